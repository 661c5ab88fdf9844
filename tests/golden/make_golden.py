#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE'S OWN CODE.

Run in the build container only (needs /root/reference):
    python tests/golden/make_golden.py

What is executed from the reference, unmodified, imported from /root/reference:
  * octree_np.encode / getDecodeFromPc / decode                      (octree_np.py)
  * pn_kit.encode_sampled_np / decode_sampled_np / binary_array_to_byte_array /
    byte_array_to_binary_array / farthest_point_sample_batch / index_points /
    normalize / denormalize / pmf_to_cdf / PointNet / MLP / SetAbstraction (pn_kit.py)
  * AE.AE / AE.ConditionalProbabilityModel                           (AE.py)
  * PPPF_AE.PPPF_AE, pppe_pcd_ae.PointCloudAE forward                 (PPPF_AE.py, pppe_pcd_ae.py)
  * eval.calc_uc (uniformity coefficient, eval.py:127-151)            (eval.py, imported with an argv that matches no file)
  * train_pppe_pcd_ae.set_model_and_loss / train_one_epoch (two iterations on CPU, scaler=None) with
    pppe_pcd_ae.RateDistortionLoss / estimate_bits_per_point_conditional (train_pppe_pcd_ae.py:171-252)
  * train.prepare_model_and_optimizer / train_one_epoch for --model AE (two iterations on CPU, scaler=None) with AE.get_loss
    (train.py:125-256, AE.py:57-70)

pn_kit.py and AE.py import pytorch3d, pyntcloud and plyfile at module import; none
of the three is installed in the image (no network).  Their NAMES are bound here
to placeholders so the imports resolve: file-I/O names raise if called; the only
third-party function the captured paths call -- pytorch3d's knn_points inside
SetAbstraction.forward (pn_kit.py:190) -- is supplied by the oracle's definition
(oracle/pcc_oracle.c: orc_knn).  Fixtures that depend on it are therefore pinned
for the reference's own arithmetic (convs, centring, max-pool) but PARITY UNPINNED
for pytorch3d's tie order; the test docstrings say so.

Fixtures are data only (inputs are regenerated from seeds; expected outputs are
stored); no reference source text is stored.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from oracle import cport, ref_families, ref_model, ref_train  # noqa: E402
from tests import synth  # noqa: E402


def _bind_absent_third_party():
    def absent(name):
        def f(*a, **k):
            raise RuntimeError(f"{name} is not installed in this image")
        return f

    p3d = types.ModuleType("pytorch3d")
    ops = types.ModuleType("pytorch3d.ops")
    knn = types.ModuleType("pytorch3d.ops.knn")
    loss = types.ModuleType("pytorch3d.loss")
    knn.knn_points = ref_model.knn_points          # oracle-defined (see module docstring)
    knn.knn_gather = absent("pytorch3d.knn_gather")
    knn._KNN = object
    ops.knn = knn
    ops.knn_points = knn.knn_points
    ops.knn_gather = knn.knn_gather
    knn.knn_gather = ref_families.knn_gather         # oracle-defined, as knn_points
    ops.knn_gather = knn.knn_gather
    ops.ball_query = lambda p1, p2, K, radius: ref_families.ball_query(p1, p2, K, radius)[1]   # pointnet_sa_module.py:18 uses the result as idx
    ops.sample_farthest_points = ref_families.sample_farthest_points
    loss.chamfer_distance = ref_train.chamfer_distance   # oracle-defined (documented pytorch3d semantics; PARITY UNPINNED)
    p3d.ops, p3d.loss = ops, loss
    pynt = types.ModuleType("pyntcloud")
    pynt.PyntCloud = absent("pyntcloud.PyntCloud")
    ply = types.ModuleType("plyfile")
    ply.PlyData = absent("plyfile.PlyData")
    for m in (p3d, ops, knn, loss, pynt, ply):
        sys.modules[m.__name__] = m


IPDAE_TRAIN_CFG = dict(N=2048, N0=1024, ALPHA=2, K=64, d=16, L=7, B=2, lr=5e-4, lamda=1000.0, rate_loss_enable_step=1, lr_decay=0.1,
                       lr_decay_steps=2)


def make_ipdae_train(ref_AE):
    """ipdae_train_step.npz: two iterations of the reference's OWN loop body for ``--model AE`` -- train.train_one_epoch (train.py:156-256)
    called once per iteration with a one-batch loader on CPU (scaler=None: the contextlib.nullcontext branch of :175, fp32), the models and
    the optimizer built as train.py:125-135 builds them, the criterion the reference's AE.get_loss behind a recorder that keeps the exact
    scalars it is handed.  Iteration 0 runs with lambda = 0 (global_step < rate_loss_enable_step, :218-219), iteration 1 with the rate term
    on (lambda chosen large enough that the probability model's gradients are comparable with the distortion's), and the learning rate
    decays after it (:250-254).  pytorch3d's knn_points and chamfer_distance are the oracle's definitions (module docstring)."""
    import argparse
    import tempfile
    import train as ref_script
    c = IPDAE_TRAIN_CFG
    targs = argparse.Namespace(device="cpu", model="AE", N=c["N"], N0=c["N0"], ALPHA=c["ALPHA"], K=c["K"], d=c["d"], L=c["L"],
                               S=c["N"] * c["ALPHA"] // c["K"], k=c["K"] // c["ALPHA"], lr=c["lr"], lamda=c["lamda"],
                               rate_loss_enable_step=c["rate_loss_enable_step"], lr_decay=c["lr_decay"], lr_decay_steps=c["lr_decay_steps"],
                               max_steps=100, step_window=10 ** 9, batch_size=c["B"], model_save_folder=tempfile.mkdtemp())
    ae, prob, crit, opt = ref_script.prepare_model_and_optimizer(targs)               # train.py:125-135
    ae.load_state_dict(ref_model.seeded_state_dict(ae, synth.AE_SEED, last_gain=synth.AE_LAST_GAIN))
    prob.load_state_dict(ref_model.seeded_state_dict(prob, synth.PROB_SEED, gain=synth.PROB_GAIN))
    xt = torch.from_numpy(synth.train_input(c["B"], c["N"]))

    class Recorder(torch.nn.Module):
        def __init__(self, inner):
            super().__init__()
            self.inner, self.log = inner, []

        def forward(self, pc_pred, pc_target, fbpp, λ):
            out = self.inner(pc_pred, pc_target, fbpp, λ)
            self.log.append((float(out), float(fbpp), float(λ), float(pc_pred.shape[1])))
            return out

    class Bar:
        def set_postfix(self, *a, **k): pass
        def update(self, *a): pass

    rec = Recorder(crit)
    named = [("ae." + k, v) for k, v in ae.named_parameters()] + [("prob." + k, v) for k, v in prob.named_parameters()]
    tr = {"param_names": np.array([k for k, _ in named])}
    gstep, starts = 0, []
    for it in range(2):
        torch.manual_seed(900 + it)                  # the one torch.randint draw of pn_kit.py:321 in this iteration (train.py:178)
        starts.append(torch.randint(0, c["N"], (c["B"],), dtype=torch.long).numpy())
        torch.manual_seed(900 + it)
        gstep = ref_script.train_one_epoch([(xt, 0)], ae, prob, rec, opt, None, targs, it, gstep, Bar())
        tr[f"params_{it}"] = np.concatenate([synth.sample64(v.detach().numpy()) for _, v in named])
        tr[f"grads_{it}"] = np.concatenate([synth.sample64(v.grad.numpy()) if v.grad is not None
                                             else np.full(synth.sample64(v.detach().numpy()).shape, np.nan, np.float32) for _, v in named])
        tr[f"grad_norms_{it}"] = np.array([float(v.grad.double().norm()) if v.grad is not None else np.nan for _, v in named])
        tr[f"lr_{it}"] = np.float64(opt.param_groups[0]["lr"])
    tr["scalars"] = np.array(rec.log, dtype=np.float64)          # per iteration: loss, fbpp, lambda, points in pc_pred
    tr["starts"] = np.stack(starts)
    np.savez_compressed(os.path.join(HERE, "ipdae_train_step.npz"), **tr)


def make_eval_uc():
    """7. eval.calc_uc (eval.py:127-151), the reference's own function: eval.py is a script (argparse + the evaluation
    loop run at import), so it is imported with an argv whose glob matches nothing and an output file in a scratch
    directory; open3d (absent) is a placeholder module -- calc_uc does not touch it; pytorch3d's knn_points inside it is
    the oracle's definition, as everywhere (PARITY UNPINNED for its tie order)."""
    import tempfile
    _bind_absent_third_party()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    sys.modules.setdefault("open3d", types.ModuleType("open3d"))
    tmp = tempfile.mkdtemp()
    argv, cwd = sys.argv, os.getcwd()
    sys.argv = ["eval.py", "--input_glob", os.path.join(tmp, "none", "*.ply"), "--output_file", os.path.join(tmp, "out.csv"),
                "--device", "cpu"]
    os.chdir(tmp)
    try:
        import eval as ref_eval                      # runs the (empty) evaluation loop, writes tmp/out.csv
    finally:
        sys.argv = argv
        os.chdir(cwd)
    torch.set_num_threads(1)
    uc = np.array([ref_eval.calc_uc(a, b) for a, b in synth.uc_cases()], dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "eval_uc.npz"), uc=uc)
    print("eval_uc.npz", uc)


def main():
    if "--only-eval-uc" in sys.argv:
        sys.argv.remove("--only-eval-uc")
        return make_eval_uc()
    if "--only-ipdae-train" in sys.argv:
        sys.argv.remove("--only-ipdae-train")
        _bind_absent_third_party()
        sys.path.insert(0, REF)
        import AE as ref_AE_
        torch.set_num_threads(1)
        return make_ipdae_train(ref_AE_)
    _bind_absent_third_party()
    sys.path.insert(0, REF)
    import octree_np as ref_octree
    import pn_kit as ref_pn_kit
    import AE as ref_AE

    torch.set_num_threads(1)

    # ---------------------------------------------------------------- 1. octree bit streams
    oc = {}
    cases = synth.octree_cases()
    bits_all, bits_off, dec_all, uniq_cnt = [], [0], [], []
    for pc, depth in cases:
        b = ref_octree.encode(pc, 1, depth)
        bits_all.append(b.astype(np.uint8))
        bits_off.append(bits_off[-1] + b.shape[0])
        dec_all.append(ref_octree.decode(b, 1).astype(np.float32))
        uniq_cnt.append(ref_octree.getDecodeFromPc(pc, 1, depth).shape[0])
    oc["bits"] = np.concatenate(bits_all)
    oc["bits_off"] = np.array(bits_off, dtype=np.int64)
    oc["decoded_reference"] = np.stack(dec_all)
    oc["unique_count"] = np.array(uniq_cnt, dtype=np.int32)
    # short / degenerate streams through decode
    short = synth.short_streams()
    oc["short_decoded"] = np.stack([ref_octree.decode(np.array(s, dtype=np.uint8), 1) for s in short])
    np.savez_compressed(os.path.join(HERE, "octree.npz"), **oc)

    # ---------------------------------------------------------------- 2. depth search + packing
    ds = {}
    sb, so, nb, packed, poff, unpk = [], [0], [], [], [0], []
    for pcs, N, K in synth.depth_search_cases():
        codes, total = ref_pn_kit.encode_sampled_np(pcs, scale=1, N=N, min_bpp=ref_pn_kit.OCTREE_BPP_DICT[K])
        assert len(codes) == 1
        sb.append(codes[0].astype(np.uint8))
        so.append(so[-1] + codes[0].shape[0])
        nb.append(total)
        by = ref_pn_kit.binary_array_to_byte_array(codes[0])
        packed.append(np.frombuffer(bytes(by), dtype=np.uint8))
        poff.append(poff[-1] + len(by))
        unpk.append(ref_pn_kit.byte_array_to_binary_array(by).astype(np.uint8))
    ds["bits"] = np.concatenate(sb)
    ds["bits_off"] = np.array(so, dtype=np.int64)
    ds["total_bits"] = np.array(nb, dtype=np.int64)
    ds["bytes"] = np.concatenate(packed)
    ds["bytes_off"] = np.array(poff, dtype=np.int64)
    ds["unpacked"] = np.concatenate(unpk)
    # tail cases len % 8 in 0..7
    tails = synth.pack_tail_cases()
    tb = [np.frombuffer(bytes(ref_pn_kit.binary_array_to_byte_array(t)), dtype=np.uint8) for t in tails]
    ds["tail_bytes"] = np.concatenate(tb)
    ds["tail_bytes_off"] = np.cumsum([0] + [len(t) for t in tb]).astype(np.int64)
    np.savez_compressed(os.path.join(HERE, "depth_search_pack.npz"), **ds)

    # ---------------------------------------------------------------- 3. FPS / normalize / gather / cdf
    fp = {}
    idxs, starts, norms, cents, longs = [], [], [], [], []
    for i, (pc, S) in enumerate(synth.fps_cases()):
        x = torch.from_numpy(pc).unsqueeze(0)
        torch.manual_seed(100 + i)
        idx = ref_pn_kit.farthest_point_sample_batch(x, S)      # random start (pn_kit.py:321)
        starts.append(int(idx[0, 0]))
        idxs.append(idx[0].numpy().astype(np.int64))
        xn, c, l = ref_pn_kit.normalize(x, margin=0.01)
        cents.append(c.numpy()); longs.append(float(l))
        norms.append(xn[0, ::257].numpy())                      # strided sample of the output
        back = ref_pn_kit.denormalize(xn, c, l, margin=0.01)
        fp[f"denorm_sample_{i}"] = back[0, ::257].numpy()
        g = ref_pn_kit.index_points(x, idx)
        fp[f"gather_{i}"] = g[0].numpy()
    fp["starts"] = np.array(starts, dtype=np.int64)
    for i, a in enumerate(idxs):
        fp[f"fps_idx_{i}"] = a
        fp[f"norm_sample_{i}"] = norms[i]
    fp["centers"] = np.stack(cents).astype(np.float32)
    fp["longest"] = np.array(longs, dtype=np.float32)
    pmf = synth.pmf_case()
    fp["cdf"] = ref_pn_kit.pmf_to_cdf(torch.from_numpy(pmf)).numpy()
    np.savez_compressed(os.path.join(HERE, "pnkit_float.npz"), **fp)

    # ---------------------------------------------------------------- 4. model modules
    md = {}
    K, k, d, L = synth.MODEL_CFG
    ae = ref_AE.AE(K=K, k=k, d=d, L=L).eval()
    ae.load_state_dict(ref_model.seeded_state_dict(ae, synth.AE_SEED, last_gain=synth.AE_LAST_GAIN))
    prob = ref_AE.ConditionalProbabilityModel(L, d).eval()
    prob.load_state_dict(ref_model.seeded_state_dict(prob, synth.PROB_SEED, gain=synth.PROB_GAIN))
    md["ae_keys"] = np.array(list(ae.state_dict().keys()))
    md["ae_shapes"] = np.array([str(tuple(v.shape)) for v in ae.state_dict().values()])
    md["prob_keys"] = np.array(list(prob.state_dict().keys()))
    md["prob_shapes"] = np.array([str(tuple(v.shape)) for v in prob.state_dict().values()])
    patches = torch.from_numpy(synth.patch_batch(K))               # (P,K,3)
    with torch.no_grad():
        xt = patches.transpose(1, 2).contiguous()                  # (P,3,K) as compress.py:107
        _, feat = ae.sa(xt)                                        # pn_kit.py:164-211 (oracle kNN inside)
        md["sa_feat_sample"] = feat[:, :, ::8].numpy()
        lat = ae.pn(torch.cat((xt, feat), dim=1))                  # pn_kit.py:124-144
        md["pn_latent_raw"] = lat.numpy()
        rec, latent, latent_q = ae(patches)                        # AE.py:34-55
        md["ae_latent"] = latent.numpy()
        md["ae_latent_q"] = latent_q.numpy()
        md["ae_recon"] = rec.numpy()
        # decoder from a fixed integer latent (decompress.py:97-102)
        lq = torch.from_numpy(synth.latent_case(patches.shape[0], d, L))
        lin = ae.inv_pool(lq).view(lq.shape[0], -1, ae.k)
        mlp_in = torch.cat((lin, lq.unsqueeze(-1).tile((1, 1, ae.k))), dim=1)
        md["dec_lin_sample"] = lin[:, ::17, ::5].numpy()
        md["dec_out"] = ae.inv_mlp(mlp_in).transpose(2, 1).numpy()
        centres = torch.from_numpy(synth.centres_case())           # (1,S,3)
        pm = prob(centres)                                         # AE.py:107-123
        md["pmf"] = pm.numpy()
        md["cdf"] = ref_pn_kit.pmf_to_cdf(pm).numpy()
    np.savez_compressed(os.path.join(HERE, "model.npz"), **md)
    # ---------------------------------------------------------------- 5. other model families (eval mode)
    import PPPF_AE as ref_PPPF
    import pppe_pcd_ae as ref_pppe
    fam = {}
    m = ref_PPPF.PPPF_AE(K=512, k=0, d=16, L=7).eval()
    m.load_state_dict(synth.family_tweak(ref_families.seeded_with_bn(m, synth.PPPF_SEED), "pppf"))
    fam["pppf_keys"] = np.array(list(m.state_dict().keys()))
    fam["pppf_shapes"] = np.array([str(tuple(v.shape)) for v in m.state_dict().values()])
    with torch.no_grad():
        rec, lat, q = m(torch.from_numpy(synth.pppf_input()))                  # PPPF_AE.py:128-150
    fam["pppf_recon"], fam["pppf_latent_sample"], fam["pppf_q"] = rec.numpy(), lat[:, ::16].numpy(), q.numpy()
    pm = ref_pppe.PointCloudAE(latent_dim=64, latent_bins=16, npoints=8192).eval()
    pm.load_state_dict(synth.family_tweak(ref_families.seeded_with_bn(pm, synth.PPPE_SEED), "pppe"))
    fam["pppe_keys"] = np.array(list(pm.state_dict().keys()))
    fam["pppe_shapes"] = np.array([str(tuple(v.shape)) for v in pm.state_dict().values()])
    x = torch.from_numpy(synth.pppe_input())
    B = x.shape[0]
    torch.manual_seed(77)      # the four torch.randint draws of pn_kit.py:321, in call order (:617-632, :596)
    fam["pppe_starts"] = np.stack([torch.randint(0, n, (B,), dtype=torch.long).numpy() for n in (8192, 8192, 512, 128)])
    torch.manual_seed(77)
    with torch.no_grad():
        coarse, fine, cond, y_q = pm(x)                                         # pppe_pcd_ae.py:858-877
    fam["pppe_coarse"], fam["pppe_fine_sample"] = coarse.numpy(), fine[:, ::16].numpy()
    fam["pppe_cond"], fam["pppe_yq"] = cond.numpy(), y_q[:, :, 0].numpy()
    np.savez_compressed(os.path.join(HERE, "families.npz"), **fam)

    # ---------------------------------------------------------------- 6. the pppe training step (configs[4])
    # The reference's own loop body: train_one_epoch is called once per iteration with a one-batch loader, the
    # criterion it is handed is the reference's get_loss("chamfer") behind a recorder that keeps the exact
    # scalars, scaler=None (the CPU branch, train_pppe_pcd_ae.py:221-224).  pytorch3d's chamfer_distance and
    # knn_points are the oracle's definitions (module docstring).
    import argparse
    import tempfile
    import train_pppe_pcd_ae as ref_script
    tr = {}
    Nt, Bt, lr = 2048, 2, 1e-3
    targs = argparse.Namespace(device="cpu", K=64, L=16, N=Nt, max_steps=100, warmup_steps=2, step_window=10 ** 9,
                               model_save_folder=tempfile.mkdtemp())
    ae, prob, crit = ref_script.set_model_and_loss(targs)                     # train_pppe_pcd_ae.py:43-49
    ae.load_state_dict(synth.family_tweak(ref_families.seeded_with_bn(ae, synth.PPPE_SEED), "pppe"))
    opt = torch.optim.Adam(list(ae.parameters()), lr=lr)                        # :274-276
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=100)          # :278
    xt = torch.from_numpy(synth.train_input(Bt, Nt))

    class Recorder(torch.nn.Module):
        def __init__(self, inner):
            super().__init__()
            self.inner, self.log = inner, []

        def forward(self, rec, tgt, fbpp, lam):
            out = self.inner(rec, tgt, fbpp, lam)
            self.log.append((float(out[0]), float(out[1]), float(out[2]), float(lam), float(fbpp)))
            return out

    class Bar:
        def set_postfix(self, *a, **k): pass
        def update(self, *a): pass

    rec = Recorder(crit)
    names = [k for k, _ in ae.named_parameters()]
    tr["param_names"] = np.array(names)
    gstep, starts = 0, []
    for it in range(2):
        torch.manual_seed(500 + it)                  # the four torch.randint draws of pn_kit.py:321 in this forward
        starts.append(np.stack([torch.randint(0, n, (Bt,), dtype=torch.long).numpy() for n in (Nt, Nt, 512, 128)]))
        torch.manual_seed(500 + it)
        gstep = ref_script.train_one_epoch([(xt, 0)], ae, prob, rec, opt, sched, None, targs, it, gstep, Bar())
        sd = dict(ae.named_parameters())
        tr[f"params_{it}"] = np.concatenate([synth.sample64(sd[k].detach().numpy()) for k in names])
        tr[f"grads_{it}"] = np.concatenate([synth.sample64(sd[k].grad.numpy()) if sd[k].grad is not None
                                             else np.full(synth.sample64(sd[k].detach().numpy()).shape, np.nan, np.float32) for k in names])
        tr[f"lr_{it}"] = np.float64(opt.param_groups[0]["lr"])
    tr["scalars"] = np.array(rec.log, dtype=np.float64)          # per iteration: loss, dist, rate, lambda_eff, fbpp
    tr["starts"] = np.stack(starts)
    tr["bn_running_mean_sample"] = np.concatenate([synth.sample64(v.numpy()) for k, v in ae.named_buffers() if k.endswith("running_mean")])
    tr["bn_running_var_sample"] = np.concatenate([synth.sample64(v.numpy()) for k, v in ae.named_buffers() if k.endswith("running_var")])
    np.savez_compressed(os.path.join(HERE, "train_step.npz"), **tr)

    make_ipdae_train(ref_AE)
    make_eval_uc()
    for f in ("octree.npz", "depth_search_pack.npz", "pnkit_float.npz", "model.npz", "families.npz", "train_step.npz", "ipdae_train_step.npz", "eval_uc.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
