"""The drop-in boundary as the reference's own scripts use it: the sub-modules are CALLED
(compress.py:113-121 ae.sa / ae.pn per patch; decompress.py:97-101 ae.inv_pool / ae.inv_mlp), torchac is imported by
name (compress.py:136, decompress.py:93), the ops are registered with the dispatcher (torch.ops.pccx.*).

The first test issues exactly the call sequence of the two loop bodies through compat/ and checks every product
against the CPU oracle (oracle/ref_pipeline.py): integer path bit-identical, latents 5e-5, reconstruction 2e-5 of the
cloud size -- the bars of tests/test_gpu_pipeline.py.
"""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import cport, ref_model, ref_pipeline
from pccx import synth as cloud_synth
from tests import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMPAT = os.path.join(ROOT, "point-cloud-compression_amd", "compat")
G = os.path.join(os.path.dirname(__file__), "golden")
K, k, d, L = synth.MODEL_CFG


def _compat():
    if COMPAT not in sys.path:
        sys.path.insert(0, COMPAT)
    import AE
    import pn_kit
    import torchac
    from pytorch3d.ops.knn import knn_points
    return AE, pn_kit, torchac, knn_points


def test_torch_library_ops_are_registered_with_schemas_and_fake_kernels():
    """CPU: torch.ops.pccx.* exist, carry schemas, propagate shapes on fake tensors, and have NO CPU kernel."""
    from torch._subclasses.fake_tensor import FakeTensorMode
    from pccx import torch_ops
    for name in torch_ops.OPS:
        assert hasattr(torch.ops.pccx, name), name
        assert str(getattr(torch.ops.pccx, name).default._schema).startswith(f"pccx::{name}(")
    with FakeTensorMode():
        x = torch.empty(2, 100, 3, device="cuda")
        d_, i_, n_ = torch.ops.pccx.knn_points(x[:, :5], x, 7)
        assert d_.shape == (2, 5, 7) and i_.dtype == torch.int64 and n_.shape == (2, 5, 7, 3)
        assert torch.ops.pccx.fps(x, 9, torch.empty(2, dtype=torch.int32, device="cuda")).shape == (2, 9)
        assert torch.ops.pccx.chamfer_distance(x, x)[0].shape == ()
        bits = torch.ops.pccx.octree_encode(x[:, :64], 8192, 0.25)
        assert bits[0].shape == (2, 1 + 8 * 64 * 16) and bits[3].shape == (2, 1025)
    with pytest.raises(NotImplementedError):
        torch.ops.pccx.fps(torch.zeros(1, 10, 3), 2, torch.zeros(1, dtype=torch.int32))      # no CPU fallback


@pytest.fixture(scope="module")
def nets():
    AE, _, _, _ = _compat()
    ae = AE.AE(K=K, k=k, d=d, L=L).to("cuda")                       # compress.py:61-63
    ae.load_state_dict(ref_model.seeded_state_dict(ae, synth.AE_SEED, last_gain=synth.AE_LAST_GAIN))
    ae.eval()
    prob = AE.ConditionalProbabilityModel(L, d).to("cuda")          # compress.py:65-67
    prob.load_state_dict(ref_model.seeded_state_dict(prob, synth.PROB_SEED, gain=synth.PROB_GAIN))
    prob.eval()
    oae = ref_model.AE(K, k, d, L).eval()
    oae.load_state_dict({k_: v.cpu() for k_, v in ae.state_dict().items()})
    oprob = ref_model.ConditionalProbabilityModel(L, d).eval()
    oprob.load_state_dict({k_: v.cpu() for k_, v in prob.state_dict().items()})
    return ae, prob, oae, oprob


@pytest.fixture(params=["f32", "bf16x3", "f16x2"])
def matmul_mode(request):
    import pccx
    old = pccx.DEFAULT_MATMUL
    pccx.DEFAULT_MATMUL = request.param
    yield request.param
    pccx.DEFAULT_MATMUL = old


@pytest.mark.gpu
def test_reference_loop_bodies_call_for_call_through_compat(nets, matmul_mode):
    """compress.py:82-152 then decompress.py:77-116, statement by statement, on the compat modules."""
    AE, pn_kit, torchac, knn_points = _compat()
    ae, prob, oae, oprob = nets
    device, ALPHA, N0, B = "cuda", 2, 1024, 1
    pc_np = cloud_synth.cad_cloud(77, 8192) * np.float32(1.7) + np.float32(0.3)

    # ---------------- compress.py:82-152
    with torch.no_grad():
        pc = torch.Tensor(pc_np).to(device)
        pc = pc.unsqueeze(0)
        pc, center, longest = pn_kit.normalize(pc, margin=0.01)
        N = pc.shape[1]
        S = (int)(N * ALPHA // K)
        torch.manual_seed(5)
        start = int(torch.randint(0, N, (B,), dtype=torch.long)[0])            # the draw of pn_kit.py:321
        torch.manual_seed(5)
        sampled_xyz = pn_kit.index_points(pc, pn_kit.farthest_point_sample_batch(pc, S))
        octree_codes, sampled_bits = pn_kit.encode_sampled_np(sampled_xyz.detach().cpu().numpy(), scale=1, N=N,
                                                               min_bpp=pn_kit.OCTREE_BPP_DICT[K])
        rec_sampled_xyz = pn_kit.decode_sampled_np(octree_codes, scale=1)
        rec_sampled_xyz = torch.Tensor(rec_sampled_xyz).to(device)
        assert rec_sampled_xyz.shape == sampled_xyz.shape

        dist, group_idx, grouped_xyz = knn_points(rec_sampled_xyz, pc, K=K, return_nn=True)      # KNN_Patching, :70-74
        grouped_xyz -= rec_sampled_xyz.view(B, S, 1, 3)
        x_patches = grouped_xyz.view(B * S, K, 3)
        x_patches = x_patches.transpose(1, 2)
        x_patches = x_patches * ((N / N0) ** (1 / 3))

        patch_features = []
        for j in range(S):                                                     # :112-116, one patch per call
            _, patch_feature = ae.sa(x_patches[j].view(1, 3, K))
            patch_features.append(patch_feature.cpu())
        patch_features = torch.cat(patch_features)
        latent = []
        for j in range(S):                                                     # :119-122
            latent.append(ae.pn(torch.cat((x_patches[j].unsqueeze(0), patch_features[j].to(device).unsqueeze(0)), dim=1)).cpu())
        latent = torch.cat(latent)

        spread = ae.L - 0.2
        latent = torch.sigmoid(latent) * spread - spread / 2
        latent_quantized = ae.quantize(latent)

        pmf = prob(rec_sampled_xyz)
        cdf = pn_kit.pmf_to_cdf(pmf).cpu()
        n_latent_quantized = latent_quantized.view(B, S, -1).to(torch.int16).cpu() + L // 2
        p_stream = torchac.encode_float_cdf(cdf, n_latent_quantized, check_input_bounds=True)
        s_stream = pn_kit.binary_array_to_byte_array(octree_codes[0])
        arr = np.zeros((4))
        arr[:3] = center.detach().cpu().numpy().flatten()
        arr[3] = longest.detach().cpu().numpy()
        c_stream = arr.astype(np.float32).tobytes()

    o, _ = ref_pipeline.compress_one(pc_np, oae, oprob, start, K=K)
    assert bytes(s_stream) == o["s"] and c_stream == o["c"]                    # .s.bin / .c.bin bit-identical
    assert np.array_equal(group_idx[0].cpu().numpy(), o["knn_idx"])
    np.testing.assert_allclose(latent.numpy(), o["latent"], rtol=0, atol=5e-5)
    q = latent_quantized.numpy()
    bad = q != o["latent_q"]
    assert (np.abs(o["latent"][bad] - np.floor(o["latent"][bad]) - 0.5) < 1e-4).all()
    assert isinstance(p_stream, bytes) and len(p_stream) > 0
    # .p.bin byte-identical with the oracle coder, unconditionally: the oracle's range encoder on the integer CDF torchac's
    # conversion gives for THIS float CDF and on THESE symbols (with the oracle's CDF and symbols that is o["p"] itself)
    ci = ref_model.cdf_float_to_int(cdf).reshape(-1, L + 1)
    assert p_stream == cport.range_encode(ci, n_latent_quantized.numpy().reshape(-1))
    if not bad.any() and np.array_equal(ci, o["cdf_int"]):
        assert p_stream == o["p"]

    # ---------------- decompress.py:77-116
    with torch.no_grad():
        octree_code = pn_kit.byte_array_to_binary_array(s_stream)
        rec = pn_kit.decode_sampled_np([octree_code], scale=1)
        rec = torch.Tensor(rec)
        S2 = rec.shape[1]
        pmf = prob(rec.to(device))
        cdf = pn_kit.pmf_to_cdf(pmf).cpu()
        sym = torchac.decode_float_cdf(cdf, p_stream)
        assert sym.dtype == torch.int16 and torch.equal(sym, n_latent_quantized)   # lossless
        latent2 = (sym - ae.L // 2).float().view(B * S2, -1)
        latent2 = latent2.to(device)
        linear_output = ae.inv_pool(latent2)
        linear_output = linear_output.view(B * S2, -1, ae.k)
        latent_q2 = latent2.unsqueeze(-1).tile((1, 1, ae.k))
        mlp_input = torch.cat((linear_output, latent_q2), dim=1)
        new_xyz = ae.inv_mlp(mlp_input)
        patches = new_xyz.transpose(2, 1)
        kk = patches.shape[1]
        N2 = S2 * kk
        patches = patches / ((N2 / N0) ** (1 / 3))
        out = (patches.cpu().view(B, S2, -1, 3) + rec.cpu().view(B, S2, 1, 3)).reshape(B, -1, 3)
        a = np.frombuffer(c_stream, dtype=np.float32)
        out = pn_kit.denormalize(out, torch.Tensor(a[:3].copy()).reshape(1, 3), torch.Tensor([a[3]]), margin=0.01)
    want, _ = ref_pipeline.decompress_one(bytes(s_stream), p_stream, c_stream, oae, oprob, latent_q_override=q.copy())
    assert out.shape == (1, 8192, 3)
    np.testing.assert_allclose(out[0].numpy(), want, rtol=0, atol=2e-5 * float(a[3]))
    # and the batched product path gives the same files for this cloud
    from pccx import codec
    comp = codec.Codec(ae, prob, K=K).compress(torch.from_numpy(pc_np)[None].cuda(), [start], keep_extras=True)
    s2, p2, c2 = comp.files(0)
    assert s2 == bytes(s_stream) and c2 == c_stream
    q2 = comp.extras["latent_q"].cpu().numpy()
    ci2 = comp.extras["cdf_int"][0].cpu().numpy().reshape(-1, L + 1)
    assert p2 == cport.range_encode(ci2, (q2.reshape(-1) + L // 2).astype(np.int16))          # unconditional: coder byte-identity
    if np.array_equal(q2, q) and np.array_equal(ci2, ci):
        assert p2 == p_stream


@pytest.mark.gpu
def test_generic_submodule_forwards_vs_oracle(matmul_mode):
    """Sub-modules outside the fused shapes run the generic HIP layers: a PointNet like prob.model_pn, an MLP, a
    SetAbstraction that samples (npoint < N, explicit FPS start), inv_pool (Linear stack) -- against torch CPU."""
    AE, pn_kit, _, _ = _compat()
    rng = np.random.default_rng(3)
    x = torch.from_numpy(rng.normal(size=(3, 3, 200)).astype(np.float32))
    pn = pn_kit.PointNet(3, [64, 128, 256], [True, True, True], False)
    opn = ref_model.PointNet(3, [64, 128, 256], [True, True, True])
    opn.load_state_dict(pn.state_dict())
    with torch.no_grad():
        np.testing.assert_allclose(pn(x.cuda()).cpu().numpy(), opn(x).numpy(), rtol=1e-4, atol=1e-5)
    mlp = pn_kit.MLP(3, [32, 16, 3], [True, True, False], False)
    omlp = ref_model.MLP(3, [32, 16, 3], [True, True, False])
    omlp.load_state_dict(mlp.state_dict())
    with torch.no_grad():
        got = mlp(x.cuda())
        assert got.shape == (3, 3, 200)
        np.testing.assert_allclose(got.cpu().numpy(), omlp(x).numpy(), rtol=1e-4, atol=1e-5)
    sa = pn_kit.SetAbstraction(npoint=50, K=8, in_channel=0, mlp=[16, 32, 64])
    osa = ref_model.SetAbstraction(npoint=50, K=8, in_channel=0, mlp=[16, 32, 64])
    osa.load_state_dict(sa.state_dict())
    with torch.no_grad():
        nx, nf = sa(x.cuda(), start_idx=[0, 0, 0])        # the oracle's SetAbstraction starts FPS at index 0
        wx, wf = osa(x)
    assert nx.shape == (3, 3, 50) and nf.shape == (3, 64, 50)
    assert np.array_equal(nx.cpu().numpy(), wx.numpy())
    np.testing.assert_allclose(nf.cpu().numpy(), wf.numpy(), rtol=1e-4, atol=1e-5)
    ae = AE.AE(K=K, k=k, d=d, L=L)
    lat = torch.from_numpy(rng.integers(-3, 4, size=(5, d)).astype(np.float32))
    with torch.no_grad():
        want = torch.nn.Sequential(*[m for m in ae.inv_pool])(lat)          # the same Linear / ReLU modules on torch CPU
        got = ae.inv_pool(lat.cuda())
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=1e-4, atol=1e-5)


@pytest.mark.gpu
def test_ste_quantize_rounds_and_passes_the_gradient_through():
    AE, _, _, _ = _compat()
    x = torch.tensor([[-2.5, -1.5, -0.5, 0.5, 1.5, 2.5, 0.49999, 3.2]], device="cuda", requires_grad=True)
    y = AE.STEQuantize.apply(x)
    assert torch.equal(y.detach().cpu(), torch.round(x.detach().cpu()))          # half to even, as torch.round
    w = torch.arange(8, device="cuda", dtype=torch.float32)[None]
    (y * w).sum().backward()
    assert torch.equal(x.grad, w)                                                # AE.py:83-85: identity
    ae = AE.AE(K=K, k=k, d=d, L=L)
    z = torch.tensor([[0.4, 1.6]], device="cuda", requires_grad=True)
    ae.quantize(z).sum().backward()
    assert torch.equal(z.grad, torch.ones_like(z))
    crit = AE.get_loss()
    a = torch.rand(2, 300, 3, device="cuda", requires_grad=True)
    b = torch.rand(2, 500, 3, device="cuda")
    loss = crit(a, b, torch.tensor(0.7, device="cuda"), 0.01)
    loss.backward()
    assert a.grad is not None and torch.isfinite(a.grad).all() and float(a.grad.abs().sum()) > 0


@pytest.mark.gpu
def test_torch_ops_agree_with_the_direct_path_and_chamfer_autograd():
    from pccx import ops, torch_ops  # noqa: F401
    pc = torch.from_numpy(cloud_synth.cad_batch(9, 2, 2048)).cuda()
    st = torch.tensor([1, 77], dtype=torch.int32, device="cuda")
    assert torch.equal(torch.ops.pccx.fps(pc, 32, st), ops.farthest_point_sample_batch(pc, 32, st))
    d0, i0, n0 = torch.ops.pccx.knn_points(pc[:, :10].contiguous(), pc, 16, 0.0)
    r = ops.knn_points(pc[:, :10].contiguous(), pc, 16)
    assert torch.equal(i0, r.idx) and torch.equal(d0, r.dists) and torch.equal(n0, r.knn)
    bits = torch.ops.pccx.octree_encode(pc[:, :64].contiguous(), 2048, 0.25)
    want = ops.octree_encode(pc[:, :64].contiguous(), 2048, 0.25)
    assert torch.equal(bits[4], want["nbytes"]) and torch.equal(bits[1], want["nbits"])
    for b in range(2):                                   # bytes past nbytes are scratch
        n = int(bits[4][b])
        assert torch.equal(bits[3][b, :n], want["bytes"][b, :n])
    x = pc[:, :300].clone().requires_grad_(True)
    y = pc[:, 300:900].clone().requires_grad_(True)
    loss = torch.ops.pccx.chamfer_distance(x, y)[0]
    loss.backward()
    x2, y2 = x.detach().clone().requires_grad_(True), y.detach().clone().requires_grad_(True)
    l2, _ = ops.chamfer_distance(x2, y2)
    l2.backward()
    assert torch.equal(loss.detach(), l2.detach())
    # the backward scatters with float atomics (several x may share a nearest y): summation order is not fixed
    torch.testing.assert_close(x.grad, x2.grad, rtol=1e-5, atol=1e-10)
    torch.testing.assert_close(y.grad, y2.grad, rtol=1e-5, atol=1e-10)
    z = torch.tensor([0.5, 1.5, -0.2], device="cuda", requires_grad=True)
    torch.ops.pccx.ste_round(z).sum().backward()
    assert torch.equal(z.grad, torch.ones(3, device="cuda"))


@pytest.mark.gpu
def test_uniformity_coefficient_vs_the_references_own_function():
    """cli/eval.calc_uc against eval.calc_uc ITSELF (tests/golden/eval_uc.npz, captured from /root/reference/eval.py by
    make_golden.py) and the oracle restatement.  Tolerance: the reference's torch.cdist forms |a|^2+|b|^2-2ab, whose
    rounding is ~1e-8 absolute on a squared distance; the kernel sums (a-b)^2.  That is 1e-5 relative on ordinary
    clouds (cases 0, 1: rtol 1e-3) and up to a percent when most nearest-neighbour distances are 3e-4 (case 2: duplicated
    points, squared distances ~1e-7; rtol 3e-2) -- the float64 value sits between the two."""
    sys.path.insert(0, os.path.join(ROOT, "point-cloud-compression_amd", "cli"))
    import importlib
    argv = sys.argv
    sys.argv = ["eval.py"]
    try:
        ev = importlib.import_module("eval")
    finally:
        sys.argv = argv
    want = np.load(os.path.join(G, "eval_uc.npz"))["uc"]
    for i, ((a, b), w, tol) in enumerate(zip(synth.uc_cases(), want, (1e-3, 1e-3, 3e-2))):
        assert abs(ref_pipeline.calc_uc(a, b) - w) <= 1e-6 * w                   # oracle == the reference's function
        got = ev.calc_uc(torch.from_numpy(a)[None].cuda(), torch.from_numpy(b)[None].cuda())
        assert abs(got - w) <= tol * w, (i, got, w)


@pytest.mark.gpu
def test_range_coder_overflow_is_reported_not_swallowed():
    from pccx import _lib, codec, models
    rng = np.random.default_rng(0)
    nsym, Lx = 1024, 7
    pmf = rng.random((1, nsym, Lx)).astype(np.float32)
    pmf /= pmf.sum(-1, keepdims=True)
    cdf = np.concatenate([np.zeros((1, nsym, 1), np.float32), np.cumsum(pmf, -1)], -1).clip(max=1)
    ci = torch.from_numpy(ref_model.cdf_float_to_int(torch.from_numpy(cdf))).cuda()
    q = torch.from_numpy(rng.integers(-3, 4, size=(1, nsym)).astype(np.float32)).cuda()
    by, nb = models.range_encode(ci, q, Lx, cap=64)                               # far too small
    assert int(nb[0]) < 0
    comp = codec.Compressed(torch.zeros(1, 8, dtype=torch.uint8, device="cuda"), torch.ones(1, dtype=torch.int32, device="cuda"),
                            by, nb, torch.zeros(1, 4, device="cuda"), 8192)
    with pytest.raises(_lib.PccxError):
        comp.files(0)
