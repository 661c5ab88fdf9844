"""Pins the oracle (oracle/) against fixtures captured from the reference's own code.

CPU only.  Fixtures were produced by tests/golden/make_golden.py running
/root/reference's octree_np.py / pn_kit.py / AE.py in the build container.
Integer results are compared bit-for-bit; float results with the tolerance stated
at each assert (the oracle and the reference both run torch CPU fp32, so they
normally agree to the last bit).
"""
import os

import numpy as np
import pytest
import torch

from oracle import cport, ref_model
from tests import synth

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def oc():
    return np.load(os.path.join(G, "octree.npz"))


def test_octree_encode_bits_match_reference(oc):
    cases = synth.octree_cases()
    off = oc["bits_off"]
    for i, (pc, depth) in enumerate(cases):
        want = oc["bits"][off[i]:off[i + 1]]
        got = cport.octree_encode(pc, 1, depth)
        assert got.shape == want.shape and np.array_equal(got, want), f"case {i}"
        assert cport.get_decode_from_pc(pc, 1, depth).shape[0] == oc["unique_count"][i]


def test_octree_decode_reference_mode_matches_reference(oc):
    cases = synth.octree_cases()
    off = oc["bits_off"]
    for i in range(len(cases)):
        got, _ = cport.octree_decode_reference(oc["bits"][off[i]:off[i + 1]], 1)
        assert np.array_equal(got, oc["decoded_reference"][i]), f"case {i}"
    for s, want in zip(synth.short_streams(), oc["short_decoded"]):
        got, _ = cport.octree_decode_reference(np.array(s, dtype=np.uint8), 1)
        assert np.array_equal(got, want), f"stream {s}"


def test_octree_full_decode_inverts_encode():
    """'full' mode is the build's extension (not in the reference): it must invert encode."""
    for pc, depth in synth.octree_cases():
        bits = cport.octree_encode(pc, 1, depth)
        cells = cport.get_decode_from_pc(pc, 1, depth)
        inside = cells[np.all((cells >= 0) & (cells <= 1), axis=1)]
        got, d = cport.octree_decode_full(bits, 1)
        if inside.shape[0] == 0:
            assert got.shape[0] == 0
            continue
        assert d == depth
        assert np.array_equal(np.unique(got, axis=0), inside)


def test_depth_search_and_packing_match_reference():
    ds = np.load(os.path.join(G, "depth_search_pack.npz"))
    bo, yo = ds["bits_off"], ds["bytes_off"]
    for i, (pcs, N, K) in enumerate(synth.depth_search_cases()):
        codes, total = cport.encode_sampled_np(pcs, 1, N, ref_model.OCTREE_BPP_DICT[K])
        assert total == ds["total_bits"][i]
        assert np.array_equal(codes[0], ds["bits"][bo[i]:bo[i + 1]]), f"case {i}"
        by = cport.pack_bits(codes[0])
        assert bytes(by) == ds["bytes"][yo[i]:yo[i + 1]].tobytes()
        un = cport.unpack_bits(by)
        assert np.array_equal(un, ds["unpacked"][8 * yo[i]:8 * yo[i + 1]])
    to = ds["tail_bytes_off"]
    for j, t in enumerate(synth.pack_tail_cases()):
        assert bytes(cport.pack_bits(t)) == ds["tail_bytes"][to[j]:to[j + 1]].tobytes()


def test_fps_normalize_gather_cdf_match_reference():
    fp = np.load(os.path.join(G, "pnkit_float.npz"))
    for i, (pc, S) in enumerate(synth.fps_cases()):
        idx = cport.fps(pc, S, int(fp["starts"][i]))
        assert np.array_equal(idx, fp[f"fps_idx_{i}"]), f"fps case {i}"          # exact index equality
        x = torch.from_numpy(pc).unsqueeze(0)
        xn, c, l = ref_model.normalize(x)
        assert np.array_equal(c.numpy(), fp["centers"][i]) and float(l) == fp["longest"][i]
        assert np.array_equal(xn[0, ::257].numpy(), fp[f"norm_sample_{i}"])       # bit-exact fp32
        back = ref_model.denormalize(xn, c, l)
        assert np.array_equal(back[0, ::257].numpy(), fp[f"denorm_sample_{i}"])
        g = ref_model.index_points(x, torch.from_numpy(idx)[None])
        assert np.array_equal(g[0].numpy(), fp[f"gather_{i}"])
    cdf = ref_model.pmf_to_cdf(torch.from_numpy(synth.pmf_case())).numpy()
    assert np.array_equal(cdf, fp["cdf"])


@pytest.fixture(scope="module")
def models():
    K, k, d, L = synth.MODEL_CFG
    ae = ref_model.AE(K, k, d, L).eval()
    ae.load_state_dict(ref_model.seeded_state_dict(ae, synth.AE_SEED, last_gain=synth.AE_LAST_GAIN))
    prob = ref_model.ConditionalProbabilityModel(L, d).eval()
    prob.load_state_dict(ref_model.seeded_state_dict(prob, synth.PROB_SEED, gain=synth.PROB_GAIN))
    return ae, prob


def test_state_dict_keys_match_reference(models):
    md = np.load(os.path.join(G, "model.npz"))
    ae, prob = models
    assert list(ae.state_dict().keys()) == list(md["ae_keys"])
    assert [str(tuple(v.shape)) for v in ae.state_dict().values()] == list(md["ae_shapes"])
    assert list(prob.state_dict().keys()) == list(md["prob_keys"])
    assert [str(tuple(v.shape)) for v in prob.state_dict().values()] == list(md["prob_shapes"])


def test_model_modules_match_reference(models):
    """SA output depends on pytorch3d's knn_points, which the fixture generator had to
    take from the oracle's definition (pytorch3d is absent): that part is PARITY
    UNPINNED for tie order; convs / centring / max-pool / decoder / prob model are the
    reference's own arithmetic.  Tolerance 1e-5 abs."""
    md = np.load(os.path.join(G, "model.npz"))
    ae, prob = models
    K, k, d, L = synth.MODEL_CFG
    torch.set_num_threads(1)
    patches = torch.from_numpy(synth.patch_batch(K))
    with torch.no_grad():
        xt = patches.transpose(1, 2).contiguous()
        _, feat = ae.sa(xt)
        np.testing.assert_allclose(feat[:, :, ::8].numpy(), md["sa_feat_sample"], atol=1e-5, rtol=0)
        lat = ae.pn(torch.cat((xt, feat), dim=1))
        np.testing.assert_allclose(lat.numpy(), md["pn_latent_raw"], atol=1e-4, rtol=1e-5)
        rec, latent, q = ae(patches)
        np.testing.assert_allclose(latent.numpy(), md["ae_latent"], atol=1e-5, rtol=0)
        assert np.array_equal(q.numpy(), md["ae_latent_q"])
        np.testing.assert_allclose(rec.numpy(), md["ae_recon"], atol=1e-5, rtol=0)
        lq = torch.from_numpy(synth.latent_case(patches.shape[0], d, L))
        np.testing.assert_allclose(ae.decode(lq).numpy(), md["dec_out"], atol=1e-5, rtol=0)
        pm = prob(torch.from_numpy(synth.centres_case()))
        np.testing.assert_allclose(pm.numpy(), md["pmf"], atol=1e-6, rtol=0)
        np.testing.assert_allclose(ref_model.pmf_to_cdf(pm).numpy(), md["cdf"], atol=1e-6, rtol=0)


def test_range_coder_round_trip_and_size():
    """torchac is absent: byte layout PARITY UNPINNED.  Pinned properties:
    lossless round trip; size within 1 % (+ 2 bytes) of sum(-log2 p)."""
    pmf = torch.from_numpy(synth.pmf_case())
    cdf = ref_model.pmf_to_cdf(pmf)
    ci = ref_model.cdf_float_to_int(cdf).reshape(-1, 8)
    rng = np.random.default_rng(3)
    p = pmf.reshape(-1, 7).numpy().astype(np.float64)
    sym = np.array([rng.choice(7, p=row / row.sum()) for row in p], dtype=np.int16)
    bs = cport.range_encode(ci, sym)
    back = cport.range_decode(ci, bs)
    assert np.array_equal(back, sym)
    ideal = -np.log2(p[np.arange(p.shape[0]), sym]).sum() / 8
    assert len(bs) <= ideal * 1.01 + 2 and len(bs) >= ideal * 0.99 - 2
    # degenerate: all-same symbol, and extreme symbols
    for s in (0, 6):
        sym2 = np.full(p.shape[0], s, dtype=np.int16)
        assert np.array_equal(cport.range_decode(ci, cport.range_encode(ci, sym2)), sym2)


def test_training_step_oracle_matches_reference_run():
    """oracle/ref_train.py + oracle/ref_families.py against two iterations of the reference's own
    train_one_epoch (tests/golden/train_step.npz: get_loss("chamfer"), lambda warm-up, clip_grad_norm_ 1.0,
    Adam, cosine LR; FPS start draws replayed from the fixture).  pytorch3d's chamfer_distance / knn_points
    are the oracle's definitions in both runs (parity unpinned for those two)."""
    import torch
    from oracle import ref_families as rf, ref_train
    g = np.load(os.path.join(G, "train_step.npz"))
    names = list(g["param_names"])
    o = rf.PointCloudAE(64, 16, 2048)
    o.load_state_dict(synth.family_tweak(rf.seeded_with_bn(o, synth.PPPE_SEED), "pppe"))
    assert [k for k, _ in o.named_parameters()] == names
    opt = torch.optim.Adam(o.parameters(), lr=1e-3)
    x = torch.from_numpy(synth.train_input(2, 2048))
    torch.set_num_threads(8)
    for it in range(2):
        st = g["starts"][it]
        if it == 1:
            opt.param_groups[0]["lr"] = float(g["lr_0"])                   # scheduler.step() after iteration 0
        loss, dist, rate = ref_train.train_step(o, opt, x, [[st[0], st[1]], st[2], st[3]], lam=float(g["scalars"][it, 3]),
                                                loss_type="chamfer")
        want = g["scalars"][it]
        tol = 1e-5 if it == 0 else 3e-3          # first-step Adam is g/|g|: entries with |g| near eps move by up to 2*lr either way
        assert abs(loss - want[0]) <= tol * abs(want[0]), (it, loss, want)
        assert abs(dist - want[1]) <= tol * abs(want[1]), (it, dist, want)
        assert abs(rate - want[2]) <= (1e-5 if it == 0 else 1e-2) * abs(want[2]), (it, rate, want)
        sd = dict(o.named_parameters())
        got_p = np.concatenate([synth.sample64(sd[k].detach().numpy()) for k in names])
        got_g = np.concatenate([synth.sample64(sd[k].grad.numpy()) if sd[k].grad is not None
                                else np.full(synth.sample64(sd[k].detach().numpy()).shape, np.nan, np.float32) for k in names])
        wp, wg = g[f"params_{it}"], g[f"grads_{it}"]
        assert np.array_equal(np.isnan(got_g), np.isnan(wg))               # the same parameters receive no gradient
        m = ~np.isnan(wg)
        if it == 0:
            assert np.abs(got_g[m] - wg[m]).max() <= 2e-3 * np.abs(wg[m]).max()   # restated layers sum in another order
        d = np.abs(got_p - wp)
        assert d.max() <= 2.2e-3 * (it + 1) and np.median(d) <= 1e-6 * (1 if it == 0 else 100), (it, d.max(), np.median(d))
