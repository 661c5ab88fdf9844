"""Other model families: PPPF_AE and the pppe PointCloudAE forward.

CPU tests pin the oracle restatement (oracle/ref_families.py) against fixtures captured from the
reference's own PPPF_AE.py / pointnet_sa_module.py / pppe_pcd_ae.py (eval mode; pytorch3d's ops are the
oracle's definitions there, so selection tie order is PARITY UNPINNED).  GPU tests compare the HIP
path (pccx.families, through the C ABI) with both.  Tolerance: activations agree to ~1e-5 relative
(fp32 fmaf chains vs oneDNN blocked sums, BatchNorm folded into the weights); symbols must be equal
unless the pre-rounding value is within 1e-3 of a rounding boundary.
"""
import os

import numpy as np
import pytest
import torch

from oracle import ref_families as rf
from tests import synth

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def fam():
    return np.load(os.path.join(G, "families.npz"))


@pytest.fixture(scope="module")
def oracle_nets():
    m = rf.PPPF_AE(512, 0, 16, 7).eval()
    m.load_state_dict(synth.family_tweak(rf.seeded_with_bn(m, synth.PPPF_SEED), "pppf"))
    p = rf.PointCloudAE(64, 16, 8192).eval()
    p.load_state_dict(synth.family_tweak(rf.seeded_with_bn(p, synth.PPPE_SEED), "pppe"))
    return m, p


def _starts(fam):
    s = fam["pppe_starts"]
    return [[s[0], s[1]], s[2], s[3]]


def test_oracle_families_match_reference(fam, oracle_nets):
    m, p = oracle_nets
    assert list(m.state_dict().keys()) == list(fam["pppf_keys"])
    assert [str(tuple(v.shape)) for v in m.state_dict().values()] == list(fam["pppf_shapes"])
    assert list(p.state_dict().keys()) == list(fam["pppe_keys"])
    assert [str(tuple(v.shape)) for v in p.state_dict().values()] == list(fam["pppe_shapes"])
    torch.set_num_threads(4)
    with torch.no_grad():
        rec, lat, q, _ = m(torch.from_numpy(synth.pppf_input()))
        coarse, fine, cond, yq, _ = p(torch.from_numpy(synth.pppe_input()), _starts(fam))
    np.testing.assert_allclose(lat[:, ::16].numpy(), fam["pppf_latent_sample"], atol=1e-5, rtol=0)
    assert np.array_equal(q.numpy(), fam["pppf_q"])
    np.testing.assert_allclose(rec.numpy(), fam["pppf_recon"], atol=1e-5, rtol=0)
    np.testing.assert_allclose(cond.numpy(), fam["pppe_cond"], atol=1e-5, rtol=1e-5)
    assert np.array_equal(yq.numpy(), fam["pppe_yq"])
    np.testing.assert_allclose(coarse.numpy(), fam["pppe_coarse"], atol=1e-5, rtol=0)
    np.testing.assert_allclose(fine[:, ::16].numpy(), fam["pppe_fine_sample"], atol=1e-5, rtol=0)


@pytest.fixture(params=["f32", "bf16x3"])
def matmul_mode(request):
    """Both arithmetic modes of the generic layers (pccx_linear / pccx_linear_b3), same tolerances."""
    import pccx
    old = pccx.DEFAULT_MATMUL
    pccx.DEFAULT_MATMUL = request.param
    yield request.param
    pccx.DEFAULT_MATMUL = old


def _near_boundary(pre, tol):
    return np.abs(pre - np.floor(pre) - 0.5) < tol


@pytest.mark.gpu
def test_pppf_ae_gpu_matches_reference_fixture_and_oracle(fam, oracle_nets, matmul_mode):
    from pccx import families
    m, _ = oracle_nets
    g = families.PPPF_AE(512, 0, 16, 7)
    assert list(g.state_dict().keys()) == list(fam["pppf_keys"])
    g.load_state_dict(m.state_dict())
    x = synth.pppf_input()
    rec, lat, q = g(torch.from_numpy(x).cuda())
    with torch.no_grad():
        orec, olat, oq, oz = m(torch.from_numpy(x))
    np.testing.assert_allclose(lat[:, ::16].cpu().numpy(), fam["pppf_latent_sample"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(lat.cpu().numpy(), olat.numpy(), atol=2e-5, rtol=0)
    bad = q.cpu().numpy() != fam["pppf_q"]
    assert _near_boundary(oz.numpy()[bad], 1e-3).all()
    if not bad.any():
        np.testing.assert_allclose(rec.cpu().numpy(), fam["pppf_recon"], atol=5e-5, rtol=1e-4)
    # a batch with ragged ball-query neighbourhoods (sparse cloud -> -1 padding) against the oracle
    rng = np.random.default_rng(2)
    xs = (rng.random((3, 512, 3)) * 1.6).astype(np.float32)
    rec2, lat2, q2 = g(torch.from_numpy(xs).cuda())
    with torch.no_grad():
        orec2, olat2, oq2, oz2 = m(torch.from_numpy(xs))
    np.testing.assert_allclose(lat2.cpu().numpy(), olat2.numpy(), atol=2e-5, rtol=0)
    bad2 = q2.cpu().numpy() != oq2.numpy()
    assert _near_boundary(oz2.numpy()[bad2], 1e-3).all()


@pytest.mark.gpu
def test_pppf_ae_f16x2_stacks_match_the_fixture_the_oracle_and_bf16x3(fam, oracle_nets):
    """The planes stacks of PPPF_AE.forward in the f16x2 arithmetic (csrc/planes.hip <2>: two fp16 pieces per operand, three products per
    fp32 product, exact power-of-two scales from per-stack interval bounds and a dynamic input normalisation from the data): the SAME
    bars as the bf16x3 / f32 runs of test_pppf_ae_gpu_matches_reference_fixture_and_oracle (reference fixture 2e-5 on the latent, symbols
    equal except within 1e-3 of a rounding boundary, reconstruction 5e-5), agreement with the bf16x3 run at the level of an fp32
    summation reorder, and inputs far outside the unit cube (x 64, x 1/64): the dynamic scale keeps every operand in fp16's range
    (no inf / nan) and the results follow the oracle at the tolerance scaled with the data."""
    import pccx
    from pccx import families
    m, _ = oracle_nets
    g = families.PPPF_AE(512, 0, 16, 7)
    g.load_state_dict(m.state_dict())
    x = synth.pppf_input()
    old = pccx.DEFAULT_MATMUL
    try:
        pccx.DEFAULT_MATMUL = "bf16x3"
        rec_b, lat_b, q_b = g(torch.from_numpy(x).cuda())
        pccx.DEFAULT_MATMUL = "f16x2"
        rec, lat, q = g(torch.from_numpy(x).cuda())
        assert "h2" in g._packed, "the f16x2 stacks did not run"
        with torch.no_grad():
            orec, olat, oq, oz = m(torch.from_numpy(x))
        np.testing.assert_allclose(lat[:, ::16].cpu().numpy(), fam["pppf_latent_sample"], atol=2e-5, rtol=0)
        np.testing.assert_allclose(lat.cpu().numpy(), olat.numpy(), atol=2e-5, rtol=0)
        np.testing.assert_allclose(lat.cpu().numpy(), lat_b.cpu().numpy(), atol=1e-5, rtol=0)
        bad = q.cpu().numpy() != fam["pppf_q"]
        assert _near_boundary(oz.numpy()[bad], 1e-3).all()
        if not bad.any():
            np.testing.assert_allclose(rec.cpu().numpy(), fam["pppf_recon"], atol=5e-5, rtol=1e-4)
            np.testing.assert_allclose(rec.cpu().numpy(), rec_b.cpu().numpy(), atol=2e-5, rtol=1e-4)
        # the dynamic normalisation read back: s <= 1 is a power of two with max|xyz| s <= 1, and 1 / s beside it
        dyn = g._packed["h2"]["dyn"].cpu().numpy()
        s0 = float(dyn[0])
        assert s0 <= 1.0 and np.log2(s0) == np.round(np.log2(s0)) and float(np.abs(x).max()) * s0 <= 1.0 + 1e-6 and dyn[1] == 1.0 / s0
        assert float(np.abs(x).max()) * s0 >= 0.5 or s0 == 1.0          # ... and no smaller than it has to be
        # call-to-call reproducibility on RECYCLED memory (the first call runs on fresh, zero-filled pages): the kernels take their B
        # operand through loads issued from inline assembly into rotating register sets; a register of a load still in flight handed to
        # something else shows up as results that change from call to call (it did: csrc/planes.hip pg_drain_loads)
        xc = torch.from_numpy(x).cuda()
        first = [t.clone() for t in g(xc)]
        for _ in range(5):
            junk = torch.full((64 << 20,), float("nan"), device="cuda")          # dirty the allocator's free blocks
            del junk
            again = g(xc)
            assert all(torch.equal(a_, b_) for a_, b_ in zip(first, again))
        # ragged neighbourhoods and inputs outside the unit cube
        rng = np.random.default_rng(2)
        for mul, tol in ((1.6, 2e-5), (64.0, 2e-3), (1.0 / 64.0, 2e-5)):
            xs = (rng.random((3, 512, 3)) * mul).astype(np.float32)
            rec2, lat2, q2 = g(torch.from_numpy(xs).cuda())
            assert torch.isfinite(rec2).all() and torch.isfinite(lat2).all()
            with torch.no_grad():
                orec2, olat2, oq2, oz2 = m(torch.from_numpy(xs))
            np.testing.assert_allclose(lat2.cpu().numpy(), olat2.numpy(), atol=tol, rtol=0)
            bad2 = q2.cpu().numpy() != oq2.numpy()
            assert _near_boundary(oz2.numpy()[bad2], 1e-3 * max(mul, 1.0)).all()
    finally:
        pccx.DEFAULT_MATMUL = old


@pytest.mark.gpu
def test_third_level_union_maximum_equals_gather_max_then_group_max_bit_for_bit(oracle_nets):
    """PPPF_AE's third set-abstraction level feeds only the maximum over its 32 centroids (PPPF_AE.py:44), and its stack acts on each
    un-centred source row by itself (pointnet_sa_module.py:73-91): max over centroids of max over samples = max over the rows that are a
    sample of any centroid.  The f16x2 path marks those rows (pccx_group_members; -1 -> row 0 as the gather's clamp does) and takes the
    maximum in the last layer's epilogue (pccx_planes_gemm_h2_member_max).  Against the per-centroid table (union_max=False: row epilogue,
    pccx_gather_max, pccx_group_max): IDENTICAL latents, symbols and reconstructions -- dense balls (every row a member), sparse clouds
    whose balls are padded, and a cloud where most rows are outside every ball.  The two entry points against torch as well."""
    import pccx
    from pccx import _lib, families
    m, _ = oracle_nets
    g = families.PPPF_AE(512, 0, 16, 7)
    g.load_state_dict(m.state_dict())
    rng = np.random.default_rng(11)
    far = (rng.random((4, 512, 3)) * 0.05).astype(np.float32)
    far[:, ::2] += rng.integers(0, 2, (4, 256, 3)).astype(np.float32) * 40.0         # clusters 40 apart: most rows outside most balls
    old = pccx.DEFAULT_MATMUL
    try:
        pccx.DEFAULT_MATMUL = "f16x2"
        for x in (synth.pppf_input(), (rng.random((5, 512, 3)) * 1.6).astype(np.float32), (rng.random((3, 512, 3)) * 30).astype(np.float32), far):
            xc = torch.from_numpy(x).cuda()
            assert families.PointnetSAModule.union_max
            a = g(xc)
            try:
                families.PointnetSAModule.union_max = False
                b = g(xc)
            finally:
                families.PointnetSAModule.union_max = True
            for u, v in zip(a, b):
                assert torch.equal(u, v)
    finally:
        pccx.DEFAULT_MATMUL = old
    # pccx_group_members: the rows named by idx (with the clamp), per batch element
    B, G, ns, N = 7, 5, 16, 128
    idx = torch.from_numpy(rng.integers(-1, N, (B, G, ns))).cuda()
    idx[3] = -1                                                                       # an element whose balls are all padding: row 0 only
    member = torch.full((B * N,), 7, device="cuda", dtype=torch.uint8)
    _lib.call("pccx_group_members", idx.data_ptr(), idx.numel(), G * ns, N, member.data_ptr(), torch.cuda.current_stream().cuda_stream)
    want = torch.zeros(B, N, dtype=torch.uint8, device="cuda")
    want.scatter_(1, idx.clamp(min=0).view(B, -1), 1)
    assert torch.equal(member.view(B, N), want) and int(want[3].sum()) == 1
    with pytest.raises(_lib.PccxError):
        _lib.call("pccx_group_members", idx.data_ptr(), idx.numel(), G * ns + 1, N, member.data_ptr(), None)
    # pccx_planes_gemm_h2_member_max against the row epilogue of the same layer reduced by torch
    stack = g._packed["sa"][2]
    layer, M = stack[-1], B * N
    rows_in = (rng.random((M, layer.K)) * (16384.0 / layer.h2["sig"])).astype(np.float32)   # sigma * input within fp16's range
    pl = torch.empty(_lib.load().pccx_planes_floats_h2(M, layer.K), device="cuda", dtype=torch.float32)
    xin = torch.from_numpy(rows_in).cuda()
    _lib.call("pccx_group_planes_h2", xin.data_ptr(), layer.K, layer.K, None, 0, 0, None, M, 1, 1, float(layer.h2["sig"]), None, pl.data_ptr(),
              torch.cuda.current_stream().cuda_stream)
    rows = layer.planes_h2(pl, M, 1)
    got = layer.planes_h2(pl, M, 2, group=N, member=member)
    ref = torch.where(want.view(B, N, 1).bool(), rows.view(B, N, -1), torch.full_like(rows.view(B, N, -1), float("-inf"))).amax(dim=1)
    assert torch.equal(got, ref)
    none = layer.planes_h2(pl, M, 2, group=N, member=torch.zeros_like(member))
    assert torch.equal(none, torch.zeros_like(none))                                  # no member: relu(-inf) = 0


@pytest.mark.gpu
def test_padded_rows_between_the_levels_equal_the_operand_plane_pass_bit_for_bit(oracle_nets):
    """Between the f16x2 levels of PPPF_AE the group maxima are written as the next level's input rows [features | xyz | 0]
    (pccx_gather_max_rows) and that level's first kernel gathers and splits them itself (pccx_planes_chain4_gather_h2 /
    pccx_planes_gemm_gather_h2 with an identity index) instead of reading operand planes built by pccx_group_planes_h2.  Against the
    plane pass (padded_levels=False): IDENTICAL latents, symbols and reconstructions, with either form of the third level; and
    pccx_gather_max_rows against pccx_gather_max + the tail it appends."""
    import pccx
    from pccx import _lib, families
    m, _ = oracle_nets
    g = families.PPPF_AE(512, 0, 16, 7)
    g.load_state_dict(m.state_dict())
    rng = np.random.default_rng(13)
    old = pccx.DEFAULT_MATMUL
    try:
        pccx.DEFAULT_MATMUL = "f16x2"
        for x in (synth.pppf_input(), (rng.random((5, 512, 3)) * 1.6).astype(np.float32), (rng.random((3, 512, 3)) * 30).astype(np.float32)):
            xc = torch.from_numpy(x).cuda()
            for um in (True, False):
                families.PointnetSAModule.union_max = um
                try:
                    assert families.PointnetSAModule.padded_levels
                    a = g(xc)
                    families.PointnetSAModule.padded_levels = False
                    b = g(xc)
                finally:
                    families.PointnetSAModule.padded_levels = True
                    families.PointnetSAModule.union_max = True
                for u, v in zip(a, b):
                    assert torch.equal(u, v)
    finally:
        pccx.DEFAULT_MATMUL = old
    B, N, C, M, ns = 3, 64, 132, 10, 7                                                # C + 3 = 135 -> rows of 160 floats
    y = torch.from_numpy(rng.standard_normal((B, N, C)).astype(np.float32)).cuda()
    idx = torch.from_numpy(rng.integers(-1, N, (B, M, ns))).cuda()
    cxyz = torch.from_numpy(rng.standard_normal((B, M, 3)).astype(np.float32)).cuda()
    pr = families.gather_max_rows(y, idx, cxyz)
    assert pr.C == C and tuple(pr.src.shape) == (B, M, 160)
    assert torch.equal(pr.src[..., :C], families.gather_max(y, idx)) and torch.equal(pr.src[..., C:C + 3], cxyz) and not pr.src[..., C + 3:].any()
    with pytest.raises(_lib.PccxError):
        _lib.call("pccx_gather_max_rows", y.data_ptr(), B, N, C, idx.data_ptr(), M, ns, cxyz.data_ptr(), pr.src.data_ptr(), 136, None)


@pytest.mark.gpu
def test_pointnet_sa_module_on_source_rows_equals_the_grouped_evaluation_bit_for_bit(oracle_nets, matmul_mode):
    """PointnetSAModule gathers features and xyz un-centred (pointnet_sa_module.py:73-85), so each grouped row is a copy of a source
    row: families.PointnetSAModule evaluates its Conv-BN-ReLU stack on the N source rows and takes every group's maximum from that
    (pccx_gather_max).  Against the literal evaluation on all npoint x nsample gathered rows (dedup=False): IDENTICAL latents,
    symbols and reconstructions, on the fixture input and on a sparse batch whose balls are padded with -1 (-> row 0, :27)."""
    from pccx import families
    m, _ = oracle_nets
    g = families.PPPF_AE(512, 0, 16, 7)
    g.load_state_dict(m.state_dict())
    rng = np.random.default_rng(7)
    for x in (synth.pppf_input(), (rng.random((5, 512, 3)) * 1.6).astype(np.float32)):
        xc = torch.from_numpy(x).cuda()
        a = g(xc)
        try:
            families.PointnetSAModule.dedup = False
            b = g(xc)
        finally:
            families.PointnetSAModule.dedup = True
        for u, v in zip(a, b):
            assert torch.equal(u, v)
    # FoldingNet's first layers as per-patch + per-point parts (PPPF_AE.split_fold) against the literal 1026- / 1027-wide rows:
    # the same numbers up to the summation order of one long dot product
    xc = torch.from_numpy(synth.pppf_input()).cuda()
    a = g(xc)
    try:
        families.PPPF_AE.split_fold = False
        b = g(xc)
    finally:
        families.PPPF_AE.split_fold = True
    assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    np.testing.assert_allclose(a[0].cpu().numpy(), b[0].cpu().numpy(), atol=2e-6, rtol=1e-5)
    # the gather-max kernel alone against torch, LDS-tiled and through-L2 forms, -1 padding, ragged channel chunks
    for (B, N, Cc, M, ns) in ((3, 128, 1024, 32, 128), (2, 512, 128, 512, 32), (2, 20000, 8, 7, 5), (1, 512, 260, 9, 64)):
        y = torch.from_numpy(rng.normal(size=(B, N, Cc)).astype(np.float32)).cuda()
        idx = torch.from_numpy(rng.integers(-1, N, size=(B, M, ns))).cuda()
        want = torch.gather(y[:, None].expand(B, M, N, Cc), 2, idx.clamp(min=0)[..., None].expand(B, M, ns, Cc)).amax(dim=2)
        assert torch.equal(families.gather_max(y, idx), want)
        try:                                                 # the form without the index table in LDS (what larger tables take)
            os.environ["PCCX_GATHER_MAX_PLAIN"] = "1"
            assert torch.equal(families.gather_max(y, idx), want)
        finally:
            del os.environ["PCCX_GATHER_MAX_PLAIN"]


@pytest.mark.gpu
def test_pppe_forward_gpu_matches_reference_fixture_and_oracle(fam, oracle_nets, matmul_mode):
    from pccx import families
    _, p = oracle_nets
    g = families.PointCloudAE(64, 16, 8192)
    assert list(g.state_dict().keys()) == list(fam["pppe_keys"])
    g.load_state_dict(p.state_dict())
    x = synth.pppe_input()
    st = _starts(fam)
    coarse, fine, cond, yq, latent = g(torch.from_numpy(x).cuda(), st)
    np.testing.assert_allclose(cond.cpu().numpy(), fam["pppe_cond"], atol=2e-5, rtol=1e-4)
    with torch.no_grad():
        oc, of, ocond, oyq, olat = p(torch.from_numpy(x), st)
    np.testing.assert_allclose(latent.cpu().numpy(), olat.numpy(), atol=2e-4, rtol=1e-4)
    bad = yq.cpu().numpy() != fam["pppe_yq"]
    assert _near_boundary(np.clip(olat.numpy(), 0, 15)[bad], 1e-3).all()
    if not bad.any():
        np.testing.assert_allclose(coarse.cpu().numpy(), fam["pppe_coarse"], atol=5e-5, rtol=1e-4)
        np.testing.assert_allclose(fine[:, ::16].cpu().numpy(), fam["pppe_fine_sample"], atol=5e-5, rtol=1e-4)
        # Chamfer loss of the fast path (pppe_pcd_ae.py:817-820) on the HIP nearest-neighbour kernel
        from pccx import ops
        from oracle import ref_model
        cd, _ = ops.chamfer_distance(fine, torch.from_numpy(x).cuda())
        want, _ = ref_model.chamfer_distance(of, torch.from_numpy(x))
        assert abs(float(cd) - want) <= 1e-4 * abs(want)


@pytest.mark.gpu
def test_generic_linear_ragged_shapes(matmul_mode):
    from pccx import families
    rng = np.random.default_rng(0)
    for M, K, N, relu in [(1, 3, 3, True), (130, 3, 64, True), (257, 131, 128, False), (33, 1027, 128, True),
                          (4, 64, 1536, False), (1000, 259, 7, True)]:
        W = rng.standard_normal((N, K)).astype(np.float32) / np.sqrt(K)
        b = rng.standard_normal(N).astype(np.float32)
        x = rng.standard_normal((M, K)).astype(np.float32)
        lyr = families.FoldedLinear(torch.from_numpy(W), torch.from_numpy(b), relu)
        got = lyr(torch.from_numpy(x).cuda()).cpu().numpy()
        want = x.astype(np.float64) @ W.T.astype(np.float64) + b
        if relu:
            want = np.maximum(want, 0)
        np.testing.assert_allclose(got, want, atol=2e-5, rtol=1e-5)
    x = rng.standard_normal((7, 33, 20)).astype(np.float32)
    assert np.array_equal(families.group_max(torch.from_numpy(x).cuda()).cpu().numpy(), x.max(1))


@pytest.mark.gpu
def test_planes_layers_ragged_shapes():
    """csrc/planes.hip against float64 matmuls at the tolerances of the generic layer test: rows -> planes -> layer(s) -> rows for
    ragged M / K / N (K, N not multiples of 32 / 16, a single row, M not a multiple of 128), a two-layer chain kept in planes,
    the gather + concat front end with -1 indices (pointnet_sa_module.py:27,73-83) and the max-over-nsample epilogue (:91)."""
    from pccx import families
    rng = np.random.default_rng(5)

    def layer(N, K, relu):
        W = rng.standard_normal((N, K)).astype(np.float32) / np.sqrt(K)
        b = rng.standard_normal(N).astype(np.float32)
        return families.FoldedLinear(torch.from_numpy(W), torch.from_numpy(b), relu, matmul="bf16x3"), W.astype(np.float64), b.astype(np.float64)

    for M, K, N, relu in [(1, 3, 3, True), (130, 3, 64, True), (257, 131, 128, False), (33, 1027, 128, True), (200, 64, 1024, False),
                          (1000, 259, 7, True), (4096, 512, 512, True)]:
        lyr, W, b = layer(N, K, relu)
        x = rng.standard_normal((M, K)).astype(np.float32)
        got = lyr.planes(families.rows_planes(torch.from_numpy(x).cuda()), M, 1).cpu().numpy()
        want = x.astype(np.float64) @ W.T + b
        if relu:
            want = np.maximum(want, 0)
        np.testing.assert_allclose(got, want, atol=2e-5, rtol=1e-5, err_msg=str((M, K, N)))
    # [a | b repeated] straight to planes (PPPF_AE.py:99-106) == the concatenated rows converted
    a, b2 = rng.standard_normal((10, 2)).astype(np.float32), rng.standard_normal((7, 37)).astype(np.float32)
    cat = np.concatenate([np.tile(a, (7, 1)), np.repeat(b2, 10, axis=0)], axis=1)
    got = families.fold_planes(torch.from_numpy(a).cuda(), 10, torch.from_numpy(b2).cuda(), 10, 70)
    assert torch.equal(got, families.rows_planes(torch.from_numpy(cat).cuda()))
    a3 = rng.standard_normal((70, 3)).astype(np.float32)
    cat = np.concatenate([a3, np.repeat(b2, 10, axis=0)], axis=1)
    got = families.fold_planes(torch.from_numpy(a3).cuda(), 0, torch.from_numpy(b2).cuda(), 10, 70)
    assert torch.equal(got, families.rows_planes(torch.from_numpy(cat).cuda()))
    # chain of two layers through planes (odd number of 16-channel tiles in the middle: 48 channels)
    l0, W0, b0 = layer(48, 35, True)
    l1, W1, b1 = layer(40, 48, False)
    x = rng.standard_normal((300, 35)).astype(np.float32)
    got = families.run_stack([l0, l1], torch.from_numpy(x).cuda()).cpu().numpy()
    want = np.maximum(x.astype(np.float64) @ W0.T + b0, 0) @ W1.T + b1
    np.testing.assert_allclose(got, want, atol=2e-5, rtol=1e-5)
    # gather + concat + layer + max over nsample
    for ns, C in [(32, 0), (64, 5), (128, 128)]:
        B, Nsrc, Mq = 3, 50, 7
        xyz = rng.standard_normal((B, Nsrc, 3)).astype(np.float32)
        feats = rng.standard_normal((B, Nsrc, C)).astype(np.float32) if C else None
        idx = rng.integers(-1, Nsrc, (B, Mq, ns))
        lyr, W, b = layer(70, C + 3, True)
        pl, rows = families.group_planes(torch.from_numpy(feats).cuda() if C else None, torch.from_numpy(xyz).cuda(), torch.from_numpy(idx).cuda())
        assert rows == B * Mq * ns
        got = lyr.planes(pl, rows, 2, ns).cpu().numpy()
        j = np.where(idx < 0, 0, idx)
        bi = np.arange(B)[:, None, None]
        g = np.concatenate(([feats[bi, j]] if C else []) + [xyz[bi, j]], axis=-1).astype(np.float64)       # (B,Mq,ns,C+3)
        want = np.maximum(g @ W.T + b, 0).max(axis=2).reshape(B * Mq, -1)
        np.testing.assert_allclose(got, want, atol=2e-5, rtol=1e-5, err_msg=str((ns, C)))
        # the same through the gathering form of the layer kernel (and of a two-layer stack): bit-identical
        f_, z_, i_ = torch.from_numpy(feats).cuda() if C else None, torch.from_numpy(xyz).cuda(), torch.from_numpy(idx).cuda()
        assert np.array_equal(families.stack_max_gather([lyr], f_, z_, i_, {}).cpu().numpy(), got)
        l2, _, _ = layer(33, 70, True)
        two = families.stack_max_gather([lyr, l2], f_, z_, i_, {}).cpu().numpy()
        assert np.array_equal(two, l2.planes(lyr.planes(pl, rows, 0), rows, 2, ns).cpu().numpy())


@pytest.mark.gpu
def test_planes_chain4_matches_layer_by_layer_and_float64():
    """pccx_planes_chain4 (four layers + max in one kernel, activations in registers) on the two width patterns of PPPF_AE.py:29-34
    and ragged variants of them: equal to the float64 stack at the layer tolerance, and BIT-identical to the same stack run layer by
    layer through pccx_planes_gemm (same products in the same order; only where the activation lives differs)."""
    from pccx import families
    rng = np.random.default_rng(9)
    for K0, widths, ns, groups in [(3, (3, 64, 64, 128), 32, 37), (131, (128, 128, 128, 256), 64, 21), (7, (20, 40, 64, 100), 32, 5),
                                   (70, (100, 128, 97, 200), 128, 3)]:
        stack, Ws = [], []
        k = K0
        for nw in widths:
            W = rng.standard_normal((nw, k)).astype(np.float32) / np.sqrt(k)
            b = rng.standard_normal(nw).astype(np.float32) * 0.1
            stack.append(families.FoldedLinear(torch.from_numpy(W), torch.from_numpy(b), True, matmul="bf16x3"))
            Ws.append((W.astype(np.float64), b.astype(np.float64)))
            k = nw
        assert families.chain4_fits(stack)
        rows = groups * ns
        x = rng.standard_normal((rows, K0)).astype(np.float32)
        pl = families.rows_planes(torch.from_numpy(x).cuda())
        got = families.stack_max_planes(stack, pl, rows, ns, {}).cpu().numpy()
        p2 = pl
        for layer in stack[:-1]:
            p2 = layer.planes(p2, rows, 0)
        ref = stack[-1].planes(p2, rows, 2, ns).cpu().numpy()
        assert np.array_equal(got, ref), (K0, widths)
        h = x.astype(np.float64)
        for W, b in Ws:
            h = np.maximum(h @ W.T + b, 0)
        np.testing.assert_allclose(got, h.reshape(groups, ns, -1).max(1), atol=3e-5, rtol=2e-5)
        # the same stack with the gather inside the kernel (pointnet_sa_module.py:73-83; -1 -> row 0): bit-identical to
        # gather -> planes -> chain
        if K0 >= 4:
            Bq, Nsrc = 3, 40
            Mq = groups // Bq
            feats = rng.standard_normal((Bq, Nsrc, K0 - 3)).astype(np.float32)
            xyz = rng.standard_normal((Bq, Nsrc, 3)).astype(np.float32)
            idx = torch.from_numpy(rng.integers(-1, Nsrc, (Bq, Mq, ns))).cuda()
            f, z = torch.from_numpy(feats).cuda(), torch.from_numpy(xyz).cuda()
            pl2, rows2 = families.group_planes(f, z, idx)
            want = families.stack_max_planes(stack, pl2, rows2, ns, {}).cpu().numpy()
            got2 = families.stack_max_gather(stack, f, z, idx, {}).cpu().numpy()
            assert np.array_equal(got2, want), (K0, widths)


@pytest.mark.gpu
def test_planes_chain_wide_matches_layer_by_layer_and_float64():
    """pccx_planes_chain_wide (three wide layers in one kernel, one row tile per wave, gather inside) on sa3's shape
    (PPPF_AE.py:32-34: 259 -> 256 -> 256 -> 512, then 512 -> 1024 + max over 128) and a ragged variant: bit-identical to the same
    stack run layer by layer through pccx_planes_gemm(_gather), and equal to the float64 stack at the layer tolerance."""
    from pccx import families
    rng = np.random.default_rng(12)
    for C, widths, ns, Bq, Mq, Nsrc in [(256, (256, 256, 512, 1024), 128, 2, 3, 128), (250, (241, 250, 500, 70), 32, 3, 5, 40)]:
        stack, Ws = [], []
        k = C + 3
        for nw in widths:
            W = rng.standard_normal((nw, k)).astype(np.float32) / np.sqrt(k)
            b = rng.standard_normal(nw).astype(np.float32) * 0.1
            stack.append(families.FoldedLinear(torch.from_numpy(W), torch.from_numpy(b), True, matmul="bf16x3"))
            Ws.append((W.astype(np.float64), b.astype(np.float64)))
            k = nw
        assert families.wide3_fits(stack) and not families.chain4_fits(stack)
        feats = rng.standard_normal((Bq, Nsrc, C)).astype(np.float32)
        xyz = rng.standard_normal((Bq, Nsrc, 3)).astype(np.float32)
        idx_np = rng.integers(-1, Nsrc, (Bq, Mq, ns))
        f, z, idx = torch.from_numpy(feats).cuda(), torch.from_numpy(xyz).cuda(), torch.from_numpy(idx_np).cuda()
        got = families.stack_max_gather(stack, f, z, idx, {}).cpu().numpy()
        src, Cc = families.padded_rows(f, z)
        rows = Bq * Mq * ns
        pl = stack[0].planes_gather(src, Cc, idx)
        for layer in stack[1:-1]:
            pl = layer.planes(pl, rows, 0)
        ref = stack[-1].planes(pl, rows, 2, ns).cpu().numpy()
        assert np.array_equal(got, ref), widths
        j = np.where(idx_np < 0, 0, idx_np)
        bi = np.arange(Bq)[:, None, None]
        h = np.concatenate([feats[bi, j], xyz[bi, j]], axis=-1).astype(np.float64)
        for W, b in Ws:
            h = np.maximum(h @ W.T + b, 0)
        np.testing.assert_allclose(got, h.max(axis=2).reshape(Bq * Mq, -1), atol=5e-5, rtol=3e-5)
