"""GPU parity: geometry / selection / integer kernels vs the oracle and the golden fixtures.

Everything here goes through the C ABI (pccx.ops -> ctypes -> libpccx.so).  Bar: bit-exact for
indices, bit streams, bytes and for the fp32 results of normalize/denormalize/kNN distances
(same fp32 expression, no FMA contraction).
"""
import os

import numpy as np
import pytest
import torch

from oracle import cport, ref_model
from pccx import ops, synth as cloud_synth
from tests import synth

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def dev(a):
    return torch.as_tensor(a).cuda()


def test_normalize_denormalize_golden_and_oracle():
    fp = np.load(os.path.join(G, "pnkit_float.npz"))
    for i, (pc, S) in enumerate(synth.fps_cases()):
        x = dev(pc)[None]
        xn, c, l = ops.normalize(x)
        assert np.array_equal(c[0].cpu().numpy(), fp["centers"][i])
        assert float(l[0]) == fp["longest"][i]
        assert np.array_equal(xn[0, ::257].cpu().numpy(), fp[f"norm_sample_{i}"])
        on, oc_, ol = ref_model.normalize(torch.from_numpy(pc)[None])
        assert np.array_equal(xn.cpu().numpy(), on.numpy())
        back = ops.denormalize(xn, c, l)
        assert np.array_equal(back[0, ::257].cpu().numpy(), fp[f"denorm_sample_{i}"])
    # batched: several clouds in one launch
    batch = cloud_synth.cad_batch(40, 5, 4096) * np.float32(3.0) - np.float32(1.0)
    xn, c, l = ops.normalize(dev(batch))
    for b in range(5):
        on, oc_, ol = ref_model.normalize(torch.from_numpy(batch[b])[None])
        assert np.array_equal(xn[b].cpu().numpy(), on[0].numpy())
        assert np.array_equal(c[b].cpu().numpy(), oc_.numpy()) and float(l[b]) == float(ol)


def test_fps_golden_indices():
    fp = np.load(os.path.join(G, "pnkit_float.npz"))
    for i, (pc, S) in enumerate(synth.fps_cases()):
        idx = ops.farthest_point_sample_batch(dev(pc)[None], S, start_idx=[int(fp["starts"][i])])
        assert np.array_equal(idx[0].cpu().numpy(), fp[f"fps_idx_{i}"]), f"case {i}"
        g = ops.index_points(dev(pc)[None], idx)
        assert np.array_equal(g[0].cpu().numpy(), fp[f"gather_{i}"])


@pytest.mark.parametrize("N,S", [(1, 1), (7, 7), (64, 10), (128, 128), (129, 40), (256, 256), (257, 32), (512, 512), (513, 9), (1000, 33),
                                 (1024, 64), (1025, 5), (3000, 100),
                                 (8192, 64), (10000, 17), (16384, 8), (20000, 9)])
def test_fps_ragged_sizes_vs_oracle(N, S):
    rng = np.random.default_rng(N * 31 + S)
    B = 3 if N > 512 or N < 64 else 7               # N <= 512: one wave per cloud, four clouds per workgroup (ragged last workgroup)
    pcs = rng.random((B, N, 3)).astype(np.float32)
    if N >= 64:
        pcs[1, 5] = pcs[1, 50]                      # duplicate points: argmax ties -> first index
        pcs[2] = np.round(pcs[2] * 8) / 8           # heavy ties on a lattice
    starts = rng.integers(0, N, size=B)
    got = ops.farthest_point_sample_batch(dev(pcs), S, start_idx=starts).cpu().numpy()
    for b in range(B):
        assert np.array_equal(got[b], cport.fps(pcs[b], S, int(starts[b]))), f"cloud {b}"


@pytest.mark.parametrize("N,M,K", [(8192, 64, 256), (300, 5, 300), (256, 256, 16), (1000, 3, 1), (5000, 7, 1000),
                                   (32768, 2, 64), (2, 2, 2)])
def test_knn_vs_oracle(N, M, K):
    rng = np.random.default_rng(N + M + K)
    ref = rng.random((2, N, 3)).astype(np.float32)
    ref[1] = np.round(ref[1] * 16) / 16             # lattice: massive distance ties -> index order
    q = ref[:, rng.choice(N, size=M, replace=M > N)] if M <= N else rng.random((2, M, 3)).astype(np.float32)
    q = np.ascontiguousarray(q)
    r = ops.knn_points(dev(q), dev(ref), K)
    for b in range(2):
        d, i = cport.knn(q[b], ref[b], K)
        assert np.array_equal(r.idx[b].cpu().numpy(), i)
        assert np.array_equal(r.dists[b].cpu().numpy(), d)
        assert np.array_equal(r.knn[b].cpu().numpy(), ref[b][i])


def test_knn_near_equidistant_and_massively_tied_clouds():
    """The fast kNN kernel narrows the K-th distance by histogram levels (11 + 11 + 9 key bits) only as far as
    needed.  A jittered shell around the query puts every distance into one 1/8-octave bin and forces the deeper
    levels; a cloud with 1500 copies of one point leaves more exact ties at the K-th distance than the tie buffer
    holds and forces the ordered lowest-index pass.  A far-away query stays on the one-level path.  All must equal
    the oracle's (distance, index) order."""
    rng = np.random.default_rng(8)
    N, K = 4096, 200
    dirs = rng.normal(size=(N, 3)); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    shell = (0.5 + 0.3 * dirs * (1.0 + rng.integers(-30, 31, size=(N, 1)) * 2.0 ** -23)).astype(np.float32)
    tied = rng.random((N, 3)).astype(np.float32)
    tied[rng.choice(N, size=1500, replace=False)] = np.float32([0.25, 0.5, 0.75])
    tied[rng.choice(N, size=50, replace=False)] = np.float32([0.26, 0.5, 0.75])       # 50 nearer ones: r < ties
    ref = np.stack([shell, tied])
    q = np.stack([np.array([[0.5, 0.5, 0.5], [2.0, -1.0, 0.25], [0.5, 0.5, 0.8]], dtype=np.float32),
                  np.array([[0.26, 0.5, 0.75], [0.25, 0.5, 0.75], [0.9, 0.1, 0.2]], dtype=np.float32)])
    r = ops.knn_points(dev(q), dev(ref), K)
    for b in range(2):
        d, i = cport.knn(q[b], ref[b], K)
        assert np.array_equal(r.idx[b].cpu().numpy(), i)
        assert np.array_equal(r.dists[b].cpu().numpy(), d)
        assert np.array_equal(r.knn[b].cpu().numpy(), ref[b][i])


def test_knn_patches_fused_centre_and_scale():
    pc = cloud_synth.cad_cloud(3, 8192)
    c = pc[:64] + np.float32(0.001)
    scale = float((8192 / 1024) ** (1 / 3))
    r = ops.knn_points(dev(c)[None], dev(pc)[None], 256, patch_scale=scale)
    d, i = cport.knn(c, pc, 256)
    want = (pc[i] - c[:, None, :]) * np.float32(scale)
    assert np.array_equal(r.knn[0].cpu().numpy(), want)
    # the codec's form (KNN_Patching, compress.py:70-74, keeps the gathered points only): distances and indices not written at all
    r2 = ops.knn_points(dev(c)[None], dev(pc)[None], 256, patch_scale=scale, return_dists=False, return_idx=False)
    assert r2.dists is None and r2.idx is None and torch.equal(r2.knn, r.knn)
    r3 = ops.knn_points(dev(c)[None], dev(pc)[None], 256, return_nn=False, return_dists=False)      # indices alone (estimate_normals)
    assert r3.dists is None and r3.knn is None and torch.equal(r3.idx, r.idx)
    # the slower general kernel (N > 8192 candidates) takes the same optional outputs
    big = np.concatenate([pc, pc[::-1] * np.float32(0.5)])
    a = ops.knn_points(dev(c)[None], dev(big)[None], 300, patch_scale=scale)
    b_ = ops.knn_points(dev(c)[None], dev(big)[None], 300, patch_scale=scale, return_dists=False, return_idx=False)
    assert torch.equal(a.knn, b_.knn)
    with pytest.raises(ops._lib.PccxError):
        ops.knn_points(dev(c)[None], dev(pc)[None], 256, return_nn=False, return_dists=False, return_idx=False)


@pytest.mark.parametrize("method", ["scan", "grid"])
@pytest.mark.parametrize("N,M,K,r", [(512, 128, 32, 0.2), (2048, 100, 64, 0.4), (100, 7, 128, 0.8), (777, 3, 5, 0.05),
                                     (20000, 300, 48, 0.04), (32768, 64, 16, 0.02), (5000, 50, 8, 2.0), (1, 4, 3, 0.5)])
def test_ball_query_vs_oracle(N, M, K, r, method):
    """Both ball-query kernels -- the ordered scan and the grid hash (cells >= r, 27-cell walk, bitmap read-out) -- against the
    oracle: the first K in-radius candidates IN INDEX ORDER, -1 padded, exact distances.  Includes a radius larger than the
    cloud (one cell), a one-point cloud, queries outside the candidates' bounding box and points on cell boundaries."""
    rng = np.random.default_rng(N + K)
    ref = rng.random((2, N, 3)).astype(np.float32)
    q = (rng.random((2, M, 3)) * 1.3 - 0.15).astype(np.float32)          # some queries lie outside the candidates' box
    if N >= 16:
        ref[0, :8] = np.round(ref[0, :8] * 8) / 8                        # points exactly on cell faces
        q[0, :min(4, M)] = ref[0, :min(4, M)]
    got = ops.ball_query(dev(q), dev(ref), K, r, method=method)
    for b in range(2):
        d, i = cport.ball_query(q[b], ref[b], K, r)
        assert np.array_equal(got.idx[b].cpu().numpy(), i)
        assert np.array_equal(got.dists[b].cpu().numpy(), d)


def test_ball_query_grid_on_a_room_scale_block_matches_the_scan():
    """The grid hash where it is meant to be used: 32 768 candidates of a room-like cloud (planes: very uneven cells), radius
    5 cm, 2048 queries -- identical to the scan."""
    from pccx import synth as cloud_synth
    pc = cloud_synth.room_cloud(100, 32768)
    ref = dev(pc[None])
    q = dev(pc[None, ::16].copy())
    a = ops.ball_query(q, ref, 32, 0.05, method="scan")
    b = ops.ball_query(q, ref, 32, 0.05, method="grid")
    assert torch.equal(a.idx, b.idx) and torch.equal(a.dists, b.dists)
    assert int((a.idx >= 0).sum()) > 2048                                   # neighbourhoods are not empty


@pytest.mark.parametrize("P,Q", [(8192, 8192), (1000, 3000), (1, 5), (5000, 1), (4097, 1025)])
def test_nn_dist_and_chamfer_vs_oracle(P, Q):
    rng = np.random.default_rng(P + Q)
    x = rng.random((2, P, 3)).astype(np.float32)
    y = rng.random((2, Q, 3)).astype(np.float32)
    y[1] = np.round(y[1] * 4) / 4
    d2, nn = ops.nn_dist(dev(x), dev(y), return_idx=True)
    for b in range(2):
        d, i = cport.nn_dist(x[b], y[b])
        assert np.array_equal(d2[b].cpu().numpy(), d)
        assert np.array_equal(nn[b].cpu().numpy(), i)
    got, _ = ops.chamfer_distance(dev(x), dev(y))
    want, _ = ref_model.chamfer_distance(torch.from_numpy(x), torch.from_numpy(y))
    assert abs(float(got) - want) <= 1e-6 * abs(want)      # fp64 means of identical fp32 terms


def _check_octree(pcs, N, min_bpp):
    r = ops.octree_encode(dev(pcs), N, min_bpp)
    nbits, depth = r["nbits"].cpu().numpy(), r["depth"].cpu().numpy()
    bits, bytes_, nbytes = r["bits"].cpu().numpy(), r["bytes"].cpu().numpy(), r["nbytes"].cpu().numpy()
    for b in range(pcs.shape[0]):
        want, wd = cport.encode_sampled(pcs[b], 1, N, min_bpp)
        assert nbits[b] == want.shape[0] and depth[b] == wd, f"cloud {b}: {nbits[b]} vs {want.shape[0]}, {depth[b]} vs {wd}"
        assert np.array_equal(bits[b, :nbits[b]], want), f"cloud {b}"
        assert bytes(bytes_[b, :nbytes[b]]) == bytes(cport.pack_bits(want))
    return r


def test_octree_depth_search_golden():
    ds = np.load(os.path.join(G, "depth_search_pack.npz"))
    bo, yo = ds["bits_off"], ds["bytes_off"]
    for i, (pcs, N, K) in enumerate(synth.depth_search_cases()):
        r = ops.octree_encode(dev(pcs), N, ops.OCTREE_BPP_DICT[K])
        nb = int(r["nbits"][0])
        assert nb == ds["total_bits"][i]
        assert np.array_equal(r["bits"][0, :nb].cpu().numpy(), ds["bits"][bo[i]:bo[i + 1]]), f"case {i}"
        ny = int(r["nbytes"][0])
        assert r["bytes"][0, :ny].cpu().numpy().tobytes() == ds["bytes"][yo[i]:yo[i + 1]].tobytes()


def test_octree_encode_batched_vs_oracle_incl_edge_cases():
    rng = np.random.default_rng(8)
    pcs = (0.005 + 0.99 * rng.random((40, 64, 3))).astype(np.float32)
    pcs[1, 3] = pcs[1, 60]                                    # duplicate centre -> depth 17, code of depth 16
    pcs[2, 1] = pcs[2, 0] + np.float32(2.0 ** -15)            # deep search
    pcs[3, 0] = [1.0, 0.3, 0.2]; pcs[3, 63] = [0.0, 0.0, 0.0]  # boundary values
    pcs[4, 0] = [-0.01, 0.5, 1.2]                             # outside the cube
    pcs[5] = 1.5 + pcs[5]                                     # nothing inside: stream is the single bit 0
    pcs[6] = (0.5 + 0.01 * rng.standard_normal((64, 3))).astype(np.float32)   # clustered
    _check_octree(pcs, 8192, 0.25)
    for S, N, K in [(32, 8192, 512), (128, 8192, 128), (256, 8192, 64), (7, 2048, 512), (1, 1024, 1024), (700, 8192, 64)]:
        p = (0.005 + 0.99 * rng.random((6, S, 3))).astype(np.float32)
        _check_octree(p, N, ops.OCTREE_BPP_DICT[K])


def test_octree_decode_reference_mode_golden_and_round_trip():
    oc = np.load(os.path.join(G, "octree.npz"))
    cases = synth.octree_cases()
    off = oc["bits_off"]
    for i in range(len(cases)):
        bits = oc["bits"][off[i]:off[i + 1]]
        by = np.frombuffer(bytes(cport.pack_bits(bits)), dtype=np.uint8)
        if bits.shape[0] < 8:
            continue        # sub-byte streams unpack differently in the reference itself (tail quirk)
        out, cnt = ops.octree_decode(dev(by)[None], dev(np.array([by.shape[0]], dtype=np.int32)), "reference")
        assert np.array_equal(out[0].cpu().numpy(), oc["decoded_reference"][i]), f"case {i}"
    # what decompress.py sees: bytes -> unpack -> decode, for every golden case incl. sub-byte ones
    for i in range(len(cases)):
        bits = oc["bits"][off[i]:off[i + 1]]
        by = cport.pack_bits(bits)
        want, _ = cport.octree_decode_reference(cport.unpack_bits(by).astype(np.uint8), 1)
        arr = np.frombuffer(bytes(by), dtype=np.uint8)
        out, _ = ops.octree_decode(dev(arr)[None], dev(np.array([arr.shape[0]], dtype=np.int32)), "reference")
        assert np.array_equal(out[0].cpu().numpy(), want)


def test_octree_full_mode_round_trip_on_device():
    rng = np.random.default_rng(9)
    for S in (64, 32, 128, 500):
        pcs = (0.005 + 0.99 * rng.random((16, S, 3))).astype(np.float32)
        r = ops.octree_encode(dev(pcs), 8192, 0.25 if S == 64 else 0.1)
        out, cnt = ops.octree_decode(r["bytes"], r["nbytes"], "full", S_out=S)
        depth = r["depth"].cpu().numpy()
        for b in range(16):
            want, d = cport.octree_decode_full(r["bits"][b, :int(r["nbits"][b])].cpu().numpy(), 1)
            assert d == depth[b] and int(cnt[b]) == want.shape[0] == S
            assert np.array_equal(out[b].cpu().numpy(), want)
            cells = cport.get_decode_from_pc(pcs[b], 1, int(depth[b]))
            assert np.array_equal(np.unique(out[b].cpu().numpy(), axis=0), cells)   # decode inverts encode


@pytest.mark.parametrize("P,Q", [(300, 500), (2048, 2048), (1, 7)])
def test_chamfer_backward_matches_autograd_of_the_definition(P, Q):
    """Loss of AE.py:57-70: the HIP forward/backward against torch autograd on the brute-force
    definition (float64, CPU).  Tolerance 1e-5 relative (fp32 atomics vs an fp64 sum)."""
    rng = np.random.default_rng(P * 7 + Q)
    x = rng.random((2, P, 3)).astype(np.float32)
    y = rng.random((2, Q, 3)).astype(np.float32)
    xg, yg = dev(x).requires_grad_(True), dev(y).requires_grad_(True)
    loss, _ = ops.chamfer_distance(xg, yg)
    (loss * 3.0).backward()
    xr = torch.from_numpy(x).double().requires_grad_(True)
    yr = torch.from_numpy(y).double().requires_grad_(True)
    d = ((xr[:, :, None, :] - yr[:, None, :, :]) ** 2).sum(-1)
    ref = (d.min(2).values.mean(1) + d.min(1).values.mean(1)).mean()
    (ref * 3.0).backward()
    assert abs(float(loss.detach()) - float(ref.detach())) <= 1e-6 * float(ref.detach())
    np.testing.assert_allclose(xg.grad.cpu().numpy(), xr.grad.numpy(), rtol=1e-4, atol=1e-9)
    np.testing.assert_allclose(yg.grad.cpu().numpy(), yr.grad.numpy(), rtol=1e-4, atol=1e-9)


def test_chamfer_backward_with_thousands_of_points_sharing_a_neighbour():
    """The scatter half of the gradient when the reconstruction is a blob (an untrained decoder): every target point's nearest
    neighbour is one of a handful of reconstructed points, so the kernel's per-wave combining rounds carry almost all of the sum;
    a second cloud pair has all rows distinct inside most waves (the plain path).  The expected gradient is the definition's
    (AE.py:57-70: 2 (x - y_nn) / (B P) to x and its negative scattered to y_nn, and symmetrically) summed in float64 over the
    assignment the GPU's own nearest-neighbour pass made -- inside the blob the float32 and float64 arg-mins differ for a few points,
    which is the search's rounding (tested above), not the gradient kernel's business."""
    rng = np.random.default_rng(77)
    P = Q = 4096
    x = np.stack([0.5 + 1e-3 * rng.random((P, 3)), rng.random((P, 3))]).astype(np.float32)
    x[0, :5] = rng.random((5, 3))                               # five points outside the blob: a few distinct heavy rows
    y = rng.random((2, Q, 3)).astype(np.float32)
    xg, yg = dev(x).requires_grad_(True), dev(y).requires_grad_(True)
    loss, _ = ops.chamfer_distance(xg, yg)
    loss.backward()
    nxy = ops.nn_dist(dev(x), dev(y), return_idx=True)[1].cpu().numpy()
    nyx = ops.nn_dist(dev(y), dev(x), return_idx=True)[1].cpu().numpy()
    assert np.unique(nyx[0]).size < 64                          # the case is what it claims: < 64 distinct neighbours for 4096 points
    gx, gy = np.zeros((2, P, 3)), np.zeros((2, Q, 3))
    for b in range(2):
        dx = 2.0 * (x[b].astype(np.float64) - y[b][nxy[b]]) / (2 * P)
        dy = 2.0 * (y[b].astype(np.float64) - x[b][nyx[b]]) / (2 * Q)
        gx[b] += dx
        np.add.at(gy[b], nxy[b], -dx)
        gy[b] += dy
        np.add.at(gx[b], nyx[b], -dy)
    scale = float(np.abs(gx).max())
    assert scale > 100 * float(np.abs(gx[1]).max())             # the heavy rows really carry thousands of contributions
    np.testing.assert_allclose(xg.grad.cpu().numpy(), gx, rtol=1e-4, atol=2e-6 * scale)
    np.testing.assert_allclose(yg.grad.cpu().numpy(), gy, rtol=1e-4, atol=2e-6 * scale)


def test_radix_sort_is_the_stable_sort_and_block_gather_scatter_are_its_inverse_pair():
    """csrc/sort.hip (configs[3]'s block partition): pccx_sort_keys_u64 = torch.sort(keys, stable=True) -- keys sorted in place and
    the permutation identical, ties in input order -- on sizes around the 4096-key tile, with heavy duplication, with keys that use
    all 63 bits and with keys confined to a few low bits; gather_blocks / unsplit_blocks against torch indexing incl. the padded last
    block and a strided (rank-style) subset."""
    from pccx import _lib, large
    from pccx.ops import _stream
    rng = np.random.default_rng(3)
    for n, hi in ((1, 2 ** 62), (255, 2 ** 62), (4096, 7), (4097, 2 ** 63 - 1), (100_003, 1000), (1_000_001, 2 ** 63 - 1)):
        keys = torch.from_numpy(rng.integers(0, hi, n, dtype=np.int64)).cuda()
        if n > 1000:
            keys[::7] = keys[3]                              # a large tie class
        want = torch.sort(keys, stable=True)
        order = torch.empty(n, device="cuda", dtype=torch.int64)
        ws = torch.empty(_lib.load().pccx_sort_keys_workspace_bytes(n), device="cuda", dtype=torch.uint8)
        k2 = keys.clone()
        _lib.call("pccx_sort_keys_u64", k2.data_ptr(), n, 63, order.data_ptr(), ws.data_ptr(), _stream())
        assert torch.equal(k2, want.values), n
        assert torch.equal(order, want.indices), n
    pc = torch.from_numpy(rng.random((20_001, 3)).astype(np.float32)).cuda()
    order = large.morton_order(pc)
    assert torch.equal(order, torch.sort(large.morton_keys(pc), stable=True).indices)
    block, nb = 4096, 5
    idx = torch.cat([order, order[-1:].expand(nb * block - pc.shape[0])])
    want = pc[idx].view(nb, block, 3)
    assert torch.equal(large.gather_blocks(pc, order, block), want)
    # several clouds at once: each sort on a side stream of its own, results as from one-by-one calls on the current stream
    many = [torch.from_numpy(rng.random((n, 3)).astype(np.float32)).cuda() for n in (70_001, 5, 8192, 300_000, 1, 4097, 65_536, 9, 12_345, 100)]
    got = large.morton_orders(many)
    total = sum(int(o.sum()) for o in got)                           # consumed on the current stream right away: the join must hold
    for pc_i, o in zip(many, got):
        assert torch.equal(o, large.morton_order(pc_i))
    assert total == sum(n * (n - 1) // 2 for n in (70_001, 5, 8192, 300_000, 1, 4097, 65_536, 9, 12_345, 100))
    assert torch.equal(large.gather_blocks(pc, order, block, first=1, stride=2), want[1::2])
    back = large.unsplit_blocks(want, list(range(nb)), order, pc.shape[0], block)
    assert torch.equal(back, pc)
    part = torch.full_like(pc, float("nan"))
    large.unsplit_blocks(want[[0, 2, 4]].contiguous(), [0, 2, 4], order, pc.shape[0], block, out=part)
    large.unsplit_blocks(want[[3, 1]].contiguous(), [3, 1], order, pc.shape[0], block, out=part)     # not a progression: per-block launches
    assert torch.equal(part, pc)
    with pytest.raises(_lib.PccxError):
        large.gather_blocks(pc, order, block, first=5, stride=1, count=1)                         # beyond the cloud
