"""Tensor-level operators over libpccx.so, named after the reference callables they replace.

PyTorch is used for device memory and streams only; every computation is a HIP kernel behind
the C ABI (include/pccx.h).  Inputs must live on a ROCm device: there is no CPU path.
"""
import collections

import torch

from . import _lib

KNN = collections.namedtuple("KNN", ["dists", "idx", "knn"])      # pytorch3d's _KNN result shape
OCTREE_BPP_DICT = {1024: 0.07, 512: 0.125, 256: 0.25, 128: 0.5, 64: 1.0}   # pn_kit.py:17-23


def _stream():
    return torch.cuda.current_stream().cuda_stream


class StageTimer:
    """Optional per-stage HIP-event timing (bench.py).  Events are recorded on torch's current
    stream, the stream every pccx kernel is launched on."""

    def __init__(self):
        self.records = []

    def totals_ms(self):
        torch.cuda.synchronize()
        out = {}
        for name, a, b in self.records:
            t, n = out.get(name, (0.0, 0))
            out[name] = (t + a.elapsed_time(b), n + 1)
        return out


_timer = None


def set_timer(t):
    global _timer
    _timer = t


class stage:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if _timer is not None:
            self.a = torch.cuda.Event(enable_timing=True)
            self.a.record()

    def __exit__(self, *exc):
        if _timer is not None:
            b = torch.cuda.Event(enable_timing=True)
            b.record()
            _timer.records.append((self.name, self.a, b))
        return False


def _dev(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.PccxError(f"{name}: expected a tensor on the GPU (pccx has no CPU fallback)")
    return t


def _f32c(t, name):
    _dev(t, name)
    if t.dtype != torch.float32:
        raise _lib.PccxError(f"{name}: expected float32, got {t.dtype}")
    return t.contiguous()


def normalize(pc, margin=0.01):
    """pn_kit.normalize (pn_kit.py:47-60), batched over B: returns (pc, center (B,3), longest (B))."""
    pc = _f32c(pc, "normalize")
    B, N, _ = pc.shape
    out = torch.empty_like(pc)
    center = torch.empty(B, 3, device=pc.device, dtype=torch.float32)
    longest = torch.empty(B, device=pc.device, dtype=torch.float32)
    _lib.call("pccx_normalize", pc.data_ptr(), B, N, float(margin), out.data_ptr(), center.data_ptr(),
              longest.data_ptr(), _stream())
    return out, center, longest


def denormalize(pc, center, longest, margin=0.01):
    """pn_kit.denormalize (pn_kit.py:62-66), batched."""
    pc = _f32c(pc, "denormalize")
    B, N, _ = pc.shape
    center = _f32c(center.reshape(B, 3), "denormalize.center")
    longest = _f32c(longest.reshape(B), "denormalize.longest")
    out = torch.empty_like(pc)
    _lib.call("pccx_denormalize", pc.data_ptr(), B, N, float(margin), center.data_ptr(), longest.data_ptr(),
              out.data_ptr(), _stream())
    return out


def farthest_point_sample_batch(xyz, npoint, start_idx=None):
    """pn_kit.farthest_point_sample_batch (pn_kit.py:309-330).  ``start_idx`` (B,) replaces the
    reference's torch.randint draw (:321); None draws it the same way the reference does."""
    xyz = _f32c(xyz, "farthest_point_sample_batch")
    B, N, _ = xyz.shape
    if start_idx is None:
        start_idx = torch.randint(0, N, (B,), dtype=torch.long)
    # start_idx == "zero": every cloud starts from its point 0 (pytorch3d's sample_farthest_points); the kernel takes a null table for
    # that, so nothing is uploaded (a pageable upload blocks the calling thread behind everything queued on its stream)
    start = None if isinstance(start_idx, str) and start_idx == "zero" else torch.as_tensor(start_idx).to(device=xyz.device, dtype=torch.int32).contiguous()
    out = torch.empty(B, npoint, device=xyz.device, dtype=torch.int64)
    work = torch.empty(B * N, device=xyz.device, dtype=torch.float32) if N > 16384 else None
    _lib.call("pccx_fps", xyz.data_ptr(), B, N, int(npoint), start.data_ptr() if start is not None else None, out.data_ptr(),
              work.data_ptr() if work is not None else None, _stream())
    return out


def sample_farthest_points(xyz, K):
    """pytorch3d.ops.sample_farthest_points as pointnet_sa_module.py:12 uses it: start index 0,
    returns (points, idx)."""
    idx = farthest_point_sample_batch(xyz, K, start_idx="zero")
    return index_points(xyz, idx), idx


def index_points(points, idx):
    """pn_kit.index_points (pn_kit.py:332-360): idx (B,S) or (B,S,K)."""
    points = _f32c(points, "index_points")
    B, N, Cc = points.shape
    idx = _dev(idx, "index_points.idx").to(torch.int64).contiguous()
    M = idx[0].numel()
    out = torch.empty(B, M, Cc, device=points.device, dtype=torch.float32)
    _lib.call("pccx_gather", points.data_ptr(), B, N, Cc, idx.data_ptr(), M, out.data_ptr(), _stream())
    return out.view(*idx.shape, Cc)


def knn_gather(x, idx):
    """pytorch3d.ops.knn_gather(x (B,N,C), idx (B,M,K)) -> (B,M,K,C)."""
    return index_points(x, idx)


def knn_points(p1, p2, K, return_nn=True, patch_scale=0.0, return_dists=True, return_idx=True):
    """pytorch3d.ops.knn_points (compress.py:71, pn_kit.py:190).  With patch_scale != 0 the third
    field holds (nn - p1) * patch_scale, i.e. compress.py:72 and :108 fused.  return_dists / return_idx = False leave that field None
    and its bytes unwritten (KNN_Patching, compress.py:70-74, keeps the gathered points only)."""
    p1, p2 = _f32c(p1, "knn_points.p1"), _f32c(p2, "knn_points.p2")
    B, M, _ = p1.shape
    N = p2.shape[1]
    dists = torch.empty(B, M, K, device=p1.device, dtype=torch.float32) if return_dists else None
    idx = torch.empty(B, M, K, device=p1.device, dtype=torch.int64) if return_idx else None
    nn = torch.empty(B, M, K, 3, device=p1.device, dtype=torch.float32) if return_nn else None
    ptr = lambda t: t.data_ptr() if t is not None else None
    _lib.call("pccx_knn", p1.data_ptr(), B, M, p2.data_ptr(), N, int(K), ptr(dists), ptr(idx), ptr(nn), float(patch_scale), _stream())
    return KNN(dists, idx, nn)


def ball_query(p1, p2, K, radius, method="auto", extent=1.0):
    """pytorch3d.ops.ball_query (pointnet_sa_module.py:18): the first K candidates in index order with d2 < radius2, idx padded
    with -1.  method: "scan" = one wave per query walks the candidates in order and stops at the K-th hit; "grid" = uniform grid
    hash of the candidates (cells of side >= radius), 27-cell walk, index order restored by a bitmap; "auto" takes the grid for
    4096 <= N <= 32768 candidates when the clouds are at least four cells wide, extent / radius >= 4 (``extent`` = the callers'
    bound on a cloud's longest side: 1.0 for this codec's normalised clouds; narrower boxes put most of the cloud in the 27-cell
    walk and the ordered scan with its early exit at the K-th hit is faster -- as on the few-hundred-point sets of PPPF_AE).
    Same results either way."""
    p1, p2 = _f32c(p1, "ball_query.p1"), _f32c(p2, "ball_query.p2")
    B, M, _ = p1.shape
    N = p2.shape[1]
    dists = torch.empty(B, M, K, device=p1.device, dtype=torch.float32)
    idx = torch.empty(B, M, K, device=p1.device, dtype=torch.int64)
    if method == "grid" or (method == "auto" and 4096 <= N <= 32768 and float(extent) >= 4.0 * float(radius)):
        ws = torch.empty(_lib.load().pccx_ball_query_grid_workspace_ints(B, N), device=p1.device, dtype=torch.int32)
        _lib.call("pccx_ball_query_grid", p1.data_ptr(), B, M, p2.data_ptr(), N, int(K), float(radius), ws.data_ptr(),
                  dists.data_ptr(), idx.data_ptr(), _stream())
    else:
        _lib.call("pccx_ball_query", p1.data_ptr(), B, M, p2.data_ptr(), N, int(K), float(radius),
                  dists.data_ptr(), idx.data_ptr(), _stream())
    return KNN(dists, idx, None)


def nn_dist(x, y, return_idx=False):
    """min_j |x_i - y_j|^2 for every i: (B,P,3),(B,Q,3) -> (B,P) [, idx (B,P) int32]."""
    x, y = _f32c(x, "nn_dist.x"), _f32c(y, "nn_dist.y")
    B, P, _ = x.shape
    d2 = torch.empty(B, P, device=x.device, dtype=torch.float32)
    nn = torch.empty(B, P, device=x.device, dtype=torch.int32) if return_idx else None
    split = _lib.load().pccx_nn_dist_split_count(B, P, int(y.shape[1])) if B > 0 else 1
    if split > 1:
        # a batch too small to fill the chip (the training step's 4 clouds): the reference cloud in `split` chunks over more workgroups
        sd = torch.empty(split, B, P, device=x.device, dtype=torch.float32)
        sn = torch.empty(split, B, P, device=x.device, dtype=torch.int32) if return_idx else None
        _lib.call("pccx_nn_dist_split", x.data_ptr(), B, P, y.data_ptr(), y.shape[1], split, sd.data_ptr(),
                  sn.data_ptr() if sn is not None else None, d2.data_ptr(), nn.data_ptr() if nn is not None else None, _stream())
        return (d2, nn) if return_idx else d2
    _lib.call("pccx_nn_dist", x.data_ptr(), B, P, y.data_ptr(), y.shape[1], d2.data_ptr(),
              nn.data_ptr() if nn is not None else None, _stream())
    return (d2, nn) if return_idx else d2


def estimate_normals(xyz, knn=30):
    """open3d estimate_normals(KDTreeSearchParamKNN(knn)) (eval.py:59-60): unoriented PCA normals (B,N,3)."""
    xyz = _f32c(xyz, "estimate_normals")
    B, N, _ = xyz.shape
    idx = knn_points(xyz, xyz, min(knn, N), return_nn=False).idx
    out = torch.empty(B, N, 3, device=xyz.device, dtype=torch.float32)
    _lib.call("pccx_estimate_normals", xyz.data_ptr(), B, N, idx.data_ptr(), idx.shape[2], out.data_ptr(), _stream())
    return out


def point_plane_err(x, y, normals_y):
    """Squared projection of (x - nearest y) on that y's normal, eval.py:79-81: (B,P)."""
    x, y, normals_y = _f32c(x, "point_plane_err.x"), _f32c(y, "point_plane_err.y"), _f32c(normals_y, "point_plane_err.n")
    B, P, _ = x.shape
    _, nn = nn_dist(x, y, return_idx=True)
    err = torch.empty(B, P, device=x.device, dtype=torch.float32)
    _lib.call("pccx_point_plane_err", x.data_ptr(), B, P, y.data_ptr(), normals_y.data_ptr(), y.shape[1], nn.data_ptr(),
              err.data_ptr(), _stream())
    return err


zeros_hook = None       # pccx.train installs its arena's allocator here: (shape, dtype, device) -> (zero tensor, from-arena?)


class _ChamferFn(torch.autograd.Function):
    """Differentiable chamfer_distance (batch mean): forward = two nn_dist launches, backward =
    pccx_chamfer_grad with the argmins saved from the forward."""

    @staticmethod
    def forward(ctx, x, y):
        dxy, nxy = nn_dist(x, y, return_idx=True)
        dyx, nyx = nn_dist(y, x, return_idx=True)
        ctx.save_for_backward(x, y, nxy, nyx)
        out = torch.empty((), device=x.device, dtype=torch.float32)
        _lib.call("pccx_chamfer_mean", dxy.data_ptr(), dyx.data_ptr(), x.shape[0], x.shape[1], y.shape[1], out.data_ptr(), _stream())
        return out

    @staticmethod
    def backward(ctx, g):
        x, y, nxy, nyx = ctx.saved_tensors
        gd = g.detach().to(torch.float32).reshape(1).contiguous()        # stays on the device: no sync inside backward
        if zeros_hook is not None:          # inside a training step: the two gradients come cleared from the step's arena (train.StepArena)
            gx, gy = zeros_hook(tuple(x.shape), torch.float32, x.device)[0], zeros_hook(tuple(y.shape), torch.float32, y.device)[0]
            _lib.call("pccx_chamfer_grad_dev_acc", x.data_ptr(), x.shape[0], x.shape[1], y.data_ptr(), y.shape[1], nxy.data_ptr(),
                      nyx.data_ptr(), gd.data_ptr(), gx.data_ptr(), gy.data_ptr(), 4, _stream())
            return gx, gy
        gx, gy = torch.empty_like(x), torch.empty_like(y)
        _lib.call("pccx_chamfer_grad_dev", x.data_ptr(), x.shape[0], x.shape[1], y.data_ptr(), y.shape[1], nxy.data_ptr(),
                  nyx.data_ptr(), gd.data_ptr(), gx.data_ptr(), gy.data_ptr(), _stream())
        return gx, gy


class _STERound(torch.autograd.Function):
    """AE.STEQuantize (AE.py:72-85): forward x.round() (pccx_round: round half to even, as torch.round), backward the
    incoming gradient unchanged (straight-through estimator)."""

    @staticmethod
    def forward(ctx, x):
        xc = _f32c(x, "STEQuantize")
        y = torch.empty_like(xc)
        _lib.call("pccx_round", xc.data_ptr(), xc.numel(), y.data_ptr(), _stream())
        return y

    @staticmethod
    def backward(ctx, g):
        return g


def ste_round(x):
    """AE.STEQuantize.apply.  compress.py:127 hands over a HOST tensor (the script moved the latents to the CPU at :121):
    such an input is uploaded, rounded by the kernel and returned on the caller's device -- there is no CPU compute path."""
    if isinstance(x, torch.Tensor) and not x.is_cuda:
        return _STERound.apply(x.cuda()).to(x.device)
    return _STERound.apply(x)


def chamfer_distance(x, y, batch_reduction="mean"):
    """pytorch3d.loss.chamfer_distance defaults (AE.py:67, eval.py:204): squared distances,
    point mean, both directions summed; returns (value, None).  Differentiable w.r.t. x and y for
    batch_reduction="mean" (the loss of AE.py:57-70 / pppe_pcd_ae.py:817-838)."""
    if batch_reduction == "mean" and (x.requires_grad or y.requires_grad):
        return _ChamferFn.apply(_f32c(x, "chamfer.x"), _f32c(y, "chamfer.y")), None
    dxy, dyx = nn_dist(x, y), nn_dist(y, x)
    per = dxy.double().mean(dim=1) + dyx.double().mean(dim=1)
    if batch_reduction == "mean":
        per = per.mean()
    elif batch_reduction == "sum":
        per = per.sum()
    return per.float(), None


def octree_bits_capacity(S):
    """Longest possible stream of S centres in bits (1 + 8*S*16); the packed rows are (cap + 7) // 8 bytes."""
    return int(_lib.load().pccx_octree_bits_capacity(int(S)))


def octree_encode(centres, N, min_bpp, out_bytes=None, out_nbytes=None):
    """pn_kit.encode_sampled_np (pn_kit.py:380-401) + binary_array_to_byte_array (:463-467),
    batched on the GPU.  centres (B,S,3).  Returns dict of device tensors:
    bits (B,cap) u8 one byte per bit, nbits (B), depth (B), bytes (B,stride) u8, nbytes (B).
    out_bytes / out_nbytes: caller-provided dense destinations of that shape (codec.Compressed's packed buffer)."""
    centres = _f32c(centres, "octree_encode")
    B, S, _ = centres.shape
    cap = octree_bits_capacity(S)
    dev = centres.device
    if out_bytes is None:
        out_bytes = torch.empty(B, (cap + 7) // 8, device=dev, dtype=torch.uint8)
    if out_nbytes is None:
        out_nbytes = torch.empty(B, device=dev, dtype=torch.int32)
    if (tuple(out_bytes.shape) != (B, (cap + 7) // 8) or out_bytes.dtype != torch.uint8 or not out_bytes.is_contiguous()
            or tuple(out_nbytes.shape) != (B,) or out_nbytes.dtype != torch.int32 or not out_nbytes.is_contiguous()):
        raise _lib.PccxError("octree_encode: out_bytes must be dense (B, (cap+7)//8) u8 and out_nbytes dense (B,) i32")
    r = dict(bits=torch.empty(B, cap, device=dev, dtype=torch.uint8),
             nbits=torch.empty(B, device=dev, dtype=torch.int32),
             depth=torch.empty(B, device=dev, dtype=torch.int32),
             bytes=out_bytes, nbytes=out_nbytes)
    _lib.call("pccx_octree_encode", centres.data_ptr(), B, S, int(N), float(min_bpp), r["bits"].data_ptr(),
              r["nbits"].data_ptr(), r["depth"].data_ptr(), r["bytes"].data_ptr(), r["nbytes"].data_ptr(), _stream())
    return r


def octree_decode(bytes_, nbytes, mode="reference", S_out=64):
    """pn_kit.decode_sampled_np (pn_kit.py:424-431) from packed streams.  mode 'reference' is
    bug-compatible with octree_np.decode as written; 'full' is the level-by-level decode.
    bytes_ (B,stride) u8, nbytes (B) i32 -> (points (B,S_out,3), count (B))."""
    _dev(bytes_, "octree_decode")
    bytes_ = bytes_.contiguous()
    B, stride = bytes_.shape
    nbytes = _dev(nbytes, "octree_decode.nbytes").to(torch.int32).contiguous()
    out = torch.empty(B, S_out, 3, device=bytes_.device, dtype=torch.float32)
    count = torch.empty(B, device=bytes_.device, dtype=torch.int32)
    _lib.call("pccx_octree_decode", bytes_.data_ptr(), stride, nbytes.data_ptr(), B,
              {"reference": 0, "full": 1}[mode], int(S_out), out.data_ptr(), count.data_ptr(), _stream())
    return out, count
