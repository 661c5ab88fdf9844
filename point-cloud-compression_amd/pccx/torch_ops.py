"""``torch.ops.pccx.*`` -- the C ABI of libpccx.so registered with the PyTorch dispatcher ("a C-ABI HIP
library wrapped by TORCH_LIBRARY(pccx, ...) ops"; north_star: "called from Python via PyTorch-ROCm custom ops").

Registration is done from Python with ``torch.library`` on top of the same ctypes-bound entry points ``pccx.ops`` /
``pccx.models`` call directly, so there is ONE implementation per op:

  * every op has a schema, a CUDA(ROCm) kernel registration and a fake (meta) implementation, so it shows up in
    ``torch.ops.pccx``, works under FakeTensor / ``torch.compile`` tracing and is visible to the profiler;
  * ops with a derivative on the reference's training path register their autograd formula here
    (``chamfer_distance``: pccx_chamfer_grad; ``ste_round``: straight-through, AE.py:83-85);
  * there is no CPU kernel: a CPU tensor gets the dispatcher's "no kernel for backend CPU" error -- pccx has no CPU
    fallback.

Importing this module registers the ops (idempotent); ``pccx.ops`` stays the thin direct path the codec uses per launch
(the dispatcher adds tens of microseconds per call, which matters for the ~20 launches of a 60 ms batch step only at
small batches).
"""
import torch
from torch import Tensor

from . import ops as _ops

_DEV = "cuda"


def _op(name, mutates=()):
    return torch.library.custom_op(f"pccx::{name}", mutates_args=mutates, device_types=_DEV)


# ---- geometry -----------------------------------------------------------------------------------------------
@_op("normalize")
def normalize(pc: Tensor, margin: float = 0.01) -> tuple[Tensor, Tensor, Tensor]:
    """pn_kit.normalize (pn_kit.py:47-60): (B,N,3) -> (normalised (B,N,3), center (B,3), longest (B))."""
    return _ops.normalize(pc, margin)


@normalize.register_fake
def _(pc, margin=0.01):
    B = pc.shape[0]
    return torch.empty_like(pc), pc.new_empty(B, 3), pc.new_empty(B)


@_op("denormalize")
def denormalize(pc: Tensor, center: Tensor, longest: Tensor, margin: float = 0.01) -> Tensor:
    """pn_kit.denormalize (pn_kit.py:62-66)."""
    return _ops.denormalize(pc, center, longest, margin)


@denormalize.register_fake
def _(pc, center, longest, margin=0.01):
    return torch.empty_like(pc)


@_op("fps")
def fps(xyz: Tensor, npoint: int, start_idx: Tensor) -> Tensor:
    """pn_kit.farthest_point_sample_batch (pn_kit.py:309-330) with the start index explicit: -> (B,npoint) int64."""
    return _ops.farthest_point_sample_batch(xyz, npoint, start_idx)


@fps.register_fake
def _(xyz, npoint, start_idx):
    return xyz.new_empty(xyz.shape[0], npoint, dtype=torch.int64)


@_op("index_points")
def index_points(points: Tensor, idx: Tensor) -> Tensor:
    """pn_kit.index_points (pn_kit.py:332-360) / pytorch3d knn_gather."""
    return _ops.index_points(points, idx)


@index_points.register_fake
def _(points, idx):
    return points.new_empty(*idx.shape, points.shape[2])


@_op("knn_points")
def knn_points(p1: Tensor, p2: Tensor, K: int, patch_scale: float = 0.0) -> tuple[Tensor, Tensor, Tensor]:
    """pytorch3d.ops.knn_points(return_nn=True) (compress.py:71): -> (dists (B,M,K), idx (B,M,K) i64, nn (B,M,K,3));
    patch_scale != 0 returns (nn - p1) * patch_scale in the third field (compress.py:72,108 fused)."""
    r = _ops.knn_points(p1, p2, K, True, patch_scale)
    return r.dists, r.idx, r.knn


@knn_points.register_fake
def _(p1, p2, K, patch_scale=0.0):
    B, M, _ = p1.shape
    return p1.new_empty(B, M, K), p1.new_empty(B, M, K, dtype=torch.int64), p1.new_empty(B, M, K, 3)


@_op("ball_query")
def ball_query(p1: Tensor, p2: Tensor, K: int, radius: float) -> tuple[Tensor, Tensor]:
    """pytorch3d.ops.ball_query (pointnet_sa_module.py:18): -> (dists (B,M,K), idx (B,M,K) i64, -1 padded)."""
    r = _ops.ball_query(p1, p2, K, radius)
    return r.dists, r.idx


@ball_query.register_fake
def _(p1, p2, K, radius):
    B, M, _ = p1.shape
    return p1.new_empty(B, M, K), p1.new_empty(B, M, K, dtype=torch.int64)


@_op("nn_dist")
def nn_dist(x: Tensor, y: Tensor) -> tuple[Tensor, Tensor]:
    """min_j |x_i - y_j|^2 and its argmin: (B,P,3),(B,Q,3) -> ((B,P) f32, (B,P) i32) (eval.py:73-81, AE.py:67)."""
    return _ops.nn_dist(x, y, return_idx=True)


@nn_dist.register_fake
def _(x, y):
    B, P, _ = x.shape
    return x.new_empty(B, P), x.new_empty(B, P, dtype=torch.int32)


# ---- differentiable ops --------------------------------------------------------------------------------------
@_op("chamfer_distance")
def chamfer_distance(x: Tensor, y: Tensor) -> tuple[Tensor, Tensor, Tensor]:
    """pytorch3d.loss.chamfer_distance defaults (AE.py:67): -> (scalar loss, argmin x->y (B,P) i32, argmin y->x (B,Q) i32).
    The argmins are outputs so that the backward can reuse them."""
    dxy, nxy = _ops.nn_dist(x, y, return_idx=True)
    dyx, nyx = _ops.nn_dist(y, x, return_idx=True)
    return (dxy.double().mean(dim=1) + dyx.double().mean(dim=1)).mean().float(), nxy, nyx


@chamfer_distance.register_fake
def _(x, y):
    return x.new_empty(()), x.new_empty(x.shape[0], x.shape[1], dtype=torch.int32), x.new_empty(y.shape[0], y.shape[1], dtype=torch.int32)


@_op("chamfer_grad")
def chamfer_grad(x: Tensor, y: Tensor, nxy: Tensor, nyx: Tensor, g: float) -> tuple[Tensor, Tensor]:
    """d(chamfer)/dx, d(chamfer)/dy scaled by g (pccx_chamfer_grad)."""
    from . import _lib
    xc, yc = _ops._f32c(x, "chamfer_grad.x"), _ops._f32c(y, "chamfer_grad.y")
    gx, gy = torch.empty_like(xc), torch.empty_like(yc)
    _lib.call("pccx_chamfer_grad", xc.data_ptr(), xc.shape[0], xc.shape[1], yc.data_ptr(), yc.shape[1], nxy.contiguous().data_ptr(),
              nyx.contiguous().data_ptr(), float(g), gx.data_ptr(), gy.data_ptr(), _ops._stream())
    return gx, gy


@chamfer_grad.register_fake
def _(x, y, nxy, nyx, g):
    return torch.empty_like(x), torch.empty_like(y)


def _chamfer_setup(ctx, inputs, output):
    x, y = inputs
    _, nxy, nyx = output
    ctx.save_for_backward(x, y, nxy, nyx)


def _chamfer_backward(ctx, g, _g1, _g2):
    x, y, nxy, nyx = ctx.saved_tensors
    gx, gy = torch.ops.pccx.chamfer_grad(x, y, nxy, nyx, float(g))
    return gx, gy


chamfer_distance.register_autograd(_chamfer_backward, setup_context=_chamfer_setup)


@_op("ste_round")
def ste_round(x: Tensor) -> Tensor:
    """AE.STEQuantize (AE.py:72-85): round half to even; straight-through gradient."""
    return _ops._STERound.forward(None, x)


@ste_round.register_fake
def _(x):
    return torch.empty_like(x)


ste_round.register_autograd(lambda ctx, g: g)


# ---- octree serialisation --------------------------------------------------------------------------------------
@_op("octree_encode")
def octree_encode(centres: Tensor, N: int, min_bpp: float) -> tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
    """pn_kit.encode_sampled_np + binary_array_to_byte_array (pn_kit.py:380-401,463-467):
    -> (bits (B,cap) u8, nbits (B) i32, depth (B) i32, bytes (B,(cap+7)//8) u8, nbytes (B) i32)."""
    r = _ops.octree_encode(centres, N, min_bpp)
    return r["bits"], r["nbits"], r["depth"], r["bytes"], r["nbytes"]


@octree_encode.register_fake
def _(centres, N, min_bpp):
    B, S, _ = centres.shape
    cap = 1 + 8 * S * 16
    i32 = dict(dtype=torch.int32)
    return (centres.new_empty(B, cap, dtype=torch.uint8), centres.new_empty(B, **i32), centres.new_empty(B, **i32),
            centres.new_empty(B, (cap + 7) // 8, dtype=torch.uint8), centres.new_empty(B, **i32))


@_op("octree_decode")
def octree_decode(bytes_: Tensor, nbytes: Tensor, full_mode: bool, S_out: int) -> tuple[Tensor, Tensor]:
    """pn_kit.decode_sampled_np (pn_kit.py:424-431): full_mode False = octree_np.decode as written, True = level by level.
    -> (points (B,S_out,3), count (B) i32)."""
    return _ops.octree_decode(bytes_, nbytes, "full" if full_mode else "reference", S_out)


@octree_decode.register_fake
def _(bytes_, nbytes, full_mode, S_out):
    B = bytes_.shape[0]
    return bytes_.new_empty(B, S_out, 3, dtype=torch.float32), bytes_.new_empty(B, dtype=torch.int32)


# ---- entropy coder ---------------------------------------------------------------------------------------------
@_op("range_encode")
def range_encode(cdf_int: Tensor, latent_q: Tensor, L: int) -> tuple[Tensor, Tensor]:
    """torchac.encode_float_cdf on integer CDFs (compress.py:136): -> (bytes (B,cap) u8, nbytes (B) i32)."""
    from . import models
    return models.range_encode(cdf_int, latent_q, L)


@range_encode.register_fake
def _(cdf_int, latent_q, L):
    B = cdf_int.shape[0]
    nsym = cdf_int[0].numel() // (L + 1)
    return cdf_int.new_empty(B, 2 * nsym + 16, dtype=torch.uint8), cdf_int.new_empty(B, dtype=torch.int32)


@_op("range_decode")
def range_decode(cdf_int: Tensor, bytes_: Tensor, nbytes: Tensor, L: int) -> Tensor:
    """torchac.decode_float_cdf (decompress.py:93) -> latent_q (B,nsym) f32 (symbol - L//2)."""
    from . import models
    return models.range_decode(cdf_int, bytes_, nbytes, L)


@range_decode.register_fake
def _(cdf_int, bytes_, nbytes, L):
    return cdf_int.new_empty(cdf_int.shape[0], cdf_int[0].numel() // (L + 1), dtype=torch.float32)


# ---- the learned transforms, on packed weight blobs (models.AE.pack / ConditionalProbabilityModel.pack) -----------
@_op("sa_forward")
def sa_forward(patches: Tensor, enc_blob: Tensor) -> Tensor:
    """ae.sa (compress.py:113-115, pn_kit.py:164-211): patches (P,K,3) -> features (P,8,K,16) (16-channel groups per point)."""
    from . import _lib
    x = _ops._f32c(patches, "sa_forward")
    P, K, _ = x.shape
    feat = torch.empty(P, 8, K, 16, device=x.device, dtype=torch.float32)
    _lib.call("pccx_sa_forward", x.data_ptr(), P, K, enc_blob.data_ptr(), feat.data_ptr(), _ops._stream())
    return feat


@sa_forward.register_fake
def _(patches, enc_blob):
    return patches.new_empty(patches.shape[0], 8, patches.shape[1], 16)


@_op("pn_forward")
def pn_forward(patches: Tensor, feat: Tensor, enc_blob: Tensor, d: int, L: int) -> tuple[Tensor, Tensor, Tensor]:
    """ae.pn + sigmoid spread + round (compress.py:120-127): -> (latent_raw, latent, latent_quantized), each (P,d)."""
    from . import _lib
    x, f = _ops._f32c(patches, "pn_forward"), _ops._f32c(feat, "pn_forward.feat")
    P, K, _ = x.shape
    outs = [torch.empty(P, d, device=x.device, dtype=torch.float32) for _ in range(3)]
    _lib.call("pccx_pn_forward", x.data_ptr(), f.data_ptr(), P, K, enc_blob.data_ptr(), d, L, outs[0].data_ptr(),
              outs[1].data_ptr(), outs[2].data_ptr(), _ops._stream())
    return outs[0], outs[1], outs[2]


@pn_forward.register_fake
def _(patches, feat, enc_blob, d, L):
    P = patches.shape[0]
    return patches.new_empty(P, d), patches.new_empty(P, d), patches.new_empty(P, d)


@_op("ae_decode")
def ae_decode(latent_q: Tensor, dec_blob: Tensor, k: int) -> Tensor:
    """ae.inv_pool + ae.inv_mlp (decompress.py:97-102, AE.py:48-53): latent_q (P,d) -> patches (P,k,3)."""
    from . import _lib
    q = _ops._f32c(latent_q, "ae_decode")
    P, d = q.shape
    ws = torch.empty(_lib.load().pccx_ae_decode_workspace_floats(P), device=q.device, dtype=torch.float32)
    out = torch.empty(P, k, 3, device=q.device, dtype=torch.float32)
    _lib.call("pccx_ae_decode", q.data_ptr(), P, d, k, dec_blob.data_ptr(), ws.data_ptr(), out.data_ptr(),
              0.0, None, None, None, 1, 0.01, None, _ops._stream())
    return out


@ae_decode.register_fake
def _(latent_q, dec_blob, k):
    return latent_q.new_empty(latent_q.shape[0], k, 3)


@_op("prob_cdf")
def prob_cdf(sampled_xyz: Tensor, prob_blob: Tensor, d: int, L: int) -> tuple[Tensor, Tensor]:
    """ConditionalProbabilityModel + pmf_to_cdf + torchac's integer CDF (compress.py:131-134): (B,S,3) ->
    (pmf (B,S,d,L) f32, cdf_int (B,S,d,L+1) i32)."""
    from . import _lib
    x = _ops._f32c(sampled_xyz, "prob_cdf")
    B, S, _ = x.shape
    pmf = torch.empty(B, S, d, L, device=x.device, dtype=torch.float32)
    ci = torch.empty(B, S, d, L + 1, device=x.device, dtype=torch.int32)
    _lib.call("pccx_prob_forward", x.data_ptr(), B, S, d, L, prob_blob.data_ptr(), pmf.data_ptr(), None, ci.data_ptr(), _ops._stream())
    return pmf, ci


@prob_cdf.register_fake
def _(sampled_xyz, prob_blob, d, L):
    B, S, _ = sampled_xyz.shape
    return sampled_xyz.new_empty(B, S, d, L), sampled_xyz.new_empty(B, S, d, L + 1, dtype=torch.int32)


OPS = ("sa_forward", "pn_forward", "ae_decode", "prob_cdf", "normalize", "denormalize", "fps", "index_points", "knn_points", "ball_query", "nn_dist", "chamfer_distance",
       "chamfer_grad", "ste_round", "octree_encode", "octree_decode", "range_encode", "range_decode")
