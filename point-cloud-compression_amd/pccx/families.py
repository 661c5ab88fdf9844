"""The reference's other two model families over libpccx.so (inference forward passes):

  * PPPF_AE (PPPF_AE.py:114-150) = PointNet++ encoder built from PointnetSAModule
    (pointnet_sa_module.py:38-93: FPS from index 0, ball query, gather, Conv-BN-ReLU stack, max)
    + FoldingNet decoder (PPPF_AE.py:50-109)                          -- SURVEY 8a rows a18, a19
  * PointCloudAE of pppe_pcd_ae.py:843-877 (MSG + two SA levels with kNN grouping, global conv,
    quantize_st, PCN decoder)                                          -- SURVEY 8a row a21

Same constructor arguments and state_dict keys as the reference (torch modules are parameter
containers only).  Every layer is a HIP kernel behind the C ABI: selection ops from ops.py, the
Conv/Linear stacks through pccx_linear with eval-mode BatchNorm folded into weight and bias at pack
time, neighbour max through pccx_group_max.  Activations are kept channels-last ((rows, C)), so the
reference's permutes disappear; torch only concatenates and reshapes buffers.
These families are correctness-first (layer by layer through HBM); the fused, tuned path is the
IPDAE codec in models.py / codec.py.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from . import _lib, ops
from .ops import _stream, stage


# ---- f16x2 scales (csrc/planes.hip, "the same layers in the f16x2 arithmetic"; the rules of csrc/pack_h2.hip) -------------------------
def _pow2_floor(x):
    m, e = math.frexp(x)                                    # x = m 2^e, m in [0.5, 1)
    return math.ldexp(1.0, e - 1)


def h2_act_scale(bound):
    """the power of two sigma with bound * sigma <= 2^15 (half of fp16's largest number)"""
    return _pow2_floor(32768.0 / max(float(bound), 2.0 ** -40))


def h2_w_scale(W):
    """the largest power of two tau with max|W| tau <= 2^14: the lo piece of all but negligible weights stays a normal fp16 number"""
    m = float(np.abs(W).max()) if W.size else 0.0
    return _pow2_floor(16384.0 / m) if m > 0 else 1.0


def ibp_layer(W, b, lo, hi, relu):
    """Interval bounds of act(W x + b s) over x in [lo, hi] (per channel) and s in (0, 1] (the stack's dynamic normalisation multiplies
    the biases by a power of two s <= 1), in float64, inflated by 1e-3 for the fp32 evaluation of the kernels."""
    Wp, Wn = np.maximum(W, 0.0), np.minimum(W, 0.0)
    nhi = Wp @ hi + Wn @ lo + np.maximum(b, 0.0)
    nlo = Wp @ lo + Wn @ hi + np.minimum(b, 0.0)
    nhi, nlo = nhi + 1e-3 * np.abs(nhi) + 1e-30, nlo - 1e-3 * np.abs(nlo) - 1e-30
    if relu:
        nhi, nlo = np.maximum(nhi, 0.0), np.maximum(nlo, 0.0)
    return nlo, nhi


def h2_prepare_stack(stack, lo, hi):
    """Give every layer of a Conv/Linear stack its f16x2 operands for inputs within [lo, hi] per channel (the stack's NORMALISED input:
    magnitudes <= 1): sigma_l from the interval bound of the layer's input, tau_l from its weights, the two fp16 planes of tau W as
    the GEMM's weight stream, the bias as sigma tau b.  Returns the bounds of the stack's output."""
    lib = _lib.load()
    lo, hi = np.asarray(lo, np.float64), np.asarray(hi, np.float64)
    for l in stack:
        sig = h2_act_scale(max(float(np.abs(lo).max()), float(np.abs(hi).max())))
        tau = h2_w_scale(l.W_host)
        wp2 = torch.empty(lib.pccx_packed_linear_h2_floats(l.N, l.K), device=l.wp.device, dtype=torch.float32)
        _lib.call("pccx_pack_linear_h2", l.wp.data_ptr(), l.N, l.K, float(tau), wp2.data_ptr(), _stream())
        ws2 = torch.empty(lib.pccx_planes_gemm_weight_floats_h2(l.N, l.K), device=l.wp.device, dtype=torch.float32)
        _lib.call("pccx_pack_planes_gemm_h2", wp2.data_ptr(), l.N, l.K, ws2.data_ptr(), _stream())
        l.h2 = dict(sig=sig, tau=tau, ws=ws2, b=(l.b * float(sig * tau)).contiguous())
        lo, hi = ibp_layer(l.W_host, l.b_host, lo, hi, l.relu)
    return lo, hi


class FoldedLinear:
    """Conv1x1 / Linear (+ eval BatchNorm) (+ ReLU) packed for pccx_linear.  matmul: "f32" (exact-fp32 MFMA) or "bf16x3" (fp32
    products from three bf16 pieces per operand, pccx_linear_b3); None = pccx.DEFAULT_MATMUL at call time (the bf16 planes are
    built on the device the first time they are needed)."""

    def __init__(self, weight, bias, relu, bn=None, device="cuda", matmul=None):
        W = weight.detach().to("cpu", torch.float32).reshape(weight.shape[0], -1).clone()
        b = bias.detach().to("cpu", torch.float32).clone() if bias is not None else torch.zeros(W.shape[0])
        if bn is not None:                                  # y = (z - mean) / sqrt(var + eps) * gamma + beta
            scale = bn.weight.detach().cpu() / torch.sqrt(bn.running_var.detach().cpu() + bn.eps)
            W = W * scale[:, None]
            b = (b - bn.running_mean.detach().cpu()) * scale + bn.bias.detach().cpu()
        self.N, self.K, self.relu = W.shape[0], W.shape[1], int(bool(relu))
        W = W.contiguous()
        self.W_host, self.b_host, self.h2 = W.numpy().astype(np.float64), b.numpy().astype(np.float64), None   # for the f16x2 bounds
        wp = torch.zeros(_lib.load().pccx_packed_linear_floats(self.N, self.K), dtype=torch.float32)
        _lib.call("pccx_pack_linear", W.data_ptr(), self.N, self.K, wp.data_ptr())
        self.wp, self.b = wp.to(device), b.contiguous().to(device)
        self.matmul, self.wp3, self.ws3 = matmul, None, None

    def mode(self):
        from . import DEFAULT_MATMUL
        m = self.matmul or DEFAULT_MATMUL
        return "bf16x3" if m == "f16x2" else m              # a layer called by itself on fp32 rows has no f16x2 form: bf16x3

    def planes_mode(self):
        """the arithmetic of the layer as part of a planes stack: "f16x2" when asked for (the stacks of PPPF_AE.forward), else mode()"""
        from . import DEFAULT_MATMUL
        return self.matmul or DEFAULT_MATMUL

    def planes_h2(self, pin, M, epilogue=0, group=0, sig_next=None, dyn=None, amax=None, member=None):
        """planes() in the f16x2 arithmetic (h2_prepare_stack first): pin = f16x2 planes of sigma * input.  epilogue 0 -> f16x2 planes of
        sig_next * output; 1 / 2 -> fp32 rows / group maxima, un-scaled (and times dyn[1]); amax: 8 floats the row epilogue folds the
        largest |value| into.  member (M bytes, with epilogue 2): the maxima run over the rows marked 1 only."""
        h = self.h2
        scale = (float(sig_next) if epilogue == 0 else 1.0) / (h["sig"] * h["tau"])
        if member is not None:
            out = torch.empty(M // group, self.N, device=pin.device, dtype=torch.float32)
            _lib.call("pccx_planes_gemm_h2_member_max", pin.data_ptr(), M, self.K, h["ws"].data_ptr(), h["b"].data_ptr(), self.N, self.relu, group,
                      member.data_ptr(), scale, dyn.data_ptr() if dyn is not None else None, out.data_ptr(), self.N, _stream())
            return out
        if epilogue == 0:
            out = torch.empty(_lib.load().pccx_planes_floats_h2(M, self.N), device=pin.device, dtype=torch.float32)
        else:
            out = torch.empty(M if epilogue == 1 else M // group, self.N, device=pin.device, dtype=torch.float32)
        _lib.call("pccx_planes_gemm_h2", pin.data_ptr(), M, self.K, h["ws"].data_ptr(), h["b"].data_ptr(), self.N, self.relu, epilogue, group,
                  scale, dyn.data_ptr() if dyn is not None else None, amax.data_ptr() if amax is not None else None, out.data_ptr(), self.N,
                  _stream())
        return out

    def _planes3(self):
        if self.wp3 is None:
            self.wp3 = torch.empty(_lib.load().pccx_packed_linear_b3_floats(self.N, self.K), device=self.wp.device, dtype=torch.float32)
            _lib.call("pccx_pack_linear_b3", self.wp.data_ptr(), self.N, self.K, self.wp3.data_ptr(), _stream())
        return self.wp3

    def _stream3(self):
        if self.ws3 is None:
            self.ws3 = torch.empty(_lib.load().pccx_planes_gemm_weight_floats(self.N, self.K), device=self.wp.device, dtype=torch.float32)
            _lib.call("pccx_pack_planes_gemm", self._planes3().data_ptr(), self.N, self.K, self.ws3.data_ptr(), _stream())
        return self.ws3

    def planes_gather(self, src, C, idx, epilogue=0, group=0):
        """planes() on rows gathered inside the kernel: src (B, N, ldp) from padded_rows(), idx (B, M, ns) int64 (-1 -> row 0)."""
        B, Mq, ns = idx.shape
        rows = B * Mq * ns
        self._stream3()
        if epilogue == 0:
            out = torch.empty(_lib.load().pccx_planes_floats(rows, self.N), device=src.device, dtype=torch.float32)
        else:
            out = torch.empty(rows if epilogue == 1 else rows // group, self.N, device=src.device, dtype=torch.float32)
        _lib.call("pccx_planes_gemm_gather", src.data_ptr(), src.shape[2], idx.data_ptr(), Mq * ns, src.shape[1], rows, C,
                  self.ws3.data_ptr(), self.b.data_ptr(), self.N, self.relu, epilogue, group, out.data_ptr(), self.N, _stream())
        return out

    def planes(self, pin, M, epilogue=0, group=0):
        """The layer on an activation kept in planes (csrc/planes.hip; bf16x3 only): pin = planes of the (M, K) input.
        epilogue 0 -> planes of the (M, N) output, 1 -> fp32 rows (M, N), 2 -> (M // group, N) max over `group` consecutive rows."""
        self._stream3()
        if epilogue == 0:
            out = torch.empty(_lib.load().pccx_planes_floats(M, self.N), device=pin.device, dtype=torch.float32)
        else:
            out = torch.empty(M if epilogue == 1 else M // group, self.N, device=pin.device, dtype=torch.float32)
        _lib.call("pccx_planes_gemm", pin.data_ptr(), M, self.K, self.ws3.data_ptr(), self.b.data_ptr(), self.N, self.relu, epilogue,
                  group, out.data_ptr(), self.N, _stream())
        return out

    def __call__(self, x):
        """x (M,K) f32 contiguous on the GPU -> (M,N)."""
        from . import DEFAULT_MATMUL
        M = x.shape[0]
        out = torch.empty(M, self.N, device=x.device, dtype=torch.float32)
        if self.mode() == "bf16x3":
            _lib.call("pccx_linear_b3", x.data_ptr(), M, self.K, x.stride(0), self._planes3().data_ptr(), self.b.data_ptr(), self.N,
                      self.relu, out.data_ptr(), self.N, _stream())
            return out
        _lib.call("pccx_linear", x.data_ptr(), M, self.K, x.stride(0), self.wp.data_ptr(), self.b.data_ptr(), self.N,
                  self.relu, out.data_ptr(), self.N, _stream())
        return out


def cat_rows(parts):
    """torch.cat(parts, dim=-1) flattened to rows, written into a buffer whose row stride is a multiple of 4 floats: the layer
    kernels then take their 16-byte vector loads whatever the channel count (3, 131, 259, 1026 ...).  Returns the (rows, C) view."""
    lead = parts[0].shape[:-1]
    rows = 1
    for d_ in lead:
        rows *= int(d_)
    C = sum(int(p.shape[-1]) for p in parts)
    buf = torch.empty(rows, (C + 3) // 4 * 4, device=parts[0].device, dtype=torch.float32)
    off = 0
    for p in parts:
        c = int(p.shape[-1])
        buf[:, off:off + c].copy_(p.reshape(rows, c))
        off += c
    return buf[:, :C]


def group_planes(feats, xyz, idx):
    """index_points(feats, idx) ++ index_points(xyz, idx) (pointnet_sa_module.py:73-83; -1 -> row 0) as the planes of the first
    layer's operand.  feats (B,N,C) channels-last or None, xyz (B,N,3) or None, idx (B,M,ns) int64.  Returns (planes, rows)."""
    B, Mq, ns = idx.shape
    rows = B * Mq * ns
    f0 = feats.contiguous() if feats is not None else None
    f1 = xyz.contiguous() if xyz is not None else None
    C0 = int(f0.shape[-1]) if f0 is not None else 0
    C1 = int(f1.shape[-1]) if f1 is not None else 0
    n_src = int((f0 if f0 is not None else f1).shape[1])
    idx = idx.contiguous()
    out = torch.empty(_lib.load().pccx_planes_floats(rows, C0 + C1), device=idx.device, dtype=torch.float32)
    _lib.call("pccx_group_planes", f0.data_ptr() if f0 is not None else None, C0, C0, f1.data_ptr() if f1 is not None else None, C1, C1,
              idx.data_ptr(), rows, Mq * ns, n_src, out.data_ptr(), _stream())
    return out, rows


def rows_planes(x):
    """fp32 rows (M, K) (row stride >= K) -> planes."""
    M, K = x.shape
    out = torch.empty(_lib.load().pccx_planes_floats(M, K), device=x.device, dtype=torch.float32)
    _lib.call("pccx_group_planes", x.data_ptr(), K, x.stride(0), None, 0, 0, None, M, 1, 1, out.data_ptr(), _stream())
    return out


def chain4_fits(stack):
    """The width patterns pccx_planes_chain4 is built for (sa1 / sa2 of PPPF_AE.py:29-34), every layer with ReLU."""
    if len(stack) != 4 or not all(l.relu for l in stack) or any(stack[i + 1].K != stack[i].N for i in range(3)):
        return False
    n = [l.N for l in stack]
    return (n[0] <= 32 and 32 < n[1] <= 64 and 32 < n[2] <= 64 and 64 < n[3] <= 128) or \
           (all(96 < v <= 128 for v in n[:3]) and 128 < n[3] <= 256)


def _chain_args(stack, cache):
    if "ws" not in cache:
        cache["ws"] = torch.cat([l._stream3() for l in stack])
    a = []
    for l in stack:
        a += [l.b.data_ptr(), l.N]
    return cache["ws"], a


def padded_rows(feats, xyz):
    """[features, xyz] of every source point as fp32 rows zero padded to a multiple of 32 channels: what the gathering kernels read.
    Returns (src (B, N, ldp), C)."""
    parts = [p for p in (feats, xyz) if p is not None]
    C = sum(int(p.shape[-1]) for p in parts)
    B, n_src = int(parts[0].shape[0]), int(parts[0].shape[1])
    src = torch.zeros(B, n_src, (C + 31) // 32 * 32, device=parts[0].device, dtype=torch.float32)
    off = 0
    for p in parts:
        src[..., off:off + p.shape[-1]] = p
        off += int(p.shape[-1])
    return src, C


def wide3_fits(stack):
    """The first three layers fit pccx_planes_chain_wide (sa3 of PPPF_AE.py:32-34: widths 241..256, 241..256, 497..512, all ReLU, input
    of 8 or 9 blocks of 32 channels)."""
    if len(stack) < 3 or not all(l.relu for l in stack[:3]) or stack[1].K != stack[0].N or stack[2].K != stack[1].N:
        return False
    return 240 < stack[0].N <= 256 and 240 < stack[1].N <= 256 and 496 < stack[2].N <= 512 and (stack[0].K + 31) // 32 in (8, 9)


def wide3_planes(stack, src, C, idx, cache):
    """relu(L2(relu(L1(relu(L0(gathered rows)))))) as planes, one kernel (pccx_planes_chain_wide).  src (B, N, ldp) from padded_rows();
    idx (B, M, ns) int64 or None (then src is (rows, ldp) and every row is its own input)."""
    if "wide" not in cache:
        ws = torch.empty(_lib.load().pccx_planes_chain_wide_weight_floats(stack[0].K), device=src.device, dtype=torch.float32)
        _lib.call("pccx_pack_planes_chain_wide", stack[0]._planes3().data_ptr(), stack[1]._planes3().data_ptr(), stack[2]._planes3().data_ptr(),
                  stack[0].K, stack[0].N, stack[1].N, stack[2].N, ws.data_ptr(), _stream())
        cache["wide"] = ws
    if idx is not None:
        B, Mq, ns = idx.shape
        rows, rpb, n_src, ip = B * Mq * ns, Mq * ns, src.shape[1], idx.data_ptr()
    else:
        rows, rpb, n_src, ip = src.shape[0], 1, 1, None
    out = torch.empty(_lib.load().pccx_planes_floats(rows, stack[2].N), device=src.device, dtype=torch.float32)
    _lib.call("pccx_planes_chain_wide", src.data_ptr(), src.shape[-1], ip, rpb, n_src, rows, C, cache["wide"].data_ptr(),
              stack[0].b.data_ptr(), stack[0].N, stack[1].b.data_ptr(), stack[1].N, stack[2].b.data_ptr(), stack[2].N, out.data_ptr(), _stream())
    return out


def stack_max_gather(stack, feats, xyz, idx, cache):
    """index_points(feats, idx) ++ index_points(xyz, idx) -> Conv-BN-ReLU x 4 -> max over nsample (pointnet_sa_module.py:73-91) in one
    kernel for the stacks chain4_fits() accepts: the gather happens inside the kernel from the (B, N, C+3) rows zero padded to a
    multiple of 32 channels, so the grouped tensor never exists.  idx (B, M, ns) int64, -1 -> row 0.  Returns (B * M, N3)."""
    B, Mq, ns = idx.shape
    src, C = padded_rows(feats, xyz)
    idx = idx.contiguous()
    rows = B * Mq * ns
    if chain4_fits(stack):
        ws, a = _chain_args(stack, cache)
        out = torch.empty(B * Mq, stack[3].N, device=idx.device, dtype=torch.float32)
        _lib.call("pccx_planes_chain4_gather", src.data_ptr(), src.shape[2], idx.data_ptr(), Mq * ns, src.shape[1], rows, C, ws.data_ptr(),
                  *a, ns, out.data_ptr(), stack[3].N, _stream())
        return out
    if len(stack) == 4 and wide3_fits(stack):
        # three wide layers in one kernel (gather inside), then the last layer with the max in its epilogue
        return stack[3].planes(wide3_planes(stack, src, C, idx, cache), rows, 2, ns)
    # layer by layer: the first layer gathers, the last reduces
    pl = stack[0].planes_gather(src, C, idx) if len(stack) > 1 else None
    for layer in stack[1:-1]:
        pl = layer.planes(pl, rows, 0)
    if len(stack) > 1:
        return stack[-1].planes(pl, rows, 2, ns)
    return stack[0].planes_gather(src, C, idx, 2, ns)


def stack_max_planes(stack, pl, rows, group, cache):
    """Conv-BN-ReLU stack + max over `group` consecutive rows on planes: one kernel when the stack fits pccx_planes_chain4, else
    layer by layer with the max in the last layer's epilogue.  cache: a dict owned by the caller (holds the concatenated stream)."""
    if chain4_fits(stack):
        ws, a = _chain_args(stack, cache)
        out = torch.empty(rows // group, stack[3].N, device=pl.device, dtype=torch.float32)
        _lib.call("pccx_planes_chain4", pl.data_ptr(), rows, stack[0].K, ws.data_ptr(), *a, group, out.data_ptr(), stack[3].N, _stream())
        return out
    for layer in stack[:-1]:
        pl = layer.planes(pl, rows, 0)
    return stack[-1].planes(pl, rows, 2, group)


def fold_planes(a, mod0, b, div1, M):
    """torch.cat([a rows, b rows repeated], -1) as planes without building it: row r = a[r % mod0 if mod0 else r] ++ b[r // div1]
    (PPPF_AE.py:99-106).  a (.., C0), b (.., C1) fp32 rows."""
    a, b = a.contiguous(), b.contiguous()
    C0, C1 = int(a.shape[-1]), int(b.shape[-1])
    out = torch.empty(_lib.load().pccx_planes_floats(M, C0 + C1), device=a.device, dtype=torch.float32)
    _lib.call("pccx_fold_planes", a.data_ptr(), C0, C0, mod0, b.data_ptr(), C1, C1, div1, M, out.data_ptr(), _stream())
    return out


def run_stack_planes(stack, pl, M):
    """A Conv/Linear stack on an input given as planes -> fp32 rows (M, N_last)."""
    for layer in stack[:-1]:
        pl = layer.planes(pl, M, 0)
    return stack[-1].planes(pl, M, 1)


def run_stack(stack, x):
    """A Conv/Linear stack on fp32 rows x (M, K): layer by layer on rows (f32), or through planes (bf16x3)."""
    if stack and stack[0].mode() == "bf16x3" and x.shape[0] > 0:
        return run_stack_planes(stack, rows_planes(x), x.shape[0])
    for layer in stack:
        x = layer(x)
    return x


def rows_affine_small(base, div, x, mod, w_small, relu, M):
    """act(base[r // div] + x[r % mod if mod else r] @ w_small.T) for r < M (pccx_rows_affine_small): base (B, C), x (.., Ks <= 4)."""
    x = x.contiguous()
    Cc, Ks = int(base.shape[-1]), int(w_small.shape[1])
    out = torch.empty(M, Cc, device=base.device, dtype=torch.float32)
    _lib.call("pccx_rows_affine_small", base.contiguous().data_ptr(), Cc, int(div), x.data_ptr(), int(x.shape[-1]), Ks, int(mod),
              w_small.data_ptr(), int(bool(relu)), int(M), out.data_ptr(), _stream())
    return out


def rows_affine_planes(base, div, x, mod, w_small, relu, M):
    """rows_affine_small's values written as the operand planes of the next layer (pccx_rows_affine_planes): no fp32 rows in between."""
    x = x.contiguous()
    Cc, Ks = int(base.shape[-1]), int(w_small.shape[1])
    out = torch.empty(_lib.load().pccx_planes_floats(M, Cc), device=base.device, dtype=torch.float32)
    _lib.call("pccx_rows_affine_planes", base.contiguous().data_ptr(), Cc, int(div), x.data_ptr(), int(x.shape[-1]), Ks, int(mod),
              w_small.data_ptr(), int(bool(relu)), int(M), out.data_ptr(), _stream())
    return out


def rows_affine_planes_h2(base, div, x, mod, w_small, relu, M, rho, dyn):
    """rows_affine_planes in the f16x2 arithmetic: the planes of (rho * dyn[0]) * act(base[r // div] + x[..] @ w_small.T)"""
    x = x.contiguous()
    Cc, Ks = int(base.shape[-1]), int(w_small.shape[1])
    out = torch.empty(_lib.load().pccx_planes_floats_h2(M, Cc), device=base.device, dtype=torch.float32)
    _lib.call("pccx_rows_affine_planes_h2", base.contiguous().data_ptr(), Cc, int(div), x.data_ptr(), int(x.shape[-1]), Ks, int(mod),
              w_small.data_ptr(), int(bool(relu)), int(M), float(rho), dyn.data_ptr(), out.data_ptr(), _stream())
    return out


def run_stack_planes_h2(stack, pl, M, dyn, amax=None):
    """A Conv/Linear stack on f16x2 planes -> fp32 rows (M, N_last), un-scaled; amax: where the rows' largest |value| is recorded"""
    for i, layer in enumerate(stack[:-1]):
        pl = layer.planes_h2(pl, M, 0, sig_next=stack[i + 1].h2["sig"], dyn=dyn)
    return stack[-1].planes_h2(pl, M, 1, dyn=dyn, amax=amax)


def gather_max(y, idx):
    """max over nsample of y[b, idx.clamp(min=0)] (pointnet_sa_module.py:27-28,91): y (B,N,C) rows, idx (B,M,ns) int64 -> (B,M,C)."""
    B, N, Cc = y.shape
    _, M, ns = idx.shape
    out = torch.empty(B, M, Cc, device=y.device, dtype=torch.float32)
    _lib.call("pccx_gather_max", y.contiguous().data_ptr(), B, N, Cc, idx.contiguous().data_ptr(), M, ns, out.data_ptr(), _stream())
    return out


class PaddedRows:
    """The input rows of a set-abstraction level as the gathering planes kernels read them: src (B, N, ldp) fp32 with ldp = 32 * ceil((C + 3) / 32),
    row = [C features | xyz | zeros] (pointnet_sa_module.py:83: features first, xyz last)."""

    def __init__(self, src, C):
        self.src, self.C = src, int(C)


def gather_max_rows(y, idx, new_xyz):
    """gather_max whose output IS the next level's input rows (pccx_gather_max_rows): -> PaddedRows((B, M, ldp), C)"""
    B, N, Cc = y.shape
    _, M, ns = idx.shape
    ldp = (Cc + 3 + 31) // 32 * 32
    out = torch.empty(B, M, ldp, device=y.device, dtype=torch.float32)
    _lib.call("pccx_gather_max_rows", y.contiguous().data_ptr(), B, N, Cc, idx.contiguous().data_ptr(), M, ns, new_xyz.contiguous().data_ptr(),
              out.data_ptr(), ldp, _stream())
    return PaddedRows(out, Cc)


_IDENTITY = {}


def identity_index(P, n, device):
    """(P * n) int64: 0 .. n - 1 for each of the P batch elements (the gathering kernels' index of the rows as they stand)"""
    key = (int(P), int(n), str(device))
    t = _IDENTITY.get(key)
    if t is None:
        if len(_IDENTITY) > 8:
            _IDENTITY.clear()
        t = _IDENTITY[key] = torch.arange(n, device=device, dtype=torch.int64).repeat(P).contiguous()
    return t


def group_max(x):
    """(G,Kn,C) -> (G,C)."""
    G, Kn, Cc = x.shape
    out = torch.empty(G, Cc, device=x.device, dtype=torch.float32)
    _lib.call("pccx_group_max", x.contiguous().data_ptr(), G, Kn, Cc, out.data_ptr(), _stream())
    return out


def sigmoid_spread(x, L, do_round=False):
    y = torch.empty_like(x)
    _lib.call("pccx_sigmoid_spread", x.contiguous().data_ptr(), x.numel(), int(L), int(do_round), y.data_ptr(), _stream())
    return y


def round_(x):
    y = torch.empty_like(x)
    _lib.call("pccx_round", x.contiguous().data_ptr(), x.numel(), y.data_ptr(), _stream())
    return y


def _fold_stack(seq, device):
    """[Conv, (BN), (ReLU), ...] -> list of FoldedLinear."""
    mods = list(seq)
    out, i = [], 0
    while i < len(mods):
        conv = mods[i]
        i += 1
        bn = None
        if i < len(mods) and isinstance(mods[i], (nn.BatchNorm1d, nn.BatchNorm2d)):
            bn = mods[i]
            i += 1
        relu = i < len(mods) and isinstance(mods[i], nn.ReLU)
        if relu:
            i += 1
        out.append(FoldedLinear(conv.weight, conv.bias, relu, bn, device))
    return out


class _Packable(nn.Module):
    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self._packed = None
        return r


# =================================================================================================
# PPPF_AE
# =================================================================================================
class PointnetSAModule(nn.Module):                      # pointnet_sa_module.py:38-56
    def __init__(self, npoint, radius, nsample, mlp, use_xyz=True, in_channels=0):
        super().__init__()
        self.npoint, self.radius, self.nsample, self.use_xyz = npoint, radius, nsample, use_xyz
        last = in_channels + (3 if use_xyz else 0)
        layers = []
        for out in mlp:
            layers += [nn.Conv2d(last, out, 1), nn.BatchNorm2d(out), nn.ReLU(inplace=True)]
            last = out
        self.mlp = nn.Sequential(*layers)

    # The module gathers features and xyz UN-CENTRED (:73-85) and its Conv-BN(eval)-ReLU stack acts on each (group, sample) row by
    # itself, so every one of the npoint * nsample grouped rows is a copy of one of the N source rows: the stack is evaluated on
    # the N rows once and each group takes the maximum over its members from that result (pccx_gather_max) -- the same fp32 chain
    # per row, hence bit-identical outputs, for 1 / (npoint * nsample / N) of the matrix work (PPPF_AE: 16384 -> 512, 8192 -> 512,
    # 4096 -> 128 rows per patch).  dedup=False keeps the literal grouped evaluation (tests compare the two bit for bit).
    dedup = True
    # A level whose output is only ever reduced over ALL its centroids (PPPF_AE's third: PPPF_AE.py:44) needs no per-centroid maxima: the
    # maximum over the centroids of the maxima over their samples is the maximum over the source rows that are a sample of any centroid.
    # run_union_max() marks those rows (pccx_group_members) and takes that maximum in the last layer's epilogue -- neither the layer's
    # fp32 rows (1 GB per 2048 patches) nor the per-centroid table exist.  union_max=False keeps gather_max + group_max (tests compare).
    union_max = True
    # Between the f16x2 levels the maxima are written as the next level's padded input rows (pccx_gather_max_rows) and that level's first
    # kernel gathers them itself (the gathering forms of the chain / GEMM with an identity index): the operand-plane pass of levels 2 and 3
    # (pccx_group_planes_h2: 0.23 + 0.11 ms per 2048 patches) is not run.  Same planes, same results; False keeps the plane pass (tests compare).
    padded_levels = True

    def run_union_max(self, stack, xyz, feats, h2):
        """max over the npoint centroids of the level's output (B, C'), f16x2 arithmetic; needs N in {32, 64, 128} source rows per element"""
        B, N = xyz.shape[0], xyz.shape[1]
        with stage("fps"):
            new_xyz, _ = ops.sample_farthest_points(xyz, self.npoint)               # :66-68 (start index 0)
        with stage("ball_query"):
            idx = ops.ball_query(new_xyz, xyz, self.nsample, self.radius).idx       # :71 (-1 padded; gather clamps, :27)
            member = torch.empty(B * N, device=xyz.device, dtype=torch.uint8)
            _lib.call("pccx_group_members", idx.data_ptr(), idx.numel(), self.npoint * self.nsample, N, member.data_ptr(), _stream())
        with stage("sa_stack_%d" % stack[-1].N):
            dyn, _ = h2
            if isinstance(feats, PaddedRows):
                pl, first = self._h2_first_layer_gathered(stack, feats, dyn, stack[1].h2["sig"]), 1
            else:
                f2 = feats.reshape(-1, feats.shape[-1]).contiguous() if feats is not None else None
                x2 = xyz.reshape(-1, 3).contiguous()
                C0 = int(f2.shape[1]) if f2 is not None else 0
                pl, first = torch.empty(_lib.load().pccx_planes_floats_h2(B * N, C0 + 3), device=x2.device, dtype=torch.float32), 0
                _lib.call("pccx_group_planes_h2", f2.data_ptr() if f2 is not None else None, C0, C0, x2.data_ptr(), 3, 3, None, B * N, 1, 1,
                          float(stack[0].h2["sig"]), dyn.data_ptr(), pl.data_ptr(), _stream())
            for i, layer in enumerate(stack[first:-1], start=first):
                pl = layer.planes_h2(pl, B * N, 0, sig_next=stack[i + 1].h2["sig"], dyn=dyn)
            return stack[-1].planes_h2(pl, B * N, 2, group=N, dyn=dyn, member=member)

    def run(self, stack, xyz, feats, h2=None, pad_out=False):
        """xyz (B,N,3); feats (B,N,C) channels-last or None -> (new_xyz (B,M,3), feats (B,M,C')).
        h2 = (dyn, amax_out): evaluate the stack in the f16x2 arithmetic (h2_prepare_stack has run): dyn = the level's dynamic input
        normalisation {s, 1 / s} on the device, amax_out = 8 floats that receive the largest value of the stack's output.  With h2, feats
        may be a PaddedRows (the previous level's pad_out) and pad_out=True returns one: the maxima written as the NEXT level's input rows
        [features | new_xyz | 0], which its first kernel gathers itself -- no operand-plane pass between the levels."""
        B = xyz.shape[0]
        with stage("fps"):
            new_xyz, _ = ops.sample_farthest_points(xyz, self.npoint)               # :66-68 (start index 0)
        with stage("ball_query"):
            idx = ops.ball_query(new_xyz, xyz, self.nsample, self.radius).idx       # :71 (-1 padded; gather clamps, :27)
        if self.dedup and B > 0 and stack[-1].N % 4 == 0:
            return new_xyz, self._run_dedup(stack, xyz, feats, idx, B, h2, new_xyz if pad_out else None)
        if h2 is not None:
            raise _lib.PccxError("PointnetSAModule: the f16x2 stacks are built for the source-row evaluation (dedup=True)")
        return self._run_grouped(stack, xyz, feats, idx, new_xyz, B)

    def _h2_first_layer_gathered(self, stack, padded, dyn, sig_next):
        """the stack's first layer on PaddedRows through the gathering GEMM (identity index): -> planes of sig_next * output"""
        l0, src = stack[0], padded.src
        Bn, n_src, ldp = src.shape
        M = Bn * n_src
        out = torch.empty(_lib.load().pccx_planes_floats_h2(M, l0.N), device=src.device, dtype=torch.float32)
        _lib.call("pccx_planes_gemm_gather_h2", src.data_ptr(), ldp, identity_index(Bn, n_src, src.device).data_ptr(), n_src, n_src, M, l0.K,
                  l0.h2["ws"].data_ptr(), l0.h2["b"].data_ptr(), l0.N, l0.relu, 0, 0, float(l0.h2["sig"]),
                  float(sig_next) / (l0.h2["sig"] * l0.h2["tau"]), dyn.data_ptr(), None, out.data_ptr(), l0.N, _stream())
        return out

    def _run_dedup_h2(self, stack, f2, x2, rows_n, C0, h2, padded=None):
        """the stack on the source rows in the f16x2 arithmetic: rows -> planes of sigma_0 s [features, xyz], then ONE chain kernel (sa1 /
        sa2) or the layers one by one (sa3), the last with the row epilogue that un-scales and records the output's maximum.  padded: the
        rows as the previous level wrote them (PaddedRows): the first kernel gathers and splits them itself, no planes pass."""
        dyn, amax_out = h2
        lib = _lib.load()
        ap = amax_out.data_ptr() if amax_out is not None else None
        if padded is not None and chain4_fits(stack):
            ws, sc, a = self._chain2_args(stack)
            src = padded.src
            y = torch.empty(rows_n, stack[3].N, device=src.device, dtype=torch.float32)
            _lib.call("pccx_planes_chain4_gather_h2", src.data_ptr(), src.shape[2], identity_index(src.shape[0], src.shape[1], src.device).data_ptr(),
                      src.shape[1], src.shape[1], rows_n, stack[0].K, ws.data_ptr(), *a, 1, sc.ctypes.data, dyn.data_ptr(), ap, y.data_ptr(),
                      stack[3].N, _stream())
            return y
        if padded is not None:
            pl = self._h2_first_layer_gathered(stack, padded, dyn, stack[1].h2["sig"])
            for i, layer in enumerate(stack[1:-1], start=1):
                pl = layer.planes_h2(pl, rows_n, 0, sig_next=stack[i + 1].h2["sig"], dyn=dyn)
            return stack[-1].planes_h2(pl, rows_n, 1, dyn=dyn, amax=amax_out)
        pl = torch.empty(lib.pccx_planes_floats_h2(rows_n, C0 + 3), device=x2.device, dtype=torch.float32)
        _lib.call("pccx_group_planes_h2", f2.data_ptr() if f2 is not None else None, C0, C0, x2.data_ptr(), 3, 3, None, rows_n, 1, 1,
                  float(stack[0].h2["sig"]), dyn.data_ptr(), pl.data_ptr(), _stream())
        if chain4_fits(stack):
            ws, sc, a = self._chain2_args(stack)
            y = torch.empty(rows_n, stack[3].N, device=x2.device, dtype=torch.float32)
            _lib.call("pccx_planes_chain4_h2", pl.data_ptr(), rows_n, stack[0].K, ws.data_ptr(), *a, 1, sc.ctypes.data, dyn.data_ptr(), ap,
                      y.data_ptr(), stack[3].N, _stream())
            return y
        for i, layer in enumerate(stack[:-1]):
            pl = layer.planes_h2(pl, rows_n, 0, sig_next=stack[i + 1].h2["sig"], dyn=dyn)
        return stack[-1].planes_h2(pl, rows_n, 1, dyn=dyn, amax=amax_out)

    def _chain2_args(self, stack):
        """the f16x2 chain's weight stream, its five scales and its (bias, width) arguments, once per pack"""
        if getattr(self, "_chain2_of", None) is not stack:
            h = [l.h2 for l in stack]
            sc = np.array([h[0]["sig"]] + [h[i]["sig"] / (h[i - 1]["sig"] * h[i - 1]["tau"]) for i in (1, 2, 3)] +
                          [1.0 / (h[3]["sig"] * h[3]["tau"])], dtype=np.float32)
            self._chain2_of, self._chain2 = stack, (torch.cat([x["ws"] for x in h]), sc)
        ws, sc = self._chain2
        a = []
        for l in stack:
            a += [l.h2["b"].data_ptr(), l.N]
        return ws, sc, a

    def _run_dedup(self, stack, xyz, feats, idx, B, h2=None, pad_xyz=None):
        with stage("sa_stack_%d" % stack[-1].N):
            if h2 is not None and isinstance(feats, PaddedRows):
                y = self._run_dedup_h2(stack, None, None, B * xyz.shape[1], feats.C, h2, padded=feats)
            elif h2 is not None:
                f2 = feats.reshape(-1, feats.shape[-1]).contiguous() if feats is not None else None
                x2 = xyz.reshape(-1, 3).contiguous()
                y = self._run_dedup_h2(stack, f2, x2, x2.shape[0], int(f2.shape[1]) if f2 is not None else 0, h2)
            elif stack[0].mode() == "bf16x3":
                # :83 features first, xyz last, one row per SOURCE point -- split straight into the first layer's operand planes from the
                # two tables (pccx_group_planes without indices): the concatenated rows (two torch copies per level in round 3) never exist
                f2 = feats.reshape(-1, feats.shape[-1]).contiguous() if feats is not None else None
                x2 = xyz.reshape(-1, 3).contiguous()
                rows_n = x2.shape[0]
                C0 = int(f2.shape[1]) if f2 is not None else 0
                pl = torch.empty(_lib.load().pccx_planes_floats(rows_n, C0 + 3), device=xyz.device, dtype=torch.float32)
                _lib.call("pccx_group_planes", f2.data_ptr() if f2 is not None else None, C0, C0, x2.data_ptr(), 3, 3, None, rows_n, 1, 1,
                          pl.data_ptr(), _stream())
                if chain4_fits(stack):
                    # :90 Conv-BN-ReLU x 4 on the source rows in ONE kernel (the chain of planes.hip with its row epilogue, group = 1)
                    if getattr(self, "_chain_of", None) is not stack:               # new pack -> new stream
                        self._chain_of, self._chain_cache = stack, {}
                    ws, a = _chain_args(stack, self._chain_cache)
                    y = torch.empty(rows_n, stack[3].N, device=xyz.device, dtype=torch.float32)
                    _lib.call("pccx_planes_chain4", pl.data_ptr(), rows_n, stack[0].K, ws.data_ptr(), *a, 1, y.data_ptr(), stack[3].N, _stream())
                else:
                    y = run_stack_planes(stack, pl, rows_n)                         # :90 Conv-BN-ReLU, (B * N, C) rows
            else:
                rows = cat_rows([feats, xyz] if feats is not None else [xyz])
                y = run_stack(stack, rows)
        with stage("gather_max"):
            if pad_xyz is not None:
                return gather_max_rows(y.view(B, xyz.shape[1], -1), idx, pad_xyz)   # :91, written as the next level's input rows
            return gather_max(y.view(B, xyz.shape[1], -1), idx)                     # :91 max over the group's members

    def _run_grouped(self, stack, xyz, feats, idx, new_xyz, B):
        if stack[0].mode() == "bf16x3" and self.nsample in (32, 64, 128) and B > 0:
            # gather + concat + split in one pass, every layer on planes, the max over nsample in the last layer's epilogue
            if getattr(self, "_chain_of", None) is not stack:                       # new pack -> new stream
                self._chain_of, self._chain_cache = stack, {}
            # :73-91 gather (features first, xyz last, not centred) inside the first kernel, max over nsample in the last
            return new_xyz, stack_max_gather(stack, feats, xyz, idx, self._chain_cache).view(B, self.npoint, -1)
        grouped = ops.index_points(xyz, idx)                                        # :81 (not centred)
        x = cat_rows([ops.index_points(feats, idx), grouped] if feats is not None else [grouped])   # :83 features first, xyz last
        for layer in stack:
            x = layer(x)                                                            # :90 Conv-BN-ReLU
        return new_xyz, group_max(x.view(B * self.npoint, self.nsample, -1)).view(B, self.npoint, -1)   # :91


class PointNetPP(nn.Module):                            # PPPF_AE.py:9-46
    def __init__(self, points=512, sa1_mlp=(64, 64, 128), sa2_mlp=(128, 128, 128, 256), sa3_mlp=(256, 256, 512),
                 feature_dim=1024):
        super().__init__()
        self.sa1 = PointnetSAModule(points, 0.2, 32, [3] + list(sa1_mlp), True, 0)
        self.sa2 = PointnetSAModule(128, 0.4, 64, list(sa2_mlp), True, 128)
        self.sa3 = PointnetSAModule(32, 0.8, 128, list(sa3_mlp) + [feature_dim], True, 256)


class FoldingNet(nn.Module):                            # PPPF_AE.py:50-80
    def __init__(self, points=512, grid_size=45, feature_dim=1024):
        super().__init__()
        self.grid_size, self.num_points = grid_size, grid_size * grid_size
        self.mlp1 = nn.Sequential(nn.Conv1d(feature_dim + 2, points, 1), nn.ReLU(), nn.Conv1d(points, points, 1), nn.ReLU(),
                                  nn.Conv1d(points, 3, 1))
        self.mlp2 = nn.Sequential(nn.Conv1d(feature_dim + 3, 128, 1), nn.ReLU(), nn.Conv1d(128, 128, 1), nn.ReLU(),
                                  nn.Conv1d(128, 3, 1))


class PPPF_AE(_Packable):
    """PPPF_AE.PPPF_AE (PPPF_AE.py:114-150)."""

    split_fold = True       # FoldingNet's first layers evaluated as per-patch + per-point parts (forward()); False = the literal rows

    def __init__(self, K=512, k=0, d=16, L=7, dim=1024):
        super().__init__()
        self.L, self.d, self.dim = L, d, dim
        self.encoder = PointNetPP(points=K, feature_dim=dim)
        self.decoder = FoldingNet(points=K, grid_size=d)
        self.enc_proj = nn.Linear(dim, d)
        self.dec_proj = nn.Linear(d, dim)
        self._packed = None

    def pack(self, device="cuda"):
        e, dcd = self.encoder, self.decoder
        self._packed = dict(sa=[_fold_stack(m.mlp, device) for m in (e.sa1, e.sa2, e.sa3)],
                            enc=FoldedLinear(self.enc_proj.weight, self.enc_proj.bias, False, None, device),
                            dec=FoldedLinear(self.dec_proj.weight, self.dec_proj.bias, False, None, device),
                            mlp1=_fold_stack(dcd.mlp1, device), mlp2=_fold_stack(dcd.mlp2, device))
        # first layers of the two folding MLPs split into their per-patch (latent) and per-point (grid / coarse) parts, see forward()
        for name, seq, ks in (("mlp1", dcd.mlp1, 2), ("mlp2", dcd.mlp2, 3)):
            w = seq[0].weight.detach().to("cpu", torch.float32).reshape(seq[0].weight.shape[0], -1)
            self._packed[name + "_lat"] = FoldedLinear(w[:, ks:], seq[0].bias, False, None, device)
            self._packed[name + "_small"] = w[:, :ks].contiguous().to(device)
        x = torch.linspace(-1, 1, dcd.grid_size)
        gx, gy = torch.meshgrid(x, x, indexing="ij")
        self._packed["grid"] = torch.stack([gx, gy], dim=-1).reshape(-1, 2).to(device)            # :82-88
        return self

    def _ensure_h2(self, device):
        """The f16x2 operands of the five planes stacks (three set-abstraction levels, the two FoldingNet chains), once per pack.  Every
        stack is bounded for a NORMALISED input -- features (post-ReLU maxima) in [0, 1], coordinates in [-1, 1] -- which forward()
        establishes per call from the data (pccx_absmax / pccx_dyn_scale: a power of two s <= 1 per stack, biases times s, outputs
        times 1 / s; Conv / ReLU stacks are positively homogeneous), so the interval bounds are rigorous whatever the input and restart
        at every stack: they stay within a few layers' looseness of the values (the lo pieces keep their bits)."""
        pk = self._packed
        if "h2" in pk:
            return pk["h2"]
        c_prev = 0
        for stack in pk["sa"]:
            h2_prepare_stack(stack, np.concatenate([np.zeros(c_prev), -np.ones(3)]), np.ones(c_prev + 3))   # :83 features first, xyz last
            c_prev = stack[-1].N
        for name in ("mlp1", "mlp2"):
            first, rest = pk[name][0], pk[name][1:]
            h2_prepare_stack(rest, np.zeros(rest[0].K) if first.relu else -np.ones(rest[0].K), np.ones(rest[0].K))
        wsum = lambda w: float(np.abs(w.detach().cpu().numpy().astype(np.float64)).sum(axis=1).max() * 1.001)
        pk["h2"] = dict(amax=torch.zeros(6 * 8, device=device, dtype=torch.float32), dyn=torch.ones(5 * 2, device=device, dtype=torch.float32),
                        wsum1=wsum(pk["mlp1_small"]), wsum2=wsum(pk["mlp2_small"]))
        return pk["h2"]

    def forward(self, xyz):
        """xyz (B,N,3) on the GPU -> (recon (B,d*d,3), latent (B,dim), latent_quantized (B,d))."""
        if self._packed is None:
            self.pack(xyz.device)
        pk = self._packed
        B = xyz.shape[0]
        pts, feats = ops._f32c(xyz, "PPPF_AE"), None
        pow2x4 = lambda n: n % 4 == 0 and n <= 1024 and (n // 4) & (n // 4 - 1) == 0     # widths pccx_rows_affine_small takes
        h2 = None
        if (B > 0 and pk["sa"][0][0].planes_mode() == "f16x2" and PointnetSAModule.dedup and self.split_fold
                and all(st[-1].N % 4 == 0 for st in pk["sa"]) and pow2x4(pk["mlp1"][0].N) and pow2x4(pk["mlp2"][0].N)):
            h2 = self._ensure_h2(pts.device)
            am, dy = (lambda i: h2["amax"][8 * i:8 * i + 8]), (lambda i: h2["dyn"][2 * i:2 * i + 2])
            scale_of = lambda m1, a1, m2, a2, add, comb, out: _lib.call(
                "pccx_dyn_scale", m1.data_ptr(), float(a1), m2.data_ptr() if m2 is not None else None, float(a2), float(add), int(comb),
                out.data_ptr(), _stream())
            absmax = lambda t, slot: _lib.call("pccx_absmax", t.data_ptr(), t.numel(), slot.data_ptr(), _stream())
            _lib.call("pccx_zero_bytes", h2["amax"].data_ptr(), h2["amax"].numel() * 4, _stream())
            absmax(pts, am(0))                                                      # max |coordinate|: every level's centroids are a subset
        for lvl, (mod, stack) in enumerate(zip((self.encoder.sa1, self.encoder.sa2, self.encoder.sa3), pk["sa"])):
            if h2 is None:
                pts, feats = mod.run(stack, pts, feats)
                continue
            # the level's input = [maxima of the previous level's output rows, coordinates]: s from the larger of the two bounds
            if lvl == 0:
                scale_of(am(0), 1.0, None, 0.0, 0.0, 1, dy(0))
            else:
                scale_of(am(lvl), 1.0, am(0), 1.0, 0.0, 1, dy(lvl))
            if (lvl == 2 and mod.union_max and pts.shape[1] in (32, 64, 128) and stack[-1].relu and not chain4_fits(stack)):
                feats = mod.run_union_max(stack, pts, feats, (dy(lvl), None))       # (B, dim): :44 and :91 of the level in one epilogue
                continue
            pts, feats = mod.run(stack, pts, feats, h2=(dy(lvl), am(lvl + 1) if lvl < 2 else None),
                                 pad_out=lvl < 2 and PointnetSAModule.padded_levels)
        with stage("latent"):
            g = group_max(feats) if feats.dim() == 3 else feats                     # :44 max over the 32 points
            latent = sigmoid_spread(g, self.L)                                      # :136-137
            q = round_(pk["enc"](latent))                                           # :139-142
            lat_dec = pk["dec"](q)                                                  # :145
        P = self.decoder.num_points
        if h2 is not None:
            # the split folding form (below) in the f16x2 arithmetic.  A chain's input relu(base + point part) is bounded from the data:
            # max |base| (+ the per-point update's largest row sum times max |point input|: the grid lies in [-1, 1], the coarse points'
            # maximum comes out of the first chain's row epilogue)
            with stage("fold_mlp1"):
                base1 = pk["mlp1_lat"](lat_dec)
                absmax(base1, am(3))
                scale_of(am(3), 1.0, None, 0.0, h2["wsum1"], 0, dy(3))
                st = pk["mlp1"][1:]
                pl = rows_affine_planes_h2(base1, P, pk["grid"], P, pk["mlp1_small"], pk["mlp1"][0].relu, B * P, st[0].h2["sig"], dy(3))
                x = run_stack_planes_h2(st, pl, B * P, dy(3), am(4))                                         # :104 coarse
            with stage("fold_mlp2"):
                base2 = pk["mlp2_lat"](lat_dec)
                absmax(base2, am(5))
                scale_of(am(5), 1.0, am(4), h2["wsum2"], 0.0, 0, dy(4))
                st = pk["mlp2"][1:]
                pl = rows_affine_planes_h2(base2, P, x, 0, pk["mlp2_small"], pk["mlp2"][0].relu, B * P, st[0].h2["sig"], dy(4))
                x = run_stack_planes_h2(st, pl, B * P, dy(4))                                                # :107 fine
            return x.view(B, P, 3), latent, q
        if self.split_fold and B > 0 and pow2x4(pk["mlp1"][0].N) and pow2x4(pk["mlp2"][0].N):
            # The folding inputs [grid | latent] and [coarse | latent] (:99-106) are never built: their 1024-wide latent part is the
            # same for the P points of a patch, so the first layer of each MLP is W_lat latent + bias once per PATCH (a Linear on B
            # rows) plus a 2- / 3-term per-point update with ReLU (pccx_rows_affine_small); the remaining layers run on the P rows.
            if pk["mlp1"][1].mode() == "bf16x3":
                # ... and the per-point update writes the next layer's operand PLANES directly (the fp32 rows of the 512-wide MLP were 1 GB
                # written, read back and split per 2048 patches)
                with stage("fold_mlp1"):
                    pl = rows_affine_planes(pk["mlp1_lat"](lat_dec), P, pk["grid"], P, pk["mlp1_small"], pk["mlp1"][0].relu, B * P)
                    x = run_stack_planes(pk["mlp1"][1:], pl, B * P)                                          # :104 coarse
                with stage("fold_mlp2"):
                    pl = rows_affine_planes(pk["mlp2_lat"](lat_dec), P, x, 0, pk["mlp2_small"], pk["mlp2"][0].relu, B * P)
                    x = run_stack_planes(pk["mlp2"][1:], pl, B * P)                                          # :107 fine
                return x.view(B, P, 3), latent, q
            h = rows_affine_small(pk["mlp1_lat"](lat_dec), P, pk["grid"], P, pk["mlp1_small"], pk["mlp1"][0].relu, B * P)
            x = run_stack(pk["mlp1"][1:], h)                                                                 # :104 coarse
            h = rows_affine_small(pk["mlp2_lat"](lat_dec), P, x, 0, pk["mlp2_small"], pk["mlp2"][0].relu, B * P)
            x = run_stack(pk["mlp2"][1:], h)                                                                 # :107 fine
            return x.view(B, P, 3), latent, q
        if pk["mlp1"][0].mode() == "bf16x3" and B > 0:
            # the literal form on operand planes: nothing is concatenated or repeated in memory either
            x = run_stack_planes(pk["mlp1"], fold_planes(pk["grid"], P, lat_dec, P, B * P), B * P)          # :104 coarse
            x = run_stack_planes(pk["mlp2"], fold_planes(x, 0, lat_dec, P, B * P), B * P)                   # :107 fine
            return x.view(B, P, 3), latent, q
        rep = lat_dec[:, None, :].expand(B, P, self.dim)
        x = cat_rows([pk["grid"][None].expand(B, P, 2), rep])                                             # :99-101
        x = run_stack(pk["mlp1"], x)                                                # :104 coarse
        x = cat_rows([x.view(B, P, 3), rep])                                                                # :106
        x = run_stack(pk["mlp2"], x)                                                # :107 fine
        return x.view(B, P, 3), latent, q


def pppf_flops_per_patch(model, executed=False, n_points=512):
    """FLOPs (2 * MACs of every Conv / Linear, rows x in x out) of one PPPF_AE forward on one patch of n_points points.
    executed=False: as the reference evaluates it, every set-abstraction stack on its npoint x nsample grouped rows
    (pointnet_sa_module.py:86-90).  executed=True: what this implementation runs -- the stacks on the source rows only
    (PointnetSAModule.dedup), everything else unchanged."""
    if model._packed is None:
        raise _lib.PccxError("pppf_flops_per_patch: pack() the model first")
    pk, e = model._packed, model.encoder
    macs, n_src = 0, n_points
    for mod, stack in zip((e.sa1, e.sa2, e.sa3), pk["sa"]):
        rows = n_src if (executed and mod.dedup) else mod.npoint * mod.nsample
        macs += sum(rows * l.N * l.K for l in stack)
        n_src = mod.npoint
    macs += pk["enc"].N * pk["enc"].K + pk["dec"].N * pk["dec"].K
    P = model.decoder.num_points
    for name, ks in (("mlp1", 2), ("mlp2", 3)):
        st = pk[name]
        if executed and model.split_fold:
            macs += st[0].N * (st[0].K - ks) + P * st[0].N * ks + sum(P * l.N * l.K for l in st[1:])
        else:
            macs += sum(P * l.N * l.K for l in st)
    return 2 * macs


# =================================================================================================
# pppe PointCloudAE
# =================================================================================================
def _c2(in_c, out_c):
    return nn.Sequential(nn.Conv2d(in_c, out_c, 1, bias=False), nn.BatchNorm2d(out_c), nn.ReLU(inplace=True))


class PointNetSetAbstraction(nn.Module):                # pppe_pcd_ae.py:573-611
    def __init__(self, npoint, K, in_channel, mlp):
        super().__init__()
        self.npoint, self.K = npoint, K
        last = in_channel + 3
        layers = []
        for out in mlp:
            layers.append(_c2(last, out))
            last = out
        self.mlp_stack = nn.ModuleList(layers)

    def run(self, stack, xyz, feats, start):
        B, N, _ = xyz.shape
        S = self.npoint
        new_xyz = xyz if S == N else ops.index_points(xyz, ops.farthest_point_sample_batch(xyz, S, start))   # :593-597
        nn_ = ops.knn_points(new_xyz, xyz, self.K, patch_scale=1.0)                  # :599-600 (nn - centre) * 1
        x = cat_rows([nn_.knn, ops.index_points(feats, nn_.idx)] if feats is not None else [nn_.knn])   # :606 xyz first
        for layer in stack:
            x = layer(x)
        return new_xyz, group_max(x.view(B * S, self.K, -1)).view(B, S, -1)          # :610


class PointNetSetAbstractionMSG(nn.Module):             # pppe_pcd_ae.py:614-632
    def __init__(self, npoint, scales, in_channel):
        super().__init__()
        self.branches = nn.ModuleList([PointNetSetAbstraction(npoint, s["K"], in_channel, s["mlp"]) for s in scales])


class PointNet2EncoderFull(nn.Module):                  # pppe_pcd_ae.py:637-667
    def __init__(self, latent_dim=256):
        super().__init__()
        self.sa_modules = nn.ModuleList([
            PointNetSetAbstractionMSG(512, [{"K": 16, "mlp": [32, 32, 64]}, {"K": 32, "mlp": [64, 64, 128]}], 0),
            PointNetSetAbstraction(128, 32, 64 + 128, [128, 128, 256]),
            PointNetSetAbstraction(32, 32, 256, [256, 256, 512])])
        self.global_conv = nn.Sequential(nn.Conv1d(512, 512, 1, bias=False), nn.BatchNorm1d(512), nn.ReLU(inplace=True),
                                         nn.Conv1d(512, latent_dim, 1))


class PCNDecoderSmall(nn.Module):                       # pppe_pcd_ae.py:691-707
    def __init__(self, latent_dim=256, coarse_points=512, final_points=8192):
        super().__init__()
        self.fc_coarse = nn.Sequential(nn.Linear(latent_dim, 512), nn.ReLU(), nn.Linear(512, coarse_points * 3))
        self.expansion_mlp = nn.Sequential(nn.Linear(coarse_points * 3 + latent_dim, 1024), nn.ReLU(),
                                           nn.Linear(1024, final_points * 3))
        self.coarse_points, self.final_points = coarse_points, final_points


class _PppeProbParams(nn.Module):                       # pppe_pcd_ae.py:751-772 (parameters only)
    def __init__(self, feature_dim=512, hidden_channels=128, latent_bins=16, latent_channels=3):
        super().__init__()
        self.cond_proj = nn.Sequential(nn.Linear(feature_dim, hidden_channels), nn.ReLU(), nn.Linear(hidden_channels, hidden_channels))
        self.combine = nn.Sequential(nn.Conv1d(latent_channels + hidden_channels, hidden_channels, 1), nn.ReLU(),
                                     nn.Conv1d(hidden_channels, hidden_channels, 1))
        self.mean_head = nn.Conv1d(hidden_channels, latent_channels, 1)
        self.scale_head = nn.Conv1d(hidden_channels, latent_channels, 1)
        self.pmf_head = nn.Conv1d(hidden_channels, latent_bins, 1)


class PointCloudAE(_Packable):
    """pppe_pcd_ae.PointCloudAE.forward (pppe_pcd_ae.py:843-877), eval mode."""

    def __init__(self, latent_dim=64, latent_bins=16, npoints=8192):
        super().__init__()
        self.encoder = PointNet2EncoderFull(latent_dim=latent_dim)
        self.decoder = PCNDecoderSmall(latent_dim=latent_dim, coarse_points=512, final_points=npoints)
        self.prob = _PppeProbParams(512, 128, latent_bins, latent_dim)
        self.latent_bins, self.latent_dim = latent_bins, latent_dim
        self.q_min, self.q_max = 0.0, latent_bins - 1.0
        self._packed = None

    def pack(self, device="cuda"):
        sa = self.encoder.sa_modules
        self._packed = dict(
            msg=[[FoldedLinear(l[0].weight, None, True, l[1], device) for l in br.mlp_stack] for br in sa[0].branches],
            sa1=[FoldedLinear(l[0].weight, None, True, l[1], device) for l in sa[1].mlp_stack],
            sa2=[FoldedLinear(l[0].weight, None, True, l[1], device) for l in sa[2].mlp_stack],
            gconv=_fold_stack(self.encoder.global_conv, device),
            coarse=_fold_stack(self.decoder.fc_coarse, device), expand=_fold_stack(self.decoder.expansion_mlp, device))
        return self

    def forward(self, x, starts):
        """x (B,N,3) on the GPU; starts = [[msg_branch0, msg_branch1], sa2, sa3], each (B,) FPS start
        indices (the reference draws them with torch.randint, pn_kit.py:321).
        -> (coarse (B,512,3), fine (B,N,3), cond_feats (B,512), y_q (B,d), latent (B,d))."""
        if self._packed is None:
            self.pack(x.device)
        pk = self._packed
        x = ops._f32c(x, "PointCloudAE")
        B = x.shape[0]
        sa = self.encoder.sa_modules
        outs, new_xyz = [], None
        for br, stack, st in zip(sa[0].branches, pk["msg"], starts[0]):             # :617-632 (last branch's centroids win)
            new_xyz, f = br.run(stack, x, None, st)
            outs.append(f)
        feats = torch.cat(outs, dim=-1).contiguous()
        xyz, feats = sa[1].run(pk["sa1"], new_xyz, feats, starts[1])
        xyz, feats = sa[2].run(pk["sa2"], xyz, feats, starts[2])
        cond = group_max(feats)                                                      # :682 global max
        latent = cond
        for layer in pk["gconv"]:
            latent = layer(latent)                                                   # :684
        # quantize_st (:719-735) then dequantise (:873); the mean over N tiled copies (:875) is the value itself
        y_q, y_deq = torch.empty_like(latent), torch.empty_like(latent)
        _lib.call("pccx_quantize_st", latent.data_ptr(), latent.numel(), float(self.q_min), float(self.q_max),
                  int(self.latent_bins), y_q.data_ptr(), y_deq.data_ptr(), _stream())
        c = y_deq
        for layer in pk["coarse"]:
            c = layer(c)                                                             # :710
        e = torch.cat([c, y_deq], dim=1).contiguous()                                # :711
        for layer in pk["expand"]:
            e = layer(e)                                                             # :712
        return c.view(B, -1, 3), e.view(B, -1, 3), cond, y_q, latent
