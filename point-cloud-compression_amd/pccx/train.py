"""The training step of train_pppe_pcd_ae.py:184-226 (BASELINE configs[4]) on libpccx.so.

    coarse, recon, feats, y = ae(batch_x)                         # forward, BatchNorm in TRAIN mode
    fbpp = estimate_bits_per_point_conditional(y, feats, prob)    # no-grad rate term
    loss, dist, rate = criterion(recon, batch_x, fbpp, lam)       # get_loss("chamfer"): Chamfer + lam * rate
    loss.backward(); clip_grad_norm_(params, 1.0); optimizer.step()   # Adam

torch.autograd only sequences the backward: every Function below is a pair of HIP launches behind the
C ABI (include/pccx.h, csrc/train.hip); parameters and optimizer state are plain tensors in HBM.
fp32 by default.  ``train_step(..., autocast=True)`` is the reference's CUDA branch (torch.cuda.amp.autocast around the forward,
train_pppe_pcd_ae.py:193-205; BASELINE configs[4] names bf16): every Linear / 1x1 Conv rounds its operands to bf16, multiplies
on the bf16 matrix cores with fp32 accumulation and rounds its result to bf16; BatchNorm, max-pool, the quantiser, the losses,
clipping and Adam stay fp32 on fp32 master weights, as under autocast; the backward GEMMs (dX, dW) round their operands the same
way.  bf16 has fp32's exponent range, so no GradScaler is needed (the reference's scaler guards fp16).  Correctness-first.
"""
import math
import os

import torch

from . import _lib, families, ops
from .ops import _stream

_SCRATCH = {}
_AUTOCAST = False          # set by train_step(autocast=True) around forward + backward


class StepArena:
    """Everything a training step ACCUMULATES into -- the weight / bias gradients (atomics over row slices), split-K dX of the skinny
    Linears, scatter-added feature gradients, the Chamfer gradients, the column-sum scratch of every BatchNorm, Adam's norm
    accumulator -- carved from ONE buffer that is cleared by ONE launch at the start of the step (pccx_zero_bytes).  Round 3 cleared
    each of them separately: 75 torch fill kernels + 35 pccx_zero_kernel launches per step, ~0.5 ms of a 5 ms step.
    The first step through an arena only measures (every request falls back to torch.zeros); the buffer is allocated when that step
    ends and used from the next one on.  Slices are valid until the next begin() on the same arena: gradients are consumed by the
    optimiser inside the step.  A captured step owns its arena (GraphedTrainStep), so eager steps never clear a graph's gradients."""

    def __init__(self):
        self.buf, self.off, self.need, self.active = None, 0, 0, False

    def begin(self, device):
        self.off, self.need, self.active = 0, 0, True
        dv = torch.device(device)
        if dv.type == "cuda" and dv.index is None:
            dv = torch.device("cuda", torch.cuda.current_device())
        if self.buf is not None and self.buf.device != dv:
            self.buf = None
        if self.buf is not None:
            _lib.call("pccx_zero_bytes", self.buf.data_ptr(), self.buf.numel(), _stream())

    def zeros(self, shape, dtype, device):
        """a zero tensor: an arena slice when the arena is live and large enough (second value True), else torch.zeros"""
        numel = 1
        for d_ in (shape if isinstance(shape, (tuple, list, torch.Size)) else (shape,)):
            numel *= int(d_)
        nbytes = (numel * torch.empty(0, dtype=dtype).element_size() + 15) // 16 * 16
        if self.active:
            self.need += nbytes
            if self.buf is not None and self.off + nbytes <= self.buf.numel():
                v = self.buf[self.off:self.off + nbytes].view(dtype)[:numel].view(shape)
                self.off += nbytes
                return v, True
        return torch.zeros(shape, dtype=dtype, device=device), False

    def end(self, device):
        self.active = False
        if (self.buf is None or self.need > self.buf.numel()) and self.need > 0 and not torch.cuda.is_current_stream_capturing():
            self.buf = torch.empty(self.need, dtype=torch.uint8, device=device)      # the next step's arena


_EAGER_ARENA = StepArena()
_ARENA = None              # the arena of the step in progress (train_step / GraphedTrainStep set it), or None outside a step


def _zeros(shape, dtype, device):
    """(tensor, cleared-by-the-arena?) -- see StepArena.zeros"""
    if _ARENA is not None:
        return _ARENA.zeros(shape, dtype, device)
    return torch.zeros(shape, dtype=dtype, device=device), False


ops.zeros_hook = _zeros       # the Chamfer backward's two gradient buffers come from the step's arena too


def _sums(C, device):
    """2*C doubles of column-sum scratch and the flag the reduction takes: 4 when the step's arena has cleared them already"""
    n = int(_lib.load().pccx_train_sums_doubles(C))          # eight replicas of the 2 C sums (csrc/train.hip: PCCX_SUM_REPLICAS)
    if _ARENA is not None:
        t, pre = _ARENA.zeros(n, torch.float64, device)
        if pre:
            return t, 4
    t = _SCRATCH.get((C, str(device)))
    if t is None:
        t = _SCRATCH[(C, str(device))] = torch.empty(n, device=device, dtype=torch.float64)
    return t, 0


def _packed(W, transpose):
    # One small launch in front of every GEMM.  Packing ALL the step's operands in one launch at its start (a table-driven kernel, round 5)
    # measured no gain at batch 4 (graph replay 2.51 ms either way) and LOST 1.4 ms per step at batches 16 / 64, where the 100 MB decoder
    # weights are packed too: packed right before its GEMM an operand is read back from the Infinity Cache (tools/experiments/r5/README.md).
    N, K = W.shape
    R, Cc = (K, N) if transpose else (N, K)
    wp = torch.empty(_lib.load().pccx_packed_linear_floats(R, Cc), device=W.device, dtype=torch.float32)
    _lib.call("pccx_pack_linear_device", W.data_ptr(), N, K, int(transpose), wp.data_ptr(), _stream())
    return wp


def _linear_raw(x, wp, bias, N, K, flags=0):
    M = x.shape[0]
    out = torch.empty(M, N, device=x.device, dtype=torch.float32)
    _lib.call("pccx_linear", x.data_ptr(), M, K, x.stride(0), wp.data_ptr(), bias.data_ptr() if bias is not None else None, N,
              int(flags), out.data_ptr(), N, _stream())
    return out


_MOMENTS = {}              # address of a LinearFn output produced WITH its column moments -> (shape, sums); consumed by the BnReluFn that
                           # takes that tensor next, emptied at the start of every forward
_BN_OF = {}                # address of a BnReluFn output y -> (shape, y, z, mean, rstd): the LinearFn that consumes y keeps it, and its
                           # backward produces dY together with the BatchNorm backward's two column sums (pccx_linear_bnback)
_BWD_SUMS = {}             # address of such a dY -> (shape, sums); consumed by that BnReluFn's backward
_FOLD_MOMENTS = os.environ.get("PCCX_NO_MOMENT_FOLD") != "1"      # experiment knob: 0 = every BatchNorm reduces its input itself


def _is_wide(M, N, K):
    """A handful of rows through a large weight matrix (the IPDAE decoder's Linear(1024, k * 128) on 64 patches, the pppe decoder's coarse
    layer at batches above 8): the generic layer gives each workgroup four of the N / 16 column tiles and walks the K / 16 weight fragments
    of those columns one dependent load after the other -- 1.5 ms for the dX of a 16384 x 1024 layer on 64 rows, a 67 MB weight read at
    45 GB/s.  With the ROLES SWAPPED the weight matrix is the row operand the kernels stream at full rate: y^T = W x^T is pccx_linear
    over N rows with the (few) activations packed as its weights, and dX^T = W^T dZ^T is pccx_linear_dw over the same N rows."""
    return 8 < M <= 256 and M % 4 == 0 and N >= 1024 and N >= 8 * M and K % 4 == 0 and N * K >= (1 << 21)


class LinearFn(torch.autograd.Function):
    """z = x W^T (+ b): nn.Linear / 1x1 Conv on channels-last rows.  want_moments (a Conv that feeds a train-mode BatchNorm): the GEMM's
    epilogue also accumulates the output's column moments (pccx_linear_moments) and BnReluFn takes them instead of reducing z again."""

    @staticmethod
    def forward(ctx, x, W, b, want_moments=False):
        x = x.contiguous()
        W2 = W.reshape(W.shape[0], -1).contiguous()
        ctx.save_for_backward(x, W2)
        ctx.has_bias, ctx.wshape = b is not None, W.shape
        ctx.flags = 2 if _AUTOCAST else 0
        bn_in = _BN_OF.get(x.data_ptr()) if _FOLD_MOMENTS else None
        ctx.bn_in = bn_in if (bn_in is not None and bn_in[0] == tuple(x.shape)) else None
        N, K = W2.shape
        # a few rows: a weight stream, not matrix work (csrc/train.hip).  The kernels take 16-byte loads of x rows and W rows: row stride and
        # base addresses are checked HERE (a (1, K) view keeps an arbitrary stride(0), a tensor with a storage offset can be misaligned) and
        # anything else takes the generic layer, as before round 3.
        ldx = K if x.shape[0] == 1 else x.stride(0)
        ctx.wide = False
        ctx.skinny = (x.shape[0] <= 8 and K % 4 == 0 and ldx % 4 == 0 and ldx >= K and x.data_ptr() % 16 == 0 and W2.data_ptr() % 16 == 0)
        if ctx.skinny:
            out = torch.empty(x.shape[0], N, device=x.device, dtype=torch.float32)
            _lib.call("pccx_linear_skinny", x.data_ptr(), x.shape[0], K, ldx, W2.data_ptr(), b.data_ptr() if b is not None else None,
                      N, ctx.flags, out.data_ptr(), N, _stream())
            return out
        ctx.wide = x.is_cuda and _is_wide(x.shape[0], N, K) and x.stride(0) == K
        if ctx.wide:
            yT = _linear_raw(W2, _packed(x, False), None, x.shape[0], K, ctx.flags)     # (N, M) = W x^T
            out = yT.t().contiguous()
            if b is not None:
                out += b
                if ctx.flags & 2:
                    out = out.bfloat16().float()                                        # the autocast layer rounds AFTER its bias
            return out
        if want_moments and b is None and _FOLD_MOMENTS and x.is_cuda:
            sums, pre = _sums(N, x.device)
            out = torch.empty(x.shape[0], N, device=x.device, dtype=torch.float32)
            _lib.call("pccx_linear_moments", x.data_ptr(), x.shape[0], K, x.stride(0), _packed(W2, False).data_ptr(), N, ctx.flags | pre,
                      out.data_ptr(), N, sums.data_ptr(), _stream())
            _MOMENTS[out.data_ptr()] = (tuple(out.shape), sums)
            return out
        return _linear_raw(x, _packed(W2, False), b, N, K, ctx.flags)

    @staticmethod
    def backward(ctx, dz):
        x, W2 = ctx.saved_tensors
        dz = dz.contiguous()
        N, K = W2.shape
        M = x.shape[0]
        dx = None
        if ctx.needs_input_grad[0] and ctx.skinny:
            # split-K with fp32 atomics: under autocast the operands are rounded to bf16 as in the generic dX, the SUM is left in fp32 (the
            # generic path rounds its result to bf16 as well; the per-layer pin of tests/test_train_step.py holds either to one bf16 ulp)
            dx, _ = _zeros((M, K), torch.float32, dz.device)
            _lib.call("pccx_linear_skinny_dx", dz.data_ptr(), M, N, dz.stride(0), W2.data_ptr(), K, ctx.flags, dx.data_ptr(), K, _stream())
        elif ctx.needs_input_grad[0] and ctx.wide:
            # dX^T (K, M) = W^T dZ^T: the weight-gradient kernel with W as its "dZ" (N rows of K) and dZ^T as its "x" (N rows of M)
            dzT = dz.t().contiguous()
            dxT, _ = _zeros((K, M), torch.float32, dz.device)
            _lib.call("pccx_linear_dw", W2.data_ptr(), dzT.data_ptr(), N, K, M, K, M, dxT.data_ptr(), ctx.flags, _stream())
            dx = dxT.t().contiguous()
        elif ctx.needs_input_grad[0] and ctx.bn_in is not None and K % 4 == 0:
            # x is the output of a train-mode BatchNorm-ReLU: dX is that layer's dY, and the GEMM's epilogue accumulates the two column
            # sums its backward needs from the rows it has just produced (13 col_reduce4<1> launches per step otherwise)
            _, y_, z_, mean_, rstd_ = ctx.bn_in
            sums, pre = _sums(K, dz.device)
            dx = torch.empty(M, K, device=dz.device, dtype=torch.float32)
            _lib.call("pccx_linear_bnback", dz.data_ptr(), M, N, dz.stride(0), _packed(W2, True).data_ptr(), K, ctx.flags | pre, dx.data_ptr(), K,
                      y_.data_ptr(), z_.data_ptr(), mean_.data_ptr(), rstd_.data_ptr(), sums.data_ptr(), _stream())
            _BWD_SUMS[dx.data_ptr()] = (tuple(dx.shape), sums)
        elif ctx.needs_input_grad[0]:
            dx = _linear_raw(dz, _packed(W2, True), None, K, N, ctx.flags)                                       # dX = dZ . W
        dW, _ = _zeros(tuple(W2.shape), torch.float32, dz.device)
        _lib.call("pccx_linear_dw", dz.data_ptr(), x.data_ptr(), M, N, K, N, x.stride(0), dW.data_ptr(), ctx.flags, _stream())
        db = None
        if ctx.has_bias:
            db = torch.empty(N, device=dz.device, dtype=torch.float32)                                             # written, not accumulated
            sums, pre = _sums(N, dz.device)
            _lib.call("pccx_col_sum_w", dz.data_ptr(), M, N, sums.data_ptr(), db.data_ptr(), pre, _stream())
        return dx, dW.view(ctx.wshape), db, None


class BnReluFn(torch.autograd.Function):
    """BatchNorm (training statistics, running buffers updated in place) followed by ReLU."""

    @staticmethod
    def forward(ctx, z, gamma, beta, bn):
        z = z.contiguous()
        M, Cc = z.shape
        mean = torch.empty(Cc, device=z.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        y = torch.empty_like(z)
        mom = _MOMENTS.pop(z.data_ptr(), None)
        if mom is not None and mom[0] == tuple(z.shape):
            sums, pre = mom[1], 4 | 8                # the producing GEMM's epilogue accumulated the moments: no reduction pass here
        else:
            sums, pre = _sums(Cc, z.device)
        # moments, then ONE kernel that finalises them (mean, rstd, running statistics) and applies the layer (csrc/train.hip)
        _lib.call("pccx_bn_relu_train_forward", z.data_ptr(), M, Cc, float(bn.eps), float(bn.momentum), sums.data_ptr(), gamma.data_ptr(),
                  beta.data_ptr(), 1, mean.data_ptr(), rstd.data_ptr(), bn.running_mean.data_ptr(), bn.running_var.data_ptr(), y.data_ptr(),
                  pre, _stream())
        if _BN_COUNTED is None:
            bn.num_batches_tracked += 1             # outside forward_train (which advances every counter of the model in one launch)
        ctx.save_for_backward(z, y, mean, rstd, gamma)
        if _FOLD_MOMENTS:
            _BN_OF[y.data_ptr()] = (tuple(y.shape), y, z, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        z, y, mean, rstd, gamma = ctx.saved_tensors
        dy = dy.contiguous()
        M, Cc = z.shape
        dz = torch.empty_like(z)
        gg, gb = torch.empty_like(gamma), torch.empty_like(gamma)                    # written by the apply kernel's first workgroup
        bs = _BWD_SUMS.pop(dy.data_ptr(), None)
        if bs is not None and bs[0] == tuple(dy.shape):
            sums, pre = bs[1], 4 | 8                 # the GEMM that produced dY accumulated the two sums in its epilogue
        else:
            sums, pre = _sums(Cc, z.device)
        _lib.call("pccx_bn_relu_train_backward", dy.data_ptr(), y.data_ptr(), z.data_ptr(), M, Cc, mean.data_ptr(), rstd.data_ptr(),
                  gamma.data_ptr(), sums.data_ptr(), dz.data_ptr(), gg.data_ptr(), gb.data_ptr(), pre, _stream())
        return dz, gg, gb, None


class ReluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z):
        y = torch.empty_like(z)
        # relu through the same elementwise kernel as its backward mask: y = max(z, 0)
        _lib.call("pccx_relu_backward", z.contiguous().data_ptr(), z.contiguous().data_ptr(), z.numel(), y.data_ptr(), _stream())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dz = torch.empty_like(y)
        _lib.call("pccx_relu_backward", dy.contiguous().data_ptr(), y.data_ptr(), y.numel(), dz.data_ptr(), _stream())
        return dz


class GroupMaxFn(torch.autograd.Function):
    """(G,Kn,C) -> (G,C), torch.max over the neighbour axis (first maximum on ties)."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        G, Kn, Cc = x.shape
        out = torch.empty(G, Cc, device=x.device, dtype=torch.float32)
        arg = torch.empty(G, Cc, device=x.device, dtype=torch.int32)
        _lib.call("pccx_group_max_arg", x.data_ptr(), G, Kn, Cc, out.data_ptr(), arg.data_ptr(), _stream())
        ctx.save_for_backward(arg)
        ctx.shape = (G, Kn, Cc)
        return out

    @staticmethod
    def backward(ctx, dout):
        (arg,) = ctx.saved_tensors
        G, Kn, Cc = ctx.shape
        dx = torch.empty(G, Kn, Cc, device=dout.device, dtype=torch.float32)
        _lib.call("pccx_group_max_backward", dout.contiguous().data_ptr(), arg.data_ptr(), G, Kn, Cc, dx.data_ptr(), _stream())
        return dx


class GatherFn(torch.autograd.Function):
    """index_points(feats (B,N,C), idx (B,S,K)) with gradient to feats."""

    @staticmethod
    def forward(ctx, feats, idx):
        ctx.save_for_backward(idx)
        ctx.shape = feats.shape
        return ops.index_points(feats, idx)

    @staticmethod
    def backward(ctx, dg):
        (idx,) = ctx.saved_tensors
        B, N, Cc = ctx.shape
        dg = dg.contiguous()
        M = idx[0].numel()
        df, pre = _zeros((B, N, Cc), torch.float32, dg.device)
        _lib.call("pccx_gather_backward_acc", dg.data_ptr(), Cc, idx.contiguous().data_ptr(), B, M, N, Cc, df.data_ptr(), 4, _stream())
        return df, None


class QuantizeSTFn(torch.autograd.Function):
    """quantize_st (pppe_pcd_ae.py:719-735) -> (y_q, y_dequant); straight-through gradient."""

    @staticmethod
    def forward(ctx, latent, qmin, qmax, levels):
        latent = latent.contiguous()
        yq, ydeq = torch.empty_like(latent), torch.empty_like(latent)
        _lib.call("pccx_quantize_st", latent.data_ptr(), latent.numel(), float(qmin), float(qmax), int(levels), yq.data_ptr(),
                  ydeq.data_ptr(), _stream())
        ctx.save_for_backward(latent)
        ctx.cfg = (float(qmin), float(qmax), int(levels))
        ctx.mark_non_differentiable(yq)
        return yq, ydeq

    @staticmethod
    def backward(ctx, _dyq, dydeq):
        (latent,) = ctx.saved_tensors
        dx = torch.empty_like(latent)
        _lib.call("pccx_quantize_st_backward", latent.data_ptr(), dydeq.contiguous().data_ptr(), latent.numel(), *ctx.cfg,
                  dx.data_ptr(), _stream())
        return dx, None, None, None


class SmoothL1Fn(torch.autograd.Function):
    """F.smooth_l1_loss(a, b, reduction='mean') (pppe_pcd_ae.py:822,826); gradient to a."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        val = torch.empty(1, device=a.device, dtype=torch.float64)
        _lib.call("pccx_smooth_l1", a.data_ptr(), b.data_ptr(), a.numel(), 0.0, val.data_ptr(), None, _stream())
        ctx.save_for_backward(a, b)
        return (val / a.numel()).float().reshape(())

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        val = torch.empty(1, device=a.device, dtype=torch.float64)
        grad = torch.empty_like(a)
        _lib.call("pccx_smooth_l1", a.data_ptr(), b.data_ptr(), a.numel(), float(g) / a.numel(), val.data_ptr(), grad.data_ptr(),
                  _stream())
        return grad, None


# ------------------------------------------------------------------------------------------------
def _select(mod, xyz, start, fps_idx=None):
    """The part of PointNetSetAbstraction.forward (pppe_pcd_ae.py:593-600) that depends on COORDINATES and the FPS start index only --
    never on a weight: FPS, the centroids, kNN grouping of the coordinates.  -> (new_xyz (B,S,3), grouped (B,S,K,3) = nn - centroid,
    idx (B,S,K) int64).  fps_idx: the module's FPS indices when the caller has drawn them already."""
    B, N, _ = xyz.shape
    S = mod.npoint
    if fps_idx is None and S != N:
        fps_idx = ops.farthest_point_sample_batch(xyz, S, start)
    new_xyz = xyz if S == N else ops.index_points(xyz, fps_idx)
    nn_ = ops.knn_points(new_xyz, xyz, mod.K, patch_scale=1.0)
    return new_xyz, nn_.knn, nn_.idx


def selection_tables(model, x, starts):
    """Every FPS / kNN result of PointNet2EncoderFull.forward (pppe_pcd_ae.py:596-632, pn_kit.py:309-330) for one batch, in module order:
    [MSG branch 0, MSG branch 1, sa_modules[1], sa_modules[2]], each (new_xyz, grouped, idx).  They are functions of the batch and of
    the explicit start indices alone, so a training loop can compute them for batch i+1 while step i runs (GraphedTrainStep(prefetch=True))."""
    sa = model.encoder.sa_modules
    brs = list(sa[0].branches)
    B = x.shape[0]
    fps_of = [None] * len(brs)
    if len(brs) > 1 and all(b_.npoint == brs[0].npoint != x.shape[1] for b_ in brs):
        # the branches draw independent FPS samples of the SAME cloud from their own start indices (pppe_pcd_ae.py:624-632): FPS is a
        # chain of npoint dependent rounds on one CU per cloud, so the draws of all branches go into ONE launch (B x branches clouds)
        st_all = torch.cat([torch.as_tensor(s_).to(device=x.device, dtype=torch.int32).reshape(-1) for s_ in starts[0]])
        idx_all = ops.farthest_point_sample_batch(x.repeat(len(brs), 1, 1), brs[0].npoint, st_all)
        fps_of = list(idx_all.view(len(brs), B, -1))
    tables = [_select(br, x, st, fi) for br, st, fi in zip(brs, starts[0], fps_of)]
    tables.append(_select(sa[1], tables[-1][0], starts[1]))          # the last branch's centroids win (pppe_pcd_ae.py:631)
    tables.append(_select(sa[2], tables[-1][0], starts[2]))
    return tables


def _sa_train(mod, table, feats):
    """PointNetSetAbstraction.forward (pppe_pcd_ae.py:586-611) in train mode, channels-last, on the module's selection table."""
    new_xyz, grouped, idx = table                                           # no gradient: xyz is data
    B, S = new_xyz.shape[0], new_xyz.shape[1]
    if feats is not None:
        grouped = torch.cat([grouped, GatherFn.apply(feats, idx)], dim=-1)
    x = grouped.reshape(-1, grouped.shape[-1])
    for layer in mod.mlp_stack:                                             # conv (no bias) -> BN -> ReLU
        x = BnReluFn.apply(LinearFn.apply(x, layer[0].weight, None, True), layer[1].weight, layer[1].bias, layer[1])
    return new_xyz, GroupMaxFn.apply(x.view(B * S, mod.K, -1)).view(B, S, -1)


def forward_train(model, x, starts, tables=None):
    """PointCloudAE.forward (pppe_pcd_ae.py:858-877) with BatchNorm in train mode.  tables: selection_tables(model, x, starts) when the
    caller has them already (then `starts` is not looked at).
    -> (coarse (B,512,3), fine (B,N,3), cond (B,512), y_q (B,d))."""
    global _BN_COUNTED
    enc, dec = model.encoder, model.decoder
    B = x.shape[0]
    sa = enc.sa_modules
    outs, new_xyz = [], None
    _MOMENTS.clear()
    _BN_OF.clear()
    _BWD_SUMS.clear()
    _BN_COUNTED = _advance_bn_counters(model, x.device)    # every BatchNorm's num_batches_tracked += 1, one launch
    try:
        with ops.stage("selection"):
            tables = selection_tables(model, x, starts) if tables is None else tables
        with ops.stage("forward"):
            return _forward_train_body(model, tables, enc, dec, B, sa)
    finally:
        _BN_COUNTED = None


_BN_COUNTED = None


def _advance_bn_counters(model, device):
    """num_batches_tracked += 1 for every BatchNorm of the model (what nn.BatchNorm does per forward in train mode, pppe_pcd_ae.py:556-568)
    by ONE kernel over a device table of the counters' addresses, cached on the model and rebuilt when a buffer has moved."""
    ctrs = [b for n, b in model.named_buffers() if n.endswith("num_batches_tracked") and b.device.type == "cuda"]
    if not ctrs:
        return None
    ptrs = [int(b.data_ptr()) for b in ctrs]
    cache = getattr(model, "_pccx_bn_table", None)
    if cache is None or cache[0] != ptrs:
        if torch.cuda.is_current_stream_capturing():
            return None                                      # no host-to-device copy inside a capture: the layers advance their own counters
        cache = (ptrs, torch.tensor(ptrs, dtype=torch.int64, device=device))
        model._pccx_bn_table = cache
    _lib.call("pccx_add_i64_table", cache[1].data_ptr(), len(ptrs), 1, _stream())
    return True


def _forward_train_body(model, tables, enc, dec, B, sa):
    outs = []
    for br, tb in zip(sa[0].branches, tables):
        _, f = _sa_train(br, tb, None)
        outs.append(f)
    feats = torch.cat(outs, dim=-1)
    _, feats = _sa_train(sa[1], tables[-2], feats)
    _, feats = _sa_train(sa[2], tables[-1], feats)
    cond = GroupMaxFn.apply(feats)                                          # global max over the 32 points
    gc = enc.global_conv
    h = BnReluFn.apply(LinearFn.apply(cond, gc[0].weight, None), gc[1].weight, gc[1].bias, gc[1])
    latent = LinearFn.apply(h, gc[3].weight, gc[3].bias)
    y_q, y_deq = QuantizeSTFn.apply(latent, model.q_min, model.q_max, model.latent_bins)
    c = LinearFn.apply(ReluFn.apply(LinearFn.apply(y_deq, dec.fc_coarse[0].weight, dec.fc_coarse[0].bias)),
                       dec.fc_coarse[2].weight, dec.fc_coarse[2].bias)
    e = torch.cat([c, y_deq], dim=1)
    fine = LinearFn.apply(ReluFn.apply(LinearFn.apply(e, dec.expansion_mlp[0].weight, dec.expansion_mlp[0].bias)),
                          dec.expansion_mlp[2].weight, dec.expansion_mlp[2].bias)
    return c.view(B, -1, 3), fine.view(B, -1, 3), cond, y_q


@torch.no_grad()
def estimate_bits_per_point(model, y_q, cond):
    """estimate_bits_per_point_conditional (pppe_pcd_ae.py:882-917): every one of the N tiled columns is
    identical, so one column per cloud is evaluated; returns the scalar fbpp."""
    pr = model.prob
    lin = lambda t, m, relu: (ReluFn.apply if relu else (lambda v: v))(LinearFn.apply(t, m.weight, m.bias))
    c = lin(lin(cond, pr.cond_proj[0], True), pr.cond_proj[2], False)
    h = lin(lin(torch.cat([y_q, c], dim=1), pr.combine[0], True), pr.combine[2], False)
    logits = lin(h, pr.pmf_head, False).contiguous()
    out = torch.empty(1, device=y_q.device, dtype=torch.float32)
    _lib.call("pccx_rate_from_logits", logits.data_ptr(), y_q.contiguous().data_ptr(), y_q.shape[0], logits.shape[1], y_q.shape[1],
              out.data_ptr(), _stream())
    return out.reshape(())


def rd_loss(fine, target, fbpp, lam, loss_type="chamfer", alpha=0.7, max_rate=100.0):
    """RateDistortionLoss (pppe_pcd_ae.py:807-838): distortion + lam * clamp(rate).  loss_type "chamfer" is what the
    training script builds (get_loss("chamfer"), train_pppe_pcd_ae.py:48); "l1" = smooth-L1; anything else is the
    class's hybrid alpha*Chamfer + (1-alpha)*smooth-L1."""
    if loss_type == "chamfer":
        dist, _ = ops.chamfer_distance(fine, target)
    elif loss_type == "l1":
        dist = SmoothL1Fn.apply(fine, target)
    else:
        chamfer, _ = ops.chamfer_distance(fine, target)
        dist = alpha * chamfer + (1 - alpha) * SmoothL1Fn.apply(fine, target)
    rate = fbpp.clamp(0.0, max_rate)
    return dist + lam * rate, dist.detach(), rate.detach()


class Adam:
    """torch.optim.Adam(lr, betas=(0.9,0.999), eps=1e-8) + clip_grad_norm_(max_norm) as HIP kernels."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.params = [p for p in params]
        self.lr, self.betas, self.eps, self.t = lr, betas, eps, 0
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]
        self.hyper = None           # device {lr, 1 - b1^t, 1 - b2^t, t, b1^t, b2^t}: set by make_capturable()
        self._table = self._graph_table = self._pending = self._keep = None   # step(): device tables of tensor addresses

    def make_capturable(self, device):
        """Keep lr, the step counter and the bias corrections in DEVICE memory (pccx_adam_advance_dev / pccx_adam_step_dev), so
        that step() has no per-step launch argument and no per-step host write: it can sit inside a captured hipGraph, and a CPU
        that runs several replays ahead cannot race the values a queued replay reads.  The state starts from the current t."""
        if self.hyper is None:
            import numpy as np
            host = np.zeros(8, dtype=np.float32)                 # float lr | 1-b1^t | 1-b2^t | int32 t | double b1^t | double b2^t
            host[0] = self.lr
            host[1], host[2] = 1.0 - self.betas[0] ** self.t, 1.0 - self.betas[1] ** self.t
            host.view(np.int32)[3] = self.t
            host.view(np.float64)[2:4] = (self.betas[0] ** self.t, self.betas[1] ** self.t)
            self.hyper = torch.from_numpy(host).to(device)       # one synchronous copy at set-up
            self._graph_table = torch.zeros(len(self.params), 6, device=device, dtype=torch.int64)
        return self

    def set_lr(self, lr):
        """A new learning rate (the cosine schedule of train_pppe_pcd_ae.py:148): stream-ordered fill of the device word, the value
        travels as a launch argument -- nothing on the host is read later."""
        self.lr = float(lr)
        if self.hyper is not None:
            self.hyper[0:1].fill_(self.lr)

    def advance(self):
        """t += 1 on the device (and on the host mirror unless the launch is only being captured)."""
        _lib.call("pccx_adam_advance_dev", self.hyper.data_ptr(), float(self.betas[0]), float(self.betas[1]), _stream())
        if not torch.cuda.is_current_stream_capturing():
            self.t += 1

    def step(self, max_norm=None):
        """clip_grad_norm_(max_norm) + Adam over every parameter that has a gradient: two launches in all (pccx_sumsq_multi,
        pccx_adam_multi) driven by a table of the tensors' addresses in device memory."""
        import numpy as np
        capturable = self.hyper is not None
        capturing = torch.cuda.is_current_stream_capturing()
        if capturable:
            self.advance()              # eager or captured alike: the step advances its own device counter
        else:
            self.t += 1
        live = [(p, p.grad.contiguous(), m, v) for p, m, v in zip(self.params, self.m, self.v) if p.grad is not None]
        if not live:
            return None
        dev = live[0][0].device
        rows, first = np.zeros((len(live), 6), dtype=np.int64), 0
        for r, (p, g, m, v) in enumerate(live):
            rows[r] = (p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), first)
            first += (p.numel() + 1023) // 1024
        self._keep = [g for _, g, _, _ in live]                      # the addresses in the table stay valid until the next step
        if capturing:
            # nothing may be copied from the host inside a capture: the kernels are recorded with the ADDRESS of a table of their own
            # (eager steps on this optimiser keep theirs), its content is uploaded right after the capture (flush_table, called by
            # GraphedTrainStep) -- the captured gradients live at fixed addresses in the graph's pool
            if self._graph_table is None:
                raise _lib.PccxError("Adam.step inside a capture needs make_capturable() first (it allocates the captured step's table)")
            table = self._graph_table                                # allocated (zeroed: n = 0 rows touch nothing) OUTSIDE the capture
            self._pending = rows
        else:
            if self._table is None:
                self._table = torch.zeros(len(self.params), 6, device=dev, dtype=torch.int64)
            table = self._table
            table[:len(live)].copy_(torch.from_numpy(rows))          # pageable source: the copy has completed when this returns
        acc = None
        if max_norm is not None:
            acc, _ = _zeros(1, torch.float64, dev)
            _lib.call("pccx_sumsq_multi", table.data_ptr(), len(live), first, acc.data_ptr(), _stream())
        _lib.call("pccx_adam_multi", table.data_ptr(), len(live), first, acc.data_ptr() if acc is not None else None,
                  float(max_norm or 0.0), self.hyper.data_ptr() if capturable else None, float(self.lr), int(max(self.t, 1)),
                  float(self.betas[0]), float(self.betas[1]), float(self.eps), _stream())
        return acc

    def state_dict(self):
        """torch.optim.Adam.state_dict()'s layout (what train.py:107 saves as optimizer_step{N}.pkl): per-parameter step / exp_avg /
        exp_avg_sq keyed by the parameter's position, one param_group with the hyper-parameters -- a checkpoint written here loads into
        torch.optim.Adam over the same parameter list, and the other way round."""
        state = {i: {"step": torch.tensor(float(self.t)), "exp_avg": m.detach().clone(), "exp_avg_sq": v.detach().clone()}
                 for i, (m, v) in enumerate(zip(self.m, self.v))} if self.t > 0 else {}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": 0, "amsgrad": False, "maximize": False,
                 "foreach": None, "capturable": False, "differentiable": False, "fused": None, "params": list(range(len(self.params)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        g = sd["param_groups"][0]
        if len(g["params"]) != len(self.params):
            raise _lib.PccxError(f"Adam.load_state_dict: {len(g['params'])} parameters in the checkpoint, {len(self.params)} here")
        if g.get("weight_decay", 0) or g.get("amsgrad", False) or g.get("maximize", False):
            raise _lib.PccxError("Adam.load_state_dict: weight_decay / amsgrad / maximize are not built (train.py:131-134 uses none of them)")
        self.betas, self.eps = tuple(g["betas"]), float(g["eps"])
        steps = {int(float(st["step"])) for st in sd["state"].values()}
        if len(steps) > 1:
            raise _lib.PccxError(f"Adam.load_state_dict: parameters at different steps {sorted(steps)} (one counter here)")
        self.t = steps.pop() if steps else 0
        for i, (m, v) in enumerate(zip(self.m, self.v)):
            st = sd["state"].get(i)
            if st is None:
                m.zero_(), v.zero_()
            else:
                m.copy_(st["exp_avg"]), v.copy_(st["exp_avg_sq"])
        if self.hyper is not None:
            self.hyper = None
            self.make_capturable(self.params[0].device)
        self.set_lr(float(g["lr"]))

    def flush_table(self):
        """Upload the tensor table recorded during a capture (step() could not copy from the host there)."""
        if self._pending is not None:
            self._graph_table[:self._pending.shape[0]].copy_(torch.from_numpy(self._pending))
            self._pending = None


def step_flops(model, batch):
    """Algorithmic FLOPs of one training step's GEMMs for ``batch`` clouds: 2 * MACs of every Linear / 1x1 Conv of the forward
    (rows x in x out), times 3 for forward + dX + dW (the first layers' dX on raw coordinates is not computed: negligible)."""
    enc, dec = model.encoder, model.decoder
    macs = 0
    rows_in = {0: None}
    sa = enc.sa_modules
    for br in sa[0].branches:
        rows = batch * br.npoint * br.K
        for layer in br.mlp_stack:
            macs += rows * layer[0].weight.shape[0] * layer[0].weight.shape[1]
    for m in (sa[1], sa[2]):
        rows = batch * m.npoint * m.K
        for layer in m.mlp_stack:
            macs += rows * layer[0].weight.shape[0] * layer[0].weight.shape[1]
    for lin in (enc.global_conv[0], enc.global_conv[3], dec.fc_coarse[0], dec.fc_coarse[2], dec.expansion_mlp[0], dec.expansion_mlp[2]):
        macs += batch * lin.weight.shape[0] * lin.weight[0].numel()
    return 3 * 2 * macs


class GraphedTrainStep:
    """The whole iteration of train_step -- forward, rate term, loss, backward, clipping, Adam: about 600 small launches at batch
    4 -- captured ONCE as a hipGraph and replayed (MI355X-first: HIP graphs for a launch-bound inner loop instead of a tracing
    compiler).  Everything that changes from step to step lives in device memory the graph reads: the batch, the FPS start indices,
    lambda, and Adam's lr / bias corrections (Adam.make_capturable).  Shapes are fixed at construction.  The ``warmup`` eager
    iterations run before the capture are REAL optimisation steps on the construction batch (they also size the scratch buffers).
    data_parallel=True (one replica per GPU under torch.distributed): the iteration is captured as TWO graphs cut at the only
    exchange of the step -- forward + backward, then clip + Adam -- and the bucketed gradient all-reduce over RCCL
    (dist.allreduce_mean_, on the gradients' fixed graph-pool addresses) runs between the two replays on the same stream.
    prefetch=True takes SELECTION out of the captured step: every FPS / kNN result of the encoder is a function of the batch and the
    start indices alone (selection_tables), FPS is 512 dependent rounds on 8 of the 256 CUs (0.56 ms of a 3.3 ms step at batch 4), so
    the tables of batch i+1 are computed on a SIDE stream while the graph of step i runs.  The graph reads the batch and the tables
    from one fixed buffer (`cur`); prefetch(batch, starts) fills a second one (`nxt`) on the side stream; __call__ moves nxt -> cur
    with one copy kernel in front of the replay, and the next prefetch waits only for that copy, not for the replay.

        step.prefetch(x0, s0)
        for i in ...:
            out = step(next_batch=(x[i+1], s[i+1]))   # consumes batch i; queues batch i+1's selection; replays

    next_batch is queued BETWEEN the copy and the replay: the side stream is ordered after whatever produced the next batch on the
    caller's stream (an event recorded there), and that point must lie in front of the replay or the selection would wait for it
    (calling prefetch() after step() does exactly that: measured 3.5 ms per step against 2.6, tools/experiments/r5/).
    __call__(batch_x, starts) without a pending prefetch runs the selection first and then the graph (the un-pipelined LATENCY of one
    step); results are the same either way (tests/test_train_step.py)."""

    def __init__(self, model, opt, batch_x, starts, lam=1.0, grad_clip=1.0, loss_type="chamfer", autocast=False, warmup=2,
                 data_parallel=False, debug_dot=None, prefetch=False):
        if loss_type != "chamfer":
            raise _lib.PccxError("GraphedTrainStep: loss_type='chamfer' (what train_pppe_pcd_ae.py:48 builds) is the captured loss; "
                                 "the smooth-L1 backward still reads its upstream gradient on the host")
        dev = batch_x.device
        self.model, self.opt, self.grad_clip, self.loss_type, self.autocast = model, opt, grad_clip, loss_type, autocast
        self.data_parallel, self.graph_opt = bool(data_parallel), None
        self.arena = StepArena()            # this step's accumulation buffers: sized by the warm-up iterations, cleared by the graph's first node
        opt.make_capturable(dev)
        as_dev = lambda s_: torch.as_tensor(s_).to(device=dev, dtype=torch.int32).contiguous().clone()
        self.x = batch_x.detach().clone().contiguous()
        self.starts = [[as_dev(s_) for s_ in starts[0]], as_dev(starts[1]), as_dev(starts[2])]
        self.lam = torch.tensor(float(lam), device=dev, dtype=torch.float32)
        self.prefetch_mode, self.tables, self._pending = bool(prefetch), None, False
        if self.prefetch_mode:
            # one flat buffer per side: the batch, then every table tensor, each at a 16-byte aligned offset
            probe = [self.x] + [t for tb in selection_tables(model, self.x, self.starts) for t in tb]
            offs, o = [], 0
            for t in probe:
                offs.append(o)
                o += (t.numel() * t.element_size() + 15) // 16 * 16
            self._flat = [torch.zeros(o, device=dev, dtype=torch.uint8) for _ in range(2)]           # cur, nxt
            views = lambda buf: [buf[a:a + t.numel() * t.element_size()].view(t.dtype).view(t.shape) for a, t in zip(offs, probe)]
            self._cur, self._nxt = views(self._flat[0]), views(self._flat[1])
            for dst, src in zip(self._cur, probe):
                dst.copy_(src)
            self.x = self._cur[0]
            self.tables = [tuple(self._cur[1 + 3 * i:4 + 3 * i]) for i in range((len(probe) - 1) // 3)]
            self._side, self._upload = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
            self._sel_done, self._copied = torch.cuda.Event(), torch.cuda.Event()
            self._copied.record(torch.cuda.current_stream())
        if warmup > 0:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    self._body()
            torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        if debug_dot:                       # hipGraphDebugDotPrint of the captured step (nodes and edges), for diagnostics
            self.graph.enable_debug_mode()
        # capture_error_mode="thread_local": only THIS thread's calls are checked against the capture.  In the default (global) mode a
        # call from any thread invalidates it -- and under a process group the RCCL watchdog thread polls the events of earlier
        # collectives (hipEventQuery) whenever it likes: a captured data-parallel step then died at random with "operation not permitted
        # when stream is capturing" raised in the WATCHDOG and hipErrorStreamCaptureInvalidated here (found by the one-rank RCCL test,
        # about one run in ten; it would have hit the N-GPU --graph runs the same way).
        mode = dict(capture_error_mode="thread_local")
        if self.data_parallel:
            with torch.cuda.graph(self.graph, **mode):
                self.out = self._fwd_bwd()
            self._dp_grads = [p.grad for p in opt.params if p.grad is not None]      # fixed addresses: what the all-reduce averages
            self.graph_opt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_opt, pool=self.graph.pool(), **mode):
                self._opt_step()
        else:
            with torch.cuda.graph(self.graph, **mode):
                self.out = self._body()
        if debug_dot:
            self.graph.debug_dump(debug_dot)
        self._grads = opt._keep            # the captured gradients: graph-pool tensors at the addresses the table holds
        opt.flush_table()
        # Leave no python handle on the captured iteration: the parameters' .grad are graph-pool tensors the replays own.  Eager
        # iterations on the same model / optimiser may alternate with replays (the optimiser's step counter lives on the device
        # and both advance it).
        for p in opt.params:
            p.grad = None

    def _fwd_bwd(self):
        global _AUTOCAST, _ARENA
        for p in self.opt.params:
            p.grad = None
        _AUTOCAST = bool(self.autocast)
        _ARENA = self.arena
        self.arena.begin(self.x.device)
        try:
            coarse, fine, cond, y_q = forward_train(self.model, self.x, self.starts, tables=self.tables)
            fbpp = estimate_bits_per_point(self.model, y_q, cond.detach())
            loss, dist, rate = rd_loss(fine, self.x, fbpp, self.lam, self.loss_type)
            _AUTOCAST = False
            loss.backward()
        finally:
            _AUTOCAST = False
            _ARENA = None
        return loss.detach(), dist, rate

    def _opt_step(self):
        global _ARENA
        _ARENA = self.arena                                          # Adam's norm accumulator comes from the same cleared buffer
        try:
            self.opt.step(max_norm=self.grad_clip)
        finally:
            _ARENA = None
            self.arena.end(self.x.device)

    def _body(self):
        out = self._fwd_bwd()
        if self.data_parallel:                                       # warm-up iterations of a replica: average, then step
            from . import dist as pdist
            pdist.allreduce_mean_([p.grad for p in self.opt.params])
        self._opt_step()
        return out

    def prefetch(self, batch_x, starts):
        """Queue the selection of the NEXT batch on the side stream (prefetch=True only): FPS / kNN tables of (batch_x, starts) into the
        `nxt` buffer.  Returns at once; the next __call__() consumes it."""
        if not self.prefetch_mode:
            raise _lib.PccxError("GraphedTrainStep.prefetch needs prefetch=True at construction")
        dev = self.x.device
        main = torch.cuda.current_stream()
        ready = torch.cuda.Event()
        ready.record(main)                                           # batch_x / starts may have been produced on the caller's stream
        # start indices that live on the HOST go up in one piece on a stream of their own: a pageable copy blocks the calling thread until
        # everything queued before it on ITS stream is done -- on the side stream that is the previous step's replay (measured: 3.15 ms
        # per step with the four small uploads there, 2.8 with the indices already on the device)
        parts = list(starts[0]) + [starts[1], starts[2]]
        if any(not (isinstance(p_, torch.Tensor) and p_.is_cuda) for p_ in parts):
            import numpy as np
            host = np.concatenate([np.asarray(p_.cpu() if isinstance(p_, torch.Tensor) else p_).astype(np.int32).reshape(-1) for p_ in parts])
            with torch.cuda.stream(self._upload):
                up = torch.from_numpy(host).to(dev)
                upl = torch.cuda.Event()
                upl.record(self._upload)
            sizes = [int(np.asarray(p_.cpu() if isinstance(p_, torch.Tensor) else p_).size) for p_ in parts]
            cut = list(torch.split(up, sizes))
            up.record_stream(self._side)
        else:
            upl, cut = None, [p_.to(torch.int32).contiguous() for p_ in parts]
        st = [cut[:len(starts[0])], cut[-2], cut[-1]]
        with torch.cuda.stream(self._side):
            self._side.wait_event(ready)
            if upl is not None:
                self._side.wait_event(upl)
            self._side.wait_event(self._copied)                      # nxt is free once the previous step has moved it into cur
            bx = batch_x.detach().to(device=dev, dtype=torch.float32).contiguous()
            flat = [bx] + [t for tb in selection_tables(self.model, bx, st) for t in tb]
            for dst, src in zip(self._nxt, flat):
                dst.copy_(src)
            self._sel_done.record(self._side)
            for t in flat:
                t.record_stream(self._side)
        self._pending = True

    def __call__(self, batch_x=None, starts=None, lam=None, sync=True, next_batch=None):
        """One iteration.  Returns (loss, dist, rate) as floats (sync=True) or the device scalars of the graph (sync=False).
        next_batch = (batch_x, starts) of the FOLLOWING iteration (prefetch=True): its selection is queued on the side stream here, in
        front of this iteration's replay, and runs under it."""
        if next_batch is not None and not self.prefetch_mode:
            raise _lib.PccxError("GraphedTrainStep: next_batch needs prefetch=True at construction")
        if self.prefetch_mode:
            if batch_x is not None or starts is not None:
                if self._pending:
                    raise _lib.PccxError("GraphedTrainStep: a prefetched batch is pending; call step() without arguments to consume it")
                if batch_x is None or starts is None:
                    raise _lib.PccxError("GraphedTrainStep(prefetch=True): batch_x and starts come together (the tables depend on both)")
                self.prefetch(batch_x, starts)                       # the un-pipelined form: selection, then the graph
            if self._pending:
                main = torch.cuda.current_stream()
                main.wait_event(self._sel_done)
                _lib.call("pccx_copy_bytes", self._flat[1].data_ptr(), self._flat[0].data_ptr(), self._flat[0].numel(), _stream())
                self._copied.record(main)
                self._pending = False
            if next_batch is not None:
                self.prefetch(*next_batch)
        else:
            if batch_x is not None:
                self.x.copy_(batch_x)
            if starts is not None:
                for dst, src in zip(self.starts[0], starts[0]):
                    dst.copy_(torch.as_tensor(src).to(dst.device, torch.int32))
                self.starts[1].copy_(torch.as_tensor(starts[1]).to(self.x.device, torch.int32))
                self.starts[2].copy_(torch.as_tensor(starts[2]).to(self.x.device, torch.int32))
        if lam is not None:
            self.lam.fill_(float(lam))
        self.graph.replay()             # carries pccx_adam_advance_dev: the device counter moves with the replay
        if self.graph_opt is not None:
            from . import dist as pdist
            pdist.allreduce_mean_(self._dp_grads)                    # RCCL, same stream, between the two graphs
            self.graph_opt.replay()
        self.opt.t += 1                 # host mirror
        return tuple(float(t) for t in self.out) if sync else self.out


def train_step(model, opt, batch_x, starts, lam=1.0, grad_clip=1.0, data_parallel=False, loss_type="chamfer", autocast=False):
    """One iteration of train_one_epoch (train_pppe_pcd_ae.py:184-226).  ``opt`` covers ae + prob
    parameters as the reference's optimizer does; returns (loss, dist, rate) as python floats.
    data_parallel=True averages the gradients over the ranks of the default process group (bucketed
    all-reduce, dist.allreduce_mean_) between backward and the clipped Adam step."""
    global _AUTOCAST, _ARENA
    for p in opt.params:
        p.grad = None
    _AUTOCAST = bool(autocast)              # the Linear layers of forward (and, through ctx.flags, of backward) take the bf16 form
    _ARENA = _EAGER_ARENA if batch_x.is_cuda else None
    if _ARENA is not None:
        _ARENA.begin(batch_x.device)        # ONE clear for everything this step accumulates into (StepArena)
    try:
        coarse, fine, cond, y_q = forward_train(model, batch_x, starts)
        fbpp = estimate_bits_per_point(model, y_q, cond.detach())
        loss, dist, rate = rd_loss(fine.float(), batch_x.float(), fbpp, lam, loss_type)      # :205 casts back to fp32 for the loss
        _AUTOCAST = False
        if data_parallel:
            # gradient averaging overlapped with backward: each bucket is all-reduced on a side stream as soon as its last gradient is
            # written (dist.GradBuckets); the clipped Adam step waits for the last bucket
            from . import dist as pdist
            if getattr(opt, "_dp", None) is None:
                opt._dp = pdist.GradBuckets(opt.params)
            opt._dp.begin()
            loss.backward()
            opt._dp.finish()
        else:
            loss.backward()
        opt.step(max_norm=grad_clip)
    finally:
        _AUTOCAST = False
        if _ARENA is not None:
            _ARENA.end(batch_x.device)
        _ARENA = None
    return float(loss.detach()), float(dist), float(rate)
