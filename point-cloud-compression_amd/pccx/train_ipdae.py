"""The training step of train.py:156-245 (``--model AE``: the IPDAE autoencoder of AE.py with its ConditionalProbabilityModel) on libpccx.so.

    batch_x, center, longest = pn_kit.normalize(batch_x, margin=0.01)                         # :171
    sampled_xyz = index_points(batch_x, farthest_point_sample_batch(batch_x, S))              # :178
    octree_codes, sampled_bits = encode_sampled_np(sampled_xyz, 1, N, OCTREE_BPP_DICT[K])     # :183
    rec_sampled_xyz = decode_sampled_np(octree_codes, 1)                                       # :184 (octree_np.decode, bug-compatible)
    x_patches = (knn_points(rec_sampled_xyz, batch_x, K) - rec_sampled_xyz) * (N / N0) ** (1/3)   # :192-199
    patches_pred, bottleneck, latent_quantized = ae(x_patches)                                 # :200 (AE.py:34-55)
    pmf = prob(rec_sampled_xyz); feature_bits = estimate_bits_from_pmf(pmf, sym) / (B * N)    # :204-208
    loss = chamfer_distance(pc_pred, batch_x) + lambda * feature_bits / (B * N)                # :211-222 (AE.py:61-70)
    loss.backward(); optimizer.step()                                                          # :229-230 (Adam over ae + prob)

Every step of the selection (normalize, FPS, octree depth search / encode / decode, kNN patches, in-patch 16-NN) is the codec's own HIP
kernel; the two networks are evaluated layer by layer through the autograd Functions of pccx.train (each a pair of HIP launches behind
the C ABI: pccx_linear / pccx_linear_dw / pccx_group_max_arg / pccx_chamfer_grad ...), so torch.autograd only sequences the backward;
Adam is pccx.train.Adam (two launches over all tensors).  The models are pccx.models.AE / ConditionalProbabilityModel: their parameters
are the reference's state_dict, and the checkpoints this trainer writes load into the codec (compress.py:58) unchanged.
fp32 by default (the reference's CPU branch, contextlib.nullcontext at :175); autocast=True is the bf16 form of pccx.train.
AE.py has no BatchNorm on this path (bn=False at AE.py:16-27).  ``--model PPPF-AE`` (train-mode BatchNorm over GROUPED rows) is not built.
"""
import torch

from . import _lib, ops, train
from .train import GroupMaxFn, LinearFn, ReluFn

OCTREE_BPP_DICT = {1024: 0.07, 512: 0.125, 256: 0.25, 128: 0.5, 64: 1.0}      # pn_kit.py:17-23


class LinearReluFn(torch.autograd.Function):
    """relu(x W^T + b) as ONE launch (pccx_linear with its ReLU epilogue) whose backward masks dY with the saved output: the separate
    ReLU pass of pccx.train (z written, read back, y written) was a third of this step's activation traffic -- 64 patches x 256 points x
    16 neighbours = 262 144 rows through the set-abstraction stack.  Same values as LinearFn + ReluFn."""

    @staticmethod
    def forward(ctx, x, W, b):
        x = x.contiguous()
        W2 = W.reshape(W.shape[0], -1).contiguous()
        N, K = W2.shape
        ctx.flags = 2 if train._AUTOCAST else 0
        y = train._linear_raw(x, train._packed(W2, False), b, N, K, ctx.flags | 1)
        ctx.save_for_backward(x, W2, y)
        ctx.has_bias, ctx.wshape = b is not None, W.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W2, y = ctx.saved_tensors
        N, K = W2.shape
        M = x.shape[0]
        dz = torch.empty_like(y)
        _lib.call("pccx_relu_backward", dy.contiguous().data_ptr(), y.data_ptr(), y.numel(), dz.data_ptr(), train._stream())
        dx = train._linear_raw(dz, train._packed(W2, True), None, K, N, ctx.flags) if ctx.needs_input_grad[0] else None
        dW, _ = train._zeros(tuple(W2.shape), torch.float32, dz.device)
        _lib.call("pccx_linear_dw", dz.data_ptr(), x.data_ptr(), M, N, K, N, x.stride(0), dW.data_ptr(), ctx.flags, train._stream())
        db = None
        if ctx.has_bias:
            db = torch.empty(N, device=dz.device, dtype=torch.float32)
            sums, pre = train._sums(N, dz.device)
            _lib.call("pccx_col_sum_w", dz.data_ptr(), M, N, sums.data_ptr(), db.data_ptr(), pre, train._stream())
        return dx, dW.view(ctx.wshape), db


def _lin(x, conv, relu):
    N_, K_ = conv.weight.shape[0], conv.weight[0].numel()
    if relu and x.shape[0] > 8 and not train._is_wide(x.shape[0], N_, K_):    # (a few rows: pccx.train's skinny / role-swapped layers)
        return LinearReluFn.apply(x, conv.weight, conv.bias)
    y = LinearFn.apply(x, conv.weight, conv.bias)
    return ReluFn.apply(y) if relu else y


def inpatch_neighbour_rows(x, nk):
    """pn_kit.py:190-191 with S == N: for every point of every patch its nk nearest neighbours inside the patch, relative to the point ->
    rows (P * K * nk, 3).  nk = 16 (AE.py:16) takes the codec's own selection kernel (pccx_patch_knn16: a byte per index; the SET of
    pytorch3d's knn_points -- the order inside a group does not matter to the max-pool that follows) and a gather."""
    P, K, _ = x.shape
    if nk == 16 and K % 16 == 0 and 16 <= K <= 1024:
        lib = _lib.load()
        ib = int(lib.pccx_patch_knn16_index_bytes(K))
        tab = torch.empty(int(lib.pccx_patch_knn16_bytes(P, K)), device=x.device, dtype=torch.uint8)
        _lib.call("pccx_patch_knn16", x.data_ptr(), P, K, tab.data_ptr(), train._stream())
        idx = tab.view(P, K, 16).to(torch.int64) if ib == 1 else (tab.view(torch.int16).view(P, K, 16).to(torch.int64) & 0xFFFF)
        return (ops.index_points(x, idx) - x.view(P, K, 1, 3)).reshape(P * K * 16, 3).contiguous()
    return ops.knn_points(x, x, nk, patch_scale=1.0).knn.reshape(P * K * nk, 3).contiguous()


def ae_forward_train(ae, x_patches):
    """AE.forward (AE.py:34-55) recording a graph: x_patches (P, K, 3) -> (new_xyz (P, k, 3), latent (P, d), latent_quantized (P, d))."""
    P, K, _ = x_patches.shape
    sa, pn = ae.sa, ae.pn
    if sa.npoint != K:
        raise _lib.PccxError(f"train_ipdae: AE was built for K={sa.npoint}, patches have {K} points")
    x = x_patches.detach().contiguous()
    with torch.no_grad():
        rows = inpatch_neighbour_rows(x, sa.K)                                        # pn_kit.py:190-191
    h = _lin(rows, sa.conv0, True)                                                    # :198
    h = _lin(h, sa.conv1, True)                                                       # :199
    h = _lin(h, sa.conv2, sa.finalRelu)                                               # :201-205
    feat = GroupMaxFn.apply(h.view(P * K, sa.K, -1))                                  # :207 max over the 16 neighbours
    h = torch.cat([x.reshape(P * K, 3), feat], dim=1)                                 # AE.py:39 (xyz first)
    for m in pn.mlp_Modules:
        h = _lin(h, m[0], len(m) > 1)                                                 # pn_kit.py:138-139
    z = GroupMaxFn.apply(h.view(P, K, -1))                                            # :141 max over the patch's points
    spread = ae.L - 0.2                                                               # AE.py:43
    latent = torch.sigmoid(z) * spread - spread / 2                                   # :44
    q = ops.ste_round(latent)                                                         # :45 (round, straight-through gradient)
    h = q
    lins = [m for m in ae.inv_pool if isinstance(m, torch.nn.Linear)]
    for m in lins:
        h = _lin(h, m, True)                                                          # :19-26: ReLU after each of the three
    k = ae.k
    lo = h.view(P, -1, k).permute(0, 2, 1).reshape(P * k, -1)                         # :49 view(BS, -1, k), as rows per output point
    h = torch.cat([lo, q[:, None, :].expand(P, k, q.shape[1]).reshape(P * k, -1)], dim=1)   # :50-51 (linear output first)
    for m in ae.inv_mlp.mlp_Modules:
        h = _lin(h, m[0], len(m) > 1)                                                 # :52
    return h.view(P, k, 3), latent, q


def prob_forward_train(prob, sampled_xyz):
    """ConditionalProbabilityModel.forward (AE.py:107-123) recording a graph: (B, S, 3) -> pmf (B, S, d, L)."""
    B, S, _ = sampled_xyz.shape
    rows = sampled_xyz.detach().reshape(B * S, 3).contiguous()
    h = rows
    for m in prob.model_pn.mlp_Modules:
        h = _lin(h, m[0], len(m) > 1)
    feat = GroupMaxFn.apply(h.view(B, S, -1))                                         # :112 (B, 256)
    h = torch.cat([rows, feat[:, None, :].expand(B, S, feat.shape[1]).reshape(B * S, -1)], dim=1)   # :115 (xyz first)
    convs = [m for m in prob.model_mlp if isinstance(m, torch.nn.Conv2d)]
    for i, m in enumerate(convs):
        h = _lin(h, m, i + 1 < len(convs))                                            # :117
    return torch.softmax(h.view(B, S, prob.d, prob.L), dim=3)                         # :118-120


def estimate_bits_from_pmf(pmf, sym):
    """pn_kit.estimate_bits_from_pmf (pn_kit.py:439-450)"""
    L = pmf.shape[-1]
    p = torch.gather(pmf.reshape(-1, L), 1, sym.reshape(-1, 1))
    return torch.sum(-torch.log2(p.clamp(min=1e-3)))


def select_patches(batch_x, S, K, N, N0, starts):
    """train.py:171-199 -- everything in front of the networks: depends on the data and the FPS start indices only.
    -> (batch_x normalised (B,N,3), rec_sampled_xyz (B,S,3), x_patches (B*S,K,3) scaled, sampled_bits (device scalar), scale)
    Nothing here reads a result on the host: the whole step can be captured into a hipGraph (GraphedIpdaeStep)."""
    B = batch_x.shape[0]
    if S != 64:
        raise _lib.PccxError(f"train_ipdae: octree_np.decode returns 64 centres whatever it is given (octree_np.py:100-111); S = N * ALPHA / K "
                             f"must be 64 (got {S})")
    if K not in OCTREE_BPP_DICT:
        raise _lib.PccxError(f"train_ipdae: --K must be one of {sorted(OCTREE_BPP_DICT)} (pn_kit.py:17-23), got {K}")
    # :171 -- pn_kit.normalize reads the centre and the longest side from pc[0] ONLY (pn_kit.py:50-53: "one point cloud"; train.py:40 says
    # the batch size must be 1) and applies them to whatever it is given: a batch is normalised by its FIRST cloud's box.  Reproduced.
    x, c0, l0 = ops.normalize(batch_x[:1].contiguous(), margin=0.01)
    if B > 1:
        rest = ((batch_x[1:].float() - c0.view(1, 1, 3)) * (1 - 0.01)) / l0.view(1, 1, 1) + 0.5       # pn_kit.py:55-57, op for op
        x = torch.cat([x, rest], dim=0).contiguous()
    sampled = ops.index_points(x, ops.farthest_point_sample_batch(x, S, starts))      # :178
    enc = ops.octree_encode(sampled, N, OCTREE_BPP_DICT[K])                           # :183
    rec, _ = ops.octree_decode(enc["bytes"], enc["nbytes"], "reference", S)           # :184
    sampled_bits = enc["nbits"].sum()                                                 # codebits: the streams' lengths in bits (device scalar)
    scale = float((N / N0) ** (1 / 3))
    patches = ops.knn_points(rec, x, K, patch_scale=scale).knn.view(B * S, K, 3)      # :192-199
    return x, rec, patches, sampled_bits, scale


def forward_loss(ae, prob, batch_x, starts, lam, S, K, N, N0):
    """the forward of train.py:171-222 -> (loss, fbpp, bpp) with the graph of loss recorded"""
    B = batch_x.shape[0]
    x, rec, patches, sampled_bits, scale = select_patches(batch_x, S, K, N, N0, starts)
    patches_pred, _, q = ae_forward_train(ae, patches)                                # :200
    patches_pred = patches_pred / scale                                               # :201
    pmf = prob_forward_train(prob, rec)                                               # :204
    sym = (q.detach().view(B, S, ae.d) + ae.L // 2).long().clamp(0, ae.L - 1)         # :205-206
    feature_bits = estimate_bits_from_pmf(pmf, sym) / (B * N)                         # :208
    bpp = (sampled_bits + feature_bits) / (B * N)                                     # :211
    fbpp = feature_bits / (B * N)                                                     # :212
    pc_pred = (patches_pred.view(B, S, -1, 3) + rec.view(B, S, 1, 3)).reshape(B, -1, 3)    # :214-216
    d, _ = ops.chamfer_distance(pc_pred, x)                                           # AE.py:67
    return d + lam * fbpp, fbpp, bpp                                                  # AE.py:68-70


class IpdaeTrainer:
    """The state of train.py's loop for ``--model AE``: the two models, Adam over both (train.py:131-134), the step counter with the
    rate term switched on at rate_loss_enable_step (:218-221) and the learning-rate decay (:250-254)."""

    def __init__(self, ae, prob, N=8192, N0=1024, ALPHA=2, K=256, lr=0.0005, lamda=1e-6, rate_loss_enable_step=40000, lr_decay=0.1,
                 lr_decay_steps=60000, autocast=False):
        self.ae, self.prob = ae, prob
        self.N, self.N0, self.K, self.S = int(N), int(N0), int(K), int(N) * int(ALPHA) // int(K)
        self.lamda, self.rate_loss_enable_step = float(lamda), int(rate_loss_enable_step)
        self.lr, self.lr_decay, self.lr_decay_steps = float(lr), float(lr_decay), int(lr_decay_steps)
        self.autocast = bool(autocast)
        self.opt = train.Adam(list(ae.parameters()) + list(prob.parameters()), lr=self.lr)
        self.global_step = 0

    def step(self, batch_x, starts):
        """one iteration (train.py:162-254) -> dict(loss, fbpp, bpp) of python floats; starts: the FPS start index per cloud (:178 draws it
        with torch.randint inside farthest_point_sample_batch, pn_kit.py:321)"""
        if batch_x.dim() != 3 or batch_x.shape[1] != self.N or batch_x.shape[2] != 3:
            raise _lib.PccxError(f"train_ipdae: batch must be (B, {self.N}, 3), got {tuple(batch_x.shape)}")
        for p in self.opt.params:
            p.grad = None                                                             # :173 optimizer.zero_grad()
        lam = 0.0 if self.global_step < self.rate_loss_enable_step else self.lamda    # :218-221
        train._AUTOCAST = self.autocast
        train._ARENA = train._EAGER_ARENA if batch_x.is_cuda else None
        if train._ARENA is not None:
            train._ARENA.begin(batch_x.device)
        try:
            loss, fbpp, bpp = forward_loss(self.ae, self.prob, batch_x, starts, lam, self.S, self.K, self.N, self.N0)
            train._AUTOCAST = False
            loss.backward()                                                           # :229
            self.opt.step(max_norm=None)                                              # :230
        finally:
            train._AUTOCAST = False
            if train._ARENA is not None:
                train._ARENA.end(batch_x.device)
            train._ARENA = None
        self.global_step += 1                                                         # :236
        if self.global_step % self.lr_decay_steps == 0:                               # :250-254
            self.lr *= self.lr_decay
            self.opt.set_lr(self.lr)
        return dict(loss=float(loss.detach()), fbpp=float(fbpp.detach()), bpp=float(bpp.detach()))

    def graphed(self, batch_x, starts, warmup=2):
        """-> GraphedIpdaeStep over this trainer's models and optimizer (the warm-up iterations are real steps on batch_x)"""
        return GraphedIpdaeStep(self, batch_x, starts, warmup)


class GraphedIpdaeStep:
    """IpdaeTrainer.step captured ONCE as a hipGraph and replayed: the iteration is ~400 small launches at the reference's batch size of 1
    (64 patches), bound by launch latency, not by arithmetic.  Selection included: FPS, the octree depth search / encode / decode and
    both kNN searches are stream-ordered kernels that read the batch and the start indices from device memory.  What changes from step to
    step lives in buffers the graph reads: the batch, the start indices, lambda (0 until rate_loss_enable_step) and Adam's lr / bias
    corrections (Adam.make_capturable).  Shapes are fixed at construction."""

    def __init__(self, trainer, batch_x, starts, warmup=2):
        tr = self.tr = trainer
        dev = batch_x.device
        self.arena = train.StepArena()
        tr.opt.make_capturable(dev)
        self.x = batch_x.detach().to(torch.float32).clone().contiguous()
        self.starts = torch.as_tensor(starts).to(device=dev, dtype=torch.int32).contiguous().clone()
        self.lam = torch.zeros((), device=dev, dtype=torch.float32)
        self.warm_out = None
        if warmup > 0:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    self._set_lam()
                    self.warm_out = self._body()                     # (loss, fbpp, bpp) of the last warm-up iteration, device scalars
                    self._advance()
            torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):        # see pccx.train.GraphedTrainStep on the mode
            self.out = self._body()
        self._grads = tr.opt._keep
        tr.opt.flush_table()
        for p in tr.opt.params:
            p.grad = None

    def _set_lam(self):
        self.lam.fill_(0.0 if self.tr.global_step < self.tr.rate_loss_enable_step else self.tr.lamda)

    def _advance(self):
        tr = self.tr
        tr.global_step += 1
        if tr.global_step % tr.lr_decay_steps == 0:
            tr.lr *= tr.lr_decay
            tr.opt.set_lr(tr.lr)

    def _body(self):
        tr = self.tr
        for p in tr.opt.params:
            p.grad = None
        train._AUTOCAST = tr.autocast
        train._ARENA = self.arena
        self.arena.begin(self.x.device)
        try:
            loss, fbpp, bpp = forward_loss(tr.ae, tr.prob, self.x, self.starts, self.lam, tr.S, tr.K, tr.N, tr.N0)
            train._AUTOCAST = False
            loss.backward()
            tr.opt.step(max_norm=None)
        finally:
            train._AUTOCAST = False
            train._ARENA = None
            self.arena.end(self.x.device)
        return loss.detach(), fbpp.detach(), bpp.detach()

    def __call__(self, batch_x=None, starts=None, sync=True):
        """one iteration on (batch_x, starts) (None = the buffers' current content) -> dict of floats, or of device scalars (sync=False)"""
        if batch_x is not None:
            self.x.copy_(batch_x)
        if starts is not None:
            self.starts.copy_(torch.as_tensor(starts).to(self.x.device, torch.int32))
        self._set_lam()
        self.graph.replay()
        self.tr.opt.t += 1
        self._advance()
        keys = ("loss", "fbpp", "bpp")
        return {k: float(v) for k, v in zip(keys, self.out)} if sync else dict(zip(keys, self.out))
