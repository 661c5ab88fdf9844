"""Host-side mirror of the reference's model classes (AE.py, pn_kit.py) over libpccx.so.

The classes keep the reference's constructor arguments and ``state_dict`` key names (SURVEY
Appendix C) so reference checkpoints load with ``load_state_dict``; torch modules are used as
parameter containers only.  Every forward computation is a HIP kernel behind the C ABI; there
is no torch / CPU compute path (inference only, as compress.py / decompress.py use the models).
"""
import torch
import torch.nn as nn

from . import _lib
from .ops import _stream, _f32c, stage


def _conv_stack(chans, relu):
    mods = nn.ModuleList()
    for i in range(len(chans) - 1):
        layers = [nn.Conv2d(chans[i], chans[i + 1], 1)]
        if relu[i]:
            layers.append(nn.ReLU())
        mods.append(nn.Sequential(*layers))
    return mods


def _pccx_default_matmul():
    from . import DEFAULT_MATMUL
    return DEFAULT_MATMUL


def _folded(conv, relu, device):
    from .families import FoldedLinear          # generic runtime-shaped fp32 MFMA layer (csrc/linear.hip)
    return FoldedLinear(conv.weight, conv.bias, relu, None, device)


class _Params(nn.Module):
    """Sub-modules of the reference's models.  They hold the parameters under the reference's state_dict keys AND are
    callable like the reference's (compress.py:113-121 calls ae.sa(x) / ae.pn(x), decompress.py:97-101 ae.inv_pool /
    ae.inv_mlp): forward runs HIP kernels -- the owning AE's fused entry points when the module is one of AE's and the
    shapes are the fused kernel's, otherwise the generic layer kernels (pccx_linear / pccx_group_max).  No torch compute."""

    _fused = None                 # set by the owning AE: a callable taking the reference's arguments, or None
    _layers = None                # generic path: packed layers, built lazily on the input's device

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self._layers = None
        return r

    def _own(self, fused):
        object.__setattr__(self, "_fused", fused)      # not a sub-module / buffer: plain attribute


class _ConvStack(_Params):
    def _stack(self, device):
        if self._layers is None or self._layers[0] != torch.device(device):
            object.__setattr__(self, "_layers", (torch.device(device), [_folded(m[0], len(m) > 1, device) for m in self.mlp_Modules]))
        return self._layers[1]

    def _rows(self, points):
        x = _f32c(points, type(self).__name__)
        if x.dim() != 3:
            raise _lib.PccxError(f"{type(self).__name__}: expected [B, C, N], got {tuple(x.shape)}")
        B, Cc, N = x.shape
        rows = x.permute(0, 2, 1).reshape(B * N, Cc).contiguous()          # channels-last rows
        for layer in self._stack(x.device):
            rows = layer(rows)
        return rows, B, N


class SetAbstraction(_Params):      # pn_kit.SetAbstraction (pn_kit.py:146-161), bn=False
    def __init__(self, npoint, K, in_channel, mlp, bn=False, finalRelu=True):
        super().__init__()
        if bn:
            raise _lib.PccxError("pccx.SetAbstraction: bn=True is not on the codec path (AE.py:16 uses bn=False)")
        self.npoint, self.K, self.finalRelu = npoint, K, finalRelu
        self.conv0 = nn.Conv2d(in_channel + 3, mlp[0], 1)
        self.conv1 = nn.Conv2d(mlp[0], mlp[1], 1)
        self.conv2 = nn.Conv2d(mlp[1], mlp[2], 1)

    def forward(self, xyz, start_idx=None):
        """pn_kit.SetAbstraction.forward (pn_kit.py:164-211): xyz [B, 3, N] -> (new_xyz [B, 3, S], new_points [B, D', S])."""
        x = _f32c(xyz, "SetAbstraction")
        B, Cc, N = x.shape
        if self._fused is not None and self.npoint == N and Cc == 3:
            return self._fused(x)
        from .families import group_max
        from . import ops
        pts = x.permute(0, 2, 1).contiguous()                                              # :173
        S = self.npoint
        new_xyz = pts if S == N else ops.index_points(pts, ops.farthest_point_sample_batch(pts, S, start_idx))   # :180-183
        nn_ = ops.knn_points(new_xyz, pts, self.K, patch_scale=1.0)                        # :190-191 (nn - centre)
        rows = nn_.knn.reshape(B * S * self.K, Cc).contiguous()
        if self._layers is None or self._layers[0] != x.device:
            object.__setattr__(self, "_layers", (x.device, [_folded(self.conv0, True, x.device), _folded(self.conv1, True, x.device),
                                                            _folded(self.conv2, self.finalRelu, x.device)]))
        for layer in self._layers[1]:
            rows = layer(rows)                                                             # :198-205
        feat = group_max(rows.view(B * S, self.K, -1)).view(B, S, -1)                      # :207
        return new_xyz.permute(0, 2, 1), feat.permute(0, 2, 1)                             # :209-211


class PointNet(_ConvStack):         # pn_kit.PointNet (pn_kit.py:98-121), bn=False
    def __init__(self, in_channel, mlps, relu, bn=False):
        super().__init__()
        if bn:
            raise _lib.PccxError("pccx.PointNet: bn=True is not on the codec path (AE.py:17 uses bn=False)")
        self.mlp_Modules = _conv_stack([in_channel] + list(mlps), relu)

    def forward(self, points):
        """pn_kit.PointNet.forward (pn_kit.py:124-144): points [B, C, N] -> [B, D] (max over the N points)."""
        if self._fused is not None:
            r = self._fused(points)
            if r is not None:
                return r
        from .families import group_max
        rows, B, N = self._rows(points)
        return group_max(rows.view(B, N, -1))


class MLP(_ConvStack):              # pn_kit.MLP (pn_kit.py:263-286), bn=False
    def __init__(self, in_channel, mlps, relu, bn=False):
        super().__init__()
        if bn:
            raise _lib.PccxError("pccx.MLP: bn=True is not on the codec path (AE.py:27 uses bn=False)")
        self.mlp_Modules = _conv_stack([in_channel] + list(mlps), relu)

    def forward(self, points):
        """pn_kit.MLP.forward (pn_kit.py:288-305): points [B, C, N] -> [B, D, N]."""
        rows, B, N = self._rows(points)
        return rows.view(B, N, -1).permute(0, 2, 1)


class LinearStack(nn.Sequential):
    """nn.Sequential of Linear / ReLU (AE.py:19-26 inv_pool) with the reference's state_dict keys; forward = one
    pccx_linear launch per Linear (ReLU fused)."""

    _layers = None

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self._layers = None
        return r

    def forward(self, x):
        x = _f32c(x, "inv_pool")
        if self._layers is None or self._layers[0] != x.device:
            mods, layers = list(self), []
            for i, m in enumerate(mods):
                if isinstance(m, nn.Linear):
                    layers.append(_folded(m, i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU), x.device))
            object.__setattr__(self, "_layers", (x.device, layers))
        lead = x.shape[:-1]
        rows = x.reshape(-1, x.shape[-1]).contiguous()
        for layer in self._layers[1]:
            rows = layer(rows)
        return rows.view(*lead, -1)


def _host(t):
    return t.detach().to("cpu", torch.float32).contiguous()


_WS = {}


def workspace(tag, numel, device):
    """Persistent scratch buffers (HBM is 288 GB: keep the big intermediates resident instead of going
    through the allocator every call).  Stream-ordered reuse is safe: every consumer of a buffer is
    enqueued before its next producer on the same stream."""
    key = (tag, str(device), torch.cuda.current_stream().cuda_stream)      # one buffer per stream: streams run concurrently
    t = _WS.get(key)
    if t is None or t.numel() < numel:
        _WS[key] = t = torch.empty(int(numel), device=device, dtype=torch.float32)
    return t[:numel]


def _pack(fn_name, size, tensors, ints):
    keep = [_host(t) for t in tensors]
    blob = torch.zeros(size, dtype=torch.float32)
    _lib.call(fn_name, *[t.data_ptr() for t in keep], *ints, blob.data_ptr())
    return blob


class AE(nn.Module):
    """AE.AE (AE.py:12-55): SetAbstraction + PointNet encoder, Linear + MLP decoder."""

    def __init__(self, K, k, d, L):
        super().__init__()
        if d < 1 or L < 1 or K % 16 != 0 or not 16 <= K <= 1024:
            raise _lib.PccxError(
                f"pccx.AE: --K must be a multiple of 16 in 16..1024 (got {K}; the reference's octree rate table pn_kit.py:17-23 has "
                f"64..1024 only) and --d, --L positive (got {d}, {L}); the reference's defaults are --K 256 --d 16 --L 7 "
                f"(compress.py:30-34)")
        # the fused transforms cover the bottleneck widths --d 1..16; wider ones (compress.py:31 accepts any) run PointNet and the
        # decoder through the generic layer kernels, chunked over the patches (correct and slow: encode_generic / decode_generic)
        self.fused_d = d <= 16
        self.sa = SetAbstraction(npoint=K, K=16, in_channel=0, mlp=[32, 64, 128])
        self.pn = PointNet(3 + 128, [128, 256, 512, d], [True, True, True, False])
        self.inv_pool = LinearStack(nn.Linear(d, 256), nn.ReLU(), nn.Linear(256, 1024), nn.ReLU(),
                                    nn.Linear(1024, k * 128), nn.ReLU())
        self.inv_mlp = MLP(d + 128, [128, 64, 32, 3], [True, True, True, False])
        self.K, self.k, self.d, self.L = K, k, d, L
        self._enc_blob = self._dec_blob = None
        self.sa._own(self._sa_call)
        self.pn._own(self._pn_call)

    def quantize(self, x):          # AE.STEQuantize.forward (AE.py:79-81); straight-through gradient (AE.py:83-85)
        from .ops import ste_round
        return ste_round(x)

    # ---- the reference's per-module calls (compress.py:113-121), on the fused kernels ----------------------------
    def _sa_call(self, x):
        """ae.sa(x): x [P, 3, K] -> (new_xyz [P, 3, K] (= x, npoint == K, pn_kit.py:180-181), features [P, 128, K])."""
        P, _, K = x.shape
        if K % 16 != 0 or not 16 <= K <= 1024:
            raise _lib.PccxError(f"ae.sa: K={K} must be a multiple of 16 in 16..1024")
        patches = x.permute(0, 2, 1).contiguous()
        feat = torch.empty(P, 8, K, 16, device=x.device, dtype=torch.float32)
        self._launch_sa(patches, feat, _pccx_default_matmul())
        return x, feat.permute(0, 1, 3, 2).reshape(P, 128, K)

    def _pn_call(self, points):
        """ae.pn(cat(x_patches, features)): [P, 3 + 128, K] -> raw latent [P, d] (before the sigmoid of compress.py:126)."""
        x = _f32c(points, "ae.pn")
        if not self.fused_d or x.dim() != 3 or x.shape[1] != 131 or x.shape[2] % 16 != 0 or not 16 <= x.shape[2] <= 1024:
            return None                                   # not the fused kernel's shape: generic path
        P, _, K = x.shape
        patches = x[:, :3].permute(0, 2, 1).contiguous()
        feat = x[:, 3:].reshape(P, 8, 16, K).permute(0, 1, 3, 2).contiguous()
        outs = [torch.empty(P, self.d, device=x.device, dtype=torch.float32) for _ in range(3)]
        self._launch_pn(patches, feat, outs, _pccx_default_matmul())
        return outs[0]

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self._enc_blob = self._dec_blob = None
        return r

    def pack(self, device="cuda"):
        """Build the MFMA fragment blobs (C ABI pccx_pack_*) and upload them."""
        sd = self.state_dict()
        w = lambda key: sd[key].reshape(sd[key].shape[0], -1)
        lib = _lib.load()
        if not self.fused_d:
            # only the SetAbstraction part of the encoder blob is used (ae.sa does not depend on --d): pack it with a 16-wide stand-in
            # for PointNet's last layer; PointNet and the decoder run through the generic layers (their own lazily packed weights)
            pad_w, pad_b = torch.zeros(16, 512), torch.zeros(16)
            enc = _pack("pccx_pack_ae_encoder", lib.pccx_ae_encoder_blob_floats(),
                        [w("sa.conv0.weight"), sd["sa.conv0.bias"], w("sa.conv1.weight"), sd["sa.conv1.bias"],
                         w("sa.conv2.weight"), sd["sa.conv2.bias"],
                         w("pn.mlp_Modules.0.0.weight"), sd["pn.mlp_Modules.0.0.bias"],
                         w("pn.mlp_Modules.1.0.weight"), sd["pn.mlp_Modules.1.0.bias"],
                         w("pn.mlp_Modules.2.0.weight"), sd["pn.mlp_Modules.2.0.bias"], pad_w, pad_b], [16])
            self._enc_blob, self._dec_blob = enc.to(device), None
            self._dec_b3 = self._sa_b3 = self._pn_b3 = self._enc_h2 = self._dec_h2 = None
            return self
        enc = _pack("pccx_pack_ae_encoder", lib.pccx_ae_encoder_blob_floats(),
                    [w("sa.conv0.weight"), sd["sa.conv0.bias"], w("sa.conv1.weight"), sd["sa.conv1.bias"],
                     w("sa.conv2.weight"), sd["sa.conv2.bias"],
                     w("pn.mlp_Modules.0.0.weight"), sd["pn.mlp_Modules.0.0.bias"],
                     w("pn.mlp_Modules.1.0.weight"), sd["pn.mlp_Modules.1.0.bias"],
                     w("pn.mlp_Modules.2.0.weight"), sd["pn.mlp_Modules.2.0.bias"],
                     w("pn.mlp_Modules.3.0.weight"), sd["pn.mlp_Modules.3.0.bias"]], [self.d])
        dec = _pack("pccx_pack_ae_decoder", lib.pccx_ae_decoder_blob_floats(self.k),
                    [sd["inv_pool.0.weight"], sd["inv_pool.0.bias"], sd["inv_pool.2.weight"], sd["inv_pool.2.bias"],
                     sd["inv_pool.4.weight"], sd["inv_pool.4.bias"],
                     w("inv_mlp.mlp_Modules.0.0.weight"), sd["inv_mlp.mlp_Modules.0.0.bias"],
                     w("inv_mlp.mlp_Modules.1.0.weight"), sd["inv_mlp.mlp_Modules.1.0.bias"],
                     w("inv_mlp.mlp_Modules.2.0.weight"), sd["inv_mlp.mlp_Modules.2.0.bias"],
                     w("inv_mlp.mlp_Modules.3.0.weight"), sd["inv_mlp.mlp_Modules.3.0.bias"]], [self.k, self.d])
        self._enc_blob, self._dec_blob = enc.to(device), dec.to(device)
        self._dec_b3 = self._sa_b3 = self._pn_b3 = self._enc_h2 = self._dec_h2 = None
        return self

    def _enc_tensors(self):
        sd = self.state_dict()
        w = lambda key: sd[key].reshape(sd[key].shape[0], -1)
        return [w("sa.conv0.weight"), sd["sa.conv0.bias"], w("sa.conv1.weight"), sd["sa.conv1.bias"],
                w("sa.conv2.weight"), sd["sa.conv2.bias"],
                w("pn.mlp_Modules.0.0.weight"), sd["pn.mlp_Modules.0.0.bias"],
                w("pn.mlp_Modules.1.0.weight"), sd["pn.mlp_Modules.1.0.bias"],
                w("pn.mlp_Modules.2.0.weight"), sd["pn.mlp_Modules.2.0.bias"],
                w("pn.mlp_Modules.3.0.weight"), sd["pn.mlp_Modules.3.0.bias"]]

    def _dec_tensors(self):
        sd = self.state_dict()
        w = lambda key: sd[key].reshape(sd[key].shape[0], -1)
        return [sd["inv_pool.0.weight"], sd["inv_pool.0.bias"], sd["inv_pool.2.weight"], sd["inv_pool.2.bias"],
                sd["inv_pool.4.weight"], sd["inv_pool.4.bias"],
                w("inv_mlp.mlp_Modules.0.0.weight"), sd["inv_mlp.mlp_Modules.0.0.bias"],
                w("inv_mlp.mlp_Modules.1.0.weight"), sd["inv_mlp.mlp_Modules.1.0.bias"],
                w("inv_mlp.mlp_Modules.2.0.weight"), sd["inv_mlp.mlp_Modules.2.0.bias"],
                w("inv_mlp.mlp_Modules.3.0.weight"), sd["inv_mlp.mlp_Modules.3.0.bias"]]

    def _enc_h2_blob(self, device):
        """f16x2 operand planes, scaled biases and layer scales of the encoder (csrc/pack_h2.hip), packed on the host."""
        enc, _ = self._blobs(device)
        if getattr(self, "_enc_h2", None) is None or self._enc_h2.device != enc.device:
            self._enc_h2 = _pack("pccx_pack_ae_encoder_h2", _lib.load().pccx_ae_encoder_h2_blob_floats(), self._enc_tensors(),
                                 [self.d]).to(enc.device)
        return self._enc_h2

    def _dec_h2_blob(self, device):
        _, dec = self._blobs(device)
        if getattr(self, "_dec_h2", None) is None or self._dec_h2.device != dec.device:
            self._dec_h2 = _pack("pccx_pack_ae_decoder_h2", _lib.load().pccx_ae_decoder_h2_blob_floats(self.k), self._dec_tensors(),
                                 [self.k, self.d]).to(dec.device)
        return self._dec_h2

    def _blobs(self, device):
        if self._enc_blob is None or self._enc_blob.device != torch.device(device):
            self.pack(device)
        return self._enc_blob, self._dec_blob

    def _sa_b3_blob(self, device):
        """bf16x3 planes of the SetAbstraction conv1 / conv2 weights, built on the device."""
        enc, _ = self._blobs(device)
        if getattr(self, "_sa_b3", None) is None or self._sa_b3.device != enc.device:
            self._sa_b3 = torch.empty(_lib.load().pccx_sa_b3_blob_floats(), device=enc.device, dtype=torch.float32)
            _lib.call("pccx_pack_sa_b3", enc.data_ptr(), self._sa_b3.data_ptr(), _stream())
        return self._sa_b3

    def _pn_b3_blob(self, device):
        """bf16x3 planes of the PointNet weight stream, built on the device."""
        enc, _ = self._blobs(device)
        if getattr(self, "_pn_b3", None) is None or self._pn_b3.device != enc.device:
            self._pn_b3 = torch.empty(_lib.load().pccx_pn_b3_blob_floats(), device=enc.device, dtype=torch.float32)
            _lib.call("pccx_pack_pn_b3", enc.data_ptr(), self._pn_b3.data_ptr(), _stream())
        return self._pn_b3

    def _launch_sa(self, x, feat, matmul):
        P, K, _ = x.shape
        enc, _ = self._blobs(x.device)
        if matmul in ("bf16x3", "f16x2"):                 # f16x2 exists for the fused transforms only: the module alone runs bf16x3
            _lib.call("pccx_sa_forward_b3", x.data_ptr(), P, K, enc.data_ptr(), self._sa_b3_blob(x.device).data_ptr(),
                      feat.data_ptr(), _stream())
        elif matmul == "f32":
            _lib.call("pccx_sa_forward", x.data_ptr(), P, K, enc.data_ptr(), feat.data_ptr(), _stream())
        else:
            raise ValueError(f"sa_matmul={matmul!r}: expected 'f32', 'bf16x3' or 'f16x2'")

    def _launch_pn(self, x, feat, outs, matmul):
        P, K, _ = x.shape
        enc, _ = self._blobs(x.device)
        if matmul in ("bf16x3", "f16x2"):
            _lib.call("pccx_pn_forward_b3", x.data_ptr(), feat.data_ptr(), P, K, enc.data_ptr(), self._pn_b3_blob(x.device).data_ptr(),
                      self.d, self.L, outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(), _stream())
        elif matmul == "f32":
            _lib.call("pccx_pn_forward", x.data_ptr(), feat.data_ptr(), P, K, enc.data_ptr(), self.d, self.L,
                      outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(), _stream())
        else:
            raise ValueError(f"pn_matmul={matmul!r}: expected 'f32', 'bf16x3' or 'f16x2'")

    def encode(self, patches, sa_matmul=None, pn_matmul=None, fused=True):
        """patches (BS,K,3), centred and scaled -> (latent_raw, latent, latent_quantized), each (BS,d).
        = ae.sa + ae.pn + sigmoid spread + round (compress.py:113-127, AE.py:37-45).
        sa_matmul / pn_matmul: "f32" (exact-fp32 MFMA), "bf16x3" (fp32 products of three bf16 pieces per operand on the
        bf16 matrix cores) or "f16x2" (two exactly scaled fp16 pieces per operand, three products on the fp16 matrix cores;
        both split modes: fp32-level error, a latent within ~1e-6 of a rounding boundary may round the other way);
        None = pccx.DEFAULT_MATMUL.  "f16x2" exists as the fused kernel only (it holds every K up to 1024): with fused=False or different modes
        for the two modules, an "f16x2" request runs the bf16x3 kernels.  fused=False forces the
        two-kernel path (feature map through HBM)."""
        x = _f32c(patches, "AE.encode")
        P, K, _ = x.shape
        if not self.fused_d:
            return self.encode_generic(x)
        outs = [torch.empty(P, self.d, device=x.device, dtype=torch.float32) for _ in range(3)]
        sa_matmul, pn_matmul = sa_matmul or _pccx_default_matmul(), pn_matmul or _pccx_default_matmul()
        if fused and sa_matmul == pn_matmul == "f16x2" and _lib.load().pccx_ae_encode_h2_fused_ok(K):
            # the fused kernel on f16x2 operands (csrc/encoder_fused_h2.hip): two fp16 pieces per operand, three MFMA passes
            enc, _ = self._blobs(x.device)
            nbytes = _lib.load().pccx_ae_encode_h2_workspace_bytes(P, K)
            ws = workspace("patch_knn16", (nbytes + 3) // 4, x.device)
            # the two launches of pccx_ae_encode_h2_ws as two calls, so that a stage timer sees each kernel on its own
            with stage("patch_knn16"):
                _lib.call("pccx_patch_knn16", x.data_ptr(), P, K, ws.data_ptr(), _stream())
            with stage("sa_pn_forward"):
                _lib.call("pccx_ae_encode_h2_tables", x.data_ptr(), P, K, enc.data_ptr(), self._enc_h2_blob(x.device).data_ptr(), self.d, self.L,
                          outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(), ws.data_ptr(), _stream())
            return tuple(outs)
        if sa_matmul == "f16x2" or pn_matmul == "f16x2":      # K beyond the fused kernel, or fused=False: the bf16x3 kernels
            sa_matmul = "bf16x3" if sa_matmul == "f16x2" else sa_matmul
            pn_matmul = "bf16x3" if pn_matmul == "f16x2" else pn_matmul
        if fused and sa_matmul == pn_matmul == "bf16x3" and _lib.load().pccx_ae_encode_b3_fused_ok(K):
            # one kernel, the (P,128,K) feature map never leaves the CU (csrc/encoder_fused.hip)
            # (csrc/encoder_fused.hip); the in-patch 16-NN tables come from a kernel of their own (csrc/patch_knn.hip) through
            # a persistent workspace: 4 KB per 256-point patch
            enc, _ = self._blobs(x.device)
            nbytes = _lib.load().pccx_ae_encode_b3_workspace_bytes(P, K)
            ws = workspace("patch_knn16", (nbytes + 3) // 4, x.device)
            with stage("patch_knn16"):
                _lib.call("pccx_patch_knn16", x.data_ptr(), P, K, ws.data_ptr(), _stream())
            with stage("sa_pn_forward"):
                _lib.call("pccx_ae_encode_b3_tables", x.data_ptr(), P, K, enc.data_ptr(), self._sa_b3_blob(x.device).data_ptr(),
                          self._pn_b3_blob(x.device).data_ptr(), self.d, self.L, outs[0].data_ptr(), outs[1].data_ptr(),
                          outs[2].data_ptr(), ws.data_ptr(), _stream())
            return tuple(outs)
        ws = workspace("sa_feat", P * K * 128, x.device)
        with stage("sa_forward"):
            self._launch_sa(x, ws, sa_matmul)
        with stage("pn_forward"):
            self._launch_pn(x, ws, outs, pn_matmul)
        return tuple(outs)

    def encode_generic(self, x, chunk=1024):
        """encode() for bottleneck widths the fused PointNet does not cover (--d > 16): ae.sa on its kernel (it does not depend on d),
        ae.pn through the generic layers (pccx_linear* + pccx_group_max), sigmoid spread and round by their kernels -- the statement
        sequence of compress.py:113-127, ``chunk`` patches at a time (the (chunk, K, 131) rows are the memory bound)."""
        from .families import sigmoid_spread, round_
        raws = []
        for i in range(0, x.shape[0], chunk):
            xt = x[i:i + chunk].permute(0, 2, 1).contiguous()                  # (p, 3, K)
            _, feat = self.sa(xt)                                              # compress.py:113-115
            raws.append(self.pn(torch.cat((xt, feat), dim=1)))                 # compress.py:116-121
        raw = torch.cat(raws) if raws else torch.empty(0, self.d, device=x.device)
        latent = sigmoid_spread(raw, self.L)                                   # AE.py:43-44
        return raw, latent, round_(latent)                                     # AE.py:45

    def decode_generic(self, q, chunk=1024):
        """latent_q (P, d) -> raw decoder output (P, k, 3) through ae.inv_pool / ae.inv_mlp on the generic layers
        (decompress.py:97-102), ``chunk`` patches at a time (inv_pool's (chunk, k * 128) rows are the memory bound)."""
        outs = []
        for i in range(0, q.shape[0], chunk):
            lq = q[i:i + chunk]
            lin = self.inv_pool(lq).view(lq.shape[0], -1, self.k)                                  # :97-98
            mlp_in = torch.cat((lin, lq.unsqueeze(-1).tile((1, 1, self.k))), dim=1)                # :99-100
            outs.append(self.inv_mlp(mlp_in).transpose(2, 1).contiguous())                         # :101-102
        return torch.cat(outs) if outs else torch.empty(0, self.k, 3, device=q.device)

    def _b3_blob(self, device):
        """bf16x3 planes of the decoder's big Linear, built on the device from the packed fp32 blob."""
        _, dec = self._blobs(device)
        if getattr(self, "_dec_b3", None) is None or self._dec_b3.device != dec.device:
            self._dec_b3 = torch.empty(_lib.load().pccx_dec_b3_blob_floats(self.k), device=dec.device, dtype=torch.float32)
            _lib.call("pccx_pack_ae_decoder_b3", dec.data_ptr(), self.k, self._dec_b3.data_ptr(), _stream())
        return self._dec_b3

    def decode(self, latent_q, centres=None, center=None, longest=None, S=None, scale=None, margin=0.01, matmul=None):
        """latent_q (BS,d) -> decoded patches (BS,k,3) (AE.py:48-53).  With centres/center/longest/S/scale
        it returns instead the reassembled, denormalised cloud (B,S*k,3) of decompress.py:104-116.
        matmul="bf16x3" / "f16x2" evaluate the matrix products as fp32 products of three bf16 / two exactly scaled fp16 pieces
        per operand on the matrix cores (fp32-level error, not bit-identical to "f32"); None = pccx.DEFAULT_MATMUL."""
        matmul = matmul or _pccx_default_matmul()
        q = _f32c(latent_q, "AE.decode")
        P = q.shape[0]
        if not self.fused_d:
            raw = self.decode_generic(q)
            if centres is None:
                return raw
            B = P // S
            pc = torch.empty(B, S * self.k, 3, device=q.device, dtype=torch.float32)
            _lib.call("pccx_reassemble", raw.data_ptr(), P, self.k, float(scale), _f32c(centres.reshape(P, 3), "AE.decode.centres").data_ptr(),
                      _f32c(center.reshape(B, 3), "AE.decode.center").data_ptr(), _f32c(longest.reshape(B), "AE.decode.longest").data_ptr(),
                      int(S), float(margin), pc.data_ptr(), _stream())
            return pc
        _, dec = self._blobs(q.device)
        if matmul == "bf16x3":
            fn, extra = "pccx_ae_decode_b3", (self._b3_blob(q.device).data_ptr(),)
            ws = workspace("dec_h2_b3", _lib.load().pccx_ae_decode_b3_workspace_floats(P), q.device)
        elif matmul == "f16x2":
            fn, extra = "pccx_ae_decode_h2", (self._dec_h2_blob(q.device).data_ptr(),)
            ws = workspace("dec_h2_h2", _lib.load().pccx_ae_decode_h2_workspace_floats(P), q.device)
        elif matmul == "f32":
            fn, extra = "pccx_ae_decode", ()
            ws = workspace("dec_h2", _lib.load().pccx_ae_decode_workspace_floats(P), q.device)
        else:
            raise ValueError(f"matmul={matmul!r}: expected 'f32', 'bf16x3' or 'f16x2'")
        if centres is None:
            out = torch.empty(P, self.k, 3, device=q.device, dtype=torch.float32)
            _lib.call(fn, q.data_ptr(), P, self.d, self.k, dec.data_ptr(), *extra, ws.data_ptr(), out.data_ptr(),
                      0.0, None, None, None, 1, float(margin), None, _stream())
            return out
        B = P // S
        centres = _f32c(centres.reshape(P, 3), "AE.decode.centres")
        center = _f32c(center.reshape(B, 3), "AE.decode.center")
        longest = _f32c(longest.reshape(B), "AE.decode.longest")
        pc = torch.empty(B, S * self.k, 3, device=q.device, dtype=torch.float32)
        _lib.call(fn, q.data_ptr(), P, self.d, self.k, dec.data_ptr(), *extra, ws.data_ptr(), None, float(scale),
                  centres.data_ptr(), center.data_ptr(), longest.data_ptr(), int(S), float(margin), pc.data_ptr(), _stream())
        return pc

    def forward(self, xyz):
        """AE.AE.forward (AE.py:34-55), inference: xyz (BS,K,3) -> (new_xyz (BS,k,3), latent, latent_quantized)."""
        _, latent, q = self.encode(xyz)
        return self.decode(q), latent, q


class ConditionalProbabilityModel(nn.Module):
    """AE.ConditionalProbabilityModel (AE.py:87-123)."""

    def __init__(self, L, d):
        super().__init__()
        self.L, self.d = L, d
        self.model_pn = PointNet(3, [64, 128, 256], [True, True, True])
        self.model_mlp = nn.Sequential(nn.Conv2d(3 + 256, 512, 1), nn.ReLU(), nn.Conv2d(512, 512, 1), nn.ReLU(),
                                       nn.Conv2d(512, d * L, 1))
        self._blob = None

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self._blob = None
        return r

    def fused_ok(self, S=16):
        """The fused kernel covers d <= 16, L <= 15, d * L <= 128 and S a multiple of 16; everything else takes the generic layers."""
        return self.d <= 16 and self.L <= 15 and self.d * self.L <= 128 and S % 16 == 0 and S >= 16

    def pack(self, device="cuda"):
        if not self.fused_ok():
            self._blob = None
            self._generic = None
            return self
        sd = self.state_dict()
        w = lambda key: sd[key].reshape(sd[key].shape[0], -1)
        blob = _pack("pccx_pack_prob", _lib.load().pccx_prob_blob_floats(),
                     [w("model_pn.mlp_Modules.0.0.weight"), sd["model_pn.mlp_Modules.0.0.bias"],
                      w("model_pn.mlp_Modules.1.0.weight"), sd["model_pn.mlp_Modules.1.0.bias"],
                      w("model_pn.mlp_Modules.2.0.weight"), sd["model_pn.mlp_Modules.2.0.bias"],
                      w("model_mlp.0.weight"), sd["model_mlp.0.bias"], w("model_mlp.2.weight"), sd["model_mlp.2.bias"],
                      w("model_mlp.4.weight"), sd["model_mlp.4.bias"]], [self.d, self.L])
        self._blob = blob.to(device)
        return self

    def run(self, sampled_xyz, want=("pmf",)):
        """sampled_xyz (B,S,3) -> dict with any of pmf (B,S,d,L), cdf (B,S,d,L+1), cdf_int (int32)."""
        x = _f32c(sampled_xyz, "ConditionalProbabilityModel")
        B, S, _ = x.shape
        if not self.fused_ok(S):
            return self._run_generic(x, want)
        if self._blob is None or self._blob.device != x.device:
            self.pack(x.device)
        r = {}
        if "pmf" in want:
            r["pmf"] = torch.empty(B, S, self.d, self.L, device=x.device, dtype=torch.float32)
        if "cdf" in want:
            r["cdf"] = torch.empty(B, S, self.d, self.L + 1, device=x.device, dtype=torch.float32)
        if "cdf_int" in want:
            r["cdf_int"] = torch.empty(B, S, self.d, self.L + 1, device=x.device, dtype=torch.int32)
        p = lambda k: r[k].data_ptr() if k in r else None
        _lib.call("pccx_prob_forward", x.data_ptr(), B, S, self.d, self.L, self._blob.data_ptr(), p("pmf"), p("cdf"),
                  p("cdf_int"), _stream())
        return r

    def _run_generic(self, x, want):
        """AE.ConditionalProbabilityModel.forward (AE.py:107-123) + pmf_to_cdf + torchac's integer CDF through the generic layers,
        for --d / --L / S outside the fused kernel's shapes: model_pn (generic PointNet), the three 1x1 convolutions as generic
        layers on the (B*S, 259) rows, then pccx_softmax_cdf."""
        from .families import FoldedLinear
        B, S, _ = x.shape
        if getattr(self, "_generic", None) is None or self._generic[0] != x.device:
            mods = list(self.model_mlp)
            layers = [FoldedLinear(m.weight, m.bias, i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU), None, x.device)
                      for i, m in enumerate(mods) if isinstance(m, nn.Conv2d)]
            object.__setattr__(self, "_generic", (x.device, layers))
        feature = self.model_pn(x.transpose(1, 2).contiguous())                                    # AE.py:111-112  (B, 256)
        rows = torch.cat((x, feature[:, None, :].expand(B, S, feature.shape[1])), dim=2).reshape(B * S, -1)   # :113-115
        buf = torch.zeros(B * S, (rows.shape[1] + 3) // 4 * 4, device=x.device, dtype=torch.float32)           # 16-byte rows
        buf[:, :rows.shape[1]] = rows
        rows = buf[:, :rows.shape[1]]
        for layer in self._generic[1]:
            rows = layer(rows)                                                                     # :117  (B*S, d*L)
        r = {}
        if "pmf" in want:
            r["pmf"] = torch.empty(B, S, self.d, self.L, device=x.device, dtype=torch.float32)
        if "cdf" in want:
            r["cdf"] = torch.empty(B, S, self.d, self.L + 1, device=x.device, dtype=torch.float32)
        if "cdf_int" in want:
            r["cdf_int"] = torch.empty(B, S, self.d, self.L + 1, device=x.device, dtype=torch.int32)
        p = lambda k_: r[k_].data_ptr() if k_ in r else None
        _lib.call("pccx_softmax_cdf", rows.contiguous().data_ptr(), B * S * self.d, self.L, p("pmf"), p("cdf"), p("cdf_int"), _stream())   # :119-123
        return r

    def forward(self, sampled_xyz):
        return self.run(sampled_xyz, ("pmf",))["pmf"]


def range_cap(nsym):
    """Default output capacity per cloud of the range coder: 16-bit frequencies with every symbol's frequency >= 1 cost at
    most 16 bits per symbol, plus the flush."""
    return int(nsym) * 2 + 16


def range_encode(cdf_int, latent_q, L, cap=None, out=None, nb=None):
    """torchac.encode_float_cdf on device: cdf_int (B,nsym,L+1) int32, latent_q (B,nsym) -> (bytes (B,cap) u8, nbytes (B)).
    A cloud whose stream does not fit ``cap`` bytes comes back with a NEGATIVE nbytes; codec.Compressed.to_host() raises on it
    (unreachable with the default cap).  out / nb: caller-provided dense destinations."""
    B = cdf_int.shape[0]
    nsym = cdf_int[0].numel() // (L + 1)
    q = _f32c(latent_q.reshape(B, nsym), "range_encode")
    if out is None:
        out = torch.empty(B, cap or range_cap(nsym), device=q.device, dtype=torch.uint8)
    if nb is None:
        nb = torch.empty(B, device=q.device, dtype=torch.int32)
    cap = out.shape[1]
    if (out.shape[0] != B or out.dtype != torch.uint8 or not out.is_contiguous() or tuple(nb.shape) != (B,)
            or nb.dtype != torch.int32 or not nb.is_contiguous()):
        raise _lib.PccxError("range_encode: out must be dense (B,cap) u8 and nb dense (B,) i32")
    _lib.call("pccx_range_encode", cdf_int.contiguous().data_ptr(), q.data_ptr(), B, nsym, int(L), out.data_ptr(), cap,
              nb.data_ptr(), _stream())
    return out, nb


def range_decode(cdf_int, bytes_, nbytes, L):
    """torchac.decode_float_cdf on device -> latent_q (B,nsym) f32 (already minus L//2)."""
    B = cdf_int.shape[0]
    nsym = cdf_int[0].numel() // (L + 1)
    bytes_ = bytes_.contiguous()
    q = torch.empty(B, nsym, device=bytes_.device, dtype=torch.float32)
    _lib.call("pccx_range_decode", cdf_int.contiguous().data_ptr(), bytes_.data_ptr(), bytes_.shape[1],
              nbytes.to(torch.int32).contiguous().data_ptr(), B, nsym, int(L), q.data_ptr(), _stream())
    return q
