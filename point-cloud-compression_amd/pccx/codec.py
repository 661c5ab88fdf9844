"""Batched compress / decompress pipeline: the loop bodies of compress.py:90-152 and
decompress.py:80-116 for B clouds per launch sequence, everything resident in HBM.

The reference processes one cloud at a time (B = 1, compress.py:48) with 128 per-patch module
calls and several host round trips per cloud; here each stage is one kernel launch over the whole
batch and nothing returns to the host until the caller asks for the byte streams.

Stream formats are the reference's: ``.s.bin`` = packed octree bits (tail byte
right-aligned), ``.p.bin`` = range-coder bytes of the S*d latent symbols, ``.c.bin`` = 4 x fp32
[cx, cy, cz, longest].
"""
import dataclasses

import numpy as np
import torch

from . import models, ops
from .ops import stage


def packed_layout(B, s_stride, p_cap):
    """Byte offsets of the five sections of one batch's packed stream buffer (SoA, so that every section is a dense
    tensor the kernels write in place): s_nbytes (B) i32 | p_nbytes (B) i32 | c (B,4) f32 | s_bytes (B,s_stride) u8 |
    p_bytes (B,p_cap) u8.  Everything the three files of compress.py:139-152 need, in ONE device->host copy."""
    o_sn, o_pn, o_c = 0, 4 * B, 8 * B
    o_sb = o_c + 16 * B
    o_pb = o_sb + B * s_stride
    return o_sn, o_pn, o_c, o_sb, o_pb, o_pb + B * p_cap


@dataclasses.dataclass
class Compressed:
    s_bytes: torch.Tensor     # (B, stride) u8  packed octree streams
    s_nbytes: torch.Tensor    # (B) i32
    p_bytes: torch.Tensor     # (B, cap) u8     range-coded latents
    p_nbytes: torch.Tensor    # (B) i32
    c: torch.Tensor           # (B, 4) f32      centre xyz + longest side
    n_points: int
    extras: dict = None       # intermediates for tests / diagnostics
    packed: torch.Tensor = None   # the u8 buffer the five fields above are views of (packed_layout), when built by Codec

    @classmethod
    def alloc(cls, B, s_stride, p_cap, n_points, device):
        return cls.from_packed(torch.empty(packed_layout(B, s_stride, p_cap)[-1], device=device, dtype=torch.uint8),
                               B, s_stride, p_cap, n_points)

    @classmethod
    def from_packed(cls, packed, B, s_stride, p_cap, n_points):
        """Views over a packed buffer (device: as produced by Codec.compress; or the same bytes uploaded from the host)."""
        o_sn, o_pn, o_c, o_sb, o_pb, end = packed_layout(B, s_stride, p_cap)
        if packed.numel() != end or packed.dtype != torch.uint8:
            raise ValueError(f"packed stream buffer: expected {end} bytes of uint8, got {packed.numel()} of {packed.dtype}")
        return cls(packed[o_sb:o_pb].view(B, s_stride), packed[o_sn:o_pn].view(torch.int32),
                   packed[o_pb:end].view(B, p_cap), packed[o_pn:o_c].view(torch.int32),
                   packed[o_c:o_sb].view(torch.float32).view(B, 4), n_points, None, packed)

    def bits(self):
        """Total bits per cloud of the three files (eval.py:189 numerator)."""
        return 8 * (self.s_nbytes.long() + self.p_nbytes.long() + 16)

    def bpp(self):
        return self.bits().double() / self.n_points

    def to_host(self):
        """ONE device->host transfer of everything the three files need (cached).  Raises PccxError when a range-coder
        output buffer was too small (the kernel reports that as a negative byte count)."""
        if getattr(self, "_host", None) is None:
            B = self.s_bytes.shape[0]
            if self.packed is None:       # assembled by hand (e.g. from files): gather the five pieces first
                comp = Compressed.alloc(B, self.s_bytes.shape[1], self.p_bytes.shape[1], self.n_points, self.s_bytes.device)
                for dst, src in ((comp.s_bytes, self.s_bytes), (comp.s_nbytes, self.s_nbytes), (comp.p_bytes, self.p_bytes),
                                 (comp.p_nbytes, self.p_nbytes), (comp.c, self.c)):
                    dst.copy_(src)
                packed = comp.packed
            else:
                packed = self.packed
            h = Compressed.from_packed(packed.cpu(), B, self.s_bytes.shape[1], self.p_bytes.shape[1], self.n_points)
            sn, pn = h.s_nbytes.numpy(), h.p_nbytes.numpy()
            if (pn < 0).any() or (sn < 0).any():
                from ._lib import PccxError
                raise PccxError("range coder / octree output buffer too small for clouds %s (negative byte count): "
                                "raise `cap`" % np.nonzero((pn < 0) | (sn < 0))[0].tolist())
            self._host = (h.s_bytes.numpy(), sn, h.p_bytes.numpy(), pn, h.c.numpy())
        return self._host

    def files(self, b):
        """The three byte strings compress.py:139-152 writes for cloud b."""
        sb, sn, pb, pn, c = self.to_host()
        return bytes(sb[b, :sn[b]]), bytes(pb[b, :pn[b]]), c[b].tobytes()


class Codec:
    def __init__(self, ae, prob, K=256, ALPHA=2, N0=1024, octree_mode="reference", margin=0.01, matmul=None,
                 decoder_matmul=None, sa_matmul=None, pn_matmul=None):
        """matmul: how the three transforms (SetAbstraction, PointNet, decoder) form their fp32 products --
        "f32" = v_mfma_f32_16x16x4_f32 (bit-for-bit a k-ordered fmaf chain), "bf16x3" = each fp32 operand split exactly into
        three bf16 pieces, six products per pair on the bf16 matrix cores, fp32 accumulate (fp32-level error, 2.6x the
        rate).  None = pccx.DEFAULT_MATMUL.  The per-stage arguments override it."""
        from . import DEFAULT_MATMUL
        matmul = matmul or DEFAULT_MATMUL
        self.ae, self.prob = ae, prob
        self.decoder_matmul = decoder_matmul or matmul
        self.sa_matmul, self.pn_matmul = sa_matmul or matmul, pn_matmul or matmul
        for m in (self.decoder_matmul, self.sa_matmul, self.pn_matmul):
            if m not in ("f32", "bf16x3", "f16x2"):
                raise ValueError(f"matmul={m!r}: expected 'f32', 'bf16x3' or 'f16x2'")
        self.K, self.ALPHA, self.N0 = K, ALPHA, N0
        self.k = K // ALPHA                                  # compress.py:46
        self.octree_mode = octree_mode
        self.margin = margin
        if ae.K != K or ae.k != self.k:
            raise ValueError("AE was built for a different K / k")

    def compress(self, pc, start_idx, keep_extras=False):
        """pc (B,N,3) f32 on the GPU; start_idx (B,) FPS start per cloud (the reference draws it
        from torch.randint, pn_kit.py:321)."""
        B, N, _ = pc.shape
        d, L = self.ae.d, self.ae.L
        with stage("normalize"):
            pcn, center, longest = ops.normalize(pc, self.margin)                    # compress.py:90
        S = int(N * self.ALPHA // self.K)                                            # compress.py:93
        if self.octree_mode == "reference" and S != 64:
            raise ValueError(f"octree_mode='reference' reproduces octree_np.decode's hard-coded S=64 "
                             f"(octree_np.py:100; compress.py:102 asserts); got S={S}. Use octree_mode='full'.")
        with stage("fps"):
            fps_idx = ops.farthest_point_sample_batch(pcn, S, start_idx)             # compress.py:96
        with stage("gather"):
            sampled = ops.index_points(pcn, fps_idx)
        comp = Compressed.alloc(B, (ops.octree_bits_capacity(S) + 7) // 8, models.range_cap(S * d), N, pc.device)
        with stage("octree_encode"):
            oc = ops.octree_encode(sampled, N, ops.OCTREE_BPP_DICT[self.K],          # compress.py:98
                                   out_bytes=comp.s_bytes, out_nbytes=comp.s_nbytes)
        with stage("octree_decode"):
            rec, _ = ops.octree_decode(oc["bytes"], oc["nbytes"], self.octree_mode, S)   # compress.py:100
        scale = float((N / self.N0) ** (1 / 3))
        with stage("knn_patches"):
            nn = ops.knn_points(rec, pcn, self.K, patch_scale=scale)                 # compress.py:105-108
        patches = nn.knn.view(B * S, self.K, 3)
        raw, latent, q = self.ae.encode(patches, sa_matmul=self.sa_matmul, pn_matmul=self.pn_matmul)                                     # compress.py:113-127
        with stage("prob"):
            cdf_int = self.prob.run(rec, ("cdf_int",))["cdf_int"]                    # compress.py:131-134
        with stage("range_encode"):
            models.range_encode(cdf_int, q.view(B, S * d), L, out=comp.p_bytes, nb=comp.p_nbytes)   # compress.py:135-136
        comp.c[:, :3].copy_(center)                                                  # compress.py:149-152
        comp.c[:, 3].copy_(longest)
        if keep_extras:
            comp.extras = dict(pcn=pcn, fps_idx=fps_idx, sampled=sampled, octree=oc, rec_sampled=rec, patches=patches,
                          latent_raw=raw, latent=latent, latent_q=q, cdf_int=cdf_int, knn_idx=nn.idx)
        return comp

    def decompress(self, comp, S=64):
        """Inverse pipeline (decompress.py:80-116) -> (B, S*k, 3) f32."""
        B = comp.s_bytes.shape[0]
        d, L = self.ae.d, self.ae.L
        with stage("octree_decode"):
            rec, cnt = ops.octree_decode(comp.s_bytes, comp.s_nbytes, self.octree_mode, S)    # decompress.py:80-85
        with stage("prob"):
            cdf_int = self.prob.run(rec, ("cdf_int",))["cdf_int"]                         # decompress.py:88-92
        with stage("range_decode"):
            q = models.range_decode(cdf_int, comp.p_bytes, comp.p_nbytes, L)              # decompress.py:93
        N = S * self.k                                                                    # decompress.py:106
        scale = float((N / self.N0) ** (1 / 3))
        with stage("ae_decode"):
            return self.ae.decode(q.view(B * S, d), rec.view(B * S, 3), comp.c[:, :3].contiguous(),
                                  comp.c[:, 3].contiguous(), S=S, scale=scale, margin=self.margin,
                                  matmul=self.decoder_matmul)


def d1_psnr(orig, recon):
    """eval.py:43-98 D1 (point-to-point) PSNR, batched: 10*log10(diag^2 / mean_recon min_orig |.|^2),
    diag = bounding-box diagonal of the original.  (B,N,3),(B,M,3) -> (B,) f64."""
    d2 = ops.nn_dist(recon, orig).double()
    mse = d2.mean(dim=1)
    rng = orig.amax(dim=1).double() - orig.amin(dim=1).double()
    diag2 = (rng * rng).sum(dim=1)
    return 10 * torch.log10(diag2 / mse)


def d2_psnr(orig, recon, knn=30):
    """eval.py:43-98 D2 (point-to-plane) PSNR: normals of the ORIGINAL by 30-NN PCA, error = squared
    projection of (recon - nearest original) on that normal.  (B,) f64."""
    normals = ops.estimate_normals(orig, knn)
    mse = ops.point_plane_err(recon, orig, normals).double().mean(dim=1)
    rng = orig.amax(dim=1).double() - orig.amin(dim=1).double()
    return 10 * torch.log10((rng * rng).sum(dim=1) / mse)


def normalized_chamfer(orig, recon):
    """eval.py:198-205: both clouds min-max normalised by the ORIGINAL's global min/max, then
    pytorch3d chamfer_distance.  Returns (B,) f64."""
    lo = orig.amin(dim=(1, 2), keepdim=True)
    hi = orig.amax(dim=(1, 2), keepdim=True)
    a = ((orig - lo) / (hi - lo)).contiguous()
    b = ((recon - lo) / (hi - lo)).contiguous()
    return ops.nn_dist(b, a).double().mean(dim=1) + ops.nn_dist(a, b).double().mean(dim=1)
