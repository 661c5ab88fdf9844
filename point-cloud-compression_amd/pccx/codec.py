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
import os

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
            self._host_packed = packed if not packed.is_cuda else packed.cpu()
            h = Compressed.from_packed(self._host_packed, B, self.s_bytes.shape[1], self.p_bytes.shape[1], self.n_points)
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

    def write_files(self, directory, names, threads=0):
        """<directory>/<names[b]>.p.bin / .s.bin / .c.bin for every cloud (compress.py:139-152), cut from ONE host copy of the packed
        buffer by the library's host threads (pccx_write_streams_host).  Returns the total number of bytes of the three files."""
        B = self.s_bytes.shape[0]
        if len(names) != B:
            raise ValueError(f"write_files: {len(names)} names for {B} clouds")
        self.to_host()                                         # ONE D2H (cached); raises on a negative byte count
        host = self._host_packed
        write_streams(host, B, self.s_bytes.shape[1], self.p_bytes.shape[1], directory, names, threads)
        return int(self._host[1].sum()) + int(self._host[3].sum()) + 16 * B

    @classmethod
    def read_files(cls, directory, names, n_points=0, s_stride=None, p_cap=None, device=None, threads=0, out=None):
        """The inverse (decompress.py:80-91,113): the three files of every name into one packed host buffer (rows sized from the largest
        file unless s_stride / p_cap are given; `out` = a host uint8 tensor to fill, e.g. pinned), then -- with `device` -- ONE upload."""
        B = len(names)
        if s_stride is None or p_cap is None:
            ss, ps = stream_sizes(directory, names, threads)
            if B and (ss.min() < 0 or ps.min() < 0):
                from ._lib import PccxError
                raise PccxError("missing .s.bin / .p.bin for: " + ", ".join(n for n, a, b_ in zip(names, ss, ps) if a < 0 or b_ < 0))
            s_stride = int(max(1, ss.max() if B else 1)) if s_stride is None else s_stride
            p_cap = int(max(1, ps.max() if B else 1)) if p_cap is None else p_cap
        need = packed_layout(B, s_stride, p_cap)[-1]
        host = out if out is not None else torch.empty(need, dtype=torch.uint8)
        if host.numel() != need or host.dtype != torch.uint8 or host.is_cuda:
            raise ValueError(f"read_files: `out` must be a host uint8 tensor of {need} bytes")
        read_streams(host, B, s_stride, p_cap, directory, names, threads)
        packed = host.to(device, non_blocking=True) if device is not None else host
        return cls.from_packed(packed, B, s_stride, p_cap, n_points)


def _name_table(names):
    blob, off, o = bytearray(), np.zeros(max(len(names), 1), dtype=np.int64), 0
    for i, n in enumerate(names):
        e = os.fsencode(n)
        if b"\0" in e or b"/" in e:
            raise ValueError(f"stream name {n!r}: a file name without directory part is expected")
        off[i] = o
        blob += e + b"\0"
        o += len(e) + 1
    return bytes(blob) or b"\0", off


def _host_u8(t, need, who):
    if not isinstance(t, torch.Tensor) or t.is_cuda or t.dtype != torch.uint8 or not t.is_contiguous() or t.numel() != need:
        raise ValueError(f"{who}: a contiguous host uint8 tensor of {need} bytes is expected (codec.packed_layout)")
    return t


def write_streams(packed_host, B, s_stride, p_cap, directory, names, threads=0):
    """pccx_write_streams_host: the files of B clouds from the host copy of a packed buffer (no GPU call)."""
    from . import _lib
    blob, off = _name_table(names)
    _host_u8(packed_host, packed_layout(B, s_stride, p_cap)[-1], "write_streams")
    _lib.call("pccx_write_streams_host", packed_host.data_ptr(), B, s_stride, p_cap, os.fsencode(directory), blob, off.ctypes.data, int(threads))


def read_streams(packed_host, B, s_stride, p_cap, directory, names, threads=0):
    """pccx_read_streams_host: the files of B clouds into a packed host buffer (counts, centres, rows; row tails cleared)."""
    from . import _lib
    blob, off = _name_table(names)
    _host_u8(packed_host, packed_layout(B, s_stride, p_cap)[-1], "read_streams")
    _lib.call("pccx_read_streams_host", packed_host.data_ptr(), B, s_stride, p_cap, os.fsencode(directory), blob, off.ctypes.data, int(threads))


def stream_sizes(directory, names, threads=0):
    """(sizes of <name>.s.bin, sizes of <name>.p.bin) as int64 arrays, -1 where the file does not exist."""
    from . import _lib
    blob, off = _name_table(names)
    ss, ps = np.zeros(max(len(names), 1), np.int64), np.zeros(max(len(names), 1), np.int64)
    _lib.call("pccx_stream_sizes_host", len(names), os.fsencode(directory), blob, off.ctypes.data, ss.ctypes.data, ps.ctypes.data, int(threads))
    return ss[:len(names)], ps[:len(names)]


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
            nn = ops.knn_points(rec, pcn, self.K, patch_scale=scale,                 # compress.py:105-108 (KNN_Patching keeps the
                                return_dists=False, return_idx=keep_extras)          # patches; the indices only for diagnostics)
        patches = nn.knn.view(B * S, self.K, 3)
        raw, latent, q = self.ae.encode(patches, sa_matmul=self.sa_matmul, pn_matmul=self.pn_matmul)                                     # compress.py:113-127
        with stage("prob"):
            cdf_int = self.prob.run(rec, ("cdf_int",))["cdf_int"]                    # compress.py:131-134
        with stage("range_encode"):
            models.range_encode(cdf_int, q.view(B, S * d), L, out=comp.p_bytes, nb=comp.p_nbytes)   # compress.py:135-136
        comp.c[:, :3].copy_(center)                                                  # compress.py:149-152
        comp.c[:, 3].copy_(longest)
        comp._cdf_int = cdf_int               # kept for decompress(reuse_cdf=True): the resident pipeline's shortcut, never part of the streams
        if keep_extras:
            comp.extras = dict(pcn=pcn, fps_idx=fps_idx, sampled=sampled, octree=oc, rec_sampled=rec, patches=patches,
                          latent_raw=raw, latent=latent, latent_q=q, cdf_int=cdf_int, knn_idx=nn.idx)
        return comp

    def decompress(self, comp, S=64, reuse_cdf=False):
        """Inverse pipeline (decompress.py:80-116) -> (B, S*k, 3) f32.
        reuse_cdf=True (resident pipelines only): when `comp` is the very object compress() returned, the integer CDF it coded with is
        still in HBM and the probability model is not evaluated a second time -- both sides evaluate the SAME function of the SAME
        decoded centres (compress.py:131, decompress.py:88), so the table is identical by construction.  From bytes (files, a
        Compressed built by from_packed / read_files) the table is always recomputed, as decompress.py does."""
        B = comp.s_bytes.shape[0]
        d, L = self.ae.d, self.ae.L
        with stage("octree_decode"):
            rec, cnt = ops.octree_decode(comp.s_bytes, comp.s_nbytes, self.octree_mode, S)    # decompress.py:80-85
        cdf_int = getattr(comp, "_cdf_int", None) if reuse_cdf else None
        if cdf_int is None or cdf_int.shape[0] != B or cdf_int.shape[1] != S:
            with stage("prob"):
                cdf_int = self.prob.run(rec, ("cdf_int",))["cdf_int"]                     # decompress.py:88-92
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
