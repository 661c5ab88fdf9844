"""Drop-in for the reference's ``pn_kit`` module: same names and argument meaning
(pn_kit.py), backed by libpccx.so.  Put this directory ahead of the reference on sys.path.
Tensors must live on the GPU (there is no CPU compute path); numpy inputs of the octree
helpers are uploaded, as the reference's callers pass numpy there (compress.py:98-100)."""
import os
import sys

import numpy as np
import torch

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

from pccx import ops, plyio  # noqa: E402
from pccx.models import MLP, PointNet, SetAbstraction  # noqa: E402,F401  (parameter containers)

OCTREE_BPP_DICT = dict(ops.OCTREE_BPP_DICT)                       # pn_kit.py:17-23
OCTREE_MODE = os.environ.get("PCCX_OCTREE_MODE", "reference")     # 'reference' = octree_np.decode as written


def read_point_cloud(filepath):                                   # pn_kit.py:25-31
    return plyio.read_point_cloud(filepath)


def read_point_clouds(file_path_list):                             # pn_kit.py:33-37 (host I/O, serial here)
    return np.array([plyio.read_point_cloud(f) for f in file_path_list])


def save_point_cloud(pc, filename, path='./viewing/'):            # pn_kit.py:39-42
    plyio.save_point_cloud(pc, os.path.join(path, filename))


def _up(t):
    """The reference's scripts shuttle tensors between host and device freely (decompress.py:110-116 denormalises on
    the CPU).  There is no CPU compute path here: a host tensor is uploaded, the HIP kernel runs, and the result goes
    back to the device the caller used."""
    t = torch.as_tensor(t)
    return (t if t.is_cuda else t.cuda()), t.device


def normalize(pc, margin=0.01):                                   # pn_kit.py:47-60, pc (1,N,3)
    x, dev = _up(pc)
    out, center, longest = (t.to(dev) for t in ops.normalize(x.float(), margin))
    if pc.shape[0] == 1:
        return out, center[0], longest[0]
    return out, center, longest


def denormalize(pc, cetner, longest, margin=0.01):                # pn_kit.py:62-66
    x, dev = _up(pc)
    B = x.shape[0]
    c = torch.as_tensor(cetner, dtype=torch.float32).to(x.device).reshape(-1, 3).expand(B, 3).contiguous()
    l = torch.as_tensor(longest, dtype=torch.float32).to(x.device).reshape(-1).expand(B).contiguous()
    return ops.denormalize(x.float(), c, l, margin).to(dev)


def farthest_point_sample_batch(xyz, npoint, start_idx=None):     # pn_kit.py:309-330
    x, dev = _up(xyz)
    return ops.farthest_point_sample_batch(x, npoint, start_idx).to(dev)


def index_points(points, idx):                                    # pn_kit.py:332-360
    x, dev = _up(points)
    return ops.index_points(x, torch.as_tensor(idx).to(x.device)).to(dev)


def encode_sampled_np(sampled_xyz, scale, N, min_bpp):            # pn_kit.py:380-401
    if scale != 1:
        raise ValueError("pccx implements the octree for scale=1 (the only value on the path, compress.py:98)")
    x = torch.as_tensor(np.asarray(sampled_xyz), dtype=torch.float32).cuda()
    r = ops.octree_encode(x, N, min_bpp)
    nb = r["nbits"].cpu().numpy()
    bits = r["bits"].cpu().numpy()
    codes = [bits[b, :nb[b]].copy() for b in range(x.shape[0])]
    return codes, int(nb.sum())


def decode_sampled_np(codes, scale, mode=None):                   # pn_kit.py:424-431
    mode = mode or OCTREE_MODE
    rows = [bytes(binary_array_to_byte_array(c)) if len(c) >= 8 or mode == "full" else None for c in codes]
    out = []
    for c, row in zip(codes, rows):
        if row is None:
            # sub-byte stream handed over as raw bits (compress side): octree_np.decode reads the bits as given
            g = list(int(v) for v in c)
            pts = [[0.75 if (7 - t) & 4 else 0.25, 0.75 if (7 - t) & 2 else 0.25, 0.75 if (7 - t) & 1 else 0.25]
                   for t, bit in enumerate(g) if bit == 1]
            if len(g) == 0:
                pts = [[0.5, 0.5, 0.5]]
            arr = np.zeros((64, 3), dtype=np.float32) if not pts else np.array(pts + [pts[-1]] * (64 - len(pts)), dtype=np.float32)
            out.append(arr)
            continue
        by = torch.from_numpy(np.frombuffer(row, dtype=np.uint8).copy())[None].cuda()
        nby = torch.tensor([len(row)], dtype=torch.int32).cuda()
        if mode == "reference":
            pts, _ = ops.octree_decode(by, nby, "reference", 64)
        else:
            _, cnt = ops.octree_decode(by, nby, "full", 1)
            pts, _ = ops.octree_decode(by, nby, "full", max(int(cnt[0]), 1))
        out.append(pts[0].cpu().numpy())
    return np.stack(out, axis=0)


# -- helpers of the training scripts (train.py:164-201), not on the compress/decompress path: tensor plumbing only
def n_scale_batch(batch_pc, margin=0.01):                         # pn_kit.py:68-87
    ext = batch_pc.max(dim=1)[0] - batch_pc.min(dim=1)[0]
    scaling = (1 - margin) / ext.max(dim=1)[0]
    return batch_pc * scaling.view(-1, 1, 1), scaling


def d_n_scale_batch(batch_pc, scaling):                           # pn_kit.py:89-96
    return batch_pc / scaling.view(-1, 1, 1)


def estimate_bits_from_pmf(pmf, sym):                             # pn_kit.py:439-450
    L = pmf.shape[-1]
    p = torch.gather(pmf.reshape(-1, L), 1, sym.reshape(-1, 1))
    return torch.sum(-torch.log2(p.clamp(min=1e-3)))


def pmf_to_cdf(pmf):                                              # pn_kit.py:452-461
    cdf = pmf.cumsum(dim=-1)
    zeros = torch.zeros(pmf.shape[:-1] + (1,), dtype=pmf.dtype, device=pmf.device)
    return torch.cat([zeros, cdf], dim=-1).clamp(max=1.)


def binary_array_to_byte_array(a):                                # pn_kit.py:463-467 (tail byte right-aligned)
    a = np.asarray(a, dtype=np.uint8).reshape(-1)
    n = a.shape[0]
    full = n // 8 * 8
    out = bytearray(np.packbits(a[:full]).tobytes())
    if n > full:
        v = 0
        for bit in a[full:]:
            v = (v << 1) | int(bit)
        out.append(v)
    return out


def byte_array_to_binary_array(byte_stream):                      # pn_kit.py:469-475
    return np.unpackbits(np.frombuffer(bytes(byte_stream), dtype=np.uint8)).astype(np.int32)
