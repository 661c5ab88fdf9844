"""Clouds larger than one codec block (BASELINE configs[3]: S3DIS rooms of 0.5-1 M points).

The reference has no such path: with N points it would need S = N*ALPHA/K patches and
octree_np.decode hard-codes S = 64 (compress.py:102 asserts).  The build keeps S = 64 by cutting a
large cloud into spatially compact blocks of ``block`` = 8192 points (Morton order over the bounding
box) and running the unchanged codec per block; blocks are independent, so they shard across ranks
exactly like files (dist.shard_indices).  The last block is completed by repeating its final point.
Ordering the keys uses torch.sort: data preparation outside the codec's timed window.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from .ops import _f32c, _stream


def morton_keys(pc):
    """pc (N,3) f32 on the GPU -> (N,) int64 keys over pc's bounding box."""
    pc = _f32c(pc, "morton_keys")
    lo = pc.amin(dim=0).cpu().numpy().astype(np.float32)
    ext = float((pc.amax(dim=0).cpu().numpy() - lo).max())
    keys = torch.empty(pc.shape[0], device=pc.device, dtype=torch.int64)
    lo_c = (ctypes.c_float * 3)(*lo.tolist())
    _lib.call("pccx_morton_keys", pc.data_ptr(), pc.shape[0], ctypes.addressof(lo_c), max(ext, 1e-30), keys.data_ptr(), _stream())
    return keys


def split_blocks(pc, block=8192):
    """pc (N,3) -> (blocks (nb,block,3), order (N,) permutation, n_valid_last).  Block j holds points
    order[j*block : (j+1)*block]; the last block is padded with copies of its final point."""
    N = pc.shape[0]
    order = torch.sort(morton_keys(pc), stable=True).indices
    nb = (N + block - 1) // block
    pad = nb * block - N
    idx = torch.cat([order, order[-1:].expand(pad)]) if pad else order
    return pc[idx].view(nb, block, 3).contiguous(), order, block - pad


def compress_large(codec, pc, seed=11, rank=0, world=1, block=8192, batch=256):
    """Compress this rank's share of the blocks of one large cloud.  Returns (list of (block index,
    Compressed batch slices), number of blocks)."""
    from . import dist
    blocks, order, n_last = split_blocks(pc, block)
    mine = dist.shard_indices(blocks.shape[0], rank, world)
    out = []
    for i in range(0, len(mine), batch):
        ids = mine[i:i + batch]
        starts = [dist.fps_start_index(seed, j, block) for j in ids]
        out.append((ids, codec.compress(blocks[ids], starts)))
    return out, blocks.shape[0], order, n_last
