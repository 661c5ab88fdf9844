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
    keys = torch.empty(pc.shape[0], device=pc.device, dtype=torch.int64)
    bbox = torch.empty(6, device=pc.device, dtype=torch.int32)
    _lib.call("pccx_morton_keys_auto", pc.data_ptr(), pc.shape[0], keys.data_ptr(), bbox.data_ptr(), _stream())   # bounding box on the device
    return keys


def morton_keys_host_bbox(pc):
    """The same keys with the bounding box reduced by torch and passed from the host (pccx_morton_keys): the definition
    morton_keys() is tested against."""
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
    """Compress this rank's share of the blocks of one large cloud.  Returns (list of (block indices,
    Compressed batch), number of blocks, order, n_valid_last): everything decompress_large needs."""
    from . import dist
    blocks, order, n_last = split_blocks(pc, block)
    mine = dist.shard_indices(blocks.shape[0], rank, world)
    out = []
    for i in range(0, len(mine), batch):
        ids = mine[i:i + batch]
        starts = [dist.fps_start_index(seed, j, block) for j in ids]
        out.append((ids, codec.compress(blocks[ids], starts)))
    return out, blocks.shape[0], order, n_last


def unsplit_blocks(rows, ids, order, n_points, block=8192, out=None):
    """Inverse of split_blocks for the blocks ``ids``: rows (len(ids), block, 3) -> written into ``out`` (n_points,3) at
    the ORIGINAL indices of those blocks' points (out[order[j*block + i]] = rows[j', i]); the padding rows of the last
    block (copies of its final point on the way in) are dropped.  Rows of a decoded block correspond to the block's
    input points as a SET (the decoder regenerates the block's points), so this restores the cloud's size, its
    block-to-region placement and, for rows == the input blocks, the cloud itself bit for bit."""
    if out is None:
        out = torch.empty(n_points, 3, device=rows.device, dtype=rows.dtype)
    ids_t = torch.as_tensor(list(ids), device=rows.device, dtype=torch.int64)
    pos = ids_t[:, None] * block + torch.arange(block, device=rows.device)[None]          # position in Morton order
    keep = pos < n_points                                                                  # drops the padding
    out[order[pos[keep]]] = rows.reshape(-1, block, 3)[keep]
    return out


def decompress_large(codec, parts, n_blocks, order, n_points, block=8192, out=None, S=64):
    """Decode the (block indices, Compressed) batches of compress_large and place them back (unsplit_blocks).  With
    world > 1 every rank fills the rows of ITS blocks in ``out``; rows of other ranks' blocks stay untouched (blocks are
    independent, so metrics are per-block sums gathered with dist.gather_summaries -- the cloud itself is only
    exchanged if a caller wants it in one place)."""
    if out is None:
        dev = parts[0][1].s_bytes.device if parts else "cuda"
        out = torch.zeros(n_points, 3, device=dev, dtype=torch.float32)
    for ids, comp in parts:
        rec = codec.decompress(comp, S=S)
        if rec.shape[1] != block:
            raise ValueError(f"decoded blocks have {rec.shape[1]} points, expected {block} (S*k must equal the block size)")
        unsplit_blocks(rec, ids, order, n_points, block, out)
    return out


def compress_large_many(codec, clouds, seed=11, rank=0, world=1, block=8192, batch=256):
    """Several large clouds at once: every cloud is cut as compress_large cuts it, but the blocks of ALL clouds form one
    sequence (global block g = blocks of cloud 0, then cloud 1, ...) that is sharded g mod world and compressed in batches of
    ``batch`` blocks regardless of cloud boundaries -- full launches instead of one short launch per room.  A block's FPS start
    index is dist.fps_start_index(seed + cloud, block-in-cloud, block), i.e. what compress_large(seed=seed + cloud) would draw, so a
    block's files do not depend on how the clouds were batched.  Returns (parts, metas): parts = list of (global block ids,
    Compressed batch), metas = per cloud (first global block, number of blocks, order, n_points)."""
    from . import dist
    metas, all_blocks, first = [], [], 0
    for pc in clouds:
        blocks, order, _ = split_blocks(pc, block)
        metas.append((first, int(blocks.shape[0]), order, int(pc.shape[0])))
        all_blocks.append(blocks)
        first += int(blocks.shape[0])
    flat = torch.cat(all_blocks) if all_blocks else torch.empty(0, block, 3)
    where = [(ci, j) for ci, (_, nb, _, _) in enumerate(metas) for j in range(nb)]
    mine = dist.shard_indices(first, rank, world)
    parts = []
    for i in range(0, len(mine), batch):
        ids = mine[i:i + batch]
        starts = [dist.fps_start_index(seed + where[g][0], where[g][1], block) for g in ids]
        parts.append((ids, codec.compress(flat[ids], starts)))
    return parts, metas


def decompress_large_many(codec, parts, metas, block=8192, outs=None, S=64):
    """Decode compress_large_many's batches and put every block back into its cloud (unsplit_blocks).  Returns the list of clouds;
    with world > 1 a rank fills the rows of its own blocks only."""
    if outs is None:
        dev = parts[0][1].s_bytes.device if parts else "cuda"
        outs = [torch.zeros(n, 3, device=dev, dtype=torch.float32) for (_, _, _, n) in metas]
    for ids, comp in parts:
        rec = codec.decompress(comp, S=S)
        if rec.shape[1] != block:
            raise ValueError(f"decoded blocks have {rec.shape[1]} points, expected {block} (S*k must equal the block size)")
        for ci, (first, nb, order, n) in enumerate(metas):
            sel = [slot for slot, g in enumerate(ids) if first <= g < first + nb]
            if sel:
                unsplit_blocks(rec[sel], [ids[slot] - first for slot in sel], order, n, block, outs[ci])
    return outs
