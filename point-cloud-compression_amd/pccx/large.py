"""Clouds larger than one codec block (BASELINE configs[3]: S3DIS rooms of 0.5-1 M points).

The reference has no such path: with N points it would need S = N*ALPHA/K patches and
octree_np.decode hard-codes S = 64 (compress.py:102 asserts).  The build keeps S = 64 by cutting a
large cloud into spatially compact blocks of ``block`` = 8192 points (Morton order over the bounding
box) and running the unchanged codec per block; blocks are independent, so they shard across ranks
exactly like files (dist.shard_indices).  The last block is completed by repeating its final point.
The partition is HIP end to end (csrc/geometry.hip, csrc/sort.hip): bounding box + Morton keys, a stable radix sort of the
(key, index) pairs (the permutation torch.sort(keys, stable=True) gives; round 3 used that call), and block gather / scatter
kernels that read and write through the permutation -- no library sort and no torch indexing in the timed step.
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


def morton_order(pc):
    """pc (N,3) -> (N,) int64: the rows of pc in Morton order, ties in input order (= torch.sort(morton_keys(pc), stable=True).indices),
    by the library's radix sort."""
    keys = morton_keys(pc)
    n = keys.shape[0]
    order = torch.empty(n, device=pc.device, dtype=torch.int64)
    ws = torch.empty(_lib.load().pccx_sort_keys_workspace_bytes(n), device=pc.device, dtype=torch.uint8)
    _lib.call("pccx_sort_keys_u64", keys.data_ptr(), n, 63, order.data_ptr(), ws.data_ptr(), _stream())
    return order


_SIDE = {}


def morton_orders(clouds, streams=8):
    """morton_order of several clouds, each on a side stream of its own (up to ``streams``): one room's sort is 16 launches of ~180
    workgroups, less than the chip's 256 CUs, so eight rooms' sorts run side by side instead of one after the other (configs[3]: 3.1 ->
    ~0.8 ms of the step).  Every buffer is allocated on the CURRENT stream before the fork and the current stream waits for all side
    streams before this returns, so callers see ordinary stream semantics."""
    if len(clouds) <= 1 or not clouds[0].is_cuda:
        return [morton_order(pc) for pc in clouds]
    dev = clouds[0].device
    main = torch.cuda.current_stream(dev)
    pool = _SIDE.setdefault(str(dev), [])
    while len(pool) < min(streams, len(clouds)):
        pool.append(torch.cuda.Stream(device=dev))
    lib = _lib.load()
    bufs = []
    for pc in clouds:
        n = int(pc.shape[0])
        bufs.append((torch.empty(n, device=dev, dtype=torch.int64), torch.empty(6, device=dev, dtype=torch.int32),
                     torch.empty(n, device=dev, dtype=torch.int64), torch.empty(lib.pccx_sort_keys_workspace_bytes(n), device=dev, dtype=torch.uint8)))
    used = pool[:min(streams, len(clouds))]
    for st in used:
        st.wait_stream(main)
    for i, (pc, (keys, bbox, order, ws)) in enumerate(zip(clouds, bufs)):
        h = used[i % len(used)].cuda_stream
        _lib.call("pccx_morton_keys_auto", pc.data_ptr(), pc.shape[0], keys.data_ptr(), bbox.data_ptr(), h)
        _lib.call("pccx_sort_keys_u64", keys.data_ptr(), pc.shape[0], 63, order.data_ptr(), ws.data_ptr(), h)
    for st in used:
        main.wait_stream(st)
    return [b[2] for b in bufs]


def gather_blocks(pc, order, block=8192, first=0, stride=1, count=None, out=None):
    """Blocks first, first + stride, ... (count of them; default: every block) of pc in the order `order`, as (count, block, 3); the
    last block of the cloud is completed with copies of its final point.  ``out``: a (count, block, 3) slice to write into."""
    pc = _f32c(pc, "gather_blocks")
    N = pc.shape[0]
    nb = (N + block - 1) // block
    if count is None:
        count = max((nb - first + stride - 1) // stride, 0)
    if out is None:
        out = torch.empty(count, block, 3, device=pc.device, dtype=torch.float32)
    if count:
        _lib.call("pccx_gather_blocks", pc.data_ptr(), order.data_ptr(), N, block, first, stride, count, out.data_ptr(), _stream())
    return out


def split_blocks(pc, block=8192):
    """pc (N,3) -> (blocks (nb,block,3), order (N,) permutation, n_valid_last).  Block j holds points
    order[j*block : (j+1)*block]; the last block is padded with copies of its final point."""
    N = pc.shape[0]
    order = morton_order(pc)
    nb = (N + block - 1) // block
    return gather_blocks(pc, order, block), order, N - (nb - 1) * block


def compress_large(codec, pc, seed=11, rank=0, world=1, block=8192, batch=256):
    """Compress this rank's share of the blocks of one large cloud.  Returns (list of (block indices,
    Compressed batch), number of blocks, order, n_valid_last): everything decompress_large needs."""
    from . import dist
    N = pc.shape[0]
    order = morton_order(pc)
    nb = (N + block - 1) // block
    mine = dist.shard_indices(nb, rank, world)                     # rank, rank + world, ...
    out = []
    for i in range(0, len(mine), batch):
        ids = mine[i:i + batch]
        starts = [dist.fps_start_index(seed, j, block) for j in ids]
        out.append((ids, codec.compress(gather_blocks(pc, order, block, first=ids[0], stride=world, count=len(ids)), starts)))
    return out, nb, order, N - (nb - 1) * block


def _progression(ids):
    """(first, stride) when ids is first, first + stride, ... (what dist.shard_indices hands a rank), else None"""
    ids = list(ids)
    if len(ids) == 1:
        return int(ids[0]), 1
    st = int(ids[1]) - int(ids[0])
    if st >= 1 and all(int(ids[i + 1]) - int(ids[i]) == st for i in range(len(ids) - 1)):
        return int(ids[0]), st
    return None


def unsplit_blocks(rows, ids, order, n_points, block=8192, out=None):
    """Inverse of split_blocks for the blocks ``ids``: rows (len(ids), block, 3) -> written into ``out`` (n_points,3) at
    the ORIGINAL indices of those blocks' points (out[order[j*block + i]] = rows[j', i]); the padding rows of the last
    block (copies of its final point on the way in) are dropped.  Rows of a decoded block correspond to the block's
    input points as a SET (the decoder regenerates the block's points), so this restores the cloud's size, its
    block-to-region placement and, for rows == the input blocks, the cloud itself bit for bit."""
    if out is None:
        out = torch.empty(n_points, 3, device=rows.device, dtype=rows.dtype)
    ids = list(ids)
    if not ids:
        return out
    rows = _f32c(rows.reshape(len(ids), block, 3), "unsplit_blocks")
    runs, prog = [], _progression(ids)
    if prog is not None:
        runs = [(0, len(ids), prog[0], prog[1])]
    else:                                                 # arbitrary ids: one launch per block
        runs = [(s_, 1, int(j), 1) for s_, j in enumerate(ids)]
    for s0, cnt, first, stride in runs:
        _lib.call("pccx_scatter_blocks", rows[s0:s0 + cnt].data_ptr(), order.data_ptr(), n_points, block, first, stride, cnt, out.data_ptr(), _stream())
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


def _segments(ids, metas):
    """Cut a batch's ascending global block ids into runs that belong to one cloud: (slot of the run's first block, count, cloud,
    block-in-cloud of the first)"""
    segs, slot = [], 0
    while slot < len(ids):
        g = ids[slot]
        ci = next(c for c, (first, nb, _, _) in enumerate(metas) if first <= g < first + nb)
        first, nb = metas[ci][0], metas[ci][1]
        cnt = 1
        while slot + cnt < len(ids) and ids[slot + cnt] < first + nb:
            cnt += 1
        segs.append((slot, cnt, ci, g - first))
        slot += cnt
    return segs


def compress_large_many(codec, clouds, seed=11, rank=0, world=1, block=8192, batch=256):
    """Several large clouds at once: every cloud is cut as compress_large cuts it, but the blocks of ALL clouds form one
    sequence (global block g = blocks of cloud 0, then cloud 1, ...) that is sharded g mod world and compressed in batches of
    ``batch`` blocks regardless of cloud boundaries -- full launches instead of one short launch per room.  A block's FPS start
    index is dist.fps_start_index(seed + cloud, block-in-cloud, block), i.e. what compress_large(seed=seed + cloud) would draw, so a
    block's files do not depend on how the clouds were batched.  Returns (parts, metas): parts = list of (global block ids,
    Compressed batch), metas = per cloud (first global block, number of blocks, order, n_points).  The clouds must stay alive and
    unchanged until the call returns; each batch's blocks are gathered straight from them through the permutation."""
    from . import dist
    metas, first = [], 0
    clouds = [_f32c(pc, "compress_large_many") for pc in clouds]
    for pc, order in zip(clouds, morton_orders(clouds)):
        nb = (int(pc.shape[0]) + block - 1) // block
        metas.append((first, nb, order, int(pc.shape[0])))
        first += nb
    mine = dist.shard_indices(first, rank, world)
    parts = []
    for i in range(0, len(mine), batch):
        ids = mine[i:i + batch]
        buf = torch.empty(len(ids), block, 3, device=clouds[0].device, dtype=torch.float32)
        starts = []
        for slot, cnt, ci, j0 in _segments(ids, metas):
            gather_blocks(clouds[ci], metas[ci][2], block, first=j0, stride=world, count=cnt, out=buf[slot:slot + cnt])
            starts += [dist.fps_start_index(seed + ci, j0 + q * world, block) for q in range(cnt)]
        parts.append((ids, codec.compress(buf, starts)))
    return parts, metas


def decompress_large_many(codec, parts, metas, block=8192, outs=None, S=64):
    """Decode compress_large_many's batches and put every block back into its cloud (unsplit_blocks).  Returns the list of clouds;
    with world > 1 a rank fills the rows of its own blocks only (the other rows are zero)."""
    if outs is None:
        dev = parts[0][1].s_bytes.device if parts else "cuda"
        have = [0] * len(metas)
        for ids, _ in parts:
            for _, cnt, ci, _ in _segments(ids, metas):
                have[ci] += cnt
        # a cloud all of whose blocks are decoded here is written row by row: no clear needed
        outs = [(torch.empty if have[ci] == nb else torch.zeros)(n, 3, device=dev, dtype=torch.float32) for ci, (_, nb, _, n) in enumerate(metas)]
    for ids, comp in parts:
        rec = codec.decompress(comp, S=S)
        if rec.shape[1] != block:
            raise ValueError(f"decoded blocks have {rec.shape[1]} points, expected {block} (S*k must equal the block size)")
        stride = ids[1] - ids[0] if len(ids) > 1 else 1
        for slot, cnt, ci, j0 in _segments(ids, metas):
            _, _, order, n = metas[ci]
            unsplit_blocks(rec[slot:slot + cnt], [j0 + q * stride for q in range(cnt)], order, n, block, outs[ci])
    return outs
