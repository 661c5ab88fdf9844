"""File sharding across the GPUs of one node: one process per GPU, cloud i -> rank
i mod world, no data-path collective.  The only exchanges are a MAX of wall time and an all-gather of
tiny per-rank summaries (sum bits, sum points, sum PSNR, sum Chamfer, files, seconds) -- tens of
bytes over RCCL/xGMI, latency-bound."""
import numpy as np
import torch
import torch.distributed as dist

SUMMARY_FIELDS = ("bits", "points", "psnr_sum", "chamfer_sum", "files", "seconds")


def shard_indices(n_files, rank, world):
    """Indices of the files rank ``rank`` owns."""
    return list(range(rank, n_files, world))


def fps_start_index(seed, file_index, n_points):
    """Deterministic replacement for the reference's per-file torch.randint draw (pn_kit.py:321):
    a pure function of (seed, file index), so results do not depend on the sharding."""
    return int(np.random.default_rng([seed, file_index]).integers(0, n_points))


def gather_summaries(local, device=None):
    """all_gather of a per-rank fp64 summary vector -> (world, len) tensor on every rank."""
    t = torch.as_tensor(local, dtype=torch.float64, device=device)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return t[None]
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return torch.stack(out)


def reduce_summaries(gathered):
    """Global metrics from the gathered per-rank sums."""
    g = gathered.sum(dim=0).cpu().numpy()
    s = dict(zip(SUMMARY_FIELDS, g))
    files = max(s["files"], 1.0)
    return {"bpp": s["bits"] / max(s["points"], 1.0), "d1_psnr_db": s["psnr_sum"] / files,
            "chamfer": s["chamfer_sum"] / files, "files": int(s["files"]),
            "points_per_s": s["points"] / max(gathered[:, 5].max().item(), 1e-12)}


def max_over_ranks(seconds, device=None):
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])


def allreduce_mean_(tensors, bucket_bytes=64 << 20):
    """Data-parallel gradient averaging (the one collective of the configs[4] training step): flatten the
    tensors into buckets of ``bucket_bytes`` (xGMI is point-to-point, ring all-reduce is per-link bound, so
    a few large messages beat many small ones; the pppe model is 116 MB of fp32 gradients = 2 buckets),
    all-reduce each bucket over the default process group (RCCL on GPUs, gloo in the CPU rehearsal) and
    scatter the averages back in place.  No-op without a process group."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return 0
    world = dist.get_world_size()
    n_buckets, cur, size = 0, [], 0

    def flush():
        nonlocal cur, size, n_buckets
        if not cur:
            return
        flat = torch.cat([t.reshape(-1) for t in cur])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat /= world
        off = 0
        for t in cur:
            t.copy_(flat[off:off + t.numel()].view_as(t))
            off += t.numel()
        cur, size = [], 0
        n_buckets += 1

    for t in tensors:
        if t is None:
            continue
        cur.append(t)
        size += t.numel() * t.element_size()
        if size >= bucket_bytes:
            flush()
    flush()
    return n_buckets
