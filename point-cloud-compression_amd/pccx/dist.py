"""File sharding across the GPUs of one node: one process per GPU, cloud i -> rank
i mod world, no data-path collective.  The only exchanges are a MAX of wall time and an all-gather of
tiny per-rank summaries (sum bits, sum points, sum PSNR, sum Chamfer, files, seconds) -- tens of
bytes over RCCL/xGMI, latency-bound."""
import os

import numpy as np
import torch
import torch.distributed as dist

SUMMARY_FIELDS = ("bits", "points", "psnr_sum", "chamfer_sum", "files", "seconds")


def collectives_active():
    """True when the exchanges below really go through the process group: more than one rank -- or ONE rank with
    PCCX_DIST_SINGLE_RANK=1, the rehearsal of the RCCL path a one-GPU box allows (RCCL refuses two ranks on one device, but a one-rank
    communicator runs the same ncclAllReduce / ncclAllGather kernels on the same streams; tests/test_gpu_dp.py)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("PCCX_DIST_SINGLE_RANK") == "1"


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
    if not collectives_active():
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
    if collectives_active():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])


def allreduce_mean_(tensors, bucket_bytes=64 << 20):
    """Data-parallel gradient averaging (the one collective of the configs[4] training step): flatten the
    tensors into buckets of ``bucket_bytes`` (xGMI is point-to-point, ring all-reduce is per-link bound, so
    a few large messages beat many small ones; the pppe model is 116 MB of fp32 gradients = 2 buckets),
    all-reduce each bucket over the default process group (RCCL on GPUs, gloo in the CPU rehearsal) and
    scatter the averages back in place.  No-op without a process group."""
    if not collectives_active():
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


class GradBuckets:
    """Gradient averaging OVERLAPPED with the backward pass (the data-parallel step of configs[4], train_pppe_pcd_ae.py:184-226 under
    DDP): a post-accumulate hook on every parameter counts its bucket down, and a bucket whose last gradient has just been written is
    all-reduced on a SIDE stream while autograd keeps producing the earlier layers' gradients on the compute stream.  Buckets follow
    the order gradients appear in (reverse parameter order): tensors of ``big_bytes`` or more travel alone and in place (the pppe
    model's 100 MB expansion layer is produced first and its ring all-reduce, about 1 ms on 7 x 153 GB/s xGMI links, hides under
    the remaining backward); smaller ones are coalesced into flat buckets of ``bucket_bytes`` (xGMI is point to point: a few large
    messages, not many small ones).  finish() makes the compute stream wait for the side stream.  Without a process group (or with
    one rank) every call is a no-op; on CPU tensors (the gloo rehearsal) the all-reduce runs inside the hook."""

    def __init__(self, params, bucket_bytes=32 << 20, big_bytes=4 << 20):
        self.params = [p for p in params if p.requires_grad]
        self.buckets, cur, size = [], [], 0
        for p in reversed(self.params):
            nbytes = p.numel() * p.element_size()
            if nbytes >= big_bytes:
                if cur:
                    self.buckets.append(cur)
                    cur, size = [], 0
                self.buckets.append([p])
                continue
            cur.append(p)
            size += nbytes
            if size >= bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
        if cur:
            self.buckets.append(cur)
        self.bucket_of = {id(p): b for b, ps in enumerate(self.buckets) for p in ps}
        self.pending = [len(ps) for ps in self.buckets]
        self.side = None
        self.launched = 0
        self.handles = [p.register_post_accumulate_grad_hook(self._hook) for p in self.params]
        self.enabled = False

    def active(self):
        return collectives_active()

    def begin(self):
        """Call before backward()."""
        self.pending = [len(ps) for ps in self.buckets]
        self.launched = 0
        self.enabled = self.active()

    def _hook(self, p):
        if not self.enabled:
            return
        b = self.bucket_of[id(p)]
        self.pending[b] -= 1
        if self.pending[b] == 0:
            self._launch(self.buckets[b])

    def _reduce(self, ps):
        world = dist.get_world_size()
        grads = [p.grad for p in ps if p.grad is not None]
        if not grads:
            return
        if len(grads) == 1:
            dist.all_reduce(grads[0], op=dist.ReduceOp.SUM)
            grads[0] /= world
            return
        flat = torch.cat([g.reshape(-1) for g in grads])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat /= world
        off = 0
        for g in grads:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()

    def _launch(self, ps):
        self.launched += 1
        if not ps[0].is_cuda:
            self._reduce(ps)
            return
        if self.side is None:
            self.side = torch.cuda.Stream(device=ps[0].device)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        with torch.cuda.stream(self.side):
            self.side.wait_event(ev)
            self._reduce(ps)
            for p in ps:
                if p.grad is not None:
                    p.grad.record_stream(self.side)

    def finish(self):
        """Call after backward(): buckets whose parameters received no gradient this step are flushed (rank-consistent: every rank
        sees the same None pattern), then the compute stream waits for the side stream.  Returns the number of all-reduces issued."""
        if self.enabled:
            for b, ps in enumerate(self.buckets):
                if self.pending[b] > 0:
                    self.pending[b] = 0
                    self._launch(ps)
            if self.side is not None:
                torch.cuda.current_stream().wait_stream(self.side)
        self.enabled = False
        return self.launched
