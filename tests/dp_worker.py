"""Worker of tests/test_gpu_dp.py: one of two ranks that SHARE cuda:0, process group over gloo with CUDA tensors (RCCL refuses two ranks
on one device; the gloo collectives take the same code path through pccx.dist: event, side stream, all_reduce + in-place divide on
p.grad, record_stream, finish()).  Prints one JSON line with the checks; exit code 0 only if it ran to the end.

What is checked is the MECHANICS of the data-parallel step, not a gradient value (on this small model the gradient of one batch
moves by tens of percent from run to run -- a latent that sits on a rounding boundary of the quantiser flips --, so "equals the mean of
two separately computed local gradients" cannot be asserted tightly): every all_reduce the step issues is recorded (a clone of its input
before, its tensor after the in-place average), the inputs are exchanged between the ranks, and
  * each averaged tensor equals the mean over ranks of what the ranks put in (to fp32 rounding),
  * the ranks' inputs really differ (the check is not vacuous),
  * after the step every parameter's .grad and every parameter is BIT-IDENTICAL on the two ranks (a gradient that missed its
    bucket, or a bucket averaged before its last gradient was written, would differ),
  * the eager step used the side stream and several buckets; the captured step is two graphs around the exchange."""
import copy
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "point-cloud-compression_amd"))
import numpy as np
import torch
import torch.distributed as dist

from pccx import families, train
from tests import synth

RECORD = []
_real_all_reduce = dist.all_reduce


def _recording_all_reduce(t, *a, **k):
    pre = t.detach().clone()
    r = _real_all_reduce(t, *a, **k)
    RECORD.append((pre, t))
    return r


BACKEND = os.environ.get("DP_BACKEND", "gloo")               # "nccl": the one-rank RCCL rehearsal (collectives on device tensors)


def _x(t):
    """where the checking collectives' tensors live: the host under gloo, the device under RCCL"""
    return t.cuda() if BACKEND == "nccl" else t.cpu()


def same_on_all_ranks(tensors, world):
    flat = _x(torch.cat([t.detach().reshape(-1).float() for t in tensors]))
    both = [torch.zeros_like(flat) for _ in range(world)]
    _real_all_reduce_gather(both, flat)
    return bool(all(torch.equal(both[0], b) for b in both[1:]))


def _real_all_reduce_gather(out, t):
    dist.all_gather(out, t)


def check_records(world):
    """(worst relative error of averaged-vs-mean-of-inputs over the recorded calls, largest relative spread of the inputs, calls)"""
    worst, spread = 0.0, 0.0
    for pre, post in RECORD:
        mine = _x(pre.reshape(-1))
        all_pre = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(all_pre, mine)
        mean = torch.stack(all_pre).double().mean(0).float()
        scale = float(mean.abs().max()) + 1e-30
        worst = max(worst, float((_x(post.detach().reshape(-1)) - mean).abs().max()) / scale)
        if world > 1:
            spread = max(spread, float((all_pre[0] - all_pre[1]).abs().max()) / scale)
    return worst, spread, len(RECORD)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    if BACKEND == "nccl":
        from pccx import launch
        assert world == 1 and os.environ.get("PCCX_DIST_SINGLE_RANK") == "1", "RCCL refuses two ranks on one device"
        launch.init_process_group("nccl", torch.device("cuda", 0))      # the product's own entry (launch.py), device_id bound
    else:
        dist.init_process_group("gloo")
    N, B = 2048, 4
    torch.manual_seed(5)
    base = families.PointCloudAE(64, 16, N)
    for k, v in base.state_dict().items():
        if k.endswith("running_var"):
            v.fill_(1.0)
    base = base.cuda()
    x = torch.from_numpy(synth.train_input(B * world, N)[rank * B:(rank + 1) * B].copy()).cuda()       # every rank its own clouds
    rng = np.random.default_rng(7 + rank)
    starts = [[rng.integers(0, N, B), rng.integers(0, N, B)], rng.integers(0, 512, B), rng.integers(0, 128, B)]
    res = {"rank": rank, "backend": dist.get_backend(), "world": world}
    print("stage: group up", file=sys.stderr, flush=True)
    dist.all_reduce = _recording_all_reduce                  # pccx.dist calls torch.distributed.all_reduce by attribute

    # 1. eager step with the overlapped buckets (GradBuckets' CUDA branch: side stream, in-place all_reduce on p.grad)
    m1 = copy.deepcopy(base)
    opt1 = train.Adam(m1.parameters(), lr=1e-4)
    RECORD.clear()
    train.train_step(m1, opt1, x, starts, lam=1e-3, data_parallel=True)
    torch.cuda.synchronize()
    res["buckets_launched"] = int(opt1._dp.launched)
    res["side_stream"] = opt1._dp.side is not None
    res["eager_avg_err"], res["eager_input_spread"], res["eager_calls"] = check_records(world)
    n_grad = sum(p.grad.numel() for p in m1.parameters() if p.grad is not None)
    res["eager_covered"] = sum(pre.numel() for pre, _ in RECORD) == n_grad          # every gradient element went through exactly one all_reduce
    res["eager_grads_equal_across_ranks"] = same_on_all_ranks([p.grad for p in m1.parameters() if p.grad is not None], world)
    res["eager_params_equal_across_ranks"] = same_on_all_ranks(list(m1.parameters()), world)

    print("stage: eager step checked", file=sys.stderr, flush=True)
    # 2. the captured step: two graphs around the all-reduce (warmup=0: the first replay starts from the same state)
    m2 = copy.deepcopy(base)
    opt2 = train.Adam(m2.parameters(), lr=1e-4)
    # ... in the form bench.py runs at N > 1: selection tables of the next batch prefetched on a side stream (prefetch=True)
    gs = train.GraphedTrainStep(m2, opt2, x, starts, lam=1e-3, warmup=0, data_parallel=True, prefetch=True)
    RECORD.clear()
    gs.prefetch(x, starts)
    gs(sync=False, next_batch=(x, starts))
    torch.cuda.synchronize()
    res["graph_avg_err"], res["graph_input_spread"], res["graph_calls"] = check_records(world)
    res["graph_covered"] = sum(pre.numel() for pre, _ in RECORD) == sum(g.numel() for g in gs._dp_grads)
    res["graph_grads_equal_across_ranks"] = same_on_all_ranks(gs._dp_grads, world)
    res["graph_params_equal_across_ranks"] = same_on_all_ranks(list(m2.parameters()), world)
    res["two_graphs"] = gs.graph_opt is not None
    gs(sync=False)                                           # the prefetched batch: a second replay pair around a second all-reduce
    torch.cuda.synchronize()
    res["second_step_params_equal_across_ranks"] = same_on_all_ranks(list(m2.parameters()), world)
    res["second_step_finite"] = bool(all(torch.isfinite(p).all() for p in m2.parameters()))
    differ = world == 1 or (res["eager_input_spread"] > 1e-2 and res["graph_input_spread"] > 1e-2)
    if BACKEND == "nccl":                                    # the file-sharded workloads' two exchanges over RCCL as well (pccx/dist.py)
        from pccx import dist as pdist
        g = pdist.gather_summaries([1.0, 2.0, 3.0, 4.0, 5.0, 6.0], torch.device("cuda", 0))
        res["summaries_ok"] = bool(g.is_cuda and g.shape == (1, 6) and g[0].tolist() == [1.0, 2.0, 3.0, 4.0, 5.0, 6.0])
        res["max_ok"] = pdist.max_over_ranks(0.125, torch.device("cuda", 0)) == 0.125
        differ = differ and res["summaries_ok"] and res["max_ok"]
    res["ok"] = bool(res["eager_avg_err"] <= 1e-5 and res["graph_avg_err"] <= 1e-5 and differ
                     and res["eager_covered"] and res["graph_covered"] and res["eager_grads_equal_across_ranks"]
                     and res["graph_grads_equal_across_ranks"] and res["eager_params_equal_across_ranks"]
                     and res["graph_params_equal_across_ranks"] and res["side_stream"] and res["buckets_launched"] >= 2 and res["two_graphs"]
                     and res["second_step_params_equal_across_ranks"] and res["second_step_finite"])
    print(json.dumps(res), flush=True)
    print("stage: checks done", file=sys.stderr, flush=True)
    torch.cuda.synchronize()
    dist.barrier()
    print("stage: barrier done", file=sys.stderr, flush=True)
    dist.destroy_process_group()
    print("stage: group destroyed", file=sys.stderr, flush=True)


if __name__ == "__main__":
    main()
