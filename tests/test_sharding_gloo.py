"""CPU, world_size 2 over gloo: the N>1 path of the file-sharded runner (pccx/dist.py).
The per-file work is replaced by a deterministic function of the file index (no GPU here); what is
tested is the partition, the sharding-independent FPS start and the summary exchange."""
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pccx import dist as pdist


def _fake_file_result(i):
    """(bits, points, psnr, chamfer) of file i."""
    start = pdist.fps_start_index(11, i, 8192)
    return 5000 + 7 * i + start % 13, 8192, 30.0 + 0.1 * i, 1e-4 * (i + 1)


def _local_summary(idx, seconds):
    r = np.array([_fake_file_result(i) for i in idx], dtype=np.float64).reshape(-1, 4)
    return [r[:, 0].sum(), r[:, 1].sum(), r[:, 2].sum(), r[:, 3].sum(), float(len(idx)), seconds]


def _worker(rank, world, port, n_files, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    idx = pdist.shard_indices(n_files, rank, world)
    g = pdist.gather_summaries(_local_summary(idx, 1.0 + rank))
    tmax = pdist.max_over_ranks(1.0 + rank)
    q.put((rank, idx, pdist.reduce_summaries(g), tmax))
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process():
    n_files, world = 37, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_files, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    owned = sorted(i for _, idx, _, _ in res for i in idx)
    assert owned == list(range(n_files))                       # every file exactly once
    single = pdist.reduce_summaries(torch.tensor([_local_summary(list(range(n_files)), 2.0)], dtype=torch.float64))
    for _, _, s, tmax in res:
        assert tmax == 2.0                                     # MAX over ranks
        assert s["files"] == n_files
        for k in ("bpp", "d1_psnr_db", "chamfer", "points_per_s"):
            assert abs(s[k] - single[k]) <= 1e-12 * abs(single[k])


def test_fps_start_is_sharding_independent():
    a = [pdist.fps_start_index(11, i, 8192) for i in range(10)]
    assert a == [pdist.fps_start_index(11, i, 8192) for i in range(10)]
    assert len(set(a)) > 5 and all(0 <= v < 8192 for v in a)


def test_single_process_fallback_needs_no_process_group():
    g = pdist.gather_summaries([1, 2, 3, 4, 5, 6])
    assert g.shape == (1, 6) and pdist.max_over_ranks(3.5) == 3.5


def _ar_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(rank)
    grads = [torch.randn(s, generator=g) for s in ((7, 3), (1000,), (64, 64), (5,), (300, 40))] + [None]
    keep = [t.clone() for t in grads if t is not None]
    nb = pdist.allreduce_mean_(grads, bucket_bytes=4096)          # forces several buckets
    q.put((rank, nb, [t.numpy() for t in grads if t is not None], [t.numpy() for t in keep]))
    dist.destroy_process_group()


def test_bucketed_gradient_allreduce_two_ranks():
    """The collective of the data-parallel training step: bucketed mean over ranks equals the plain mean."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_ar_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1] >= 2
    for i in range(len(res[0][2])):
        want = (res[0][3][i] + res[1][3][i]) / 2
        np.testing.assert_allclose(res[0][2][i], want, rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(res[1][2][i], want, rtol=1e-6, atol=1e-7)


def test_bench_self_launch_two_ranks_gloo():
    """`python bench.py --gpus 2` with NO torchrun environment starts its own two rank processes (pccx/launch.py) before
    anything touches a GPU, rendezvous on 127.0.0.1, and rank 0 prints exactly one JSON line carrying n_gpus = 2 and the
    all-gathered dist.SUMMARY_FIELDS reduction.  (launch-check does no GPU work; gloo stands in for RCCL on CPU.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["PCCX_ASSERT_PARENT_GPU_FREE"] = "1"                    # the parent exits non-zero if torch / the HIP library got imported
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--workload",
                        "launch-check", "--steps", "3"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]      # gloo itself prints a connection note on stdout
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3
    assert "launcher parent is GPU-free" in r.stderr           # the sentinel ran in the parent and found no torch import
    assert j["rccl_world"] == 2 and j["dist_backend"] == "gloo" and j["rank_devices"] == [None, None]
    s = j["summary"]                                            # ranks contributed (1000, 8192, 30, 1e-4, 1, 0.5) * (r+1)-ish
    assert s["files"] == 3 and abs(s["bpp"] - 3000.0 / (3 * 8192)) < 1e-12
    assert abs(s["d1_psnr_db"] - (30.0 + 31.0) / 3) < 1e-12
    assert abs(s["points_per_s"] - 3 * 8192 / 1.5) < 1e-9      # sum of points / MAX of seconds
    # the secondary block of the default line (stubs here: no GPU): every rank ran the legs, rank 0 emitted them ONCE, inside the one
    # line; with N > 1 no leg carries a CPU baseline (bench.cpu_leg_allowed) and every rank's NUMA placement is listed
    sec = j["secondary"]
    assert set(sec) == {"stub_a", "stub_b", "wall_s"} and sec["stub_a"]["n_gpus"] == 2
    assert j["cpu_baseline"] is None and sec["stub_a"]["cpu_baseline"] is None and sec["stub_b"]["cpu_baseline"] is None
    assert [r_["rank"] for r_ in j["rank_numa"]] == [0, 1]
    assert r.stdout.count('"secondary"') == 1
    # ... and at N = 1 the same command carries the CPU legs
    r1 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--dist-backend", "gloo", "--workload",
                         "launch-check", "--steps", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert r1.returncode == 0, r1.stderr[-2000:]
    j1 = json.loads([l for l in r1.stdout.splitlines() if l.startswith("{")][0])
    assert j1["cpu_baseline"] is not None and j1["secondary"]["stub_b"]["cpu_baseline"] is not None and j1["n_gpus"] == 1


def test_launcher_propagates_a_failing_rank(tmp_path):
    from pccx import launch
    script = tmp_path / "w.py"
    script.write_text("import os, sys, time\nr = int(os.environ['RANK'])\nassert os.environ['WORLD_SIZE'] == '2' and os.environ['MASTER_ADDR'] == '127.0.0.1'\n"
                      "if r == 1:\n    sys.exit(7)\ntime.sleep(30)\n")
    import time
    t0 = time.time()
    assert launch.spawn_ranks(str(script), [], 2) == 7          # rank 1 fails -> rank 0 is terminated, its code reported
    assert time.time() - t0 < 20


def test_bench_rank_without_a_device_exits_nonzero():
    """The launcher parent no longer counts GPUs: a rank whose LOCAL_RANK has no device must say so and exit non-zero
    (here: no GPU at all, one rank, the rank environment set by hand as a launcher would)."""
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() > 1:
        import pytest
        pytest.skip("needs a box with fewer than 2 GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RANK="1", LOCAL_RANK="1", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "1", "--cpu-clouds", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "GPU(s) visible" in r.stderr


def _gb_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(8, 300), torch.nn.ReLU(), torch.nn.Linear(300, 200), torch.nn.ReLU(), torch.nn.Linear(200, 4))
    unused = torch.nn.Parameter(torch.zeros(5))                    # a parameter that never receives a gradient
    params = list(net.parameters()) + [unused]
    gb = pdist.GradBuckets(params, bucket_bytes=2048, big_bytes=100_000)      # the 300x200 weight travels alone, the rest in flat buckets
    x = torch.randn(16, 8, generator=torch.Generator().manual_seed(100 + rank))
    res = []
    for it in range(2):                                            # hooks must re-arm
        for p in params:
            p.grad = None
        gb.begin()
        (net(x) ** 2).mean().backward()
        launched = gb.finish()
        res.append((launched, [None if p.grad is None else p.grad.numpy().copy() for p in params]))
    local = []
    for p in params:
        p.grad = None
    (net(x) ** 2).mean().backward()
    local = [None if p.grad is None else p.grad.numpy().copy() for p in params]
    q.put((rank, len(gb.buckets), res, local))
    dist.destroy_process_group()


def test_overlapped_gradient_buckets_two_ranks():
    """dist.GradBuckets (the all-reduce of the data-parallel training step, launched bucket by bucket from post-accumulate hooks
    while backward is still running): after finish() every gradient is the mean over ranks of the local gradients, a large tensor
    travels alone, parameters without a gradient are skipped consistently, and the hooks re-arm for the next step."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_gb_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, nb0, r0, l0), (_, nb1, r1, l1) = res
    assert nb0 == nb1 >= 3
    for it in range(2):
        assert r0[it][0] == r1[it][0] == nb0                        # every bucket issued exactly once per step
        for a, b, la, lb in zip(r0[it][1], r1[it][1], l0, l1):
            if la is None:
                assert a is None and b is None
                continue
            want = (la + lb) / 2
            np.testing.assert_allclose(a, want, rtol=1e-6, atol=1e-7)
            np.testing.assert_allclose(b, want, rtol=1e-6, atol=1e-7)
