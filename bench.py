#!/usr/bin/env python3
"""bench.py -- points/s of compress + decompress (BASELINE.json metric) on N MI355X GPUs.

A "step" is one pass of the hot path over one batch of synthetic clouds: the loop bodies of
compress.py:90-152 and decompress.py:80-116 for ``--batch`` clouds of 8192 points (IPDAE K=256,
configs[1]), inputs already resident in HBM when the timed region starts.  Multi-GPU: clouds are
sharded by file across ranks (one process per GPU, no data-path collective; the only collective
is the MAX of the wall time), so scaling is weak.

Prints ONE JSON line on rank 0.  Besides the driver's contract it carries
  roofline     -- the dominant kernel: algorithmic FLOPs per launch / its HIP-event duration,
                  against the fp32 matrix-core peak (MI355X_MICROARCH.md: 157.3 TFLOP/s);
  cpu_baseline -- the CPU restatement of the reference loop (oracle/ref_pipeline.py), timed on
                  this node's host cores on a bounded sample, rank 0 at N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "point-cloud-compression_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

N_POINTS, K_PATCH, ALPHA, N0, D_LAT, L_LEV = 8192, 256, 2, 1024, 16, 7
S_PATCH = N_POINTS * ALPHA // K_PATCH
AE_SEED, PROB_SEED = 11, 12
AE_LAST_GAIN = {"pn.mlp_Modules.3.0": 40.0}
PROB_GAIN = 2.0
F32_MATRIX_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md, "Peak FP32 (matrix)"

# algorithmic FLOPs per PATCH (2*MACs), from the layer shapes of AE.py:16-27
FLOP_SA = K_PATCH * 16 * (3 * 32 + 32 * 64 + 64 * 128) * 2
FLOP_PN = K_PATCH * (131 * 128 + 128 * 256 + 256 * 512 + 512 * D_LAT) * 2
K_SMALL = K_PATCH // ALPHA
FLOP_DEC = (D_LAT * 256 + 256 * 1024 + 1024 * K_SMALL * 128) * 2 + K_SMALL * (144 * 128 + 128 * 64 + 64 * 32 + 32 * 3) * 2
STAGE_FLOP = {"sa_forward": FLOP_SA, "pn_forward": FLOP_PN, "ae_decode": FLOP_DEC}
STAGE_KERNEL = {"sa_forward": "sa_forward_kernel", "pn_forward": "pn_forward_kernel", "ae_decode": "dec_main_kernel"}


def measured_traffic(stage, batch):
    """HBM bytes per launch of the stage's kernel from the committed PMC passes (profiles/round1_traffic.json,
    collected at --batch 256 with rocprofv3 --pmc in separate runs); None when not applicable."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "round1_traffic.json")))["kernels"][STAGE_KERNEL[stage]]
        return int(t["hbm_bytes_per_launch"] * batch / 256)
    except (OSError, KeyError, ValueError):
        return None


def seeded_state_dict(module, seed, gain=1.0, last_gain=None):
    """Same deterministic fill as oracle.ref_model.seeded_state_dict (kept local: the product side
    of bench.py must not import the oracle)."""
    rng = np.random.default_rng(seed)
    sd = {}
    cur = module.state_dict()
    for k, v in cur.items():
        shape = tuple(v.shape)
        if len(shape) == 0:
            sd[k] = v.clone()
            continue
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else int(shape[0])
        if k.endswith("bias"):
            fan_in = int(np.prod(tuple(cur[k[:-4] + "weight"].shape)[1:]))
        b = gain / np.sqrt(max(fan_in, 1))
        a = rng.uniform(-b, b, size=shape).astype(np.float32)
        for sub, g in (last_gain or {}).items():
            if sub in k:
                a = a * np.float32(g)
        sd[k] = torch.from_numpy(a)
    return sd


def host_cores():
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    # a one-GPU box is given a 16-core share of its host (more threads only thrash)
    return max(1, min(n, int(os.environ.get("PCCX_CPU_THREADS", "16"))))


def cpu_baseline(max_clouds, budget_s):
    """Reference-structured CPU loop (oracle) on a bounded sample of the same workload."""
    from oracle import ref_model, ref_pipeline
    from pccx import synth
    torch.set_num_threads(host_cores())
    ae = ref_model.AE(K_PATCH, K_SMALL, D_LAT, L_LEV).eval()
    ae.load_state_dict(ref_model.seeded_state_dict(ae, AE_SEED, last_gain=AE_LAST_GAIN))
    prob = ref_model.ConditionalProbabilityModel(L_LEV, D_LAT).eval()
    prob.load_state_dict(ref_model.seeded_state_dict(prob, PROB_SEED, gain=PROB_GAIN))
    ref_pipeline.compress_one(synth.cad_cloud(11, N_POINTS), ae, prob, 0)          # warm-up, discarded
    tot, n, t_start = 0.0, 0, time.time()
    while n < max_clouds and (n < 4 or time.time() - t_start < budget_s):
        pc = synth.cad_cloud(11 + n, N_POINTS)
        o, tc = ref_pipeline.compress_one(pc, ae, prob, (n * 97) % N_POINTS)
        _, td = ref_pipeline.decompress_one(o["s"], o["p"], o["c"], ae, prob)
        tot += tc + td
        n += 1
    return {"value": n * N_POINTS / tot, "unit": "points/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} synthetic 8192-pt clouds, compress+decompress windows of compress.py:85-154 / "
                      f"decompress.py:77-118, CPU restatement of the reference loop (torch CPU fp32 + C oracle)",
            "ms_per_cloud": 1e3 * tot / n}


def bench_pppe_train(args, world, rank, dev, cdev):
    """Secondary workload: one optimisation step of the pppe fast path per "step" (forward in train mode,
    Chamfer rate-distortion loss as the script builds it, backward, clip, Adam; data-parallel gradient all-reduce when N > 1)."""
    import torch.distributed as dist
    from pccx import families, synth, train
    Bt = 4                                                   # train_pppe_pcd_ae.py: batch_size 4
    model = families.PointCloudAE(64, 16, N_POINTS)
    model.load_state_dict(seeded_state_dict(model, 32))
    for k, v in model.state_dict().items():                  # sane BatchNorm statistics
        if k.endswith("running_var"):
            v.fill_(1.0)
    model = model.to(dev)
    opt = train.Adam(model.parameters(), lr=1e-3)
    x = torch.from_numpy(np.stack([synth.cad_cloud(900 + rank * Bt + i, N_POINTS) for i in range(Bt)])).to(dev)
    rng = np.random.default_rng(rank)
    starts = [[rng.integers(0, N_POINTS, Bt), rng.integers(0, N_POINTS, Bt)], rng.integers(0, 512, Bt), rng.integers(0, 128, Bt)]
    for _ in range(args.warmup):
        out = train.train_step(model, opt, x, starts, lam=1e-3, data_parallel=world > 1)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = train.train_step(model, opt, x, starts, lam=1e-3, data_parallel=world > 1)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=cdev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0])
    if rank == 0:
        print(json.dumps({
            "metric": "clouds/sec, pppe fast-path training step (forward+backward+Adam)", "value": world * Bt * args.steps / dt,
            "unit": "clouds/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "pppe PointCloudAE training step (configs[4]), batch 4 x 8192 points per GPU, fp32",
                       "parallelism": f"dp{world}", "weights": "seeded random"},
            "roofline": None, "cpu_baseline": None, "loss": out[0],
            "note": "secondary, correctness-first path: unfused layers, weights re-packed every step"}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=1024, help="clouds per GPU per step")
    ap.add_argument("--octree-mode", default="reference", choices=["reference", "full"])
    ap.add_argument("--sa-matmul", default="f32", choices=["f32", "bf16x3"], help="EXPERIMENTAL, as --decoder-matmul, for SetAbstraction")
    ap.add_argument("--pn-matmul", default="f32", choices=["f32", "bf16x3"], help="EXPERIMENTAL, as --decoder-matmul, for PointNet")
    ap.add_argument("--decoder-matmul", default="f32", choices=["f32", "bf16x3"],
                    help="bf16x3 = EXPERIMENTAL: the decoder's big Linear as fp32 products of three bf16 pieces per operand on the "
                         "bf16 matrix cores (fp32-level error, not bit-identical); the default f32 is the measured configuration")
    ap.add_argument("--cpu-clouds", type=int, default=64, help="max clouds in the CPU baseline sample (0 = skip)")
    ap.add_argument("--cpu-budget", type=float, default=20.0, help="seconds of CPU work for the baseline sample")
    ap.add_argument("--workload", default="ipdae", choices=["ipdae", "pppe-train"],
                    help="ipdae = the headline compress+decompress path (default); pppe-train = the training step of "
                         "configs[4] (train_pppe_pcd_ae.py:184-226, batch 4 x 8192 per GPU), a secondary measurement")
    ap.add_argument("--pcie", action="store_true",
                    help="also move the clouds host->device and the streams / reconstruction device->host inside the "
                         "timed region (the PCIe-inclusive rate quoted in DESIGN.md; never the headline value)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (production); gloo only to rehearse the N>1 path on one GPU")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.dist_backend == "gloo":
        local = local % max(torch.cuda.device_count(), 1)      # rehearsal: ranks may share a GPU
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cdev = dev if args.dist_backend == "nccl" else torch.device("cpu")   # where the tiny collectives live
    if world > 1:
        import torch.distributed as dist
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    from pccx import codec, models, ops, synth

    if args.workload == "pppe-train":
        return bench_pppe_train(args, world, rank, dev, cdev)

    ae = models.AE(K_PATCH, K_SMALL, D_LAT, L_LEV)
    ae.load_state_dict(seeded_state_dict(ae, AE_SEED, last_gain=AE_LAST_GAIN))
    prob = models.ConditionalProbabilityModel(L_LEV, D_LAT)
    prob.load_state_dict(seeded_state_dict(prob, PROB_SEED, gain=PROB_GAIN))
    ae.pack(dev)
    prob.pack(dev)
    cd = codec.Codec(ae, prob, K=K_PATCH, ALPHA=ALPHA, N0=N0, octree_mode=args.octree_mode, decoder_matmul=args.decoder_matmul, sa_matmul=args.sa_matmul, pn_matmul=args.pn_matmul)

    B = args.batch
    # shard by file: global cloud i -> rank i % world; 32 distinct shapes per rank, tiled
    base = np.stack([synth.cad_cloud(11 + rank + world * i, N_POINTS) for i in range(min(B, 32))])
    clouds = torch.from_numpy(np.concatenate([base] * ((B + base.shape[0] - 1) // base.shape[0]))[:B]).to(dev)
    starts = torch.from_numpy((np.arange(B) * 97 + rank) % N_POINTS).to(dev)

    host_clouds = clouds.cpu().pin_memory() if args.pcie else None

    def step():
        if args.pcie:
            x = host_clouds.to(dev, non_blocking=True)
            comp = cd.compress(x, starts)
            comp.to_host()                                   # .s.bin / .p.bin / .c.bin bytes to the host
            out = cd.decompress(comp)
            out.cpu()                                        # reconstructed XYZ to the host
            return comp, out
        comp = cd.compress(clouds, starts)
        return comp, cd.decompress(comp)

    for _ in range(args.warmup):
        comp, out = step()
    torch.cuda.synchronize()
    timer = ops.StageTimer()
    ops.set_timer(timer)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        comp, out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    ops.set_timer(None)
    if world > 1:
        t = torch.tensor([dt], device=cdev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0])

    stages = {k: (ms / n, n) for k, (ms, n) in timer.totals_ms().items()}
    bpp = float(comp.bpp().mean())
    psnr = float(codec.d1_psnr(clouds, out).mean())
    if world > 1:
        # the only data-path-adjacent exchange: tiny per-rank quality summaries
        summ = torch.tensor([bpp, psnr], device=cdev, dtype=torch.float64)
        gathered = [torch.zeros_like(summ) for _ in range(world)]
        dist.all_gather(gathered, summ)
        bpp, psnr = [float(x) for x in torch.stack(gathered).mean(0)]

    if rank == 0:
        P = B * S_PATCH
        per_step_ms = {k: v[0] * v[1] / args.steps for k, v in stages.items()}
        dom = max(STAGE_FLOP, key=lambda k: per_step_ms.get(k, 0.0))
        dur_ms = stages[dom][0]
        achieved = STAGE_FLOP[dom] * P / (dur_ms * 1e-3) / 1e12
        res = {
            "metric": "points/sec compress+decompress (ModelNet40-shaped 8192 K=256)",
            "value": world * B * N_POINTS * args.steps / dt, "unit": "points/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.decoder_matmul == args.sa_matmul == args.pn_matmul == "f32" else
                     "f32 (EXPERIMENTAL: bf16x3-split operands, fp32 accumulate, in: %s)" % "+".join(
                         n for n, v in (("decoder", args.decoder_matmul), ("sa", args.sa_matmul), ("pn", args.pn_matmul)) if v != "f32"),
            "data": "synthetic",
            "config": {"workload": "IPDAE K=256 d=16 L=7, 8192-pt CAD-like synthetic clouds (configs[1])",
                       "clouds_per_gpu_per_step": B, "points_per_cloud": N_POINTS, "patches_per_cloud": S_PATCH,
                       "octree_mode": args.octree_mode, "sharding": f"file-sharded x{world}", "weights": "seeded random",
                       "pcie_inclusive": bool(args.pcie), "decoder_matmul": args.decoder_matmul, "sa_matmul": args.sa_matmul, "pn_matmul": args.pn_matmul},
            "roofline": {"kernel": dom, "bound": "mfma", "achieved": achieved, "peak": F32_MATRIX_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / F32_MATRIX_PEAK_TFLOPS, "traffic": measured_traffic(dom, B),
                         "launch_ms": dur_ms, "flop_per_launch": STAGE_FLOP[dom] * P},
            "stage_ms_per_step": {k: round(v, 4) for k, v in sorted(per_step_ms.items(), key=lambda kv: -kv[1])},
            "mfma_stage_tflops": {k: STAGE_FLOP[k] * P / (stages[k][0] * 1e-3) / 1e12 for k in STAGE_FLOP if k in stages},
            "bpp": bpp, "d1_psnr_db": psnr,
        }
        res["cpu_baseline"] = None
        print("[bench] gpu leg done: %.3e points/s, %.2f ms/step; dominant %s %.1f TFLOP/s" %
              (res["value"], res["ms_per_step"], dom, achieved), file=sys.stderr, flush=True)
        if world == 1 and args.cpu_clouds > 0:
            try:
                res["cpu_baseline"] = cpu_baseline(args.cpu_clouds, args.cpu_budget)
                res["gpu_over_cpu"] = res["value"] / res["cpu_baseline"]["value"]
            except Exception as e:   # the GPU line must survive a broken host toolchain
                res["cpu_baseline_error"] = repr(e)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
