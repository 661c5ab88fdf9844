#!/usr/bin/env python3
"""Per-kernel roofline table for the hot path: every kernel of SURVEY 8(a) on its configs[1] shape
(1024 clouds x 8192 points, S=64 patches of K=256; PPPF shapes for ball query), timed with HIP events
on the stream the kernels are launched on, against the bound SURVEY 8(d) assigns to it.

  python tools/kernel_rooflines.py [--clouds 1024] [--iters 5] > profiles/<round>_kernel_rooflines.json

`achieved` = algorithmic bytes (or flops) per launch / average launch time; the algorithmic figures are
SURVEY 8(d)'s per-cloud numbers, repeated in DESIGN.md.  Peaks: HBM 8 TB/s, fp32 matrix 157.3 TFLOP/s,
fp32 vector without FMA 39.3 T instr-lanes/s x2 for packed math (MI355X_MICROARCH.md).
No oracle, no reference: this only times the product path.
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "point-cloud-compression_amd"))
from pccx import models, ops, synth  # noqa: E402

HBM_PEAK = 8000.0          # GB/s
MFMA_F32_PEAK = 157.3      # TFLOP/s
MFMA_B3_PEAK = 16 * 157.3 / 6   # fp32-equivalent TFLOP/s of the bf16x3 arithmetic: bf16 dense peak / six products
MFMA_H2_PEAK = 16 * 157.3 / 3   # ... of the f16x2 arithmetic: fp16 dense peak (= bf16's) / three products
VALU_F32_PEAK = 78.6       # TFLOP/s of non-fused packed fp32 (256 CU x 4 SIMD x 16 lanes x 2 x 2.4 GHz)


def timed(fn, iters):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clouds", type=int, default=1024)
    ap.add_argument("--iters", type=int, default=5)
    args = ap.parse_args()
    B, N, S, K, k, d, L = args.clouds, 8192, 64, 256, 128, 16, 7
    dev = torch.device("cuda:0")
    base = np.stack([synth.cad_cloud(11 + i, N) for i in range(32)])
    clouds = torch.from_numpy(np.concatenate([base] * ((B + 31) // 32))[:B]).to(dev)
    starts = torch.from_numpy((np.arange(B) * 97) % N).to(dev)
    rows = []

    def row(name, ms, bound, work, unit, peak, note):
        ach = work / (ms * 1e-3) / (1e9 if unit == "GB/s" else 1e12)
        rows.append({"kernel": name, "ms_per_launch": round(ms, 4), "bound": bound, "achieved": round(ach, 2), "peak": peak,
                     "unit": unit, "frac": round(ach / peak, 4), "algorithmic_work_per_launch": work, "note": note})

    xyz, center, longest = ops.normalize(clouds)
    row("normalize", timed(lambda: ops.normalize(clouds), args.iters), "hbm", B * N * 12 * 3, "GB/s", HBM_PEAK,
        "2 reads + 1 write of the cloud")
    ms = timed(lambda: ops.farthest_point_sample_batch(xyz, S, starts), args.iters)
    row("fps (LDS/register side)", ms, "lds", B * S * N * 20, "GB/s", HBM_PEAK, "S*N*20 B per cloud; resident in registers+LDS, "
        "so this exceeds the HBM peak by design: latency-bound on S dependent rounds")
    row("fps (HBM minimum)", ms, "hbm", B * N * 12, "GB/s", HBM_PEAK, "the cloud is read once")
    idx = ops.farthest_point_sample_batch(xyz, S, starts)
    centres = ops.index_points(xyz, idx)
    row("gather (index_points)", timed(lambda: ops.index_points(xyz, idx), args.iters), "hbm", B * S * (12 + 8 + 12), "GB/s",
        HBM_PEAK, "S rows per cloud: launch-latency bound")
    ms = timed(lambda: ops.knn_points(centres, xyz, K, True, 2.0), args.iters)
    row("knn patching K=256", ms, "hbm", B * (S * N * 12 + S * K * (12 + 4 + 8)), "GB/s", HBM_PEAK,
        "S*N*12 B of candidate reads per cloud (L2/LDS side; HBM minimum 98 KB) + outputs")
    qb = xyz[:, :512].contiguous()
    cb = ops.sample_farthest_points(qb, 128)[0]
    ms = timed(lambda: ops.ball_query(cb, qb, 64, 0.2, method="scan"), args.iters)
    row("ball_query scan 128x512 r=0.2 nsample=64", ms, "hbm", B * (128 * 512 * 12 + 128 * 64 * 8), "GB/s", HBM_PEAK,
        "PPPF sa1 shape (pointnet_sa_module.py:18): ordered scan with early exit")
    nb8 = min(B, 64)
    big = xyz[:nb8].reshape(nb8 // 4, 4 * N, 3).contiguous()                      # 32768-point candidate sets
    qbig = big[:, ::4].contiguous()
    for meth in ("scan", "grid"):
        ms = timed(lambda: ops.ball_query(qbig, big, 32, 0.05, method=meth), args.iters)
        row(f"ball_query {meth} 8192x32768 r=0.05 nsample=32", ms, "hbm", (nb8 // 4) * (8192 * 27 * 12 * 40 + 32768 * 24 + 8192 * 32 * 12), "GB/s",
            HBM_PEAK, "grid hash: ~27 cells x ~40 candidates of 12 B per query + the binning pass + outputs (the scan reads up to "
            "all 32768 candidates per query from L2)")
    patches = ops.knn_points(centres, xyz, K, True, 2.0)[2]
    ms = timed(lambda: ops.octree_encode(centres, N, 0.25), args.iters)
    row("octree_encode (depth search + bits + bytes)", ms, "hbm", B * (S * 12 + 400), "GB/s", HBM_PEAK, "one wave per cloud: latency bound")
    other = clouds.roll(1, 0).contiguous()
    ms = timed(lambda: ops.nn_dist(clouds, other), args.iters)
    row("nn_dist (Chamfer / D1, one direction)", ms, "valu", B * N * N * 8, "TFLOP/s", VALU_F32_PEAK,
        "8 flop per pair, packed fp32 without FMA; fused-tiled HBM traffic is 196 KB per cloud")
    row("nn_dist as unfused bytes", ms, "hbm", B * N * N * 12, "GB/s", HBM_PEAK, "8192^2*12 B per cloud per direction if read from HBM "
        "per pair (SURVEY 8d) -- served from LDS tiles instead")

    # ---- round 3: the in-patch 16-NN selection as a kernel of its own (csrc/patch_knn.hip): vector-ALU bound
    from pccx import _lib
    pt0 = patches.reshape(B * S, K, 3)
    tab = torch.empty(_lib.load().pccx_patch_knn16_bytes(B * S, K), dtype=torch.uint8, device=dev)
    st_ = torch.cuda.current_stream().cuda_stream
    ms = timed(lambda: _lib.call("pccx_patch_knn16", pt0.data_ptr(), B * S, K, tab.data_ptr(), st_), args.iters)
    winst = B * S * K * K * 27.0 / 64                                          # 27 vector instructions per (point, candidate) pair
    rows.append({"kernel": "patch_knn16 (16-NN inside every 256-point patch)", "ms_per_launch": round(ms, 4), "bound": "valu",
                 "achieved": round(winst / (ms * 1e-3) / 1e9, 1), "peak": round(1024 * 2.4 / 2, 1), "unit": "G wave-instr/s",
                 "frac": round(winst / (ms * 1e-3) / 1e9 / (1024 * 2.4 / 2), 4), "algorithmic_work_per_launch": winst,
                 "note": "pn_kit.py:186-190; peak = 1024 SIMDs x one wave64 instruction per 2 cycles at 2.4 GHz; 8 waves per SIMD"})

    ae = models.AE(K, k, d, L)
    prob = models.ConditionalProbabilityModel(L, d)
    ae.pack(dev)
    prob.pack(dev)
    pt = patches.reshape(B * S, K, 3)
    for mode, peak in (("f32", MFMA_F32_PEAK), ("bf16x3", MFMA_B3_PEAK), ("f16x2", MFMA_H2_PEAK)):
        for _ in range(2):
            ae.encode(pt, sa_matmul=mode, pn_matmul=mode)
        t = ops.StageTimer()
        ops.set_timer(t)
        for _ in range(args.iters):
            lq = ae.encode(pt, sa_matmul=mode, pn_matmul=mode)[2]
        st = t.totals_ms()
        ops.set_timer(None)
        for name, flop in (("sa_forward", 84.7e6), ("pn_forward", 96.7e6), ("sa_pn_forward", 181.3e6)):
            if name in st:
                row(f"{name} [{mode}]", st[name][0] / st[name][1], "mfma", B * S * flop, "TFLOP/s", round(peak, 1),
                    "flop per patch from the layer shapes of AE.py:16-17" + ("; fused kernel, feature map kept in the CU" if name == "sa_pn_forward" else ""))
        row(f"ae_decode (head + main) [{mode}]", timed(lambda: ae.decode(lq, matmul=mode), args.iters), "mfma", B * S * 41.4e6, "TFLOP/s",
            round(peak, 1), "AE.py:19-27")
    ms = timed(lambda: prob.run(centres, ("cdf_int",)), args.iters)
    # executed FLOPs per cloud: model_pn 5.27 M + model_mlp per centre 2 (3 x 512 + 512 x 512 + 512 x 112) x 64 + the first Conv's feature part
    # ONCE per cloud (2 x 256 x 512) = 46.6 M; the reference's formulation (feature part per centre) counts 63.1 M
    row("prob_forward", ms, "mfma", B * 0.0466e9, "TFLOP/s", MFMA_F32_PEAK, "one workgroup per cloud, NT=1; executed FLOPs (the reference's per-centre form counts 1.35x)")
    cdf = prob.run(centres, ("cdf_int",))["cdf_int"]
    q = torch.randint(-3, 4, (B, S * d), device=dev).float()
    ms = timed(lambda: models.range_encode(cdf, q, L), args.iters)
    rows.append({"kernel": "range_encode", "ms_per_launch": round(ms, 4), "bound": "latency", "achieved": round(B * S * d / ms / 1e6, 2),
                 "unit": "G symbols/s", "note": "serial per cloud: %.0f ns per symbol per stream" % (ms * 1e6 / (S * d))})
    by, nb = models.range_encode(cdf, q, L)
    ms = timed(lambda: models.range_decode(cdf, by, nb, L), args.iters)
    rows.append({"kernel": "range_decode", "ms_per_launch": round(ms, 4), "bound": "latency", "achieved": round(B * S * d / ms / 1e6, 2),
                 "unit": "G symbols/s", "note": "serial per cloud: %.0f ns per symbol per stream" % (ms * 1e6 / (S * d))})
    # ---- the PointNet++ stacks of PPPF_AE (configs[2]) on operand planes (csrc/planes.hip), 2048 patches of 512 points
    from pccx import families
    rng = np.random.default_rng(3)
    PB = 2048
    pp = torch.from_numpy(rng.random((PB, 512, 3)).astype(np.float32)).to(dev)
    ms = timed(lambda: ops.sample_farthest_points(pp, 512), args.iters)
    row("fps one wave per cloud, 512 of 512 points x 2048 patches", ms, "lds", PB * 512 * 512 * 20, "GB/s", HBM_PEAK,
        "pointnet_sa_module.py:66-68 on K=512 patches: 512 dependent rounds per patch, no workgroup barrier")

    def stack(widths, k0):
        out, kk = [], k0
        for nw in widths:
            W = torch.from_numpy((rng.standard_normal((nw, kk)) / np.sqrt(kk)).astype(np.float32))
            out.append(families.FoldedLinear(W, torch.zeros(nw), True, matmul="bf16x3"))
            kk = nw
        return out
    for name, k0, widths, npoint, ns, nsrc in (("sa1 3-3-64-64-128, 32 samples", 3, (3, 64, 64, 128), 512, 32, 512),
                                                ("sa2 131-128-128-128-256, 64 samples", 131, (128, 128, 128, 256), 128, 64, 512)):
        st = stack(widths, k0)
        feats = torch.randn(PB, nsrc, k0 - 3, device=dev) if k0 > 3 else None
        xyz_s = torch.rand(PB, nsrc, 3, device=dev)
        idx = torch.randint(0, nsrc, (PB, npoint, ns), device=dev)
        cache = {}
        ms = timed(lambda: families.stack_max_gather(st, feats, xyz_s, idx, cache), args.iters)
        fl = 2.0 * PB * npoint * ns * sum(l.N * l.K for l in st)
        row(f"planes_chain4 (gather inside) {name}", ms, "mfma", fl, "TFLOP/s", round(MFMA_B3_PEAK, 1),
            "pointnet_sa_module.py:73-91 in one kernel; algorithmic flops of the unpadded layers")
    M3 = PB * 32 * 128
    for K_, N_, epi, grp in ((512, 1024, 2, 128), (256, 512, 0, 0), (256, 256, 0, 0)):
        lyr = stack((N_,), K_)[0]
        pin = torch.empty(families._lib.load().pccx_planes_floats(M3, K_), device=dev, dtype=torch.float32).normal_()
        ms = timed(lambda: lyr.planes(pin, M3, epi, grp), args.iters)
        row(f"planes_gemm {K_}->{N_} on {M3} rows" + (" + max over 128" if epi == 2 else ""), ms, "mfma", 2.0 * M3 * K_ * N_, "TFLOP/s",
            round(MFMA_B3_PEAK, 1), "sa3 of PPPF_AE.py:32-34; 6 B per activation in" + ("" if epi == 2 else " and out"))
        del pin
    # ---- round 5: the layers of PPPF_AE.forward in the f16x2 arithmetic, on the rows it runs them on (the N source rows per patch: 128 for
    # sa3 = 262144 rows, 256 grid points for FoldingNet = 524288 rows); operand planes built from random rows in [0, 1]
    lib = families._lib.load()
    dyn1 = torch.ones(2, device=dev)
    for K_, N_, M_, epi, what in ((259, 256, PB * 128, 0, "sa3 layer 0"), (256, 256, PB * 128, 0, "sa3 layer 1"), (256, 512, PB * 128, 0, "sa3 layer 2"),
                                  (512, 1024, PB * 128, 1, "sa3 layer 3, fp32 rows out (the form before the union maximum)"),
                                  (512, 1024, PB * 128, 3, "sa3 layer 3 as shipped: maximum over the member rows of each patch in the epilogue"),
                                  (512, 512, PB * 256, 0, "FoldingNet mlp1 layer 1")):
        lyr = stack((N_,), K_)[0]
        families.h2_prepare_stack([lyr], np.zeros(K_), np.ones(K_))
        src = torch.rand(M_, K_, device=dev)
        pin = torch.empty(lib.pccx_planes_floats_h2(M_, K_), device=dev, dtype=torch.float32)
        families._lib.call("pccx_group_planes_h2", src.data_ptr(), K_, K_, None, 0, 0, None, M_, 1, 1, float(lyr.h2["sig"]), None, pin.data_ptr(), st_)
        del src
        if epi == 3:
            member = (torch.rand(M_, device=dev) < 0.9).to(torch.uint8)
            ms = timed(lambda: lyr.planes_h2(pin, M_, 2, group=128, dyn=dyn1, member=member), args.iters)
        else:
            ms = timed(lambda: lyr.planes_h2(pin, M_, epi, 0, sig_next=lyr.h2["sig"], dyn=dyn1), args.iters)
        row(f"planes_gemm f16x2 {K_}->{N_} on {M_} rows ({what})", ms, "mfma", 2.0 * M_ * K_ * N_, "TFLOP/s", round(MFMA_H2_PEAK, 1),
            "csrc/planes.hip <2>: three fp16 MFMA products per fp32 product; 4 B per activation in" +
            (" and out" if epi == 0 else ", fp32 rows out" if epi == 1 else ", one fp32 row per 128 out"))
        del pin
    for name, k0, widths, nsrc in (("sa1 3-3-64-64-128", 3, (3, 64, 64, 128), 512), ("sa2 131-128-128-128-256", 131, (128, 128, 128, 256), 512)):
        st = stack(widths, k0)
        families.h2_prepare_stack(st, np.concatenate([np.zeros(k0 - 3), -np.ones(3)]), np.ones(k0))
        mod = families.PointnetSAModule(nsrc, 0.2, 32, list(widths), True, k0 - 3)
        f2 = torch.rand(PB * nsrc, k0 - 3, device=dev) if k0 > 3 else None
        x2 = torch.rand(PB * nsrc, 3, device=dev)
        ms = timed(lambda: mod._run_dedup_h2(st, f2, x2, PB * nsrc, k0 - 3, (dyn1, None)), args.iters)
        row(f"rows -> planes + planes_chain4 f16x2 {name} on {PB * nsrc} source rows", ms, "mfma", 2.0 * PB * nsrc * sum(l.N * l.K for l in st), "TFLOP/s",
            round(MFMA_H2_PEAK, 1), "the stack once per SOURCE row (PointnetSAModule.dedup), fp32 rows out; algorithmic flops of the unpadded layers")
    # ---- round 3: PointnetSAModule on source rows (families.PointnetSAModule.dedup): the group maxima and FoldingNet's per-point update
    for name, nsrc, Cc, npoint, ns in (("sa1", 512, 128, 512, 32), ("sa2", 512, 256, 128, 64), ("sa3", 128, 1024, 32, 128)):
        y = torch.randn(PB, nsrc, Cc, device=dev)
        idx = torch.randint(-1, nsrc, (PB, npoint, ns), device=dev)
        ms = timed(lambda: families.gather_max(y, idx), args.iters)
        row(f"gather_max {name}: {npoint} groups x {ns} samples x {Cc} ch from {nsrc} rows, 2048 patches", ms, "lds", PB * npoint * ns * Cc * 4.0,
            "GB/s", 256 * 256 * 2.4, "pointnet_sa_module.py:27-28,91 without the gathered tensor; bytes read from the LDS tile (peak 256 B/clk/CU "
            "at 2.4 GHz); HBM minimum is the (B, N, C) input once")
        del y, idx
    base_ = torch.randn(PB, 512, device=dev)
    grid_ = torch.rand(256, 2, device=dev)
    wsm = torch.randn(512, 2, device=dev)
    ms = timed(lambda: families.rows_affine_small(base_, 256, grid_, 256, wsm, True, PB * 256), args.iters)
    row("rows_affine_small (FoldingNet layer 0, per-point part) 524288 x 512", ms, "hbm", PB * 256 * 512 * 4.0, "GB/s", HBM_PEAK,
        "PPPF_AE.py:99-104: one write of the (B*P, 512) rows; the per-patch part is a Linear on B rows")
    # ---- round 3: the training step's weight-stream Linears and column reductions (csrc/train.hip)
    Wb = torch.randn(24576, 1024, device=dev)
    xb = torch.randn(4, 1024, device=dev)
    ob = torch.empty(4, 24576, device=dev)
    ms = timed(lambda: _lib.call("pccx_linear_skinny", xb.data_ptr(), 4, 1024, 1024, Wb.data_ptr(), None, 24576, 0, ob.data_ptr(), 24576, st_), args.iters)
    row("linear_skinny forward 4 x 1024 -> 24576 (pppe expansion layer)", ms, "hbm", 24576 * 1024 * 4.0, "GB/s", HBM_PEAK, "pppe_pcd_ae.py:706-714: the "
        "100 MB weight read once")
    dzb = torch.randn(4, 24576, device=dev)
    dxb = torch.zeros(4, 1024, device=dev)
    ms = timed(lambda: _lib.call("pccx_linear_skinny_dx", dzb.data_ptr(), 4, 24576, 24576, Wb.data_ptr(), 1024, 0, dxb.data_ptr(), 1024, st_), args.iters)
    row("linear_skinny dX 4 x 24576 . (24576 x 1024), split-K", ms, "hbm", 24576 * 1024 * 4.0, "GB/s", HBM_PEAK, "the same weight read once; round 2's "
        "generic layer took 2.2 ms for this call")
    zb = torch.randn(131072, 64, device=dev)
    mean_, rstd_ = torch.empty(64, device=dev), torch.empty(64, device=dev)
    sums_ = torch.zeros(128, device=dev, dtype=torch.float64)
    ms = timed(lambda: _lib.call("pccx_bn_train_stats", zb.data_ptr(), 131072, 64, 1e-5, 0.1, sums_.data_ptr(), mean_.data_ptr(), rstd_.data_ptr(), None,
                                 None, st_), args.iters)
    row("bn_train_stats (column moments) 131072 x 64", ms, "hbm", 131072 * 64 * 4.0, "GB/s", HBM_PEAK, "one read of the activation; zero-fill + reduce "
        "+ finalize launches included")
    print(json.dumps({"device": torch.cuda.get_device_name(0), "clouds_per_launch": B, "points_per_cloud": N, "rows": rows}, indent=1))


if __name__ == "__main__":
    main()
