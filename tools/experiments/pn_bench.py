#!/usr/bin/env python3
"""Time the PointNet bf16x3 kernel variants (PCCX_PN_B3_VARIANT) on the bench shape and check them against the exact-fp32 kernel.
usage: python tools/experiments/pn_bench.py [clouds=1024] [variants=-1,0,1,...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "point-cloud-compression_amd"))
import numpy as np
import torch
from bench import seeded_state_dict, AE_SEED, AE_LAST_GAIN
from pccx import models, synth, ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
variants = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [-1, 100]
K, k, d, L = 256, 128, 16, 7
ae = models.AE(K, k, d, L)
ae.load_state_dict(seeded_state_dict(ae, AE_SEED, last_gain=AE_LAST_GAIN))
ae.pack("cuda")
base = np.stack([synth.cad_cloud(11 + i, 8192) for i in range(32)])
clouds = torch.from_numpy(np.concatenate([base] * (B // 32 + 1))[:B]).cuda()
pcn, _, _ = ops.normalize(clouds)
cent = ops.index_points(pcn, ops.farthest_point_sample_batch(pcn, 64, torch.zeros(B, dtype=torch.int32)))
patches = ops.knn_points(cent, pcn, K, patch_scale=2.0).knn.view(B * 64, K, 3).contiguous()
P = patches.shape[0]
feat = torch.empty(P * K * 128, device="cuda")
ae._launch_sa(patches, feat, "f32")
outs = lambda: [torch.empty(P, d, device="cuda") for _ in range(3)]
ref = outs()
ae._launch_pn(patches, feat, ref, "f32")
torch.cuda.synchronize()
for v in variants:
    os.environ["PCCX_PN_B3_VARIANT"] = str(v)
    o = outs()
    for _ in range(2):
        ae._launch_pn(patches, feat, o, "bf16x3")
    torch.cuda.synchronize()
    a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    n = 5
    for _ in range(n):
        ae._launch_pn(patches, feat, o, "bf16x3")
    b_.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b_) / n
    err = float((o[0] - ref[0]).abs().max())
    flips = int((o[2] != ref[2]).sum())
    if v in ():
        import ctypes
        from pccx import _lib
        buf = (ctypes.c_ulonglong * 8)()
        lib = _lib.load()
        lib.pccx_debug_pn_b3_stamps(buf)
        ae._launch_pn(patches, feat, o, "bf16x3")
        torch.cuda.synchronize()
        lib.pccx_debug_pn_b3_stamps(buf)
        w_, b2, i_, nb, life, waves = [int(x) for x in buf[:6]]
        print(f"   stamps per wave: lifetime {life / waves:.0f} cyc; per boundary: wait {w_ / nb:.0f}  barrier {b2 / nb:.0f}  issue {i_ / nb:.0f}; "
              f"boundaries per wave {nb / waves:.1f}; fractions wait {w_ / life:.3f} barrier {b2 / life:.3f} issue {i_ / life:.3f}")
    print(f"variant {v:3d}: {ms:8.3f} ms per launch of {P} patches  max|raw - f32| {err:.3e}  symbol flips {flips}/{o[2].numel()}", flush=True)
