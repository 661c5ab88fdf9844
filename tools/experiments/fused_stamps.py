"""Phase shares of the fused encoder from a DIAGNOSTIC build (s_memtime stamps at the phase boundaries, thread 0 of every workgroup;
the product kernel carries no stamps).  Needs libpccx_stamps.so in place of libpccx.so (built from encoder_fused.hip with the FU_STAMP
edits recorded in DESIGN.md section 4).  python tools/experiments/fused_stamps.py"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "point-cloud-compression_amd"))
from pccx import _lib, models, ops, synth  # noqa: E402

lib = _lib.load()
fn = lib.pccx_debug_fused_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
B, N, S, K = 256, 8192, 64, 256
dev = torch.device("cuda:0")
base = np.stack([synth.cad_cloud(11 + i, N) for i in range(32)])
clouds = torch.from_numpy(np.concatenate([base] * (B // 32))).to(dev)
xyz, _, _ = ops.normalize(clouds)
idx = ops.farthest_point_sample_batch(xyz, S, torch.zeros(B, dtype=torch.int32, device=dev))
centres = ops.index_points(xyz, idx)
patches = ops.knn_points(centres, xyz, K, True, 2.0)[2].reshape(B * S, K, 3)
ae = models.AE(K, 128, 16, 7)
ae.pack(dev)
for _ in range(2):
    ae.encode(patches)
torch.cuda.synchronize()
fn(None, 1)
reps = 4
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    ae.encode(patches)
e1.record()
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 16)()
fn(out, 0)
v = [int(x) for x in out]
tot = sum(v[:5])
names = ["stage patch into LDS", "kNN-16 in the patch", "SetAbstraction + hand-over", "PointNet pass", "latent epilogue"]
print(f"{v[7]} patches, {e0.elapsed_time(e1) / reps:.2f} ms per launch of {B} clouds (stamped build)")
for n_, c in zip(names, v[:5]):
    print(f"{n_:30s} {c / v[7]:10.0f} ticks per patch  {100.0 * c / tot:5.1f} %")

if v[6]:
    print(f"  observed wave took {v[6] / v[7]:.2f} SetAbstraction units per patch (16 = an equal share)")
    for n_, c in zip(["SetAbstraction units of the observed wave", "wait at the barrier after them", "hand-over", "wait at the barrier before PointNet"], v[8:12]):
        print(f"  per pass: {n_:45s} {c / (v[7] * 2):9.0f} cycles")
