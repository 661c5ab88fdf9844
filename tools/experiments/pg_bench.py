"""Times pccx_planes_gemm on the PPPF layer shapes (rows of 2048 patches).  python tools/experiments/pg_bench.py [reps]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "point-cloud-compression_amd"))
from pccx import families  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
SHAPES = [(8388608, 512, 1024, 2, 128), (8388608, 256, 512, 0, 0), (8388608, 256, 256, 0, 0), (16777216, 128, 128, 0, 0),
          (16777216, 128, 256, 2, 64), (33554432, 64, 128, 2, 32)]
rng = np.random.default_rng(0)
for M, K, N, epi, grp in SHAPES:
    W = torch.from_numpy(rng.standard_normal((N, K)).astype(np.float32) / np.sqrt(K))
    lyr = families.FoldedLinear(W, torch.zeros(N), True, matmul="bf16x3")
    pin = torch.empty(families._lib.load().pccx_planes_floats(M, K), device="cuda", dtype=torch.float32).normal_()
    out = lyr.planes(pin, M, epi, grp)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = lyr.planes(pin, M, epi, grp)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = 2.0 * M * K * N
    by = 6.0 * M * K + (6.0 * M * N if epi == 0 else 0)
    print(f"M={M} K={K} N={N} epi={epi}: {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s ({fl / ms / 1e9 / 419.5:.3f} of bf16x3 peak)  {by / ms / 1e9:6.2f} TB/s activations", flush=True)
    del pin, out
