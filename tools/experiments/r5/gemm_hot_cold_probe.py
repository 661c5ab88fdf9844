#!/usr/bin/env python3
"""planes GEMM 512 -> 1024 (f16x2, member-max epilogue): does the time per row depend on where the input planes come from?  The same layer on
M rows whose planes fit the 256 MB Infinity Cache (replayed: hot) and on M rows that do not (cold from HBM every time)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "point-cloud-compression_amd"))
import numpy as np, torch
import pccx
from pccx import _lib, families, ops
pccx.DEFAULT_MATMUL = "f16x2"
lib = _lib.load()
K_, N_ = 512, 1024
W = (np.random.default_rng(0).standard_normal((N_, K_)) / np.sqrt(K_)).astype(np.float32)
lyr = families.FoldedLinear(torch.from_numpy(W), torch.zeros(N_), True, None, "cuda", "f16x2")
families.h2_prepare_stack([lyr], np.zeros(K_), np.ones(K_))
dyn = torch.ones(2, device="cuda")
for M in (8192, 32768, 65536, 131072, 262144, 524288):
    src = torch.rand(M, K_, device="cuda")
    pin = torch.empty(lib.pccx_planes_floats_h2(M, K_), device="cuda", dtype=torch.float32)
    _lib.call("pccx_group_planes_h2", src.data_ptr(), K_, K_, None, 0, 0, None, M, 1, 1, float(lyr.h2["sig"]), None, pin.data_ptr(), ops._stream())
    member = torch.ones(M, device="cuda", dtype=torch.uint8)
    f = lambda: lyr.planes_h2(pin, M, 2, group=128, dyn=dyn, member=member)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("M %7d  planes %6.1f MB  %.3f ms  %.3f of the f16x2 peak" % (M, M * K_ * 4 / 2**20, ms, 2.0 * M * K_ * N_ / (ms * 1e-3) / 838.9e12), flush=True)
