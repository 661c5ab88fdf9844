#!/usr/bin/env python3
"""Per-call wall time of the PPPF_AE forward from a fresh process (is a one-time stall inside a short timed region?)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "point-cloud-compression_amd"))
import numpy as np, torch
import bench, pccx
from pccx import families, ops, synth
Kp, N, B = 512, 2048, 256
S = N * 2 // Kp
model = families.PPPF_AE(K=Kp, k=Kp // 2, d=16, L=7)
model.load_state_dict(bench.seeded_state_dict(model, 21))
model.pack("cuda")
clouds = torch.from_numpy(np.stack([synth.cad_cloud(300 + i, N) for i in range(32)])).cuda().repeat(B // 32, 1, 1).contiguous()
cent = ops.index_points(clouds, ops.farthest_point_sample_batch(clouds, S, torch.zeros(B, dtype=torch.int32)))
patches = ops.knn_points(cent, clouds, Kp, patch_scale=float((N / 1024) ** (1 / 3))).knn.view(B * S, Kp, 3).contiguous()
pccx.DEFAULT_MATMUL = "f16x2"
if os.environ.get("PRIME"):
    from pccx import _lib
    buf = torch.zeros(64, device="cuda")
    t0 = time.perf_counter()
    for _ in range(int(os.environ["PRIME"])):
        _lib.call("pccx_zero_bytes", buf.data_ptr(), 64, ops._stream())
    torch.cuda.synchronize()
    print("primed with", os.environ["PRIME"], "launches in %.1f ms" % (1e3 * (time.perf_counter() - t0)), flush=True)
for i in range(12):
    timer = ops.StageTimer() if (i < 4 and not os.environ.get("NOTIMER")) else None
    ops.set_timer(timer)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = model(patches)
    torch.cuda.synchronize()
    ops.set_timer(None)
    if timer is not None:
        print("   stages", {k: round(ms, 2) for k, (ms, n) in timer.totals_ms().items()}, flush=True)
    st = torch.cuda.memory_stats()
    print(i, "%.2f ms" % (1e3 * (time.perf_counter() - t0)), "reserved %.2f GB" % (torch.cuda.memory_reserved() / 2**30), "mallocs", st.get("num_device_alloc"), "frees", st.get("num_device_free"), flush=True)
