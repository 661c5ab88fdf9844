#!/usr/bin/env python3
"""PPPF_AE forward (f16x2), levels 2 and 3 fed by padded rows (pccx_gather_max_rows + gathering kernels) against the operand-plane pass.
Same box, alternating; 2048 patches of 512 points (the bench shape)."""
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
for k_, v in model.state_dict().items():
    if k_.endswith("running_var"):
        v.fill_(1.0)
model.pack("cuda")
clouds = torch.from_numpy(np.stack([synth.cad_cloud(300 + i, N) for i in range(32)])).cuda().repeat(B // 32, 1, 1).contiguous()
cent = ops.index_points(clouds, ops.farthest_point_sample_batch(clouds, S, torch.zeros(B, dtype=torch.int32)))
patches = ops.knn_points(cent, clouds, Kp, patch_scale=float((N / 1024) ** (1 / 3))).knn.view(B * S, Kp, 3).contiguous()
pccx.DEFAULT_MATMUL = "f16x2"
res = {}
for um in (False, True, False, True):  # here: padded_levels
    families.PointnetSAModule.padded_levels = um
    for _ in range(3):
        out = model(patches)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        out = model(patches)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    res[um] = [t.clone() for t in out]
    timer = ops.StageTimer(); ops.set_timer(timer)
    for _ in range(5):
        model(patches)
    ops.set_timer(None)
    st = {k: round(ms / 5, 3) for k, (ms, n) in sorted(timer.totals_ms().items(), key=lambda kv: -kv[1][0])}
    print("padded_levels", um, "%.3f ms" % (1e3 * dt), st, flush=True)
print("identical:", all(torch.equal(a, b) for a, b in zip(res[True], res[False])))
