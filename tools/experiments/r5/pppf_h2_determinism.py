#!/usr/bin/env python3
"""Is the f16x2 PPPF forward deterministic call to call, and where does it leave the bf16x3 run?  (small ragged batches of the test)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "point-cloud-compression_amd"))
import numpy as np, torch
import bench, pccx
from pccx import families
g = families.PPPF_AE(512, 0, 16, 7)
if os.environ.get("PROBE_WEIGHTS") == "test":           # the weights of tests/test_families.py (tweaked BatchNorm statistics)
    from oracle import ref_families as rf
    from tests import synth as tsynth
    m = rf.PPPF_AE(512, 0, 16, 7).eval()
    m.load_state_dict(tsynth.family_tweak(rf.seeded_with_bn(m, tsynth.PPPF_SEED), "pppf"))
    g.load_state_dict(m.state_dict())
else:
    m = None
    g.load_state_dict(bench.seeded_state_dict(g, 21))
    for k_, v in g.state_dict().items():
        if k_.endswith("running_var"):
            v.fill_(1.0)
g.pack("cuda")
rng = np.random.default_rng(2)
for mul in (1.6, 64.0, 1.0 / 64.0, 1.0):
    xs = torch.from_numpy((rng.random((3, 512, 3)) * mul).astype(np.float32)).cuda()
    pccx.DEFAULT_MATMUL = "bf16x3"
    ref = [t.clone() for t in g(xs)]
    pccx.DEFAULT_MATMUL = "f16x2"
    outs = []
    for i in range(6):
        o = [t.clone() for t in g(xs)]
        outs.append(o)
        torch.cuda.synchronize()
    same = [all(torch.equal(a, b) for a, b in zip(outs[0], o)) for o in outs[1:]]
    d = [float((o[1] - ref[1]).abs().max()) for o in outs]
    per_patch = (outs[-1][1] - ref[1]).abs().amax(dim=1).cpu().numpy()
    if m is not None:
        with torch.no_grad():
            ol = m(xs.cpu())[1]
        print("   vs oracle: f16x2 per patch %s, bf16x3 per patch %s" % ((outs[-1][1].cpu() - ol).abs().amax(dim=1).numpy(), (ref[1].cpu() - ol).abs().amax(dim=1).numpy()))
    print("mul %g: repeat-identical %s; max |latent - bf16x3| per call %s; per patch (last call) %s; dyn %s" %
          (mul, same, ["%.2e" % v for v in d], per_patch, g._packed["h2"]["dyn"].cpu().numpy()[:6]), flush=True)
