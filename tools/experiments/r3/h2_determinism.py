"""run-to-run determinism of the f16x2 decoder with both ring chunk sizes (separate processes: the choice is read once)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "..", "point-cloud-compression_amd"))
import torch
from pccx import models
torch.manual_seed(1)
ae = models.AE(256, 128, 16, 7).to("cuda"); ae.pack("cuda")
g = torch.Generator(device="cuda"); g.manual_seed(2)
P = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
q = torch.randint(-3, 4, (P, 16), device="cuda", generator=g).float()
ref32 = ae.decode(q, matmul="f32")
outs = [ae.decode(q, matmul="f16x2").clone() for _ in range(6)]
torch.cuda.synchronize()
print("CH", os.environ.get("PCCX_DEC_H2_CH"), "run-to-run identical:", [bool(torch.equal(outs[0], o)) for o in outs[1:]],
      "max|d - f32|", [float((o - ref32).abs().max()) for o in outs], "checksum", float(outs[0].double().sum()))
bad = (outs[0] != outs[1])
if bad.any():
    idx = bad.nonzero()
    print("differing elements", int(bad.sum()), "first", idx[:5].tolist(), "patches", sorted(set((idx[:, 0] // 16).tolist()))[:20])
