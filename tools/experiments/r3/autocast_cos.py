import sys, copy
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/point-cloud-compression_amd")
import numpy as np, torch
import tests.test_train_step as T
from tests import synth
from pccx import families, train, synth as cs
o = T._models(2048)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
xb = torch.from_numpy(np.stack([cs.cad_cloud(500 + i, 2048) for i in range(B)]).astype(np.float32)).cuda()
rng = np.random.default_rng(8)
starts = [[rng.integers(0, 2048, B), rng.integers(0, 2048, B)], rng.integers(0, 512, B), rng.integers(0, 128, B)]
ga = families.PointCloudAE(64, 16, 2048); ga.load_state_dict(o.state_dict()); ga = ga.cuda()
gb = copy.deepcopy(ga)
la, da, ra = train.train_step(ga, train.Adam(ga.parameters(), lr=0.0), xb, starts, lam=1e-3)
lb, db, rb = train.train_step(gb, train.Adam(gb.parameters(), lr=0.0), xb, starts, lam=1e-3, autocast=True)
print("loss", la, lb, "rel", abs(la - lb) / la, "dist", da, db, "rate", ra, rb)
cs_ = []
for (k, p), (_, q) in zip(ga.named_parameters(), gb.named_parameters()):
    if p.grad is None: continue
    a, b = p.grad.double().reshape(-1), q.grad.double().reshape(-1)
    c = float((a @ b) / (a.norm() * b.norm() + 1e-300))
    cs_.append((c, k, float(a.norm())))
va = torch.cat([p.grad.reshape(-1) for p in ga.parameters() if p.grad is not None]).double()
vb = torch.cat([p.grad.reshape(-1) for p in gb.parameters() if p.grad is not None]).double()
print("full cosine", float((va @ vb) / (va.norm() * vb.norm())))
for c, k, n in sorted(cs_)[:12]:
    print(round(c, 4), k, "|g|", n)
print("layers >= 0.98:", sum(c >= 0.98 for c, _, _ in cs_), "of", len(cs_), "min", min(c for c, _, _ in cs_))
