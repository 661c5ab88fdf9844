import sys, numpy as np
sys.path.insert(0, '/root/repo/point-cloud-compression_amd')
from pccx import synth
rng = np.random.default_rng(0)
def morton(p):
    q = ((p - p.min(0)) / (np.ptp(p, 0).max() + 1e-9) * 1023).astype(np.int64)
    def spread(v):
        v = (v | (v << 16)) & 0x030000FF; v = (v | (v << 8)) & 0x0300F00F; v = (v | (v << 4)) & 0x030C30C3; v = (v | (v << 2)) & 0x09249249
        return v
    return spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
res = {"radius": [], "morton": []}
for seed in range(6):
    cloud = synth.cad_cloud(100 + seed, 8192)
    for c in rng.integers(0, 8192, 6):
        d = ((cloud - cloud[c]) ** 2).sum(1)
        idx = np.argsort(d, kind="stable")[:256]
        P = cloud[idx] - cloud[c]
        D = ((P[:, None, :] - P[None, :, :]) ** 2).sum(-1)
        r16 = np.sort(D, axis=1)[:, 15]                      # 16th smallest incl. self
        for name in ("radius", "morton"):
            order = np.arange(256) if name == "radius" else np.argsort(morton(P), kind="stable")
            Dq = D[order][:, order]; rq = r16[order]
            need = Dq <= rq[:, None]                          # query needs candidate (final radius: a lower bound of the work)
            tot = skip = 0
            for w in range(4):
                nw = need[64 * w:64 * w + 64]                 # (64 queries, 256 candidates)
                g = nw.reshape(64, 64, 4).any(axis=(0, 2))    # group of 4 candidates needed by any lane
                tot += 64; skip += int((~g).sum())
            res[name].append(skip / tot)
for k, v in res.items():
    print(k, "groups skippable at the FINAL radius: mean %.3f min %.3f max %.3f" % (np.mean(v), np.min(v), np.max(v)))
