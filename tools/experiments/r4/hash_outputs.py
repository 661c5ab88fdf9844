"""Bit-level fingerprint of compress + decompress of a fixed batch in the three arithmetic modes: run once per library build
(PCCX_LIB=... python tools/experiments/r4/hash_outputs.py) and compare the lines -- how a kernel variant that claims to be bit-identical
is checked against the build it replaces (raw latents, symbols, packed streams and the reconstruction all enter the hash)."""
import argparse
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "point-cloud-compression_amd"))
import numpy as np
import torch

import bench
from pccx import synth


def main():
    out = []
    rk = argparse.Namespace(dev=torch.device("cuda", 0))
    clouds = torch.from_numpy(np.stack([synth.cad_cloud(40 + i, 8192) for i in range(32)])).cuda()
    start = torch.arange(32, device="cuda", dtype=torch.int32) * 17
    for mm in ("f16x2", "bf16x3", "f32"):
        cd, _, _ = bench.build_codec(rk, mm, "full")
        comp = cd.compress(clouds, start, keep_extras=True)
        rec = cd.decompress(comp)
        h = hashlib.sha256()
        for k in ("latent_raw", "latent_q"):
            h.update(comp.extras[k].detach().cpu().numpy().tobytes())
        h.update(comp.packed.cpu().numpy().tobytes() if comp.packed is not None else b"")
        h.update(rec.detach().cpu().numpy().tobytes())
        out.append(f"{mm} {h.hexdigest()[:16]}")
    print(" | ".join(out))


if __name__ == "__main__":
    main()
