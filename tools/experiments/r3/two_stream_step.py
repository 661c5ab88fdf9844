"""compress(i) on one compute stream and decompress(i-1) on another, against both on one stream (the bench's resident leg)
    python tools/experiments/r3/two_stream_step.py [clouds=1024] [steps=10]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "point-cloud-compression_amd"))
import numpy as np, torch
import bench
from pccx import synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
class RK: dev = torch.device("cuda", 0)
cd, _, _ = bench.build_codec(RK, "f16x2", "reference")
base = np.stack([synth.cad_cloud(11 + i, 8192) for i in range(32)])
clouds = torch.from_numpy(np.concatenate([base] * (B // 32 + 1))[:B]).cuda()
starts = torch.from_numpy((np.arange(B) * 97) % 8192).cuda()
S = 64

def one_stream(n):
    comp = cd.compress(clouds, starts)
    for _ in range(n):
        c2 = cd.compress(clouds, starts)
        out = cd.decompress(comp, S=S)
        comp = c2
    return out

def two_streams(n, sa, sb):
    with torch.cuda.stream(sa):
        comp = cd.compress(clouds, starts)
        ev = torch.cuda.Event(); ev.record(sa)
    out = None
    for _ in range(n):
        with torch.cuda.stream(sb):
            sb.wait_event(ev)
            out = cd.decompress(comp, S=S)
            done = torch.cuda.Event(); done.record(sb)
        with torch.cuda.stream(sa):
            c2 = cd.compress(clouds, starts)
            ev = torch.cuda.Event(); ev.record(sa)
        comp = c2
    sa.synchronize(); sb.synchronize()
    return out

sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
ref = one_stream(2); torch.cuda.synchronize()
o2 = two_streams(2, sa, sb); torch.cuda.synchronize()
print("two-stream output equals one-stream:", bool(torch.equal(ref, o2)))
for name, fn in (("one stream", lambda: one_stream(steps)), ("two streams", lambda: two_streams(steps, sa, sb)), ("one stream", lambda: one_stream(steps)), ("two streams", lambda: two_streams(steps, sa, sb))):
    torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / (steps + 0.5)          # the extra leading compress counted as half a step
    print(f"{name}: {dt * 1e3:.2f} ms per step  {B * 8192 / dt / 1e6:.1f} M points/s")
