"""f16x2 against f32 / bf16x3 on the two fused AE transforms: differences and times (run on the GPU box).
    python tools/experiments/r3/h2_check.py [P]
"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "..", "point-cloud-compression_amd"))
import torch
from pccx import models

P = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = "cuda"
torch.manual_seed(1)
ae = models.AE(256, 128, 16, 7).to(dev)
# give the biases some size (default init is small) so that the scaled-bias path is exercised
with torch.no_grad():
    for n, p in ae.named_parameters():
        if n.endswith("bias"):
            p.mul_(3.0)
ae.pack(dev)
g = torch.Generator(device=dev); g.manual_seed(2)
x = torch.randn(P, 256, 3, device=dev, generator=g) * 0.35
x = x - x.mean(1, keepdim=True)

def enc(mode, xx):
    return ae.encode(xx, sa_matmul=mode, pn_matmul=mode)

def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3

for scale in (1.0, 37.0, 1e-3, 3000.0):
    xs = x[:4096] * scale
    r32, l32, q32 = enc("f32", xs)
    for mode in ("bf16x3", "f16x2"):
        r, l, q = enc(mode, xs)
        ok = torch.isfinite(r).all().item()
        print(f"encode scale {scale:g} {mode}: finite {ok} max|raw-raw32| {(r - r32).abs().max().item():.3e} rel {((r - r32).abs().max() / r32.abs().max()).item():.3e} "
              f"rms rel {((r - r32).pow(2).mean().sqrt() / r32.pow(2).mean().sqrt()).item():.3e} symbols differing {(q != q32).sum().item()} of {q.numel()}")

q = torch.randint(-3, 4, (4096, 16), device=dev, generator=g).float()
for qs in (1.0, 50.0):
    d32 = ae.decode(q * qs, matmul="f32")
    for mode in ("bf16x3", "f16x2"):
        dd = ae.decode(q * qs, matmul=mode)
        print(f"decode latent x{qs:g} {mode}: finite {torch.isfinite(dd).all().item()} max|d-d32| {(dd - d32).abs().max().item():.3e} "
              f"rel {((dd - d32).abs().max() / d32.abs().max()).item():.3e} rms rel {((dd - d32).pow(2).mean().sqrt() / d32.pow(2).mean().sqrt()).item():.3e}")

for mode in ("bf16x3", "f16x2"):
    print(f"encode {P} patches {mode}: {timeit(lambda: enc(mode, x)):.2f} ms")
qq = torch.randint(-3, 4, (P, 16), device=dev, generator=g).float()
for mode in ("bf16x3", "f16x2"):
    print(f"decode {P} patches {mode}: {timeit(lambda: ae.decode(qq, matmul=mode)):.2f} ms")
