"""CPU property test of the arithmetic behind the EXPERIMENTAL bf16x3 kernels: an fp32 value
splits into three bf16 pieces with x == hi + mid + lo EXACTLY, and the six products of weight i + j <= 4 recover an
fp32 dot product to fp32 accuracy.  numpy restatement of b3_split / b3_split8 (round to nearest even at each level)."""
import numpy as np


def bf16_rne(x):
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16
    return (u.astype(np.uint32) << 16).view(np.float32)


def split3(x):
    hi = bf16_rne(x)
    r1 = (x - hi).astype(np.float32)
    mid = bf16_rne(r1)
    r2 = (r1 - mid).astype(np.float32)
    lo = bf16_rne(r2)
    return hi, mid, lo, r1, r2


def test_split_is_exact_for_normal_floats():
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(200000).astype(np.float32) * np.float32(10.0) ** rng.integers(-20, 20, 200000).astype(np.float32),
                        np.float32([0.0, 1.0, -1.0, 1.0 + 2.0 ** -23, 1.0 - 2.0 ** -24, 3.3e38, 1.2e-30, 0.1, 255.99998])])
    hi, mid, lo, r1, r2 = split3(x)
    # every residual is exact in fp32 (float64 check), and the last piece is itself a bf16 value
    assert np.array_equal(r1.astype(np.float64), x.astype(np.float64) - hi.astype(np.float64))
    assert np.array_equal(r2.astype(np.float64), r1.astype(np.float64) - mid.astype(np.float64))
    assert np.array_equal(lo, r2)
    assert np.array_equal(hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64), x.astype(np.float64))
    for p in (hi, mid, lo):
        assert np.all(p.view(np.uint32) & 0xFFFF == 0)                      # representable in bf16


def test_six_products_give_fp32_level_dot_products():
    rng = np.random.default_rng(1)
    K = 1024
    a = (rng.random((64, K), dtype=np.float32) * 2 - 1)
    b = (rng.random((K, 64), dtype=np.float32) * 2 - 1)
    ref = a.astype(np.float64) @ b.astype(np.float64)
    mag = np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64)
    ah, am, al, _, _ = split3(a)
    bh, bm, bl, _, _ = split3(b)
    acc = np.zeros((64, 64), np.float32)
    for k0 in range(0, K, 32):                                              # per K = 32 block, smallest products first, fp32 accumulate
        for pa, pb in ((al, bh), (ah, bl), (am, bm), (am, bh), (ah, bm), (ah, bh)):
            acc = (acc + (pa[:, k0:k0 + 32].astype(np.float64) @ pb[k0:k0 + 32].astype(np.float64)).astype(np.float32)).astype(np.float32)
    seq = np.zeros((64, 64), np.float32)
    for k in range(K):                                                      # an fp32 fma chain, for scale
        seq = (seq + (a[:, k:k + 1].astype(np.float64) * b[k:k + 1].astype(np.float64)).astype(np.float32)).astype(np.float32)
    e6 = np.abs(acc - ref).max() / mag.max()
    e32 = np.abs(seq - ref).max() / mag.max()
    assert e6 <= 4 * e32 + 1e-9, (e6, e32)                                  # same order as an fp32 summation
    assert e6 < 5e-7
    # dropping the three small products is visibly worse
    acc3 = np.zeros((64, 64), np.float32)
    for pa, pb in ((am, bh), (ah, bm), (ah, bh)):
        acc3 = (acc3 + (pa.astype(np.float64) @ pb.astype(np.float64)).astype(np.float32)).astype(np.float32)
    assert np.abs(acc3 - ref).max() / mag.max() > 2 * e6
