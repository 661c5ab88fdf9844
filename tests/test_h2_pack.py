"""CPU tests of the f16x2 mode's host side (csrc/pack_h2.hip; no GPU needed: the packing is pure host code behind the C ABI).

The mode forms every fp32 product from two fp16 pieces per operand; fp16 has five exponent bits, so the packing chooses exact
power-of-two scales from interval bounds of the layers.  Checked here:
  * every scale is a power of two (so scaling commutes with fp32 rounding),
  * the two planes of every packed weight reproduce it to 2^-22 relative (or fp16's subnormal spacing for negligible weights),
  * the bounds are RIGOROUS: the reference's layers (float64, oracle-free restatement of the Conv/ReLU stacks) evaluated on random
    and on adversarial corner inputs of magnitude <= 1 never exceed 2^15 after scaling -- half of fp16's largest number.
"""
import numpy as np
import pytest
import torch

from oracle import ref_model
from pccx import _lib, models
from tests import synth

K, k, d, L = synth.MODEL_CFG

ENC_META, ENC_SA_B1, ENC_SA_W = 0, 16, 16 + 64 + 128 + 128 + 256 + 512 + 16
ENC_PN_STREAM = ENC_SA_W + 40 * 256


def _pow2(v):
    m, _ = np.frexp(np.asarray(v, dtype=np.float64))
    return bool(np.all(m == 0.5))


def _tau(W):
    return 2.0 ** np.floor(np.log2(16384.0 / np.abs(W).max()))


def _planes(blob, off, nfrag):
    """fragments [nfrag/2][2 planes][64 lanes][8 halves] -> (hi, lo) float64 arrays [nfrag/2][64][8]"""
    a = blob[off:off + nfrag * 256].view(np.uint32).view(np.float16).reshape(nfrag // 2, 2, 64, 8).astype(np.float64)
    return a[:, 0], a[:, 1]


@pytest.fixture(scope="module", params=[1.0, 5.0])
def packed(request):
    ae = models.AE(K, k, d, L)
    sd = ref_model.seeded_state_dict(ae, synth.AE_SEED, last_gain=synth.AE_LAST_GAIN)
    sd = {n: (v * request.param if n.endswith("bias") else v) for n, v in sd.items()}
    ae.load_state_dict(sd)
    lib = _lib.load()
    enc = models._pack("pccx_pack_ae_encoder_h2", lib.pccx_ae_encoder_h2_blob_floats(), ae._enc_tensors(), [d]).numpy()
    dec = models._pack("pccx_pack_ae_decoder_h2", lib.pccx_ae_decoder_h2_blob_floats(k), ae._dec_tensors(), [k, d]).numpy()
    return ae, enc, dec


def test_scales_are_powers_of_two(packed):
    _, enc, dec = packed
    assert _pow2(enc[:8]) and _pow2(dec[:7])
    assert (enc[:8] > 0).all() and (dec[:7] > 0).all()


def test_weight_planes_reproduce_the_weights(packed):
    ae, enc, _ = packed
    sd = ae.state_dict()
    # SetAbstraction conv1 (64 x 32): fragments [kt32 = 0][mt 0..3][plane]; lane (m, kg), slot j <-> channel 16 (j >> 2) + 4 kg + (j & 3)
    W1 = sd["sa.conv1.weight"].reshape(64, 32).numpy().astype(np.float64)
    hi, lo = _planes(enc, ENC_SA_W, 8)
    tau = _tau(W1)
    lane = np.arange(64)
    m, kg = lane & 15, lane >> 4
    j = np.arange(8)
    ch = 16 * (j[None, :] >> 2) + 4 * kg[:, None] + (j[None, :] & 3)
    for mt in range(4):
        want = W1[16 * mt + m[:, None], ch] * tau
        got = hi[mt] + lo[mt]
        assert np.all(np.abs(got - want) <= np.maximum(2.0 ** -22 * np.abs(want), 2.0 ** -25))
        assert np.all(np.abs(hi[mt]) <= 16384.0)
    # PointNet layer 1 (256 x 128) in the stream: after L0's 5 x 8 x 2 fragments, [kt32 0..3][mt 0..15][plane]
    Wp = sd["pn.mlp_Modules.1.0.weight"].reshape(256, 128).numpy().astype(np.float64)
    hi, lo = _planes(enc, ENC_PN_STREAM + 80 * 256, 128)
    tau = _tau(Wp)
    for t in range(4):
        for mt in range(16):
            want = Wp[16 * mt + m[:, None], 32 * t + ch] * tau
            got = hi[t * 16 + mt] + lo[t * 16 + mt]
            assert np.all(np.abs(got - want) <= np.maximum(2.0 ** -22 * np.abs(want), 2.0 ** -25))


def _relu(x):
    return np.maximum(x, 0.0)


def test_encoder_bounds_hold_on_random_and_corner_inputs(packed):
    """sigma of every split, recovered from the packed multipliers (rho_l = sigma_l / (sigma_{l-1} tau_{l-1})), times the layer's
    float64 activations stays <= 2^15 for coordinates of magnitude <= 1, any patch scale s <= 1."""
    ae, enc, _ = packed
    sd = {n: v.numpy().astype(np.float64) for n, v in ae.state_dict().items()}
    W0, b0 = sd["sa.conv0.weight"].reshape(32, 3), sd["sa.conv0.bias"]
    W1, b1 = sd["sa.conv1.weight"].reshape(64, 32), sd["sa.conv1.bias"]
    W2, b2 = sd["sa.conv2.weight"].reshape(128, 64), sd["sa.conv2.bias"]
    P = [(sd[f"pn.mlp_Modules.{i}.0.weight"].reshape(sd[f"pn.mlp_Modules.{i}.0.weight"].shape[0], -1), sd[f"pn.mlp_Modules.{i}.0.bias"]) for i in range(4)]
    rho0, rho1, inv2, rho_in, rp1, rp2, rp3, inv_out = [float(v) for v in enc[:8]]
    sig0 = rho0
    sig1 = rho1 * sig0 * _tau(W1)
    assert np.isclose(inv2, 1.0 / (sig1 * _tau(W2)), rtol=0, atol=0)
    sig_in = rho_in
    sp0 = rp1 * sig_in * _tau(P[0][0])
    sp1 = rp2 * sp0 * _tau(P[1][0])
    sp2 = rp3 * sp1 * _tau(P[2][0])
    assert np.isclose(inv_out, 1.0 / (sp2 * _tau(P[3][0])), rtol=0, atol=0)
    rng = np.random.default_rng(5)
    worst = np.zeros(6)
    for trial in range(40):
        s = 1.0 if trial % 2 == 0 else 2.0 ** -int(rng.integers(0, 12))
        n = 64
        if trial < 20:
            xi, xj = rng.uniform(-1, 1, (n, 3)), rng.uniform(-1, 1, (n, 3))
        else:                                                     # corners: differences of +-2 aligned with rows of conv0
            r = rng.integers(0, 32, n)
            xi, xj = -np.sign(W0[r]), np.sign(W0[r])
        rel = xj - xi
        h0 = _relu(rel @ W0.T + b0 * s)
        y1 = _relu(h0 @ W1.T + b1 * s)
        feat = _relu(y1 @ W2.T + b2 * s)                          # (the neighbour max of such rows is bounded alike)
        xyz = rng.uniform(-1, 1, (n, 3)) if trial < 20 else np.sign(P[0][0][rng.integers(0, 128, n), :3])
        z0 = _relu(np.concatenate([xyz, feat], 1) @ P[0][0].T + P[0][1] * s)
        z1 = _relu(z0 @ P[1][0].T + P[1][1] * s)
        z2 = _relu(z1 @ P[2][0].T + P[2][1] * s)
        worst = np.maximum(worst, [h0.max() * sig0, y1.max() * sig1, max(feat.max(), 1.0) * sig_in, z0.max() * sp0, z1.max() * sp1, z2.max() * sp2])
    assert (worst <= 32768.0).all(), worst
    assert (worst >= 2.0 ** 3).all(), worst                     # and the scales are not absurdly loose (lo pieces stay normal)


def test_decoder_bounds_hold_on_random_and_corner_inputs(packed):
    ae, _, dec = packed
    sd = {n: v.numpy().astype(np.float64) for n, v in ae.state_dict().items()}
    Wg, bg = sd["inv_pool.4.weight"], sd["inv_pool.4.bias"]
    M = [(sd[f"inv_mlp.mlp_Modules.{i}.0.weight"].reshape(sd[f"inv_mlp.mlp_Modules.{i}.0.weight"].shape[0], -1), sd[f"inv_mlp.mlp_Modules.{i}.0.bias"]) for i in range(4)]
    sig_h, rho0, sig_q, rho1, rho2, rho3, inv_out = [float(v) for v in dec[:7]]
    assert sig_h == 32768.0
    sig0 = rho0 * sig_h * _tau(Wg)
    assert sig0 == sig_q
    sig1 = rho1 * sig0 * _tau(M[0][0])
    sig2 = rho2 * sig1 * _tau(M[1][0])
    sig3 = rho3 * sig2 * _tau(M[2][0])
    assert inv_out == 1.0 / (sig3 * _tau(M[3][0]))
    rng = np.random.default_rng(6)
    worst = np.zeros(4)
    for trial in range(12):
        s = 1.0 if trial % 2 == 0 else 2.0 ** -int(rng.integers(0, 12))
        n = 8
        h2 = rng.uniform(0, 1, (n, 1024)) if trial < 8 else (Wg[rng.integers(0, Wg.shape[0], n)] > 0).astype(np.float64)
        q = rng.uniform(-1, 1, (n, d))
        g = _relu(h2 @ Wg.T + bg * s).reshape(n, 128, k)         # AE.py:49: channel c, point p = row c*k + p
        for p in rng.integers(0, k, 4):
            y = np.concatenate([g[:, :, p], q], 1)
            m0 = _relu(y @ M[0][0].T + M[0][1] * s)
            m1 = _relu(m0 @ M[1][0].T + M[1][1] * s)
            m2 = _relu(m1 @ M[2][0].T + M[2][1] * s)
            worst = np.maximum(worst, [max(y.max(), 1.0) * sig0, m0.max() * sig1, m1.max() * sig2, m2.max() * sig3])
    assert (worst <= 32768.0).all(), worst
    assert (worst >= 2.0 ** 3).all(), worst


def test_pack_rejects_non_finite_weights():
    ae = models.AE(K, k, d, L)
    sd = ref_model.seeded_state_dict(ae, synth.AE_SEED, last_gain=synth.AE_LAST_GAIN)
    sd["pn.mlp_Modules.1.0.weight"] = sd["pn.mlp_Modules.1.0.weight"].clone()
    sd["pn.mlp_Modules.1.0.weight"][3, 5] = float("inf")
    ae.load_state_dict(sd)
    with pytest.raises(_lib.PccxError):
        models._pack("pccx_pack_ae_encoder_h2", _lib.load().pccx_ae_encoder_h2_blob_floats(), ae._enc_tensors(), [d])


def test_two_piece_split_arithmetic_property():
    """The arithmetic behind the mode, in numpy (IEEE binary16, round to nearest even -- what v_cvt_pk_f16_f32 does): for operands
    scaled into fp16's normal range, hi = rn16(x), lo = rn16(x - hi) reproduce x to 2^-22 |x|, the residual x - hi is exact in fp32,
    and hi*hi + hi*lo + lo*hi (each product exact in fp32: 11 x 11 bits) differs from x*y by at most 3 * 2^-22 |x y|."""
    rng = np.random.default_rng(9)
    mag = lambda emax: rng.uniform(0.25, 2.0, 200000) * np.exp2(rng.integers(0, emax, 200000)) * rng.choice([-1.0, 1.0], 200000)
    x, y = mag(14).astype(np.float32), mag(13).astype(np.float32)          # |x| >= 2^-2: the lo piece is a normal fp16 number

    def split(v):
        hi = v.astype(np.float16)
        assert np.isfinite(hi).all()
        r = v - hi.astype(np.float32)                                   # fp32 subtraction
        assert np.array_equal(r.astype(np.float64), v.astype(np.float64) - hi.astype(np.float64))     # ... is exact
        lo = r.astype(np.float16)
        return hi.astype(np.float64), lo.astype(np.float64)

    xh, xl = split(x)
    yh, yl = split(y)
    x64, y64 = x.astype(np.float64), y.astype(np.float64)
    assert (np.abs(x64 - xh - xl) <= 2.0 ** -22 * np.abs(x64)).all()
    for a, b in ((xh, yh), (xh, yl), (xl, yh)):
        p = a * b
        assert np.array_equal(p, p.astype(np.float32).astype(np.float64))       # every partial product is an fp32 number
    approx = xh * yh + xh * yl + xl * yh
    assert (np.abs(approx - x64 * y64) <= 3 * 2.0 ** -22 * np.abs(x64 * y64)).all()
