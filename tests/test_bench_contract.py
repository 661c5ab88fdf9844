"""CPU tests of bench.py's bookkeeping (no GPU): the roofline of the dominant kernel in each arithmetic mode -- peak, fraction, which
kernels the timed stage covers, where the traffic figure comes from -- computed from a synthetic stage table."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def _stages(enc_ms, dec_ms):
    # name -> (mean ms per launch, launches) as bench_ipdae builds it from ops.StageTimer
    return {"sa_pn_forward": (enc_ms, 20), "patch_knn16": (2.5, 20), "ae_decode": (dec_ms, 20), "knn_patches": (1.3, 20), "prob": (0.5, 40)}


def test_roofline_per_mode():
    P, B, steps = 65536, 1024, 20
    flop = (bench.FLOP_SA + bench.FLOP_PN) * P
    for mode, peak in (("bf16x3", 16 * 157.3 / 6), ("f16x2", 16 * 157.3 / 3)):
        rf, per_step, tfl = bench.roofline_of(_stages(27.4, 6.5), steps, P, mode, B)
        assert rf["kernel"] == "sa_pn_forward" and rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s"
        assert abs(rf["peak"] - peak) < 1e-6
        assert abs(rf["achieved"] - flop / 27.4e-3 / 1e12) < 1e-6 and abs(rf["frac"] - rf["achieved"] / peak) < 1e-12
        # the timed stage is ONE kernel (the neighbour tables are a stage of their own, "patch_knn16", with no matrix work)
        assert rf["stage_kernels"] == ["sa_pn_forward_h2_kernel" if mode == "f16x2" else "sa_pn_forward_b3_kernel"]
        assert abs(per_step["patch_knn16"] - 2.5) < 1e-9
        assert rf["flop_per_launch"] == flop and rf["arithmetic"] == mode
        assert abs(per_step["sa_pn_forward"] - 27.4) < 1e-9 and abs(per_step["prob"] - 1.0) < 1e-9
        if rf["traffic"] is not None:                                    # read from a committed PMC pass, never measured in the run
            assert "profiles/" in rf["traffic_source"] and rf["traffic"] > 0
    rf, _, _ = bench.roofline_of({"sa_forward": (39.0, 20), "pn_forward": (44.5, 20), "ae_decode": (19.8, 20)}, steps, P, "f32", B)
    assert rf["kernel"] == "pn_forward" and abs(rf["peak"] - 157.3) < 1e-9 and rf["stage_kernels"] == ["pn_forward_kernel"]


def test_flop_counts_follow_the_layer_shapes():
    K, k, d = 256, 128, 16
    assert bench.FLOP_SA == K * 16 * (3 * 32 + 32 * 64 + 64 * 128) * 2                       # AE.py:16, 16 neighbours per point
    assert bench.FLOP_PN == K * (131 * 128 + 128 * 256 + 256 * 512 + 512 * d) * 2            # AE.py:17
    assert bench.FLOP_DEC == (d * 256 + 256 * 1024 + 1024 * k * 128) * 2 + k * (144 * 128 + 128 * 64 + 64 * 32 + 32 * 3) * 2   # AE.py:19-27
    assert set(bench.MODE_DTYPE) == {"f32", "bf16x3", "f16x2"}
