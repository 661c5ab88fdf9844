"""pccx -- host side of the MI355X compress / decompress path.

DEFAULT_MATMUL selects how the three MLP transforms form their fp32 products when a caller does not say
(codec.Codec(matmul=...), models.AE.encode/decode): "f32" (exact-fp32 MFMA) or "bf16x3" (three-way bf16 split on the
bf16 matrix cores, fp32 accumulate).  The default is "bf16x3": every oracle / golden parity test of tests/test_gpu_model.py,
tests/test_gpu_pipeline.py, tests/test_boundary.py and the smoke run pass in BOTH modes at the same tolerances (the tests are
parametrised over the mode), and it is 1.6x faster end to end.  Overridable with the environment variable PCCX_MATMUL.

"f16x2" (round 3): two fp16 pieces per operand (22-23 significant bits, the operand precision of "3xTF32"), three products per
fp32 product instead of six, with exact power-of-two operand scales from rigorous layer bounds (csrc/pack_h2.hip).  It exists for
the two fused AE transforms (encoder_fused_h2.hip, decoder_h2.hip); every other kernel runs its bf16x3 form in this mode.
"""
import os

MATMUL_MODES = ("f32", "bf16x3", "f16x2")
DEFAULT_MATMUL = os.environ.get("PCCX_MATMUL", "bf16x3")
if DEFAULT_MATMUL not in MATMUL_MODES:
    raise ValueError(f"PCCX_MATMUL={DEFAULT_MATMUL!r}: expected one of {MATMUL_MODES}")
