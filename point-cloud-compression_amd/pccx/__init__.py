"""pccx -- host side of the MI355X compress / decompress path.

DEFAULT_MATMUL selects how the three MLP transforms form their fp32 products when a caller does not say
(codec.Codec(matmul=...), models.AE.encode/decode):
  "f32"     exact-fp32 MFMA (v_mfma_f32_16x16x4_f32): bit-for-bit a k-ordered fmaf chain;
  "bf16x3"  every fp32 operand split EXACTLY into three bf16 pieces, six products per fp32 product on the bf16 matrix cores, fp32
            accumulate (the default of round 2);
  "f16x2"   every fp32 operand split into two fp16 pieces (22-23 significant bits, the operand precision of "3xTF32"), three products
            per fp32 product on the fp16 matrix cores, fp32 accumulate, with exact power-of-two operand scales from rigorous layer
            bounds (csrc/pack_h2.hip) and one per patch from the data.  It exists for the two fused AE transforms
            (encoder_fused_h2.hip, decoder_h2.hip); every other kernel runs its bf16x3 form in this mode.
The default is "f16x2" (round 3): every oracle / golden parity test of tests/test_gpu_model.py, tests/test_gpu_pipeline.py,
tests/test_boundary.py and the smoke run pass in ALL THREE modes at the same tolerances (the tests are parametrised over the mode);
against the exact-fp32 kernels its results differ no more than bf16x3's do (both at the level of an fp32 summation reorder: the fp32
accumulation, not the operand representation, dominates), no symbol of 524 288 differs, and it is 1.5x faster end to end than bf16x3.
Overridable with the environment variable PCCX_MATMUL.
"""
import os

MATMUL_MODES = ("f32", "bf16x3", "f16x2")
DEFAULT_MATMUL = os.environ.get("PCCX_MATMUL", "f16x2")
if DEFAULT_MATMUL not in MATMUL_MODES:
    raise ValueError(f"PCCX_MATMUL={DEFAULT_MATMUL!r}: expected one of {MATMUL_MODES}")
