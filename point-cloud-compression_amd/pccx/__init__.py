"""pccx -- host side of the MI355X compress / decompress path.

DEFAULT_MATMUL selects how the three MLP transforms form their fp32 products when a caller does not say
(codec.Codec(matmul=...), models.AE.encode/decode): "f32" (exact-fp32 MFMA) or "bf16x3" (three-way bf16 split on the
bf16 matrix cores, fp32 accumulate).  The default is "bf16x3": every oracle / golden parity test of tests/test_gpu_model.py,
tests/test_gpu_pipeline.py, tests/test_boundary.py and the smoke run pass in BOTH modes at the same tolerances (the tests are
parametrised over the mode), and it is 1.6x faster end to end.  Overridable with the environment variable PCCX_MATMUL.
"""
import os

DEFAULT_MATMUL = os.environ.get("PCCX_MATMUL", "bf16x3")
if DEFAULT_MATMUL not in ("f32", "bf16x3"):
    raise ValueError(f"PCCX_MATMUL={DEFAULT_MATMUL!r}: expected 'f32' or 'bf16x3'")
