"""pccx -- host side of the MI355X compress / decompress path.

DEFAULT_MATMUL selects how the three MLP transforms form their fp32 products when a caller does not say
(codec.Codec(matmul=...), models.AE.encode/decode): "f32" (exact-fp32 MFMA) or "bf16x3" (three-way bf16 split on the
bf16 matrix cores, fp32 accumulate).  Overridable with the environment variable PCCX_MATMUL.
"""
import os

DEFAULT_MATMUL = os.environ.get("PCCX_MATMUL", "f32")
if DEFAULT_MATMUL not in ("f32", "bf16x3"):
    raise ValueError(f"PCCX_MATMUL={DEFAULT_MATMUL!r}: expected 'f32' or 'bf16x3'")
