# round-3 variants of the fused encoder (exec'd by enc_variants.py)
VARIANTS["nnan"] = (PRODUCT, ["-fno-honor-nans"])
