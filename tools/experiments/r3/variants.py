# round-3 variants of the fused encoder (exec'd by enc_variants.py)
VARIANTS["nnan"] = (PRODUCT, ["-fno-honor-nans"])
KN = os.path.join(R3, "enc_knobs.hip")
for tag, fl in {"k_base": [], "k_onlysa": ["-DK_NO_KNN", "-DK_NO_PN"], "k_onlypn": ["-DK_NO_KNN", "-DK_NO_SA"],
                "k_onlysa_nosplit": ["-DK_NO_KNN", "-DK_NO_PN", "-DK_NOSPLIT"], "k_onlysa_noldsw": ["-DK_NO_KNN", "-DK_NO_PN", "-DK_SA_NOLDSW"],
                "k_onlysa_nosplit_noldsw": ["-DK_NO_KNN", "-DK_NO_PN", "-DK_NOSPLIT", "-DK_SA_NOLDSW"],
                "k_onlypn_nosplit": ["-DK_NO_KNN", "-DK_NO_SA", "-DK_NOSPLIT"], "k_onlypn_nodma": ["-DK_NO_KNN", "-DK_NO_SA", "-DK_NODMA"],
                "k_onlypn_noldsw": ["-DK_NO_KNN", "-DK_NO_SA", "-DK_PN_NOLDSW"],
                "k_onlypn_nodma_nobar": ["-DK_NO_KNN", "-DK_NO_SA", "-DK_NODMA", "-DK_NOBAR"],
                "k_onlypn_nodma_nobar_noldsw": ["-DK_NO_KNN", "-DK_NO_SA", "-DK_NODMA", "-DK_NOBAR", "-DK_PN_NOLDSW"],
                "k_onlypn_floor": ["-DK_NO_KNN", "-DK_NO_SA", "-DK_NODMA", "-DK_NOSPLIT", "-DK_PN_NOLDSW", "-DK_NOBAR"]}.items():
    VARIANTS[tag] = (KN, ["-fno-honor-nans"] + fl)
VARIANTS["ch32"] = (PRODUCT, ["-fno-honor-nans", "-DFU_CHUNK=32"])
VARIANTS["ch16"] = (PRODUCT, ["-fno-honor-nans", "-DFU_CHUNK=16"])
VARIANTS["nb3"] = (PRODUCT, ["-fno-honor-nans", "-DFU_NB=3"])
VARIANTS["ch16nb4"] = (PRODUCT, ["-fno-honor-nans", "-DFU_CHUNK=16", "-DFU_NB=4"])
VARIANTS["ch8nb8"] = (PRODUCT, ["-fno-honor-nans", "-DFU_CHUNK=8", "-DFU_NB=8"])
VARIANTS["round2"] = (KN, [])                     # the round-2 kernel as it was (in-kernel selection, canonicalising v_max)
VARIANTS["product"] = (PRODUCT, ["-fno-honor-nans"])   # what libpccx.so builds
VARIANTS["cachepl"] = (PRODUCT, ["-fno-honor-nans", "-DFU_CACHE_PLANES"])
VARIANTS["pwait"] = (PRODUCT, ["-fno-honor-nans", "-DFU_COUNTED_WAITS"])       # counted lgkmcnt waits on the ring reads (inline-asm ds_read)
