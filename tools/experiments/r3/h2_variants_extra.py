# diagnostic builds of the f16x2 encoder (exec'd by h2_variants.py): garbage results, timing only
_NODMA = ("mfma_chain.h", "__builtin_amdgcn_global_load_lds((const void *)(src + voff), (lds_u32 *)(uintptr_t)dst, 16, 0, 0);", "(void)src; (void)dst;")
_NOBAR = ("mfma_chain.h", 'asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");\n        __syncthreads();\n        issue_ahead(c);',
          'asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");\n        issue_ahead(c);')
_ONLYPN = ("encoder_fused_h2.hip", "if (unit >= units) break;", "break;")
_NOSPLIT = [("encoder_fused_h2.hip", '#ifndef FH_CHUNK', FAKE_DEF + '#ifndef FH_CHUNK'), ("encoder_fused_h2.hip",) + FAKE_SPLIT]
VARIANTS["nosplit"] = ([], _NOSPLIT + [("decoder_h2.hip", '#ifndef DEC_GROUP', FAKE_DEF + '#ifndef DEC_GROUP'), ("decoder_h2.hip",) + FAKE_SPLIT])
# (builds that removed the LDS-DMA itself faulted on the box -- the ring is then never written and the compiler is free to treat its reads as undefined --
#  and are not kept; the phases are separated instead)
_ONLYSA = ("encoder_fused_h2.hip", "#define FH_DENSE(KT, MT, in, acc) dense_h2_rd<KT, MT, 1>(rd, f, in, acc)", "#define FH_DENSE(KT, MT, in, acc) (void)0")
VARIANTS["onlypn"] = ([], [_ONLYPN])
VARIANTS["onlysa"] = ([], [_ONLYSA])
VARIANTS["onlypn_nosplit"] = ([], [_ONLYPN] + _NOSPLIT)
VARIANTS["onlysa_nosplit"] = ([], [_ONLYSA] + _NOSPLIT)
VARIANTS["onlypn_nobar"] = ([], [_ONLYPN, _NOBAR])
# DIAGNOSTIC: only every second weight fragment is actually read from the LDS ring (the others keep stale registers): is the PointNet phase
# bound by LDS bandwidth?
_HALFREADS = ("mfma_chain.h", 'asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[fi % D])', 'if (!(fi & 1)) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[fi % D])')
_NOREADS = ("mfma_chain.h", 'asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[fi % D])', 'if (fi < 8) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[fi % D])')
VARIANTS["onlypn_halfreads"] = ([], [_ONLYPN, _HALFREADS])
VARIANTS["onlypn_noreads"] = ([], [_ONLYPN, _NOREADS])
VARIANTS["onlypn_noreads_nosplit"] = ([], [_ONLYPN, _NOREADS] + _NOSPLIT)
