"""CPU test over the COMPILED kernels that issue loads from inline assembly (tools/asm_load_lint.py): between such a load and the
wait that covers it, no instruction may read or overwrite its destination registers.  The source arranges that; a register copy or
a reuse inserted by the compiler would break it silently, and only on cold caches (it did once: decoder_h2.hip, a build with a
larger ring chunk had the tail of the kernel reuse the registers of the last, unused operand loads while they were in flight).
hipcc cross-compiles for gfx950 without a GPU, so this runs in the CPU suite."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_no_instruction_touches_a_pending_asm_load(tmp_path):
    import subprocess

    import asm_load_lint as L
    total = 0
    for f, prefixes in L.TARGETS.items():
        out = str(tmp_path / (f + ".s"))
        subprocess.run(["/opt/rocm/bin/hipcc"] + L.FLAGS + [os.path.join(L.CSRC, f), "-o", out], check=True, stderr=subprocess.DEVNULL)
        found = 0
        for name, body in L.kernels(out, prefixes):
            n_asm, problems = L.scan(name, body)
            assert n_asm > 0, f"{name}: no asm loads found (the lint's parser no longer matches the compiler's output?)"
            assert not problems, f"{f}: {name}: " + "; ".join(f"#{i} {s} <- {w}" for i, s, w in problems[:5])
            found += 1
        assert found >= 1, f"{f}: none of {prefixes} found"
        total += found
    assert total >= 4 + 16          # + the planes kernels (12 GEMM instantiations, 4 chains, and more as they are added)


def test_lint_flags_a_premature_use():
    """the scanner itself, on a hand-written fragment: a read of an asm-loaded register before the wait is reported, after it is not"""
    import asm_load_lint as L
    frag = ["\t;;#ASMSTART", "\tglobal_load_dwordx4 v[10:13], v[2:3], off", "\t;;#ASMEND",
            "\tv_add_f32_e32 v20, v11, v21", "\ts_waitcnt vmcnt(0)", "\tv_add_f32_e32 v22, v12, v21"]
    n, problems = L.scan("k", frag)
    assert n == 1 and len(problems) == 1 and "v_add_f32_e32 v20" in problems[0][1]
    frag2 = ["\t;;#ASMSTART", "\tds_read_b128 v[10:13], v1 offset:0", "\t;;#ASMEND", "\t;;#ASMSTART", "\tds_read_b128 v[14:17], v1 offset:1024", "\t;;#ASMEND",
             "\t;;#ASMSTART", "\ts_waitcnt lgkmcnt(1)", "\t;;#ASMEND", "\tv_mov_b32_e32 v30, v10", "\tv_mov_b32_e32 v31, v14"]
    n, problems = L.scan("k", frag2)
    assert n == 2 and len(problems) == 1 and "v31" in problems[0][1]
