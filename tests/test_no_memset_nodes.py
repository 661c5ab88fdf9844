"""CPU lint: no hipMemsetAsync / hipMemcpyAsync may enter a code path that is captured into a hipGraph.

Round 3 found that memset NODES between the kernel nodes of the captured training step gave non-finite gradients on a replay issued
after an idle gap; buffers are cleared by a kernel since (common.h: pccx_zero_async).  Nothing in the sources
stops a new hipMemsetAsync from being added, so this test does: in csrc/ the call may appear only inside weight-PACKING entry points
(pccx_pack_*: run once at set-up, never inside a captured step) and in the experiment-only branch of pccx_zero_async."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "point-cloud-compression_amd", "csrc")
FUNC = re.compile(r'^(?:extern "C" )?(?:static )?(?:inline )?[\w\s\*]*?\b(pccx_\w+)\s*\(')


def _calls(path, what):
    """(line number, enclosing pccx_* function) of every call of `what` in a source file, comments stripped"""
    out, cur, guarded = [], None, False
    for no, line in enumerate(open(path), 1):
        code = line.split("//")[0]
        m = FUNC.match(code)
        if m:
            cur = m.group(1)
        if code.startswith("#ifdef PCCX_ZERO_WITH_MEMSET"):
            guarded = True
        elif code.startswith("#endif"):
            guarded = False
        if what + "(" in code and not guarded:
            out.append((no, cur))
    return out


def test_memset_and_memcpy_only_in_packing_entry_points():
    bad = []
    for path in sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h"))):
        for what in ("hipMemsetAsync", "hipMemset", "hipMemcpyAsync", "hipMemcpy"):
            for no, fn in _calls(path, what):
                if not (fn or "").startswith("pccx_pack"):
                    bad.append(f"{os.path.basename(path)}:{no} {what} in {fn}")
    assert not bad, "memset / memcpy calls outside the packing entry points (they would become graph nodes in a captured step): " + "; ".join(bad)


def test_the_lint_sees_the_experiment_guard_and_the_packers():
    # the guarded experiment branch of pccx_zero_async is the only hipMemsetAsync of common.h; the packers keep theirs
    assert _calls(os.path.join(CSRC, "common.h"), "hipMemsetAsync") == []
    assert "hipMemsetAsync(" in open(os.path.join(CSRC, "common.h")).read()
    assert [fn for _, fn in _calls(os.path.join(CSRC, "decoder.hip"), "hipMemsetAsync")] == ["pccx_pack_ae_decoder_b3", "pccx_pack_pn_b3"]
