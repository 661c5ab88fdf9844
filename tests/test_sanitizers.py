"""CPU: the sanitizer target of SURVEY section 5.  `make -C oracle asan-test` builds the C oracle and the host side of the C ABI (the weight
packers + error plumbing: pure host code in csrc/pack.hip, pack_h2.hip, error.hip, compiled as plain C++) with AddressSanitizer + UBSan
and runs the CPU tests that drive them (oracle vs golden fixtures, f16x2 packing, split property) against those builds in a child
process.  GPU code cannot be sanitized on this pool (no GPU ASan / xnack+), so this covers the native code that runs on the host."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_and_host_packers_are_clean_under_asan_and_ubsan():
    gcc = shutil.which("gcc")
    if not gcc or not os.path.exists(subprocess.run([gcc, "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()):
        pytest.skip("no gcc / libasan on this machine")
    env = {k: v for k, v in os.environ.items() if not k.startswith("PCCX_")}
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan-test"], capture_output=True, text=True, env=env, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert " passed" in tail and "ERROR: AddressSanitizer" not in tail and "runtime error" not in tail, tail
