"""CPU: the C-ABI library loads and exports every symbol include/pccx.h declares (no compute)."""
import ctypes
import os
import re

import pytest

from pccx import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "pccx.h")).read()
    return sorted(set(re.findall(r"PCCX_API\s+[\w\s\*]+?\b(pccx_\w+)\s*\(", txt)))


def test_header_declares_symbols():
    names = _declared()
    assert "pccx_fps" in names and "pccx_octree_encode" in names and len(names) >= 10


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        from pccx import build
        build.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared():
        assert hasattr(lib, name), f"libpccx.so does not export {name}"


def test_python_binding_covers_header():
    assert sorted(_lib.declared_symbols()) == _declared()


def test_ops_fail_loudly_without_gpu_tensor():
    import torch
    from pccx import ops
    with pytest.raises(_lib.PccxError):
        ops.normalize(torch.zeros(1, 8, 3))     # CPU tensor: no fallback


def test_version_and_error_string():
    lib = _lib.load()
    assert lib.pccx_version() >= 100
    # argument validation happens on the host, before any HIP call
    rc = lib.pccx_octree_encode(None, 1, 64, 8192, 0.25, None, None, None, None, None, None)
    assert rc == -1 and b"null pointer" in lib.pccx_last_error()


def test_model_limits_name_the_reference_flags():
    """compress.py:30-34 accepts any --K / --d / --L.  --d and --L are unrestricted here too (widths beyond the fused kernels' 16 /
    d * L <= 128 take the generic layers, tests/test_gpu_model.py); --K must be a multiple of 16 in 16..1024 (the reference's own
    octree rate table, pn_kit.py:17-23, stops at 1024): refused loudly, naming the flag, before any GPU work."""
    import pytest as _pt
    from pccx import models
    for K, k, d, L in ((250, 125, 16, 7), (2048, 1024, 16, 7), (256, 128, 0, 7)):
        with _pt.raises(_lib.PccxError, match="--K"):
            models.AE(K, k, d, L)
    assert models.AE(256, 128, 32, 7).fused_d is False and models.AE(512, 256, 8, 16).fused_d is True
    assert models.ConditionalProbabilityModel(9, 16).fused_ok(64) is False and models.ConditionalProbabilityModel(7, 16).fused_ok(64) is True
