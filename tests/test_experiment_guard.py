"""CPU: the experiment harness refuses diagnostic variants that delete a producer and keep its consumers (round-3 review: a build of
tools/experiments/r3/h2_variants.py that removed the LDS-DMA left the weight ring unwritten and ended in a GPU memory fault)."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load():
    spec = importlib.util.spec_from_file_location("h2_variants", os.path.join(ROOT, "tools", "experiments", "r3", "h2_variants.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_no_dma_variant_is_refused_and_the_kept_variants_pass():
    m = _load()
    for tag, (flags, patches) in m.VARIANTS.items():
        m.check_variant(tag, patches)                                    # everything that is kept is admissible
    nodma = ("mfma_chain.h", "__builtin_amdgcn_global_load_lds((const void *)(src + voff), (lds_u32 *)(uintptr_t)dst, 16, 0, 0);", "(void)src; (void)dst;")
    with pytest.raises(SystemExit, match="producer"):
        m.check_variant("nodma", [nodma])
    with pytest.raises(SystemExit, match="producer"):
        m.check_variant("nodma_onlypn", [nodma, ("encoder_fused_h2.hip", "if (unit >= units) break;", "break;")])
