"""The C-ABI library loads and exports every symbol include/edynhip.h declares; without a GPU it fails loudly
(no CPU fallback). No compute calls here."""
import ctypes
import os
import re
import pytest
from conftest import ROOT, HAVE_GPU


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "edynhip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(edynhip_[a-z_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from edyn_amd import _capi
    lib = ctypes.CDLL(_capi.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in edynhip.h but not exported"
    assert sorted(_capi.SYMBOLS) == syms, "edyn_amd/_capi.py SYMBOLS out of sync with include/edynhip.h"
    assert lib.edynhip_abi_version() == 15


def test_record_layouts_match():
    from edyn_amd import _capi
    from oracle import binding as ob
    assert _capi.MANIFOLD_DTYPE.itemsize == 16 + 4 * 80 == ob.MANIFOLD_DTYPE.itemsize
    assert _capi.MANIFOLD_DTYPE == ob.MANIFOLD_DTYPE


@pytest.mark.skipif(HAVE_GPU, reason="a GPU is present: creation succeeds")
def test_create_fails_loudly_without_gpu():
    import edyn_amd
    w = edyn_amd.World()
    with pytest.raises(edyn_amd.EdynHipError) as ei:
        w.attach(16)
    assert ei.value.code == -2 and "no CPU fallback" in str(ei.value)


def test_product_never_imports_oracle():
    """The product package must not reference anything under oracle/."""
    pkg = os.path.join(ROOT, "edyn_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle" not in src.lower(), f"{f} mentions the oracle"
