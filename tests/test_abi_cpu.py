"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol
include/ptr.h declares, and refuses to run without a GPU (no CPU fallback)."""
import os
import re

import pytest

import ptr_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "ptr.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ptr_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = ptr_amd.lib()
    declared = _declared_symbols()
    assert len(declared) >= 16
    for s in declared:
        assert hasattr(L, s), s
    assert sorted(ptr_amd.SYMBOLS) == declared
    assert L.ptr_abi_version() == 1


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(ptr_amd.PtrError) as e:
        ptr_amd.Renderer(0)
    assert e.value.code == ptr_amd.ERR_NO_DEVICE
    assert "no CPU fallback" in str(e.value)


def test_product_never_touches_the_oracle():
    # the oracle is test infrastructure: nothing under path-tracer_amd/ may mention it
    bad = []
    for dp, _, fns in os.walk(os.path.join(ROOT, "path-tracer_amd")):
        for fn in fns:
            if fn.endswith((".so", ".o", ".node", ".pyc")):
                continue
            txt = open(os.path.join(dp, fn), errors="ignore").read()
            if "oracle" in txt.lower():
                bad.append(os.path.join(dp, fn))
    assert not bad, bad
