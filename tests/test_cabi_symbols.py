"""CPU: the C-ABI library builds/loads and exports every symbol include/totsu_f32hip.h declares; the product
path fails loudly without a GPU (no CPU fallback)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "totsu_f32hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = set(re.findall(r"\b(thip_[a-z0-9_]+)\s*\(", txt))
    names -= {"thip_allreduce_fn"}
    return sorted(names)


def test_library_exports_every_declared_symbol():
    from totsu_amd import _lib
    lib = _lib.load()
    decl = _declared()
    assert len(decl) >= 50
    for name in decl:
        assert hasattr(lib, name), "missing symbol %s" % name
    # and the Python binding knows every one of them
    assert set(decl) == set(_lib.PROTOTYPES), set(decl) ^ set(_lib.PROTOTYPES)


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from totsu_amd import _lib
    with pytest.raises(_lib.ThipError) as e:
        _lib.init(0)
    assert e.value.code == _lib.E_NOGPU
    import numpy as np
    from totsu_amd import F32HIP
    with pytest.raises(_lib.ThipError):
        F32HIP.Sl.new_ref(np.zeros(4, dtype=np.float32))


def test_product_never_imports_oracle():
    # the oracle is test infrastructure: nothing under totsu_amd/ may reference it
    pkg = os.path.join(ROOT, "totsu_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f
                assert "libtotsu_oracle" not in src, f


def test_authored_rust_binding_declares_every_symbol():
    # rust/totsu_f32hip cannot be compiled here (no cargo); at least its extern block stays one-to-one with the header
    src = open(os.path.join(ROOT, "rust", "totsu_f32hip", "src", "ffi.rs")).read()
    rust = set(re.findall(r"pub fn (thip_[a-z0-9_]+)\s*\(", src))
    assert rust == set(_declared()), rust ^ set(_declared())
