"""CPU: the C-ABI library builds/loads and exports every symbol include/totsu_f32hip.h declares; the product
path fails loudly without a GPU (no CPU fallback)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _headers():
    # the interface (totsu_f32hip.h) and the test hooks / probes the same library exports (totsu_f32hip_test.h)
    return "\n".join(open(os.path.join(ROOT, "include", f)).read() for f in ("totsu_f32hip.h", "totsu_f32hip_test.h"))


def _declared():
    txt = _headers()
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
            if f.endswith((".py", ".hip", ".h", ".cpp", ".inc")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f
                assert "libtotsu_oracle" not in src, f


def test_test_hooks_are_not_in_the_interface_header():
    # thip_test_* (fault injection, engine switches, single-kernel entry points) live in totsu_f32hip_test.h only
    main = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "totsu_f32hip.h")).read(), flags=re.S)
    assert not re.findall(r"\bthip_test_[a-z0-9_]+\s*\(", main)
    hooks = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "totsu_f32hip_test.h")).read(), flags=re.S)
    assert len(set(re.findall(r"\b(thip_test_[a-z0-9_]+)\s*\(", hooks))) >= 8
    # and no host built on the interface (examples/, the C++ mirrors) needs them
    for d, files in (("include", ["totsu_f32hip.hpp", "totsu_f32hip_prob.hpp"]), ("examples", os.listdir(os.path.join(ROOT, "examples")))):
        for f in files:
            p = os.path.join(ROOT, d, f)
            if os.path.isfile(p) and f.endswith((".c", ".cpp", ".hpp", ".h")):
                assert "thip_test_" not in open(p).read(), p


def test_authored_rust_binding_declares_every_symbol():
    # rust/totsu_f32hip cannot be compiled here (no cargo); at least its extern block stays one-to-one with the header
    src = open(os.path.join(ROOT, "rust", "totsu_f32hip", "src", "ffi.rs")).read()
    rust = set(re.findall(r"pub fn (thip_[a-z0-9_]+)\s*\(", src))
    assert rust == set(_declared()), rust ^ set(_declared())


def test_authored_rust_binding_has_the_header_arities():
    # second line of defence for the uncompiled crate: every extern declaration takes as many arguments as the header says
    hdr = re.sub(r"/\*.*?\*/", "", _headers(), flags=re.S)
    arity = {}
    for m in re.finditer(r"\b(thip_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.S):
        args = m.group(2).strip()
        arity[m.group(1)] = 0 if args in ("void", "") else args.count(",") + 1
    for path in (os.path.join(ROOT, "rust", "totsu_f32hip", "src", "ffi.rs"), os.path.join(ROOT, "INTEGRATION.md")):
        src = open(path).read()
        for m in re.finditer(r"pub fn (thip_[a-z0-9_]+)\(([^;]*?)\)\s*(->[^;]*)?;", src, flags=re.S):
            if m.group(1) in arity:
                n = 0 if not m.group(2).strip() else m.group(2).count(":")
                assert n == arity[m.group(1)], (path, m.group(1), n, arity[m.group(1)])
