"""CPU: the C-ABI library loads and exports every function include/sclens_hip.h declares (no GPU compute)."""
import os
import re

from sclens_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "sclens_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sclens_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported_and_bound():
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 40
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/sclens_hip.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert set(_lib.SIGNATURES) == set(names)


def test_version_and_no_device_is_an_error_code():
    import ctypes as C

    lib = _lib.load()
    assert b"gfx950" in lib.sclens_hip_version()
    import torch

    if not torch.cuda.is_available():  # build container: creating a context must fail loudly, not fall back
        h = C.c_void_p()
        assert lib.sclens_hip_create(C.byref(h), 0) == 2  # SCLENS_ERR_NO_DEVICE
        try:
            _lib.Context(0)
            assert False
        except _lib.SclensHipError as e:
            assert e.code == 2


def test_product_path_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "sclens_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f


def test_julia_shim_binds_only_declared_symbols():
    """julia/scLENS_hip.jl cannot be executed here (no julia): at least every symbol it ccall's must be declared in the
    header and exported by the library, with the library path it expects."""
    src = open(os.path.join(ROOT, "julia", "scLENS_hip.jl")).read()
    used = sorted(set(re.findall(r"\(:(sclens_[a-z0-9_]+),\s*LIB\)", src)))
    assert len(used) >= 10
    declared = set(_declared())
    lib = _lib.load()
    for n in used:
        assert n in declared, f"{n} used by the Julia shim but not declared in include/sclens_hip.h"
        assert hasattr(lib, n)
    assert "libsclens_hip.so" in src
