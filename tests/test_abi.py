"""CPU: the C-ABI library loads and exports every function include/sclens_hip.h declares (no GPU compute)."""
import os

import numpy as np
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


# ---- ccall signatures of the Julia shim against the header ------------------------------------------------------------
def _balanced(src, i):
    """src[i] == '(' -> index just past its matching ')'"""
    depth = 0
    for j in range(i, len(src)):
        depth += src[j] in "([{"
        depth -= src[j] in ")]}"
        if depth == 0:
            return j + 1
    raise ValueError("unbalanced")


def _split_top(s):
    out, depth, cur, in_str = [], 0, "", False
    for ch in s:
        if ch == '"':
            in_str = not in_str
        if not in_str:
            depth += ch in "([{"
            depth -= ch in ")]}"
            if ch == "," and depth == 0:
                out.append(cur.strip())
                cur = ""
                continue
        cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def _julia_ccalls():
    src = open(os.path.join(ROOT, "julia", "scLENS_hip.jl")).read()
    src = "\n".join(line.split("#")[0] if '"' not in line.split("#")[0][-1:] else line for line in src.split("\n"))
    calls = []
    for m in re.finditer(r"\bccall\(", src):
        end = _balanced(src, m.end() - 1)
        parts = _split_top(src[m.end(): end - 1])
        name = re.match(r"\(:(\w+),\s*LIB\)", parts[0]).group(1)
        ret = parts[1]
        types = _split_top(parts[2].strip()[1:-1]) if parts[2].strip() != "()" else []
        calls.append((name, ret, types, parts[3:]))
    return calls


def _header_prototypes():
    src = open(os.path.join(ROOT, "include", "sclens_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    protos = {}
    for m in re.finditer(r"([\w\s\*]+?)\b(sclens_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        params = [] if args in ("", "void") else [re.sub(r"\s+", " ", a.strip()) for a in args.split(",")]
        protos[name] = (re.sub(r"\s+", " ", ret), params)
    return protos


def _c_kind(decl):
    """C parameter declaration -> a category that the Julia type of the same position must match"""
    d = decl.replace("const ", "").strip()
    d = re.sub(r"\b\w+$", "", d).strip() if not d.endswith("*") else d  # drop the parameter name
    d = d.replace(" *", "*")
    if "(*" in decl or d.endswith("_fn"):
        return "fnptr"
    table = {"int": "int", "int64_t": "i64", "uint64_t": "u64", "double": "f64", "float": "f32", "char*": "cstr",
             "float*": "p_f32", "double*": "p_f64", "int64_t*": "p_i64", "int32_t*": "p_i32", "uint32_t*": "p_u32",
             "uint8_t*": "p_u8", "int*": "p_int", "void*": "p_void", "void**": "pp_void", "uint64_t*": "p_u64"}
    if d in table:
        return table[d]
    if re.fullmatch(r"sclens_hip_\w+\*\*", d):
        return "pp_void"
    if re.fullmatch(r"sclens_hip_\w+\*", d):
        return "p_void"
    raise AssertionError(f"unmapped C type {decl!r}")


_JL = {"Cint": "int", "Int64": "i64", "UInt64": "u64", "Cdouble": "f64", "Float64": "f64", "Cfloat": "f32", "Float32": "f32",
       "Cstring": "cstr", "Ptr{Float32}": "p_f32", "Ptr{Float64}": "p_f64", "Ptr{Int64}": "p_i64", "Ref{Int64}": "p_i64",
       "Ptr{Int32}": "p_i32", "Ptr{UInt32}": "p_u32", "Ptr{UInt8}": "p_u8", "Ref{Cint}": "p_int", "Ptr{Cint}": "p_int",
       "Ptr{Cvoid}": "p_void", "Ref{Ptr{Cvoid}}": "pp_void", "Ptr{Ptr{Cvoid}}": "pp_void", "Ptr{UInt64}": "p_u64"}
_JL_RET = {"Cint": "int", "Cvoid": "void", "Cstring": "const char*", "Int64": "int64_t", "Cdouble": "double"}


def test_julia_shim_ccall_signatures_match_the_header():
    """The shim cannot run here, so its ccalls are checked statically: every call passes as many values as it declares types,
    and the declared return / argument types are the header's, position by position (a wrong width or a missing argument in a
    ccall is silent memory corruption at run time, not an error)."""
    protos = _header_prototypes()
    calls = _julia_ccalls()
    assert len(calls) >= 20
    for name, ret, types, values in calls:
        assert name in protos, name
        c_ret, c_params = protos[name]
        assert len(values) == len(types), f"{name}: {len(types)} types but {len(values)} values"
        assert len(types) == len(c_params), f"{name}: the header has {len(c_params)} parameters, the ccall {len(types)}"
        assert _JL_RET[ret] == c_ret, f"{name}: returns {c_ret}, ccall says {ret}"
        for pos, (jt, cd) in enumerate(zip(types, c_params)):
            assert jt in _JL, f"{name}: unmapped Julia type {jt}"
            assert _JL[jt] == _c_kind(cd), f"{name} argument {pos}: header `{cd}` vs ccall `{jt}`"


def test_ctypes_signatures_match_the_header():
    """the same check for the ctypes table of sclens_amd/_lib.py"""
    import ctypes as C

    protos = _header_prototypes()
    kind = {C.c_int: "int", C.c_int64: "i64", C.c_uint64: "u64", C.c_double: "f64", C.c_float: "f32", C.c_char_p: "cstr",
            _lib.c_f32p: "p_f32", _lib.c_f64p: "p_f64", _lib.c_i64p: "p_i64", _lib.c_i32p: "p_i32", _lib.c_u32p: "p_u32",
            _lib.c_u8p: "p_u8", C.POINTER(C.c_int): "p_int", C.c_void_p: "p_void", C.POINTER(C.c_void_p): "pp_void",
            C.POINTER(C.c_uint64): "p_u64", _lib.ALLREDUCE_FN: "fnptr", _lib.REDUCE_FN: "fnptr"}
    for name, (res, args) in _lib.SIGNATURES.items():
        c_ret, c_params = protos[name]
        assert len(args) == len(c_params), f"{name}: header {len(c_params)} parameters, ctypes {len(args)}"
        for pos, (a, cd) in enumerate(zip(args, c_params)):
            ck = _c_kind(cd)
            got = kind[a]
            ck, got = ck.replace("p_int", "p_i32"), got.replace("p_int", "p_i32")  # `int` is 32 bits wide: one ctypes class
            # typed data pointers may be bound as void* (raw device pointers / numpy .ctypes.data)
            assert got == ck or (got == "p_void" and ck.startswith("p_")), f"{name} argument {pos}: header `{cd}` vs ctypes {a}"
        want_ret = {"int": C.c_int, "void": None, "const char*": C.c_char_p, "int64_t": C.c_int64, "double": C.c_double,
                    "void*": C.c_void_p}[c_ret]
        assert res is want_ret or (res == want_ret), f"{name}: returns {c_ret}"


def test_julia_shim_takes_the_signal_cut_from_the_top_and_falls_back_like_the_reference():
    """The shim must be the Python twin where decisions are made: after the float64 refinement the number of
    signals is NOT `sum(L .> lambda_c)` (a close pair may come out in the other order and the retained set would stop being
    contiguous) but the count from the top (api.cut_with_guard_band); codes 2 / 3 fall back to device_="cpu"; an unsupported
    `centering` string prints the reference's warning and runs the mean branch (scLENS.jl:655-657)."""
    src = open(os.path.join(ROOT, "julia", "scLENS_hip.jl")).read()
    code = "\n".join(line.split("#", 1)[0] for line in src.splitlines())  # comments stripped
    assert "sum(L .> lambda_c)" not in code
    assert "cut_with_guard_band!(L, lambda_c" in code and "findlast(x -> !(x > lambda_c), L)" in code
    assert "e.code in (2, 3)" in code and 'device_="cpu"' in code
    assert "not supported in the current algorithm" in src
    # the helper agrees with the Python twin on a hand case: a refined pair that swaps order around the cut
    from sclens_amd import api

    L = np.array([0.5, 1.0, 1.5, 2.0 + 1e-7, 2.0 + 2e-7, 3.0, 9.0])
    Lw, k, nL, guard = api.cut_with_guard_band(L, 2.0 + 1.5e-7, 4.0, lambda lo, hi: np.array([2.0 + 3e-7, 2.0 + 1e-7][: hi - lo]))
    assert k == 2 and nL.tolist() == [9.0, 3.0] and guard["refined"]
