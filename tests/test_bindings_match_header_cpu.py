"""The three descriptions of the C ABI must say the same thing: the prototypes of include/loraine_hip.h, the ctypes
signatures the Python host binds (loraine.jl_amd/_capi.py) and the `ccall`s of the Julia glue a Loraine.jl maintainer
adds (julia/LoraineHIP.jl).  The build image has no Julia, so the glue is never executed here: this test
at least pins every one of its `ccall`s -- symbol, return type, number, order and kind of the arguments -- to the header,
argument by argument."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---------------------------------------------------------------------------------------------- the header
def _kind_of_c(t):
    """C parameter / return type -> a kind both bindings can be compared with."""
    t = re.sub(r"\bconst\b", "", t).strip()
    t = re.sub(r"\s+", " ", t).replace(" *", "*")
    if t in ("lrn_host_allreduce_fn", "lrn_host_allgather_fn"):
        return "fnptr"
    stars = t.count("*")
    base = t.replace("*", "").strip()
    if base == "lrn_ctx":
        return {1: "ctx", 2: "ctx*"}[stars]
    scal = {"int": "i32", "int32_t": "i32", "double": "f64", "int64_t": "i64", "uint64_t": "i64", "long": "i64",
            "char": "char", "void": "void", "unsigned char": "u8"}[base]
    return scal + "*" * stars


def _header_prototypes():
    txt = open(os.path.join(ROOT, "include", "loraine_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"//[^\n]*", "", txt)
    protos = {}
    for ret, name, params in re.findall(r"^\s*((?:const\s+)?[a-z0-9_]+\s*\**)\s*(lrn_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", txt,
                                        flags=re.M | re.S):
        params = " ".join(params.split())
        kinds = []
        if params and params != "void":
            for p in params.split(","):
                p = p.strip()
                m = re.match(r"^(.*?)([A-Za-z_][A-Za-z0-9_]*)$", p)          # type, then the parameter's name
                assert m, (name, p)
                kinds.append(_kind_of_c(m.group(1)))
        protos[name] = (_kind_of_c(ret), kinds)
    return protos


def test_the_header_parser_sees_every_declared_symbol():
    protos = _header_prototypes()
    txt = open(os.path.join(ROOT, "include", "loraine_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = set(re.findall(r"\b(lrn_[a-z0-9_]+)\s*\(", txt))
    assert names == set(protos), sorted(names ^ set(protos))
    assert protos["lrn_create"] == ("i32", ["ctx*", "i32"])
    assert protos["lrn_last_error"] == ("char*", ["ctx"])
    assert protos["lrn_make_rhs"] == ("i32", ["ctx", "f64*", "f64**", "f64*"])
    assert protos["lrn_schur_shard_doubles"] == ("i64", ["ctx"])


# ---------------------------------------------------------------------------------------------- ctypes
def _ctypes_accepts(ct, kind):
    from loraine_jl_amd import _capi
    if ct is C.c_int:
        return kind == "i32"
    if ct is C.c_double:
        return kind == "f64"
    if ct in (C.c_int64, C.c_uint64):
        return kind == "i64"
    if ct is C.c_char_p:
        return kind == "char*"
    if ct is _capi.c_ctx:                                  # c_void_p: the context, or any data pointer passed as an address
        return kind == "ctx" or kind.endswith("*") or kind == "fnptr"
    if ct is _capi.PD:
        return kind == "f64*"
    if ct is _capi.PI:
        return kind == "i32*"
    if ct is _capi.PI64:
        return kind == "i64*"
    if ct is _capi.PPD:                                    # POINTER(c_void_p): an array of addresses, or the lrn_ctx** of lrn_create
        return kind.endswith("**") or kind == "ctx*"
    if isinstance(ct, type) and issubclass(ct, C._CFuncPtr):
        return kind == "fnptr"
    return False


def test_ctypes_signatures_match_the_header():
    from loraine_jl_amd import _capi
    protos = _header_prototypes()
    assert set(protos) == set(_capi.SIGNATURES)
    for name, (ret, kinds) in protos.items():
        restype, argtypes = _capi.SIGNATURES[name]
        assert len(argtypes) == len(kinds), f"{name}: {len(argtypes)} ctypes arguments, {len(kinds)} in the header"
        assert _ctypes_accepts(restype, ret), f"{name}: return type {restype} vs {ret}"
        for pos, (ct, kind) in enumerate(zip(argtypes, kinds)):
            assert _ctypes_accepts(ct, kind), f"{name}: argument {pos} is {ct} in _capi.py, {kind} in the header"


# ---------------------------------------------------------------------------------------------- Julia
_JULIA_KIND = {
    "Cint": {"i32"}, "Cdouble": {"f64"}, "Int64": {"i64"}, "Cstring": {"char*"},
    "Ptr{Cvoid}": {"ctx", "void*"}, "Ref{Ptr{Cvoid}}": {"ctx*"},
    "Ptr{Float64}": {"f64*"}, "Ptr{Int64}": {"i64*"}, "Ref{Cint}": {"i32*"}, "Ptr{Cint}": {"i32*"},
    "Ptr{Ptr{Float64}}": {"f64**"}, "Ptr{Ptr{Int64}}": {"i64**"}, "Ptr{UInt8}": {"void*", "u8*"},
}


def _split_top_level(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "{(":
            depth += 1
        elif ch in "})":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def _julia_ccalls():
    src = open(os.path.join(ROOT, "julia", "LoraineHIP.jl")).read()
    src = re.sub(r"#[^\n]*", "", src)
    calls = []
    for m in re.finditer(r"ccall\(\(:(lrn_[a-z0-9_]+),\s*LIB\),\s*([A-Za-z0-9{}]+),\s*\(", src):
        # the argument-type tuple: up to its matching parenthesis
        i, depth = m.end(), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(src[i], 0)
            i += 1
        types = _split_top_level(src[m.end():i - 1])
        # the actual arguments: up to the parenthesis that closes the ccall
        j, depth = i, 1
        while depth:
            depth += {"(": 1, ")": -1, "[": 1, "]": -1}.get(src[j], 0)
            j += 1
        args = _split_top_level(src[i:j - 1].lstrip().lstrip(","))
        calls.append((m.group(1), m.group(2), types, args))
    return calls


def test_julia_ccalls_match_the_header():
    protos = _header_prototypes()
    calls = _julia_ccalls()
    assert len(calls) >= 40
    for name, ret, types, args in calls:
        assert name in protos, f"LoraineHIP.jl calls {name}, which include/loraine_hip.h does not declare"
        hret, kinds = protos[name]
        assert ret in _JULIA_KIND and hret in _JULIA_KIND[ret], f"{name}: returns {ret} in Julia, {hret} in the header"
        assert len(types) == len(kinds), f"{name}: {len(types)} argument types in Julia, {len(kinds)} in the header"
        assert len(args) == len(types), f"{name}: {len(args)} arguments passed for {len(types)} declared types"
        for pos, (jt, kind) in enumerate(zip(types, kinds)):
            assert jt in _JULIA_KIND, f"{name}: argument {pos}: unknown Julia type {jt}"
            assert kind in _JULIA_KIND[jt], f"{name}: argument {pos} is {jt} in Julia, {kind} in the header"


def test_julia_glue_covers_the_entry_points_of_the_host_loop():
    """Every entry point the Python host (solvers.py / resident.py / device.py, the code the parity tests run) calls on
    the per-iteration path has a `ccall` in the Julia glue -- except synthetic-data generators, probes and debug hooks."""
    called = {c[0] for c in _julia_ccalls()}
    protos = _header_prototypes()
    exempt = {n for n in protos if n.startswith("lrn_dbg_") or n.startswith("lrn_synthetic_")} | {
        "lrn_version", "lrn_device_count", "lrn_get_constraint", "lrn_set_option", "lrn_set_scaling", "lrn_schur_get",
        "lrn_schur_plan", "lrn_comm_init_host", "lrn_get_timing", "lrn_get_count", "lrn_mfma_f64_peak", "lrn_xcc_probe",
        "lrn_hbm_copy_peak"}
    missing = sorted(set(protos) - called - exempt)
    assert not missing, f"no ccall in julia/LoraineHIP.jl for: {missing}"


def test_julia_source_is_balanced():
    """A parser-free sanity check of the glue nobody can run here: brackets balance and every block opener has its `end`."""
    src = open(os.path.join(ROOT, "julia", "LoraineHIP.jl")).read()
    code = re.sub(r'"(?:\\.|[^"\\])*"', '""', re.sub(r"#[^\n]*", "", src))
    for a, b in ("()", "[]", "{}"):
        assert code.count(a) == code.count(b), f"unbalanced {a}{b}"
    # block openers: keywords at the start of a statement (a `for` inside a comprehension has no `end`), `begin` / `do`
    # anywhere; `end` as the last index inside brackets (a[end]) does not occur in this file
    openers = len(re.findall(r"(?m)^[ \t]*(?:mutable[ \t]+struct|function|module|struct|if|for|while|let|try)\b", code))
    openers += len(re.findall(r"\b(?:begin|do)[ \t]*$", code, flags=re.M))
    ends = len(re.findall(r"(?<![A-Za-z0-9_.!:])end(?![A-Za-z0-9_!])", code))
    assert openers == ends, f"{openers} block openers, {ends} `end`s"
