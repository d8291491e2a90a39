"""Static check of the Julia shim (qpcontrol.jl_amd/julia/*.jl) against the C ABI: no Julia toolchain exists here, so the `ccall`s
have never run -- but every one of them can be PARSED and held against the prototype it binds in include/qpc.h: the symbol is
declared there and exported by the library, the argument tuple has the prototype's length, every argument and the return type are of
the prototype's kind (pointer / 32-bit integer / 64-bit integer / double).  (What `test/runtests.jl:35-43` would plug into.)"""
import ctypes
import glob
import os
import re

import pytest

import emu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "qpc.h")
JULIA = sorted(glob.glob(os.path.join(ROOT, "qpcontrol.jl_amd", "julia", "*.jl")))

C_KIND = {"int32_t": "i32", "int": "i32", "int64_t": "i64", "uint64_t": "u64", "double": "f64", "void": "void"}
JL_KIND = {"Int32": "i32", "Cint": "i32", "Int64": "i64", "UInt64": "u64", "Csize_t": "u64", "Float64": "f64", "Cdouble": "f64", "Cvoid": "void",
           "Cstring": "ptr"}


def c_prototypes(text):
    """name -> (return kind, [argument kinds]) of every function declared in the header"""
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    text = re.sub(r"^\s*#[^\n]*$", " ", text, flags=re.M)  # preprocessor lines
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(qpc_\w+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if "typedef" in ret or "struct" in ret and "*" not in ret and name.endswith("_t"):
            continue

        def kind(decl):
            decl = decl.strip()
            if "*" in decl or "[" in decl:
                return "ptr"
            base = [w for w in re.sub(r"\b(const|extern)\b", " ", decl).split() if w][0]
            return C_KIND[base]
        arglist = [] if args in ("", "void") else [kind(a) for a in args.split(",")]
        protos[name] = (kind(ret), arglist)
    return protos


def split_top(s):
    """split at top-level commas (braces and parentheses nest)"""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "({[":
            depth += 1
        elif ch in ")}]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip()); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def julia_ccalls(text):
    """(symbol, return type, [argument types], number of actual arguments, line) of every ccall((:sym, lib), ...)"""
    calls = []
    for m in re.finditer(r"ccall\(", text):
        i = m.end()
        depth, j = 1, i
        while depth:
            depth += {"(": 1, ")": -1}.get(text[j], 0)
            j += 1
        parts = split_top(text[i:j - 1])
        sym = re.match(r"\(\s*:(\w+)\s*,", parts[0])
        assert sym, parts[0]
        types = parts[2].strip()
        assert types.startswith("(") and types.endswith(")"), types
        argt = [t for t in split_top(types[1:-1]) if t]
        calls.append((sym.group(1), parts[1].strip(), argt, len(parts) - 3, text.count("\n", 0, m.start()) + 1))
    return calls


def jl_kind(t):
    t = t.strip()
    if t.startswith(("Ptr{", "Ref{")) or t == "Cstring":
        return "ptr"
    return JL_KIND[t]


def check(text, protos, exported):
    problems = []
    for sym, ret, argt, nactual, line in julia_ccalls(text):
        if sym not in protos:
            problems.append(f"line {line}: {sym} is not declared in include/qpc.h"); continue
        if not exported(sym):
            problems.append(f"line {line}: {sym} is not exported by the library")
        pret, pargs = protos[sym]
        if len(argt) != len(pargs) or nactual != len(pargs):
            problems.append(f"line {line}: {sym} takes {len(pargs)} arguments, the ccall declares {len(argt)} and passes {nactual}"); continue
        if jl_kind(ret) != pret:
            problems.append(f"line {line}: {sym} returns {pret}, the ccall says {ret}")
        for k, (a, b) in enumerate(zip(argt, pargs)):
            if jl_kind(a) != b:
                problems.append(f"line {line}: {sym} argument {k + 1} is {b} in the header, {a} in the ccall")
    return problems


@pytest.fixture(scope="module")
def abi():
    protos = c_prototypes(open(HEADER).read())
    lib = os.path.join(ROOT, "qpcontrol.jl_amd", "libqpcontrol_hip.so")
    if not os.path.exists(lib):  # (the product library is built by __graft_entry__.build(); the emulation exports the same ABI minus device-only calls)
        lib = emu.build()
    L = ctypes.CDLL(lib)
    return protos, lambda s: hasattr(L, s)


def test_header_parses(abi):
    protos, _ = abi
    assert len(protos) > 60
    assert protos["qpc_control_batch"] == ("i32", ["ptr", "i64"] + ["ptr"] * 12)
    assert protos["qpc_qp_setup"] == ("i32", ["i32", "i32", "i64", "ptr", "ptr", "i64", "ptr", "ptr", "i32", "ptr"])
    assert protos["qpc_status_string"][0] == "ptr" and protos["qpc_qp_free"][0] == "void"


@pytest.mark.parametrize("path", JULIA, ids=[os.path.basename(p) for p in JULIA])
def test_every_ccall_matches_its_prototype(path, abi):
    protos, exported = abi
    text = open(path).read()
    calls = julia_ccalls(text)
    if os.path.basename(path) == "QPControlHIP.jl":
        assert len(calls) >= 36 and len({c[0] for c in calls}) >= 31
    problems = check(text, protos, exported)
    assert not problems, "\n".join(problems)


def test_a_broken_ccall_is_caught(abi):
    """the checker is red when a ccall is wrong: a missing argument, a scalar where the ABI takes a pointer, an Int64 where it takes an
    int32_t, an unknown symbol"""
    protos, exported = abi
    good = 'ccall((:qpc_qp_analyze, libqpc), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), p.handle, l, u)'
    assert check(good, protos, exported) == []
    for bad in ('ccall((:qpc_qp_analyze, libqpc), Int32, (Ptr{Cvoid}, Ptr{Float64}), p.handle, l)',
                'ccall((:qpc_qp_analyze, libqpc), Int32, (Ptr{Cvoid}, Float64, Ptr{Float64}), p.handle, l, u)',
                'ccall((:qpc_qp_plan_info, libqpc), Int32, (Ptr{Cvoid}, Ptr{Int64}, Int64), p.handle, out, 12)',
                'ccall((:qpc_qp_analyse, libqpc), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), p.handle, l, u)',
                'ccall((:qpc_qp_free, libqpc), Int32, (Ptr{Cvoid},), p.handle)'):
        assert check(bad, protos, exported), bad
