"""Header ↔ Julia binding drift check.

julia/ThunderboltHIPBackend.jl cannot be executed in this image (no Julia toolchain), so the only verification its `ccall`s can get
is a static one: every `ccall((:tb_xxx, libtbhip), Ret, (ArgTypes…), …)` must name a symbol that include/tbhip.h declares, with the
same number of arguments and, argument by argument, the same ABI class (32-bit integer, 64-bit integer, size_t, double, float,
pointer).  The same classes are checked against the ctypes prototypes of thunderbolt.jl_amd/_lib.py where those are declared, and
against the exported symbols of the built library."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "tbhip.h")
JULIA = os.path.join(ROOT, "julia", "ThunderboltHIPBackend.jl")


def strip_c_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def c_class(decl):
    """ABI class of one C parameter declaration."""
    d = decl.strip()
    if d == "void":
        return None
    if "*" in d or "[" in d:
        return "ptr"
    d = re.sub(r"\bconst\b", " ", d)
    toks = d.split()
    ty = " ".join(toks[:-1]) if len(toks) > 1 else toks[0]   # drop the parameter name
    table = {"int": "i32", "int32_t": "i32", "uint32_t": "i32", "unsigned": "i32", "unsigned int": "i32", "int64_t": "i64", "uint64_t": "i64",
             "long long": "i64", "size_t": "size", "double": "f64", "float": "f32"}
    assert ty in table, "unclassified C parameter %r" % decl
    return table[ty]


def header_prototypes():
    text = strip_c_comments(open(HEADER, encoding="utf-8").read())
    protos = {}
    for m in re.finditer(r"(?m)^\s*((?:const\s+)?(?:char|int|int32_t|int64_t|double)\s*\*?)\s*(tb_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        classes = [c_class(a) for a in args.replace("\n", " ").split(",")]
        classes = [c for c in classes if c is not None]
        protos[name] = ("ptr" if "*" in ret else c_class(ret + " _"), classes)
    return protos


def split_top_level(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "({[":
            depth += 1
        elif ch in ")}]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return [x.strip() for x in out]


def julia_class(ty):
    t = ty.strip()
    if t.startswith(("Ptr{", "Ref{")) or t in ("Cstring", "Ptr", "Ref"):
        return "ptr"
    table = {"Cint": "i32", "Int32": "i32", "UInt32": "i32", "Cuint": "i32", "Int64": "i64", "Clonglong": "i64", "UInt64": "i64", "Csize_t": "size",
             "Cdouble": "f64", "Float64": "f64", "Tv": "f64", "Cfloat": "f32", "Float32": "f32"}
    assert t in table, "unclassified Julia ccall type %r" % ty
    return table[t]


def matching_paren(text, start):
    depth = 0
    for k in range(start, len(text)):
        if text[k] == "(":
            depth += 1
        elif text[k] == ")":
            depth -= 1
            if depth == 0:
                return k
    raise AssertionError("unbalanced parentheses in the Julia file")


def julia_ccalls():
    text = re.sub(r"#=.*?=#", " ", open(JULIA, encoding="utf-8").read(), flags=re.S)   # block comments first, then line comments
    text = re.sub(r"#[^\n]*", "", text)
    calls = []
    for m in re.finditer(r"ccall\s*\(", text):
        end = matching_paren(text, m.end() - 1)
        parts = split_top_level(text[m.end():end])
        sym = re.match(r"\(\s*:(tb_[a-z0-9_]+)\s*,\s*libtbhip\s*\)", parts[0])
        assert sym, "ccall without a (:symbol, libtbhip) target: %r" % parts[0]
        ret = parts[1]
        tup = parts[2].strip()
        assert tup.startswith("(") and tup.endswith(")"), tup
        argtypes = [a for a in split_top_level(tup[1:-1]) if a]
        nvalues = len(parts) - 3
        line = text.count("\n", 0, m.start()) + 1
        calls.append((sym.group(1), ret, argtypes, nvalues, line))
    return calls


def test_header_parses_every_entry_point():
    protos = header_prototypes()
    raw = strip_c_comments(open(HEADER, encoding="utf-8").read())
    declared = set(re.findall(r"\b(tb_[a-z0-9_]+)\s*\(", raw))
    assert declared == set(protos), sorted(declared ^ set(protos))
    assert len(protos) >= 85


def test_julia_ccalls_match_the_header():
    protos = header_prototypes()
    calls = julia_ccalls()
    assert len(calls) >= 30
    for name, ret, argtypes, nvalues, line in calls:
        assert name in protos, "julia:%d calls %s, which include/tbhip.h does not declare" % (line, name)
        cret, cargs = protos[name]
        assert julia_class(ret) == cret, "julia:%d %s: return type %s vs header %s" % (line, name, ret, cret)
        assert len(argtypes) == len(cargs), "julia:%d %s: %d argument types, header has %d" % (line, name, len(argtypes), len(cargs))
        assert nvalues == len(argtypes), "julia:%d %s: %d values passed for %d argument types" % (line, name, nvalues, len(argtypes))
        for k, (jt, ct) in enumerate(zip(argtypes, cargs)):
            assert julia_class(jt) == ct, "julia:%d %s: argument %d is %s, header says %s" % (line, name, k + 1, jt, ct)


def test_every_bound_symbol_is_documented_in_integration_md():
    text = open(os.path.join(ROOT, "INTEGRATION.md"), encoding="utf-8").read() + open(HEADER, encoding="utf-8").read()
    for name in {c[0] for c in julia_ccalls()}:
        assert name in text, name


def test_ctypes_prototypes_match_the_header():
    """thunderbolt.jl_amd/_lib.py declares argtypes for the entries the Python mirror calls: same arity and ABI classes as the header."""
    import ctypes as C
    lib_path = os.path.join(ROOT, "thunderbolt.jl_amd", "libtbhip.so")
    if not os.path.exists(lib_path):
        pytest.skip("libtbhip.so not built")
    import thunderbolt_jl_amd as tb
    lib = tb.lib()
    protos = header_prototypes()

    def ctypes_class(t):
        if t in (C.c_int, C.c_int32, C.c_uint32, C.c_uint):
            return "i32"
        if t in (C.c_int64, C.c_longlong, C.c_uint64):
            return "i64"
        if t is C.c_size_t:
            return "size"
        if t is C.c_double:
            return "f64"
        if t is C.c_float:
            return "f32"
        return "ptr"

    # c_size_t and c_uint64 / c_long may be the same ctypes class on LP64: compare with size ≡ i64 folded
    fold = lambda c: "i64" if c == "size" else c
    checked = 0
    for name, (_, cargs) in protos.items():
        fn = getattr(lib, name)          # every declared symbol is exported
        if fn.argtypes is None:
            continue
        got = [fold(ctypes_class(t)) for t in fn.argtypes]
        assert got == [fold(c) for c in cargs], "%s: ctypes %s vs header %s" % (name, got, cargs)
        checked += 1
    assert checked >= 20
