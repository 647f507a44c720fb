"""Static cross-check of the Julia package extension (julia/ext/NonuniformFFTsMI355XExt.jl) against the C header and the
reference source — there is no Julia runtime in the build image, so nothing else would notice a drifted `ccall`.

Checked: every `ccall((:sym, libnufft), Ret, (ArgTypes...), ...)` against the prototype of `sym` in include/nufft_mi355x.h
(symbol, return type, argument count and types); `struct CParams` / `struct CCallbacks` against `nufft_params` /
`nufft_callbacks` (field names, order, types, size); `kernel_id` against NUFFT_KERNEL_*; `check`'s return-code sets against
the header's enum; and every `NonuniformFFTs.<name>` the extension adds a method to against the reference source (a method
of that name with the same number of positional arguments exists) when /root/reference is present."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "julia", "ext", "NonuniformFFTsMI355XExt.jl")
HEADER = os.path.join(ROOT, "include", "nufft_mi355x.h")
REFERENCE = "/root/reference"


def _strip_c_comments(text):
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def _header_prototypes():
    """name -> (return type, [argument types]) with C types normalised (no names, single spaces)."""
    text = _strip_c_comments(open(HEADER).read())
    protos = {}
    for m in re.finditer(r"^\s*((?:const\s+)?\w+\s*\**)\s*(nufft_\w+)\s*\(([^;{}]*)\)\s*;", text, flags=re.M):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        argt = []
        for a in [x.strip() for x in args.split(",")] if args.strip() not in ("", "void") else []:
            a = re.sub(r"\s+", " ", a)
            mm = re.match(r"^(.*?[\*\s])(\w+)$", a)        # drop the parameter name
            t = mm.group(1) if mm else a
            argt.append(re.sub(r"\s*\*\s*", "*", t).strip())
        protos[name] = (re.sub(r"\s*\*\s*", "*", re.sub(r"\s+", " ", ret)).strip(), argt)
    return protos


# C type (normalised) -> the Julia types a ccall may legitimately use for it
_JULIA_FOR_C = {
    "int": {"Cint"}, "int64_t": {"Int64"}, "double": {"Float64", "Cdouble"},
    "const char*": {"Cstring"},
    "nufft_plan*": {"Ptr{Cvoid}"}, "const nufft_plan*": {"Ptr{Cvoid}"},
    "nufft_plan**": {"Ref{Ptr{Cvoid}}", "Ptr{Ptr{Cvoid}}"},
    "void*": {"Ptr{Cvoid}"}, "const void*": {"Ptr{Cvoid}"},
    "void*const*": {"Ptr{Ptr{Cvoid}}"}, "const void*const*": {"Ptr{Ptr{Cvoid}}"},
    "const nufft_params*": {"Ref{CParams}"}, "const nufft_callbacks*": {"Ref{CCallbacks}"},
    "nufft_info*": {"Ref{CInfo}"},
}


def _norm_c(t):
    return re.sub(r"\s*\*\s*", "*", re.sub(r"\s+", " ", t)).replace("* const", "*const").replace("*const *", "*const*").strip()


def _shim_ccalls():
    text = open(SHIM).read()
    calls = []
    for m in re.finditer(r"ccall\(\(:(\w+),\s*libnufft\),\s*([\w{}]+),\s*\(([^()]*)\)", text):
        args = [a.strip() for a in m.group(3).split(",") if a.strip()]
        calls.append((m.group(1), m.group(2), args))
    return calls


def test_every_ccall_matches_its_prototype():
    protos = _header_prototypes()
    calls = _shim_ccalls()
    assert len(calls) >= 7, calls
    assert {"nufft_plan_create_ex", "nufft_plan_destroy", "nufft_set_points", "nufft_exec_type1_cb", "nufft_exec_type2_cb",
            "nufft_sizeof_params", "nufft_last_error_message"} <= {c[0] for c in calls}
    for name, ret, args in calls:
        assert name in protos, f"{name}: not declared in include/nufft_mi355x.h"
        cret, cargs = protos[name]
        assert ret in _JULIA_FOR_C[_norm_c(cret)], (name, ret, cret)
        assert len(args) == len(cargs), (name, args, cargs)
        for ja, ca in zip(args, cargs):
            assert ja in _JULIA_FOR_C[_norm_c(ca)], (name, ja, ca)


def _header_struct(name):
    text = _strip_c_comments(open(HEADER).read())
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), text, flags=re.S).group(1)
    fields = []
    for decl in [d.strip() for d in body.split(";") if d.strip()]:
        m = re.match(r"^((?:const\s+)?\w+\s*\*?)\s*(.*)$", decl)
        ctype = re.sub(r"\s+", " ", m.group(1)).strip()
        for item in [x.strip() for x in m.group(2).split(",")]:
            mm = re.match(r"^(\w+)(?:\[(\d+)\])?$", item)
            fields.append((mm.group(1), ctype, int(mm.group(2)) if mm.group(2) else 0))
    return fields


def _julia_struct(name):
    text = open(SHIM).read()
    body = re.search(r"struct %s\b(.*?)\bend\b" % name, text, flags=re.S).group(1)
    fields = []
    for item in re.split(r"[;\n]", body):
        item = item.split("#")[0].strip()
        if not item:
            continue
        fname, ftype = [x.strip() for x in item.split("::")]
        fields.append((fname, ftype))
    return fields


_CT = {"int32_t": ("Int32", C.c_int32), "int64_t": ("Int64", C.c_int64), "double": ("Float64", C.c_double),
       "const void*": ("Ptr{Cvoid}", C.c_void_p)}


def _check_struct(cname, jname):
    cf, jf = _header_struct(cname), _julia_struct(jname)
    assert [f[0] for f in cf] == [f[0] for f in jf], (cname, [f[0] for f in cf], [f[0] for f in jf])
    ctypes_fields = []
    for (fname, ctype, n), (_, jt) in zip(cf, jf):
        jbase, ct = _CT[ctype]
        expect = f"NTuple{{{n}, {jbase}}}" if n else jbase
        assert jt.replace(" ", "") == expect.replace(" ", ""), (cname, fname, jt, expect)
        ctypes_fields.append((fname, ct * n if n else ct))
    return type("S", (C.Structure,), {"_fields_": ctypes_fields})


def test_mirrored_structs_match_the_header_and_the_library():
    S = _check_struct("nufft_params", "CParams")
    _check_struct("nufft_callbacks", "CCallbacks")
    # the layout a C compiler gives the header's struct (ctypes applies the same alignment rules) = what the library reports
    from nufft_pkg import nufft
    assert C.sizeof(S) == nufft.lib.nufft_sizeof_params()


def test_enums_and_return_codes_used_by_the_shim():
    text, hdr = open(SHIM).read(), _strip_c_comments(open(HEADER).read())
    enum = {m.group(1): int(m.group(2)) for m in re.finditer(r"\b(NUFFT_\w+)\s*=\s*(-?\d+)", hdr)}
    ids = {m.group(1): int(m.group(2)) for m in re.finditer(r"kernel_id\(::(\w+)\)\s*=\s*(\d+)", text)}
    assert ids == {"BackwardsKaiserBesselKernel": enum["NUFFT_KERNEL_BACKWARDS_KAISER_BESSEL"], "KaiserBesselKernel": enum["NUFFT_KERNEL_KAISER_BESSEL"],
                   "GaussianKernel": enum["NUFFT_KERNEL_GAUSSIAN"], "BSplineKernel": enum["NUFFT_KERNEL_BSPLINE"]}
    # check(rc): ArgumentError for the codes the header maps to ArgumentError, DimensionMismatch for NUFFT_ERR_DIM_MISMATCH
    arg = re.search(r"rc in \(([\d, ]+)\) \? throw\(ArgumentError", text).group(1)
    assert {int(x) for x in arg.split(",")} == {enum[k] for k in ("NUFFT_ERR_INVALID_ARG", "NUFFT_ERR_SIZE_TOO_SMALL", "NUFFT_ERR_LDS_TOO_SMALL",
                                                                   "NUFFT_ERR_UNSUPPORTED", "NUFFT_ERR_NO_POINTS", "NUFFT_ERR_NO_DEVICE")}
    assert int(re.search(r"rc == (\d+) \? throw\(DimensionMismatch", text).group(1)) == enum["NUFFT_ERR_DIM_MISMATCH"]
    # evalmode / dtype / point-transform encodings used when CParams is filled
    assert enum["NUFFT_EVAL_DIRECT"] == 0 and enum["NUFFT_F64"] == 1 and enum["NUFFT_POINT_TRANSFORM_IDENTITY"] == 0
    assert "p.kernel_evalmode isa Direct ? 0 : 1" in text and "T === Float64 ? 1 : 0" in text and "pt === identity ? 0 : 1" in text


def _positional_arity(sig):
    """Number of positional parameters of a Julia signature `(a, b::T{X, Y}; kw...)` (top-level commas before `;`)."""
    depth, n, seen = 0, 0, False
    for ch in sig:
        if ch in "({[":
            depth += 1
        elif ch in ")}]":
            depth -= 1
        elif ch == ";" and depth == 0:
            break
        elif ch == "," and depth == 0:
            n += 1
            continue
        if not ch.isspace():
            seen = True
    return n + 1 if seen else 0


def _signature_after(text, pos):
    """The parenthesised argument list that starts at text[pos] == '('."""
    depth = 0
    for i in range(pos, len(text)):
        if text[i] in "({[":
            depth += 1
        elif text[i] in ")}]":
            depth -= 1
            if depth == 0:
                return text[pos + 1:i]
    raise ValueError("unbalanced signature")


def _shim_overloads():
    text = open(SHIM).read()
    out = []
    for m in re.finditer(r"^(?:function\s+)?(?:NonuniformFFTs|KA)\.(\w+!?)\(", text, flags=re.M):
        if text[m.start():m.start() + 3] == "KA.":
            continue
        out.append((m.group(1), _positional_arity(_signature_after(text, m.end() - 1))))
    return out


def test_overloaded_functions_exist_in_the_reference_with_that_arity():
    overloads = _shim_overloads()
    names = {n for n, _ in overloads}
    assert {"default_block_size", "BlockDataGPU", "init_plan_data", "set_points!", "exec_type1!", "exec_type2!", "output_field", "gpu_method",
            "default_kernel", "default_kernel_evalmode"} <= names, names
    if not os.path.isdir(os.path.join(REFERENCE, "src")):
        pytest.skip("reference checkout not present (GPU box): arity check runs in the build container")
    src = ""
    for d in ("src", "ext"):
        for dirpath, _, files in os.walk(os.path.join(REFERENCE, d)):
            for f in files:
                if f.endswith(".jl"):
                    src += open(os.path.join(dirpath, f)).read() + "\n"
    for name, arity in overloads:
        found = set()
        for m in re.finditer(r"(?:^|[\s.])%s\(" % re.escape(name), src, flags=re.M):
            # definitions only: `function name(` or `name(args...) =` / `name(args...) where`
            line_start = src.rfind("\n", 0, m.start()) + 1
            prefix = src[line_start:m.start() + 1]
            sig = _signature_after(src, m.end() - 1)
            tail = src[m.end() + len(sig):m.end() + len(sig) + 40]
            is_def = prefix.strip().startswith("function") or re.match(r"^\)\s*(where\s+[^=\n]+)?=(?!=)", tail) is not None
            if "@kernel" in prefix:
                is_def = False
            if is_def:
                found.add(_positional_arity(sig))
        assert found, f"NonuniformFFTs.{name}: no definition found in the reference source"
        assert arity in found, f"NonuniformFFTs.{name}: the shim's method takes {arity} positional arguments, the reference defines {sorted(found)}"


def test_integration_md_points_to_the_file():
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert "julia/ext/NonuniformFFTsMI355XExt.jl" in md and "test_julia_shim_static.py" in md
