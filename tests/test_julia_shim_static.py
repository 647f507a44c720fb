"""Static cross-check of the Julia package extension (julia/ext/NonuniformFFTsMI355XExt.jl) against the C header and the
reference source — there is no Julia runtime in the build image, so nothing else would notice a drifted `ccall`, a call of a
function the reference does not have, a wrong enum encoding or a constructor with the wrong number of arguments.

Checked against include/nufft_mi355x.h:
  * every `ccall((:sym, libnufft), Ret, (ArgTypes...), ...)`: symbol, return type, argument count and types;
  * `struct CParams` / `struct CCallbacks` against `nufft_params` / `nufft_callbacks` (field names, order, types, size);
  * every `const NUFFT_X = Int32(n)` against the header's enum, every `NUFFT_X` token the file uses has such a constant, and the enum
    slots of the `CParams(...)` call are filled from those constants only (no integer literal can encode an enum);
Checked against the reference source (/root/reference; skipped where the checkout is absent):
  * every `NonuniformFFTs.f(...)` / `Kernels.f(...)` the file CALLS: a definition with that positional arity exists;
    every `AbstractNFFTs.f` it uses is one the reference itself uses;
  * every field it reads from a reference struct (kernel data, PlanNUFFT, NUFFTCallbacks, the fold closure's capture);
  * every method it ADDS to a reference function repeats one reference signature in every positional slot except those it narrows
    to its own types (MI355X...), including the bounds of the type variables — so the method is strictly more specific than
    the reference's and cannot be ambiguous with it;
Checked within the file: every constructor call of its own structs passes as many arguments as the struct has fields."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "julia", "ext", "NonuniformFFTsMI355XExt.jl")
HEADER = os.path.join(ROOT, "include", "nufft_mi355x.h")
REFERENCE = "/root/reference"

needs_reference = pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "src")),
                                     reason="reference checkout not present (GPU box): runs in the build container")


def _strip_c_comments(text):
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def _strip_julia_comments(text):
    """Drop `# ...` comments (not inside string literals; good enough for this file and the reference's style)."""
    out = []
    for line in text.split("\n"):
        in_str, i, cut = False, 0, len(line)
        while i < len(line):
            ch = line[i]
            if ch == '"' and (i == 0 or line[i - 1] != "\\"):
                in_str = not in_str
            elif ch == "#" and not in_str:
                cut = i
                break
            i += 1
        out.append(line[:cut])
    return "\n".join(out)


def _shim_text():
    return _strip_julia_comments(open(SHIM).read())


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


def _header_enums():
    hdr = _strip_c_comments(open(HEADER).read())
    out = {m.group(1): int(m.group(2)) for m in re.finditer(r"\b(NUFFT_\w+)\s*=\s*(-?\d+)", hdr)}
    # ... and the ABI version macro: the shim refuses an older library
    out.update({m.group(1): int(m.group(2)) for m in re.finditer(r"#define\s+(NUFFT_MI355X_VERSION)\s+(\d+)", open(HEADER).read())})
    return out


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
    assert {"nufft_plan_create_ex", "nufft_plan_destroy", "nufft_set_points", "nufft_set_callbacks", "nufft_spread_deferred", "nufft_fft_forward",
            "nufft_deconvolve_truncate", "nufft_deconvolve_pad", "nufft_fft_backward", "nufft_interpolate",
            "nufft_sizeof_params", "nufft_version", "nufft_last_error_message"} <= {c[0] for c in calls}
    # a ccall's (symbol, library) pair must be a literal: none through a variable
    assert not re.search(r"ccall\(\(\s*[a-z]\w*\s*,", open(SHIM).read())
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


def _julia_struct(name, text=None):
    """[(field, type)] of `struct name ... end` (one-line `struct A; x::T; y::U; end` or multi-line)."""
    text = _shim_text() if text is None else text
    m = re.search(r"struct %s\b([^\n;]*)[\n;](.*?)(?:^|[\n;])\s*end\b" % re.escape(name), text, flags=re.S)
    assert m, f"struct {name} not found"
    body = m.group(2)
    # drop inner constructors (function ... end blocks) of reference structs
    body = re.sub(r"\n\s*function\b.*", "", body, flags=re.S)
    fields = []
    for item in re.split(r"[;\n]", body):
        item = item.strip()
        if not item or "::" not in item:
            continue
        fname, ftype = [x.strip() for x in item.split("::", 1)]
        if re.match(r"^\w+$", fname):
            fields.append((fname, ftype))
    return fields


_CT = {"int32_t": ("Int32", C.c_int32), "int64_t": ("Int64", C.c_int64), "double": ("Float64", C.c_double),
       "const void*": ("Ptr{Cvoid}", C.c_void_p), "const char*": ("Ptr{UInt8}", C.c_char_p)}


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


# ---- constants and encodings -------------------------------------------------------------------------------------------

def _shim_constants():
    return {m.group(1): int(m.group(2)) for m in re.finditer(r"^const (NUFFT_\w+) = Int32\((-?\d+)\)", _shim_text(), flags=re.M)}


def test_every_constant_equals_the_header_enum_and_every_use_is_a_constant():
    enum, consts, text = _header_enums(), _shim_constants(), _shim_text()
    assert len(consts) >= 18
    for name, value in consts.items():
        assert name in enum, f"{name}: not an enumerator of include/nufft_mi355x.h"
        assert enum[name] == value, f"{name} = {value} in the shim, {enum[name]} in the header"
    used = set(re.findall(r"\bNUFFT_[A-Z0-9_]+\b", text))
    assert used <= set(consts), f"used but not defined: {sorted(used - set(consts))}"
    # the families a binding needs are complete
    for fam in ("NUFFT_KERNEL_", "NUFFT_EVAL_", "NUFFT_METHOD_", "NUFFT_POINT_TRANSFORM_"):
        assert {k for k in enum if k.startswith(fam)} == {k for k in consts if k.startswith(fam)}, fam
    # check(rc): ArgumentError for the codes the header maps to ArgumentError, DimensionMismatch for NUFFT_ERR_DIM_MISMATCH
    arg = re.search(r"rc in \(([\w, ]+)\) \?\s*throw\(ArgumentError", text).group(1)
    assert {x.strip() for x in arg.split(",")} == {"NUFFT_ERR_INVALID_ARG", "NUFFT_ERR_SIZE_TOO_SMALL", "NUFFT_ERR_LDS_TOO_SMALL",
                                                   "NUFFT_ERR_UNSUPPORTED", "NUFFT_ERR_NO_POINTS", "NUFFT_ERR_NO_DEVICE"}
    assert re.search(r"rc == NUFFT_ERR_DIM_MISMATCH \? throw\(DimensionMismatch", text)


def _split_top(s, sep=","):
    """Split on `sep` at bracket depth 0 (strings are not expected to contain brackets here)."""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "({[":
            depth += 1
        elif ch in ")}]":
            depth -= 1
        if ch == sep and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    out.append(cur)
    return [x.strip() for x in out]


def _balanced(text, pos):
    """The bracketed group that starts at text[pos] (one of ( { [), without the brackets."""
    depth = 0
    for i in range(pos, len(text)):
        if text[i] in "({[":
            depth += 1
        elif text[i] in ")}]":
            depth -= 1
            if depth == 0:
                return text[pos + 1:i]
    raise ValueError("unbalanced")


def test_enum_slots_of_cparams_are_filled_from_constants_only():
    text = _shim_text()
    fields = [f for f, _ in _julia_struct("CParams")]
    m = re.search(r"prm = CParams\(", text)
    args = _split_top(_balanced(text, m.end() - 1))
    assert len(args) == len(fields), (len(args), len(fields))
    byname = dict(zip(fields, args))
    family = {"dtype": "NUFFT_F", "evalmode": "NUFFT_EVAL_", "point_transform": "NUFFT_POINT_TRANSFORM_", "gpu_method": "NUFFT_METHOD_"}
    for field, fam in family.items():
        var = byname[field]
        assert re.match(r"^[a-z]\w*$", var), f"CParams.{field} is filled with `{var}`: expected a local built from {fam}* constants"
        rhs = re.search(r"^\s*%s = (.*)$" % re.escape(var), text, flags=re.M).group(1)
        branches = re.search(r"\?\s*(\w+)\s*:\s*(\w+)\s*$", rhs)
        assert branches, (field, rhs)
        for b in branches.groups():
            assert b.startswith(fam), f"CParams.{field}: `{b}` is not a {fam}* constant"
    assert byname["spread_method"] == "NUFFT_SPREAD_AUTO"
    assert byname["kernel"].startswith("kernel_id(")
    # ABI 104: the caller states how much of the struct it knows; no development switches from a Julia process
    assert byname["struct_size"] == "Int32(sizeof(CParams))" and byname["reserved"] == "Int32(0)" and byname["options"] == "Ptr{UInt8}(C_NULL)"
    # the host-only probe of BlockDataGPU (plan-time errors at plan time): same struct, every enum slot a constant or a ternary of
    # constants of the right family, device = -1 (no GPU call), the reference's own Ñs forwarded
    mp = re.search(r"probe = CParams\(", text)
    pargs = _split_top(_balanced(text, mp.end() - 1))
    assert len(pargs) == len(fields), (len(pargs), len(fields))
    pb = dict(zip(fields, pargs))
    for field, fam in dict(family, kernel="NUFFT_KERNEL_", spread_method="NUFFT_SPREAD_").items():
        toks = re.findall(r"NUFFT_\w+", pb[field])
        assert toks and all(tk.startswith(fam) for tk in toks), (field, pb[field])
        assert not re.search(r"(?<![\w.])\d+(?![\w.])", re.sub(r"NUFFT_\w+", "", pb[field]).replace("Float64", "")), (field, pb[field])
    assert pb["device"] == "Int32(-1)" and pb["struct_size"] == "Int32(sizeof(CParams))" and pb["options"] == "Ptr{UInt8}(C_NULL)"
    assert "Ñs[d]" in pb["N_over"] and pb["half_support"] == "Int32(M)" and pb["ndim"] == "Int32(D)"
    assert re.search(r"probe_parameters\(Z, Ñs, Val\(M\), method\)", text)
    # the semantic pairing of each ternary: Float64 -> F64, Direct -> DIRECT, identity -> IDENTITY, :shared_memory -> SHARED_MEMORY
    assert re.search(r"dtype = T === Float64 \? NUFFT_F64 : NUFFT_F32", text)
    assert re.search(r"evalmode = p\.kernel_evalmode isa Direct \? NUFFT_EVAL_DIRECT : NUFFT_EVAL_FAST_APPROXIMATION", text)
    assert re.search(r"ptrans = pt === identity \? NUFFT_POINT_TRANSFORM_IDENTITY : NUFFT_POINT_TRANSFORM_NFFT", text)
    assert re.search(r"=== :shared_memory \? NUFFT_METHOD_SHARED_MEMORY : NUFFT_METHOD_GLOBAL_MEMORY", text)
    ids = dict(re.findall(r"kernel_id\(::AbstractKernelData\{(\w+)\}\) = (NUFFT_KERNEL_\w+)", text))
    assert ids == {"BackwardsKaiserBesselKernel": "NUFFT_KERNEL_BACKWARDS_KAISER_BESSEL", "KaiserBesselKernel": "NUFFT_KERNEL_KAISER_BESSEL",
                   "GaussianKernel": "NUFFT_KERNEL_GAUSSIAN", "BSplineKernel": "NUFFT_KERNEL_BSPLINE"}
    # the library accepts both gpu_method encodings and the per-dimension fields this call fills (host-only plan, no GPU)
    from nufft_pkg import nufft
    import numpy as np
    for method in ("shared_memory", "global_memory"):
        q = nufft.PlanNUFFT(np.float64, (16, 12), gpu_method=method, backend=None)
        assert q.shape == (12, 9)          # torch order: reversed (N₁÷2+1, N₂)


# ---- constructor calls of the shim's own structs --------------------------------------------------------------------------

def _positional_arity(sig):
    """(min, max) number of positional parameters / arguments of `(a, b::T{X, Y} = 1, c...; kw...)`; max = None for varargs."""
    pos = _split_top(sig.split(";")[0] if ";" not in _strip_nested(sig) else _split_top(sig, ";")[0])
    pos = [a for a in pos if a]
    lo = hi = 0
    for a in pos:
        if a.endswith("..."):
            return lo, None
        hi += 1
        # a default value: `=` at depth 0 that is not part of `==`, `<=`, `=>`
        depth, has_default = 0, False
        for i, ch in enumerate(a):
            if ch in "({[":
                depth += 1
            elif ch in ")}]":
                depth -= 1
            elif ch == "=" and depth == 0 and a[i - 1] not in "=<>!" and a[i + 1:i + 2] not in ("=", ">"):
                has_default = True
        if not has_default:
            lo = hi
    return lo, hi


def _strip_nested(s):
    out, depth = "", 0
    for ch in s:
        if ch in "({[":
            depth += 1
        elif ch in ")}]":
            depth -= 1
        elif depth == 0:
            out += ch
    return out


def _call_argc(args):
    lo, hi = _positional_arity(args)
    return hi if hi is not None else lo


def test_own_struct_constructors_get_as_many_arguments_as_fields():
    text = _shim_text()
    structs = re.findall(r"^(?:mutable )?struct (\w+)", text, flags=re.M)
    assert {"MI355XBackend", "MI355XBlockData", "MI355XData", "SpectrumShape", "CParams", "CCallbacks", "PointWeights", "ModeFactors"} <= set(structs)
    checked = 0
    for name in structs:
        nfields = len(_julia_struct(name, text))
        for m in re.finditer(r"(?<![\w.])%s(\{[^()\n]*\})?\(" % name, text):
            line_start = text.rfind("\n", 0, m.start()) + 1
            prefix = text[line_start:m.start()]
            args = _balanced(text, m.end() - 1)
            tail = text[m.end() + len(args) + 1:m.end() + len(args) + 4]
            if re.search(r"\bstruct\s*$", prefix) or "function" in prefix or re.match(r"^\s*=(?!=)", tail) or prefix.strip().startswith("("):
                continue        # a definition (struct header, outer constructor `Name(...) = ...`, functor method `(c::Name)(...)`)
            if "::" in args and name in ("PointWeights", "ModeFactors"):
                continue        # `(c::PointWeights)(v, n)` functor definitions
            argc = _call_argc(args)
            if name == "MI355XBackend" and argc == 0:
                continue        # the outer convenience constructor MI355XBackend() defined in the file
            assert argc == nfields, f"{name}({args}) passes {argc} arguments, the struct has {nfields} fields"
            checked += 1
    assert checked >= 6, checked


# ---- names the shim calls in the reference ---------------------------------------------------------------------------------

def _reference_source():
    src = ""
    for d in ("src", "ext"):
        for dirpath, _, files in os.walk(os.path.join(REFERENCE, d)):
            for f in sorted(files):
                if f.endswith(".jl"):
                    src += _strip_julia_comments(open(os.path.join(dirpath, f)).read()) + "\n"
    return src


def _definitions(src, name):
    """[(signature text, where text)] of every method definition of `name` in `src` (long and short form)."""
    out = []
    for m in re.finditer(r"(?<![\w.!])(?:\w+\.)*%s(?:\{[^()\n]*\})?\(" % re.escape(name), src):
        line_start = src.rfind("\n", 0, m.start()) + 1
        prefix = src[line_start:m.start()]
        if "@kernel" in prefix:
            continue
        sig = _balanced(src, m.end() - 1)
        after = src[m.end() + len(sig) + 1:m.end() + len(sig) + 200]
        long_form = re.search(r"\bfunction\s*$", prefix) is not None
        short = re.match(r"^\s*(where\s*(\{[^}]*\}|\w+)\s*)?=(?![=>])", after)
        if not (long_form or short):
            continue
        w = re.match(r"^\s*where\s*(\{[^}]*\}|\w+)", after)
        out.append((sig, w.group(1) if w else ""))
    return out


def _struct_arity(src, name):
    m = re.search(r"^(?:mutable )?struct %s\b" % re.escape(name), src, flags=re.M)
    if not m:
        return None
    try:
        return len(_julia_struct(name, src))
    except AssertionError:
        return None


def _shim_calls(prefixes):
    """[(module, name, args)] of qualified calls `Mod.name(args)` that are not method definitions."""
    text = _shim_text()
    out = []
    for m in re.finditer(r"(?<![\w.])(%s)\.(\w+!?)\(" % "|".join(prefixes), text):
        line_start = text.rfind("\n", 0, m.start()) + 1
        prefix = text[line_start:m.start()]
        args = _balanced(text, m.end() - 1)
        after = text[m.end() + len(args) + 1:m.end() + len(args) + 120]
        if re.search(r"\bfunction\s*$", prefix) or (prefix.strip() == "" and re.match(r"^\s*(where\s*(\{[^}]*\}|\w+)\s*)?=(?![=>])", after)):
            continue
        out.append((m.group(1), m.group(2), args))
    return out


@needs_reference
def test_every_reference_function_the_shim_calls_exists_with_that_arity():
    src = _reference_source()
    calls = _shim_calls(["NonuniformFFTs", "Kernels"])
    names = {c[1] for c in calls}
    assert {"get_timer_nowarn", "maybe_synchronise", "check_nufft_uniform_data", "check_nufft_nonuniform_data", "gridsize", "gridstep",
            "gpu_method", "get_batch_size", "default_block_size", "convert_window_function", "_split_accuracy_params", "NFFTPlan",
            "default_kernel"} <= names, names
    own = _shim_text()      # methods the file adds itself (NFFTPlan with the backend in front) count as definitions too
    for mod, name, args in calls:
        argc = _call_argc(args)
        ranges = [_positional_arity(sig) for sig, _ in _definitions(src, name)]
        ranges += [_positional_arity(sig) for sig, _ in _definitions(own, name)] if _definitions(src, name) else []
        sa = _struct_arity(src, name)
        if sa is not None:
            ranges.append((sa, sa))
        assert ranges, f"{mod}.{name}: no definition in the reference source"
        assert any(lo <= argc and (hi is None or argc <= hi) for lo, hi in ranges), \
            f"{mod}.{name} called with {argc} positional arguments; the reference defines {sorted(set(ranges), key=str)}"
    # AbstractNFFTs is a dependency outside the checkout: only names the reference itself uses
    for _, name, _ in _shim_calls(["AbstractNFFTs"]):
        assert re.search(r"AbstractNFFTs\.%s(?![\w!])|using AbstractNFFTs:[^\n]*\b%s(?![\w!])" % (re.escape(name), re.escape(name)), src), name
    # non-call uses of internal names
    text = _shim_text()
    for name in re.findall(r"NonuniformFFTs\.(_\w+)\b(?!\()", text):
        assert _definitions(src, name), name
    for name in ("AbstractBlockData", "AbstractNUFFTData", "AbstractKernelData", "NUFFTCallbacks", "default_callback", "StaticBool", "HalfSupport"):
        assert re.search(r"\b%s\b" % name, src) and re.search(r"\b%s\b" % name, text), name


@needs_reference
def test_every_field_read_from_a_reference_struct_exists():
    src, text = _reference_source(), _shim_text()
    plan_fields = {f for f, _ in _julia_struct("PlanNUFFT", src)} | {"points", "timer"}      # getproperty, src/plan.jl:412-420
    used = set(re.findall(r"\bp\.(\w+)", text))
    assert {"data", "kernels", "blocks", "points_ref", "kernel_evalmode", "fftshift", "point_transform_fold", "σ"} <= used
    assert used <= plan_fields, used - plan_fields
    cb_fields = {f for f, _ in _julia_struct("NUFFTCallbacks", src)}
    assert set(re.findall(r"\bcb\.(\w+)", text)) <= cb_fields
    # shape parameters: the field each shape_param method reads is a field of that kernel's data struct
    reads = re.findall(r"shape_param\(g::AbstractKernelData\{(\w+)\}\) = Float64\(g\.(\w+)", text)
    assert len(reads) == 3
    for kernel, field in reads:
        fields = {f for f, _ in _julia_struct(kernel + "Data", src)}
        assert field in fields, (kernel, field, fields)
    # ... and the kernel is the first parameter of AbstractKernelData, with these four concrete kernels
    assert re.search(r"abstract type AbstractKernelData\{K <: AbstractKernel, M, T <: AbstractFloat\}", src)
    for kernel in ("BackwardsKaiserBesselKernel", "KaiserBesselKernel", "GaussianKernel", "BSplineKernel"):
        assert re.search(r"<: AbstractKernelData\{%s, M, T\}" % kernel, src), kernel
    # the fold closure captures a variable called point_transform (src/plan.jl:459-464)
    m = re.search(r"function generate_point_transform_fold_function\(([^)]*)\)(.*?)\nend", src, flags=re.S)
    assert "point_transform::F" in m.group(1) and "point_transform(x)" in m.group(2)
    assert "fold.point_transform" in text
    # output_field returns a tuple in the reference (`first(output_field(...))`, src/plan.jl:531) and in the shim
    assert re.search(r"first\(output_field\(nufft_data\)\)", src)
    assert re.search(r"shape::NTuple\{Nc, SpectrumShape\{T, N\}\}", text) and "output_field(data::MI355XData) = data.shape" in text
    # the storage-free spectrum has the oversampled spectrum's size, as the reference's arrays (src/plan.jl:43,55)
    assert "(Ñs[1] ÷ 2 + 1, Base.tail(Ñs)...)" in text and "dims_out = (Ñs[1] ÷ 2 + 1, Base.tail(Ñs)...)" in src


def _slots(sig):
    """Normalised positional type annotations of a signature: `a::T` -> `T`, `::T` -> `T`, `a` -> `Any`, defaults dropped."""
    pos = _split_top(_split_top(sig, ";")[0])
    out = []
    for a in [x for x in pos if x]:
        a = re.split(r"(?<![=<>!])=(?![=>])", a)[0].strip()
        t = a.split("::", 1)[1] if "::" in a else "Any"
        out.append(re.sub(r"\s+", "", t))
    return out


def _where(w):
    w = w.strip()
    if w.startswith("{"):
        w = w[1:-1]
    out = {}
    for item in _split_top(w):
        if not item:
            continue
        mm = re.match(r"^(\w+)\s*(?:<:\s*(.*))?$", item)
        out[mm.group(1)] = re.sub(r"\s+", "", mm.group(2) or "Any")
    return out


@needs_reference
def test_every_overload_repeats_a_reference_signature_except_where_it_narrows():
    src, text = _reference_source(), _shim_text()
    overloads = []
    for m in re.finditer(r"^(?:function\s+)?NonuniformFFTs\.(\w+!?)\(", text, flags=re.M):
        sig = _balanced(text, m.end() - 1)
        after = text[m.end() + len(sig) + 1:m.end() + len(sig) + 200]
        w = re.match(r"^\s*where\s*(\{[^}]*\}|\w+)", after)
        overloads.append((m.group(1), sig, w.group(1) if w else ""))
    names = [o[0] for o in overloads]
    assert {"default_kernel", "default_kernel_evalmode", "BlockDataGPU", "init_plan_data", "output_field", "gpu_method", "with_blocking",
            "get_batch_size", "set_points!", "exec_type1!", "exec_type2!", "NFFTPlan"} <= set(names), names
    assert names.count("init_plan_data") == 2        # one per reference method: a single ::Type{Z} method would be ambiguous with both
    matched_refs = []
    for name, sig, w in overloads:
        mine, mine_w = _slots(sig), _where(w)
        if name == "NFFTPlan":                       # the reference's constructor with the backend in front
            assert "MI355X" in mine[0]
            mine = mine[1:]
        ok = False
        for rsig, rw in _definitions(src, name):
            theirs, theirs_w = _slots(rsig), _where(rw)
            if len(theirs) != len(mine):
                continue
            good = True
            for a, b in zip(mine, theirs):
                if "MI355X" in a:
                    # narrowed slot: the reference's annotation must be a prefix pattern of it (PlanNUFFT{Z,N} -> PlanNUFFT{Z,N,Nc,M,MI355XBackend})
                    head = re.match(r"^[\w.]+", b).group(0) if b != "Any" else ""
                    if head in ("PlanNUFFT",) and not a.startswith(b.rstrip("}")):
                        good = False
                    continue
                if a != b:
                    good = False
                    break
                for tv in re.findall(r"\b[A-Z]\w*\b", a):        # bounds of the type variables used in a repeated slot
                    if tv in mine_w or tv in theirs_w:
                        if mine_w.get(tv, "Any") != theirs_w.get(tv, "Any"):
                            good = False
            if good:
                ok = True
                matched_refs.append((name, rsig))
                break
        assert ok, f"NonuniformFFTs.{name}({sig}): no reference method with the same slots; reference has {[_slots(s) for s, _ in _definitions(src, name)]}"
    # the two init_plan_data methods mirror two different reference methods
    assert len({r for n, r in matched_refs if n == "init_plan_data"}) == 2


def test_integration_md_points_to_the_file_and_walks_the_constructor():
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert "julia/ext/NonuniformFFTsMI355XExt.jl" in md and "test_julia_shim_static.py" in md
    # the commented walk of _PlanNUFFT names every call of src/plan.jl:467-541 that involves the backend
    for needle in ("default_kernel_evalmode", "default_block_size", "default_gpu_batch_size", "optimal_kernel", "init_fourier_coefficients!",
                   "KA.allocate", "BlockDataGPU", "init_plan_data", "output_field", "non_oversampled_indices!",
                   "generate_point_transform_fold_function", "to_unit_cell"):
        assert needle in md, needle


@needs_reference
def test_stage_timer_labels_are_the_references():
    """exec_type1! / exec_type2! nest the reference's own TimerOutputs labels, in its order, each stage followed by maybe_synchronise(p)
    (src/NonuniformFFTs.jl:157-186, 246-283): p.timer shows the same tree whichever backend runs (missing 2)."""
    ref = open(os.path.join(REFERENCE, "src", "NonuniformFFTs.jl")).read()
    shim = _shim_text()

    def labels(src, fname, first_arg):
        m = re.search(r"function (?:NonuniformFFTs\.)?%s\(\s*\S+::NTuple" % re.escape(fname), src)      # (the tuple form: the one with the stages)
        assert m, fname
        end = re.search(r"\n    \S+\nend\n", src[m.start():])
        body = src[m.start():m.start() + end.end()]
        return re.findall(r'@timeit timer "([^"]+)"', body), body

    for fname, first in (("exec_type1!", None), ("exec_type2!", None)):
        want, _ = labels(ref, fname, first)
        got, body = labels(shim, fname, first)
        assert len(want) == 5 and got == want, (fname, got, want)
        # one stage call and one maybe_synchronise per label below the outer one
        assert body.count("NonuniformFFTs.maybe_synchronise(p)") == 4, fname
        assert "set_callbacks(h, cb)" in body and "set_callbacks(h, Ref(no_callbacks))" in body
    b1 = labels(shim, "exec_type1!", None)[1]
    assert re.search(r'"\(1\) Spreading" begin\s+spread_deferred\(', b1) and re.search(r'"\(2\) Forward FFT" begin\s+fft_forward\(', b1)
    assert re.search(r'"\(3\) Deconvolution" begin\s+deconvolve_truncate\(', b1)
    b2 = labels(shim, "exec_type2!", None)[1]
    assert re.search(r'"\(1\) Deconvolution" begin\s+deconvolve_pad\(', b2) and re.search(r'"\(2\) Backward FFT" begin\s+fft_backward\(', b2)
    assert re.search(r'"\(3\) Interpolation" begin\s+interpolate_points\(', b2)
    # the stage sequence IS nufft_exec_type1 / nufft_exec_type2 (csrc/plan.cpp): nothing is lost by not calling them
    plan = open(os.path.join(ROOT, "nonuniformffts.jl_amd", "csrc", "plan.cpp")).read()
    e1 = plan[plan.index("int nufft_exec_type1(nufft_plan* p"):plan.index("// Fused callback menu")]
    assert re.findall(r"nufft_(\w+)\(p,", e1) == ["spread_deferred", "fft_forward", "deconvolve_truncate"]
    e2 = plan[plan.index("int nufft_exec_type2(nufft_plan* p"):plan.index("int nufft_grid_ptr(")]
    assert re.findall(r"nufft_(\w+)\(p,", e2) == ["deconvolve_pad", "fft_backward", "interpolate"]
