#!/usr/bin/env python3
"""Generates rust/ffi.rs -- the `extern "C"` block a RustSolver maintainer links against -- from include/rustsolver_amd.h and
include/rustsolver_amd_diag.h (the job `bindgen` would do; there is no rustc / bindgen in the build image, so the output is checked by
tests/test_abi_cpu.py against the headers instead of by a compiler: names, arities, struct fields).

    python tools/gen_rust_ffi.py            # rewrites rust/ffi.rs
    python tools/gen_rust_ffi.py --check    # exit 1 if rust/ffi.rs is not what the headers generate
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADERS = [("rustsolver_amd.h", "the drop-in surface"), ("rustsolver_amd_diag.h", "diagnostics (bench / tests), not needed by a solver")]

SCALARS = {
    "int": "c_int", "unsigned": "c_uint", "unsigned int": "c_uint", "char": "c_char", "void": "c_void", "float": "f32", "double": "f64",
    "size_t": "usize", "uint8_t": "u8", "uint16_t": "u16", "uint32_t": "u32", "uint64_t": "u64", "int8_t": "i8", "int16_t": "i16",
    "int32_t": "i32", "int64_t": "i64",
}
RESERVED = {"abs": "abs_", "type": "type_", "in": "in_", "mod": "mod_", "ref": "ref_", "box": "box_", "move": "move_", "loop": "loop_"}


def strip_comments(src):
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    return re.sub(r"//[^\n]*", " ", src)


def rust_type(ctype):
    """'const float *' -> '*const f32';  'rs_table **' -> '*mut *mut rs_table';  'rs_card_abs *const *' -> '*const *mut rs_card_abs'"""
    toks = re.findall(r"\*|\w+", ctype)
    base, i, const_base = [], 0, False
    while i < len(toks) and toks[i] != "*":
        if toks[i] == "const":
            const_base = True
        elif toks[i] != "struct":
            base.append(toks[i])
        i += 1
    name = " ".join(base)
    out = SCALARS.get(name, name)
    pending_const = const_base
    while i < len(toks):
        assert toks[i] == "*", ctype
        i += 1
        out = ("*const " if pending_const else "*mut ") + out
        pending_const = False
        while i < len(toks) and toks[i] == "const":
            pending_const = True
            i += 1
    return out


def split_decl(decl):
    """'const uint32_t n_clusters[RS_MAX_ROUNDS][RS_MAX_PLAYERS]' -> (ctype, name, [dims])"""
    decl = decl.strip()
    dims = re.findall(r"\[([^\]]*)\]", decl)
    decl = re.sub(r"\[[^\]]*\]", "", decl).strip()
    m = re.match(r"^(.*?)(\w+)$", decl, flags=re.S)
    return m.group(1).strip(), m.group(2), dims


def field_type(ctype, dims):
    t = rust_type(ctype)
    for d in reversed(dims):
        t = "[%s; %s]" % (t, d if not d.isdigit() else d)
    return t


def param_type(ctype, dims):
    """an array parameter decays to a pointer to its element (of the remaining dimensions)"""
    if not dims:
        return rust_type(ctype)
    const = bool(re.search(r"\bconst\b", ctype))
    inner = field_type(re.sub(r"\bconst\b", "", ctype), dims[1:])
    return ("*const " if const else "*mut ") + inner


def parse(src):
    src = strip_comments(src)
    items = {"defines": [], "enums": [], "opaque": [], "structs": [], "fns": []}
    for m in re.finditer(r"^#define\s+(RS_\w+)\s+(-?(?:0x)?[0-9a-fA-F]+)\s*$", src, flags=re.M):
        items["defines"].append((m.group(1), m.group(2)))
    for m in re.finditer(r"enum\s*\{(.*?)\}\s*;", src, flags=re.S):
        for part in m.group(1).split(","):
            if "=" in part:
                k, v = part.split("=")
                items["enums"].append((k.strip(), v.strip()))
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s+(\w+)\s*;", src):
        items["opaque"].append(m.group(1))
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", src, flags=re.S):
        fields = []
        for line in m.group(2).split(";"):
            line = line.strip()
            if not line:
                continue
            first = split_decl(line.split(",")[0])
            fields.append((first[1], field_type(first[0], first[2])))
            for more in line.split(",")[1:]:                     # `uint32_t world, rank;`
                fields.append((more.strip(), field_type(first[0], [])))
        items["structs"].append((m.group(1), fields))
    body = re.sub(r"^[ \t]*#[^\n]*(?:\\\n[^\n]*)*$", " ", src, flags=re.M)   # preprocessor lines (#ifdef / #endif / #define, continuations included) are not part of any prototype
    body = re.sub(r"typedef\s+struct\s+\w+\s*\{.*?\}\s*\w+\s*;", " ", body, flags=re.S)
    body = re.sub(r"enum\s*\{.*?\}\s*;", " ", body, flags=re.S)
    for m in re.finditer(r"([\w\s\*]+?)\b(rs_\w+)\s*\(([^()]*)\)\s*;", body):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if ret.startswith("typedef") or not ret:
            continue
        params = []
        if args and args != "void":
            for a in args.split(","):
                ctype, pname, dims = split_decl(a)
                params.append((RESERVED.get(pname, pname), param_type(ctype, dims)))
        rt = None if ret == "void" else rust_type(ret)
        if rt is not None and not re.fullmatch(r"(?:\*(?:const|mut) )*\w+", rt):   # anything else is a parse accident (rustc would reject it; there is none here to say so)
            raise ValueError("return type of %s parsed as %r" % (name, rt))
        items["fns"].append((name, params, rt))
    return items


def generate():
    out = ["//! rust/ffi.rs -- GENERATED by tools/gen_rust_ffi.py from include/rustsolver_amd.h and include/rustsolver_amd_diag.h; do not edit.",
           "//! The binding a RustSolver maintainer adds (e.g. as src/solver/gpu.rs) to route the info-set hot path through librustsolver_amd.so;",
           "//! INTEGRATION.md shows the call sites.  There is no rustc in the build image, so this file is not compiled here: tests/test_abi_cpu.py",
           "//! checks every name, arity and struct field against the C headers, and the same ABI is exercised from C (examples/solver_main.c) and ctypes.",
           "#![allow(non_camel_case_types, non_upper_case_globals, dead_code)]",
           "use std::os::raw::{c_char, c_int, c_uint, c_void};", ""]
    seen_opaque, n_fns = set(), 0
    for header, what in HEADERS:
        it = parse(open(os.path.join(ROOT, "include", header)).read())
        out.append("// ======== %s: %s ========" % (header, what))
        for k, v in it["defines"]:
            out.append("pub const %s: usize = %s;" % (k, v))
        for k, v in it["enums"]:
            out.append("pub const %s: c_int = %s;" % (k, v))
        full = {s[0] for s in it["structs"]}
        for o in it["opaque"]:
            if o not in full and o not in seen_opaque:
                seen_opaque.add(o)
                out.append("#[repr(C)] pub struct %s { _private: [u8; 0] }" % o)
        for name, fields in it["structs"]:
            out.append("#[repr(C)] #[derive(Clone, Copy)]")
            out.append("pub struct %s {" % name)
            for f, t in fields:
                out.append("    pub %s: %s," % (RESERVED.get(f, f), t))
            out.append("}")
        out.append('#[link(name = "rustsolver_amd")]')
        out.append('extern "C" {')
        for name, params, ret in it["fns"]:
            sig = ", ".join("%s: %s" % p for p in params)
            out.append("    pub fn %s(%s)%s;" % (name, sig, " -> " + ret if ret else ""))
            n_fns += 1
        out.append("}")
        out.append("")
    return "\n".join(out), n_fns


def main():
    text, n = generate()
    path = os.path.join(ROOT, "rust", "ffi.rs")
    if "--check" in sys.argv:
        ok = os.path.exists(path) and open(path).read() == text
        print("rust/ffi.rs %s (%d functions)" % ("is up to date" if ok else "is STALE: run python tools/gen_rust_ffi.py", n))
        sys.exit(0 if ok else 1)
    with open(path, "w") as f:
        f.write(text)
    print("wrote %s: %d functions" % (path, n))


if __name__ == "__main__":
    main()
