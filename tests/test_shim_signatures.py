"""CPU: the three statements of the drop-in boundary must say the same thing, argument by argument.

  * include/nbody_mi355x.h            the C prototypes (level 1 = the six reference symbols, level 2 = nbx_*)
  * integration/rs-shim/nbody.rs      the Rust shim a maintainer adds (cannot be compiled here: no rustc): its `extern "C"`
                                      imports of level 2 and its six `#[no_mangle] pub extern fn nb_*` exports (the exports
                                      rs-src/nbody.rs:34-35, :39-40, :73-74, :106-107, :186-187, :482-483 must keep)
  * INTEGRATION.md                    the Haskell imports quoted in the symbol map (hs-src/RustNBodyExperiment.hs:101-106);
                                      when /root/reference is present (this container) they are also checked against that file
  * nm -D libnbody_mi355x.so          what the library really exports

Every signature is reduced to a tuple of ABI classes (i32, f32, u64, ptr(const|mut, pointee), void) and compared.  The test
fails if any ONE argument type, pointer constness, arity or return type is changed in the shim, the header or the quoted
imports -- the last test mutates each statement in turn to prove that."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "nbody_mi355x.h")
SHIM = os.path.join(ROOT, "integration", "rs-shim", "nbody.rs")
INTEGRATION = os.path.join(ROOT, "INTEGRATION.md")
LIB = os.path.join(ROOT, "rust-exp_amd", "lib", "libnbody_mi355x.so")
HS = "/root/reference/hs-src/RustNBodyExperiment.hs"

LEVEL1 = ["nb_num_particles", "nb_random_disk", "nb_stable_orbits", "nb_step_brute_force", "nb_step_barnes_hut", "nb_draw"]


# ---- C header ------------------------------------------------------------------------------------------------------------
C_SCALARS = {"int32_t": "i32", "int": "i32", "uint32_t": "u32", "float": "f32", "double": "f64", "int64_t": "i64",
             "uint64_t": "u64", "size_t": "usize", "char": "i8", "void": "void"}


def c_type(text, has_name):
    """ABI class of a C declarator: `const float *px` -> ptr(const, f32).  has_name: the last identifier is a parameter name."""
    toks = text.replace("*", " * ").split()
    toks = [t for t in toks if t != "struct"]
    if has_name and len([t for t in toks if t not in ("*", "const")]) >= 2:
        assert re.match(r"[A-Za-z_]\w*$", toks[-1]), text
        toks = toks[:-1]
    stars = toks.count("*")
    const = "const" in toks
    base_toks = [t for t in toks if t not in ("*", "const")]
    assert len(base_toks) == 1, (text, toks)
    kind = C_SCALARS.get(base_toks[0], "opaque:" + base_toks[0])
    for level in range(stars):
        kind = ("ptr", "const" if (const and level == 0) else "mut", kind)
    return kind


def parse_header(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    text = "\n".join(line for line in text.splitlines() if not line.lstrip().startswith("#"))
    text = re.sub(r'extern\s+"C"\s*\{', " ", text)
    protos = {}
    for stmt in text.split(";"):
        stmt = " ".join(stmt.split())
        m = re.match(r"^([A-Za-z_][\w\s\*]*?[\s\*])(nbx?_\w+)\s*\(([^(){}]*)\)$", stmt)
        if not m or "typedef" in m.group(1) or "{" in stmt:
            continue
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        params = [] if args in ("", "void") else [c_type(a, True) for a in args.split(",")]
        protos[name] = (c_type(ret, False), params)
    return protos


# ---- Rust shim -----------------------------------------------------------------------------------------------------------
RS_SCALARS = {"i32": "i32", "u32": "u32", "f32": "f32", "f64": "f64", "i64": "i64", "u64": "u64", "usize": "usize",
              "c_char": "i8", "c_void": "void", "u8": "u8"}


def rs_type(text):
    t = text.strip()
    m = re.match(r"\*(const|mut)\s+(.*)$", t)
    if m:
        inner = rs_type(m.group(2))
        return ("ptr", m.group(1), inner)
    if t in RS_SCALARS:
        return RS_SCALARS[t]
    return "opaque:" + {"NbxEngine": "nbx_engine"}.get(t, t)


def split_args(args):
    out, depth, cur = [], 0, ""
    for ch in args:
        if ch in "(<[":
            depth += 1
        if ch in ")>]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return out


def parse_shim(text):
    text = re.sub(r"//[^\n]*", " ", text)
    block = re.search(r'extern\s+"C"\s*\{(.*?)\n\}', text, flags=re.S)
    assert block, "no extern \"C\" block in the shim"
    imports = {}
    for m in re.finditer(r"fn\s+(\w+)\s*\((.*?)\)\s*(?:->\s*([^;]+))?;", block.group(1), flags=re.S):
        params = [rs_type(a.split(":", 1)[1]) for a in split_args(m.group(2))]
        imports[m.group(1)] = (rs_type(m.group(3)) if m.group(3) else "void", params)
    exports = {}
    for m in re.finditer(r"#\[no_mangle\]\s*pub\s+extern\s+fn\s+(\w+)\s*\((.*?)\)\s*(?:->\s*([\w\*\s]+?))?\s*\{", text, flags=re.S):
        params = [rs_type(a.split(":", 1)[1]) for a in split_args(m.group(2))]
        exports[m.group(1)] = (rs_type(m.group(3)) if m.group(3) else "void", params)
    link = re.search(r'#\[link\(name\s*=\s*"(\w+)"\)\]', text)
    return imports, exports, (link.group(1) if link else None)


# ---- Haskell imports -----------------------------------------------------------------------------------------------------
HS_TYPES = {"CInt": "i32", "CFloat": "f32", "CUInt": "u32", "CDouble": "f64"}


def hs_signature(sig):
    parts = [p.strip() for p in sig.split("->")]
    last = parts[-1]
    assert last.startswith("IO"), sig
    ret = last[2:].strip().strip("()").strip()
    ret = "void" if ret == "" else HS_TYPES[ret]
    params = []
    for p in parts[:-1]:
        m = re.match(r"Ptr\s+(\w+)$", p)
        if m:
            params.append(("ptr", "mut", {"Word32": "u32", "CFloat": "f32"}[m.group(1)]))
        else:
            params.append(HS_TYPES[p])
    return ret, params


def parse_hs_imports(text):
    out = {}
    for m in re.finditer(r'foreign import ccall "(\w+)"\s+\w+\s*::\s*([^\n]+)', text):
        out[m.group(1)] = hs_signature(m.group(2).strip())
    return out


def parse_integration_symbol_map(text):
    out = {}
    for m in re.finditer(r"^\|\s*`(nb_\w+)\s*::\s*([^`]+)`\s*\|[^|]*\|\s*`([^`]+)`\s*\|", text, flags=re.M):
        out[m.group(1)] = (hs_signature(m.group(2).strip()), m.group(3).strip())
    return out


def without_constness(sig):
    def strip(t):
        return ("ptr", "any", strip(t[2])) if isinstance(t, tuple) else t
    return strip(sig[0]), [strip(p) for p in sig[1]]


# ---- the checks ----------------------------------------------------------------------------------------------------------
def check_all(header_text, shim_text, integration_text, hs_text=None, exported=None):
    """Raises AssertionError on the first disagreement; returns what it compared."""
    protos = parse_header(header_text)
    imports, exports, link = parse_shim(shim_text)
    assert link == "nbody_mi355x", link
    # 1. every level-2 function the shim imports exists in the header with the same ABI, constness of pointers included
    assert imports, "the shim imports nothing"
    for name, sig in imports.items():
        assert name in protos, f"shim imports {name}, which the header does not declare"
        assert sig == protos[name], f"{name}: shim {sig} != header {protos[name]}"
    # 2. the shim exports exactly the six reference symbols, with the header's level-1 signatures
    assert sorted(exports) == sorted(LEVEL1), sorted(exports)
    for name in LEVEL1:
        assert name in protos, name
        assert exports[name] == protos[name], f"{name}: shim export {exports[name]} != header {protos[name]}"
    # 3. the Haskell imports quoted in INTEGRATION.md are the header's level 1 (Haskell's Ptr has no constness)
    table = parse_integration_symbol_map(integration_text)
    assert sorted(table) == sorted(LEVEL1), sorted(table)
    for name, (hs_sig, c_text) in table.items():
        assert without_constness(hs_sig) == without_constness(protos[name]), f"{name}: quoted Haskell import {hs_sig} != header {protos[name]}"
        quoted = parse_header(c_text + ";")
        assert quoted.get(name) == protos[name], f"{name}: prototype quoted in INTEGRATION.md {quoted.get(name)} != header {protos[name]}"
    # 3b. the copy of the shim printed in INTEGRATION.md (option B) declares what the shim file declares
    block = re.search(r"```rust\n(.*?)```", integration_text, flags=re.S)
    assert block, "INTEGRATION.md no longer prints the shim"
    doc_imports, doc_exports, doc_link = parse_shim(block.group(1))
    assert (doc_imports, doc_exports, doc_link) == (imports, exports, link), "INTEGRATION.md's copy of the shim differs from integration/rs-shim/nbody.rs"
    # 4. ... and the reference's own file, when it is here
    if hs_text is not None:
        live = parse_hs_imports(hs_text)
        assert sorted(live) == sorted(LEVEL1), sorted(live)
        for name in LEVEL1:
            assert live[name] == table[name][0], f"{name}: RustNBodyExperiment.hs says {live[name]}, INTEGRATION.md quotes {table[name][0]}"
    # 5. the library exports every symbol the shim imports and the six it replaces
    if exported is not None:
        for name in list(imports) + LEVEL1:
            assert name in exported, f"{name} is not exported by libnbody_mi355x.so"
    return protos, imports, exports


def _read(path):
    with open(path) as f:
        return f.read()


def _exported():
    out = subprocess.run(["nm", "-D", "--defined-only", LIB], check=True, stdout=subprocess.PIPE, text=True).stdout
    return {line.split()[-1] for line in out.splitlines() if line.strip()}


def test_shim_header_and_quoted_imports_agree():
    import rust_exp_amd

    rust_exp_amd.lib()           # builds the library if need be
    hs = _read(HS) if os.path.exists(HS) else None
    protos, imports, exports = check_all(_read(HEADER), _read(SHIM), _read(INTEGRATION), hs, _exported())
    assert protos["nb_draw"] == ("void", ["i32", "i32", ("ptr", "mut", "u32")])
    assert protos["nb_num_particles"] == ("i32", [])
    assert imports["nbx_set_particles"][1][2] == ("ptr", "const", "f32") and imports["nbx_get_particles"][1][2] == ("ptr", "mut", "f32")
    assert imports["nbx_create"] == ("i32", [("ptr", "mut", ("ptr", "mut", "opaque:nbx_engine")), "i32"])


MUTATIONS = [
    # (which text, pattern, replacement): each changes ONE argument type, constness, arity or return type
    ("shim", r"fn nbx_stable_orbits\(e: \*mut NbxEngine, n: i32, rmin: f32, rmax: f32\)", "fn nbx_stable_orbits(e: *mut NbxEngine, n: i32, rmin: f64, rmax: f32)"),
    ("shim", r"fn nbx_num_particles\(e: \*const NbxEngine\)", "fn nbx_num_particles(e: *mut NbxEngine)"),
    ("shim", r"fn nbx_draw\(e: \*mut NbxEngine, w: i32, h: i32, fb: \*mut u32\) -> i32;", "fn nbx_draw(e: *mut NbxEngine, w: i32, h: i32, fb: *mut u32) -> i64;"),
    ("shim", r"fn nbx_step_barnes_hut\(e: \*mut NbxEngine, theta: f32, dt: f32, nthreads: i32\)", "fn nbx_step_barnes_hut(e: *mut NbxEngine, theta: f32, dt: f32)"),
    ("shim", r"px: \*const f32, py: \*const f32,\n", "px: *mut f32, py: *const f32,\n"),
    ("shim", r"pub extern fn nb_draw\(w: i32, h: i32, fb: \*mut u32\)", "pub extern fn nb_draw(w: i32, h: i32, fb: *mut u8)"),
    ("shim", r"pub extern fn nb_step_brute_force\(dt: f32\)", "pub extern fn nb_step_brute_force(dt: f64)"),
    ("shim", r"pub extern fn nb_num_particles\(\) -> i32", "pub extern fn nb_num_particles() -> u32"),
    ("shim", r"#\[no_mangle\] pub extern fn nb_random_disk", "pub extern fn nb_random_disk"),
    ("header", r"int32_t nbx_step_barnes_hut\(nbx_engine \*e, float theta, float dt, int32_t nthreads\);", "int32_t nbx_step_barnes_hut(nbx_engine *e, float theta, double dt, int32_t nthreads);"),
    ("header", r"void nb_stable_orbits\(int32_t num_particles, float rmin, float rmax\);", "void nb_stable_orbits(int32_t num_particles, float rmin);"),
    ("header", r"int32_t nb_num_particles\(void\);", "int64_t nb_num_particles(void);"),
    ("header", r"void nb_draw\(int32_t w, int32_t h, uint32_t \*fb\);", "void nb_draw(int32_t w, int32_t h, uint64_t *fb);"),
    ("integration", r"`nb_step_barnes_hut :: CFloat -> CFloat -> CInt -> IO \(\)`", "`nb_step_barnes_hut :: CFloat -> CInt -> CInt -> IO ()`"),
    ("integration", r"`void nb_random_disk\(int32_t n\)`", "`void nb_random_disk(float n)`"),
    ("integration", r"`nb_num_particles :: IO CInt`", "`nb_num_particles :: IO ()`"),
]


@pytest.mark.parametrize("which,pattern,replacement", MUTATIONS)
def test_a_single_changed_type_is_caught(which, pattern, replacement):
    texts = {"header": _read(HEADER), "shim": _read(SHIM), "integration": _read(INTEGRATION)}
    mutated, count = re.subn(pattern, replacement, texts[which], count=1)
    assert count == 1, f"the mutation's pattern no longer matches the {which}: update the test"
    texts[which] = mutated
    with pytest.raises(AssertionError):
        check_all(texts["header"], texts["shim"], texts["integration"], _read(HS) if os.path.exists(HS) else None)
