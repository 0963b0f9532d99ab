"""rust-shim/src/lib.rs cannot be compiled in this image (no cargo): this test is the compiler's stand-in for the part that can be checked
without one -- every `extern "C"` declaration of the shim against the prototype of the same name in include/sumcheck_hip.h (arity, every
parameter's type, the return type), `sc_poly_desc` field by field, and the constants.  A header change that the shim does not follow, or a
typo in the shim, fails here instead of at a maintainer's first `cargo build`."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = open(os.path.join(ROOT, "include", "sumcheck_hip.h")).read()
RS = open(os.path.join(ROOT, "rust-shim", "src", "lib.rs")).read()

C_BASE = {"int": "c_int", "uint32_t": "u32", "uint64_t": "u64", "int64_t": "i64", "size_t": "usize", "double": "f64", "float": "f32", "uint8_t": "u8",
          "char": "c_char", "void": "c_void", "sc_poly_desc": "sc_poly_desc", "sc_prover": "sc_prover", "sc_rng": "sc_rng", "sc_comm": "sc_comm",
          "sc_allreduce_u64_fn": "sc_allreduce_u64_fn", "sc_allgather_fn": "sc_allgather_fn"}


def c_type_to_rust(t):
    """'const uint64_t *const *' -> '*const *const u64'; 'sc_prover **' -> '*mut *mut sc_prover'; 'uint32_t' -> 'u32'"""
    t = t.strip()
    toks = re.findall(r"[A-Za-z_][A-Za-z_0-9]*|\*", t)
    base = [x for x in toks if x not in ("const", "*", "struct", "unsigned")]
    assert len(base) == 1, t
    out = C_BASE[base[0]]
    # walk the declarator left to right: a '*' makes a pointer to what is on its left; it is *const if 'const' qualifies the pointee
    pointee_const = toks[0] == "const" or (len(toks) > 1 and toks[1] == "const" and toks[0] == base[0])
    i = toks.index(base[0]) + 1
    if i < len(toks) and toks[i] == "const":
        pointee_const = True
        i += 1
    while i < len(toks):
        if toks[i] == "*":
            out = ("*const " if pointee_const else "*mut ") + out
            pointee_const = False
        elif toks[i] == "const":
            pointee_const = True
        i += 1
    return out


def header_prototypes():
    protos = {}
    for m in re.finditer(r"SC_API\s+([^;(]+?)\b(sc_[a-z0-9_]+)\s*\(([^;]*?)\)\s*;", HDR, re.S):
        ret, name, params = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        plist = []
        if params and params != "void":
            for prm in params.split(","):
                prm = prm.strip()
                mm = re.match(r"(.*?)([A-Za-z_][A-Za-z_0-9]*)$", prm)  # the last identifier is the parameter's name
                plist.append(c_type_to_rust(mm.group(1)))
        protos[name] = (plist, None if ret == "void" else c_type_to_rust(ret))
    return protos


def rust_externs():
    blocks = re.findall(r'extern "C" \{(.*?)\n\}', RS, re.S)
    out = {}
    for b in blocks:
        for m in re.finditer(r"pub fn (sc_[a-z0-9_]+)\s*\((.*?)\)\s*(?:->\s*([^;]+?))?\s*;", b, re.S):
            name, params, ret = m.group(1), " ".join(m.group(2).split()), m.group(3)
            plist = [p.split(":", 1)[1].strip() for p in params.split(",") if p.strip()]
            out[name] = (plist, ret.strip() if ret else None)
    return out


def test_c_type_mapping_examples():
    assert c_type_to_rust("const uint64_t *const *") == "*const *const u64"
    assert c_type_to_rust("sc_prover **") == "*mut *mut sc_prover"
    assert c_type_to_rust("const char *") == "*const c_char" and c_type_to_rust("uint32_t") == "u32" and c_type_to_rust("const void *") == "*const c_void"


def test_every_extern_of_the_shim_matches_the_header():
    protos, ext = header_prototypes(), rust_externs()
    assert len(protos) >= 60 and len(ext) >= 35, (len(protos), len(ext))
    for name, (rp, rr) in ext.items():
        assert name in protos, f"the shim declares {name}, the header does not"
        hp, hr = protos[name]
        assert len(rp) == len(hp), f"{name}: {len(rp)} parameters in the shim, {len(hp)} in the header"
        for i, (a, b) in enumerate(zip(rp, hp)):
            assert a == b, f"{name}, parameter {i}: shim `{a}`, header `{b}`"
        assert rr == hr, f"{name}: returns `{rr}` in the shim, `{hr}` in the header"


def test_poly_desc_and_constants_match_the_header():
    hs = re.search(r"typedef struct sc_poly_desc \{(.*?)\} sc_poly_desc;", HDR, re.S) or re.search(r"struct sc_poly_desc \{(.*?)\};", HDR, re.S)
    assert hs, "sc_poly_desc not found in the header"
    body = re.sub(r"/\*.*?\*/", "", hs.group(1), flags=re.S)
    hfields = []
    for decl in body.split(";"):
        decl = " ".join(decl.split())
        if not decl:
            continue
        mm = re.match(r"(.*?)([A-Za-z_][A-Za-z_0-9]*)$", decl)
        hfields.append((mm.group(2), c_type_to_rust(mm.group(1))))
    rsb = re.search(r"pub struct sc_poly_desc \{(.*?)\}", RS, re.S).group(1)
    rfields = [(m.group(1), m.group(2).strip()) for m in re.finditer(r"pub ([a-z_]+):\s*([^,]+),", rsb)]
    assert rfields == hfields, (rfields, hfields)
    def c_value(expr):  # "5", "0x100", "1u << 3"
        expr = expr.strip()
        m = re.match(r"^(\d+)u?\s*<<\s*(\d+)$", expr)
        return (int(m.group(1)) << int(m.group(2))) if m else int(expr.rstrip("u"), 0)

    consts = re.findall(r"pub const (SC_[A-Z_0-9]+): (?:c_int|u32) = (\d+);", RS)
    assert len(consts) >= 10
    for name, val in consts:
        m = re.search(r"#define\s+%s\s+(0x[0-9a-fA-F]+|\d+)" % name, HDR) or re.search(r"\b%s\s*=\s*([0-9a-fx]+u?(?:\s*<<\s*\d+)?)" % name, HDR)
        assert m, f"{name} is not in the header"
        assert c_value(m.group(1)) == int(val), f"{name}: {val} in the shim, {m.group(1)} in the header"
