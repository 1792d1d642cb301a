"""CPU-only checks of the product library: it loads, exports every symbol include/mpvss_hip.h declares,
its host-only helpers are correct, and it refuses to compute without a GPU (no silent fallback)."""
import hashlib
import os
import re

import pytest

from mpvss_rs_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = capi.load_library()
    header = open(os.path.join(ROOT, "include", "mpvss_hip.h")).read()
    declared = set(re.findall(r"\b(mpvss_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(capi.EXPORTED_SYMBOLS)
    for sym in declared:
        assert hasattr(lib, sym), sym


def test_host_sha256_matches_hashlib():
    for n in (0, 1, 55, 56, 63, 64, 65, 119, 120, 1000, 70001):
        d = bytes((i * 7 + n) & 0xFF for i in range(n))
        assert capi.sha256(d) == hashlib.sha256(d).digest()


def test_transcript_framing_minimal_length():
    """dleq.rs:58-61 + modp.rs:150-152: u64-BE length then minimal-length bytes; zero -> one 0x00 byte."""
    elems = [0, 1, 255, 256, 1 << 36, (1 << 2040) - 1, 1 << 2040, (1 << 2048) - 1]
    raw = b"".join(e.to_bytes(256, "big") for e in elems)
    st = capi.transcript_absorb(capi.transcript_init(), raw)
    verdict, digest = capi.transcript_verdict(st, bytes(256))
    h = hashlib.sha256()
    for e in elems:
        b = e.to_bytes(max(1, (e.bit_length() + 7) // 8), "big")
        h.update(len(b).to_bytes(8, "big") + b)
    assert digest == h.digest() and verdict is False
    # verdict compares int(SHA256(digest)) with the challenge (modp.rs:142-148, dleq.rs:119-126)
    c = hashlib.sha256(digest).digest().rjust(256, b"\0")
    assert capi.transcript_verdict(st, c)[0] is True
    # absorbing in two pieces == absorbing at once (state is resumable across ranks)
    st2 = capi.transcript_absorb(capi.transcript_absorb(capi.transcript_init(), raw[:768]), raw[768:])
    assert capi.transcript_verdict(st2, c) == (True, digest)


def test_no_cpu_fallback():
    lib = capi.load_library()
    if lib.mpvss_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(capi.EngineError):
        capi.Engine(0)


def test_ec_hash_to_scalar_host():
    import mpvss_oracle as O
    for name, gid in (("secp256k1", capi.GROUP_SECP256K1), ("ristretto255", capi.GROUP_RISTRETTO255)):
        G = O.GROUPS[name]()
        for data in (b"", b"test data", bytes(range(32)), b"\xff" * 100):
            assert capi.ec_hash_to_scalar(gid, data) == G.scalar_to_bytes(G.hash_to_scalar(data))


def test_rust_ffi_declares_the_exported_symbols():
    """rust/src/ffi.rs (never compiled here: no Rust toolchain) must declare exactly the functions include/mpvss_hip.h
    exports, so that the shipped binding cannot drift from the library."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ffi = open(os.path.join(root, "rust", "src", "ffi.rs")).read()
    declared = set(re.findall(r"pub fn (mpvss_\w+)\(", ffi))
    from mpvss_rs_amd import EXPORTED_SYMBOLS
    assert declared == set(EXPORTED_SYMBOLS), (sorted(declared - set(EXPORTED_SYMBOLS)), sorted(set(EXPORTED_SYMBOLS) - declared))


def _c_decls(header_text):
    """include/mpvss_hip.h -> ({function: (return type, [(parameter name, C type)])}, {struct: [(field, C type)]})"""
    import re
    text = re.sub(r"/\*.*?\*/", " ", header_text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    text = re.sub(r"^\s*#.*$", " ", text, flags=re.M)
    structs = {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*\w+\s*;", text, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            # "const uint8_t *a, *b" / "size_t a, b" / "double kernel_ms[4]"
            base = re.match(r"((?:const\s+)?(?:unsigned\s+long\s+long|\w+))\s*(.*)", decl)
            for item in base.group(2).split(","):
                item = item.strip()
                stars = item.count("*")
                name = item.replace("*", "").strip()
                arr = re.match(r"(\w+)\[(\d+)\]", name)
                if arr:
                    fields.append((arr.group(1), f"{base.group(1)}[{arr.group(2)}]"))
                else:
                    fields.append((name, base.group(1) + "*" * stars))
        structs[m.group(1)] = fields
    text = re.sub(r"typedef\s+struct\s+\w+\s*\{.*?\}\s*\w+\s*;", " ", text, flags=re.S)
    funcs = {}
    for m in re.finditer(r"([\w\s\*]+?)\b(mpvss_\w+)\s*\(([^()]*)\)\s*;", text):
        ret = " ".join(m.group(1).replace("extern", "").split())
        params = []
        body = " ".join(m.group(3).split())
        if body and body != "void":
            for prm in body.split(","):
                prm = prm.strip()
                arr = re.match(r"(.*?)(\w+)\s*\[\d*\]$", prm)          # uint8_t out32[32] decays to a pointer
                if arr:
                    params.append((arr.group(2), " ".join(arr.group(1).split()) + "*"))
                    continue
                pm = re.match(r"(.*?)(\w+)$", prm)
                ty = "".join(pm.group(1).split())
                if not ty.endswith("*const*"):
                    ty = ty.replace("const", "const ")
                params.append((pm.group(2), ty.replace("unsignedlonglong", "unsigned long long")))
        funcs[m.group(2)] = (ret.replace(" *", "*"), params)
    return funcs, structs


_C2RUST = {"int": "c_int", "double": "c_double", "size_t": "usize", "unsigned long long": "c_ulonglong", "uint8_t": "u8",
           "int64_t": "i64", "char": "c_char", "void": "c_void", "uint16_t": "u16", "uint32_t": "u32"}


def _rust_type(ctype):
    """the Rust FFI spelling of a C type of the header: `const uint8_t*` -> `*const u8`, `mpvss_ctx**` -> `*mut *mut mpvss_ctx`"""
    import re
    ctype = ctype.strip()
    if ctype.replace(" ", "").endswith("*const*"):            # T* const*: a read-only array of pointers
        return "*const " + _rust_type(ctype.replace(" ", "")[:-len("const*")])
    arr = re.match(r"(.*)\[(\d+)\]$", ctype)
    if arr:
        return f"[{_rust_type(arr.group(1))}; {arr.group(2)}]"
    const = ctype.startswith("const ")
    core = ctype[6:] if const else ctype
    stars = core.count("*")
    base = core.replace("*", "").strip()
    rust = _C2RUST.get(base, base)
    for k in range(stars):
        rust = ("*const " if (const and k == 0) else "*mut ") + rust
    return rust


def _rust_decls(ffi_text):
    import re
    text = re.sub(r"//[^\n]*", " ", ffi_text)
    funcs = {}
    for m in re.finditer(r"pub fn (mpvss_\w+)\s*\(([^()]*)\)\s*(?:->\s*([^;]+?))?\s*;", text):
        params = []
        for prm in " ".join(m.group(2).split()).split(","):
            prm = prm.strip()
            if prm:
                name, ty = prm.split(":", 1)
                params.append((name.strip(), " ".join(ty.split())))
        funcs[m.group(1)] = ((m.group(3) or "").strip(), params)
    structs = {}
    for m in re.finditer(r"pub struct (\w+)\s*\{(.*?)\}", text, flags=re.S):
        fields = []
        for f in m.group(2).split(","):
            f = " ".join(f.split())
            if f:
                name, ty = f.split(":", 1)
                fields.append((name.strip().removeprefix("pub ").strip() if hasattr(str, "removeprefix") else name.strip()[4:].strip(), ty.strip()))
        structs[m.group(1)] = fields
    return funcs, structs


def test_rust_ffi_signatures_match_the_header():
    """Names alone do not protect a binding that is never compiled here: an argument added to one side only is undefined
    behaviour at the first call.  Every prototype of include/mpvss_hip.h and every `pub fn` of rust/src/ffi.rs is parsed
    into (name, return type, [(parameter name, type)]) with the C types mapped to their Rust FFI spelling, and the `#[repr(C)]`
    structs field by field; the two sides must agree exactly (src/group.rs:24-124 is what the crate implements on top)."""
    cf, cs = _c_decls(open(os.path.join(ROOT, "include", "mpvss_hip.h")).read())
    rf, rs = _rust_decls(open(os.path.join(ROOT, "rust", "src", "ffi.rs")).read())
    from mpvss_rs_amd import EXPORTED_SYMBOLS
    assert set(cf) == set(EXPORTED_SYMBOLS) == set(rf), (sorted(set(cf) ^ set(EXPORTED_SYMBOLS)), sorted(set(rf) ^ set(cf)))
    for name, (ret, params) in sorted(cf.items()):
        want_ret = "" if ret == "void" else _rust_type(ret)
        want = [(pn, _rust_type(pt)) for pn, pt in params]
        assert rf[name][0] == want_ret, (name, rf[name][0], want_ret)
        assert rf[name][1] == want, (name, [a for a, b in zip(rf[name][1], want) if a != b] or (len(rf[name][1]), len(want)))
    for sname in ("mpvss_modp_box", "mpvss_ec_box", "mpvss_box_view", "mpvss_pipeline_stats"):
        want = [(fn, _rust_type(ft)) for fn, ft in cs[sname]]
        assert rs[sname] == want, (sname, rs[sname], want)


def test_signature_parsers_notice_a_drifted_argument():
    """the check above must fail for the drift it exists for: an extra argument, a reordered pair, a changed pointer kind"""
    cf, _ = _c_decls("int mpvss_x(mpvss_ctx* ctx, const uint8_t* a, size_t n, uint8_t out32[32]);")
    assert cf["mpvss_x"] == ("int", [("ctx", "mpvss_ctx*"), ("a", "const uint8_t*"), ("n", "size_t"), ("out32", "uint8_t*")])
    want = [(pn, _rust_type(pt)) for pn, pt in cf["mpvss_x"][1]]
    good, _ = _rust_decls("pub fn mpvss_x(ctx: *mut mpvss_ctx, a: *const u8, n: usize, out32: *mut u8) -> c_int;")
    assert good["mpvss_x"] == ("c_int", want)
    for bad in ("pub fn mpvss_x(ctx: *mut mpvss_ctx, a: *const u8, n: usize) -> c_int;",
                "pub fn mpvss_x(ctx: *mut mpvss_ctx, n: usize, a: *const u8, out32: *mut u8) -> c_int;",
                "pub fn mpvss_x(ctx: *mut mpvss_ctx, a: *mut u8, n: usize, out32: *mut u8) -> c_int;",
                "pub fn mpvss_x(ctx: *mut mpvss_ctx, a: *const u8, n: usize, out32: *mut u8);"):
        got, _ = _rust_decls(bad)
        assert got["mpvss_x"] != ("c_int", want), bad


def _rust_sources():
    import glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "rust", "src", "**", "*.rs"), recursive=True) + glob.glob(os.path.join(root, "rust", "examples", "*.rs")))
    return {os.path.relpath(f, root): open(f).read() for f in files}


def _reference_api():
    import json
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_api", "reference_api.json")) as fh:
        return json.load(fh)


def _strip_rust_comments(text):
    import re
    return re.sub(r"//[^\n]*", "", text)


def test_rust_names_only_public_paths_of_the_reference():
    """Every `mpvss_rs::...` path rust/ uses (code, not comments) must be PUBLIC in the reference (tests/reference_api/reference_api.json,
    made by tools/gen_reference_api.py from src/lib.rs and the files of its `pub mod`s): round 5 called `mpvss_rs::util::Util`, a
    private module (src/lib.rs:27) -- E0603, the crate would not have built."""
    import re
    api = _reference_api()
    public = set(api["public_paths"])
    used = {}
    for rel, text in _rust_sources().items():
        code = _strip_rust_comments(text)
        for m in re.finditer(r"mpvss_rs::((?:[a-z_0-9]+::)*)(\{[^}]*\}|[A-Za-z_0-9]+)", code):
            prefix, last = m.group(1), m.group(2)
            names = [x.strip() for x in last.strip("{}").split(",")] if last.startswith("{") else [last]
            for nm in names:
                if nm:
                    used.setdefault("mpvss_rs::" + prefix + nm, set()).add(rel)
    assert used, "rust/ no longer names the reference crate at all?"
    not_public = {p: sorted(f) for p, f in used.items() if p not in public}
    assert not not_public, f"rust/ names paths that are not public in the reference: {not_public}"
    for mod in api["private_modules"]:
        for rel, text in _rust_sources().items():
            assert f"mpvss_rs::{mod}" not in _strip_rust_comments(text), (rel, mod)
    # the manifest's dependency is the crate the listing was made from
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cargo = open(os.path.join(root, "rust", "Cargo.toml")).read()
    assert re.search(r'^mpvss-rs = "%s' % api["version"].split(".")[0], cargo, re.M), "rust/Cargo.toml depends on another major version"


def test_rust_participant_has_the_reference_method_surface():
    """rust/src/participant.rs: a `Participant<G>` with the reference's fields (privatekey, publickey public; group private), the
    generic constructors, and for each of the three groups an impl block with the reference's methods -- same names, same receiver,
    same parameter names in the same order (participant.rs:158, 1085, 1564)."""
    import re
    api = _reference_api()
    text = _rust_sources()[os.path.join("rust", "src", "participant.rs")]
    code = _strip_rust_comments(text)
    fields = re.search(r"pub struct Participant<G: Group> \{(.*?)\}", code, re.S).group(1)
    assert re.findall(r"pub ([a-z_]+):", fields) == api["participant_fields"]["public"]
    assert [f for f in re.findall(r"^\s+([a-z_]+):", fields, re.M)] == api["participant_fields"]["private"]
    starts = [(m.start(), m.group(1)) for m in re.finditer(r"^impl(?:<[^>]*>)? Participant<([A-Za-z0-9_]+)>", code, re.M)]
    ours = {}
    for k, (pos, who) in enumerate(starts):
        end = starts[k + 1][0] if k + 1 < len(starts) else len(code)
        for m in re.finditer(r"^    pub fn ([a-z_0-9]+)\s*\(([^)]*)\)", code[pos:end], re.M | re.S):
            params = [p.strip().split(":")[0].strip() for p in m.group(2).split(",") if p.strip()]
            ours.setdefault(who, {})[m.group(1)] = {"receiver": params[0] if params and "self" in params[0] else None,
                                                    "params": [p for p in params if "self" not in p]}
    pairs = {"G": "G", "ModpGroup": "HipModpGroup", "Secp256k1Group": "HipSecp256k1Group", "Ristretto255Group": "HipRistretto255Group"}
    for ref_group, our_group in pairs.items():
        want, got = api["participant_impls"][ref_group], ours.get(our_group, {})
        for name, sig in want.items():
            assert name in got, f"Participant<{our_group}> lacks {name}"
            assert got[name] == sig, (our_group, name, got[name], sig)
    # nothing but an accessor beyond the reference's surface in the generic block
    assert set(ours["G"]) - set(api["participant_impls"]["G"]) <= {"group"}
    # the crate root offers what a program written against the reference imports from it
    lib = _strip_rust_comments(_rust_sources()[os.path.join("rust", "src", "lib.rs")])
    for name in ("Participant", "ModpParticipant", "Secp256k1Participant", "Ristretto255Participant", "string_to_secret", "string_from_secret"):
        assert re.search(r"pub use [^;]*\b%s\b" % name, lib), name


def test_header_is_plain_c_and_cxx():
    """The boundary is a C ABI: include/mpvss_hip.h must compile on its own as C99 and as C++11 (what cgo / bindgen /
    a C++ host would feed it to), without warnings."""
    import shutil
    import subprocess
    header = os.path.join(ROOT, "include", "mpvss_hip.h")
    for cc, args in (("gcc", ["-std=c99", "-x", "c", "-pedantic"]), ("g++", ["-std=c++11", "-x", "c++"])):
        if shutil.which(cc) is None:
            pytest.skip(f"{cc} not available")
        out = subprocess.run([cc, "-fsyntax-only", "-Wall", "-Wextra", "-Werror", *args, header], capture_output=True, text=True)
        assert out.returncode == 0, out.stderr
