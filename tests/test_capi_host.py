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
