"""The flat wire format of a box ("MPVSSBX1", include/mpvss_hip.h): serialisation round trip on the
golden fixtures of the three groups, section alignment, and rejection of everything that is not exactly one box."""
import glob
import json
import os

import pytest

from mpvss_rs_amd import capi

HERE = os.path.dirname(os.path.abspath(__file__))
GID = {"modp2048": 0, "secp256k1": capi.GROUP_SECP256K1, "ristretto255": capi.GROUP_RISTRETTO255}


def fixture_box(path):
    fx = json.load(open(path))
    b = fx["box"]
    cat = lambda hs: bytes.fromhex("".join(hs))
    u = int(b["U"], 16) if isinstance(b.get("U"), str) else int(b.get("U", 0))
    ub = u.to_bytes(max(1, (u.bit_length() + 7) // 8), "big")
    return fx, GID[fx["group"]], dict(commitments=cat(b["commitments"]), positions=b["positions"], pubkeys=cat(b["publickeys"]),
                                     shares=cat(b["shares"]), responses=cat(b["responses"]),
                                     challenge=bytes.fromhex(b["challenge"]), u_be=ub)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(HERE, "golden", "*.json"))), ids=os.path.basename)
def test_round_trip_and_layout(path):
    fx, gid, box = fixture_box(path)
    wire = capi.box_serialize(gid, **box)
    e = 256 if gid == 0 else capi.EC_ENC[gid]
    s = 256 if gid == 0 else 32
    n, t = fx["n"], fx["t"]
    assert wire[:8] == b"MPVSSBX1"
    assert int.from_bytes(wire[8:12], "little") == gid and int.from_bytes(wire[12:16], "little") == e
    assert (int.from_bytes(wire[16:24], "little"), int.from_bytes(wire[24:32], "little")) == (n, t)
    pad8 = lambda x: (x + 7) & ~7
    off = 40
    assert wire[off:off + t * e] == box["commitments"]
    off = pad8(off + t * e)
    assert [int.from_bytes(wire[off + 8 * i:off + 8 * i + 8], "little", signed=True) for i in range(n)] == box["positions"]
    off = pad8(off + 8 * n)
    assert wire[off:off + n * e] == box["pubkeys"]
    assert len(wire) == pad8(pad8(pad8(pad8(off + n * e) + n * e) + n * s) + s) + len(box["u_be"])
    back = capi.box_parse(wire)
    assert back["group"] == gid and back["n"] == n and back["t"] == t
    for k in ("commitments", "positions", "pubkeys", "shares", "responses", "challenge"):
        assert back[k] == box[k], k
    assert back["U"] == box["u_be"]
    # canonical: serialising the parsed box gives the same bytes
    assert capi.box_serialize(gid, back["commitments"], back["positions"], back["pubkeys"], back["shares"], back["responses"],
                              back["challenge"], back["U"]) == wire
    for bad in (wire[:-1], wire + b"\0", b"MPVSSBX2" + wire[8:], wire[:8] + (9).to_bytes(4, "little") + wire[12:],
                wire[:12] + (e + 1).to_bytes(4, "little") + wire[16:], wire[:16] + (n + 1).to_bytes(8, "little") + wire[24:],
                wire[:39], b""):
        with pytest.raises(capi.EngineError):
            capi.box_parse(bad)


def test_empty_box_and_size_limits():
    wire = capi.box_serialize(0, b"", [], b"", b"", b"", bytes(256))
    back = capi.box_parse(wire)
    assert back["n"] == 0 and back["t"] == 0 and back["challenge"] == bytes(256) and back["U"] == b""
    lib = capi.load_library()
    assert lib.mpvss_box_wire_size(7, 1, 1, 0) == 0
    assert lib.mpvss_box_wire_size(0, 1 << 41, 1, 0) == 0


def test_parser_never_trusts_the_header():
    """The parser reads untrusted bytes: every accepted blob must be exactly one canonical box (sections inside the
    buffer, re-serialising gives the same bytes), everything else must be refused -- no crash, no out-of-bounds view.
    Random header mutations, truncations and extensions of valid blobs of the three groups, and random noise."""
    import random
    rng = random.Random(0xB0C5)
    blobs = []
    for path in sorted(glob.glob(os.path.join(HERE, "golden", "*.json"))):
        fx, gid, box = fixture_box(path)
        blobs.append(capi.box_serialize(gid, **box))
    accepted = 0
    for it in range(4000):
        base = bytearray(rng.choice(blobs))
        kind = it % 5
        if kind == 0:                                   # a header field replaced by an arbitrary value
            off = rng.choice((8, 12, 16, 24, 32))
            width = 4 if off in (8, 12) else 8
            val = rng.choice((0, 1, rng.randrange(1 << 16), rng.randrange(1 << (8 * width)), (1 << (8 * width)) - 1))
            base[off:off + width] = val.to_bytes(width, "little")
        elif kind == 1:                                 # truncated or extended
            cut = rng.randrange(len(base) + 1)
            base = base[:cut] + bytearray(rng.randbytes(rng.choice((0, 0, 1, 7, 64))))
        elif kind == 2:                                 # a flipped bit anywhere
            i = rng.randrange(len(base))
            base[i] ^= 1 << rng.randrange(8)
        elif kind == 3:                                 # noise behind a plausible magic
            base = bytearray(b"MPVSSBX1" + rng.randbytes(rng.randrange(0, 200)))
        else:                                           # padding bytes must be zero: spoil one byte of a header-implied gap
            i = rng.randrange(40, len(base))
            base[i] = (base[i] + rng.randrange(1, 256)) & 0xFF
        wire = bytes(base)
        try:
            back = capi.box_parse(wire)
        except capi.EngineError:
            continue
        accepted += 1
        again = capi.box_serialize(back["group"], back["commitments"], back["positions"], back["pubkeys"], back["shares"],
                                   back["responses"], back["challenge"], back["U"])
        assert again == wire, (it, kind)                # canonical: an accepted blob is its own serialisation
    assert accepted > 100                               # payload-only changes are still boxes
