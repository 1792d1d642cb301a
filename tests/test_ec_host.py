"""CPU unit tests of the kernels' elliptic-curve arithmetic: mpvss_rs_amd/csrc/ec_{field,curves}.h are plain
C++, so the very code the gfx950 kernels run is compiled here with g++ (tests/ec_host_shim.cpp) and compared
with the oracle: field ops, complete group law incl. P+P, P+(-P) and the identity, encodings, scalar mults."""
import ctypes as C
import os
import random
import subprocess

import pytest

import mpvss_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "libec_host.so")


@pytest.fixture(scope="module")
def shim():
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    src = os.path.join(HERE, "ec_host_shim.cpp")
    deps = [src] + [os.path.join(HERE, "..", "mpvss_rs_amd", "csrc", f) for f in ("ec_field.h", "ec_curves.h", "ec_consts.h", "ec_glv.h", "ec_scalar.h")]
    if not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", src, "-o", LIB])
    return C.CDLL(LIB)


def buf(b):
    return (C.c_uint8 * len(b)).from_buffer_copy(b)


@pytest.mark.parametrize("curve,p", [(0, O.SECP_P), (1, O.ED_P)])
def test_field_ops(shim, curve, p):
    rng = random.Random(curve + 1)
    for it in range(200):
        a, b = rng.randrange(p), rng.randrange(p)
        if it < 4:
            a = b = p - 1
        if it == 5:
            a = 0
        if it == 6:
            a = b = (1 << 256) - 1          # non-canonical input is reduced
        for op, want in ((0, a * b % p), (1, a * a % p), (2, (a - b) % p), (4, (a + b) % p)):
            out = (C.c_uint8 * 32)()
            shim.fe_op(curve, op, buf(a.to_bytes(32, "little")), buf(b.to_bytes(32, "little")), out)
            assert int.from_bytes(bytes(out), "little") == want
    a = rng.randrange(1, p)
    out = (C.c_uint8 * 32)()
    shim.fe_op(curve, 3, buf(a.to_bytes(32, "little")), buf(bytes(32)), out)
    want = pow(a, p - 2, p) if curve == 0 else pow(a, (p - 5) // 8, p)
    assert int.from_bytes(bytes(out), "little") == want


@pytest.mark.parametrize("curve,p", [(0, O.SECP_P), (1, O.ED_P)])
def test_field_product_at_the_limb_bounds(shim, curve, p):
    """mul / sqr take lazily added operands: every limb up to 2^30 - 1 (ec_field.h).  Worst-case and random limb
    patterns at that bound: the value is right and every output limb is below 2^27 + 2^12 (reduce_columns)."""
    rng = random.Random(77 + curve)
    top = (1 << 30) - 1
    shim.fe_limb_op.restype = C.c_uint32
    pats = [[top] * 10, [top] * 9 + [0], [0] * 9 + [top], [top if i % 2 else 0 for i in range(10)], [1 << 26] * 10]
    pats += [[rng.randrange(top + 1) for _ in range(10)] for _ in range(300)]
    pats += [[rng.choice((0, top, top - 1, 1 << 29)) for _ in range(10)] for _ in range(200)]
    val = lambda limbs: sum(v << (26 * i) for i, v in enumerate(limbs))
    for it, a in enumerate(pats):
        b = pats[(it * 7 + 3) % len(pats)]
        for op, want in ((0, val(a) * val(b) % p), (1, val(a) * val(a) % p)):
            out = (C.c_uint8 * 32)()
            biggest = shim.fe_limb_op(curve, op, (C.c_uint32 * 10)(*a), (C.c_uint32 * 10)(*b), out)
            assert int.from_bytes(bytes(out), "little") == want
            assert biggest < (1 << 27) + (1 << 12)


@pytest.mark.parametrize("curve,name", [(0, "secp256k1"), (1, "ristretto255")])
def test_group_law_and_encodings(shim, curve, name):
    G = O.GROUPS[name]()
    rng = random.Random(7 + curve)
    order = G.group_order_int()
    out = (C.c_uint8 * G.elem_len)()
    shim.ec_gen(curve, out)
    assert bytes(out) == G.element_to_bytes(G.generator())
    B = G.generator()
    pts = [G.exp(B, rng.randrange(1, order)) for _ in range(6)]
    ident = G.identity()
    e = G.element_to_bytes
    for a, b in [(pts[0], pts[1]), (pts[0], pts[0]), (pts[0], G.element_inverse(pts[0])), (ident, pts[2]),
                 (pts[2], ident), (ident, ident)]:
        assert shim.ec_add(curve, buf(e(a)), buf(e(b)), out, 0) == 0
        assert bytes(out) == e(G.mul(a, b))
    for a in pts + [ident]:
        shim.ec_add(curve, buf(e(a)), buf(e(a)), out, 1)
        assert bytes(out) == e(G.mul(a, a))
    s = G.scalar_to_bytes
    for k in [0, 1, 2, 3, order - 1, rng.randrange(order)]:
        assert shim.ec_dual(curve, buf(e(pts[3])), buf(s(k)), None, None, out) == 0
        assert bytes(out) == e(G.exp(pts[3], k))
    k1, k2 = rng.randrange(order), rng.randrange(order)
    shim.ec_dual(curve, buf(e(pts[4])), buf(s(k1)), buf(e(pts[5])), buf(s(k2)), out)
    assert bytes(out) == e(G.mul(G.exp(pts[4], k1), G.exp(pts[5], k2)))
    for k in [0, 1, 5, 65536, 65535, (1 << 40) + 77]:
        shim.ec_small(curve, buf(e(pts[0])), C.c_uint64(k), out)
        assert bytes(out) == e(G.exp(pts[0], k))
    # the same through the non-adjacent form (the seed kernel of a lone box): runs of ones, alternating bits, the largest position
    for k in [0, 1, 2, 3, 5, 7, 0x5555, 0xAAAA, 65535, 65536, 65537, (1 << 40) + 77, (1 << 61) - 1, (1 << 61) - 2] + \
            [rng.getrandbits(rng.randrange(1, 62)) for _ in range(12)]:
        shim.ec_small_naf(curve, buf(e(pts[0])), C.c_uint64(k), out)
        assert bytes(out) == e(G.exp(pts[0], k)), k
    shim.ec_small_naf(curve, buf(e(ident)), C.c_uint64(12345), out)
    assert bytes(out) == e(ident)
    # invalid encodings are rejected
    bad = [b"\x05" + bytes(32), b"\x02" + (O.SECP_P).to_bytes(32, "big")] if curve == 0 else \
          [bytes.fromhex("01" + "00" * 31), bytes.fromhex("ed" + "ff" * 30 + "7f")]
    for x in bad:
        assert shim.ec_decode_ok(curve, buf(x)) == 0


@pytest.mark.parametrize("curve,name", [(0, "secp256k1"), (1, "ristretto255")])
def test_windowed_paths(shim, curve, name):
    """The signed-4-bit Straus routine over cached tables, the fixed-base comb of the generator and the batched SEC1
    encoding (the routines the windowed kernels run) against the oracle, incl. scalars 0, 1, order - 1, digits that
    hit -8 / +8 and the identity as a base."""
    G = O.GROUPS[name]()
    rng = random.Random(17 + curve)
    order = G.group_order_int()
    e, s = G.element_to_bytes, G.scalar_to_bytes
    B = G.generator()
    P, Q = G.exp(B, rng.randrange(1, order)), G.exp(B, rng.randrange(1, order))
    out = (C.c_uint8 * G.elem_len)()
    special = [0, 1, 2, 7, 8, 9, 15, 16, 0x88888888, 0x77777777, 0x8 << 252, order - 1, order - 2,
               int("8" * 62, 16), int("7" * 63, 16) % order, int("f" * 63, 16) % order]
    ks = special + [rng.randrange(order) for _ in range(6)]
    for k in ks:
        assert shim.ec_comb(curve, buf(s(k)), out) == 0
        assert bytes(out) == e(G.exp(B, k)), hex(k)
        assert shim.ec_dual_win(curve, buf(e(P)), buf(s(k)), None, None, out) == 0
        assert bytes(out) == e(G.exp(P, k)), hex(k)
    for k1, k2 in zip(ks, reversed(ks)):
        assert shim.ec_dual_win(curve, buf(e(P)), buf(s(k1)), buf(e(Q)), buf(s(k2)), out) == 0
        assert bytes(out) == e(G.mul(G.exp(P, k1), G.exp(Q, k2)))
    ident = G.identity()
    k1, k2 = rng.randrange(order), rng.randrange(order)
    assert shim.ec_dual_win(curve, buf(e(ident)), buf(s(k1)), buf(e(Q)), buf(s(k2)), out) == 0
    assert bytes(out) == e(G.exp(Q, k2))
    assert shim.ec_dual_win(curve, buf(e(P)), buf(s(k1)), buf(e(G.element_inverse(P))), buf(s(k1)), out) == 0
    assert bytes(out) == e(ident)
    if curve == 0:
        pts = [P, ident, Q, G.exp(B, 5)]
        out4 = (C.c_uint8 * (4 * 33))()
        assert shim.secp_encode_batch4(buf(b"".join(e(x) for x in pts)), out4) == 0
        assert bytes(out4) == b"".join(e(G.exp(x, 3)) for x in pts)


def test_secp256k1_glv_split_and_windowed_multiplication(shim):
    """ec_glv.h: k = k1 + k2 lambda (mod n) with |k1|, |k2| < 2^128 for every k -- 0, 1, n - 1, lambda, n - lambda, values around
    n / 2 and 2^128, unreduced inputs (n, 2^256 - 1), 3000 random ones -- and the 33-window double multiplication built on it
    (phi(P) = (beta x, y) made from P's table) equals the oracle's k1 P + k2 Q, incl. the identity as a base and P, -P."""
    G = O.GROUPS["secp256k1"]()
    n = G.group_order_int()
    lam = 0x5363ad4cc05c30e0a5261c028812645a122e22ea20816678df02967c1b23bd72
    assert pow(lam, 3, n) == 1
    rng = random.Random(0x61F)
    ks = [0, 1, 2, n - 1, n - 2, lam, n - lam, lam + 1, (n - 1) // 2, (n + 1) // 2, 1 << 128, (1 << 128) - 1, (1 << 128) + 1, 1 << 255,
          n - (1 << 128), n, (1 << 256) - 1, int("8" * 64, 16), int("7" * 64, 16)] + [rng.randrange(n) for _ in range(3000)]
    out = (C.c_uint8 * 36)()
    for k in ks:
        assert shim.secp_glv_split_bytes(buf(k.to_bytes(32, "big")), out) == 0
        raw = bytes(out)
        m1, s1, m2, s2 = int.from_bytes(raw[0:17], "little"), raw[17], int.from_bytes(raw[18:35], "little"), raw[35]
        assert m1 < (1 << 128) and m2 < (1 << 128), hex(k)
        assert ((-m1 if s1 else m1) + (-m2 if s2 else m2) * lam - k) % n == 0, hex(k)
    e, s = G.element_to_bytes, G.scalar_to_bytes
    B = G.generator()
    P, Q = G.exp(B, rng.randrange(1, n)), G.exp(B, rng.randrange(1, n))
    outp = (C.c_uint8 * 33)()
    sample = ks[:17] + ks[19:40]                       # (canonical scalars: the group API rejects the others before any kernel runs)
    for k1, k2 in zip(sample, reversed(sample)):
        assert shim.ec_dual_win_glv(buf(e(P)), buf(s(k1)), buf(e(Q)), buf(s(k2)), outp) == 0
        assert bytes(outp) == e(G.mul(G.exp(P, k1), G.exp(Q, k2))), (hex(k1), hex(k2))
        assert shim.ec_dual_win_glv(buf(e(P)), buf(s(k1)), None, None, outp) == 0
        assert bytes(outp) == e(G.exp(P, k1)), hex(k1)
    ident = G.identity()
    k1, k2 = rng.randrange(n), rng.randrange(n)
    assert shim.ec_dual_win_glv(buf(e(ident)), buf(s(k1)), buf(e(Q)), buf(s(k2)), outp) == 0
    assert bytes(outp) == e(G.exp(Q, k2))
    assert shim.ec_dual_win_glv(buf(e(P)), buf(s(k1)), buf(e(G.element_inverse(P))), buf(s(k1)), outp) == 0
    assert bytes(outp) == e(ident)
