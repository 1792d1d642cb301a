"""The scalar-field entry points of the C ABI (host only: scalar_mul / scalar_sub, batched DLEQ responses, polynomial
values) against Python integers following the reference's definitions:
  src/groups/modp.rs:180-192, secp256k1.rs:173-181, ristretto255.rs:244-252, src/dleq.rs:42-50, src/polynomial.rs:50-58."""
import random

import pytest

import mpvss_oracle as O
from mpvss_rs_amd import capi

GROUPS = [("modp2048", 0), ("secp256k1", capi.GROUP_SECP256K1), ("ristretto255", capi.GROUP_RISTRETTO255)]


def codec(G, gid):
    if gid == 0:
        return (lambda v: v.to_bytes(256, "big")), (lambda b: int.from_bytes(b, "big")), 256
    return G.scalar_to_fixed, G.scalar_from_fixed, 32


@pytest.mark.parametrize("name,gid", GROUPS)
def test_scalar_mul_and_sub(name, gid):
    G = O.GROUPS[name]()
    enc, dec, w = codec(G, gid)
    order = G.group_order_int()
    rng = random.Random(5 + gid)
    cases = [(0, 0), (1, order - 1), (order - 1, order - 1), (order - 1, 1), (5, 7), (7, 5), (0, 1)]
    cases += [(rng.randrange(order), rng.randrange(order)) for _ in range(40)]
    for a, b in cases:
        assert dec(capi.scalar_mul(gid, enc(a), enc(b))) == G.scalar_mul(a, b) % order
        assert dec(capi.scalar_sub(gid, enc(a), enc(b))) == G.scalar_sub(a, b) % order
    if gid == 0:      # unreduced MODP operands follow the reference's formulas as well (modp.rs:180-192)
        a, b = (1 << 2048) - 1, (1 << 2047) + 12345
        assert dec(capi.scalar_mul(0, enc(a), enc(b))) == (a * b) % order
        assert dec(capi.scalar_sub(0, enc(a), enc(b))) == (a - b) % order
        assert dec(capi.scalar_sub(0, enc(b), enc(a))) == b - a + order


@pytest.mark.parametrize("name,gid", GROUPS)
def test_batched_responses_and_polynomial_values(name, gid):
    G = O.GROUPS[name]()
    enc, dec, w = codec(G, gid)
    order = G.group_order_int()
    rng = random.Random(9 + gid)
    n = 700
    ws = [rng.randrange(order) for _ in range(n)]
    al = [rng.randrange(order) for _ in range(n)]
    c = rng.randrange(min(order, 1 << 256))
    cat = lambda xs: b"".join(enc(x) for x in xs)
    out = capi.dleq_responses(gid, cat(ws), cat(al), enc(c), threads=3)
    want = [O.dleq_response(G, a_w, a_al, c) % order for a_w, a_al in zip(ws, al)]                  # dleq.rs:42-50
    assert [dec(out[i * w:(i + 1) * w]) for i in range(n)] == want
    cs = [rng.randrange(order) for _ in range(n)]
    out = capi.dleq_responses(gid, cat(ws), cat(al), cat(cs))
    assert [dec(out[i * w:(i + 1) * w]) for i in range(n)] == [O.dleq_response(G, x, y, z) % order for x, y, z in zip(ws, al, cs)]
    for t in (1, 2, 17):
        coeffs = [rng.randrange(order) for _ in range(t)]
        positions = [1, 2, 3, 65536, (1 << 40) + 9, (1 << 62) + 1] + [rng.randrange(1, 1 << 20) for _ in range(300)]
        out = capi.poly_eval(gid, cat(coeffs), positions, threads=2)
        want = [O.poly_get_value(coeffs, i) % order for i in positions]                             # polynomial.rs:50-58
        assert [dec(out[i * w:(i + 1) * w]) for i in range(len(positions))] == want
    # the reference's known answer: P(278) mod 15486967 = 4115179 is a different modulus; instead pin position 0
    coeffs = [rng.randrange(order) for _ in range(5)]
    assert dec(capi.poly_eval(gid, cat(coeffs), [0])) == coeffs[0]
    if gid == 0:
        with pytest.raises(capi.EngineError):
            capi.poly_eval(0, cat(coeffs), [3, -1])
