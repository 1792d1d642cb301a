"""Batched extract_secret_share against the oracle, all three groups."""
import random

import pytest

import mpvss_oracle as O
from helpers import cat, make_modp_instance, modp_keygen, split
from mpvss_rs_amd import capi

pytestmark = pytest.mark.gpu


def test_modp_extract_shares(engine):
    g, privs, pks, coeffs, ws, box = make_modp_instance(7, 3, 13)
    rng = random.Random(2)
    wit = [modp_keygen(g, rng) for _ in privs]
    keys = [g.element_to_bytes(p) for p in pks]
    Y = [box["shares"][k] for k in keys]
    xinv = [O.mod_inverse(x, g.group_order_int()) for x in privs]
    S, c = engine.extract_shares(cat(g, pks), cat(g, Y), cat(g, xinv), cat(g, wit))
    exp = [O.extract_secret_share(g, box, x, w) for x, w in zip(privs, wit)]          # participant.rs:294-353
    assert split(S) == [e["share"] for e in exp]
    assert split(c) == [e["challenge"] for e in exp]
    # the host-side response closes the proof (dleq.rs:42-50) and the batch verifier accepts it
    r = [O.dleq_response(g, w, x, ci) for w, x, ci in zip(wit, privs, split(c))]
    assert r == [e["response"] for e in exp]
    assert list(engine.verify_shares(cat(g, pks), S, cat(g, Y), c, cat(g, r))) == [1] * 7


@pytest.mark.parametrize("name", ["secp256k1", "ristretto255"])
def test_ec_extract_shares(engine, name):
    G = O.GROUPS[name]()
    gid = capi.GROUP_SECP256K1 if name == "secp256k1" else capi.GROUP_RISTRETTO255
    rng = random.Random(17)
    order = G.group_order_int()
    n, t = 6, 3
    privs = [rng.randrange(1, order) for _ in range(n)]
    pks = [G.generate_public_key(k) for k in privs]
    box = O.distribute_secret(G, 0x1234, pks, t, [rng.randrange(order) for _ in range(t)],
                              [rng.randrange(1, order) for _ in range(n)])
    wit = [rng.randrange(1, order) for _ in range(n)]
    keys = [G.element_to_bytes(p) for p in pks]
    enc = lambda pts: b"".join(G.element_to_bytes(p) for p in pts)
    sc = lambda ks: b"".join(G.scalar_to_bytes(k) for k in ks)
    Y = [box["shares"][k] for k in keys]
    S, c = engine.ec_extract_shares(gid, enc(pks), enc(Y), sc([G.scalar_inverse(x) for x in privs]), sc(wit))
    exp = [O.extract_secret_share(G, box, x, w) for x, w in zip(privs, wit)]
    assert S == enc([e["share"] for e in exp])
    assert c == sc([e["challenge"] for e in exp])
    r = [O.dleq_response(G, w, x, e["challenge"]) for w, x, e in zip(wit, privs, exp)]
    assert list(engine.ec_verify_shares(gid, enc(pks), S, enc(Y), c, sc(r))) == [1] * n


def test_modp_verify_shares_large_batch_uses_wide_windows(engine):
    """Batches of 1024 share boxes and more take the 6-bit-window path for S^r (per-share challenges): honest proofs
    built with the engine's own extract_shares must verify, tampered ones must not."""
    g = O.ModpGroup()
    rng = random.Random(91)
    n = 1100
    order = g.group_order_int()
    privs = [modp_keygen(g, rng) for _ in range(8)]
    privs = [privs[i % 8] for i in range(n)]                           # few distinct keys (Python pow is slow)
    pk_of = {x: g.generate_public_key(x) for x in set(privs)}
    pks = [pk_of[x] for x in privs]
    ps = [rng.randrange(order) for _ in range(16)]
    Y_of = {(x, p): pow(pk_of[x], p, g.q) for x in set(privs) for p in ps}
    pvals = [ps[i % 16] for i in range(n)]
    Y = [Y_of[(x, p)] for x, p in zip(privs, pvals)]
    wit = [rng.randrange(1, order) for _ in range(n)]
    xinv_of = {x: O.mod_inverse(x, order) for x in set(privs)}
    S, c = engine.extract_shares(cat(g, pks), cat(g, Y), cat(g, [xinv_of[x] for x in privs]), cat(g, wit))
    assert split(S)[:3] == [pow(2, p, g.q) for p in pvals[:3]]         # S_i = Y_i^(1/x_i) = G^p_i
    r = [O.dleq_response(g, w, x, ci) for w, x, ci in zip(wit, privs, split(c))]
    rb = bytearray(cat(g, r))
    for i in (5, 600, 1099):
        rb[i * 256 + 200] ^= 1
    verdicts = list(engine.verify_shares(cat(g, pks), S, cat(g, Y), c, bytes(rb)))
    assert verdicts == [0 if i in (5, 600, 1099) else 1 for i in range(n)]
