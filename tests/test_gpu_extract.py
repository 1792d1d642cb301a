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


def test_modp_extract_shares_block_form(engine):
    """mpvss_modp_extract_shares_compute / _absorb: three batches in flight (320 participants each, other witnesses per
    batch), absorbed in order -- S and c equal to the synchronous call's, a few shares against the oracle, the proofs verify;
    a batch holding an encrypted share that is 0 mod q is refused (it is no group element) and the context stays usable."""
    n, t = 320, 4
    g, privs, pks, coeffs, ws, box = make_modp_instance(n, t, 29)
    keys = [g.element_to_bytes(p) for p in pks]
    Y = [box["shares"][k] for k in keys]
    xinv = cat(g, [O.mod_inverse(x, g.group_order_int()) for x in privs])
    pkb, Yb = cat(g, pks), cat(g, Y)
    batches = []
    for b in range(3):
        rng = random.Random(100 + b)
        batches.append([modp_keygen(g, rng) for _ in privs])
    for wit in batches:
        assert engine.extract_shares_compute(pkb, Yb, xinv, cat(g, wit)) == n
    assert engine.blocks_in_flight()[0] == 3
    got = [engine.extract_shares_absorb(n) for _ in batches]
    assert engine.blocks_in_flight() == (0, 0)
    for b, (wit, (S, c)) in enumerate(zip(batches, got)):
        # the synchronous call takes the two dependent exponentiations at this size: a second path to the same bytes
        assert (S, c) == engine.extract_shares(pkb, Yb, xinv, cat(g, wit)), f"batch {b}"
        for i in (0, 7, n - 1):
            e = O.extract_secret_share(g, box, privs[i], wit[i])                          # participant.rs:294-353
            assert split(S)[i] == e["share"] and split(c)[i] == e["challenge"]
        r = [O.dleq_response(g, w, x, ci) for w, x, ci in zip(wit, privs, split(c))]
        assert list(engine.verify_shares(pkb, S, Yb, c, cat(g, r))) == [1] * n
    zero = bytearray(Yb); zero[5 * 256:6 * 256] = g.q.to_bytes(256, "big")
    with pytest.raises(capi.EngineError, match="0 mod q"):
        engine.extract_shares_compute(pkb, bytes(zero), xinv, cat(g, batches[0]))
    assert engine.blocks_in_flight() == (0, 0)
    assert engine.extract_shares_compute(pkb, Yb, xinv, cat(g, batches[0])) == n
    assert engine.extract_shares_absorb(n)[0] == engine.extract_shares(pkb, Yb, xinv, cat(g, batches[0]))[0]


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
    # compute / absorb form: three batches in flight (honest, a tampered response, a tampered share), absorbed in order;
    # then a batch with an element that is no encoding -- reported by the absorbing call, the context stays usable
    rb = bytearray(sc(r)); rb[2 * 32 + (31 if name == "secp256k1" else 0)] ^= 1
    Sb = bytearray(S); L = len(S) // n; Sb[4 * L:5 * L] = S[3 * L:4 * L]
    for args in ((enc(pks), S, enc(Y), c, sc(r)), (enc(pks), S, enc(Y), c, bytes(rb)), (enc(pks), bytes(Sb), enc(Y), c, sc(r))):
        assert engine.ec_verify_shares_compute(gid, *args) == n
    assert list(engine.ec_verify_shares_absorb(n)) == [1] * n
    assert list(engine.ec_verify_shares_absorb(n)) == [1, 1, 0, 1, 1, 1]
    assert list(engine.ec_verify_shares_absorb(n)) == [1, 1, 1, 1, 0, 1]
    with pytest.raises(capi.EngineError):
        engine.ec_verify_shares_absorb(n)                    # nothing left in flight
    broken = bytearray(enc(pks)); broken[1 * L:2 * L] = (b"\x05" + bytes(32)) if name == "secp256k1" else bytes.fromhex("01" + "00" * 31)
    engine.ec_verify_shares_compute(gid, bytes(broken), S, enc(Y), c, sc(r))
    engine.ec_verify_shares_compute(gid, enc(pks), S, enc(Y), c, sc(r))
    with pytest.raises(capi.EngineError, match="share boxes: element 1"):
        engine.ec_verify_shares_absorb(n)
    assert list(engine.ec_verify_shares_absorb(n)) == [1] * n


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


def test_modp_extract_shares_large_batch_shares_the_squarings(engine):
    """From 1024 shares (host buffers) S = Y^(1/x) and a2 = S^w = Y^(w/x) come out of one chain of squarings, with
    e2 = w * (1/x) mod (q-1) formed on the host.  The challenges hash a1 and a2, so they pin a2: compared with the
    two-exponentiation path on a slice (< 1024 shares) of the same inputs and with Python's pow; exponents whose
    product is a multiple of q-1, w = 0 and an unreduced 1/x are among them.  A batch that contains a Y that is
    0 mod q must take the two-step path (0^e is not periodic in e): S = 0 there, and the results still agree."""
    g = O.ModpGroup()
    q, order = g.q, g.group_order_int()
    rng = random.Random(0xE77)
    n = 1024 + 36
    ys = [rng.randrange(2, q) for _ in range(n)]
    xinv = [rng.randrange(1, order) for _ in range(n)]
    wit = [rng.randrange(1, order) for _ in range(n)]
    xinv[1], wit[1] = order // 2, 2                  # product = q - 1: reduced exponent 0
    wit[2] = 0
    xinv[3] = (1 << 2048) - 1                        # not reduced
    xinv[4], wit[4] = order, 5                       # 1/x = q - 1 itself
    pks = [pow(2, rng.randrange(order), q) for _ in range(8)]
    pk = [pks[i % 8] for i in range(n)]
    S, c = engine.extract_shares(cat(g, pk), cat(g, ys), cat(g, xinv), cat(g, wit))
    assert split(S) == [pow(y, e, q) for y, e in zip(ys, xinv)]
    m = 300                                          # the slice goes through two dependent exponentiations
    S2, c2 = engine.extract_shares(cat(g, pk[:m]), cat(g, ys[:m]), cat(g, xinv[:m]), cat(g, wit[:m]))
    assert S2 == S[:m * 256] and c2 == c[:m * 256]
    for i in (0, 1, 2, 3, 4, n - 1):                 # participant.rs:329-343 with a1 = G^w, a2 = S^w
        a1, a2 = pow(2, wit[i], q), pow(pow(ys[i], xinv[i], q), wit[i], q)
        digest = O.sha256(O.append_transcript(g, pk[i], ys[i], a1, a2))
        assert split(c)[i] == g.hash_to_scalar(digest), i
    ys0 = list(ys); ys0[7] = 0; ys0[9] = q
    S0, c0 = engine.extract_shares(cat(g, pk), cat(g, ys0), cat(g, xinv), cat(g, wit))
    assert split(S0)[7] == 0 and split(S0)[9] == 0 and split(S0)[8] == split(S)[8]
    assert c0[:7 * 256] == c[:7 * 256] and c0[10 * 256:] == c[10 * 256:]
