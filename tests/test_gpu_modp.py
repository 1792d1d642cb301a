"""GPU parity tests (MODP-2048): the HIP engine, called through its C ABI, against the CPU oracle.
Bit-exact equality is the bar (integer/byte work)."""
import random

import pytest

import mpvss_oracle as O
from helpers import EB, cat, make_modp_instance, modp_keygen, split
from mpvss_rs_amd import capi

pytestmark = pytest.mark.gpu

G = O.ModpGroup()
Q = G.q
EDGE = [0, 1, 2, 4, Q - 1, Q, Q + 1, (1 << 2048) - 1, 1 << 2047, (1 << 2040) - 1, 0xFFFFFFF, 1 << 28, (1 << 56) - 1]


def fx(v):
    return v.to_bytes(EB, "big")


def test_batch_mul_matches_oracle(engine):
    rng = random.Random(11)
    a = [rng.randrange(1 << 2048) for _ in range(70)] + EDGE + EDGE
    b = [rng.randrange(1 << 2048) for _ in range(70)] + EDGE + EDGE[::-1]
    out = split(engine.batch_mul(b"".join(map(fx, a)), b"".join(map(fx, b))))
    assert out == [(x * y) % Q for x, y in zip(a, b)]          # modp.rs:130-132


def test_batch_exp_matches_oracle(engine):
    rng = random.Random(12)
    bases = [rng.randrange(1 << 2048) for _ in range(20)] + EDGE + [2, 4, 3]
    exps = [rng.randrange(1 << 2048) for _ in range(20)] + EDGE[::-1] + [0, 1, (1 << 2048) - 1]
    out = split(engine.batch_exp(b"".join(map(fx, bases)), b"".join(map(fx, exps))))
    assert out == [pow(x, e, Q) for x, e in zip(bases, exps)]  # modp.rs:122-128


def test_batch_exp_fixed_base(engine):
    rng = random.Random(13)
    exps = [rng.randrange(Q - 1) for _ in range(33)] + [0, 1, 2]
    for base in (2, 4, rng.randrange(Q)):
        out = split(engine.batch_exp_fixed_base(fx(base), b"".join(map(fx, exps))))
        assert out == [pow(base, e, Q) for e in exps]


def test_single_element_and_ragged_counts(engine):
    rng = random.Random(14)
    for n in (1, 2, 15, 16, 17, 31, 33):
        a = [rng.randrange(Q) for _ in range(n)]
        b = [rng.randrange(Q) for _ in range(n)]
        assert split(engine.batch_mul(b"".join(map(fx, a)), b"".join(map(fx, b)))) == [x * y % Q for x, y in zip(a, b)]
    assert engine.batch_mul(b"", b"") == b""


def test_commit_eval_matches_reference_loop(engine):
    rng = random.Random(15)
    t = 5
    commitments = [pow(4, rng.randrange(Q - 1), Q) for _ in range(t)]
    positions = list(range(1, 41)) + [0, 12345, (1 << 40) + 5, (1 << 62) + 1, 65536, 65535]
    out = split(engine.commit_eval(cat(G, commitments), positions))
    assert out == [O.commitment_eval(G, commitments, i) for i in positions]   # participant.rs:423-434


def test_commit_eval_edge_commitments(engine):
    # unreduced, zero and one commitments; t = 1
    for commitments in ([0, 5, 7], [Q + 3, Q - 1, 1], [1, 1, 1], [9]):
        positions = [1, 2, 3, 17]
        out = split(engine.commit_eval(cat(G, [c % (1 << 2048) for c in commitments]), positions))
        assert out == [O.commitment_eval(G, commitments, i) for i in positions]


def test_dleq_commitments(engine):
    rng = random.Random(16)
    n = 19
    h1 = [rng.randrange(Q) for _ in range(n)]
    g2 = [rng.randrange(Q) for _ in range(n)]
    h2 = [rng.randrange(Q) for _ in range(n)]
    r = [rng.randrange(Q - 1) for _ in range(n)]
    c = rng.randrange(1 << 256)
    a1, a2 = engine.dleq_commitments(fx(4), cat(G, h1), cat(G, g2), cat(G, h2), cat(G, r), fx(c), False)
    exp = [O.dleq_verifier_commitments(G, 4, h1[i], g2[i], h2[i], r[i], c) for i in range(n)]   # dleq.rs:66-84
    assert split(a1) == [e[0] for e in exp]
    assert split(a2) == [e[1] for e in exp]
    # per-share, full-width challenges
    cs = [rng.randrange(Q - 1) for _ in range(n)]
    a1, a2 = engine.dleq_commitments(fx(2), cat(G, h1), cat(G, g2), cat(G, h2), cat(G, r), cat(G, cs), True)
    exp = [O.dleq_verifier_commitments(G, 2, h1[i], g2[i], h2[i], r[i], cs[i]) for i in range(n)]
    assert split(a1) == [e[0] for e in exp]
    assert split(a2) == [e[1] for e in exp]


@pytest.mark.parametrize("n,t,seed", [(3, 3, 1), (20, 4, 2), (37, 7, 3)])
def test_verify_distribution_honest_and_tampered(engine, n, t, seed):
    g, privs, pks, coeffs, ws, box = make_modp_instance(n, t, seed)
    flat = O.box_to_flat(g, box)
    res = engine.verify_distribution(flat["commitments"], flat["positions"], flat["publickeys"], flat["shares"],
                                     flat["responses"], flat["challenge"], dump=True)
    trace = {}
    assert O.verify_distribution_shares(g, box, trace) is True
    assert res["verdict"] is True
    assert res["digest"] == trace["digest"] == box["_digest"]
    assert split(res["X"]) == trace["X"] == box["_X"]
    assert split(res["a1"]) == trace["a1"] == box["_a1"]
    assert split(res["a2"]) == trace["a2"] == box["_a2"]
    # negative tests the reference lacks: one flipped bit anywhere => reject, same digest as oracle
    rng = random.Random(seed)
    for field in ("responses", "shares", "commitments", "publickeys", "challenge"):
        bad = dict(flat)
        buf = bytearray(flat[field])
        idx = rng.randrange(len(buf))
        buf[idx] ^= 1 << rng.randrange(8)
        bad[field] = bytes(buf)
        res = engine.verify_distribution(bad["commitments"], bad["positions"], bad["publickeys"], bad["shares"],
                                         bad["responses"], bad["challenge"], dump=False)
        # oracle on the same tampered flat box
        obox = flat_to_box(g, bad)
        otrace = {}
        overdict = O.verify_distribution_shares(g, obox, otrace)
        assert res["verdict"] == overdict
        assert res["digest"] == otrace["digest"]
        assert res["verdict"] is False


def flat_to_box(g, flat):
    n, t = flat["n"], flat["t"]
    cm = split(flat["commitments"])
    pk = split(flat["publickeys"])
    sh = split(flat["shares"])
    rs = split(flat["responses"])
    keys = [g.element_to_bytes(p) for p in pk]
    return {"commitments": cm, "publickeys": pk,
            "positions": dict(zip(keys, flat["positions"])),
            "shares": dict(zip(keys, sh)), "responses": dict(zip(keys, rs)),
            "challenge": int.from_bytes(flat["challenge"], "big"), "U": 0}


def test_verify_distribution_minimal_length_framing(engine):
    """X_1 = 2^36 is hashed as 5 bytes, not 256."""
    g = O.ModpGroup()
    pks = [g.generate_public_key(k) for k in (5, 9)]
    box = O.distribute_secret(g, 0x4869, pks, 2, [7, 11], [13, 17])
    assert box["_digest"].hex() == "5b1da312869d00956a82fb60dc26147ec3d37e2b751fa633f11f74b8f567e333"
    flat = O.box_to_flat(g, box)
    res = engine.verify_distribution(flat["commitments"], flat["positions"], flat["publickeys"], flat["shares"],
                                     flat["responses"], flat["challenge"], dump=True)
    assert res["verdict"] is True
    assert res["digest"] == box["_digest"]
    assert split(res["X"])[0] == 1 << 36


def test_verify_distribution_empty_box(engine):
    import hashlib
    res = engine.verify_distribution(b"", [], b"", b"", b"", bytes(256))
    assert res["digest"] == hashlib.sha256(b"").digest()
    assert res["verdict"] is False


def test_verify_shares_batch(engine):
    g, privs, pks, coeffs, ws, box = make_modp_instance(9, 3, 21)
    rng = random.Random(5)
    sbs = [O.extract_secret_share(g, box, k, modp_keygen(g, rng)) for k in privs]
    keys = [g.element_to_bytes(p) for p in pks]
    Y = [box["shares"][k] for k in keys]
    S = [sb["share"] for sb in sbs]
    c = [sb["challenge"] for sb in sbs]
    r = [sb["response"] for sb in sbs]
    # tamper a few
    r[2] ^= 1
    S[5] = S[5] * 2 % Q
    c[7] ^= 1 << 200
    verdicts = engine.verify_shares(cat(g, pks), cat(g, S), cat(g, Y), cat(g, c), cat(g, r))
    exp = [O.dleq_verify(g, g.generator(), pks[i], S[i], Y[i], c[i], r[i]) for i in range(9)]   # participant.rs:361-386
    assert list(verdicts) == [int(v) for v in exp]
    assert exp == [True, True, False, True, True, False, True, False, True]


def test_verify_shares_device_hash_framing_edge_cases(engine):
    """K7 hashes on the device: elements with leading zero bytes are framed with their minimal length (255 bytes,
    1 byte, zero as one 0x00 byte -- modp.rs:150-152), elements above q as given, and a challenge >= 2^256 never
    verifies even when its low 256 bits equal the hash."""
    rng = random.Random(77)
    pk = [5, 0, 1 << 2039, (1 << 2040) - 1, rng.randrange(Q), Q + 3, (1 << 2048) - 1, 255, 256, rng.randrange(Q)]
    Y = [1 << 2039, 7, 0, rng.randrange(1 << 2033), 1, rng.randrange(Q), 2, (1 << 2047), 65535, rng.randrange(Q)]
    n = len(pk)
    S = [rng.randrange(Q) for _ in range(n)]
    r = [rng.randrange(Q - 1) for _ in range(n)]
    r[3] = 0
    c = [rng.randrange(1 << 256) for _ in range(n)]
    exp = [O.dleq_verify(G, G.generator(), pk[i], S[i], Y[i], c[i], r[i]) for i in range(n)]
    got = engine.verify_shares(cat(G, pk), cat(G, S), cat(G, Y), cat(G, c), cat(G, r))
    assert list(got) == [int(v) for v in exp] and not any(exp)
    # honest proofs, then two challenges lifted above 2^256: their low 256 bits still equal the hash
    g, privs, pks, coeffs, ws, box = make_modp_instance(6, 3, 23)
    sbs = [O.extract_secret_share(g, box, k, modp_keygen(g, rng)) for k in privs]
    keys = [g.element_to_bytes(p) for p in pks]
    Yh = [box["shares"][k] for k in keys]
    Sh = [sb["share"] for sb in sbs]
    ch = [sb["challenge"] for sb in sbs]
    rh = [sb["response"] for sb in sbs]
    ch[1] += 1 << 256
    ch[4] += 1 << 2047
    got = engine.verify_shares(cat(g, pks), cat(g, Sh), cat(g, Yh), cat(g, ch), cat(g, rh))
    assert list(got) == [1, 0, 1, 1, 0, 1]
    assert [O.dleq_verify(g, g.generator(), pks[i], Sh[i], Yh[i], ch[i], rh[i]) for i in range(6)] == \
        [True, False, True, True, False, True]


def test_verify_shares_transcript_with_short_elements_verifies(engine):
    """Share boxes whose pk and Y have one-byte encodings (0 and 1) and whose c IS the hash of the resulting transcript
    -- possible because 0^c and 1^c do not depend on c: the verdict must be 1, i.e. the device frames short elements
    exactly as the reference does (a wrong length prefix or padding would change the hash)."""
    rng = random.Random(78)
    pk, Y, S, r, c = [], [], [], [], []
    for p, y in ((1, 1), (0, 1), (1, 0), (0, 0)):
        s_, r_ = rng.randrange(Q), rng.randrange(1, Q - 1)
        a1 = pow(2, r_, Q) * p % Q              # G^r * pk^c with pk in {0, 1} and c > 0
        a2 = pow(s_, r_, Q) * y % Q
        c_ = G.hash_to_scalar(O.sha256(O.append_transcript(G, p, y, a1, a2)))
        assert c_ > 0 and O.dleq_verify(G, G.generator(), p, s_, y, c_, r_) is True
        pk.append(p); Y.append(y); S.append(s_); r.append(r_); c.append(c_)
    pk += [5, 1 << 2039]; Y += [1 << 2039, 3]; S += [7, 9]; r += [11, 13]; c += [rng.randrange(1 << 256)] * 2
    got = engine.verify_shares(cat(G, pk), cat(G, S), cat(G, Y), cat(G, c), cat(G, r))
    assert list(got) == [1, 1, 1, 1, 0, 0]


def test_verify_shares_block_api_keeps_batches_in_flight(engine):
    """verify_shares_compute / _absorb: several batches enqueued, absorbed in FIFO order; a device buffer receives the
    same verdict bytes; the two block kinds do not mix."""
    import torch
    g, privs, pks, coeffs, ws, box = make_modp_instance(9, 3, 21)
    rng = random.Random(5)
    sbs = [O.extract_secret_share(g, box, k, modp_keygen(g, rng)) for k in privs]
    keys = [g.element_to_bytes(p) for p in pks]
    Y = [box["shares"][k] for k in keys]
    S = [sb["share"] for sb in sbs]
    c = [sb["challenge"] for sb in sbs]
    rs = [sb["response"] for sb in sbs]
    batches, want = [], []
    for k in range(5):
        r = list(rs)
        r[k] ^= 1                                  # batch k: share k tampered
        batches.append((cat(g, pks), cat(g, S), cat(g, Y), cat(g, c), cat(g, r)))
        want.append(bytes(0 if i == k else 1 for i in range(9)))
    dev = [torch.zeros(9, dtype=torch.uint8, device="cuda") for _ in batches]
    torch.cuda.synchronize()              # torch's fills run on torch's stream, the engine's kernels on the engine's
    for b, d in zip(batches, dev):
        engine.verify_shares_compute(*b, verdicts_dev_ptr=d.data_ptr())
    with pytest.raises(capi.EngineError):          # the oldest block is a verify_share batch
        engine.verify_block_absorb(capi.transcript_init())
    got = [engine.verify_shares_absorb(9) for _ in batches]
    assert got == want
    assert [bytes(d.cpu().numpy().tobytes()) for d in dev] == want
    with pytest.raises(capi.EngineError):
        engine.verify_shares_absorb(9)


def test_distribute_group_part(engine):
    g, privs, pks, coeffs, ws, box = make_modp_instance(11, 4, 31)
    order = g.group_order_int()
    positions = list(range(1, 12))
    p_vals = [O.poly_get_value(coeffs, i) % order for i in positions]
    res = engine.distribute(cat(g, box["commitments"]), positions, cat(g, pks), cat(g, p_vals), cat(g, ws))
    keys = [g.element_to_bytes(p) for p in pks]
    assert split(res["X"]) == box["_X"]
    assert split(res["Y"]) == [box["shares"][k] for k in keys]
    assert split(res["a1"]) == box["_a1"]
    assert split(res["a2"]) == box["_a2"]
    assert res["digest"] == box["_digest"]


def test_dealer_shared_squarings_bucket_path(engine):
    """From 1024 shares per block the dealer computes Y_i = y_i^P(i) (participant.rs:219) and a2_i = y_i^w_i
    (dleq.rs:213-216) with ONE chain of squarings per share and right-to-left 5-bit buckets (k_modp_twin_exp_buckets,
    k_modp_bucket_combine).  Every output against Python's pow, including the exponents that leave buckets empty, fill
    only one, put every window in the same bucket, or are 0 / not reduced; the 4-bit-window path on the same inputs."""
    rng = random.Random(0xB0C4E7)
    n = 1024 + 40
    same = sum(d << (5 * k) for k in range(410) for d in [9]) % (1 << 2048)          # every window the same digit
    top = 7 << 2045                                                                  # only the (3-bit) top window
    edge_e = [0, 1, 31, 32, Q - 2, Q - 1, Q, (1 << 2048) - 1, 1 << 2047, same, top, (1 << 2045) - 1, 1 << 5, 17 << 2040]
    ys = [rng.randrange(2, Q) for _ in range(n - len(EDGE))] + EDGE
    ps = [rng.randrange(1 << 2048) for _ in range(n - 2 * len(edge_e))] + edge_e + edge_e[::-1]
    ws = [rng.randrange(Q - 1) for _ in range(n - 2 * len(edge_e))] + edge_e[::-1] + edge_e
    rng.shuffle(ys)
    args = (b"".join(map(fx, ys)), b"".join(map(fx, ps)), b"".join(map(fx, ws)))
    engine.distribute_compute(None, None, *args)
    st, X, Y, a1, a2 = engine.distribute_absorb(capi.transcript_init(), n)
    from helpers import modp_dual_pow_chunk, parallel_map

    def pows(bases, exps):                 # base^exp mod q for every pair, on oracle-only worker processes
        items = [(b, e, 1, 0) for b, e in zip(bases, exps)]
        step = max(1, len(items) // 32)
        return [v for part in parallel_map(modp_dual_pow_chunk, [items[k:k + step] for k in range(0, len(items), step)]) for v in part]
    assert split(Y) == pows(ys, ps)
    assert split(a2) == pows(ys, ws)
    assert split(X) == pows([4] * n, ps) and split(a1) == pows([4] * n, ws)     # g = 4, modp.rs:65-66
    # below 1024 shares: left-to-right windows, same results
    m = 200
    engine.distribute_compute(None, None, *(a[-m * EB:] for a in args))
    st2, X2, Y2, a12, a22 = engine.distribute_absorb(capi.transcript_init(), m)
    assert Y2 == Y[-m * EB:] and a22 == a2[-m * EB:] and X2 == X[-m * EB:]


def test_distribute_block_api_and_dealer_shortcut(engine):
    """The dealer's blocks in compute / absorb form: several in flight, X_i either from the commitments (the reference's
    loop, participant.rs:207-215) or as g^P(i) through the comb (commitments = None) -- identical outputs and transcript;
    a box cut into two blocks carries the running hash state from one to the next."""
    g, privs, pks, coeffs, ws, box = make_modp_instance(11, 4, 31)
    order = g.group_order_int()
    positions = list(range(1, 12))
    p_vals = [O.poly_get_value(coeffs, i) % order for i in positions]
    keys = [g.element_to_bytes(p) for p in pks]
    cm = cat(g, box["commitments"])
    args = (cat(g, pks), cat(g, p_vals), cat(g, ws))
    engine.distribute_compute(cm, positions, *args)
    engine.distribute_compute(None, None, *args)
    for _ in range(2):
        st, X, Y, a1, a2 = engine.distribute_absorb(capi.transcript_init(), 11)
        assert split(X) == box["_X"] and split(Y) == [box["shares"][k] for k in keys]
        assert split(a1) == box["_a1"] and split(a2) == box["_a2"]
        assert capi.transcript_verdict(st, bytes(256))[1] == box["_digest"]
    # two blocks of one box (5 + 6 shares), state carried across
    cut = 5 * 256
    engine.distribute_compute(cm, positions[:5], args[0][:cut], args[1][:cut], args[2][:cut])
    engine.distribute_compute(None, None, args[0][cut:], args[1][cut:], args[2][cut:])
    st = capi.transcript_init()
    st, X1, _, _, _ = engine.distribute_absorb(st, 5)
    st, X2, _, _, _ = engine.distribute_absorb(st, 6)
    assert split(X1 + X2) == box["_X"]
    assert capi.transcript_verdict(st, bytes(256))[1] == box["_digest"]
    # scalar side behind the C ABI closes the box: P(i), challenge, responses (participant.rs:200-202, 251-264)
    pv = capi.poly_eval(0, cat(g, coeffs), positions)
    assert split(pv) == p_vals
    r = capi.dleq_responses(0, cat(g, ws), pv, fx(box["challenge"]))
    assert split(r) == [box["responses"][k] for k in keys]


def test_rejects_bad_arguments(engine):
    from mpvss_rs_amd import EngineError
    with pytest.raises(EngineError):
        engine.commit_eval(fx(4), [-1])
    with pytest.raises(EngineError):
        engine.distribute(cat(G, [4, 4, 4]), [1, 2], cat(G, [2, 2]), cat(G, [1, 1]), cat(G, [1, 1]))   # t > n


def test_every_pair_layout_kernel_against_the_dealer_and_the_fast_form():
    """MPVSS_PAIR=31 sends the window tables, g^r and a1 through the pair layout as well (by default only a2 goes there: the
    pipeline is faster that way).  An 8200-share box is dealt and verified with dumps in a child process under that
    switch (it is read once per process): verdict, the dealer's digest, X / a1 / a2 equal to the dealer's everywhere and to
    the fast form of the reference arithmetic on a sample; one flipped response bit is rejected."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys, random, hashlib, math
sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)
from helpers import MODP_Q as Q, MODP_ORDER as ORDER, modp_fast_share, poly_values
from mpvss_rs_amd import Engine
fx = lambda v: v.to_bytes(256, "big")
eng = Engine(0)
rng = random.Random(515)
n, t = 8200, 17
def keygen():
    while True:
        k = rng.randrange(Q)
        if math.gcd(k, ORDER) == 1: return k
coeffs = [rng.randrange(ORDER) for _ in range(t)]
privs = [keygen() for _ in range(n)]; wits = [keygen() for _ in range(n)]
pos = list(range(11, 11 + n))
pv = poly_values(coeffs, pos, ORDER)
cm = eng.batch_exp_fixed_base(fx(4), b"".join(map(fx, coeffs)))
pk = eng.batch_exp_fixed_base(fx(2), b"".join(map(fx, privs)))
d = eng.distribute(cm, pos, pk, b"".join(map(fx, pv)), b"".join(map(fx, wits)))
c = int.from_bytes(hashlib.sha256(d["digest"]).digest(), "big") %% ((Q - 1) // 2)
r = b"".join(fx((w - p * c) %% ORDER) for w, p in zip(wits, pv))
res = eng.verify_distribution(cm, pos, pk, d["Y"], r, fx(c), dump=True)
assert res["verdict"] is True and res["digest"] == d["digest"]
assert (res["X"], res["a1"], res["a2"]) == (d["X"], d["a1"], d["a2"])
for i in sorted(random.Random(3).sample(range(n), 12)) + [0, n - 1]:
    s = slice(i * 256, (i + 1) * 256)
    assert modp_fast_share((cm, pos[i], pk[s], d["Y"][s], r[s], fx(c))) == (res["X"][s], res["a1"][s], res["a2"][s]), i
bad = bytearray(r); bad[7321 * 256 + 200] ^= 8
assert eng.verify_distribution(cm, pos, pk, d["Y"], bytes(bad), fx(c))["verdict"] is False
print("pair15 ok")
""" % (root, os.path.join(root, "oracle"), os.path.join(root, "tests"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, MPVSS_PAIR="31"), timeout=900)
    assert out.returncode == 0 and "pair15 ok" in out.stdout, out.stdout[-1500:] + out.stderr[-3000:]
