"""The row layout (mpvss_rs_amd/csrc/bn_row.h: 16 lanes per number, the latency-bound launches of a box that has the chip to
itself): (a) one Montgomery product / squaring per operand pair against Python integers on operands at the bounds of the integer
model; (b) the Horner seed kernel k_modp_commit_eval_row behind the X path of a lone call -- the same X, bit for bit, as the quad
kernel (MPVSS_FD_ROW=0), as Horner's rule for every position (MPVSS_FD=0), and as Python's pow on a sample (participant.rs:423-434)."""
import os
import random
import struct
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, L = 29, 72
M29 = (1 << W) - 1
LIM = M29 + 512


def test_row_product_and_square_on_edge_operands(tmp_path):
    import mpvss_oracle as O
    N = O.ModpGroup().q
    R = 1 << (W * L)
    exe = os.path.join(ROOT, "tests", "_build", "row_unit")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "mpvss_rs_amd", "csrc"), "../../tests/_build/row_unit"])
    rng = random.Random(0x50E)
    canon = lambda v: [(v >> (W * j)) & M29 for j in range(L)]
    value = lambda l: sum(x << (W * j) for j, x in enumerate(l))

    def loose():          # a value below 2N whose limbs are "almost normalised" (up to 2^29 - 1 + 2^9), as a product leaves them
        while True:
            l = canon(rng.randrange(2 * N))
            for j in range(L - 1):
                if rng.random() < 0.3 and l[j + 1] > 0:
                    l[j] += 1 << W
                    l[j + 1] -= 1
                    l[j] = min(l[j], LIM)
            if value(l) < 2 * N:
                return l
    maxl = [LIM] * 70 + [0, 0]
    while value(maxl) >= 2 * N:
        maxl[69] //= 2
    half = 1 << 1044
    named = [canon(v) for v in (0, 1, 2, N - 1, N, N + 1, 2 * N - 1, 2 * N - 2, half, half - 1, half + 1, M29, 1 << W,
                                (1 << (W * 5)) - 1, 1 << (W * 5), (1 << (W * 70)) - 1)] + [maxl]
    pairs = [(a, b) for a in named for b in named]
    for _ in range(1500):
        pairs.append((canon(rng.randrange(2 * N)), canon(rng.randrange(2 * N))))
    for _ in range(1500):
        pairs.append((loose(), loose()))
    pairs.append((maxl, maxl))
    n = len(pairs)
    assert n % 4 != 0                                                    # the last wave is ragged
    inp, outp = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(inp, "wb") as f:
        f.write(struct.pack("<72I", *canon(N)))
        for a, b in pairs:
            f.write(struct.pack("<72I", *a) + struct.pack("<72I", *b))
    res = subprocess.run([exe, str(inp), str(outp), str(n)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    raw = open(outp, "rb").read()
    assert len(raw) == n * 2 * 288
    Rinv = pow(R, -1, N)
    for i, (a, b) in enumerate(pairs):
        x, y = value(a), value(b)
        for which, (p, q) in enumerate(((x, y), (x, x))):
            limbs = struct.unpack_from("<72I", raw, (2 * i + which) * 288)
            v = value(limbs)
            assert max(limbs) <= LIM and limbs[71] == 0, (i, which, hex(max(limbs)))
            assert v < 2 * N and v % N == p * q * Rinv % N, (i, which)


CHILD = r"""
import os, sys, random, hashlib
sys.path.insert(0, %r)
from mpvss_rs_amd import Engine
EB = 256
eng = Engine(0)
rng = random.Random(77)
sc = lambda k: b"".join(rng.randrange(1 << 2048).to_bytes(EB, "big") for _ in range(k))
out = []
for n, t, p0 in ((4096, 16, 1), (8229, 40, 3), (16384 + 5, 256, 70001), (4200, 17, (1 << 40) + 5)):
    cm = eng.batch_exp_fixed_base((4).to_bytes(EB, "big"), sc(t))
    pos = list(range(p0, p0 + n))
    X = eng.commit_eval(cm, pos)                          # stand-alone X path: row-layout seeds
    out.append(hashlib.sha256(X).hexdigest())
    # the same X inside a verifier's block (quad-layout seeds): the dumped X of a box with arbitrary keys, shares, responses
    junk = sc(n)
    res = eng.verify_distribution(cm, pos, junk, junk, junk, (12345).to_bytes(EB, "big"), dump=True)
    assert res["X"] == X, (n, t, "X of the block path differs from the stand-alone call")
    if os.environ.get("CHECK_POW") == "1":
        Q = int(os.environ["MODP_Q"])
        cs = [int.from_bytes(cm[j * EB:(j + 1) * EB], "big") for j in range(t)]
        for i in (0, 1, n // 2, n - 1):
            want = 1
            for j, c in enumerate(cs):
                want = want * pow(c, (p0 + i) ** j, Q) %% Q
            assert int.from_bytes(X[i * EB:(i + 1) * EB], "big") == want, (n, t, i)
print("DIGESTS " + " ".join(out))
"""


def _child(env):
    import mpvss_oracle as O
    e = dict(os.environ, MODP_Q=str(O.ModpGroup().q), **env)
    out = subprocess.run([sys.executable, "-c", CHILD % ROOT], capture_output=True, text=True, env=e, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("DIGESTS ")][0]
    return line.split()[1:]


def test_row_seeds_give_the_same_x_as_the_quad_seeds_and_as_horner():
    """A stand-alone mpvss_modp_commit_eval over consecutive positions takes forward differences with ALL seeds by Horner's rule in
    the row layout; inside a verifier's block the same X path takes the quad-layout seed kernel.  Four shapes (ragged seed counts,
    a threshold that is not a multiple of 4, positions beyond 2^40): the two X arrays are equal byte for byte, they hash the same
    with MPVSS_FD=0 (Horner for every position, no forward differences at all), and four positions per shape equal Python's pow."""
    row = _child({"CHECK_POW": "1"})
    horner = _child({"MPVSS_FD": "0"})
    assert row == horner


def test_small_batches_through_the_row_layout_on_edge_operands(engine):
    """Small batches (up to 4096 numbers: everything the reference's own tests and examples use) take the row-layout kernels -- the latency
    of one number's chain: ModpGroup::exp (k_modp_dual_exp_row, modp.rs:122-128), the DLEQ verifier's commitments with per-share bases
    and a shared or per-share challenge (dleq.rs:66-84), a fixed non-generator base (one shared table), and X_i by Horner for a small box
    (k_modp_commit_eval_row + k_modp_from_mont, participant.rs:423-434) -- against Python integers on edge operands, at 1, 5, 4096
    numbers (one wave per SIMD) and at 4097 (the quad layout again: the same bytes on the first 4096)."""
    import mpvss_oracle as O
    Q = O.ModpGroup().q
    fx = lambda v: v.to_bytes(256, "big")
    rng = random.Random(0x70A)
    edge_b = [1, 2, Q - 1, Q - 2, 0, (1 << 2047) + 12345, 3, int("f0" * 256, 16) % Q]
    edge_e = [0, 1, 2, Q - 2, 15, 16, (1 << 2047), int("ff" * 256, 16) % (Q - 1), int("0f" * 256, 16), int("f0" * 256, 16) % (Q - 1)]
    for n in (1, 5, 4096, 4097):
        bases = [rng.randrange(2, Q) for _ in range(n)]
        exps = [rng.randrange(Q - 1) for _ in range(n)]
        k = 0
        for b in edge_b:
            for e in edge_e:
                if k < n:
                    bases[k], exps[k] = b, e
                    k += 1
        out = engine.batch_exp(b"".join(map(fx, bases)), b"".join(map(fx, exps)))
        sample = sorted(set(list(range(min(n, k))) + [rng.randrange(n) for _ in range(30)] + [n - 1]))
        for i in sample:
            assert int.from_bytes(out[i * 256:(i + 1) * 256], "big") == pow(bases[i], exps[i], Q), (n, i)
    # the same operands through both layouts: 4097 numbers (quad) against their first 4096 (row)
    n = 4097
    bases = [rng.randrange(2, Q) for _ in range(n)]
    exps = [rng.randrange(Q - 1) for _ in range(n)]
    bb, eb = b"".join(map(fx, bases)), b"".join(map(fx, exps))
    assert engine.batch_exp(bb, eb)[:4096 * 256] == engine.batch_exp(bb[:4096 * 256], eb[:4096 * 256])
    # DLEQ commitments: a1 = g1^r h1^c with a non-generator g1 (one shared table, stride 0), a2 = g2^r h2^c with per-share bases;
    # shared and per-share challenges; a challenge beyond 256 bits (512 windows)
    n = 37
    h1 = [rng.randrange(2, Q) for _ in range(n)]
    g2 = [rng.randrange(2, Q) for _ in range(n)]
    h2 = [rng.randrange(2, Q) for _ in range(n)]
    r = [rng.randrange(Q - 1) for _ in range(n)]
    r[0], r[1], r[2] = 0, Q - 2, 1
    g1 = 7
    flat = lambda xs: b"".join(map(fx, xs))
    for cs in ([rng.randrange(1 << 256) for _ in range(n)], [rng.randrange(1 << 256)] * n, [(1 << 300) + 77] * n):
        per = len(set(cs)) > 1
        cs[0] = 0 if per else cs[0]
        a1, a2 = engine.dleq_commitments(fx(g1), flat(h1), flat(g2), flat(h2), flat(r), flat(cs) if per else fx(cs[0]), per)
        for i in range(n):
            assert int.from_bytes(a1[i * 256:(i + 1) * 256], "big") == pow(g1, r[i], Q) * pow(h1[i], cs[i], Q) % Q, i
            assert int.from_bytes(a2[i * 256:(i + 1) * 256], "big") == pow(g2[i], r[i], Q) * pow(h2[i], cs[i], Q) % Q, i
    # X_i by Horner on the row layout: a box of 300 shares at scattered positions (0 and 2^40 among them), t = 9
    t, n = 9, 300
    cm = [pow(4, rng.randrange(1, Q - 1), Q) for _ in range(t)]
    pos = [rng.randrange(1, 1 << 20) for _ in range(n)]
    pos[0], pos[1], pos[2] = 0, 1 << 40, 1
    X = engine.commit_eval(flat(cm), pos)
    order = Q - 1
    for i in list(range(6)) + [n - 1]:
        want = 1
        for j, c in enumerate(cm):
            want = want * pow(c, pow(pos[i], j, order), Q) % Q
        assert int.from_bytes(X[i * 256:(i + 1) * 256], "big") == want, i


@pytest.mark.parametrize("n,t", [(4096, 64), (3000, 40), (1025, 16)])
def test_one_box_calls_of_up_to_4096_shares_take_the_direct_path(engine, n, t):
    """A one-box verify_distribution call of up to 4096 shares on an uncrowded context: X by Horner on the row layout, a2 by the row-layout
    double exponentiation on the second stream, a1 behind X (BASELINE config C2 as ONE call; mpvss_capi.cpp `small_direct`).  The dealer's
    transcript digest -- computed from X = g^P(i), a1 = g^w, a2 = y^w, i.e. by other kernels and another formula -- must come out of the
    verifier's X_i = prod C_j^(i^j), a1 = g^r X^c, a2 = y^r Y^c for every share; a flipped response or share bit is rejected; twelve
    callers at once (the context is crowded: forward differences and the pair layout again) get the same verdicts and digests."""
    import threading
    EB = 256
    fx = lambda v: v.to_bytes(EB, "big")
    rng = random.Random(n * 7 + t)
    sc = lambda k: b"".join(fx(rng.randrange(1, 1 << 2040)) for _ in range(k))
    pos = list(range(1, n + 1))
    coeffs, wit = sc(t), sc(n)
    pk = engine.batch_exp_fixed_base(fx(2), sc(n))
    cm = engine.batch_exp_fixed_base(fx(4), coeffs)
    box = engine.deal(coeffs, pos, pk, wit)
    ok = engine.verify_distribution(cm, pos, pk, box["Y"], box["responses"], box["challenge"], dump=True)
    assert ok["verdict"] and ok["digest"] == box["digest"]
    assert ok["X"] == box["X"]                      # the verifier's Horner against the dealer's comb
    flip = lambda b, at: b[:at] + bytes([b[at] ^ 1]) + b[at + 1:]
    assert not engine.verify_distribution(cm, pos, pk, box["Y"], flip(box["responses"], (n - 1) * EB + 255), box["challenge"])["verdict"]
    assert not engine.verify_distribution(cm, pos, pk, flip(box["Y"], 7 * EB + 100), box["responses"], box["challenge"])["verdict"]
    res = [None] * 12

    def work(k):
        res[k] = [engine.verify_distribution(cm, pos, pk, box["Y"], box["responses"], box["challenge"]) for _ in range(3)]
    ths = [threading.Thread(target=work, args=(k,)) for k in range(12)]
    [th.start() for th in ths]
    [th.join() for th in ths]
    assert all(v["verdict"] and v["digest"] == box["digest"] for r in res for v in r)
    assert engine.blocks_in_flight() == (0, 0)
