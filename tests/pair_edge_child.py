"""Child process of tests/test_gpu_pair_edges.py: the environment switches that send MORE kernels through the pair layout
(MPVSS_PAIR, MPVSS_FD_PAIR_MIN_T) are read once per process, so the boxes below are verified here, with dumps, and every
X_i / a1_i / a2_i is compared with Python integers (CPython pow on spawned oracle-only workers).

    python pair_edge_child.py tables   -- window tables, g^r (wide comb), a1 = g^r X^c and a2 = y^r Y^c on hostile operands
    python pair_edge_child.py fd       -- k_modp_fd_step_pair at t = 512 with commitments in {1, Q-1, unreduced, 0}

Reference semantics: no validation, modpow reduces (src/groups/modp.rs:122-132,154-156); a1 = g^r X^c, a2 = y^r Y^c
(src/dleq.rs:66-84); X_i = prod_j C_j^(i^j) (src/participant.rs:423-434)."""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

from helpers import MODP_Q as Q, modp_fast_share, modp_reference_x, parallel_map  # noqa: E402

# (torch and the engine are imported by main only: the spawned oracle workers re-import this module)

EB = 256
EDGE = [0, 1, 2, 4, Q - 1, Q, Q + 1, (1 << 2048) - 1, 1 << 2047, (1 << 2040) - 1, 0xFFFFFFF, 1 << 28, (1 << 56) - 1]
EDGE_R = [0, 1, Q - 2, Q - 1, Q, (1 << 2048) - 1, (1 << 2046) - 1, 63 << 2040, 1 << 6, (1 << 2048) - (1 << 2042)]


def fx(v):
    return v.to_bytes(EB, "big")


def cyc(vals, n, rng, every=2):
    """edge values at every `every`-th place, random numbers between them"""
    return [vals[(i // every) % len(vals)] if i % every == 0 else rng.randrange(1 << 2048) for i in range(n)]


def check_box(eng, cm, pos, ys, Ys, rs, c, what, sample=None):
    n = len(pos)
    res = eng.verify_distribution(b"".join(map(fx, cm)), pos, b"".join(map(fx, ys)), b"".join(map(fx, Ys)),
                                  b"".join(map(fx, rs)), fx(c), dump=True)
    idx = list(range(n)) if sample is None else sample
    cmb = b"".join(map(fx, cm))
    outs = parallel_map(modp_fast_share, [(cmb, pos[i], fx(ys[i]), fx(Ys[i]), fx(rs[i]), fx(c)) for i in idx])
    for i, (x, a1, a2) in zip(idx, outs):
        s = slice(i * EB, (i + 1) * EB)
        assert res["X"][s] == x, f"{what}: X of share {i}"
        assert res["a1"][s] == a1, f"{what}: a1 of share {i} (X = {int.from_bytes(x, 'big'):#x}, r = {rs[i]:#x})"
        assert res["a2"][s] == a2, f"{what}: a2 of share {i} (y = {ys[i]:#x}, Y = {Ys[i]:#x}, r = {rs[i]:#x})"
    return res


def tables():
    assert int(os.environ.get("MPVSS_PAIR", "0")) & 15 == 15
    from mpvss_rs_amd import Engine
    eng = Engine(0)
    rng = random.Random(0xED6E)
    n = 8192 + 37                      # wide comb (n >= 8192), the last workgroup ragged: 8229 = 128 * 64 + 37
    pos = list(range(3, 3 + n))
    ys, Ys, rs = cyc(EDGE, n, rng), cyc(EDGE[::-1], n, rng, 3), cyc(EDGE_R, n, rng)
    ys[-1], Ys[-1], rs[-1] = Q - 1, (1 << 2048) - 1, (1 << 2048) - 1          # the share the padding lanes repeat
    # X_i in {1, Q-1}: C_0 = 1, C_1 = Q-1; then random commitments (t = 5: Horner, no forward differences)
    c = rng.randrange(1 << 256)
    check_box(eng, [1, Q - 1], pos, ys, Ys, rs, c, "X = +-1, shared 256-bit c")
    part = [i for i in range(n) if i % 4 == 0 or i >= n - 70]           # (the oracle's time: a quarter of the shares + the ragged end)
    m = 4133                           # (the wide comb of g^r was exercised above; these two differ in the schedule of c only)
    check_box(eng, [1, Q - 1], pos[:m], ys[:m], Ys[:m], rs[:m], 0, "X = +-1, c = 0", [i for i in part if i < m] + list(range(m - 70, m)))
    check_box(eng, [1, Q - 1], pos[:m], ys[:m], Ys[:m], rs[:m], (1 << 256) - 1, "X = +-1, c = 2^256 - 1", [i for i in part if i < m] + list(range(m - 70, m)))
    cm = [pow(4, rng.randrange(Q - 1), Q) for _ in range(5)]
    check_box(eng, cm, pos, ys, Ys, rs, c | 1, "random X", part)
    check_box(eng, [0, 7], pos[:4200], ys[:4200], Ys[:4200], rs[:4200], c, "X = 0")
    check_box(eng, [Q + 1, Q, 1], pos[:4133], ys[:4133], Ys[:4133], rs[:4133], rng.randrange(Q - 1), "full-width shared c")
    eng.close()
    print("pair tables ok")


def fd():
    assert os.environ.get("MPVSS_FD_PAIR_MIN_T") == "16"
    from mpvss_rs_amd import Engine
    eng = Engine(0)
    rng = random.Random(0xFD51)
    t, n, p0 = 512, 9000, 40000
    pos = list(range(p0, p0 + n))
    spread = sorted(set([0, 1, 511, 512, n - 1] + [rng.randrange(n) for _ in range(1)]))
    # random commitments g^a from the engine's fixed-base comb (oracle-checked elsewhere): 1500 Python modpows otherwise
    raw = eng.batch_exp_fixed_base(fx(4), b"".join(fx(rng.randrange(Q - 1)) for _ in range(3 * t)))
    rnd = [int.from_bytes(raw[k * EB:(k + 1) * EB], "big") for k in range(3 * t)]
    cases = {
        "+-1 only": [1 if rng.random() < 0.5 else Q - 1 for _ in range(t)],
        "random with 1, Q-1, Q+1": rnd[:t],
        "a zero commitment": rnd[t:2 * t],
        "C_0 = Q": rnd[2 * t:],
    }
    cases["random with 1, Q-1, Q+1"][3] = 1
    cases["random with 1, Q-1, Q+1"][7] = Q - 1
    cases["random with 1, Q-1, Q+1"][100] = Q + 1
    cases["random with 1, Q-1, Q+1"][511] = Q - 1
    cases["a zero commitment"][200] = 0
    cases["C_0 = Q"][0] = Q
    ones = fx(1) * n
    for what, cm in cases.items():
        before = eng.fd_stats()
        # (the block path, whose absorb step reports whether the forward differences held; y = Y = 1, r = c = 0)
        out = eng.verify_distribution(b"".join(map(fx, cm)), pos, ones, ones, bytes(n * EB), bytes(EB), dump=True)["X"]
        after = eng.fd_stats()
        got = [int.from_bytes(out[i * EB:(i + 1) * EB], "big") for i in spread]
        want = parallel_map(modp_reference_x, [([v % Q for v in cm], pos[i]) for i in spread])    # reference order
        assert got == want, what
        # the fast form (Horner in the exponent) on a spread 2 % of the box
        more = sorted(rng.sample(range(n), 72))
        cmb = b"".join(map(fx, cm))
        fast = parallel_map(modp_fast_share, [(cmb, pos[i], fx(1), fx(1), fx(0), fx(0)) for i in more])
        for i, (x, _, _) in zip(more, fast):
            assert out[i * EB:(i + 1) * EB] == x, (what, i)
        print(what, "fd calls", after[0] - before[0], "fallbacks", after[1] - before[1])
        assert after[0] - before[0] == 1, "the box was expected to take the forward-difference path"
        if "zero" in what or what.endswith("Q"):
            assert after[1] - before[1] == 1, "X = 0 has no inverse: the device flag must send the box down Horner's rule"
        else:
            assert after[1] == before[1], "forward differences were expected to hold"
    eng.close()
    print("pair fd ok")


if __name__ == "__main__":
    import torch  # noqa: F401  (torch's HIP runtime first, as tests/conftest.py)
    {"tables": tables, "fd": fd}[sys.argv[1]]()
