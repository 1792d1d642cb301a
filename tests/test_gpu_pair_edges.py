"""Edge-operand parity of the PAIR-layout kernels (Montgomery reduction on the matrix cores, mpvss_rs_amd/csrc/bn_pair.h,
modp_pair_kernels.hip) on the device, at the batch sizes that actually select them.

The reference validates nothing on this path -- `ModpGroup::exp` is `modpow`, which reduces whatever it is given
(src/groups/modp.rs:122-132,154-156), and the verifier's a1 = g^r X^c, a2 = y^r Y^c take dealer-controlled y, Y, r, c
(src/dleq.rs:66-84) -- so hostile operands (0, 1, q-1, q, q+1, 2^2048-1, exponents 0 / >= q-1 / all ones) must come out
exactly as Python's pow gives them, from the kernel that produces the headline and not only from the small-batch VALU
kernels (tests/test_gpu_modp.py::EDGE runs below 1024 shares, where the quad kernels are selected)."""
import os
import random
import struct
import subprocess
import sys

import pytest

from helpers import EB, MODP_Q as Q, modp_dual_pow_chunk, parallel_map

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools", "mfma_mont"))

EDGE = [0, 1, 2, 4, Q - 1, Q, Q + 1, (1 << 2048) - 1, 1 << 2047, (1 << 2040) - 1, 0xFFFFFFF, 1 << 28, (1 << 56) - 1]
EDGE_R = [0, 1, Q - 2, Q - 1, Q, (1 << 2048) - 1, (1 << 2046) - 1, 63 << 2040, 1 << 6, (1 << 2048) - (1 << 2042)]


def fx(v):
    return v.to_bytes(EB, "big")


# ---- (a) one Montgomery operation at the bounds of the integer model --------------------------------------------------
def test_mont_pair_unit_operations_at_the_model_bounds(tmp_path):
    """tests/pair_unit.hip runs ONE product, ONE squaring and ONE product-with-the-neighbour's-operand (the stepping form of
    phase A) per operand pair through bn_pair.h; every result must be a * b * R^-1 mod N as an almost-normalised value below
    2N (limbs <= 2^29 - 1 + 2^9), and on the named edge pairs exactly the integer the model computes."""
    import model as M
    N, R, W, L = M.N, M.R, M.W, M.L
    M29 = (1 << W) - 1
    LIM = M29 + (1 << 9)
    exe = os.path.join(ROOT, "tests", "_build", "pair_unit")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "mpvss_rs_amd", "csrc"), "examples", "-s"])
    rng = random.Random(0x9A12)

    def canon(v):
        return [(v >> (W * k)) & M29 for k in range(L)]

    def value(limbs):
        return sum(x << (W * k) for k, x in enumerate(limbs))

    # every limb at the almost-normalised maximum: limbs 0..69 = 2^29 - 1 + 2^9, limb 70 as large as 2N allows
    maxl = [LIM] * 70 + [0, 0]
    maxl[70] = (2 * N - 1 - value(maxl)) >> (W * 70)
    assert 0 < maxl[70] < (1 << W) and value(maxl) < 2 * N

    def loose(v=None):
        """a random almost-normalised representation: limbs uniform up to the bound, the value still below 2N"""
        l = [rng.randrange(LIM + 1) for _ in range(70)] + [0, 0]
        l[70] = rng.randrange(((2 * N - 1 - value(l)) >> (W * 70)) + 1)
        return l

    half = 1 << 1044
    named = [canon(v) for v in (0, 1, 2, N - 1, N, N + 1, 2 * N - 1, 2 * N - 2, half, half - 1, half + 1, 3 * half,
                                (1 << 2048) - 1, 1 << 2048, M29, 1 << W, (1 << (W * 36)) - 1, 1 << (W * 36))]
    named.append(maxl)
    pairs = [(a, b) for a in named for b in named]                       # includes T_lo = 0 (half * half = R) and
    edge_count = len(pairs)                                              # T_lo = R - 1 ((half - 1) * (half + 1))
    assert value(canon(half)) ** 2 % R == 0 and (half - 1) * (half + 1) % R == R - 1
    for _ in range(2000):
        pairs.append((canon(rng.randrange(2 * N)), canon(rng.randrange(2 * N))))
    for _ in range(2000):
        pairs.append((loose(), loose()))
    n = len(pairs)
    assert n % 64 != 0                                                   # the last workgroup is ragged
    inp, outp = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(inp, "wb") as f:
        for a, b in pairs:
            f.write(struct.pack("<72I", *a) + struct.pack("<72I", *b))
    res = subprocess.run([exe, str(inp), str(outp), str(n)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    raw = open(outp, "rb").read()
    assert len(raw) == n * 3 * 288
    Rinv = pow(R, -1, N)
    vals = [(value(a), value(b)) for a, b in pairs]
    for i, (a, b) in enumerate(vals):
        wave_last = (i % 32 == 31) or i == n - 1
        nb = b if wave_last else vals[i + 1][1]                          # operand of the stepping form: the NEXT number's b
        for which, (x, y) in enumerate(((a, b), (a, a), (a, nb))):
            limbs = struct.unpack_from("<72I", raw, (3 * i + which) * 288)
            v = value(limbs)
            assert max(limbs) <= LIM and limbs[71] == 0, (i, which, hex(max(limbs)))
            assert v < 2 * N and v % N == x * y * Rinv % N, (i, which)
            if i < edge_count and which < 2 and (i % 4 == 0 or x * y % R in (0, R - 1)):
                assert v == M.mont_model(x, y, check=False), (i, which)  # the exact integer of the model's pipeline


# ---- (b) the shipped a2 kernel on hostile y, Y, r, c ------------------------------------------------------------------
def _expected(b1, e1, b2, e2):
    items = list(zip(b1, e1, b2, e2))
    step = max(1, len(items) // 64)
    chunks = [items[k:k + step] for k in range(0, len(items), step)]
    return [v for part in parallel_map(modp_dual_pow_chunk, chunks) for v in part]


def _split(b):
    return [int.from_bytes(b[i:i + EB], "big") for i in range(0, len(b), EB)]


def _cyc(vals, n, rng, every=2):
    return [vals[(i // every) % len(vals)] if i % every == 0 else rng.randrange(1 << 2048) for i in range(n)]


def test_a2_pair_kernel_on_hostile_operands(engine):
    """mpvss_modp_dleq_commitments with n >= 1024 per-share bases runs a2 = g2^r h2^c through k_modp_dual_exp_w6_pair (and the
    64-entry tables of g2 it reads): bases cycling through EDGE, exponents through EDGE_R, n not a multiple of 64 with the
    padding lanes repeating a hostile last share; a shared 256-bit c, c = 0, c = 2^256 - 1, a 256-bit c per share (the c_stride
    path), and full-width per-share c (which takes the 4-bit-window kernel: same results).  Every a2 and a1 against pow."""
    rng = random.Random(0xA2ED6E)
    n = 4096 + 37
    y, Y, r = _cyc(EDGE, n, rng), _cyc(EDGE[::-1], n, rng, 3), _cyc(EDGE_R, n, rng)
    X = _cyc(EDGE, n, rng, 5)
    y[-1], Y[-1], r[-1] = (1 << 2048) - 1, Q - 1, (1 << 2048) - 1
    cat = lambda v: b"".join(map(fx, v))
    by, bY, br, bX = cat(y), cat(Y), cat(r), cat(X)
    for c in (rng.randrange(1 << 256), 0, (1 << 256) - 1):
        a1, a2 = engine.dleq_commitments(fx(4), bX, by, bY, br, fx(c), False)
        assert _split(a2) == _expected(y, r, Y, [c] * n), f"a2, shared c = {c:#x}"
        assert _split(a1) == _expected([4] * n, r, X, [c] * n), f"a1, shared c = {c:#x}"
    m = 1024 + 37
    cs = [rng.randrange(1 << 256) for _ in range(m)]
    cs[0], cs[1], cs[2], cs[-1] = 0, 1, (1 << 256) - 1, (1 << 256) - 1
    a1, a2 = engine.dleq_commitments(fx(2), bX[:m * EB], by[:m * EB], bY[:m * EB], br[:m * EB], cat(cs), True)
    assert _split(a2) == _expected(y[:m], r[:m], Y[:m], cs), "a2, one 256-bit c per share"
    assert _split(a1) == _expected([2] * m, r[:m], X[:m], cs), "a1, one 256-bit c per share"
    cw = _cyc(EDGE_R, m, rng)
    a1, a2 = engine.dleq_commitments(fx(4), bX[:m * EB], by[:m * EB], bY[:m * EB], br[:m * EB], cat(cw), True)
    assert _split(a2) == _expected(y[:m], r[:m], Y[:m], cw), "a2, full-width c per share"
    assert _split(a1) == _expected([4] * m, r[:m], X[:m], cw), "a1, full-width c per share"


# ---- (c), (d) the other pair kernels, selected by environment switches that are read once per process -----------------
def _child(mode, env):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "pair_edge_child.py"), mode], capture_output=True, text=True,
                         env=dict(os.environ, **env), timeout=1200)
    assert out.returncode == 0 and f"pair {mode} ok" in out.stdout, out.stdout[-1500:] + out.stderr[-3000:]


def test_table_comb_and_a1_pair_kernels_on_hostile_operands():
    """MPVSS_PAIR=63: k_modp_build_table_pair (64 entries of y, odd powers of Y and X), k_modp_comb16_exp_pair (g^r),
    k_modp_sched_exp_mul_pair (a1 = g^r X^c along a shared sliding-window schedule) and the a2 kernel on boxes of 8229
    shares with X in {1, q-1, 0, random}, y / Y / r cycling through the edge sets, c in {random 256-bit, 0, 2^256-1, full
    width}: every X, a1, a2 against pow."""
    _child("tables", {"MPVSS_PAIR": "63"})


def test_fd_step_pair_kernel_with_edge_commitments():
    """k_modp_fd_step_pair at t = 512 (MPVSS_FD_PAIR_MIN_T=16 also sends the stride-1 seeding chain through it): commitments
    in {1, q-1} only, random ones with 1 / q-1 / q+1 among them, and boxes whose X is 0 (a zero or unreduced-q commitment:
    no inverse, the device flag must send them down Horner's rule) -- against O.commitment_eval in the reference order on 6
    positions per box (24 in all) and the fast form on 72 more each."""
    _child("fd", {"MPVSS_FD_PAIR_MIN_T": "16", "MPVSS_FD_L1": "2"})


def test_fd_step_pair_tile_kernel_with_edge_commitments():
    """The same boxes through k_modp_fd_step_pair_tile (round 5: the stepping as wide launches over the anti-diagonals of the
    (stage, block of steps) grid, MPVSS_FD_TILE=2; ragged blocks of 50 steps): same oracle positions, same fall-back for the
    boxes whose X is 0."""
    _child("fd", {"MPVSS_FD_PAIR_MIN_T": "16", "MPVSS_FD_L1": "2", "MPVSS_FD_TILE": "2", "MPVSS_FD_TILE_STEPS": "50"})
