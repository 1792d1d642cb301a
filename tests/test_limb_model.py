"""Integer model of the kernels' quad-lane radix-2^29 Montgomery product (mpvss_rs_amd/csrc/bn_quad.h):
same step order, same lazy carries, same two-pass normalisation.  Checks (a) the result, (b) that no
64-bit column accumulator can overflow even for worst-case "almost normalised" limbs, (c) the limb
bound the next product relies on."""
import random

import mpvss_oracle as O

N = O.ModpGroup().q
W, L, LPL = 29, 72, 18
M = (1 << W) - 1
R = 1 << (W * L)
NL = [(N >> (W * j)) & M for j in range(L)]
N0INV = (-pow(N, -1, 1 << W)) % (1 << W)
LIMB_BOUND = M + 512


def mont_mul_model(a, b, stats, bound_only=False, square=False):
    """square=True: the dedicated squaring (b is a): row r = 18 o + rr visits only the local positions k >= rr of
    every lane, k > rr with the doubled limb."""
    T = [0] * L
    for i in range(L):
        bi = b[i]
        rr = i % LPL
        for j in range(L):
            k = j % LPL
            if not square:
                T[j] += a[j] * bi
            elif k >= rr:
                T[j] += a[j] * (2 * bi if k > rr else bi)
        m = ((T[0] & 0xFFFFFFFF) * N0INV) & M
        for j in range(L):
            T[j] += m * NL[j]
        assert T[0] & M == 0
        stats["maxacc"] = max(stats["maxacc"], max(T))
        for q in range(4):       # every lane: upper bits of its lowest column into its next column, low limb handed down
            j = q * LPL
            T[j + 1] += T[j] >> W
            T[j] &= M
        stats["maxacc"] = max(stats["maxacc"], max(T))
        assert max(T) < (1 << 64)
        T = T[1:] + [0]
    if bound_only:
        return None
    # pass 1 inside each lane, pass 2 across lanes
    limbs, couts = [0] * L, [0] * 4
    for q in range(4):
        c = 0
        for k in range(LPL):
            v = T[q * LPL + k] + c
            assert v < (1 << 64)
            limbs[q * LPL + k] = v & M
            c = v >> W
        couts[q] = c
    assert couts[3] == 0
    for q in range(1, 4):
        v = limbs[q * LPL] + couts[q - 1]
        limbs[q * LPL] = v & M
        limbs[q * LPL + 1] += v >> W
    return limbs


def val(l):
    return sum(x << (W * j) for j, x in enumerate(l))


def tolimbs(v):
    return [(v >> (W * j)) & M for j in range(L)]


def test_model_matches_montgomery_product_and_bounds():
    assert N0INV == 1
    rng = random.Random(5)
    rinv = pow(R, -1, N)
    stats = {"maxacc": 0}
    for it in range(60):
        a, b = rng.randrange(2 * N), rng.randrange(2 * N)
        if it < 3:
            a = b = 2 * N - 1
        r = mont_mul_model(tolimbs(a), tolimbs(b), stats)
        v = val(r)
        assert v % N == (a * b * rinv) % N and v < 2 * N
        assert max(r) <= LIMB_BOUND
    assert stats["maxacc"] < (1 << 64)


def test_dedicated_squaring_matches_and_stays_in_bounds():
    rng = random.Random(6)
    rinv = pow(R, -1, N)
    stats = {"maxacc": 0}
    for it in range(40):
        a = rng.randrange(2 * N)
        if it < 3:
            a = 2 * N - 1
        r = mont_mul_model(tolimbs(a), tolimbs(a), stats, square=True)
        v = val(r)
        assert v % N == (a * a * rinv) % N and v < 2 * N
        assert max(r) <= LIMB_BOUND
    mont_mul_model([LIMB_BOUND] * L, [LIMB_BOUND] * L, stats, bound_only=True, square=True)
    assert stats["maxacc"].bit_length() <= 64


def test_worst_case_limbs_do_not_overflow():
    stats = {"maxacc": 0}
    mont_mul_model([LIMB_BOUND] * L, [LIMB_BOUND] * L, stats, bound_only=True)     # far above 2N as an integer: bound check only
    assert stats["maxacc"].bit_length() <= 64


# ---- the row layout (mpvss_rs_amd/csrc/bn_row.h): 16 lanes x 5 limb slots, the same 72 rows ----------------------------------
def row_mont_mul_model(a, b, stats, square=False):
    """bn_row.h::mont_mul as integers: 80 column slots (72..79 hold zeros on input), 72 rows, every lane carries its lowest column
    every row and hands its low 29 bits to the lane below; two final carry passes."""
    LPLR, LANES, COLS = 5, 16, 80
    a = list(a) + [0] * (COLS - L)
    nl = NL + [0] * (COLS - L)
    T = [0] * COLS
    for i in range(L):
        bi = b[i]
        rr = i % LPLR
        for j in range(COLS):
            k = j % LPLR
            if not square:
                T[j] += a[j] * bi
            elif k >= rr:
                T[j] += a[j] * (2 * bi if k > rr else bi)
        m = ((T[0] & 0xFFFFFFFF) * N0INV) & M
        for j in range(COLS):
            T[j] += m * nl[j]
        assert T[0] & M == 0
        stats["maxacc"] = max(stats["maxacc"], max(T))
        for q in range(LANES):
            j = q * LPLR
            T[j + 1] += T[j] >> W
            T[j] &= M
        stats["maxacc"] = max(stats["maxacc"], max(T))
        assert max(T) < (1 << 64)
        T = T[1:] + [0]
    limbs, couts = [0] * COLS, [0] * LANES
    for q in range(LANES):
        c = 0
        for k in range(LPLR):
            v = T[q * LPLR + k] + c
            limbs[q * LPLR + k] = v & M
            c = v >> W
        couts[q] = c
        assert c < (1 << 36)
    assert couts[LANES - 1] == 0
    for q in range(1, LANES):
        v = limbs[q * LPLR] + couts[q - 1]
        limbs[q * LPLR] = v & M
        limbs[q * LPLR + 1] += v >> W
    assert all(x == 0 for x in limbs[L:])
    return limbs[:L]


def test_row_layout_model():
    """Same results as the quad model's pipeline for ordinary and worst-case operands, accumulators far from 2^64, limbs back
    inside the bound the next product relies on -- for the product and for the dedicated squaring."""
    rng = random.Random(6)
    rinv = pow(R, -1, N)
    stats = {"maxacc": 0}
    worst = [LIMB_BOUND] * 70 + [0, 0]
    while val(worst) >= 2 * N:
        worst[69] //= 2
    cases = [(tolimbs(2 * N - 1), tolimbs(2 * N - 1)), (worst, worst), (tolimbs(0), worst), (tolimbs(1), tolimbs(N))]
    for _ in range(40):
        cases.append((tolimbs(rng.randrange(2 * N)), tolimbs(rng.randrange(2 * N))))
    for a, b in cases:
        for sq in (False, True):
            bb = a if sq else b
            r = row_mont_mul_model(a, bb, stats, square=sq)
            v = val(r)
            assert v < 2 * N and v % N == val(a) * val(bb) * rinv % N
            assert max(r) <= LIMB_BOUND and r[71] == 0
    assert stats["maxacc"] < (1 << 62)
