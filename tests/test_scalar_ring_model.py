"""Integer model of the device's scalar ring Z/(q-1) (k_modq_poly_eval / k_modq_responses in
mpvss_rs_amd/csrc/modp_kernels.hip; src/polynomial.rs:50-58, src/dleq.rs:42-50): residues mod q' = (q-1)/2 in the
Montgomery product of the group kernels -- Horner's rule with an 18-row product that leaves a factor 2^-522 per step,
cancelled in the coefficients -- and the parity beside, lifted by the Chinese remainder.  Checks the generated constants
of q', the host-side coefficient transform and the identities the kernels rely on, on Python integers."""
import os
import random
import re

import mpvss_oracle as O

Q = O.ModpGroup().q
ORDER = Q - 1
QH = ORDER // 2
W, L = 29, 72
R = 1 << (W * L)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_array(name):
    txt = open(os.path.join(ROOT, "mpvss_rs_amd", "csrc", "modp2048_consts.h")).read()
    body = re.search(r"%s\[72\] = \{(.*?)\};" % name, txt, re.S).group(1)
    limbs = [int(x.rstrip("u"), 16) for x in re.findall(r"0x[0-9a-fA-F]+u?", body)]
    assert len(limbs) == L and all(x < (1 << W) for x in limbs)
    return sum(x << (W * j) for j, x in enumerate(limbs))


def test_constants_of_the_half_order():
    assert QH % 2 == 1 and 2 * QH == ORDER
    assert header_array("MODQH_N_LIMBS") == QH
    assert header_array("MODQH_R2_LIMBS") == (R * R) % QH
    assert header_array("MODQH_ONE_M_LIMBS") == R % QH
    # the kernels use ONE n0inv for both moduli: q and q' are both -1 mod 2^29
    assert (-pow(QH, -1, 1 << W)) % (1 << W) == (-pow(Q, -1, 1 << W)) % (1 << W) == 1


def rows18(a, b):
    """mont_mul<.., OUTER = 1>: 18 rows, b < 2^522 -> a b 2^-522 mod q' (any representative)"""
    assert b < (1 << (18 * W))
    return a * b * pow(1 << (18 * W), -1, QH) % QH


def lift(v, parity):
    v %= QH
    return v if v % 2 == parity else v + QH


def test_horner_with_scaled_coefficients_and_the_parity_lift():
    rng = random.Random(3)
    for t in (1, 2, 5, 33):
        coeffs = [rng.randrange(ORDER) for _ in range(t)]
        coeffs[rng.randrange(t)] = rng.choice([0, 1, ORDER - 1, QH, QH + 1])
        scaled = [a % ORDER % QH * pow(1 << (18 * W), j, QH) % QH for j, a in enumerate(coeffs)]     # the host's a'_j
        par_even, par_odd = coeffs[0] % 2, sum(a % 2 for a in coeffs) % 2
        for x in [0, 1, 2, 3, (1 << 29) - 1, 1 << 29, (1 << 63) - 1] + [rng.randrange(1 << 40) for _ in range(5)]:
            acc = scaled[t - 1]
            for j in range(t - 2, -1, -1):
                acc = rows18(acc, x) + scaled[j]          # limb-wise add, values stay far below 2^2088
                assert acc < 3 * QH
            want = sum(a * pow(x, j, ORDER) for j, a in enumerate(coeffs)) % ORDER
            assert lift(acc, par_odd if x & 1 else par_even) == want


def test_responses_identity():
    rng = random.Random(4)
    for _ in range(50):
        w, a = rng.randrange(1 << 2048), rng.randrange(ORDER)
        c = rng.choice([rng.randrange(1 << 256), 0, 1, QH, ORDER - 1, rng.randrange(1 << 2048)])
        cneg = (QH - c % ORDER % QH) % QH
        s = a * cneg % QH + w                      # alpha * (-c) mod q' plus w as limbs: below 2^2050
        assert s < (1 << 2050)
        parity = (w ^ (a & (c % ORDER))) & 1       # -alpha c = alpha c (mod 2)
        assert lift(s, parity) == (w - a * c) % ORDER
