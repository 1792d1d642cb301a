"""The pair-layout Montgomery product (mpvss_rs_amd/csrc/bn_pair.h) moves the reduction onto the matrix cores: two int8 GEMMs
against constant digit matrices of N' and N.  Its arithmetic -- signed limb-aligned digits, C-init corrections, the carry-free
re-digitisation of m, the guard limb, the bias of the high columns -- is proven here on exact Python integers by the model the
constant tables are generated from (tools/mfma_mont/model.py); the committed header must be what the model emits.  The kernel
itself is compared with the oracle by the GPU tests (every a2 of a lone call with >= 4096 shares goes through it)."""
import os
import random
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools", "mfma_mont"))
import model as M  # noqa: E402


def test_reduction_identities_and_bounds_on_edge_values():
    rng = random.Random(2029)
    N = M.N
    cases = [(0, 0), (1, 1), (2 * N - 1, 2 * N - 1), (N - 1, N + 1), (2 * N - 1, 1), (rng.randrange(2 * N), rng.randrange(2 * N))]
    for a, b in cases:
        r = M.mont_model(a, b)            # asserts every intermediate bound and the exact carry of the low columns
        assert r < 2 * N and (r * M.R - a * b) % N == 0


def test_squaring_column_bound_of_the_pair_layout():
    """phase A of a squaring doubles the limb above the diagonal: between two carries of a column (36 rows) the 64-bit
    accumulator collects at most 19 doubled products of almost-normalised limbs -- worst case below 2^64."""
    LP, lim = 36, (1 << 29) - 1 + (1 << 9)
    worst = 0
    for i0 in range(72):                      # a column enters a lane's window at row i0 (local position 35) and leaves 35 rows later
        acc = 0
        for s in range(LP):
            i, k = i0 + s, LP - 1 - s
            if i >= 72:
                break
            rr = i % LP
            if k >= rr:
                acc += lim * lim * (2 if k > rr else 1)
        worst = max(worst, acc)
    assert worst + (1 << 40) < (1 << 64)     # plus the carry handed up from the column below


def test_committed_tables_are_what_the_model_emits(tmp_path):
    out = tmp_path / "tables.h"
    M.emit(str(out))
    committed = open(os.path.join(ROOT, "mpvss_rs_amd", "csrc", "modp_mfma_tables.h")).read()
    assert out.read_text() == committed
