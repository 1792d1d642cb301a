"""bench.py's work model (modp_work: Montgomery operations and VALU issue slots per verified share, what `compute.frac` is
made of) against its own invariants: the layout a kernel runs in changes the slots, never the operation count; the
stepping products of the X path move to the pair layout from MPVSS_FD_PAIR_MIN_T commitments only."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_bench():
    sys.path.insert(0, ROOT)
    try:
        return importlib.import_module("bench")
    finally:
        sys.path.pop(0)


def test_slots_follow_the_layout_and_operations_do_not():
    bench = load_bench()
    cs = [0x1234567890ABCDEF << 190 | 12345, (1 << 255) | 1]
    saved = bench.PAIR_MASK
    try:
        for n, t in ((65536, 256), (131072, 1024), (4096, 64)):
            pos = list(range(1, n + 1))
            res = {}
            for mask in (0, 1, 17, 49, 63):
                bench.PAIR_MASK = mask
                res[mask] = bench.modp_work(n, t, pos, cs)
            assert len({round(r["mm_total"], 6) for r in res.values()}) == 1          # the work is the work
            assert len({(round(r["ops"]["squarings"], 6), round(r["ops"]["products"], 6)) for r in res.values()}) == 1
            assert res[1]["slots"] < res[0]["slots"] and res[17]["slots"] == res[1]["slots"]      # bit 4 is the dealer's kernel
            if t >= bench.FD_PAIR_MIN_T:
                assert res[49]["slots"] < res[17]["slots"]
                step = res[17]["slots"] - res[49]["slots"]
                assert abs(step / (bench.QUAD_MUL_SLOTS - bench.PAIR_MUL_SLOTS) / n - t) < 0.25 * t     # about t stepping products per share
            else:
                assert res[49]["slots"] == res[17]["slots"]
            assert res[63]["slots"] < res[49]["slots"]
            # ~2 700 product equivalents per share at the headline shape, more with more commitments
            per_share = res[49]["mm_total"] / n
            assert 2500 < per_share < 4500, per_share
    finally:
        bench.PAIR_MASK = saved
