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


def test_the_mix_weighted_budget_agrees_with_the_flat_slot_count_when_every_class_costs_a_slot():
    """`compute.peak_mix_weighted` prices every instruction class with its own measured rate; with all classes at one slot (4 cycles at
    2.4 GHz) it must give back the flat issue-slot accounting of the same operations to within the few instructions the two counts
    differ by (table addressing in the shipped kernel), and with the rates round 4 measured (a v_mad_u64_u32 4.38 cycles at 1.9 GHz,
    the 64-bit shifts two passes) the a2 launch needs more time than its slot count says."""
    bench = load_bench()
    slot = 4.0 / 2.4e9
    flat = {c: slot for c in bench.MIX_CLASSES}
    wk = bench.modp_work(65536, 256, list(range(1, 65537)), [(1 << 255) | 12345])
    by_slots = wk["slots"] * slot
    by_mix = bench.mix_seconds_per_simd(wk["by_layout"], flat)
    assert abs(by_mix / by_slots - 1) < 0.04, (by_mix, by_slots)
    assert sum(wk["by_layout"].values()) == pytest_approx(65536 * (wk["ops"]["squarings"] + wk["ops"]["products"]))
    measured = dict(flat, mad64=4.38 / 1.9e9, shift64=2 * 4.0 / 2.35e9, swap=6.0 / 2.35e9)
    a2_flat = bench.mix_seconds_per_simd(wk["a2_by_layout"], flat)
    assert bench.mix_seconds_per_simd(wk["a2_by_layout"], measured) > 1.2 * a2_flat
    assert abs(a2_flat / (wk["a2_slots"] * slot) - 1) < 0.04


def pytest_approx(x):
    import pytest
    return pytest.approx(x, rel=1e-9)
