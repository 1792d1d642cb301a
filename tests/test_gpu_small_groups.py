"""Runs of same-shaped SMALL boxes -- the reference's own sizes, far below the forward differences' 4096 shares -- travel through
mpvss_modp_verify_many as ONE block per run (round 6: X by Horner with the box as the second grid dimension; a short run on an otherwise
idle context takes the row-layout a2).  Same verdicts and digests as one mpvss_modp_verify_distribution call per box
(src/participant.rs:399-455), whatever the mixture of shapes."""
import random

import pytest

from helpers import EB

pytestmark = pytest.mark.gpu
fx = lambda v: v.to_bytes(EB, "big")


def test_runs_of_small_boxes_as_groups_equal_the_one_box_calls(engine):
    rng = random.Random(0x5A11)
    sc = lambda k: b"".join(fx(rng.randrange(1, 1 << 2040)) for _ in range(k))
    shapes = [(5, 3), (1, 1), (10, 10), (64, 17), (300, 2), (1030, 33)]
    made = {}
    for n, t in shapes:
        pos = [rng.randrange(1, 1 << 30) for _ in range(n)] if n == 10 else list(range(2, 2 + n))      # (10, 10): scattered positions
        pk = engine.batch_exp_fixed_base(fx(2), sc(n))
        boxes = []
        for _ in range(3):
            co, wi = sc(t), sc(n)
            d = engine.deal(co, pos, pk, wi)
            boxes.append(dict(commitments=engine.batch_exp_fixed_base(fx(4), co), positions=pos, pubkeys=pk, shares=d["Y"],
                              responses=d["responses"], challenge=d["challenge"], digest=d["digest"]))
        made[(n, t)] = boxes
    flip = lambda b, at: b[:at] + bytes([b[at] ^ 1]) + b[at + 1:]
    seq = []
    for n, t in shapes:                                   # a run of 7 per shape, the third tampered (a response bit), the fifth too (a share bit)
        run = [made[(n, t)][k % 3] for k in range(7)]
        run[2] = dict(run[2], responses=flip(run[2]["responses"], (n - 1) * EB + 255))
        run[4] = dict(run[4], shares=flip(run[4]["shares"], 200))
        seq += run
    seq.insert(9, made[(64, 17)][0])                      # a lone box of another shape inside the (1, 1) run: two shorter runs around it
    seq.append(dict(made[(5, 3)][1], challenge=fx((1 << 270) + 3)))      # a challenge beyond 256 bits: not groupable, verdict 0
    one_by_one = [engine.verify_distribution(b["commitments"], b["positions"], b["pubkeys"], b["shares"], b["responses"], b["challenge"])
                  for b in seq]
    want = [(r["verdict"], r["digest"]) for r in one_by_one]
    assert [v for v, _ in want].count(False) == 2 * len(shapes) + 1
    for depth, threads in ((10, 8), (1, 1), (3, 2)):
        assert engine.verify_many(seq, depth=depth, hash_threads=threads) == want, (depth, threads)
    honest = [b for b, (v, _) in zip(seq, want) if v]
    got = engine.verify_many(honest, depth=6, hash_threads=4)
    assert all(v and dg == b["digest"] for (v, dg), b in zip(got, honest))
    assert engine.blocks_in_flight() == (0, 0)
