"""Forward differences in the exponent (consecutive positions) must give exactly Horner's results."""
import os
import subprocess
import sys
import time

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (t, n, first position, statement that doctors the commitments)
CASES = [(16, 4096, 1, ""), (64, 8192, 1, ""), (33, 5000, 777, ""), (256, 8192, 1, ""), (33, 9001, 777, ""),
         (17, 20011, 123456789, ""), (100, 12345, 2, "cm[99] = Q - 1"), (1024, 16384, 1, ""), (512, 9000, 40000, ""),
         (64, 2048, 1, ""), (16, 2100, 7, ""), (257, 5000, 3, ""), (64, 8192, 1, "cm[5] = 0"), (64, 8192, 1, "cm[0] = Q")]

# one process evaluates every case (importing torch and creating the context once) and prints one hash per case
CODE = r'''
import os, sys, random, hashlib
sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)
import mpvss_oracle as O
from helpers import modp_fast_share
from mpvss_rs_amd import Engine
G = O.ModpGroup(); Q = G.q
fx = lambda v: v.to_bytes(256, "big")
eng = Engine(0)
for t, n, p0, extra in %r:
    rng = random.Random(t * 1000 + n)
    # commitments g^a_j from the engine's fixed-base comb (oracle-checked elsewhere): seconds of Python pow per process otherwise
    raw = eng.batch_exp_fixed_base(fx(4), b"".join(fx(rng.randrange(Q - 1)) for _ in range(t)))
    cm = [int.from_bytes(raw[k * 256:(k + 1) * 256], "big") for k in range(t)]
    exec(extra)
    pos = list(range(p0, p0 + n))
    cmb = b"".join(map(fx, cm))
    out = eng.commit_eval(cmb, pos)
    # spot-check against the oracle (every position is compared with Horner's result through the hash; the oracle pins a few of
    # them: the reference-order loop for small t, the fast form of the same arithmetic -- tests/helpers.py -- for large t)
    for i in (((0, t, n - 1) if t < 100 else (1, n - 1)) if os.environ.get("CHECK_ORACLE") else ()):
        got = out[i * 256:(i + 1) * 256]
        if t < 100:
            assert int.from_bytes(got, "big") == O.commitment_eval(G, cm, pos[i]), (t, n, i)
        else:
            assert got == modp_fast_share((cmb, pos[i], fx(1), fx(1), fx(0), fx(0)))[0], (t, n, i)
    print(hashlib.sha256(out).hexdigest())
'''


def run(cases, env_extra):
    code = CODE % (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), cases)
    env = dict(os.environ, **env_extra)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=1800)
    assert out.returncode == 0, out.stderr[-3000:]
    return out.stdout.split()


def test_fd_equals_horner():
    # MPVSS_FD_L1=2: two-level seeding (what pipelined boxes use); 0: Horner's rule for every seed (what a lone call uses)
    a = run(CASES, {"MPVSS_FD": "1", "MPVSS_FD_L1": "2", "MPVSS_FD_MIN_SHARES": "2048", "CHECK_ORACLE": "1"})
    b = run(CASES, {"MPVSS_FD": "0"})
    c = run(CASES, {"MPVSS_FD": "1", "MPVSS_FD_L1": "0", "MPVSS_FD_MIN_SHARES": "2048"})
    # the stepping kernels on the pair layout (stages of 32 levels; by default only from t = 512) for every case
    pair_cases = [k for k, cs in enumerate(CASES) if cs[0] != 64 or cs[3]]       # (three plain t = 64 cases less: time)
    # (with its own oracle positions: the pair-layout stepping is not only compared with the other runs)
    d = run([CASES[k] for k in pair_cases], {"MPVSS_FD": "1", "MPVSS_FD_L1": "2", "MPVSS_FD_MIN_SHARES": "2048", "MPVSS_FD_PAIR_MIN_T": "16",
                                             "CHECK_ORACLE": "1"})
    assert len(a) == len(CASES) and all(len(h) == 64 for h in a) and len(d) == len(pair_cases)
    for case, ha, hb, hc in zip(CASES, a, b, c):
        assert ha == hb == hc, case
    for k, hd in zip(pair_cases, d):
        assert hd == a[k], CASES[k]
    # ... and the same stepping as wide launches over the anti-diagonals of the (stage, block of steps) grid (round 5:
    # k_modp_fd_step_pair_tile, no wave waits for another): blocks of 64 steps with two-level seeding, ragged blocks of 37 steps
    # with Horner's rule for every seed -- every case, the seeding chain and both directions of the strided chains
    for env in ({"MPVSS_FD_L1": "2", "MPVSS_FD_TILE_STEPS": "64", "CHECK_ORACLE": "1"}, {"MPVSS_FD_L1": "0", "MPVSS_FD_TILE_STEPS": "37"}):
        e = run([CASES[k] for k in pair_cases], dict({"MPVSS_FD": "1", "MPVSS_FD_MIN_SHARES": "2048", "MPVSS_FD_PAIR_MIN_T": "16",
                                                      "MPVSS_FD_TILE": "2"}, **env))
        for k, he in zip(pair_cases, e):
            assert he == a[k], (CASES[k], env)


@pytest.mark.parametrize("mode", ["1", "2", "3", "4", "1p", "2p", "3p"])
def test_a_stage_that_gives_up_falls_back_to_horner(mode):
    """MPVSS_FD_TEST_FAULT makes one pipeline stage behave as if its wait had timed out (1: top stage of the first
    forward stepping chain, 2: a middle stage of the last backward chain, 3: a stage of the stride-1 seeding chain,
    4: the top stage of the table pipeline): it clears the device flag and poisons its output; the stages below must
    give up at once and the gated Horner launch must produce every X."""
    case = [(64, 8192, 1, "")]
    pair = {"MPVSS_FD_PAIR_MIN_T": "16"} if mode.endswith("p") else {}      # "p": the pair-layout stepping kernel's stages
    t0 = time.time()
    b = run(case, {"MPVSS_FD": "0"})
    ref = time.time() - t0
    t0 = time.time()
    a = run(case, dict({"MPVSS_FD": "1", "MPVSS_FD_L1": "2", "MPVSS_FD_TEST_FAULT": mode.rstrip("p")}, **pair))
    took = time.time() - t0
    assert a == b and len(a[0]) == 64
    # poisoned stages must give up at once, not wait for their 2 s timeouts one after the other (16 stages per chain)
    assert took < ref + 20, (took, ref)
