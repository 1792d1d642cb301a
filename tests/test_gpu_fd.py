"""Forward differences in the exponent (consecutive positions) must give exactly Horner's results."""
import os
import random
import subprocess
import sys

import pytest

import mpvss_oracle as O
from helpers import EB, cat, split

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = O.ModpGroup()
Q = G.q


def run(code, env_extra):
    env = dict(os.environ, **env_extra)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    return out.stdout


CODE = r'''
import sys, random, hashlib
sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)
import mpvss_oracle as O
from mpvss_rs_amd import Engine
G = O.ModpGroup(); Q = G.q
fx = lambda v: v.to_bytes(256, "big")
rng = random.Random(%d)
t, n, p0 = %d, %d, %d
cm = [pow(4, rng.randrange(Q - 1), Q) for _ in range(t)]
%s
eng = Engine(0)
pos = list(range(p0, p0 + n))
out = eng.commit_eval(b"".join(map(fx, cm)), pos)
h = hashlib.sha256(out).hexdigest()
# spot-check against the oracle
for i in ((0, 1, t - 1, t, n // 2, n - 1) if t < 512 else (0, n // 2 + 1, n - 1)):
    assert int.from_bytes(out[i * 256:(i + 1) * 256], "big") == O.commitment_eval(G, cm, pos[i]), i
print(h)
'''


@pytest.mark.parametrize("t,n,p0,extra", [(16, 4096, 1, ""), (64, 8192, 1, ""), (33, 5000, 777, ""),
                                           (256, 8192, 1, ""), (33, 9001, 777, ""), (17, 20011, 123456789, ""),
                                           (100, 12345, 2, "cm[99] = Q - 1"), (1024, 16384, 1, ""), (512, 9000, 40000, ""), (64, 2048, 1, ""),
                                           (16, 2100, 7, ""), (257, 5000, 3, ""), (64, 8192, 1, "cm[5] = 0"), (64, 8192, 1, "cm[0] = Q")])
def test_fd_equals_horner(t, n, p0, extra):
    code = CODE % (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), t * 1000 + n, t, n, p0, extra)
    a = run(code, {"MPVSS_FD": "1", "MPVSS_FD_MIN_SHARES": "2048"}).strip()
    b = run(code, {"MPVSS_FD": "0"}).strip()
    assert a == b and len(a) == 64


def test_a_stage_that_gives_up_falls_back_to_horner():
    """MPVSS_FD_TEST_FAULT=1 makes one pipeline stage behave as if its wait had timed out: it clears the device flag
    and poisons its output; the stages below must give up at once and the gated Horner launch must produce every X."""
    code = CODE % (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), 4242, 64, 8192, 1, "")
    import time
    t0 = time.time()
    b = run(code, {"MPVSS_FD": "0"}).strip()
    ref = time.time() - t0
    t0 = time.time()
    a = run(code, {"MPVSS_FD": "1", "MPVSS_FD_TEST_FAULT": "1"}).strip()
    took = time.time() - t0
    assert a == b and len(a) == 64
    # poisoned stages must give up at once, not wait for their 2 s timeouts one after the other (16 stages per chain)
    assert took < ref + 20, (took, ref)
