"""Forward differences on the elliptic-curve groups (consecutive positions) must give exactly Horner's results."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = [(name, t, n, p0, extra) for name in ("secp256k1", "ristretto255")
         for t, n, p0, extra in [(16, 4096, 1, ""), (20, 4200, 5, ""), (64, 8192, 1, "cm[3] = cm[2]"),
                                 (200, 4100, 123456789, "")]]

CODE = r'''
import os, sys, random, hashlib
sys.path.insert(0, %r); sys.path.insert(0, %r)
import mpvss_oracle as O
from mpvss_rs_amd import Engine, capi
eng = Engine(0)
for name, t, n, p0, extra in %r:
    G = O.GROUPS[name]()
    gid = capi.GROUP_SECP256K1 if name == "secp256k1" else capi.GROUP_RISTRETTO255
    rng = random.Random(t * 7 + n)
    order = G.group_order_int()
    cm = [G.generate_public_key(rng.randrange(1, order)) for _ in range(t)]
    exec(extra)
    enc = b"".join(G.element_to_bytes(c) for c in cm)
    pos = list(range(p0, p0 + n))
    out = eng.ec_commit_eval(gid, enc, pos)
    L = len(out) // n
    for i in ((0, 1, t, n // 2, n - 1) if os.environ.get("CHECK_ORACLE") else ()):
        assert out[i * L:(i + 1) * L] == G.element_to_bytes(O.commitment_eval(G, cm, pos[i])), (name, t, n, i)
    print(hashlib.sha256(out).hexdigest())
'''


def run(env_extra):
    code = CODE % (ROOT, os.path.join(ROOT, "oracle"), CASES)
    env = dict(os.environ, **env_extra)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=1800)
    assert out.returncode == 0, out.stderr[-3000:]
    return out.stdout.split()


def test_ec_fd_equals_horner():
    a = run({"MPVSS_EC_FD": "1", "MPVSS_EC_FD_L1": "0", "CHECK_ORACLE": "1"})     # Horner for every seed
    b = run({"MPVSS_EC_FD": "0"})
    c = run({"MPVSS_EC_FD": "1", "MPVSS_EC_FD_L1": "2", "CHECK_ORACLE": "1"})     # two-level seeding (what pipelined boxes use)
    assert len(a) == len(CASES) and all(len(h) == 64 for h in a)
    for case, ha, hb, hc in zip(CASES, a, b, c):
        assert ha == hb == hc, case
