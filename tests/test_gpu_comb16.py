"""The wide fixed-base comb (16-bit windows, used for batches >= MPVSS_COMB16_MIN) must give exactly the results
of the narrow one and of the oracle, for both generators, including extreme exponents."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r'''
import sys, random, hashlib
sys.path.insert(0, %r); sys.path.insert(0, %r)
import mpvss_oracle as O
from mpvss_rs_amd import Engine
G = O.ModpGroup(); Q = G.q
fx = lambda v: v.to_bytes(256, "big")
rng = random.Random(99)
n = 8192
exps = [rng.randrange(G.q_minus_1) for _ in range(n)]
exps[0] = 0; exps[1] = 1; exps[2] = G.q_minus_1 - 1; exps[3] = (1 << 2048) - 1; exps[4] = 65535; exps[5] = 65536
exps[6] = int("ffff0000" * 64, 16)
eng = Engine(0)
h = hashlib.sha256()
for base in (4, 2):
    out = eng.batch_exp_fixed_base(fx(base), b"".join(map(fx, exps)))
    h.update(out)
    for i in list(range(8)) + [n // 2, n - 1]:
        assert int.from_bytes(out[i * 256:(i + 1) * 256], "big") == pow(base, exps[i], Q), (base, i)
print(h.hexdigest())
'''


def run(env_extra):
    env = dict(os.environ, **env_extra)
    code = CODE % (ROOT, os.path.join(ROOT, "oracle"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    return out.stdout.strip()


def test_wide_comb_equals_narrow_comb_and_oracle():
    wide = run({"MPVSS_COMB16_MIN": "8192"})
    narrow = run({"MPVSS_COMB16_MIN": "0"})
    assert wide == narrow and len(wide) == 64
