"""Forward differences on the elliptic-curve groups (consecutive positions) must give exactly Horner's results."""
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
from mpvss_rs_amd import Engine, capi
name, t, n, p0 = %r, %d, %d, %d
G = O.GROUPS[name]()
gid = capi.GROUP_SECP256K1 if name == "secp256k1" else capi.GROUP_RISTRETTO255
rng = random.Random(t * 7 + n)
order = G.group_order_int()
cm = [G.generate_public_key(rng.randrange(1, order)) for _ in range(t)]
%s
enc = b"".join(G.element_to_bytes(c) for c in cm)
eng = Engine(0)
pos = list(range(p0, p0 + n))
out = eng.ec_commit_eval(gid, enc, pos)
L = len(out) // n
for i in (0, 1, t, n // 2, n - 1):
    assert out[i * L:(i + 1) * L] == G.element_to_bytes(O.commitment_eval(G, cm, pos[i])), i
print(hashlib.sha256(out).hexdigest())
'''


def run(code, env_extra):
    env = dict(os.environ, **env_extra)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=1200)
    assert out.returncode == 0, out.stderr[-3000:]
    return out.stdout.strip()


@pytest.mark.parametrize("name", ["secp256k1", "ristretto255"])
@pytest.mark.parametrize("t,n,p0,extra", [(16, 4096, 1, ""), (20, 4200, 5, ""), (64, 8192, 1, "cm[3] = cm[2]"),
                                           (200, 4100, 123456789, "")])
def test_ec_fd_equals_horner(name, t, n, p0, extra):
    code = CODE % (ROOT, os.path.join(ROOT, "oracle"), name, t, n, p0, extra)
    a = run(code, {"MPVSS_EC_FD": "1"})
    b = run(code, {"MPVSS_EC_FD": "0"})
    assert a == b and len(a) == 64
