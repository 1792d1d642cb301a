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


def test_comb_dual_exp_on_the_pair_layout_with_edge_operands(engine):
    """verify_share's a1 = G^r pk^c and the participant's / dealer's fixed-base powers on the pair layout (k_modp_comb16_dual_exp_pair,
    k_modp_comb16_twin_exp_pair: batches of 8192 numbers and more, the wide comb): mpvss_modp_dleq_commitments with a generator as g1 and a
    challenge PER SHARE, 8197 shares (a ragged last wave), against Python integers -- r in {0, 1, q - 2, 2^2047, 0xffff.., 0x0000ffff..}
    (comb digits 0 and 0xffff), c in {0, 1, 15, 16, 2^256 - 1, 2^255, nibble patterns}, pk in {1, q - 1, 2, random} -- and the same a1
    through the quad-layout kernel (MPVSS_PAIR bit 0 is not read again, so: the narrow path of 48 shares)."""
    import random
    import mpvss_oracle as O
    Q = O.ModpGroup().q
    fx = lambda v: v.to_bytes(256, "big")
    rng = random.Random(20261)
    n = 8197
    r = [rng.randrange(Q - 1) for _ in range(n)]
    c = [rng.randrange(1 << 256) for _ in range(n)]
    pk = [pow(2, rng.randrange(1, Q - 1), Q) for _ in range(40)]
    pk = [pk[i % 40] for i in range(n)]
    edge_r = [0, 1, Q - 2, 1 << 2047, int("ffff" * 128, 16) % (Q - 1), int("0000ffff" * 64, 16), int("ffff0000" * 64, 16) % (Q - 1), 65535, 65536]
    edge_c = [0, 1, 15, 16, (1 << 256) - 1, 1 << 255, int("f0" * 32, 16), int("0f" * 32, 16), 255]
    edge_pk = [1, Q - 1, 2, 4, pk[0], pk[1], Q - 2, 3, 5]
    at = [0, 1, 31, 32, 63, 64, 4097, 8191, n - 1]
    for i, a, b, y in zip(at, edge_r, edge_c, edge_pk):
        r[i], c[i], pk[i] = a, b, y
    # every edge exponent against every edge challenge once more, further in
    k = 100
    for a in edge_r:
        for b in edge_c:
            r[k], c[k] = a, b
            k += 1
    sample = sorted(set(at + list(range(100, k)) + [rng.randrange(n) for _ in range(40)]))
    flat = lambda xs: b"".join(map(fx, xs))
    # g1 = G = 2 (a generator: the comb), h1 = pk; g2 / h2: a per-share base pair for the second commitment (the a2 kernel's business)
    a1, a2 = engine.dleq_commitments(fx(2), flat(pk), flat(pk), flat(pk), flat(r), flat(c), True)
    for i in sample:
        assert int.from_bytes(a1[i * 256:(i + 1) * 256], "big") == pow(2, r[i], Q) * pow(pk[i], c[i], Q) % Q, i
        assert int.from_bytes(a2[i * 256:(i + 1) * 256], "big") == pow(pk[i], r[i], Q) * pow(pk[i], c[i], Q) % Q, i
    # the quad-layout kernel on the first 48 (below the pair layout's 64-share switch and the wide comb's 8192): same bytes
    b1, _ = engine.dleq_commitments(fx(2), flat(pk[:48]), flat(pk[:48]), flat(pk[:48]), flat(r[:48]), flat(c[:48]), True)
    assert b1 == a1[:48 * 256]
    # the subgroup generator g = 4 with ONE shared challenge (stride 0)
    a1s, _ = engine.dleq_commitments(fx(4), flat(pk), flat(pk), flat(pk), flat(r), fx(c[5]), False)
    for i in sample[:60]:
        assert int.from_bytes(a1s[i * 256:(i + 1) * 256], "big") == pow(4, r[i], Q) * pow(pk[i], c[5], Q) % Q, i
