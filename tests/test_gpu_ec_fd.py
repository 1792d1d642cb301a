"""Forward differences on the elliptic-curve groups (consecutive positions) must give exactly Horner's results."""
import os
import random
import struct
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = [(name, t, n, p0, extra) for name in ("secp256k1", "ristretto255")
         for t, n, p0, extra in [(16, 4096, 1, ""), (20, 4200, 5, ""), (64, 8192, 1, "cm[3] = cm[2]"),
                                 (200, 4100, 123456789, ""), (256, 16384, 77, "")]]

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
    for i in ((0, 1, t, n // 2, n - 1) if os.environ.get("CHECK_ORACLE") and t <= 200 else ()):
        assert out[i * L:(i + 1) * L] == G.element_to_bytes(O.commitment_eval(G, cm, pos[i])), (name, t, n, i)
    print(hashlib.sha256(out).hexdigest())
'''


def run(env_extra):
    code = CODE % (ROOT, os.path.join(ROOT, "oracle"), CASES)
    env = dict(os.environ, **env_extra)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=1800)
    assert out.returncode == 0, out.stderr[-3000:]
    return out.stdout.split()


_HORNER = []


def horner():
    """Horner's rule for every case (MPVSS_EC_FD=0), once per session"""
    if not _HORNER:
        _HORNER.append(run({"MPVSS_EC_FD": "0"}))
    return _HORNER[0]


def test_ec_fd_equals_horner():
    # MPVSS_EC_FD_QUAD=2: the stepping launches as pipelines of quad-lane stages (ec_quad.h; what a call that has the chip to itself
    # uses), 0: one workgroup per chain (what batched boxes use).  MPVSS_EC_FD_L1=0: Horner for every seed, 2: two-level seeding.
    b = horner()
    assert len(b) == len(CASES) and all(len(h) == 64 for h in b)
    for quad, l1 in (("2", "0"), ("2", "2"), ("0", "2")):
        a = run({"MPVSS_EC_FD": "1", "MPVSS_EC_FD_L1": l1, "MPVSS_EC_FD_QUAD": quad, "CHECK_ORACLE": "1" if quad == "2" else ""})
        for case, ha, hb in zip(CASES, a, b):
            assert ha == hb, (case, quad, l1)
    # secp256k1 with eight lanes per point addition (OctSecp: stepping and tables; off by default), here with the seeds by 8 lanes
    # bit by bit
    a = run({"MPVSS_EC_FD": "1", "MPVSS_EC_FD_L1": "0", "MPVSS_EC_FD_QUAD": "2", "MPVSS_EC_FD_OCT": "1", "MPVSS_EC_FD_SEEDS_WIN": "0"})
    assert a == b


@pytest.mark.parametrize("fault", ["1", "2"])
def test_a_quad_stage_that_gives_up_falls_back_to_horner(fault):
    """MPVSS_EC_FD_TEST_FAULT: the second stage of the first chain of the stepping (1) / table (2) pipeline behaves as if its
    wait had timed out -- it clears the box's gate and poisons its output; the stages below give up at once and the gated
    Horner launch produces every X."""
    a = run({"MPVSS_EC_FD": "1", "MPVSS_EC_FD_L1": "2", "MPVSS_EC_FD_QUAD": "2", "MPVSS_EC_FD_TEST_FAULT": fault})
    assert a == horner()


def test_quad_lane_addition_against_the_one_lane_formulas(tmp_path):
    """tests/ec_quad_unit.hip: P = a G, Q = +-b G per quad of lanes; P + Q and (-P) + Q by mpvss_rs_amd/csrc/ec_quad.h against
    the complete one-lane formulas of ec_curves.h, compared projectively on the device -- random pairs, doublings, P + (-P),
    the identity on either side and on both, small multiples."""
    exe = os.path.join(ROOT, "tests", "_build", "ec_quad_unit")
    assert os.path.exists(exe), "build it with `make -C mpvss_rs_amd/csrc examples` (__graft_entry__.build() does)"
    rng = random.Random(5)
    rows = []
    for i in range(4096):
        a, b, fl = rng.getrandbits(64), rng.getrandbits(64), rng.getrandbits(1)
        k = i % 16
        if k == 1: b, fl = a, 0            # a doubling
        if k == 2: b, fl = a, 1            # P + (-P)
        if k == 3: a = 0                   # the identity + Q
        if k == 4: b = 0                   # P + the identity
        if k == 5: a = b = 0
        if k == 6: a, b, fl = 1, 1, 0
        if k == 7: a, b = 1, 2
        if k == 8: a, b = 2**64 - 1, 1
        rows.append((a, b, fl))
    path = tmp_path / "pairs.bin"
    path.write_bytes(b"".join(struct.pack("<QQQ", *r) for r in rows))
    for group in (1, 2, 3):       # secp256k1 by quads, ristretto255 by quads, secp256k1 by eight lanes
        out = subprocess.run([exe, str(group), str(path), str(len(rows))], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0 and out.stdout.startswith("bad 0"), (group, out.stdout[:600], out.stderr[-600:])
