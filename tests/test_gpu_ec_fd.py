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


NOMEM_CODE = r'''
import os, sys, random, hashlib
sys.path.insert(0, %r)
from mpvss_rs_amd import Engine, capi
eng = Engine(0)
for gid, sb in ((capi.GROUP_SECP256K1, "big"), (capi.GROUP_RISTRETTO255, "little")):
    rng = random.Random(gid)
    n, t = 4200, 20
    sc = lambda: rng.randrange(1, 2**250).to_bytes(32, sb)
    pk = eng.ec_batch_exp_generator(gid, b"".join(sc() for _ in range(n)))
    coeffs = b"".join(sc() for _ in range(t))
    cm = eng.ec_batch_exp_generator(gid, coeffs)
    pos = list(range(3, 3 + n))
    box = eng.ec_deal(gid, coeffs, pos, pk, b"".join(sc() for _ in range(n)))
    one = eng.ec_verify_distribution(gid, cm, pos, pk, box["Y"], box["responses"], box["challenge"], dump=True)
    assert one["verdict"] and one["digest"] == box["digest"] and one["X"] == box["X"], "lone box"
    b = dict(commitments=cm, positions=pos, pubkeys=pk, shares=box["Y"], responses=box["responses"], challenge=box["challenge"])
    bad = dict(b, responses=bytes([box["responses"][0] ^ 1]) + box["responses"][1:])
    res = eng.ec_verify_many(gid, [b, bad, b, b], depth=4, hash_threads=2)
    assert [v for v, _ in res] == [True, False, True, True] and all(d == box["digest"] for v, d in res if v), "batched boxes"
    print(hashlib.sha256(one["X"]).hexdigest())
assert "hipMalloc" not in eng.last_error() and "memory" not in eng.last_error().lower(), eng.last_error()
'''


def test_optional_buffers_that_cannot_be_allocated_cost_speed_not_the_call():
    """the hand-over space of the quad-lane pipelines (177 MB at n = 65536, up to 4 GB per slot) and the seeds'
    window tables are OPTIONAL -- without them the one-workgroup-per-chain kernels run.  MPVSS_TEST_OPTIONAL_NOMEM=1 makes every such
    allocation fail through hipMalloc itself, so HIP's sticky last error is really set; before the fix the next launcher's
    `return hipGetLastError()` turned it into MPVSS_E_DEVICE for the whole call.  Same X as Horner, same verdicts and digests, rc OK,
    nothing recorded as the context's last error: X alone, a lone box, and boxes whose X paths are batched (xb.hand)."""
    env = {"MPVSS_EC_FD": "1", "MPVSS_EC_FD_L1": "0", "MPVSS_EC_FD_QUAD": "2", "MPVSS_TEST_OPTIONAL_NOMEM": "1"}
    assert run(env) == horner()
    out = {}
    for nomem in ("1", "0"):
        r = subprocess.run([sys.executable, "-c", NOMEM_CODE % ROOT], capture_output=True, text=True, timeout=900,
                           env=dict(os.environ, MPVSS_EC_FD_QUAD="2", MPVSS_TEST_OPTIONAL_NOMEM=nomem))
        assert r.returncode == 0, r.stderr[-3000:]
        out[nomem] = r.stdout.split()
    assert out["1"] == out["0"] and len(out["1"]) == 2
