"""The MODP scalar ring Z/(q-1) on the device: mpvss_modp_poly_eval_device (src/polynomial.rs:50-58 followed by the caller's
`% order`, participant.rs:202) and mpvss_modp_dleq_responses_device (src/dleq.rs:42-50) against Python integers and against
the host functions of the same C ABI.  The kernels work mod q' = (q-1)/2 with the parity on the side, so the cases aim at
both halves of the Chinese remainder: even / odd positions and coefficients, values just below q-1, operands that are not
reduced."""
import random

import pytest
import torch

from helpers import EB, MODP_ORDER as ORDER
from mpvss_rs_amd import capi
from mpvss_rs_amd.capi import EngineError

pytestmark = pytest.mark.gpu
fx = lambda v: v.to_bytes(EB, "big")
QH = ORDER // 2


def dev_u8(b):
    return torch.frombuffer(bytearray(b), dtype=torch.uint8).to("cuda:0")


def poly_eval_device(engine, coeffs, positions):
    n = len(positions)
    d_pos = torch.tensor(positions, dtype=torch.int64, device="cuda:0")
    out = torch.full((n * EB,), 0xA5, dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()          # torch's fill runs on torch's stream, the engine's kernels on the engine's
    engine.poly_eval_device(b"".join(map(fx, coeffs)), d_pos.data_ptr(), n, out.data_ptr())
    raw = bytes(out.cpu().numpy().tobytes())
    return [int.from_bytes(raw[i * EB:(i + 1) * EB], "big") for i in range(n)]


@pytest.mark.parametrize("t", [1, 2, 7, 64])
def test_poly_eval_device_against_integers(engine, t):
    rng = random.Random(100 + t)
    positions = ([0, 1, 2, 3, 4, 65535, 65536, 2**29 - 1, 2**29, 2**29 + 1, 2**40 + 5, 2**58, 2**62 + 3, 2**63 - 1] +
                 [rng.randrange(1, 1 << 20) for _ in range(37)])
    specials = [0, 1, ORDER - 1, QH, QH + 1, QH - 1, 2**2048 - 1, ORDER, ORDER + 1]          # the last three are not reduced
    for trial in range(3):
        coeffs = [rng.randrange(ORDER) for _ in range(t)]
        for k in range(min(t, 3)):
            if trial:
                coeffs[rng.randrange(t)] = specials[(trial * 3 + k) % len(specials)]
        want = [sum(a * pow(i, j, ORDER) for j, a in enumerate(coeffs)) % ORDER for i in positions]
        assert poly_eval_device(engine, coeffs, positions) == want
        if all(a < 2**2048 for a in coeffs):
            host = capi.poly_eval(0, b"".join(map(fx, coeffs)), positions)
            assert [int.from_bytes(host[i * EB:(i + 1) * EB], "big") for i in range(len(positions))] == want


def test_poly_eval_device_of_a_full_box(engine):
    """the dealer's shape: 65536 consecutive positions, t = 256 -- every value equal to the host function's (which takes the
    forward-difference route for such a run)"""
    rng = random.Random(7)
    coeffs = [rng.randrange(ORDER) for _ in range(256)]
    positions = list(range(1, 65537))
    got = poly_eval_device(engine, coeffs, positions)
    host = capi.poly_eval(0, b"".join(map(fx, coeffs)), positions)
    assert got == [int.from_bytes(host[i * EB:(i + 1) * EB], "big") for i in range(65536)]
    for i in (1, 2, 40000, 65536):
        assert got[i - 1] == sum(a * pow(i, j, ORDER) for j, a in enumerate(coeffs)) % ORDER


def test_poly_eval_device_rejects_a_negative_position(engine):
    with pytest.raises(EngineError):
        poly_eval_device(engine, [5, 6], [1, 2, -3, 4])
    assert "negative position" in engine.last_error()
    assert poly_eval_device(engine, [5, 6], [1, 2]) == [11, 17]


def test_dleq_responses_device_against_integers(engine):
    rng = random.Random(11)
    n = 203
    w = [rng.randrange(ORDER) for _ in range(n)]
    a = [rng.randrange(ORDER) for _ in range(n)]
    w[:6] = [0, ORDER - 1, 1, QH, 2**2048 - 1, 0]
    a[:6] = [0, ORDER - 1, QH, QH, ORDER - 1, 1]
    d_w, d_a = dev_u8(b"".join(map(fx, w))), dev_u8(b"".join(map(fx, a)))
    for c in (rng.randrange(1 << 256), rng.randrange(1 << 256) | 1, 0, 1, 2, QH, QH + 1, ORDER - 1, ORDER, 2**2048 - 1, rng.randrange(ORDER)):
        out = torch.full((n * EB,), 0x5A, dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()
        engine.dleq_responses_device(d_w.data_ptr(), d_a.data_ptr(), fx(c), n, out.data_ptr())
        raw = bytes(out.cpu().numpy().tobytes())
        got = [int.from_bytes(raw[i * EB:(i + 1) * EB], "big") for i in range(n)]
        assert got == [(wi - ai * c) % ORDER for wi, ai in zip(w, a)], f"c = {c:#x}"
        if c < ORDER and max(w) < ORDER:
            pass
    host = capi.dleq_responses(0, b"".join(map(fx, [x % ORDER for x in w])), b"".join(map(fx, a)), fx(12345))
    out = torch.zeros(n * EB, dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    engine.dleq_responses_device(dev_u8(b"".join(map(fx, [x % ORDER for x in w]))).data_ptr(), d_a.data_ptr(), fx(12345), n, out.data_ptr())
    assert bytes(out.cpu().numpy().tobytes()) == host


def test_deal_compute_is_the_oracles_dealer(engine):
    """mpvss_modp_deal_compute + mpvss_modp_distribute_absorb + mpvss_modp_dleq_responses_device == the oracle's
    distribute_secret (participant.rs:160-286) for the same polynomial and witnesses: X, Y, a1, a2 of every share, the
    transcript digest, the challenge and the responses; two blocks in flight; a negative position fails its block at absorb."""
    import hashlib
    import mpvss_oracle as O
    from helpers import make_modp_instance
    n, t = 23, 4
    g, privs, pks, coeffs, ws, box = make_modp_instance(n, t, 41)
    flat = O.box_to_flat(g, box)
    d_pos = torch.tensor(flat["positions"], dtype=torch.int64, device="cuda:0")
    d_pk, d_w = dev_u8(flat["publickeys"]), dev_u8(b"".join(map(fx, ws)))
    d_p = [torch.zeros(n * EB, dtype=torch.uint8, device="cuda:0") for _ in range(2)]
    cb = b"".join(map(fx, coeffs))
    torch.cuda.synchronize()
    engine.deal_compute(cb, d_pos.data_ptr(), d_pk.data_ptr(), d_w.data_ptr(), n, d_p[0].data_ptr())
    engine.deal_compute(cb, d_pos.data_ptr(), d_pk.data_ptr(), d_w.data_ptr(), n, d_p[1].data_ptr())
    for k in range(2):
        st, X, Y, a1, a2 = engine.distribute_absorb(capi.transcript_init(), n)
        digest = capi.transcript_verdict(st, bytes(EB))[1]
        assert digest == box["_digest"] and Y == flat["shares"]
        assert bytes(d_p[k].cpu().numpy().tobytes()) == b"".join(fx(sum(a * pow(i, j, ORDER) for j, a in enumerate(coeffs)) % ORDER)
                                                                   for i in flat["positions"])
        c = int.from_bytes(hashlib.sha256(digest).digest(), "big") % QH
        assert fx(c) == flat["challenge"]
        d_r = torch.zeros(n * EB, dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()
        engine.dleq_responses_device(d_w.data_ptr(), d_p[k].data_ptr(), fx(c), n, d_r.data_ptr())
        assert bytes(d_r.cpu().numpy().tobytes()) == flat["responses"]
    bad = d_pos.clone(); bad[7] = -8
    torch.cuda.synchronize()
    engine.deal_compute(cb, bad.data_ptr(), d_pk.data_ptr(), d_w.data_ptr(), n, d_p[0].data_ptr())
    with pytest.raises(EngineError):
        engine.distribute_absorb(capi.transcript_init(), n)
    assert "negative position" in engine.last_error() and engine.blocks_in_flight() == (0, 0)


def test_deal_in_one_call_from_host_buffers(engine):
    """mpvss_modp_deal == the oracle's distribute_secret for the same polynomial and witnesses (every output, the challenge and
    the responses), and a 2100-share box dealt that way (twin-exponentiation path) verifies with the dealer's digest."""
    import hashlib
    import mpvss_oracle as O
    from helpers import make_modp_instance
    n, t = 23, 4
    g, privs, pks, coeffs, ws, box = make_modp_instance(n, t, 43)
    flat = O.box_to_flat(g, box)
    d = engine.deal(b"".join(map(fx, coeffs)), flat["positions"], flat["publickeys"], b"".join(map(fx, ws)))
    assert d["digest"] == box["_digest"] and d["Y"] == flat["shares"]
    assert d["challenge"] == flat["challenge"] and d["responses"] == flat["responses"]
    ref = engine.verify_distribution(flat["commitments"], flat["positions"], flat["publickeys"], d["Y"], d["responses"], d["challenge"],
                                     dump=True)
    assert ref["verdict"] is True and (ref["X"], ref["a1"], ref["a2"]) == (d["X"], d["a1"], d["a2"])
    rng = random.Random(9)
    n, t = 2100, 7
    coeffs = [rng.randrange(ORDER) for _ in range(t)]
    keys = [rng.randrange(1, ORDER) for _ in range(n)]
    wit = [rng.randrange(1, ORDER) for _ in range(n)]
    pk = engine.batch_exp_fixed_base(fx(2), b"".join(map(fx, keys)))
    cm = engine.batch_exp_fixed_base(fx(4), b"".join(map(fx, coeffs)))
    pos = list(range(1, n + 1))
    d = engine.deal(b"".join(map(fx, coeffs)), pos, pk, b"".join(map(fx, wit)))
    assert d["challenge"] == fx(int.from_bytes(hashlib.sha256(d["digest"]).digest(), "big") % QH)
    p = [sum(a * pow(i, j, ORDER) for j, a in enumerate(coeffs)) % ORDER for i in pos]
    c = int.from_bytes(d["challenge"], "big")
    assert d["responses"] == b"".join(fx((w - pi * c) % ORDER) for w, pi in zip(wit, p))
    res = engine.verify_distribution(cm, pos, pk, d["Y"], d["responses"], d["challenge"])
    assert res["verdict"] is True and res["digest"] == d["digest"]
    with pytest.raises(EngineError):
        engine.deal(b"".join(map(fx, coeffs)), [1, -2] + pos[2:], pk, b"".join(map(fx, wit)))


def test_deal_of_a_box_larger_than_one_block():
    """src/participant.rs:160-286 has no size limit: mpvss_modp_deal cuts a box into blocks of MAX_CHUNK shares internally
    (one transcript, the responses once the challenge is known).  Under MPVSS_MAX_CHUNK=16 (read once per process) a 75-share
    box is five blocks: every output, the digest, the challenge and the responses must be the oracle's distribute_secret, a
    second deal on the same context must not see anything of the first, and the C++ mirror of the crate API (whose
    distribute_secret binds this call, as rust/src/batch.rs does) passes the reference's own tests the same way."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)
import torch
import mpvss_oracle as O
from helpers import make_modp_instance
from mpvss_rs_amd import Engine
fx = lambda v: v.to_bytes(256, "big")
eng = Engine(0)
for n, t, seed in ((75, 5, 61), (16, 16, 62), (33, 2, 63)):
    g, privs, pks, coeffs, ws, box = make_modp_instance(n, t, seed)
    flat = O.box_to_flat(g, box)
    d = eng.deal(b"".join(map(fx, coeffs)), flat["positions"], flat["publickeys"], b"".join(map(fx, ws)))
    assert d["digest"] == box["_digest"] and d["Y"] == flat["shares"], (n, t)
    assert d["challenge"] == flat["challenge"] and d["responses"] == flat["responses"], (n, t)
    assert [int.from_bytes(d["X"][i * 256:(i + 1) * 256], "big") for i in range(n)] == box["_X"]
    assert [int.from_bytes(d["a1"][i * 256:(i + 1) * 256], "big") for i in range(n)] == box["_a1"]
    assert [int.from_bytes(d["a2"][i * 256:(i + 1) * 256], "big") for i in range(n)] == box["_a2"]
    res = eng.verify_distribution(flat["commitments"], flat["positions"], flat["publickeys"], d["Y"], d["responses"], d["challenge"])
    assert res["verdict"] is True and res["digest"] == d["digest"]
try:
    eng.deal(b"".join(map(fx, coeffs)), [1, 2, -3] + flat["positions"][3:], flat["publickeys"], b"".join(map(fx, ws)))
    raise SystemExit("a negative position was accepted")
except Exception as e:
    assert "negative position" in str(e) or "negative position" in eng.last_error(), e
assert eng.blocks_in_flight() == (0, 0)
print("chunked deal ok")
""" % (root, os.path.join(root, "oracle"), os.path.join(root, "tests"))
    env = dict(os.environ, MPVSS_MAX_CHUNK="16")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0 and "chunked deal ok" in out.stdout, out.stdout[-1500:] + out.stderr[-3000:]
    exe = os.path.join(root, "tests", "_build", "host_mirror_tests")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(root, "mpvss_rs_amd", "csrc"), "examples", "-s"])
    out = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0 and "all passed" in out.stdout, out.stdout[-1500:] + out.stderr[-3000:]


def test_lone_deal_in_chained_blocks_equals_the_one_block_dealer(engine):
    """A dealer that has the chip to itself deals a large box in blocks, block k + 1 starting on the GPU when block k is done, so that the
    transcript of block k is hashed while block k + 1 computes (capi_scalar.inc): blocks of 65536 shares, of 32768 with the participants'
    key tables.  131109 shares = 65536 + 65536 + 37 = 4 x 32768 + 37 (a ragged last block of less than one wave): every output, the digest,
    the challenge and the responses equal the ONE-block dealer's (mpvss_modp_distribute with P(i) from the host's scalar ring, and the
    responses by the host functions); the box verifies; the same with key tables (cross-call cache: blocks at key offsets 0, 32768, ...);
    a negative position in the last block fails the call and leaves nothing in the ring."""
    n, t = 131109, 5
    rng = random.Random(9091)
    sc = lambda k: b"".join(rng.randrange(1, 1 << 2040).to_bytes(EB, "big") for _ in range(k))
    pos = list(range(2, 2 + n))
    pk = engine.batch_exp_fixed_base(fx(2), sc(n))
    coeffs, wit = sc(t), sc(n)
    cm = engine.batch_exp_fixed_base(fx(4), coeffs)
    p_values = capi.poly_eval(0, coeffs, pos)
    assert engine.blocks_in_flight() == (0, 0)
    one = engine.distribute(cm, pos, pk, p_values, wit)                # one block of n shares (X from the commitments)
    got = engine.deal(coeffs, pos, pk, wit)                           # chained blocks
    for k in ("X", "Y", "a1", "a2", "digest"):
        assert got[k] == one[k], k
    import hashlib
    c = int.from_bytes(hashlib.sha256(got["digest"]).digest(), "big") % QH
    assert got["challenge"] == fx(c)
    assert got["responses"] == capi.dleq_responses(0, wit, p_values, fx(c))
    r = engine.verify_distribution(cm, pos, pk, got["Y"], got["responses"], got["challenge"])
    assert r["verdict"] and r["digest"] == got["digest"]
    assert engine.set_key_cache_lru(1, 1) == 0
    try:
        keyed = engine.deal(coeffs, pos, pk, wit)
    finally:
        assert engine.set_key_cache_lru(0) == 1
    assert keyed == got
    with pytest.raises(EngineError):
        engine.deal(coeffs, pos[:-1] + [-4], pk, wit)
    assert engine.blocks_in_flight() == (0, 0)
