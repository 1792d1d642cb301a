"""Registered public keys: verify_block_compute_keyset must produce exactly the X, a1, a2 arrays and transcript
state of verify_block_compute on the same keys (and a2 must be y^r * Y^c by the oracle's arithmetic)."""
import random

import pytest

import mpvss_oracle as O
from helpers import EB
from mpvss_rs_amd import capi

pytestmark = pytest.mark.gpu
Q = O.ModpGroup().q


def fx(v):
    return v.to_bytes(EB, "big")


# (8192: whole pair-layout waves; 8200 and 4133: a ragged last wave whose padding lanes repeat the last share; 48: below the
#  64-share switch, the quad-layout key-table kernel -- ADVICE r5)
@pytest.mark.parametrize("n,t,key_offset", [(8192, 16, 0), (8192, 32, 64), (8200, 16, 3), (4133, 16, 0), (48, 3, 5)])
def test_keyset_block_equals_plain_block(engine, n, t, key_offset):
    rng = random.Random(1000 * t + key_offset)
    nk = n + key_offset + 3
    keys = [pow(2, rng.randrange(Q - 1), Q) for _ in range(40)]
    keys = [keys[rng.randrange(40)] * pow(2, i, Q) % Q for i in range(nk)]            # cheap distinct group elements
    keys[key_offset + 5] = 1
    keys[key_offset + 6] = Q - 1
    pk = b"".join(map(fx, keys))
    cm = b"".join(fx(pow(4, rng.randrange(Q - 1), Q)) for _ in range(t))
    shares = rng.randbytes(EB * n)
    resp = bytearray(rng.randbytes(EB * n))
    resp[0:EB] = bytes(EB)                                   # r = 0
    resp[EB:2 * EB] = b"\xff" * EB                           # r = 2^2048 - 1
    resp = bytes(resp)
    chal = fx(rng.randrange(1 << 256))
    pos = list(range(1, n + 1))
    sub = pk[key_offset * EB:(key_offset + n) * EB]

    blocks0, fallbacks0 = engine.fd_stats()
    engine.verify_block_compute(cm, pos, sub, shares, resp, chal)
    st0, X0, A10, A20 = engine.verify_block_absorb_dump(capi.transcript_init(), n)
    if n >= 4096:
        assert engine.fd_stats() == (blocks0 + 1, fallbacks0)      # forward differences were used and held

    ks = engine.keyset_create(pk)
    try:
        assert engine.keyset_bytes(ks) == nk * (8 * 128 * 288 + 256)            # 7-bit windows of the eight 256-bit rows of r
        engine.verify_block_compute_keyset(cm, pos, ks, key_offset, shares, resp, chal)
        st1, X1, A11, A21 = engine.verify_block_absorb_dump(capi.transcript_init(), n)
        # a challenge that does not fit 256 bits takes the plain path inside the same entry point
        big = fx((1 << 300) + 5)
        engine.verify_block_compute_keyset(cm, pos, ks, key_offset, shares, resp, big)
        st2, _, _, A22 = engine.verify_block_absorb_dump(capi.transcript_init(), n)
        engine.verify_block_compute(cm, pos, sub, shares, resp, big)
        st3, _, _, A23 = engine.verify_block_absorb_dump(capi.transcript_init(), n)
        with pytest.raises(capi.EngineError):                # shares beyond the registered keys
            engine.verify_block_compute_keyset(cm, pos, ks, key_offset + 4, shares, resp, chal)
    finally:
        engine.keyset_destroy(ks)
    assert (X1, A11) == (X0, A10)
    assert A21 == A20 and st1 == st0
    assert A22 == A23 and st2 == st3
    c = int.from_bytes(chal, "big")
    for i in (0, 1, 2, 5, 6, n // 2, n - 2, n - 1):
        y = keys[key_offset + i]
        Y = int.from_bytes(shares[i * EB:(i + 1) * EB], "big")
        r = int.from_bytes(resp[i * EB:(i + 1) * EB], "big")
        assert int.from_bytes(A21[i * EB:(i + 1) * EB], "big") == pow(y, r, Q) * pow(Y, c, Q) % Q, i


def test_key_cache_behind_the_unchanged_verify_many(engine):
    """mpvss_ctx_set_key_cache(ctx, 3): mpvss_modp_verify_many registers by itself a public-key array that at least three large boxes
    of the call present (same pointer, same n) and verifies those boxes against the tables -- same verdicts and digests as with the
    cache off: honest boxes of three dealers against one key array, a tampered response, a tampered share, a box with a second key
    array (only one box has it: left alone), a box whose challenge does not fit 256 bits (plain path inside the same call), host
    memory.  The launch counter of the table builds says the cache was really used."""
    import ctypes as C
    n, t = 16500, 16
    rng = random.Random(77)
    sc = lambda k: b"".join(rng.randrange(1, 1 << 2040).to_bytes(EB, "big") for _ in range(k))
    pos = list(range(9, 9 + n))
    pkA = engine.batch_exp_fixed_base(fx(2), sc(n))
    pkB = engine.batch_exp_fixed_base(fx(2), sc(n))
    def deal(pk):
        coeffs = sc(t)
        d = engine.deal(coeffs, pos, pk, sc(n))
        return dict(cm=engine.batch_exp_fixed_base(fx(4), coeffs), Y=d["Y"], r=d["responses"], c=d["challenge"], digest=d["digest"])
    d1, d2, d3, dB = deal(pkA), deal(pkA), deal(pkA), deal(pkB)
    flip = lambda b, at: b[:at] + bytes([b[at] ^ 1]) + b[at + 1:]
    specs = [(pkA, d1), (pkA, dict(d2, r=flip(d2["r"], 5 * EB + 200))), (pkA, d2), (pkB, dB), (pkA, dict(d3, Y=flip(d3["Y"], 77))), (pkA, d3),
             (pkA, dict(d1, c=fx((1 << 300) + 5))), (pkA, d1)]
    bufs = {id(pkA): (C.c_uint8 * len(pkA)).from_buffer_copy(pkA), id(pkB): (C.c_uint8 * len(pkB)).from_buffer_copy(pkB)}
    posb = (C.c_int64 * n)(*pos)
    keep, arr = [], (capi.ModpBox * len(specs))()
    for i, (pk, d) in enumerate(specs):
        b = [(C.c_uint8 * len(d[k])).from_buffer_copy(d[k]) for k in ("cm", "Y", "r", "c")]
        keep.append(b)
        arr[i] = capi.ModpBox(C.addressof(b[0]), t, C.addressof(posb), C.addressof(bufs[id(pk)]), C.addressof(b[1]), C.addressof(b[2]), n,
                              C.addressof(b[3]), None, 0)

    def run():
        verdicts = (C.c_int * len(specs))()
        digests = (C.c_uint8 * (32 * len(specs)))()
        engine._check(engine.lib.mpvss_modp_verify_many(engine.ctx, capi.MPVSS_HOST, arr, len(specs), 4, 3, verdicts, C.cast(digests, C.c_void_p)),
                      "verify_many")
        return [(bool(verdicts[i]), bytes(digests)[32 * i:32 * i + 32]) for i in range(len(specs))]

    assert engine.set_key_cache(0) == 0
    engine.pipeline_stats(reset=True)
    plain = run()
    tables_plain = engine.pipeline_stats(reset=True)["kernel_launches"][2]
    assert [v for v, _ in plain] == [True, False, True, True, False, True, False, True]
    assert plain[0][1] == d1["digest"] and plain[3][1] == dB["digest"] and plain[5][1] == d3["digest"]
    import torch
    table_b = n * 1024 * 72 * 4                       # 295 KB per key
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    assert engine.set_key_cache(3) == 0
    try:
        cached = run()
        launches = engine.pipeline_stats(reset=True)["kernel_launches"][2]
        # the tables' BUFFER outlives the call (allocating it costs more than building the tables): held while the cache is on, taken by
        # the next call's tables (no second buffer), given back when the cache is switched off
        free1 = torch.cuda.mem_get_info()[0]
        assert free0 - free1 > 0.9 * table_b, (free0, free1)
        assert run() == plain
        free2 = torch.cuda.mem_get_info()[0]
        assert abs(free1 - free2) < 0.2 * table_b, (free1, free2)
    finally:
        assert engine.set_key_cache(0) == 3
    assert torch.cuda.mem_get_info()[0] - free1 > 0.8 * table_b
    assert cached == plain
    # six of the eight boxes went through the key tables: no 64-entry table of y was built for them
    assert launches == tables_plain - 6
    with pytest.raises(capi.EngineError):
        engine.set_key_cache(1)


def test_key_tables_at_the_headline_shape():
    """The configuration bench.py times as `registered_keys` (missing 4): n = 65536, t = 256, 19.3 GB of tables.  Three
    dealers' boxes against one key array through mpvss_modp_verify_many with the context's key cache on, and through an explicit key
    set: the digests are the dealers', a seeded 1 % of the shares (X, a1, a2 dumped through the key-set block path) equal Python's
    pow (helpers.modp_fast_share), a flipped response bit and a flipped share bit are rejected with the digests the plain path
    gives.  Own process: the 19 GB go back to the device afterwards."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys, random, ctypes as C
sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)
from helpers import modp_fast_share, parallel_map
from mpvss_rs_amd import Engine, capi
EB = 256
def main():
    eng = Engine(0)
    rng = random.Random(65536)
    fx = lambda v: v.to_bytes(EB, "big")
    sc = lambda k: b"".join(rng.randrange(1, 1 << 2040).to_bytes(EB, "big") for _ in range(k))
    n, t = 65536, 256
    pos = list(range(1, n + 1))
    pk = eng.batch_exp_fixed_base(fx(2), sc(n))
    wit = sc(n)
    boxes = []
    for b in range(3):
        coeffs = sc(t)
        d = eng.deal(coeffs, pos, pk, wit)
        boxes.append(dict(commitments=eng.batch_exp_fixed_base(fx(4), coeffs), positions=pos, pubkeys=pk, shares=d["Y"], responses=d["responses"],
                          challenge=d["challenge"], digest=d["digest"], X=d["X"], a1=d["a1"], a2=d["a2"]))
    flip = lambda b, at: b[:at] + bytes([b[at] ^ 1]) + b[at + 1:]
    bad_r = dict(boxes[1], responses=flip(boxes[1]["responses"], 40000 * EB + 77))
    bad_Y = dict(boxes[2], shares=flip(boxes[2]["shares"], 65535 * EB + 255))
    seq = [boxes[0], bad_r, boxes[1], bad_Y, boxes[2]]
    pkb = (C.c_uint8 * len(pk)).from_buffer_copy(pk)
    posb = (C.c_int64 * n)(*pos)
    keep, arr = [], (capi.ModpBox * len(seq))()
    for i, b in enumerate(seq):
        bufs = [(C.c_uint8 * len(b[k])).from_buffer_copy(b[k]) for k in ("commitments", "shares", "responses", "challenge")]
        keep.append(bufs)
        arr[i] = capi.ModpBox(C.addressof(bufs[0]), t, C.addressof(posb), C.addressof(pkb), C.addressof(bufs[1]), C.addressof(bufs[2]), n,
                              C.addressof(bufs[3]), None, 0)
    def run():
        vd, dg = (C.c_int * len(seq))(), (C.c_uint8 * (32 * len(seq)))()
        eng._check(eng.lib.mpvss_modp_verify_many(eng.ctx, capi.MPVSS_HOST, arr, len(seq), 5, 4, vd, C.cast(dg, C.c_void_p)), "verify_many")
        return [(bool(vd[i]), bytes(dg)[32 * i:32 * i + 32]) for i in range(len(seq))]
    plain = run()
    assert [v for v, _ in plain] == [True, False, True, False, True]
    assert [plain[i][1] for i in (0, 2, 4)] == [b["digest"] for b in boxes]
    eng.pipeline_stats(reset=True)
    assert eng.set_key_cache(3) == 0
    cached = run()
    assert eng.set_key_cache(0) == 3
    assert cached == plain, "key cache at (65536, 256): verdicts or digests differ from the plain path"
    assert eng.pipeline_stats(reset=True)["kernel_launches"][2] < 3 * len(seq), "the key tables were not used"
    # the explicit key set, block form, with dumps: every X / a1 / a2 equals the dealer's; a seeded 1 %% against Python's pow
    ks = eng.keyset_create(pk)
    try:
        assert eng.keyset_bytes(ks) == n * (8 * 128 * 288 + 256)
        b = boxes[0]
        eng.verify_block_compute_keyset(b["commitments"], pos, ks, 0, b["shares"], b["responses"], b["challenge"])
        st, X, A1, A2 = eng.verify_block_absorb_dump(capi.transcript_init(), n)
        assert capi.transcript_verdict(st, b["challenge"]) == (True, b["digest"])
        assert (X, A1, A2) == (b["X"], b["a1"], b["a2"])
        idx = sorted(random.Random(7).sample(range(n), 655)) + [0, n - 1]
        outs = parallel_map(modp_fast_share, [(b["commitments"], pos[i], pk[i * EB:(i + 1) * EB], b["shares"][i * EB:(i + 1) * EB],
                                               b["responses"][i * EB:(i + 1) * EB], b["challenge"]) for i in idx])
        for i, (x, a1, a2) in zip(idx, outs):
            s = slice(i * EB, (i + 1) * EB)
            assert (X[s], A1[s], A2[s]) == (x, a1, a2), i
    finally:
        eng.keyset_destroy(ks)
    assert eng.fd_stats()[1] == 0
    print("keyset headline ok")
if __name__ == "__main__":
    main()
""" % (root, os.path.join(root, "oracle"), os.path.join(root, "tests"))
    script = os.path.join(root, "tests", "_build", "keyset_headline_child.py")
    os.makedirs(os.path.dirname(script), exist_ok=True)
    with open(script, "w") as fh:
        fh.write(code)
    out = subprocess.run([sys.executable, script], capture_output=True, text=True, timeout=1200,
                         env=dict(os.environ, GPU_MAX_HW_QUEUES="8"))
    assert out.returncode == 0 and "keyset headline ok" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_key_cache_across_one_box_calls(engine):
    """mpvss_ctx_set_key_cache_lru(ctx, 1, 2): one-box calls (mpvss_modp_verify_distribution, host buffers: the crate's call shape) against
    the same participants -- the second box against a key array builds its tables, later ones take them; same verdicts and digests as
    with the cache off (honest boxes of three dealers, a flipped response bit, a flipped share bit, a challenge beyond 256 bits);
    a SECOND key array with room for one set only evicts the first (least recently used, nothing in flight) and the first is rebuilt
    when it comes back; the same array in another buffer (another pointer, same bytes) is recognised, one changed key byte is another
    array; 6 threads at once get the sequential results; device-memory boxes are left alone; switching the cache off frees the sets."""
    import ctypes as C
    import threading
    n, t = 16500, 16
    rng = random.Random(99)
    sc = lambda k: b"".join(rng.randrange(1, 1 << 2040).to_bytes(EB, "big") for _ in range(k))
    pos = list(range(5, 5 + n))
    pkA = engine.batch_exp_fixed_base(fx(2), sc(n))
    pkB = engine.batch_exp_fixed_base(fx(2), sc(n))

    def deal(pk):
        coeffs = sc(t)
        d = engine.deal(coeffs, pos, pk, sc(n))
        return dict(commitments=engine.batch_exp_fixed_base(fx(4), coeffs), pubkeys=pk, shares=d["Y"], responses=d["responses"],
                    challenge=d["challenge"], digest=d["digest"])
    a1, a2, a3, b1 = deal(pkA), deal(pkA), deal(pkA), deal(pkB)
    flip = lambda b, at: b[:at] + bytes([b[at] ^ 1]) + b[at + 1:]
    cases = [a1, a2, dict(a2, responses=flip(a2["responses"], 700 * EB + 9)), a3, dict(a3, shares=flip(a3["shares"], 16499 * EB + 255)),
             dict(a1, challenge=fx((1 << 300) + 5)), b1, a1, dict(a2, pubkeys=bytes(bytearray(pkA))),      # the same keys in another buffer
             dict(a1, pubkeys=flip(pkA, 100 * EB + 200))]                                                  # one key changed: another array
    verify = lambda b: (lambda r: (r["verdict"], r["digest"]))(engine.verify_distribution(b["commitments"], pos, b["pubkeys"], b["shares"],
                                                                                          b["responses"], b["challenge"]))
    assert engine.set_key_cache_lru(0) == 0
    plain = [verify(b) for b in cases]
    assert [v for v, _ in plain] == [True, True, False, True, False, False, True, True, True, False]
    assert plain[0][1] == a1["digest"] and plain[6][1] == b1["digest"]
    engine.pipeline_stats(reset=True)
    [verify(b) for b in cases]
    tables_plain = engine.pipeline_stats(reset=True)["kernel_launches"][2]          # three table launches per box without the cache
    assert tables_plain == 3 * len(cases)
    assert engine.set_key_cache_lru(1, 2) == 0
    try:
        cached = [verify(b) for b in cases]
        assert cached == plain
        # with tables: two launches per box (Y and X).  Plain: a1 (first sighting of A), the wide challenge, b1 (first sighting of B) and
        # the changed array (first sighting); a2 is the second sighting of A: built there and used from then on
        launches = engine.pipeline_stats(reset=True)["kernel_launches"][2]
        assert launches == 3 * 4 + 2 * (len(cases) - 4), launches
        # B twice more: its second sighting evicts A (room for one set); A again: first sighting after the eviction is plain, then rebuilt
        seq = [b1, b1, a1, a1, a1]
        want = [plain[6], plain[6], plain[0], plain[0], plain[0]]
        assert [verify(b) for b in seq] == want
        launches = engine.pipeline_stats(reset=True)["kernel_launches"][2]
        assert launches == 2 + 2 + 3 + 2 + 2, launches            # b1: built at its second sighting overall (the first was above)
        # six threads, mixed boxes
        res = [None] * 6

        def work(k):
            res[k] = [verify(cases[(k + j) % len(cases)]) for j in range(4)]
        ths = [threading.Thread(target=work, args=(k,)) for k in range(6)]
        [th.start() for th in ths]
        [th.join() for th in ths]
        for k in range(6):
            assert res[k] == [plain[(k + j) % len(cases)] for j in range(4)], k
        assert engine.blocks_in_flight() == (0, 0)
    finally:
        assert engine.set_key_cache_lru(0) == 1
    assert verify(a1) == plain[0]
    with pytest.raises(capi.EngineError):
        engine.set_key_cache_lru(9)


def test_dealer_takes_key_tables_from_the_cross_call_cache(engine):
    """mpvss_ctx_set_key_cache_lru with dealers (mpvss_modp_deal, mpvss_modp_distribute: host buffers, the crate's distribute_secret):
    a dealer to participants whose keys have tables computes Y_i = y_i^P(i) and a2_i = y_i^w_i from them (k_modp_keyset_twin_exp_pair:
    full-width exponents, eight 256-bit rows) -- byte-identical X, Y, a1, a2, digest, challenge and responses to the bucket kernels'
    (cache off), at a ragged size, with edge witnesses (0, 1, q - 2, 2^2047, row boundaries); Y and a2 of those against Python integers;
    the box verifies; four dealers at once; dealers and verifiers share one set of tables."""
    import threading
    n, t = 16500, 7
    rng = random.Random(4242)
    sc = lambda k: b"".join(rng.randrange(1, 1 << 2040).to_bytes(EB, "big") for _ in range(k))
    pos = list(range(3, 3 + n))
    pk = engine.batch_exp_fixed_base(fx(2), sc(n))
    q = Q
    edge = [0, 1, q - 2, 1 << 2047, (1 << 256) - 1, 1 << 256, (1 << 1792) - 1, (1 << 2047) + (1 << 255), 127, 128]
    wit = bytearray(sc(n))
    at = [0, 1, 31, 32, 63, 64, 8191, 16383, 16384, n - 1]
    for i, w in zip(at, edge):
        wit[i * EB:(i + 1) * EB] = w.to_bytes(EB, "big")
    wit = bytes(wit)
    coeffs = sc(t)
    assert engine.set_key_cache_lru(0) == 0
    engine.pipeline_stats(reset=True)
    plain = engine.deal(coeffs, pos, pk, wit)
    ms_plain = engine.pipeline_stats(reset=True)["kernel_ms"][3]
    p_values = capi.poly_eval(0, coeffs, pos)
    for i, w in zip(at, edge):
        y = int.from_bytes(pk[i * EB:(i + 1) * EB], "big")
        assert int.from_bytes(plain["a2"][i * EB:(i + 1) * EB], "big") == pow(y, w, q), i
        # (the comb's pair-layout twin launch: a1 = g^w with digits 0 and 0xffff, X = g^P(i))
        assert int.from_bytes(plain["a1"][i * EB:(i + 1) * EB], "big") == pow(4, w, q), i
        assert int.from_bytes(plain["X"][i * EB:(i + 1) * EB], "big") == pow(4, int.from_bytes(p_values[i * EB:(i + 1) * EB], "big"), q), i
    assert engine.set_key_cache_lru(1, 1) == 0
    try:
        cached = engine.deal(coeffs, pos, pk, wit)             # first sighting builds the tables (min_sightings 1) and uses them
        ms_cached = engine.pipeline_stats(reset=True)["kernel_ms"][3]
        for k in plain:
            assert cached[k] == plain[k], k
        assert ms_cached < 0.8 * ms_plain, (ms_cached, ms_plain)          # 2 x 548 instead of 2 865 operations per share
        # P(i) through the scalar ring on the host, Y against Python integers at the edge positions
        for i in at:
            y = int.from_bytes(pk[i * EB:(i + 1) * EB], "big")
            assert int.from_bytes(cached["Y"][i * EB:(i + 1) * EB], "big") == pow(y, int.from_bytes(p_values[i * EB:(i + 1) * EB], "big"), q), i
        cm = engine.batch_exp_fixed_base(fx(4), coeffs)
        r = engine.verify_distribution(cm, pos, pk, cached["Y"], cached["responses"], cached["challenge"])     # (the verifier takes the tables too)
        assert r["verdict"] and r["digest"] == plain["digest"]
        # the one-call distribute with P(i) given
        d = engine.distribute(cm, pos, pk, p_values, wit)
        for k in ("X", "Y", "a1", "a2", "digest"):
            assert d[k] == plain[k], k
        # four dealers at once, each its own polynomial and witnesses
        jobs = [(sc(t), sc(n)) for _ in range(4)]
        res = [None] * 4

        def work(k):
            res[k] = engine.deal(jobs[k][0], pos, pk, jobs[k][1])
        ths = [threading.Thread(target=work, args=(k,)) for k in range(4)]
        [th.start() for th in ths]
        [th.join() for th in ths]
        assert engine.blocks_in_flight() == (0, 0)
        # dealers and verifiers of the same participants at once on the one set of tables: three of each
        mixed = [None] * 6

        def mix(k):
            if k < 3:
                mixed[k] = engine.deal(jobs[k][0], pos, pk, jobs[k][1])
            else:
                mixed[k] = [engine.verify_distribution(cm, pos, pk, cached["Y"], cached["responses"], cached["challenge"]) for _ in range(2)]
        ths = [threading.Thread(target=mix, args=(k,)) for k in range(6)]
        [th.start() for th in ths]
        [th.join() for th in ths]
        assert engine.blocks_in_flight() == (0, 0)
        for k in range(3):
            assert mixed[k] == res[k], k
        for k in range(3, 6):
            assert all(v["verdict"] and v["digest"] == plain["digest"] for v in mixed[k]), k
    finally:
        assert engine.set_key_cache_lru(0) == 1
    for k in range(4):
        want = engine.deal(jobs[k][0], pos, pk, jobs[k][1])
        for f in want:
            assert res[k][f] == want[f], (k, f)


@pytest.mark.parametrize("n,t,key_offset", [(8200, 9, 3), (4133, 4, 0), (48, 3, 5)])
def test_deal_compute_keyset_equals_deal_compute(engine, n, t, key_offset):
    """mpvss_modp_deal_compute_keyset (a dealer's block to REGISTERED keys, device buffers) == mpvss_modp_deal_compute with those keys:
    P(i), X, Y, a1, a2 and the transcript state byte for byte -- ragged last waves, a key offset into a larger set, and 48 shares
    (below the 64-share switch: the plain kernels with the set's device copy of the keys)."""
    import torch
    rng = random.Random(n * 31 + t)
    sc = lambda k: b"".join(rng.randrange(1, 1 << 2040).to_bytes(EB, "big") for _ in range(k))
    dev = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to("cuda:0")
    total = n + key_offset + 2
    pk = engine.batch_exp_fixed_base(fx(2), sc(total))
    pos = list(range(7, 7 + n))
    coeffs, wit = sc(t), sc(n)
    d_pos = torch.tensor(pos, dtype=torch.int64, device="cuda:0")
    d_pk, d_w = dev(pk[key_offset * EB:(key_offset + n) * EB]), dev(wit)
    d_p = [torch.zeros(n * EB, dtype=torch.uint8, device="cuda:0") for _ in range(2)]
    ks = engine.keyset_create(pk)
    try:
        torch.cuda.synchronize()
        engine.deal_compute(coeffs, d_pos.data_ptr(), d_pk.data_ptr(), d_w.data_ptr(), n, d_p[0].data_ptr())
        plain = engine.distribute_absorb(capi.transcript_init(), n)
        engine.deal_compute_keyset(coeffs, d_pos.data_ptr(), ks, key_offset, d_w.data_ptr(), n, d_p[1].data_ptr())
        keyed = engine.distribute_absorb(capi.transcript_init(), n)
        assert keyed == plain
        assert bytes(d_p[0].cpu().numpy().tobytes()) == bytes(d_p[1].cpu().numpy().tobytes())
        for i in (0, n // 2, n - 1):              # Y_i = y_i^P(i), a2_i = y_i^w_i against Python integers
            y = int.from_bytes(pk[(key_offset + i) * EB:(key_offset + i + 1) * EB], "big")
            p_i = int.from_bytes(bytes(d_p[1][i * EB:(i + 1) * EB].cpu().numpy().tobytes()), "big")
            assert int.from_bytes(keyed[2][i * EB:(i + 1) * EB], "big") == pow(y, p_i, Q)
            assert int.from_bytes(keyed[4][i * EB:(i + 1) * EB], "big") == pow(y, int.from_bytes(wit[i * EB:(i + 1) * EB], "big"), Q)
        with pytest.raises(capi.EngineError):     # shares beyond the set
            engine.deal_compute_keyset(coeffs, d_pos.data_ptr(), ks, total - n + 1, d_w.data_ptr(), n, d_p[1].data_ptr())
        assert engine.blocks_in_flight() == (0, 0)
    finally:
        engine.keyset_destroy(ks)
