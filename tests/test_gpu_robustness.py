"""Paths the parity tests do not reach by themselves: device-resident buffers (torch tensors' data_ptr), the
multi-chunk loop, two blocks in flight (pipelining), concurrent host threads on one context, a caller stream."""
import ctypes as C
import os
import random
import subprocess
import sys
import threading

import pytest

import mpvss_oracle as O
from helpers import EB, cat, make_modp_instance, split
from mpvss_rs_amd import capi

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_device_resident_buffers_match_host_buffers(engine):
    import torch
    g, privs, pks, coeffs, ws, box = make_modp_instance(21, 4, 5)
    flat = O.box_to_flat(g, box)
    host = engine.verify_distribution(flat["commitments"], flat["positions"], flat["publickeys"], flat["shares"],
                                      flat["responses"], flat["challenge"], dump=True)
    dev = torch.device("cuda", 0)
    t8 = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    d_cm, d_pk, d_sh, d_rs = t8(flat["commitments"]), t8(flat["publickeys"]), t8(flat["shares"]), t8(flat["responses"])
    d_pos = torch.tensor(flat["positions"], dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    n, t = flat["n"], flat["t"]
    ch = (C.c_uint8 * EB).from_buffer_copy(flat["challenge"])
    verdict = C.c_int(0)
    dg = (C.c_uint8 * 32)()
    X = (C.c_uint8 * (n * EB))(); A1 = (C.c_uint8 * (n * EB))(); A2 = (C.c_uint8 * (n * EB))()
    vp = lambda x: C.c_void_p(x.data_ptr())
    rc = engine.lib.mpvss_modp_verify_distribution(engine.ctx, capi.MPVSS_DEVICE, vp(d_cm), t, vp(d_pos), vp(d_pk), vp(d_sh),
                                                   vp(d_rs), n, C.cast(ch, C.c_void_p), C.byref(verdict), dg, X, A1, A2)
    assert rc == 0 and verdict.value == 1
    assert bytes(dg) == host["digest"] and bytes(X) == host["X"] and bytes(A1) == host["a1"] and bytes(A2) == host["a2"]
    # device outputs of a Group-level call
    out = torch.empty(n * EB, dtype=torch.uint8, device=dev)
    rc = engine.lib.mpvss_modp_commit_eval(engine.ctx, capi.MPVSS_DEVICE, vp(d_cm), t, vp(d_pos), n, vp(out))
    assert rc == 0
    torch.cuda.synchronize()
    assert bytes(out.cpu().numpy().tobytes()) == host["X"]
    # negative position in a device array is reported, not silently computed
    d_bad = d_pos.clone(); d_bad[3] = -7
    torch.cuda.synchronize()
    rc = engine.lib.mpvss_modp_commit_eval(engine.ctx, capi.MPVSS_DEVICE, vp(d_cm), t, vp(d_bad), n, vp(out))
    assert rc == -1


def test_two_blocks_in_flight_then_absorbed_in_order(engine):
    g, privs, pks, coeffs, ws, box = make_modp_instance(10, 3, 6)
    flat = O.box_to_flat(g, box)
    a = slice(0, 4 * EB); b = slice(4 * EB, 10 * EB)
    engine.verify_block_compute(flat["commitments"], flat["positions"][:4], flat["publickeys"][a], flat["shares"][a],
                                flat["responses"][a], flat["challenge"])
    engine.verify_block_compute(flat["commitments"], flat["positions"][4:], flat["publickeys"][b], flat["shares"][b],
                                flat["responses"][b], flat["challenge"])
    for _ in range(capi.BLOCK_SLOTS - 2):      # MPVSS_BLOCK_SLOTS blocks may be in flight ...
        engine.verify_block_compute(flat["commitments"], flat["positions"][:4], flat["publickeys"][a], flat["shares"][a],
                                    flat["responses"][a], flat["challenge"])
    with pytest.raises(capi.EngineError):      # ... one more is refused until one is absorbed
        engine.verify_block_compute(flat["commitments"], flat["positions"][:4], flat["publickeys"][a], flat["shares"][a],
                                    flat["responses"][a], flat["challenge"])
    st = engine.verify_block_absorb(capi.transcript_init())
    st = engine.verify_block_absorb(st)
    for _ in range(capi.BLOCK_SLOTS - 2):
        engine.verify_block_absorb(capi.transcript_init())
    verdict, digest = capi.transcript_verdict(st, flat["challenge"])
    assert verdict is True and digest == box["_digest"]
    with pytest.raises(capi.EngineError):
        engine.verify_block_absorb(st)         # nothing left in flight


def test_concurrent_host_threads_share_one_context(engine):
    rng = random.Random(3)
    q = O.ModpGroup().q
    jobs = []
    for k in range(4):
        a = [rng.randrange(q) for _ in range(40)]
        e = [rng.randrange(1 << 200) for _ in range(40)]
        jobs.append((a, e))
    results = [None] * 4

    def work(i):
        a, e = jobs[i]
        results[i] = split(engine.batch_exp(b"".join(x.to_bytes(EB, "big") for x in a),
                                            b"".join(x.to_bytes(EB, "big") for x in e)))
    th = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    [t.start() for t in th]
    [t.join() for t in th]
    for (a, e), r in zip(jobs, results):
        assert r == [pow(x, y, q) for x, y in zip(a, e)]


def test_multi_chunk_path_in_a_subprocess():
    """MPVSS_MAX_CHUNK=16 makes a 45-share box run as three chunks through every chunked entry point."""
    code = r'''
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)
import mpvss_oracle as O
from helpers import make_modp_instance, cat, split
from mpvss_rs_amd import Engine
eng = Engine(0)
g, privs, pks, coeffs, ws, box = make_modp_instance(45, 3, 9)
flat = O.box_to_flat(g, box)
res = eng.verify_distribution(flat["commitments"], flat["positions"], flat["publickeys"], flat["shares"],
                              flat["responses"], flat["challenge"], dump=True)
assert res["verdict"] is True and res["digest"] == box["_digest"]
assert split(res["X"]) == box["_X"] and split(res["a1"]) == box["_a1"] and split(res["a2"]) == box["_a2"]
order = g.group_order_int()
pv = [O.poly_get_value(coeffs, i) %% order for i in flat["positions"]]
d = eng.distribute(flat["commitments"], flat["positions"], flat["publickeys"], cat(g, pv), cat(g, ws))
assert d["digest"] == box["_digest"]
out = split(eng.batch_exp(flat["publickeys"], flat["responses"]))
assert out == [pow(y, r, g.q) for y, r in zip(pks, split(flat["responses"]))]
print("chunked ok")
''' % (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"))
    env = dict(os.environ, MPVSS_MAX_CHUNK="16")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr
    assert "chunked ok" in out.stdout


def test_caller_provided_stream(engine):
    import torch
    from mpvss_rs_amd import Engine
    eng = Engine(0)
    s = torch.cuda.Stream()
    assert eng.lib.mpvss_ctx_set_stream(eng.ctx, C.c_void_p(s.cuda_stream)) == 0
    q = O.ModpGroup().q
    out = split(eng.batch_mul((5).to_bytes(EB, "big") * 3, (7).to_bytes(EB, "big") * 3))
    assert out == [35, 35, 35]
    assert eng.lib.mpvss_ctx_synchronize(eng.ctx) == 0
    eng.close()


def test_blocks_absorbed_by_several_host_threads(engine):
    """Several host threads may absorb consecutive blocks side by side (the engine releases its lock while it waits
    and hashes): every block must come out with its own, correct transcript."""
    import concurrent.futures
    g, privs, pks, coeffs, ws, box = make_modp_instance(12, 4, 21)
    flat = O.box_to_flat(g, box)
    tampered = bytearray(flat["responses"]); tampered[700] ^= 4
    kinds = [flat["responses"], bytes(tampered)] * 4                 # 8 blocks, alternately honest and tampered
    for resp in kinds:
        engine.verify_block_compute(flat["commitments"], flat["positions"], flat["publickeys"], flat["shares"], resp,
                                    flat["challenge"])

    def absorb(_):
        st = engine.verify_block_absorb(capi.transcript_init())
        return capi.transcript_verdict(st, flat["challenge"])

    with concurrent.futures.ThreadPoolExecutor(max_workers=4) as pool:
        out = list(pool.map(absorb, range(len(kinds))))
    # blocks are handed out in FIFO order at entry, but the threads enter in any order: compare as multisets
    assert sorted(v for v, _ in out) == [False] * 4 + [True] * 4
    assert all(d == box["_digest"] for v, d in out if v)
    assert all(d != box["_digest"] for v, d in out if not v)
    with pytest.raises(capi.EngineError):
        engine.verify_block_absorb(capi.transcript_init())


def test_claimed_blocks_are_absorbed_by_ticket(engine):
    """mpvss_block_claim / mpvss_modp_verify_block_absorb_claimed: a thread learns WHICH block it holds before it has to
    provide the transcript state (a rank of a sharded verification fetches that state from the previous rank in
    between).  Tickets count the blocks in enqueue order; claimed blocks may be absorbed in any order and from several
    threads; a block cut in two carries its state from the first ticket to the second."""
    import concurrent.futures
    g, privs, pks, coeffs, ws, box = make_modp_instance(12, 4, 22)
    flat = O.box_to_flat(g, box)
    tampered = bytearray(flat["shares"]); tampered[300] ^= 1
    kinds = [flat["shares"], bytes(tampered), flat["shares"], flat["shares"], bytes(tampered)]
    for sh in kinds:
        engine.verify_block_compute(flat["commitments"], flat["positions"], flat["publickeys"], sh, flat["responses"],
                                    flat["challenge"])
    tickets = [engine.block_claim() for _ in kinds]
    assert tickets == list(range(tickets[0], tickets[0] + len(kinds)))
    with pytest.raises(capi.EngineError):
        engine.block_claim()                                             # nothing unclaimed is left
    with pytest.raises(capi.EngineError):
        engine.verify_block_absorb(capi.transcript_init())               # ... for the one-step call either
    with pytest.raises(capi.EngineError):
        engine.verify_block_absorb_claimed(tickets[-1] + 1, capi.transcript_init())

    def absorb(i):
        st = engine.verify_block_absorb_claimed(tickets[i], capi.transcript_init())
        return capi.transcript_verdict(st, flat["challenge"])

    with concurrent.futures.ThreadPoolExecutor(max_workers=3) as pool:
        out = list(pool.map(absorb, reversed(range(len(kinds)))))[::-1]
    assert [v for v, _ in out] == [True, False, True, True, False]       # by ticket, not by arrival
    assert all((d == box["_digest"]) == v for v, d in out)
    with pytest.raises(capi.EngineError):
        engine.verify_block_absorb_claimed(tickets[0], capi.transcript_init())      # a ticket is good once
    # one box as two blocks (5 + 7 shares), claimed together, state carried from the first to the second
    a = slice(0, 5 * EB); b = slice(5 * EB, 12 * EB)
    for sl, pos in ((a, flat["positions"][:5]), (b, flat["positions"][5:])):
        engine.verify_block_compute(flat["commitments"], pos, flat["publickeys"][sl], flat["shares"][sl],
                                    flat["responses"][sl], flat["challenge"])
    t1, t2 = engine.block_claim(), engine.block_claim()
    st = engine.verify_block_absorb_claimed(t1, capi.transcript_init())
    st = engine.verify_block_absorb_claimed(t2, st)
    assert capi.transcript_verdict(st, flat["challenge"]) == (True, box["_digest"])
    assert engine.blocks_in_flight() == (0, 0)


def test_verify_many_pipelines_boxes_inside_the_library(engine):
    """mpvss_modp_verify_many == one verify_distribution per box, in box order: honest, tampered, empty and
    different-sized boxes mixed; more boxes than block slots; a malformed box gets verdict False and the run goes on."""
    g, privs, pks, coeffs, ws, box = make_modp_instance(12, 4, 21)
    flat = O.box_to_flat(g, box)
    g2, _, _, _, _, box2 = make_modp_instance(5, 3, 22)
    flat2 = O.box_to_flat(g2, box2)
    as_box = lambda f, **kw: dict({"commitments": f["commitments"], "positions": f["positions"], "pubkeys": f["publickeys"],
                                   "shares": f["shares"], "responses": f["responses"], "challenge": f["challenge"]}, **kw)
    tampered = bytearray(flat["responses"]); tampered[700] ^= 4
    empty = as_box(flat, positions=[], pubkeys=b"", shares=b"", responses=b"")
    boxes = ([as_box(flat), as_box(flat, responses=bytes(tampered)), as_box(flat2), empty] * 6)[:21]
    want = []
    for b in boxes:
        r = engine.verify_distribution(b["commitments"], b["positions"], b["pubkeys"], b["shares"], b["responses"],
                                       b["challenge"])
        want.append((r["verdict"], r["digest"]))
    assert want[0] == (True, box["_digest"]) and want[1][0] is False and want[2] == (True, box2["_digest"])
    engine.pipeline_stats(reset=True)
    for depth, threads in ((1, 1), (3, 2), (16, 8)):
        assert engine.verify_many(boxes, depth=depth, hash_threads=threads) == want
    st = engine.pipeline_stats()
    assert st["blocks"] == 3 * sum(1 for b in boxes if b["positions"])
    assert st["hash_ms"] > 0 and st["enqueue_ms"] > 0
    bad = as_box(flat, positions=[-1] + flat["positions"][1:])
    # a malformed box (the reference would panic on its negative exponent) costs only itself: verdict False, zero digest
    assert engine.verify_many([bad] + [as_box(flat)] * 3 + [bad] + [as_box(flat)] * 3 + [bad], depth=4, hash_threads=2) == \
        [(False, bytes(32))] + [want[0]] * 3 + [(False, bytes(32))] + [want[0]] * 3 + [(False, bytes(32))]
    assert "negative position" in engine.last_error()
    # the engine is usable afterwards (no slot left busy)
    assert engine.verify_many([as_box(flat)], depth=1, hash_threads=1) == [want[0]]


def test_block_wellformedness_bytes(engine):
    """mpvss_modp_verify_block_compute_flags: one byte per share in device memory, 1 iff y_i, Y_i, r_i are canonical
    encodings (0 < y, Y < q, r < q - 1); the box verdict and digest are those of the plain call (the reference validates
    nothing here, src/groups/modp.rs:154-156) -- an honest box gives all ones."""
    import torch
    g, privs, pks, coeffs, ws, box = make_modp_instance(40, 3, 31)
    flat = O.box_to_flat(g, box)
    n, q = 40, g.q
    dev = torch.device("cuda", 0)

    def run(pk, sh, rs):
        flags = torch.full((n,), 7, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()      # torch's fill runs on torch's stream, the block's kernels on the slot's
        engine.verify_block_compute_flags(flat["commitments"], flat["positions"], pk, sh, rs, flat["challenge"], flags.data_ptr())
        st = engine.verify_block_absorb(capi.transcript_init())
        return list(flags.cpu().numpy()), capi.transcript_verdict(st, flat["challenge"])

    good, verdict = run(flat["publickeys"], flat["shares"], flat["responses"])
    assert good == [1] * n and verdict == (True, box["_digest"])
    pk, sh, rs = bytearray(flat["publickeys"]), bytearray(flat["shares"]), bytearray(flat["responses"])
    fx = lambda v: v.to_bytes(256, "big")
    pk[0:256] = fx(0); pk[256:512] = fx(q); pk[512:768] = fx(q - 1)               # 0: no, q: no, q-1: yes
    sh[3 * 256:4 * 256] = fx(q + 1); sh[4 * 256:5 * 256] = fx(2**2048 - 1); sh[5 * 256:6 * 256] = fx(1)   # no, no, yes
    rs[6 * 256:7 * 256] = fx(q - 1); rs[7 * 256:8 * 256] = fx(q - 2); rs[8 * 256:9 * 256] = fx(0)       # no, yes, yes
    rs[39 * 256:40 * 256] = fx(q)
    got, verdict = run(bytes(pk), bytes(sh), bytes(rs))
    assert got == [0, 0, 1, 0, 0, 1, 0, 1, 1] + [1] * 30 + [0]
    ref = engine.verify_distribution(flat["commitments"], flat["positions"], bytes(pk), bytes(sh), bytes(rs), flat["challenge"])
    assert verdict == (ref["verdict"], ref["digest"]) and verdict[0] is False
