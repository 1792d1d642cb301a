"""GPU parity tests for the secp256k1 and ristretto255 groups: engine (C ABI) vs oracle and vs golden fixtures."""
import glob
import json
import os
import random

import pytest

import mpvss_oracle as O
from mpvss_rs_amd import capi

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
GID = {"secp256k1": capi.GROUP_SECP256K1, "ristretto255": capi.GROUP_RISTRETTO255}


def mk(name):
    return O.GROUPS[name](), GID[name]


def enc(G, pts):
    return b"".join(G.element_to_bytes(p) for p in pts)


def sc(G, ks):
    return b"".join(G.scalar_to_bytes(k) for k in ks)


def split(b, n):
    return [b[i:i + n] for i in range(0, len(b), n)]


@pytest.mark.parametrize("name", ["secp256k1", "ristretto255"])
def test_ec_exp_and_mul(engine, name):
    G, gid = mk(name)
    rng = random.Random(41)
    order = G.group_order_int()
    B = G.generator()
    pts = [G.exp(B, rng.randrange(1, order)) for _ in range(70)] + [G.identity(), B]
    ks = [rng.randrange(order) for _ in range(68)] + [0, 1, order - 1, 2]
    out = engine.ec_batch_exp(gid, enc(G, pts), sc(G, ks))
    assert split(out, G.elem_len) == [G.element_to_bytes(G.exp(p, k)) for p, k in zip(pts, ks)]
    qs = pts[1:] + pts[:1]
    qs[5] = G.element_inverse(pts[5])     # P + (-P)
    qs[6] = pts[6]                        # P + P
    out = engine.ec_batch_mul(gid, enc(G, pts), enc(G, qs))
    assert split(out, G.elem_len) == [G.element_to_bytes(G.mul(p, q)) for p, q in zip(pts, qs)]


@pytest.mark.parametrize("name", ["secp256k1", "ristretto255"])
def test_ec_commit_eval_and_dleq(engine, name):
    G, gid = mk(name)
    rng = random.Random(42)
    order = G.group_order_int()
    B = G.generator()
    cm = [G.exp(B, rng.randrange(order)) for _ in range(5)]
    positions = list(range(1, 40)) + [0, 65536, 65535, (1 << 40) + 3]
    out = engine.ec_commit_eval(gid, enc(G, cm), positions)
    assert split(out, G.elem_len) == [G.element_to_bytes(O.commitment_eval(G, cm, i)) for i in positions]
    n = 33
    h1 = [G.exp(B, rng.randrange(order)) for _ in range(n)]
    g2 = [G.exp(B, rng.randrange(order)) for _ in range(n)]
    h2 = [G.exp(B, rng.randrange(order)) for _ in range(n)]
    r = [rng.randrange(order) for _ in range(n)]
    c = rng.randrange(order)
    a1, a2 = engine.ec_dleq_commitments(gid, G.element_to_bytes(B), enc(G, h1), enc(G, g2), enc(G, h2), sc(G, r),
                                        G.scalar_to_bytes(c), False)
    exp = [O.dleq_verifier_commitments(G, B, h1[i], g2[i], h2[i], r[i], c) for i in range(n)]
    assert split(a1, G.elem_len) == [G.element_to_bytes(e[0]) for e in exp]
    assert split(a2, G.elem_len) == [G.element_to_bytes(e[1]) for e in exp]


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(HERE, "golden", "secp256k1_*.json")) +
                                        glob.glob(os.path.join(HERE, "golden", "ristretto255_*.json"))),
                         ids=os.path.basename)
def test_ec_fixture_on_gpu(engine, path):
    fx = json.load(open(path))
    gid = GID[fx["group"]]
    b = fx["box"]
    cat = lambda hs: bytes.fromhex("".join(hs))
    res = engine.ec_verify_distribution(gid, cat(b["commitments"]), b["positions"], cat(b["publickeys"]), cat(b["shares"]),
                                        cat(b["responses"]), bytes.fromhex(b["challenge"]), dump=True)
    assert res["verdict"] is True
    assert res["digest"].hex() == fx["expected"]["transcript_digest"]
    assert res["X"].hex() == "".join(fx["expected"]["X"])
    assert res["a1"].hex() == "".join(fx["expected"]["a1"])
    assert res["a2"].hex() == "".join(fx["expected"]["a2"])
    for tam in fx["tampered"]:
        res = engine.ec_verify_distribution(gid, cat(tam["commitments"]), b["positions"], cat(b["publickeys"]),
                                            cat(tam["shares"]), cat(tam["responses"]), bytes.fromhex(tam["challenge"]))
        assert res["verdict"] is False
        assert res["digest"].hex() == tam["transcript_digest"]
    sb = fx["expected"]["share_boxes"]
    verdicts = engine.ec_verify_shares(gid, cat(b["publickeys"]), cat([s["share"] for s in sb]), cat(b["shares"]),
                                       cat([s["challenge"] for s in sb]), cat([s["response"] for s in sb]))
    assert list(verdicts) == [1] * fx["n"]
    # one tampered share box
    rs = [bytearray.fromhex(s["response"]) for s in sb]
    rs[1][7] ^= 1
    verdicts = engine.ec_verify_shares(gid, cat(b["publickeys"]), cat([s["share"] for s in sb]), cat(b["shares"]),
                                       cat([s["challenge"] for s in sb]), b"".join(bytes(x) for x in rs))
    assert list(verdicts) == [1, 0] + [1] * (fx["n"] - 2)
    # dealer side from the recorded randomness
    G = O.GROUPS[fx["group"]]()
    order = G.group_order_int()
    coeffs = [int(c, 16) for c in fx["inputs"]["coefficients"]]
    ws = [int(x, 16) for x in fx["inputs"]["witnesses"]]
    pvals = [G.scalar_from_bigint(O.poly_get_value(coeffs, i) % order) for i in b["positions"]]
    cm = engine.ec_batch_exp(gid, G.element_to_bytes(G.generator()) * fx["t"],
                             sc(G, [G.scalar_from_bigint(c) for c in coeffs]))
    assert cm.hex() == "".join(b["commitments"])
    d = engine.ec_distribute(gid, cm, b["positions"], cat(b["publickeys"]), sc(G, pvals), sc(G, ws))
    assert d["X"].hex() == "".join(fx["expected"]["X"]) and d["Y"].hex() == "".join(b["shares"])
    assert d["a1"].hex() == "".join(fx["expected"]["a1"]) and d["a2"].hex() == "".join(fx["expected"]["a2"])
    assert d["digest"].hex() == fx["expected"]["transcript_digest"]


@pytest.mark.parametrize("name", ["secp256k1", "ristretto255"])
def test_ec_rejects_invalid_encodings_and_scalars(engine, name):
    G, gid = mk(name)
    good = G.element_to_bytes(G.generator())
    bad = (b"\x05" + bytes(32)) if name == "secp256k1" else bytes.fromhex("01" + "00" * 31)
    with pytest.raises(capi.EngineError):
        engine.ec_batch_mul(gid, good + bad, good + good)
    with pytest.raises(capi.EngineError):
        engine.ec_batch_exp(gid, good, G.scalar_to_bytes(G.group_order_int()) if name == "secp256k1"
                            else G.group_order_int().to_bytes(32, "little"))


@pytest.mark.parametrize("name", ["secp256k1", "ristretto255"])
def test_ec_fixed_base_comb_and_window_digits(engine, name):
    """k * G through the fixed-base comb (mpvss_ec_batch_exp_generator) and k * P through the signed-window tables,
    with scalars whose digits hit the window boundaries (-8, +8, carries into the 65th window)."""
    G, gid = mk(name)
    rng = random.Random(43)
    order = G.group_order_int()
    special = [0, 1, 2, 7, 8, 9, 15, 16, 0x88888888, 0x77777777, (0x8 << 252) % order, order - 1, order - 2,
               int("8" * 62, 16), int("7" * 63, 16) % order, int("f" * 63, 16) % order]
    ks = special + [rng.randrange(order) for _ in range(70)]
    out = engine.ec_batch_exp_generator(gid, sc(G, ks))
    assert split(out, G.elem_len) == [G.element_to_bytes(G.exp(G.generator(), k)) for k in ks]
    P = G.exp(G.generator(), rng.randrange(1, order))
    out = engine.ec_batch_exp(gid, enc(G, [P] * len(ks)), sc(G, ks))
    assert split(out, G.elem_len) == [G.element_to_bytes(G.exp(P, k)) for k in ks]
    # generic (non-generator) shared base g1 in dleq_commitments
    n = 20
    h1 = [G.exp(G.generator(), rng.randrange(order)) for _ in range(n)]
    g2 = [G.exp(G.generator(), rng.randrange(order)) for _ in range(n)]
    h2 = [G.exp(G.generator(), rng.randrange(order)) for _ in range(n)]
    h2[3] = G.identity()
    g2[4] = G.identity()
    r = [rng.randrange(order) for _ in range(n)]
    cs = [rng.randrange(order) for _ in range(n)]
    for g1 in (P, G.generator()):
        a1, a2 = engine.ec_dleq_commitments(gid, G.element_to_bytes(g1), enc(G, h1), enc(G, g2), enc(G, h2), sc(G, r), sc(G, cs), True)
        exp = [O.dleq_verifier_commitments(G, g1, h1[i], g2[i], h2[i], r[i], cs[i]) for i in range(n)]
        assert split(a1, G.elem_len) == [G.element_to_bytes(e[0]) for e in exp]
        assert split(a2, G.elem_len) == [G.element_to_bytes(e[1]) for e in exp]


@pytest.mark.parametrize("name", ["secp256k1", "ristretto255"])
def test_ec_block_api_and_verify_many(engine, name):
    """Several curve boxes in flight inside one context: compute, compute, absorb, absorb; the library-pipelined
    mpvss_ec_verify_many; an invalid encoding is reported by the absorbing call; negative positions wrap (u64) as
    in the reference instead of failing."""
    fx = json.load(open(os.path.join(HERE, "golden", f"{name}_n8_t4.json")))
    gid = GID[name]
    b = fx["box"]
    cat = lambda hs: bytes.fromhex("".join(hs))
    good = {"commitments": cat(b["commitments"]), "positions": b["positions"], "pubkeys": cat(b["publickeys"]),
            "shares": cat(b["shares"]), "responses": cat(b["responses"]), "challenge": bytes.fromhex(b["challenge"])}
    tam = fx["tampered"][0]
    bad = dict(good, commitments=cat(tam["commitments"]), shares=cat(tam["shares"]), responses=cat(tam["responses"]),
               challenge=bytes.fromhex(tam["challenge"]))
    want_good = (True, bytes.fromhex(fx["expected"]["transcript_digest"]))
    want_bad = (False, bytes.fromhex(tam["transcript_digest"]))
    order_args = lambda x: (x["commitments"], x["positions"], x["pubkeys"], x["shares"], x["responses"], x["challenge"])
    for x in (good, bad, good):
        engine.ec_verify_block_compute(gid, *order_args(x))
    got = []
    for x in (good, bad, good):
        st = engine.ec_verify_block_absorb(capi.transcript_init())
        got.append(capi.ec_transcript_verdict(gid, st, x["challenge"]))
    assert got == [want_good, want_bad, want_good]
    boxes = [good, bad] * 9 + [good]
    for depth, threads in ((1, 1), (4, 2), (16, 4)):
        assert engine.ec_verify_many(gid, boxes, depth=depth, hash_threads=threads) == [want_good, want_bad] * 9 + [want_good]
    G = O.GROUPS[name]()
    L = G.elem_len
    broken = bytearray(good["pubkeys"])
    broken[2 * L:3 * L] = (b"\x05" + bytes(32)) if name == "secp256k1" else bytes.fromhex("01" + "00" * 31)
    # a malformed box (an encoding that is no group element, a response that is not reduced) is that box's business: verdict
    # False with a zero digest -- the reference answers `false` to structural problems (participant.rs:415-420) -- while
    # the other boxes of the run are verified as usual; mpvss_last_error names the reason
    malformed = (False, bytes(32))
    assert engine.ec_verify_many(gid, [good, dict(good, pubkeys=bytes(broken)), good, bad], depth=3, hash_threads=2) == \
        [want_good, malformed, want_good, want_bad]
    assert "public keys: element 2" in engine.last_error()
    big = G.group_order_int().to_bytes(32, "big" if name == "secp256k1" else "little")
    assert engine.ec_verify_many(gid, [dict(good, responses=good["responses"][:32] + big + good["responses"][64:]), good]) == \
        [malformed, want_good]
    assert "responses: scalar 1" in engine.last_error()
    # the single-box entry points still report such a box as an error
    with pytest.raises(capi.EngineError, match="public keys: element 2"):
        engine.ec_verify_distribution(gid, good["commitments"], good["positions"], bytes(broken), good["shares"], good["responses"],
                                      good["challenge"])
    assert engine.ec_verify_many(gid, [good]) == [want_good]
    # Scalar::from(position as u64): position -3 is the scalar 2^64 - 3 (participant.rs:1419, 1862)
    cm = [G.element_from_fixed(good["commitments"][i * L:(i + 1) * L]) for i in range(fx["t"])]
    xs = engine.ec_commit_eval(gid, good["commitments"], [-3, 5])
    assert xs[:L] == G.element_to_bytes(O.commitment_eval(G, cm, (1 << 64) - 3))
    assert xs[L:] == G.element_to_bytes(O.commitment_eval(G, cm, 5))


@pytest.mark.parametrize("name", ["secp256k1", "ristretto255"])
def test_ec_device_resident_positions_pick_the_path_on_the_device(engine, name):
    """Inputs in HBM: whether the forward-difference path applies is decided by a kernel (consecutive positions) and a
    device flag gates the two X paths; both answers must equal the host-buffer call."""
    import ctypes as C

    import torch
    G, gid = mk(name)
    rng = random.Random(44)
    order = G.group_order_int()
    L = G.elem_len
    t, n = 16, 4200
    gen = G.element_to_bytes(G.generator())
    cm = engine.ec_batch_exp_generator(gid, sc(G, [rng.randrange(order) for _ in range(t)]))
    dev = torch.device("cuda", 0)
    d_cm = torch.frombuffer(bytearray(cm), dtype=torch.uint8).to(dev)
    for positions in (list(range(7, 7 + n)), list(range(7, 7 + n - 1)) + [99999]):
        want = engine.ec_commit_eval(gid, cm, positions)
        d_pos = torch.tensor(positions, dtype=torch.int64, device=dev)
        d_out = torch.zeros(n * L, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()          # torch's fill runs on torch's stream, the engine's kernels on the engine's
        rc = engine.lib.mpvss_ec_commit_eval(engine.ctx, gid, capi.MPVSS_DEVICE, C.c_void_p(d_cm.data_ptr()), t,
                                             C.c_void_p(d_pos.data_ptr()), n, C.c_void_p(d_out.data_ptr()))
        engine._check(rc, "ec_commit_eval(device)")
        engine.lib.mpvss_ctx_synchronize(engine.ctx)
        assert bytes(d_out.cpu().numpy().tobytes()) == want
    i = n - 1
    cmp = [G.element_from_fixed(cm[k * L:(k + 1) * L]) for k in range(t)]
    assert want[i * L:(i + 1) * L] == G.element_to_bytes(O.commitment_eval(G, cmp, 99999))
    assert len(gen) == L


@pytest.mark.parametrize("name", ["secp256k1", "ristretto255"])
def test_ec_verify_many_batches_the_x_paths_of_several_boxes(engine, name):
    """mpvss_ec_verify_many computes the X paths of runs of consecutive boxes of one shape (up to MPVSS_EC_X_BATCH) by the
    same launches (box = second grid dimension).  Ten boxes from different dealers, one with a tampered share, one whose
    positions are not consecutive (host buffers: it ends the run and takes the per-box path; device buffers: it stays in
    the batch and falls back to Horner's rule through its own gate), one of another size near the end: every verdict
    and transcript digest must equal the one-box-at-a-time call's, which the other tests pin."""
    import ctypes as C

    import torch
    G, gid = mk(name)
    order = G.group_order_int()
    L = G.elem_len
    sb = G.scalar_to_fixed
    gen = G.element_to_bytes(G.generator())
    n, t = 4096 + 64, 16
    rng = random.Random(0xBA7C + gid)
    privs = [rng.randrange(1, order) for _ in range(n)]
    pks = engine.ec_batch_exp(gid, gen * n, b"".join(map(sb, privs)))

    def deal(positions, m=n):
        coeffs = [rng.randrange(order) for _ in range(t)]
        wits = [rng.randrange(1, order) for _ in range(m)]
        pvals = [sum(c * pow(p % order, j, order) for j, c in enumerate(coeffs)) % order for p in positions]
        cm = engine.ec_batch_exp(gid, gen * t, b"".join(map(sb, coeffs)))
        d = engine.ec_distribute(gid, cm, positions, pks[:m * L], b"".join(map(sb, pvals)), b"".join(map(sb, wits)))
        c = G.hash_to_scalar(d["digest"])
        resp = b"".join(sb((w - p * c) % order) for w, p in zip(wits, pvals))
        return {"commitments": cm, "positions": positions, "pubkeys": pks[:m * L], "shares": d["Y"], "responses": resp,
                "challenge": sb(c)}, d["digest"]

    consecutive = list(range(1, n + 1))
    boxes, digests = zip(*[deal(consecutive) for _ in range(5)])
    boxes, digests = list(boxes), list(digests)
    tampered = dict(boxes[1])
    y = bytearray(tampered["shares"]); y[7 * L:8 * L] = pks[9 * L:10 * L]; tampered["shares"] = bytes(y)
    scattered, dg_scattered = deal(list(range(1, n)) + [3 * n])                 # last position out of line
    small, dg_small = deal(list(range(5, 5 + 4096)), 4096)                      # another shape inside a batch
    # host: a batch of 5, the out-of-line box alone, a batch of 2, two single boxes; device: a batch of 8, two single boxes
    seq = [boxes[0], tampered, boxes[2], boxes[3], boxes[4], scattered, boxes[0], boxes[2], small, boxes[3]]
    one_by_one = []
    for b in seq:
        r = engine.ec_verify_distribution(gid, b["commitments"], b["positions"], b["pubkeys"], b["shares"], b["responses"],
                                          b["challenge"])
        one_by_one.append((r["verdict"], r["digest"]))
    assert [v for v, _ in one_by_one] == [True, False] + [True] * 8
    assert one_by_one[0][1] == digests[0] and one_by_one[5][1] == dg_scattered and one_by_one[8][1] == dg_small
    for depth, threads in ((2, 1), (8, 3)):
        assert engine.ec_verify_many(gid, seq, depth=depth, hash_threads=threads) == one_by_one
    # two boxes with nothing else in flight: one batch of two through the quad-lane stage pipelines (box = second grid dimension
    # of the stepping / table launches, hand-over space and tickets per box)
    assert engine.ec_verify_many(gid, seq[:2], depth=2, hash_threads=1) == one_by_one[:2]
    assert engine.ec_verify_many(gid, [boxes[3], boxes[4]], depth=1, hash_threads=1) == [one_by_one[3], one_by_one[4]]
    # a commitment that is no group element, inside a batch: reported by the box it belongs to, as on the per-box path
    bad_cm = bytearray(boxes[2]["commitments"])
    bad_cm[3 * L:4 * L] = (b"\x05" + bytes(32)) if name == "secp256k1" else bytes.fromhex("01" + "00" * 31)
    assert engine.ec_verify_many(gid, [boxes[0], dict(boxes[2], commitments=bytes(bad_cm)), boxes[3]], depth=3, hash_threads=2) == \
        [one_by_one[0], (False, bytes(32)), one_by_one[3]]
    assert "commitments: element 3" in engine.last_error()
    assert engine.ec_verify_many(gid, seq[:3], depth=3, hash_threads=2) == one_by_one[:3]      # the context is usable again
    # the same boxes resident in HBM: positions are judged on the device, box by box
    dev = torch.device("cuda", 0)
    keep = []

    def dptr(b):
        tns = torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
        keep.append(tns)
        return tns.data_ptr()

    arr = (capi.EcBox * len(seq))()
    for i, b in enumerate(seq):
        pos = torch.tensor(b["positions"], dtype=torch.int64, device=dev)
        ch = (C.c_uint8 * 32).from_buffer_copy(b["challenge"])
        keep += [pos, ch]
        arr[i] = capi.EcBox(dptr(b["commitments"]), t, pos.data_ptr(), dptr(b["pubkeys"]), dptr(b["shares"]),
                            dptr(b["responses"]), len(b["positions"]), C.cast(ch, C.c_void_p))
    verdicts = (C.c_int * len(seq))()
    out = (C.c_uint8 * (32 * len(seq)))()
    engine._check(engine.lib.mpvss_ec_verify_many(engine.ctx, gid, capi.MPVSS_DEVICE, arr, len(seq), 6, 2, verdicts,
                                                  C.cast(out, C.c_void_p)), "ec_verify_many(device)")
    raw = bytes(out)
    assert [(bool(verdicts[i]), raw[32 * i:32 * i + 32]) for i in range(len(seq))] == one_by_one


@pytest.mark.parametrize("name", ["secp256k1", "ristretto255"])
def test_ec_dealer_blocks(engine, name):
    """Dealer side of distribute_secret for the curve groups in compute / absorb form (participant.rs:1094-1274,
    1573-1717): several blocks in flight; X from the commitments or as P(i) * G through the comb (commitments None);
    outputs and transcript digest equal to the synchronous call and to the oracle's box; one box as two blocks carries
    the hash state; a witness that is no scalar is reported by the absorbing call and the context stays usable."""
    fx = json.load(open(os.path.join(HERE, "golden", f"{name}_n8_t4.json")))
    G, gid = mk(name)
    order = G.group_order_int()
    L = G.elem_len
    sb = G.scalar_to_fixed
    coeffs = [int(c, 16) for c in fx["inputs"]["coefficients"]]
    wits = [int(x, 16) for x in fx["inputs"]["witnesses"]]
    b = fx["box"]
    cat = lambda hs: bytes.fromhex("".join(hs))
    cm, pks, positions = cat(b["commitments"]), cat(b["publickeys"]), b["positions"]
    n = len(positions)
    pvals = [sum(c * pow(p, j, order) for j, c in enumerate(coeffs)) % order for p in positions]
    pv, wt = b"".join(map(sb, pvals)), b"".join(map(sb, wits))
    want = engine.ec_distribute(gid, cm, positions, pks, pv, wt)
    assert want["Y"] == cat(b["shares"]) and want["digest"] == bytes.fromhex(fx["expected"]["transcript_digest"])
    engine.ec_distribute_compute(gid, cm, positions, pks, pv, wt)
    engine.ec_distribute_compute(gid, None, None, pks, pv, wt)
    for _ in range(2):
        st, X, Y, a1, a2 = engine.ec_distribute_absorb(gid, capi.transcript_init(), n)
        assert (X, Y, a1, a2) == (want["X"], want["Y"], want["a1"], want["a2"])
        assert capi.ec_transcript_verdict(gid, st, bytes(32))[1] == want["digest"]
    cut = 3
    engine.ec_distribute_compute(gid, cm, positions[:cut], pks[:cut * L], pv[:cut * 32], wt[:cut * 32])
    engine.ec_distribute_compute(gid, None, None, pks[cut * L:], pv[cut * 32:], wt[cut * 32:])
    st = capi.transcript_init()
    st, X1, Y1, _, _ = engine.ec_distribute_absorb(gid, st, cut)
    st, X2, Y2, _, _ = engine.ec_distribute_absorb(gid, st, n - cut)
    assert X1 + X2 == want["X"] and Y1 + Y2 == want["Y"]
    assert capi.ec_transcript_verdict(gid, st, bytes(32))[1] == want["digest"]
    big = order.to_bytes(32, "big" if name == "secp256k1" else "little")
    engine.ec_distribute_compute(gid, cm, positions, pks, pv, wt[:64] + big + wt[96:])
    engine.ec_distribute_compute(gid, cm, positions, pks, pv, wt)
    with pytest.raises(capi.EngineError, match="witnesses: scalar 2"):
        engine.ec_distribute_absorb(gid, capi.transcript_init(), n)
    st, X, _, _, _ = engine.ec_distribute_absorb(gid, capi.transcript_init(), n)
    assert X == want["X"]
    with pytest.raises(capi.EngineError, match="threshold"):
        engine.ec_distribute(gid, cm, positions[:2], pks[:2 * L], pv[:64], wt[:64])              # a whole box with t = 4 > n = 2
