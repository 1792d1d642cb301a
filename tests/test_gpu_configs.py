"""BASELINE.json configs at their full sizes on the GPU: C2 (MODP n=4096 t=64, every share against the C port),
the headline shape (MODP n=65536 t=256, a seeded 1 % sample), C3 / C4 (secp256k1 / ristretto255 n=65536 t=256) and
one GPU's slice of C5 (MODP n=131072 t=1024).  Size-independent properties: a box produced by the engine's dealer
path must verify with the dealer's transcript digest; one flipped bit anywhere must be rejected; sampled shares must
equal the oracle evaluated in the REFERENCE operation order (src/participant.rs:399-455, 1384-1442, 1827-1885)."""
import concurrent.futures
import hashlib
import math
import random

import pytest

import mpvss_oracle as O
from helpers import (EB, MODP_ORDER as ORDER, MODP_Q as Q, ec_reference_share, ec_reference_x, modp_fast_share, parallel_map,
                     poly_values, worker_count)
from mpvss_rs_amd import capi

pytestmark = pytest.mark.gpu


def fx(v):
    return v.to_bytes(EB, "big")


def keygen(rng):
    while True:                                   # modp.rs:162-174
        k = rng.randrange(Q)
        if math.gcd(k, ORDER) == 1:
            return k


_BOXES = {}


def make_modp_box(engine, n, t, seed):
    """one dealer's box of the given shape (cached per session: the headline-shape box is shared by the tests that need one)"""
    key = (n, t, seed)
    if key not in _BOXES:
        _BOXES[key] = _make_modp_box(engine, n, t, seed)
    return _BOXES[key]


def _make_modp_box(engine, n, t, seed):
    rng = random.Random(seed)
    coeffs = [rng.randrange(ORDER) for _ in range(t)]
    privs = [keygen(rng) for _ in range(n)]
    wits = [keygen(rng) for _ in range(n)]
    positions = list(range(1, n + 1))
    pvals = poly_values(coeffs, positions, ORDER)
    cm = engine.batch_exp_fixed_base(fx(4), b"".join(map(fx, coeffs)))
    pks = engine.batch_exp_fixed_base(fx(2), b"".join(map(fx, privs)))
    d = engine.distribute(cm, positions, pks, b"".join(map(fx, pvals)), b"".join(map(fx, wits)))
    c = int.from_bytes(hashlib.sha256(d["digest"]).digest(), "big") % ((Q - 1) // 2)
    responses = b"".join(fx((w - p * c) % ORDER) for w, p in zip(wits, pvals))
    return {"cm": cm, "pos": positions, "pk": pks, "Y": d["Y"], "r": responses, "c": fx(c), "d": d}


def check_modp_box(engine, box, sample_idx, fast_idx=()):
    """round trip, three tampers, and the C port (reference operation sequence) on the sampled shares"""
    n = len(box["pos"])
    res = engine.verify_distribution(box["cm"], box["pos"], box["pk"], box["Y"], box["r"], box["c"], dump=True)
    assert res["verdict"] is True
    assert res["digest"] == box["d"]["digest"]            # verifier transcript == dealer transcript
    assert res["X"] == box["d"]["X"] and res["a1"] == box["d"]["a1"] and res["a2"] == box["d"]["a2"]
    rng = random.Random(5)
    for field in ("r", "Y", "cm"):
        buf = bytearray(box[field])
        buf[rng.randrange(len(buf))] ^= 1 << rng.randrange(8)
        args = dict(box)
        args[field] = bytes(buf)
        bad = engine.verify_distribution(args["cm"], args["pos"], args["pk"], args["Y"], args["r"], args["c"])
        assert bad["verdict"] is False and bad["digest"] != box["d"]["digest"]
    from modp_ref import ModpRef
    ref = ModpRef()

    def work(i):
        s = slice(i * EB, (i + 1) * EB)
        return ref.share_work(box["cm"], box["pos"][i], box["pk"][s], box["Y"][s], box["r"][s], box["c"])

    with concurrent.futures.ThreadPoolExecutor(max_workers=worker_count(64)) as ex:     # ctypes releases the GIL
        outs = list(ex.map(work, sample_idx))
    for i, (x, a1, a2) in zip(sample_idx, outs):
        s = slice(i * EB, (i + 1) * EB)
        assert (x, a1, a2) == (res["X"][s], res["a1"][s], res["a2"][s]), f"share {i} of {n}"
    if fast_idx:
        # a full 1 % sample through the fast form of the same arithmetic (Horner in the exponent + CPython pow), on spawned
        # oracle-only processes
        def arg(i):
            s = slice(i * EB, (i + 1) * EB)
            return (box["cm"], box["pos"][i], box["pk"][s], box["Y"][s], box["r"][s], box["c"])
        outs = parallel_map(modp_fast_share, [arg(i) for i in fast_idx])
        for i, (x, a1, a2) in zip(fast_idx, outs):
            s = slice(i * EB, (i + 1) * EB)
            assert (x, a1, a2) == (res["X"][s], res["a1"][s], res["a2"][s]), f"share {i} of {n} (fast form)"


def test_c2_every_share_against_the_c_port(engine):
    """BASELINE config C2: n=4096, t=64 -- all 4096 shares' X, a1, a2 against oracle/modp_ref.c."""
    box = make_modp_box(engine, 4096, 64, seed=4096 + 64)
    check_modp_box(engine, box, list(range(4096)))


def test_c2_boxes_grouped_into_one_block(engine):
    """mpvss_modp_verify_many enqueues runs of consecutive small boxes of one shape as ONE block (every launch covers all of
    them: the box is the second grid dimension of the forward-difference kernels, one challenge per box spread to one per
    share, fixed windows of c instead of the lone box's sliding schedule, one transcript per box at the end).  21 boxes of
    BASELINE config C2's shape with other shapes in between -- three dealers' boxes, one of them tampered in a response, one
    in a share, a (4100, 64) box that ends a run, a box whose challenge does not fit 256 bits, a box with two participants
    swapped (positions not consecutive: the whole group falls back to Horner's rule on the device) -- give exactly the
    verdicts and transcript digests of one verify_distribution per box; the honest ones are the dealer's digests."""
    b0, b1, b2 = (make_modp_box(engine, 4096, 64, 60 + i) for i in range(3))
    other = make_modp_box(engine, 4100, 64, 63)
    as_box = lambda b, **kw: dict({"commitments": b["cm"], "positions": b["pos"], "pubkeys": b["pk"], "shares": b["Y"],
                                   "responses": b["r"], "challenge": b["c"]}, **kw)
    rs = bytearray(b1["r"]); rs[4095 * EB + 255] ^= 1
    sh = bytearray(b2["Y"]); sh[2000 * EB + 17] ^= 0x40
    wide = (int.from_bytes(b0["c"], "big") + (1 << 300)).to_bytes(EB, "big")

    def swapped(b, i, j):
        out = as_box(b)
        out["positions"] = list(b["pos"]); out["positions"][i], out["positions"][j] = b["pos"][j], b["pos"][i]
        for key, src in (("pubkeys", b["pk"]), ("shares", b["Y"]), ("responses", b["r"])):
            buf = bytearray(src)
            buf[i * EB:(i + 1) * EB], buf[j * EB:(j + 1) * EB] = src[j * EB:(j + 1) * EB], src[i * EB:(i + 1) * EB]
            out[key] = bytes(buf)
        return out

    boxes = ([as_box(b0), as_box(b1), as_box(b1, responses=bytes(rs)), as_box(b2)] * 4 + [as_box(other), as_box(b0)] +
             [as_box(b2, shares=bytes(sh)), as_box(b0, challenge=wide), as_box(b1), swapped(b2, 5, 4000), as_box(b0)])
    one = lambda b: (lambda r: (r["verdict"], r["digest"]))(engine.verify_distribution(
        b["commitments"], b["positions"], b["pubkeys"], b["shares"], b["responses"], b["challenge"]))
    want = [one(b) for b in boxes[:4]] * 4 + [one(b) for b in boxes[16:]]
    assert want[0] == (True, b0["d"]["digest"]) and want[1] == (True, b1["d"]["digest"]) and want[3] == (True, b2["d"]["digest"])
    assert want[2][0] is False and want[16] == (True, other["d"]["digest"]) and want[18][0] is False and want[19][0] is False
    assert want[21][0] is False and want[21][1] != b2["d"]["digest"]       # every share verifies, the transcript order differs
    before = engine.fd_stats()
    for depth, threads in ((24, 6), (3, 2)):
        assert engine.verify_many(boxes, depth=depth, hash_threads=threads) == want
    assert engine.blocks_in_flight() == (0, 0)
    after = engine.fd_stats()
    assert after[1] - before[1] == 2              # the group with the swapped box fell back (once per run), no other block did


def test_grouped_boxes_with_a_large_threshold(engine):
    """A group whose threshold is large enough for the pair-layout stepping kernel (t >= 512: k_modp_fd_step_pair with the
    box as blockIdx.y): three (8192, 512) boxes of two dealers, one of them with a flipped commitment bit, against one
    verify_distribution per box and the dealers' digests."""
    b0, b1 = make_modp_box(engine, 8192, 512, 80), make_modp_box(engine, 8192, 512, 81)
    as_box = lambda b, **kw: dict({"commitments": b["cm"], "positions": b["pos"], "pubkeys": b["pk"], "shares": b["Y"],
                                   "responses": b["r"], "challenge": b["c"]}, **kw)
    cm = bytearray(b1["cm"]); cm[300 * EB + 200] ^= 2
    boxes = [as_box(b0), as_box(b1), as_box(b1, commitments=bytes(cm)), as_box(b0)]
    one = lambda b: (lambda r: (r["verdict"], r["digest"]))(engine.verify_distribution(
        b["commitments"], b["positions"], b["pubkeys"], b["shares"], b["responses"], b["challenge"]))
    want = [one(b) for b in boxes[:3]] + [None]
    want[3] = want[0]
    assert want[0] == (True, b0["d"]["digest"]) and want[1] == (True, b1["d"]["digest"]) and want[2][0] is False
    before = engine.fd_stats()
    assert engine.verify_many(boxes, depth=4, hash_threads=2) == want
    after = engine.fd_stats()
    assert after[1] == before[1]                 # no fallback: the forward-difference pipelines held


def test_grouped_boxes_in_device_memory_with_a_negative_position(engine):
    """Groups of boxes handed over in HBM (what bench.py's `configs.c2` times): positions are only looked at when the block
    is absorbed -- a negative one costs its own box (verdict False, zero digest; the reference would panic) and nothing
    else of the group."""
    import ctypes as C
    import torch
    dev = torch.device("cuda", 0)
    bs = [make_modp_box(engine, 4096, 64, 70 + i) for i in range(2)]
    u8 = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    pos_ok = torch.tensor(bs[0]["pos"], dtype=torch.int64, device=dev)
    pos_bad = pos_ok.clone(); pos_bad[4095] = -4096
    keep, arr = [], (capi.ModpBox * 5)()
    for i, (b, pos) in enumerate([(bs[0], pos_ok), (bs[1], pos_bad), (bs[1], pos_ok), (bs[0], pos_ok), (bs[1], pos_ok)]):
        t = [u8(b[k]) for k in ("cm", "pk", "Y", "r")]
        ch = (C.c_uint8 * EB).from_buffer_copy(b["c"])
        keep.append((t, ch))
        arr[i] = capi.ModpBox(t[0].data_ptr(), 64, pos.data_ptr(), t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(), 4096,
                              C.cast(ch, C.c_void_p), None, 0)
    torch.cuda.synchronize()
    verdicts, digests = (C.c_int * 5)(), (C.c_uint8 * 160)()
    engine._check(engine.lib.mpvss_modp_verify_many(engine.ctx, capi.MPVSS_DEVICE, arr, 5, 4, 2, verdicts, C.cast(digests, C.c_void_p)),
                  "verify_many")
    raw = bytes(digests)
    got = [(bool(verdicts[i]), raw[32 * i:32 * i + 32]) for i in range(5)]
    d0, d1 = bs[0]["d"]["digest"], bs[1]["d"]["digest"]
    assert got == [(True, d0), (False, bytes(32)), (True, d1), (True, d0), (True, d1)]
    assert "negative position" in engine.last_error()
    assert engine.blocks_in_flight() == (0, 0)


def test_headline_shape_seeded_one_percent_sample(engine):
    """n=65536, t=256 (the metric's shape): a seeded 1 % of the shares -- 200 of them and the first and last two against the C
    port in the reference operation order, 456 more through the fast form of the same arithmetic (helpers.modp_fast_share)."""
    n = 65536
    box = make_modp_box(engine, n, 256, seed=n + 256)
    pick = random.Random(99).sample(range(n), 656)
    idx = sorted(set(pick[:200]) | {0, 1, n - 2, n - 1})
    check_modp_box(engine, box, idx, sorted(pick[200:]))


def test_c5_slice_of_one_gpu(engine):
    """BASELINE config C5 is n=2^20, t=1024 over 8 GPUs: this is ONE GPU's block (131072 consecutive positions, the
    block of rank 3, so positions start at 393217) -- multi-GPU run itself is the driver's.  48 seeded shares against
    the C port in the reference order (~30 core-seconds per share at t=1024) and a seeded 1 % (1311 shares) against the
    fast form of the same arithmetic (helpers.modp_fast_share)."""
    n, t, lo = 131072, 1024, 3 * 131072
    rng = random.Random(1024 + n)
    coeffs = [rng.randrange(ORDER) for _ in range(t)]
    privs = [keygen(rng) for _ in range(n)]
    wits = [keygen(rng) for _ in range(n)]
    positions = list(range(lo + 1, lo + n + 1))
    pvals = poly_values(coeffs, positions, ORDER)
    cm = engine.batch_exp_fixed_base(fx(4), b"".join(map(fx, coeffs)))
    pks = engine.batch_exp_fixed_base(fx(2), b"".join(map(fx, privs)))
    d = engine.distribute(cm, positions, pks, b"".join(map(fx, pvals)), b"".join(map(fx, wits)))
    c = int.from_bytes(hashlib.sha256(d["digest"]).digest(), "big") % ((Q - 1) // 2)
    responses = b"".join(fx((w - p * c) % ORDER) for w, p in zip(wits, pvals))
    box = {"cm": cm, "pos": positions, "pk": pks, "Y": d["Y"], "r": responses, "c": fx(c), "d": d}
    idx = sorted(set(random.Random(7).sample(range(n), 22)) | {0, n - 1})
    fast = sorted(random.Random(8).sample(range(n), 1311))          # 1 % of the slice, fast form (SURVEY 8d's gate)
    blocks0, fallbacks0 = engine.fd_stats()
    check_modp_box(engine, box, idx, fast)
    blocks1, fallbacks1 = engine.fd_stats()
    assert blocks1 > blocks0 and fallbacks1 == fallbacks0      # the forward-difference path ran and held


def test_forward_differences_hold_with_a_dozen_boxes_in_flight(engine):
    """The forward-difference pipelines are chains of single-wave stages that must all be resident; their stage time-outs
    would only bite on a saturated GPU.  Thirteen headline-shape blocks (65536, 256) are enqueued back to back through the
    block API -- the chip is then as full as bench.py ever makes it -- and absorbed in order: no block may fall back to
    Horner's rule, every transcript must be the dealer's, and the X / a1 / a2 of a block from the middle of the queue must
    equal the fast form of the reference arithmetic on a seeded sample (and the dealer's outputs everywhere)."""
    import ctypes as C
    n, t, in_flight, probe = 65536, 256, 13, 6
    box = make_modp_box(engine, n, t, seed=n + 256)          # (the box of test_headline_shape_seeded_one_percent_sample)
    blocks0, fallbacks0 = engine.fd_stats()
    engine.lib.mpvss_ctx_synchronize(engine.ctx)
    for _ in range(in_flight):
        engine.verify_block_compute(box["cm"], box["pos"], box["pk"], box["Y"], box["r"], box["c"])
    X, A1, A2 = ((C.c_uint8 * (n * EB))() for _ in range(3))
    for b in range(in_flight):
        st = (C.c_uint8 * capi.TRANSCRIPT_STATE_BYTES).from_buffer_copy(capi.transcript_init())
        outs = (X, A1, A2) if b == probe else (None, None, None)
        engine._check(engine.lib.mpvss_modp_verify_block_absorb(engine.ctx, st, *outs), "verify_block_absorb")
        assert capi.transcript_verdict(bytes(st), box["c"]) == (True, box["d"]["digest"]), f"block {b}"
    blocks1, fallbacks1 = engine.fd_stats()
    assert blocks1 - blocks0 == in_flight and fallbacks1 == fallbacks0, "a forward-difference pipeline gave up under load"
    Xb, A1b, A2b = bytes(X), bytes(A1), bytes(A2)
    assert (Xb, A1b, A2b) == (box["d"]["X"], box["d"]["a1"], box["d"]["a2"])
    idx = sorted(random.Random(13).sample(range(n), 96))

    def arg(i):
        s = slice(i * EB, (i + 1) * EB)
        return (box["cm"], box["pos"][i], box["pk"][s], box["Y"][s], box["r"][s], box["c"])
    for i, (x, a1, a2) in zip(idx, parallel_map(modp_fast_share, [arg(i) for i in idx])):
        s = slice(i * EB, (i + 1) * EB)
        assert (x, a1, a2) == (Xb[s], A1b[s], A2b[s]), f"share {i}"


GID = {"secp256k1": capi.GROUP_SECP256K1, "ristretto255": capi.GROUP_RISTRETTO255}


@pytest.mark.parametrize("name", ["secp256k1", "ristretto255"])
def test_c3_c4_curve_groups_full_size(engine, name):
    """BASELINE configs C3 (secp256k1) and C4 (ristretto255): n=65536, t=256.  Engine dealer box -> verify, digest
    equality, three tamper cases, 64 seeded positions of X / a1 / a2 against oracle/mpvss_oracle.py in the reference
    operation order, and X over doctored commitments (a duplicate and the identity) at 16 positions."""
    n, t = 65536, 256
    G = O.GROUPS[name]()
    gid = GID[name]
    order = G.group_order_int()
    L = G.elem_len
    sb = G.scalar_to_fixed
    rng = random.Random(0xC3C4 + gid)
    coeffs = [rng.randrange(order) for _ in range(t)]
    privs = [rng.randrange(1, order) for _ in range(n)]
    wits = [rng.randrange(1, order) for _ in range(n)]
    positions = list(range(1, n + 1))
    pvals = poly_values(coeffs, positions, order)
    gen = G.element_to_bytes(G.generator())
    cm = engine.ec_batch_exp(gid, gen * t, b"".join(map(sb, coeffs)))
    pks = engine.ec_batch_exp(gid, gen * n, b"".join(map(sb, privs)))
    d = engine.ec_distribute(gid, cm, positions, pks, b"".join(map(sb, pvals)), b"".join(map(sb, wits)))
    c = G.hash_to_scalar(d["digest"])                        # secp256k1.rs:121-131 / ristretto255.rs:196-205
    assert sb(c) == capi.ec_hash_to_scalar(gid, d["digest"])
    responses = b"".join(sb((w - p * c) % order) for w, p in zip(wits, pvals))     # dleq.rs:42-50
    res = engine.ec_verify_distribution(gid, cm, positions, pks, d["Y"], responses, sb(c), dump=True)
    assert res["verdict"] is True and res["digest"] == d["digest"]
    assert res["X"] == d["X"] and res["a1"] == d["a1"] and res["a2"] == d["a2"]
    trng = random.Random(5)
    for field in ("r", "Y", "cm"):
        args = {"cm": cm, "Y": d["Y"], "r": responses}
        buf = bytearray(args[field])
        if field == "r":                                     # stay below the group order: flip a low bit of a response
            k = trng.randrange(n)
            buf[k * 32 + (31 if name == "secp256k1" else 0)] ^= 1
        elif field == "Y":                                   # another valid point in place of one share
            k = trng.randrange(n)
            buf[k * L:(k + 1) * L] = pks[((k + 1) % n) * L:((k + 1) % n + 1) * L]
        else:
            k = trng.randrange(t)
            buf[k * L:(k + 1) * L] = pks[k * L:(k + 1) * L]
        args[field] = bytes(buf)
        bad = engine.ec_verify_distribution(gid, args["cm"], positions, pks, args["Y"], args["r"], sb(c))
        assert bad["verdict"] is False and bad["digest"] != d["digest"], field
    idx = sorted(set(random.Random(11).sample(range(n), 62)) | {0, n - 1})
    jobs = [(name, cm, positions[i], pks[i * L:(i + 1) * L], d["Y"][i * L:(i + 1) * L], responses[i * 32:(i + 1) * 32], sb(c))
            for i in idx]
    for i, (x, a1, a2) in zip(idx, parallel_map(ec_reference_share, jobs)):
        s = slice(i * L, (i + 1) * L)
        assert (x, a1, a2) == (res["X"][s], res["a1"][s], res["a2"][s]), f"{name} share {i}"
    # a seeded 1 % sample (656 shares) against the independent C restatement (oracle/ec_ref.c, reference order)
    from ec_ref import EcRef
    ref = EcRef()
    idx3 = sorted(set(random.Random(13).sample(range(n), 656)))

    def work(i):
        return ref.share_work(gid, cm, positions[i], pks[i * L:(i + 1) * L], d["Y"][i * L:(i + 1) * L],
                              responses[i * 32:(i + 1) * 32], sb(c))

    with concurrent.futures.ThreadPoolExecutor(max_workers=worker_count(64)) as ex:
        for i, (x, a1, a2) in zip(idx3, ex.map(work, idx3)):
            s = slice(i * L, (i + 1) * L)
            assert (x, a1, a2) == (res["X"][s], res["a1"][s], res["a2"][s]), f"{name} share {i} (C port)"
    # adversarial commitments: a duplicate and the identity (X accumulation starts from the identity and meets P + P)
    ident = G.element_to_bytes(G.identity())
    doctored = bytearray(cm)
    doctored[3 * L:4 * L] = cm[2 * L:3 * L]
    doctored[5 * L:6 * L] = ident
    doctored = bytes(doctored)
    xs = engine.ec_commit_eval(gid, doctored, positions)
    idx2 = sorted(set(random.Random(12).sample(range(n), 14)) | {0, n - 1})
    for i, x in zip(idx2, parallel_map(ec_reference_x, [(name, doctored, positions[i]) for i in idx2])):
        assert x == xs[i * L:(i + 1) * L], f"{name} doctored X {i}"
