"""BASELINE config C5 as ONE box on one GPU: ModpGroup 2048-bit, n = 2^20 participants, t = 1024.

Rounds 1-4 only ever ran one rank's 131072-share slice.  Here the whole box is dealt (`mpvss_modp_deal`, any n: what the crate's
`distribute_secret` binds, src/participant.rs:160-286), and verified (src/participant.rs:399-455) four ways that must agree:

  (a) ONE `mpvss_modp_verify_distribution` call over all 2^20 shares (what the crate's `verify_distribution_shares` binds);
  (b) the 8 contiguous 131072-position blocks of SURVEY 8(e) through the block API, the 128-byte running hash state carried from
      block to block into ONE transcript -- every X / a1 / a2 of every share compared with the dealer's;
  (c) `mpvss_modp_verify_many_chained` on 8 in-process "ranks" (8 engines, each its block of every box, the state handed on through
      the callbacks): the honest box and three copies with ONE bit flipped in block 0 / 3 / 7;
  (d) positions are 1-based insertion order (sharebox.rs:74-86, participant.rs:186,247): block g holds positions g*131072+1 ...

Oracle: 16 shares (two per block, the box's first and last among them) in the REFERENCE operation order through oracle/modp_ref.c,
a seeded 0.1 % (1049 shares) through helpers.modp_fast_share, the dealer's P(i) / X / Y / a1 / a2 / r of 64 shares against Python
integers, and the whole transcript (4 x 2^20 framed minimal-length elements, dleq.rs:58-61,87-99) re-hashed with hashlib.
No forward-difference pipeline may fall back."""
import concurrent.futures
import ctypes as C
import hashlib
import queue
import random
import threading

import pytest

from helpers import EB, MODP_ORDER as ORDER, MODP_Q as Q, modp_fast_share, parallel_map, worker_count
from mpvss_rs_amd import Engine, capi

pytestmark = pytest.mark.gpu

N, T, RANKS = 1 << 20, 1024, 8
BLK = N // RANKS


def fx(v):
    return v.to_bytes(EB, "big")


def below_q(raw):
    """n random 2048-bit strings with the top bit cleared (< q; the engine and the reference reduce exponents the same way)"""
    b = bytearray(raw)
    b[0::EB] = bytes(x & 0x7F for x in b[0::EB])
    return bytes(b)


def poly(coeffs, i):
    acc = 0
    for a in reversed(coeffs):                                   # polynomial.rs:50-58, then `% order` (participant.rs:202)
        acc = (acc * i + a) % ORDER
    return acc


@pytest.fixture(scope="module")
def c5():
    # engines of this module's own, closed when their work is done: a block slot of this shape is several GB, and the eight ranks of
    # the last test need the HBM (the session's engine keeps what it once allocated)
    engine = Engine(0)
    rng = random.Random(0xC5)
    coeffs = [rng.randrange(ORDER) for _ in range(T)]
    priv = below_q(b"".join(rng.randbytes(1 << 24) for _ in range(N * EB >> 24)))      # (randbytes cannot make 256 MB at once)
    wit = below_q(b"".join(rng.randbytes(1 << 24) for _ in range(N * EB >> 24)))
    pos = list(range(1, N + 1))
    pk = engine.batch_exp_fixed_base(fx(2), priv)
    cm = engine.batch_exp_fixed_base(fx(4), b"".join(map(fx, coeffs)))
    box = engine.deal(b"".join(map(fx, coeffs)), pos, pk, wit)
    engine.close()
    return {"coeffs": coeffs, "wit": wit, "pos": pos, "pk": pk, "cm": cm, "box": box}


def sl(buf, lo, hi):
    return buf[lo * EB:hi * EB]


def test_the_dealers_box_against_python_integers_and_hashlib(c5):
    box, pk, wit = c5["box"], c5["pk"], c5["wit"]
    # the whole transcript, framed as dleq.rs:58-61 does (u64 big-endian length, minimal-length big-endian magnitude)
    h = hashlib.sha256()
    X, Y, A1, A2 = (memoryview(box[k]) for k in ("X", "Y", "a1", "a2"))
    for i in range(N):
        for arr in (X, Y, A1, A2):
            e = bytes(arr[i * EB:(i + 1) * EB]).lstrip(b"\0") or b"\0"
            h.update(len(e).to_bytes(8, "big") + e)
    assert h.digest() == box["digest"]
    c = int.from_bytes(hashlib.sha256(box["digest"]).digest(), "big") % ((Q - 1) // 2)          # modp.rs:142-148
    assert box["challenge"] == fx(c)
    idx = sorted(set(random.Random(1).sample(range(N), 60)) | {0, 1, N - 2, N - 1})
    for i in idx:
        p = poly(c5["coeffs"], i + 1)
        y, w = (int.from_bytes(sl(b, i, i + 1), "big") for b in (pk, wit))
        assert sl(box["X"], i, i + 1) == fx(pow(4, p, Q)) and sl(box["Y"], i, i + 1) == fx(pow(y, p, Q)), i
        assert sl(box["a1"], i, i + 1) == fx(pow(4, w, Q)) and sl(box["a2"], i, i + 1) == fx(pow(y, w, Q)), i
        assert sl(box["responses"], i, i + 1) == fx((w - p * c) % ORDER), i                   # dleq.rs:42-50


def test_one_call_and_eight_chained_blocks_agree_with_the_dealer_and_the_oracle(c5):
    box, pk, cm, pos = c5["box"], c5["pk"], c5["cm"], c5["pos"]
    Y, r, c = box["Y"], box["responses"], box["challenge"]
    engine = Engine(0)
    blocks0, fallbacks0 = engine.fd_stats()
    # (a) one call, as the crate's verify_distribution_shares would make it
    one = engine.verify_distribution(cm, pos, pk, Y, r, c)
    assert one == {"verdict": True, "digest": box["digest"]}
    # (b) 8 contiguous blocks, one transcript: enqueue all, absorb in order with the carried state
    for g in range(RANKS):
        lo, hi = g * BLK, (g + 1) * BLK
        engine.verify_block_compute(cm, pos[lo:hi], sl(pk, lo, hi), sl(Y, lo, hi), sl(r, lo, hi), c)
    state = capi.transcript_init()
    for g in range(RANKS):
        lo, hi = g * BLK, (g + 1) * BLK
        state, X, a1, a2 = engine.verify_block_absorb_dump(state, BLK)
        assert X == sl(box["X"], lo, hi), f"X of block {g}"
        assert a1 == sl(box["a1"], lo, hi) and a2 == sl(box["a2"], lo, hi), f"a1 / a2 of block {g}"      # g^r X^c == g^w, y^r Y^c == y^w
        if g < RANKS - 1:
            assert capi.transcript_verdict(state, c)[0] is False        # a prefix of the transcript is not the transcript
    assert capi.transcript_verdict(state, c) == (True, box["digest"])
    blocks1, fallbacks1 = engine.fd_stats()
    assert blocks1 > blocks0 and fallbacks1 == fallbacks0, "a forward-difference pipeline gave up"
    engine.close()
    # the oracle on sampled shares: the verifier's X / a1 / a2 equal the dealer's everywhere (above), so the dealer's arrays are compared
    from modp_ref import ModpRef
    ref = ModpRef()
    rng = random.Random(16)
    idx = sorted({0, N - 1} | {g * BLK + rng.randrange(BLK) for g in range(RANKS) for _ in range(2)})[:18]
    assert len(idx) >= 16

    def work(i):
        return ref.share_work(cm, pos[i], sl(pk, i, i + 1), sl(Y, i, i + 1), sl(r, i, i + 1), c)

    with concurrent.futures.ThreadPoolExecutor(max_workers=worker_count(64)) as ex:     # ctypes releases the GIL
        fut = ex.map(work, idx)
        fast = sorted(random.Random(17).sample(range(N), N // 1000 + 1))
        outs_fast = parallel_map(modp_fast_share, [(cm, pos[i], sl(pk, i, i + 1), sl(Y, i, i + 1), sl(r, i, i + 1), c) for i in fast])
        outs = list(fut)
    for i, (x, a1, a2) in zip(idx, outs):
        assert (x, a1, a2) == (sl(box["X"], i, i + 1), sl(box["a1"], i, i + 1), sl(box["a2"], i, i + 1)), f"share {i} (reference order)"
    for i, (x, a1, a2) in zip(fast, outs_fast):
        assert (x, a1, a2) == (sl(box["X"], i, i + 1), sl(box["a1"], i, i + 1), sl(box["a2"], i, i + 1)), f"share {i} (fast form)"


def test_eight_in_process_ranks_chained_and_one_flipped_bit_per_block(c5):
    """mpvss_modp_verify_many_chained, one engine per "rank": boxes = [honest, bit flipped in block 0, in block 3, in block 7]."""
    box, pk, cm, pos = c5["box"], c5["pk"], c5["cm"], c5["pos"]
    Y, r, c = box["Y"], box["responses"], box["challenge"]
    SB = capi.TRANSCRIPT_STATE_BYTES
    trng = random.Random(5)

    def flipped(buf, g):
        b = bytearray(sl(buf, g * BLK, (g + 1) * BLK))
        b[trng.randrange(len(b))] ^= 1 << trng.randrange(8)
        return bytes(b)

    tampers = {1: (0, "r"), 2: (3, "Y"), 3: (7, "r")}                  # box -> (block, field)
    links = [[queue.Queue() for _ in range(4)] for _ in range(RANKS)]   # links[g][box]: state from rank g-1 to rank g
    results, errors = [None] * RANKS, []

    def rank(g):
        try:
            eng = Engine(0)
            lo, hi = g * BLK, (g + 1) * BLK
            keep = []
            arr = (capi.ModpBox * 4)()
            posb = (C.c_int64 * BLK)(*pos[lo:hi])
            for b in range(4):
                Yb, rb = sl(Y, lo, hi), sl(r, lo, hi)
                if b in tampers and tampers[b][0] == g:
                    if tampers[b][1] == "r":
                        rb = flipped(r, g)
                    else:
                        Yb = flipped(Y, g)
                bufs = [(C.c_uint8 * len(x)).from_buffer_copy(x) for x in (cm, sl(pk, lo, hi), Yb, rb, c)]
                keep.append(bufs)
                arr[b] = capi.ModpBox(C.addressof(bufs[0]), T, C.addressof(posb), C.addressof(bufs[1]), C.addressof(bufs[2]),
                                      C.addressof(bufs[3]), BLK, C.addressof(bufs[4]), None, 0)

            def cb_in(user, b, state, ok):
                okp, raw = links[g][b].get(timeout=600)
                C.memmove(state, raw, SB)
                return 0 if okp else 1

            def cb_out(user, b, state, ok):
                if g + 1 < RANKS:
                    links[g + 1][b].put((bool(ok), C.string_at(state, SB)))
                return 0

            c_in = capi.CHAIN_CB(cb_in) if g > 0 else capi.CHAIN_CB()
            c_out = capi.CHAIN_CB(cb_out)
            verdicts = (C.c_int * 4)()
            digests = (C.c_uint8 * 128)()
            f0 = eng.fd_stats()
            rc = eng.lib.mpvss_modp_verify_many_chained(eng.ctx, capi.MPVSS_HOST, arr, 4, 1, 2, None, c_in, c_out, None, verdicts,
                                                        C.cast(digests, C.c_void_p))
            eng._check(rc, "verify_many_chained")
            f1 = eng.fd_stats()
            results[g] = ([bool(v) for v in verdicts], bytes(digests), f1[0] - f0[0], f1[1] - f0[1])
            eng.close()
        except Exception as exc:      # noqa: BLE001 - reported by the main thread
            errors.append((g, exc))
            if g + 1 < RANKS:
                for b in range(4):
                    links[g + 1][b].put((False, bytes(SB)))

    threads = [threading.Thread(target=rank, args=(g,)) for g in range(RANKS)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    verdicts, digests, _, _ = results[RANKS - 1]
    assert verdicts == [True, False, False, False]
    assert digests[:32] == box["digest"] and all(digests[32 * b:32 * b + 32] != box["digest"] for b in (1, 2, 3))
    for g in range(RANKS):
        assert results[g][2] >= 4 and results[g][3] == 0, f"rank {g}: forward-difference blocks {results[g][2]}, fall-backs {results[g][3]}"
