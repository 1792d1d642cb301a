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


@pytest.mark.parametrize("n,t,key_offset", [(8192, 16, 0), (8192, 32, 64)])
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
    for i in (0, 1, 2, 5, 6, n // 2, n - 1):
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
    assert engine.set_key_cache(3) == 0
    try:
        cached = run()
    finally:
        assert engine.set_key_cache(0) == 3
    assert cached == plain
    # six of the eight boxes went through the key tables: no 64-entry table of y was built for them
    assert engine.pipeline_stats(reset=True)["kernel_launches"][2] == tables_plain - 6
    with pytest.raises(capi.EngineError):
        engine.set_key_cache(1)
