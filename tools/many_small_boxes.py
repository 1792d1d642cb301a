#!/usr/bin/env python3
"""Many dealers' SMALL boxes (the sizes the reference's own tests use) through ONE mpvss_modp_verify_many call: runs of same-shaped boxes travel as
one block whatever their size (round 6: also below the forward differences' 4096 shares).  The library call alone over ctypes boxes made once;
one tampered box must be the only rejected one.  usage: many_small_boxes.py"""
import os, sys, time, random, ctypes as C
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.getcwd())
import torch
from mpvss_rs_amd import Engine, capi
eng = Engine(0)
fx = lambda v: v.to_bytes(256, "big")
rng = random.Random(5)
sc = lambda k: b"".join(fx(rng.randrange(1, 1 << 2040)) for _ in range(k))
for n, t, K in ((5, 3, 400), (100, 34, 200), (1024, 32, 120)):
    pos = list(range(1, n + 1))
    pk = eng.batch_exp_fixed_base(fx(2), sc(n))
    boxes = []
    for k in range(min(K, 12)):
        co, wi = sc(t), sc(n)
        d = eng.deal(co, pos, pk, wi)
        boxes.append(dict(commitments=eng.batch_exp_fixed_base(fx(4), co), positions=pos, pubkeys=pk, shares=d["Y"], responses=d["responses"],
                          challenge=d["challenge"], digest=d["digest"]))
    seq = [boxes[i % len(boxes)] for i in range(K)]
    bad = dict(seq[3]); bad["responses"] = bad["responses"][:300] + bytes([bad["responses"][300] ^ 1]) + bad["responses"][301:]
    seq[3] = bad
    keep, arr = [], (capi.ModpBox * K)()
    made = {}
    for i, b in enumerate(seq):
        if id(b) not in made:
            posb = (C.c_int64 * n)(*b["positions"])
            bufs = [(C.c_uint8 * len(b[k])).from_buffer_copy(b[k]) for k in ("commitments", "pubkeys", "shares", "responses", "challenge")]
            made[id(b)] = (posb, bufs)
        posb, bufs = made[id(b)]
        arr[i] = capi.ModpBox(C.addressof(bufs[0]), t, C.addressof(posb), C.addressof(bufs[1]), C.addressof(bufs[2]), C.addressof(bufs[3]), n,
                              C.addressof(bufs[4]), None, 0)
    verdicts = (C.c_int * K)()
    digests = (C.c_uint8 * (32 * K))()

    def call():
        eng._check(eng.lib.mpvss_modp_verify_many(eng.ctx, capi.MPVSS_HOST, arr, K, 10, 8, verdicts, C.cast(digests, C.c_void_p)), "verify_many")
    call()
    el = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        call()
        el = min(el, time.perf_counter() - t0)
    v = [bool(verdicts[i]) for i in range(K)]
    assert v.count(False) == 1 and not v[3], v[:8]
    assert all(bytes(digests)[32 * i:32 * i + 32] == seq[i]["digest"] for i in range(K) if i != 3)
    print(f"n={n} t={t}: {K} boxes in {el*1e3:.1f} ms = {el/K*1e3:.3f} ms/box = {K/el:.0f} boxes/s = {n*K/el/1e6:.3f} M share verifications/s", flush=True)
