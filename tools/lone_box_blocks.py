#!/usr/bin/env python3
"""ONE box at the headline shape verified as B blocks of one context in flight (mpvss_modp_verify_block_compute x B, then the absorbs in
order): the transcript hash of block b runs while blocks b+1.. compute -- against the single mpvss_modp_verify_distribution call, whose hash
(35 ms) starts when the whole box's GPU work (86 ms) is done.
  python3 tools/lone_box_blocks.py [n] [t] [reps] [B,B,...]"""
import ctypes as C
import os
import random
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401
from mpvss_rs_amd import Engine, capi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
t = int(sys.argv[2]) if len(sys.argv) > 2 else 256
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
splits = [int(x) for x in (sys.argv[4] if len(sys.argv) > 4 else "1,2,4,8").split(",")]
EB = 256
eng = Engine(0)
rng = random.Random(1)
pos = list(range(1, n + 1))
sc = lambda k: b"".join(rng.randrange(1, 1 << 2040).to_bytes(EB, "big") for _ in range(k))
coeffs, wit = sc(t), sc(n)
pk = eng.batch_exp_fixed_base((2).to_bytes(EB, "big"), sc(n))
cm = eng.batch_exp_fixed_base((4).to_bytes(EB, "big"), coeffs)
box = eng.deal(coeffs, pos, pk, wit)
bufs = [(C.c_uint8 * len(b)).from_buffer_copy(b) for b in (cm, pk, box["Y"], box["responses"], box["challenge"])]
parr = (C.c_int64 * n)(*pos)
verdict, dg = C.c_int(0), (C.c_uint8 * 32)()
u8p, i64p = C.POINTER(C.c_uint8), C.POINTER(C.c_int64)
adr = lambda buf, off, ty=u8p: C.cast(C.c_void_p(C.addressof(buf) + off), ty)
for k in range(reps + 1):
    t0 = time.perf_counter()
    eng._check(eng.lib.mpvss_modp_verify_distribution(eng.ctx, capi.MPVSS_HOST, bufs[0], t, C.cast(parr, C.c_void_p), bufs[1], bufs[2], bufs[3], n,
                                                      C.cast(bufs[4], C.c_void_p), C.byref(verdict), dg, None, None, None), "verify_distribution")
    print(f"one call {k}: {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
    assert verdict.value == 1 and bytes(dg) == box["digest"]
for B in splits:
    per = (n + B - 1) // B
    for k in range(reps + 1):
        state = (C.c_uint8 * 128).from_buffer_copy(capi.transcript_init())
        t0 = time.perf_counter()
        for b in range(B):
            lo, cnt = b * per, min(per, n - b * per)
            eng._check(eng.lib.mpvss_modp_verify_block_compute(eng.ctx, capi.MPVSS_HOST, bufs[0], t, adr(parr, lo * 8, i64p), adr(bufs[1], lo * EB),
                                                               adr(bufs[2], lo * EB), adr(bufs[3], lo * EB), cnt, adr(bufs[4], 0)),
                       "verify_block_compute")
        t1 = time.perf_counter()
        for b in range(B):
            eng._check(eng.lib.mpvss_modp_verify_block_absorb(eng.ctx, state, None, None, None), "verify_block_absorb")
        ms = (time.perf_counter() - t0) * 1e3
        ok, digest = capi.transcript_verdict(bytes(state), box["challenge"])
        assert ok and digest == box["digest"], (B, k)
        print(f"{B} blocks {k}: {ms:.1f} ms (enqueue {(t1 - t0) * 1e3:.1f})", flush=True)
print("fd", eng.fd_stats())
eng.close()
