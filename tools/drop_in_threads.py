#!/usr/bin/env python3
"""The reference's call shape under concurrency: T host threads, each calling the ONE-box entry point
mpvss_modp_verify_distribution (what rust/src/batch.rs::verify_distribution_shares binds; participant.rs:399-455) on ONE
context, over K distinct dealers' boxes -- what a crate user gets who parallelises over dealers the way participant.rs:490-500
parallelises (rayon).  ctypes releases the GIL for the duration of a foreign call, so the Python threads here behave like
compiled callers; buffers are made once.
usage: drop_in_threads.py [n] [t] [K] [threads,comma,separated] [host|device|both]"""
import ctypes as C
import os
import random
import sys
import threading
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")    # as bench.py / mpvss_process_init(): the block pipeline is tuned for 8 hardware queues
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402  (its HIP runtime first)
from mpvss_rs_amd import Engine, capi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
t = int(sys.argv[2]) if len(sys.argv) > 2 else 256
K = int(sys.argv[3]) if len(sys.argv) > 3 else 20
TS = [int(x) for x in (sys.argv[4] if len(sys.argv) > 4 else "1,4,8,12,16").split(",")]
spaces = sys.argv[5] if len(sys.argv) > 5 else "both"
EB = 256
eng = Engine(0)
lib, ctx = eng.lib, eng.ctx
rng = random.Random(1)
pos = list(range(1, n + 1))
sc = lambda k: b"".join(rng.randrange(1, 1 << 2040).to_bytes(EB, "big") for _ in range(k))
pk = eng.batch_exp_fixed_base((2).to_bytes(EB, "big"), sc(n))
wit = sc(n)
dev = torch.device("cuda", 0)


class Bx:
    pass


boxes = []
parr = (C.c_int64 * n)(*pos)
d_pos = torch.tensor(pos, dtype=torch.int64, device=dev)
pk_h = (C.c_uint8 * len(pk)).from_buffer_copy(pk)
d_pk = torch.frombuffer(bytearray(pk), dtype=torch.uint8).to(dev)
for b in range(K):
    coeffs = sc(t)
    cm = eng.batch_exp_fixed_base((4).to_bytes(EB, "big"), coeffs)
    box = eng.deal(coeffs, pos, pk, wit)
    bx = Bx()
    bx.digest = box["digest"]
    bx.h = [(C.c_uint8 * len(x)).from_buffer_copy(x) for x in (cm, box["Y"], box["responses"], box["challenge"])]
    bx.d = [torch.frombuffer(bytearray(x), dtype=torch.uint8).to(dev) for x in (cm, box["Y"], box["responses"])]
    boxes.append(bx)
torch.cuda.synchronize()


def verify_one(bx, space):
    verdict, dg = C.c_int(0), (C.c_uint8 * 32)()
    if space == capi.MPVSS_HOST:
        rc = lib.mpvss_modp_verify_distribution(ctx, space, bx.h[0], t, C.cast(parr, C.c_void_p), pk_h, bx.h[1], bx.h[2], n,
                                                C.cast(bx.h[3], C.c_void_p), C.byref(verdict), dg, None, None, None)
    else:
        vp = lambda x: C.c_void_p(x.data_ptr())
        rc = lib.mpvss_modp_verify_distribution(ctx, space, vp(bx.d[0]), t, vp(d_pos), vp(d_pk), vp(bx.d[1]), vp(bx.d[2]), n,
                                                C.cast(bx.h[3], C.c_void_p), C.byref(verdict), dg, None, None, None)
    eng._check(rc, "verify_distribution")
    assert verdict.value == 1 and bytes(dg) == bx.digest, "drop-in: verdict or digest wrong"


def run(T, space, seq):
    """T threads share the sequence of boxes round-robin; returns seconds"""
    errs = []

    def worker(k):
        try:
            torch.cuda.set_device(dev)
            for i in range(k, len(seq), T):
                verify_one(seq[i], space)
        except Exception as exc:  # noqa: BLE001
            errs.append(exc)

    ths = [threading.Thread(target=worker, args=(k,)) for k in range(T)]
    t0 = time.perf_counter()
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    el = time.perf_counter() - t0
    if errs:
        raise errs[0]
    return el


def many(space, seq, depth=10, threads=8):
    k = len(seq)
    if space == capi.MPVSS_HOST:
        arr = (capi.ModpBox * k)(*[capi.ModpBox(C.addressof(bx.h[0]), t, C.addressof(parr), C.addressof(pk_h), C.addressof(bx.h[1]),
                                                C.addressof(bx.h[2]), n, C.cast(bx.h[3], C.c_void_p), None, 0) for bx in seq])
    else:
        arr = (capi.ModpBox * k)(*[capi.ModpBox(bx.d[0].data_ptr(), t, d_pos.data_ptr(), d_pk.data_ptr(), bx.d[1].data_ptr(),
                                                bx.d[2].data_ptr(), n, C.cast(bx.h[3], C.c_void_p), None, 0) for bx in seq])
    vd = (C.c_int * k)()
    dg = (C.c_uint8 * (32 * k))()
    t0 = time.perf_counter()
    eng._check(lib.mpvss_modp_verify_many(ctx, space, arr, k, depth, threads, vd, C.cast(dg, C.c_void_p)), "verify_many")
    el = time.perf_counter() - t0
    assert all(vd[i] == 1 for i in range(k))
    return el


for sname, space in (("host", capi.MPVSS_HOST), ("device", capi.MPVSS_DEVICE)):
    if spaces not in ("both", sname):
        continue
    # slot initialisation: as many boxes at once as the largest run keeps in flight
    run(max(TS) + 2, space, (boxes * 3)[: max(TS) + 2])
    many(space, (boxes * 2)[:24])
    fb0 = eng.fd_stats()
    for rep in range(2):
        el = many(space, boxes)
        print(f"[{sname}] verify_many depth 10, 8 hash threads, K={K}: {el / K * 1e3:.1f} ms/box -> {n * K / el / 1e6:.3f} M/s", flush=True)
    for T in TS:
        for rep in range(2):
            el = run(T, space, boxes)
            print(f"[{sname}] T={T:2d} threads x verify_distribution, K={K}: {el / K * 1e3:.1f} ms/box -> {n * K / el / 1e6:.3f} M/s", flush=True)
    fb1 = eng.fd_stats()
    print(f"[{sname}] fd blocks {fb1[0] - fb0[0]}, fallbacks {fb1[1] - fb0[1]}")
eng.close()
