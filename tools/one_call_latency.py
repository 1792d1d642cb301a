#!/usr/bin/env python3
"""What ONE call from host buffers costs a compiled caller (the Rust crate's distribute_secret / verify_distribution_shares over its
own Vec<u8>s): mpvss_modp_deal, mpvss_ec_deal and mpvss_modp_verify_distribution at the headline shape, the library call alone over
ctypes buffers made once.  MPVSS_TRACE_DEAL=1 prints the phases of every deal on stderr.  usage: one_call_latency.py [n] [t] [reps]"""
import ctypes as C
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402  (its HIP runtime first)
from mpvss_rs_amd import Engine, capi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
t = int(sys.argv[2]) if len(sys.argv) > 2 else 256
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
EB = 256
eng = Engine(0)
rng = random.Random(1)
pos = list(range(1, n + 1))


def times(call):
    call()
    out = []
    for _ in range(reps):
        t0 = time.perf_counter()
        call()
        out.append((time.perf_counter() - t0) * 1e3)
    return out


sc = lambda k: b"".join(rng.randrange(1, 1 << 2040).to_bytes(EB, "big") for _ in range(k))
coeffs, wit = sc(t), sc(n)
pk = eng.batch_exp_fixed_base((2).to_bytes(EB, "big"), sc(n))
cm = eng.batch_exp_fixed_base((4).to_bytes(EB, "big"), coeffs)
call, outputs = eng.deal_call(coeffs, pos, pk, wit)
ms = times(call)
print(f"mpvss_modp_deal n={n} t={t}: {min(ms):.1f} ms best, {sum(ms) / len(ms):.1f} mean -> {n / (sum(ms) / len(ms)) * 1e3 / 1e6:.3f} M shares dealt/s")
box = outputs()
# one verify_distribution call, buffers made once
bufs = [(C.c_uint8 * len(b)).from_buffer_copy(b) for b in (cm, pk, box["Y"], box["responses"], box["challenge"])]
parr = (C.c_int64 * n)(*pos)
verdict, dg = C.c_int(0), (C.c_uint8 * 32)()


def verify():
    eng._check(eng.lib.mpvss_modp_verify_distribution(eng.ctx, capi.MPVSS_HOST, bufs[0], t, C.cast(parr, C.c_void_p), bufs[1], bufs[2], bufs[3], n,
                                                      C.cast(bufs[4], C.c_void_p), C.byref(verdict), dg, None, None, None), "verify_distribution")


ms = times(verify)
assert verdict.value == 1 and bytes(dg) == box["digest"]
print(f"mpvss_modp_verify_distribution n={n} t={t}: {min(ms):.1f} ms best, {sum(ms) / len(ms):.1f} mean -> {n / (sum(ms) / len(ms)) * 1e3 / 1e6:.3f} M share verifications/s")
for gid, name, be in ((capi.GROUP_SECP256K1, "secp256k1", "big"), (capi.GROUP_RISTRETTO255, "ristretto255", "little")):
    s32 = lambda k: b"".join(rng.randrange(1, 1 << 250).to_bytes(32, be) for _ in range(k))
    pk = eng.ec_batch_exp_generator(gid, s32(n))
    call, outputs = eng.ec_deal_call(gid, s32(t), pos, pk, s32(n))
    ms = times(call)
    print(f"mpvss_ec_deal {name} n={n} t={t}: {min(ms):.1f} ms best, {sum(ms) / len(ms):.1f} mean -> {n / (sum(ms) / len(ms)) * 1e3 / 1e6:.3f} M shares dealt/s")
