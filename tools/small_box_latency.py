#!/usr/bin/env python3
"""What ONE call costs for the box sizes the reference's own tests and examples use (a handful to a few hundred participants): the latency of
mpvss_modp_deal, mpvss_modp_verify_distribution, mpvss_modp_verify_shares and the curve twins on an idle chip, host buffers, best of `reps`.
usage: small_box_latency.py [reps] [n,t n,t ...]"""
import os
import random
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401
from mpvss_rs_amd import Engine, capi  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[2:]] or [(4, 3), (16, 8), (64, 16), (256, 32), (1024, 32), (4096, 64)]
EB = 256
eng = Engine(0)
rng = random.Random(3)
fx = lambda v: v.to_bytes(EB, "big")
sc = lambda k: b"".join(rng.randrange(1, 1 << 2040).to_bytes(EB, "big") for _ in range(k))


def best(f):
    f()
    out = []
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        out.append((time.perf_counter() - t0) * 1e3)
    return min(out)


for n, t in shapes:
    pos = list(range(1, n + 1))
    coeffs, wit, priv = sc(t), sc(n), sc(n)
    pk = eng.batch_exp_fixed_base(fx(2), priv)
    cm = eng.batch_exp_fixed_base(fx(4), coeffs)
    box = eng.deal(coeffs, pos, pk, wit)
    ms_deal = best(lambda: eng.deal(coeffs, pos, pk, wit))
    ms_ver = best(lambda: eng.verify_distribution(cm, pos, pk, box["Y"], box["responses"], box["challenge"]))
    assert eng.verify_distribution(cm, pos, pk, box["Y"], box["responses"], box["challenge"])["verdict"]
    ms_exp = best(lambda: eng.batch_exp(pk, wit))
    line = f"modp n={n:5d} t={t:3d}: deal {ms_deal:6.2f} ms  verify_distribution {ms_ver:6.2f} ms  batch_exp {ms_exp:6.2f} ms"
    for gid, name, be in ((capi.GROUP_SECP256K1, "secp", "big"), (capi.GROUP_RISTRETTO255, "rist", "little")):
        s32 = lambda k: b"".join(rng.randrange(1, 1 << 250).to_bytes(32, be) for _ in range(k))
        c32, w32 = s32(t), s32(n)
        pke = eng.ec_batch_exp_generator(gid, s32(n))
        cme = eng.ec_batch_exp_generator(gid, c32)
        b = eng.ec_deal(gid, c32, pos, pke, w32)
        md = best(lambda: eng.ec_deal(gid, c32, pos, pke, w32))
        mv = best(lambda: eng.ec_verify_distribution(gid, cme, pos, pke, b["Y"], b["responses"], b["challenge"]))
        line += f" | {name}: deal {md:5.2f} verify {mv:5.2f}"
    print(line, flush=True)
eng.close()
