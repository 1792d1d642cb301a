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
Q = int("ffffffffffffffffc90fdaa22168c234c4c6628b80dc1cd129024e088a67cc74020bbea63b139b22514a08798e3404ddef9519b3cd3a431b302b0a6df25f1437"
        "4fe1356d6d51c245e485b576625e7ec6f44c42e9a637ed6b0bff5cb6f406b7edee386bfb5a899fa5ae9f24117c4b1fe649286651ece45b3dc2007cb8a163bf05"
        "98da48361c55d39a69163fa8fd24cf5f83655d23dca3ad961c62f356208552bb9ed529077096966d670c354e4abc9804f1746c08ca18217c32905e462e36ce3b"
        "e39e772c180e86039b2783a2ec07a28fb5c55df06f4c52c9de2bcbf6955817183995497cea956ae515d2261898fa051015728e5a8aacaa68ffffffffffffffff", 16)   # RFC 3526 group 14 (modp.rs:47-58)
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
    # the participants' side: extract_secret_share (S = Y^(1/x), the DLEQ commitments, the per-share challenge) and verify_share
    # (any odd x below q - 1 that is not a multiple of (q - 1) / 2 is invertible mod q - 1)
    xs = [rng.randrange(3, Q - 1) | 1 for _ in range(n)]
    xinv = b"".join(fx(pow(x, -1, Q - 1)) for x in xs)
    pkx = eng.batch_exp_fixed_base(fx(2), b"".join(map(fx, xs)))
    ms_ext = best(lambda: eng.extract_shares(pkx, box["Y"], xinv, wit))
    S, c = eng.extract_shares(pkx, box["Y"], xinv, wit)
    r = capi.dleq_responses(0, wit, b"".join(map(fx, xs)), c)
    ms_vs = best(lambda: eng.verify_shares(pkx, S, box["Y"], c, r))
    assert all(eng.verify_shares(pkx, S, box["Y"], c, r))
    line = (f"modp n={n:5d} t={t:3d}: deal {ms_deal:6.2f} ms  verify_distribution {ms_ver:6.2f} ms  batch_exp {ms_exp:6.2f} ms  "
            f"extract_shares {ms_ext:6.2f} ms  verify_shares {ms_vs:6.2f} ms")
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
