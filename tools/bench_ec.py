#!/usr/bin/env python3
"""Throughput of verify_distribution_shares on the elliptic-curve groups (BASELINE configs C3 / C4:
n=65536, t=256).  Not the headline bench (bench.py is); prints one JSON line per group."""
import argparse
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")    # as bench.py: more queues than that slow every kernel down
import torch  # noqa: E402,F401  (initialises HIP before the library, see tests/conftest.py)
from mpvss_rs_amd import capi  # noqa: E402

ORDERS = {
    "secp256k1": 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141,
    "ristretto255": 2**252 + 27742317777372353535851937790883648493,
}
GEN = {
    "secp256k1": bytes.fromhex("0279BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798"),
    "ristretto255": bytes.fromhex("e2f2ae0a6abc4e71a884a961c500515f58e30b6aa582dd8db6a65945e08d2d76"),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=65536)
    ap.add_argument("--t", type=int, default=256)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--groups", default="secp256k1,ristretto255")
    ap.add_argument("--threads", type=lambda s: [int(x) for x in s.split(",")], default=[],
                    help="also measure with this many contexts / host threads verifying boxes side by side")
    args = ap.parse_args()
    eng = capi.Engine(0)
    for name in args.groups.split(","):
        gid = capi.GROUP_SECP256K1 if name == "secp256k1" else capi.GROUP_RISTRETTO255
        order = ORDERS[name]
        sb = (lambda k: k.to_bytes(32, "big")) if name == "secp256k1" else (lambda k: k.to_bytes(32, "little"))
        n, t = args.n, args.t
        rng = random.Random(0x6D70767373 + gid)
        coeffs = [rng.randrange(order) for _ in range(t)]
        privs = [rng.randrange(1, order) for _ in range(n)]
        wits = [rng.randrange(1, order) for _ in range(n)]
        positions = list(range(1, n + 1))
        pvals = []
        rc = list(reversed(coeffs))
        for i in positions:
            acc = 0
            for a in rc:
                acc = (acc * i + a) % order
            pvals.append(acc)
        cm = eng.ec_batch_exp(gid, GEN[name] * t, b"".join(map(sb, coeffs)))
        pks = eng.ec_batch_exp(gid, GEN[name] * n, b"".join(map(sb, privs)))
        d = eng.ec_distribute(gid, cm, positions, pks, b"".join(map(sb, pvals)), b"".join(map(sb, wits)))
        c = int.from_bytes(capi.ec_hash_to_scalar(gid, d["digest"]), "big" if name == "secp256k1" else "little")
        responses = b"".join(sb((w - p * c) % order) for w, p in zip(wits, pvals))
        res = eng.ec_verify_distribution(gid, cm, positions, pks, d["Y"], responses, sb(c))
        assert res["verdict"] is True and res["digest"] == d["digest"]
        t0 = time.perf_counter()
        for _ in range(args.steps):
            res = eng.ec_verify_distribution(gid, cm, positions, pks, d["Y"], responses, sb(c))
        dt = (time.perf_counter() - t0) / args.steps
        assert res["verdict"] is True
        line = {"group": name, "n": n, "t": t, "share_verifications_per_s": n / dt, "ms_per_box": dt * 1e3,
                "kernel_ms": {"commit_eval": eng.kernel_ms(0), "dual_mul_x2": eng.kernel_ms(1)},
                "buffers": "host (PCIe inclusive)"}
        # several boxes in flight: one context (workspace + stream) and one host thread per box -- the kernels of
        # different boxes share the chip (one box alone is one wave per SIMD) and the host hashes in parallel
        import ctypes as C
        import threading

        import torch
        dev = torch.device("cuda", 0)
        dbuf = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
        d_cm, d_pk, d_Y, d_r = dbuf(cm), dbuf(pks), dbuf(d["Y"]), dbuf(responses)
        d_pos = torch.tensor(positions, dtype=torch.int64, device=dev)
        chal = (C.c_uint8 * 32).from_buffer_copy(sb(c))
        vp = lambda x: C.c_void_p(x.data_ptr())
        torch.cuda.synchronize()
        for nthreads in args.threads:
            engines = [capi.Engine(0) for _ in range(nthreads)]
            oks = []

            def verify(e):      # inputs resident in HBM; the library copies X, Y, a1, a2 back and hashes them
                verdict = C.c_int(0)
                dg = (C.c_uint8 * 32)()
                e._check(e.lib.mpvss_ec_verify_distribution(e.ctx, gid, capi.MPVSS_DEVICE, vp(d_cm), t, vp(d_pos), vp(d_pk),
                                                            vp(d_Y), vp(d_r), n, C.cast(chal, C.c_void_p), C.byref(verdict),
                                                            C.cast(dg, C.c_void_p), None, None, None), "ec_verify_distribution")
                return bool(verdict.value) and bytes(dg) == d["digest"]

            def work(e):
                for _ in range(args.steps):
                    oks.append(verify(e))

            for e in engines:
                assert verify(e)                                        # warm-up, workspace
            ths = [threading.Thread(target=work, args=(e,)) for e in engines]
            t1 = time.perf_counter()
            for th in ths:
                th.start()
            for th in ths:
                th.join()
            dtm = time.perf_counter() - t1
            assert all(oks) and len(oks) == nthreads * args.steps
            line[f"share_verifications_per_s_{nthreads}_contexts_hbm_inputs"] = n * nthreads * args.steps / dtm
            for e in engines:
                e.close()
        print(json.dumps(line))
    eng.close()


if __name__ == "__main__":
    main()
