#!/usr/bin/env python3
"""Wall time of ONE curve-group X path (n = 65536, t = 256, consecutive positions) through the synchronous call
mpvss_ec_commit_eval -- the chip to itself -- for the stepping variants: MPVSS_EC_FD_QUAD=0/2, MPVSS_EC_FD_L1=0/2.
  python3 tools/ec_x_latency.py            (ENV=VALUE arguments go to every child; VARIANTS=quad:l1,... picks the variants; one child per variant: the switches are read once per process)"""
import hashlib
import os
import random
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import bench
    from mpvss_rs_amd import capi
    eng = capi.Engine(0)
    n, t = int(os.environ.get("EC_N", "65536")), 256
    for name in ("secp256k1", "ristretto255"):
        cfg = bench.EC[name]
        gid, order = cfg["gid"], cfg["order"]
        sb = (lambda k: k.to_bytes(32, "big")) if cfg["be"] else (lambda k: k.to_bytes(32, "little"))
        rng = random.Random(7 + gid)
        cm = eng.ec_batch_exp_generator(gid, b"".join(sb(rng.randrange(order)) for _ in range(t)))
        pos = list(range(1, n + 1))
        best, h = 1e9, None
        for rep in range(5):
            t0 = time.perf_counter()
            out = eng.ec_commit_eval(gid, cm, pos)
            best = min(best, time.perf_counter() - t0)
            h = hashlib.sha256(out).hexdigest()[:16]
        print(f"{name:13} {best * 1e3:7.2f} ms  {h}  x_path kernel_ms {eng.kernel_ms(0):.2f}")
    sys.exit(0)
variants = [v.split(":") for v in os.environ.get("VARIANTS", "0:0,0:2,2:0,2:2").split(",")]      # quad:l1
for quad, l1 in variants:
    env = dict(os.environ, MPVSS_EC_FD_QUAD=quad, MPVSS_EC_FD_L1=l1)
    extra = [a.split("=") for a in sys.argv[1:]]
    for k, v in extra:
        env[k] = v
    print(f"== MPVSS_EC_FD_QUAD={quad} MPVSS_EC_FD_L1={l1} " + " ".join(f"{k}={v}" for k, v in extra), flush=True)
    subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env)
