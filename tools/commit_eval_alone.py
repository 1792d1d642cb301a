#!/usr/bin/env python3
"""The X path ALONE on the chip: mpvss_modp_commit_eval over consecutive positions (the commitment multi-exp as a stand-alone call,
src/mpvss.rs:110-123 / participant.rs:423-434), nothing else running.  Prints the call's wall time and the X path's event time.
usage: commit_eval_alone.py [n] [t] [reps]      (MPVSS_FD_ROW=0: quad-layout seeds)"""
import os
import random
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401
from mpvss_rs_amd import Engine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
t = int(sys.argv[2]) if len(sys.argv) > 2 else 256
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
EB = 256
eng = Engine(0)
rng = random.Random(3)
cm = eng.batch_exp_fixed_base((4).to_bytes(EB, "big"), b"".join(rng.randrange(1, 1 << 2040).to_bytes(EB, "big") for _ in range(t)))
pos = list(range(1, n + 1))
best = (1e9, 0)
for k in range(reps + 1):
    t0 = time.perf_counter()
    X = eng.commit_eval(cm, pos)
    dt = (time.perf_counter() - t0) * 1e3
    if k:
        best = min(best, (dt, eng.kernel_ms(0)))
print(f"commit_eval n={n} t={t} MPVSS_FD_ROW={os.environ.get('MPVSS_FD_ROW', '1')}: call {best[0]:.1f} ms (with Python marshalling), X path on the GPU {best[1]:.1f} ms")
