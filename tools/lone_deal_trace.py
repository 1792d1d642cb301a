#!/usr/bin/env python3
"""A few LONE mpvss_modp_deal calls at the headline shape (nothing else on the GPU), for a rocprofv3 kernel trace:
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/lone_deal -- python3 tools/lone_deal_trace.py [n] [t] [reps] [key cache 0|1]
  python3 tools/lone_verify_trace.py --timeline gpurun_out/lone_deal/*/*kernel_trace.csv     (kernels of the last call, ms from its start)"""
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
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
cache = int(sys.argv[4]) if len(sys.argv) > 4 else 0
EB = 256
eng = Engine(0)
rng = random.Random(1)
pos = list(range(1, n + 1))
sc = lambda k: b"".join(rng.randrange(1, 1 << 2040).to_bytes(EB, "big") for _ in range(k))
coeffs, wit = sc(t), sc(n)
pk = eng.batch_exp_fixed_base((2).to_bytes(EB, "big"), sc(n))
call, outputs = eng.deal_call(coeffs, pos, pk, wit)
call()
want = outputs()
if cache:
    eng.set_key_cache_lru(1, 1)
    call()
for k in range(reps):
    time.sleep(0.05)
    t0 = time.perf_counter()
    call()
    print(f"deal {k} (key cache {cache}): {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
    assert outputs() == want
eng.set_key_cache_lru(0)
eng.close()
