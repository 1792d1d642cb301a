#!/usr/bin/env python3
"""What a fresh process pays before its calls settle at the latency of tools/small_box_latency.py: context creation (HIP start-up), the
fixed-base comb of each generator at its first use, the first deal's allocations -- at the reference example's size (n = 3, t = 3)."""
import os, sys, time, random
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.getcwd())
import torch
from mpvss_rs_amd import Engine, capi
t0 = time.perf_counter(); eng = Engine(0); t1 = time.perf_counter()
fx = lambda v: v.to_bytes(256, "big")
rng = random.Random(1); sc = lambda k: b"".join(fx(rng.randrange(1, 1 << 2040)) for _ in range(k))
n, t = 3, 3
pos = [1, 2, 3]; co, wi, pr = sc(t), sc(n), sc(n)
ta = time.perf_counter(); pk = eng.batch_exp_fixed_base(fx(2), pr); tb = time.perf_counter()
cm = eng.batch_exp_fixed_base(fx(4), co); tc = time.perf_counter()
box = eng.deal(co, pos, pk, wi); td = time.perf_counter()
r = eng.verify_distribution(cm, pos, pk, box["Y"], box["responses"], box["challenge"]); te = time.perf_counter()
box = eng.deal(co, pos, pk, wi); tf = time.perf_counter()
print(f"ctx_create {1e3*(t1-t0):.1f} ms | first keygen (comb G built) {1e3*(tb-ta):.1f} | first commitments (comb g built) {1e3*(tc-tb):.1f} | first deal {1e3*(td-tc):.1f} | first verify {1e3*(te-td):.1f} | second deal {1e3*(tf-te):.1f}", r["verdict"])
