"""Dealer blocks on one GPU: lone boxes (kernel durations) and a pipelined run.  usage: bench_dealer.py [n] [boxes] [depth]"""
import ctypes as C, os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpvss_rs_amd import capi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
boxes = int(sys.argv[2]) if len(sys.argv) > 2 else 12
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 8
EB = 256
eng = capi.Engine(0); lib, ctx = eng.lib, eng.ctx
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu"); g.manual_seed(5)
def rnd(): return torch.randint(0, 256, (n * EB,), dtype=torch.uint8, generator=g).to(dev)
pk, pv, wt = rnd(), rnd(), rnd()
vp = lambda t: C.c_void_p(t.data_ptr())
def compute():
    eng._check(lib.mpvss_modp_distribute_compute(ctx, capi.MPVSS_DEVICE, None, 0, None, vp(pk), vp(pv), vp(wt), n, None, None, None, None), "compute")
def absorb():
    st = (C.c_uint8 * capi.TRANSCRIPT_STATE_BYTES).from_buffer_copy(capi.transcript_init())
    eng._check(lib.mpvss_modp_distribute_absorb(ctx, st, None, None, None, None), "absorb")
for _ in range(2):
    compute(); absorb()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    compute(); absorb()
lone = (time.perf_counter() - t0) / 3
print("lone box ms", round(lone * 1e3, 1), "kernel_ms by kind", [round(x, 2) for x in eng.last_kernel_ms()] if hasattr(eng, "last_kernel_ms") else "")
import concurrent.futures
pool = concurrent.futures.ThreadPoolExecutor(max_workers=4)
for d in (depth,):
    for _ in range(16): compute()
    for _ in range(16): absorb()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    issued = 0; pend = []
    while issued < min(d, boxes): compute(); issued += 1; pend.append(pool.submit(absorb))
    while pend:
        pend.pop(0).result()
        if issued < boxes: compute(); issued += 1; pend.append(pool.submit(absorb))
    dt = (time.perf_counter() - t0) / boxes
    print("pipelined depth", d, "ms/box", round(dt * 1e3, 1), "shares/s", round(n / dt))
