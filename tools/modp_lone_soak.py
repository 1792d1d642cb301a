#!/usr/bin/env python3
"""Soak of the MODP lone-box path (forward-difference pipelines of single-wave stages that hand numbers down through HBM, 2 s stage
time-outs): the same headline-shape box verified N times through the synchronous call, every digest checked, fall-backs counted
(mpvss_modp_fd_stats), the slowest calls reported.
  python3 tools/modp_lone_soak.py [N=60] [load]     load: a second context keeps 10 boxes of the same shape in flight meanwhile"""
import ctypes as C
import os
import random
import sys
import threading
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from mpvss_rs_amd import capi  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
LOAD = len(sys.argv) > 2 and sys.argv[2] == "load"
EB = 256
n, t = 65536, 256
eng = capi.Engine(0)
eng2 = capi.Engine(0) if LOAD else None
dev = torch.device("cuda", 0)
rng = random.Random(5)
sc = lambda k: b"".join(rng.randrange(1, 1 << 2040).to_bytes(EB, "big") for _ in range(k))
coeffs, pos = sc(t), list(range(1, n + 1))
pk = eng.batch_exp_fixed_base((2).to_bytes(EB, "big"), sc(n))
cm = eng.batch_exp_fixed_base((4).to_bytes(EB, "big"), coeffs)
d = eng.deal(coeffs, pos, pk, sc(n))
t8 = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
d_cm, d_pk, d_Y, d_r = t8(cm), t8(pk), t8(d["Y"]), t8(d["responses"])
d_pos = torch.tensor(pos, dtype=torch.int64, device=dev)
chal = (C.c_uint8 * EB).from_buffer_copy(d["challenge"])
torch.cuda.synchronize()
stop, loaded = [False], [0]
if LOAD:
    box = capi.ModpBox(d_cm.data_ptr(), t, d_pos.data_ptr(), d_pk.data_ptr(), d_Y.data_ptr(), d_r.data_ptr(), n, C.cast(chal, C.c_void_p), None, 0)

    def hammer():
        K = 20
        arr = (capi.ModpBox * K)(*([box] * K))
        while not stop[0]:
            verdicts = (C.c_int * K)()
            digests = (C.c_uint8 * (32 * K))()
            eng2._check(eng2.lib.mpvss_modp_verify_many(eng2.ctx, capi.MPVSS_DEVICE, arr, K, 10, 8, verdicts, C.cast(digests, C.c_void_p)), "load")
            assert all(verdicts[i] == 1 for i in range(K))
            loaded[0] += K
    th = threading.Thread(target=hammer)
    th.start()
    time.sleep(1.5)
times, on_gpu = [], []
fd0 = eng.fd_stats()
for k in range(N):
    verdict, dg = C.c_int(0), (C.c_uint8 * 32)()
    t0 = time.perf_counter()
    eng._check(eng.lib.mpvss_modp_verify_distribution(eng.ctx, capi.MPVSS_DEVICE, d_cm.data_ptr(), t, d_pos.data_ptr(), d_pk.data_ptr(), d_Y.data_ptr(),
                                                      d_r.data_ptr(), n, C.cast(chal, C.c_void_p), C.byref(verdict), dg, None, None, None),
               "verify_distribution")
    times.append(time.perf_counter() - t0)
    assert verdict.value == 1 and bytes(dg) == d["digest"], k
if LOAD:
    stop[0] = True
    th.join()
fd1 = eng.fd_stats()
ts = sorted(times)
print(f"MODP (65536, 256): {N} boxes through the synchronous call, every digest right; call wall ms min {ts[0] * 1e3:.1f} median {ts[N // 2] * 1e3:.1f} "
      f"p99 {ts[int(N * 0.99)] * 1e3:.1f} max {ts[-1] * 1e3:.1f}; forward-difference blocks {fd1[0] - fd0[0]}, FELL BACK to Horner {fd1[1] - fd0[1]}"
      + (f"; {loaded[0]} boxes verified by the second context meanwhile" if LOAD else ""))
eng.close()
if eng2:
    eng2.close()
