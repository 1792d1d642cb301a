#!/usr/bin/env python3
"""Soak of the curve groups' lone-box path (quad-lane stage pipelines, hand-over through HBM): the same box verified N times
through the synchronous call, every digest checked, the slowest call reported (a stage that times out costs 2 s and sends
the box down Horner's rule; a stale hand-over would show as a wrong digest).
  python3 tools/ec_lone_soak.py [N=200] [load]     load: a second context keeps 16 boxes of the same group in flight meanwhile
                                                   (the pipelines' waves then share their SIMDs with wide launches of another process's worth)"""
import ctypes as C
import os
import random
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from mpvss_rs_amd import capi  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
LOAD = len(sys.argv) > 2 and sys.argv[2] == "load"
eng = capi.Engine(0)
eng2 = capi.Engine(0) if LOAD else None
dev = torch.device("cuda", 0)
for name in ("secp256k1", "ristretto255"):
    cfg = bench.EC[name]
    gid, order = cfg["gid"], cfg["order"]
    n, t = 65536, 256
    sb = (lambda k: k.to_bytes(32, "big")) if cfg["be"] else (lambda k: k.to_bytes(32, "little"))
    rng = random.Random(bench.SEED + gid)
    coeffs = [rng.randrange(order) for _ in range(t)]
    privs = [rng.randrange(1, order) for _ in range(n)]
    wits = [rng.randrange(1, order) for _ in range(n)]
    positions = list(range(1, n + 1))
    pv = capi.poly_eval(gid, b"".join(map(sb, coeffs)), positions)
    cm = eng.ec_batch_exp_generator(gid, b"".join(map(sb, coeffs)))
    pks = eng.ec_batch_exp_generator(gid, b"".join(map(sb, privs)))
    d = eng.ec_distribute(gid, cm, positions, pks, pv, b"".join(map(sb, wits)))
    cbytes = capi.ec_hash_to_scalar(gid, d["digest"])
    responses = capi.dleq_responses(gid, b"".join(map(sb, wits)), pv, cbytes)
    dbuf = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    d_cm, d_pk, d_Y, d_r = dbuf(cm), dbuf(pks), dbuf(d["Y"]), dbuf(responses)
    d_pos = torch.tensor(positions, dtype=torch.int64, device=dev)
    chal = (C.c_uint8 * 32).from_buffer_copy(cbytes)
    torch.cuda.synchronize()
    stop, loaded = [False], [0]
    if LOAD:
        import threading
        box = capi.EcBox(d_cm.data_ptr(), t, d_pos.data_ptr(), d_pk.data_ptr(), d_Y.data_ptr(), d_r.data_ptr(), n, C.cast(chal, C.c_void_p))

        def hammer():
            K = 32
            arr = (capi.EcBox * K)(*([box] * K))
            while not stop[0]:
                verdicts = (C.c_int * K)()
                digests = (C.c_uint8 * (32 * K))()
                eng2._check(eng2.lib.mpvss_ec_verify_many(eng2.ctx, gid, capi.MPVSS_DEVICE, arr, K, 16, 6, verdicts, C.cast(digests, C.c_void_p)), "load")
                assert all(verdicts[i] == 1 for i in range(K))
                loaded[0] += K
        th = threading.Thread(target=hammer)
        th.start()
        time.sleep(1.0)
    times, on_gpu = [], []
    fd0 = eng.fd_stats()
    for k in range(N):
        verdict = C.c_int(0)
        dg = (C.c_uint8 * 32)()
        t0 = time.perf_counter()
        eng._check(eng.lib.mpvss_ec_verify_distribution(eng.ctx, gid, capi.MPVSS_DEVICE, d_cm.data_ptr(), t, d_pos.data_ptr(), d_pk.data_ptr(),
                                                        d_Y.data_ptr(), d_r.data_ptr(), n, C.cast(chal, C.c_void_p), C.byref(verdict),
                                                        C.cast(dg, C.c_void_p), None, None, None), "ec_verify_distribution")
        times.append(time.perf_counter() - t0)
        st_ = eng.pipeline_stats(reset=True)
        on_gpu.append(st_["enqueue_ms"] + st_["wait_ms"])          # the box on the GPU: first launch to last copy
        assert verdict.value == 1 and bytes(dg) == d["digest"], (name, k)
    if LOAD:
        stop[0] = True
        th.join()
    ts, gs = sorted(times), sorted(on_gpu)
    fd1 = eng.fd_stats()
    print(f"{name}: {N} boxes alone, every digest right; call wall ms min {ts[0] * 1e3:.2f} median {ts[N // 2] * 1e3:.2f} "
          f"p99 {ts[int(N * 0.99)] * 1e3:.2f} max {ts[-1] * 1e3:.2f}; box on the GPU ms median {gs[N // 2]:.2f} p99 {gs[int(N * 0.99)]:.2f} "
          f"max {gs[-1]:.2f}; X paths gated {fd1[0] - fd0[0]}, FELL BACK to Horner {fd1[1] - fd0[1]}"
          + (f"; {loaded[0]} boxes verified by the second context meanwhile" if LOAD else ""))
    slow = [(k, round(times[k] * 1e3, 1), round(on_gpu[k], 1)) for k in range(N) if times[k] > 10 * ts[N // 2]]
    if slow:
        print(f"  calls slower than 10 x median (index, wall ms, on-GPU ms): {slow[:20]}")
eng.close()
if eng2:
    eng2.close()
