#!/usr/bin/env python3
"""A few LONE mpvss_modp_verify_distribution calls at the headline shape (nothing else on the GPU), for a rocprofv3 kernel trace:
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/lone -- python3 tools/lone_verify_trace.py
  python3 tools/lone_verify_trace.py --timeline gpurun_out/lone/*/*kernel_trace.csv     (kernels of the last call, ms from its start)"""
import ctypes as C
import os
import random
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 2 and sys.argv[1] == "--timeline":
    import csv
    rows = list(csv.DictReader(open(sys.argv[2])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    last = int(rows[-1]["End_Timestamp"])
    window = float(sys.argv[3]) if len(sys.argv) > 3 else 100.0
    sel = [r for r in rows if (last - int(r["Start_Timestamp"])) / 1e6 < window]
    t0 = int(sel[0]["Start_Timestamp"])
    for r in sel:
        s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
        if e - s > 0.15:
            wg = int(r["Workgroup_Size_X"])
            print(f"{s:8.2f} {e:8.2f} {e - s:7.2f} q{r['Queue_Id']:>2} {r['Kernel_Name'][:36]:36} wgs {int(r['Grid_Size_X']) // wg}x{r['Grid_Size_Y']} x{wg} "
                  f"vgpr {r['VGPR_Count']}+{r['Accum_VGPR_Count']} lds {r['LDS_Block_Size']}")
    sys.exit(0)

import torch  # noqa: E402,F401
from mpvss_rs_amd import Engine, capi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
t = int(sys.argv[2]) if len(sys.argv) > 2 else 256
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
EB = 256
eng = Engine(0)
rng = random.Random(1)
pos = list(range(1, n + 1))
sc = lambda k: b"".join(rng.randrange(1, 1 << 2040).to_bytes(EB, "big") for _ in range(k))
coeffs, wit = sc(t), sc(n)
pk = eng.batch_exp_fixed_base((2).to_bytes(EB, "big"), sc(n))
cm = eng.batch_exp_fixed_base((4).to_bytes(EB, "big"), coeffs)
box = eng.deal(coeffs, pos, pk, wit)
bufs = [(C.c_uint8 * len(b)).from_buffer_copy(b) for b in (cm, pk, box["Y"], box["responses"], box["challenge"])]
parr = (C.c_int64 * n)(*pos)
verdict, dg = C.c_int(0), (C.c_uint8 * 32)()
for k in range(reps + 1):
    t0 = time.perf_counter()
    eng._check(eng.lib.mpvss_modp_verify_distribution(eng.ctx, capi.MPVSS_HOST, bufs[0], t, C.cast(parr, C.c_void_p), bufs[1], bufs[2], bufs[3], n,
                                                      C.cast(bufs[4], C.c_void_p), C.byref(verdict), dg, None, None, None), "verify_distribution")
    print(f"call {k}: {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
    assert verdict.value == 1 and bytes(dg) == box["digest"]
    time.sleep(0.05)
eng.close()
