#!/usr/bin/env python3
"""The dominant kernel alone on the chip: a2 = y^r Y^c of n = 65536 shares through the synchronous mpvss_modp_dleq_commitments
(a1 through the comb first, then the 64-entry tables of y, then k_modp_dual_exp_w6[_pair]), five times; prints the kernel's
milliseconds (HIP events around the launch) per repetition.  MPVSS_HIP_LIB selects a variant library (tools/build_lib_variant.sh)."""
import ctypes as C
import os
import random
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from mpvss_rs_amd import capi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
eng = capi.Engine(0)
rng = random.Random(5)
dev = torch.device("cuda", 0)
rnd = lambda: torch.frombuffer(bytearray(rng.randbytes(n * 256)), dtype=torch.uint8).to(dev)
X, y, Y, r = rnd(), rnd(), rnd(), rnd()
for t in (X, y, Y, r):
    t.view(n, 256)[:, 0] = 0x7F          # below q
c = (C.c_uint8 * 256).from_buffer_copy(bytes(224) + rng.randbytes(32))
g = (C.c_uint8 * 256).from_buffer_copy(bytes(255) + b"\x04")
o1, o2 = torch.empty(n * 256, dtype=torch.uint8, device=dev), torch.empty(n * 256, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
vp = lambda t: C.c_void_p(t.data_ptr())
ms = []
for rep in range(6):
    eng._check(eng.lib.mpvss_modp_dleq_commitments(eng.ctx, capi.MPVSS_DEVICE, C.cast(g, C.c_void_p), vp(X), vp(y), vp(Y), vp(r),
                                                   C.cast(c, C.c_void_p), 0, n, vp(o1), vp(o2)), "dleq_commitments")
    ms.append((eng.kernel_ms(3) / max(eng.kernel_launches(3), 1), eng.kernel_ms(1), eng.kernel_ms(2)))
print(os.environ.get("MPVSS_HIP_LIB", "default").split("/")[-1], "a2 ms:", " ".join(f"{m[0]:.2f}" for m in ms[1:]),
      "| a1 (comb) ms:", f"{ms[-1][1]:.2f}", "| tables ms:", f"{ms[-1][2]:.2f}")
eng.close()
