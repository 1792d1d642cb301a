#!/usr/bin/env python3
"""Builds ONE curve-group box of the headline shape (n=65536, t=256) exactly as bench.py's bench_ec does and verifies it K times
through mpvss_ec_verify_many with bench.py's own depth and hash threads (MPVSS_BENCH_EC_DEPTH = 16 boxes in flight, X paths of 16
boxes per launch, 6 hash threads: the call bench.py times) -- the program the PMC passes of tools/run_profiles_ec.sh run twice per
group (K = 0 and K = 32): the difference of the summed counters is what K verifications cost, free of the set-up kernels.
  python3 tools/ec_box_for_pmc.py <secp256k1|ristretto255> <K>"""
import ctypes as C
import os
import random
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from mpvss_rs_amd import capi  # noqa: E402

name, K = sys.argv[1], int(sys.argv[2])
cfg = bench.EC[name]
gid, L, order = cfg["gid"], cfg["enc"], cfg["order"]
n, t = 65536, 256
sb = (lambda k: k.to_bytes(32, "big")) if cfg["be"] else (lambda k: k.to_bytes(32, "little"))
rng = random.Random(bench.SEED + gid)
coeffs = [rng.randrange(order) for _ in range(t)]
privs = [rng.randrange(1, order) for _ in range(n)]
wits = [rng.randrange(1, order) for _ in range(n)]
positions = list(range(1, n + 1))
eng = capi.Engine(0)
pv = capi.poly_eval(gid, b"".join(map(sb, coeffs)), positions)
cm = eng.ec_batch_exp_generator(gid, b"".join(map(sb, coeffs)))
pks = eng.ec_batch_exp_generator(gid, b"".join(map(sb, privs)))
d = eng.ec_distribute(gid, cm, positions, pks, pv, b"".join(map(sb, wits)))
cbytes = capi.ec_hash_to_scalar(gid, d["digest"])
responses = capi.dleq_responses(gid, b"".join(map(sb, wits)), pv, cbytes)
dev = torch.device("cuda", 0)
dbuf = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
d_cm, d_pk, d_Y, d_r = dbuf(cm), dbuf(pks), dbuf(d["Y"]), dbuf(responses)
d_pos = torch.tensor(positions, dtype=torch.int64, device=dev)
chal = (C.c_uint8 * 32).from_buffer_copy(cbytes)
torch.cuda.synchronize()
if K > 0:
    box = capi.EcBox(d_cm.data_ptr(), t, d_pos.data_ptr(), d_pk.data_ptr(), d_Y.data_ptr(), d_r.data_ptr(), n, C.cast(chal, C.c_void_p))
    arr = (capi.EcBox * K)(*([box] * K))
    verdicts = (C.c_int * K)()
    digests = (C.c_uint8 * (32 * K))()
    depth, threads = int(os.environ.get("MPVSS_BENCH_EC_DEPTH", "16")), int(os.environ.get("MPVSS_BENCH_EC_HASH_THREADS", "6"))
    for _ in range(int(os.environ.get("MPVSS_BOX_REPEAT", "1"))):      # (tools/ec_lone_box_trace.sh: the same K boxes again, warm)
        eng._check(eng.lib.mpvss_ec_verify_many(eng.ctx, gid, capi.MPVSS_DEVICE, arr, K, depth, threads, verdicts, C.cast(digests, C.c_void_p)), "ec_verify_many")
    assert all(verdicts[i] == 1 for i in range(K)) and bytes(digests)[:32] == d["digest"]
print(name, "verified", K)
eng.close()
