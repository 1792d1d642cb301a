import ctypes as C, os, random, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from mpvss_rs_amd import Engine, capi
n, t, EB = 65536, 256, 256
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
def call():
    t0 = time.perf_counter()
    eng._check(eng.lib.mpvss_modp_verify_distribution(eng.ctx, capi.MPVSS_HOST, bufs[0], t, C.cast(parr, C.c_void_p), bufs[1], bufs[2], bufs[3], n,
                                                      C.cast(bufs[4], C.c_void_p), C.byref(verdict), dg, None, None, None), "vd")
    assert verdict.value == 1 and bytes(dg) == box["digest"]
    return (time.perf_counter() - t0) * 1e3
plain = [call() for _ in range(4)]
eng.set_key_cache_lru(1, 2)
warm = [call() for _ in range(3)]
cached = [call() for _ in range(5)]
print(f"TMP_ROW_KEYS={os.environ.get('MPVSS_TMP_ROW_KEYS','0')} plain best {min(plain[1:]):.1f} ms; cache warm-up {[round(x,1) for x in warm]}; cached best {min(cached):.1f} ms, all {[round(x,1) for x in cached]}")
