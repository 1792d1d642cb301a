"""ctypes loader for oracle/modp_ref.c (TEST INFRASTRUCTURE / bench cpu_baseline only)."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "libmodp_ref.so")


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def load():
    if not os.path.exists(_LIB):
        build()
    lib = C.CDLL(_LIB)
    vp, sz = C.c_void_p, C.c_size_t
    lib.ref_modpow.argtypes = [vp, vp, vp]
    lib.ref_mulmod_q.argtypes = [vp, vp, vp]
    lib.ref_verify_share_work.argtypes = [vp, sz, C.c_int64, vp, vp, vp, vp, vp, vp, vp]
    lib.ref_verify_distribution.argtypes = [vp, sz, vp, vp, vp, vp, sz, vp, vp, vp, vp, vp]
    lib.ref_verify_distribution.restype = C.c_int
    lib.ref_sha256.argtypes = [vp, sz, vp]
    for f in (lib.ref_modpow, lib.ref_mulmod_q, lib.ref_verify_share_work, lib.ref_sha256):
        f.restype = None
    return lib


def _b(b):
    return (C.c_uint8 * max(len(b), 1)).from_buffer_copy(b if b else b"\0")


class ModpRef:
    def __init__(self):
        self.lib = load()

    def modpow(self, base: int, exp: int) -> int:
        out = (C.c_uint8 * 256)()
        self.lib.ref_modpow(_b(base.to_bytes(256, "big")), _b(exp.to_bytes(256, "big")), out)
        return int.from_bytes(bytes(out), "big")

    def mulmod(self, a: int, b: int) -> int:
        out = (C.c_uint8 * 256)()
        self.lib.ref_mulmod_q(_b(a.to_bytes(256, "big")), _b(b.to_bytes(256, "big")), out)
        return int.from_bytes(bytes(out), "big")

    def share_work(self, commitments: bytes, position: int, y: bytes, Y: bytes, r: bytes, c: bytes):
        X, a1, a2 = ((C.c_uint8 * 256)() for _ in range(3))
        self.lib.ref_verify_share_work(_b(commitments), len(commitments) // 256, position, _b(y), _b(Y), _b(r), _b(c),
                                       X, a1, a2)
        return bytes(X), bytes(a1), bytes(a2)

    def verify_distribution(self, flat: dict, dump=False):
        n, t = flat["n"], flat["t"]
        pos = (C.c_int64 * max(n, 1))(*flat["positions"])
        digest = (C.c_uint8 * 32)()
        outs = [(C.c_uint8 * max(n * 256, 1))() for _ in range(3)] if dump else [None] * 3
        v = self.lib.ref_verify_distribution(_b(flat["commitments"]), t, pos, _b(flat["publickeys"]),
                                             _b(flat["shares"]), _b(flat["responses"]), n, _b(flat["challenge"]),
                                             digest, *outs)
        res = {"verdict": bool(v), "digest": bytes(digest)}
        if dump:
            res.update(X=bytes(outs[0])[: n * 256], a1=bytes(outs[1])[: n * 256], a2=bytes(outs[2])[: n * 256])
        return res

    def sha256(self, data: bytes) -> bytes:
        out = (C.c_uint8 * 32)()
        self.lib.ref_sha256(_b(data), len(data), out)
        return bytes(out)
