"""ctypes loader for oracle/ec_ref.c (TEST INFRASTRUCTURE / bench cpu_baseline only)."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "libec_ref.so")
GROUP_ID = {"secp256k1": 1, "ristretto255": 2}
ENC = {1: 33, 2: 32}


def load():
    if not os.path.exists(_LIB):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    lib = C.CDLL(_LIB)
    vp = C.c_void_p
    lib.ec_ref_verify_share_work.argtypes = [C.c_int, vp, C.c_size_t, C.c_int64, vp, vp, vp, vp, vp, vp, vp]
    lib.ec_ref_verify_share_work.restype = C.c_int
    lib.ec_ref_exp.argtypes = [C.c_int, vp, vp, vp]
    lib.ec_ref_exp.restype = C.c_int
    lib.ec_ref_init.restype = None
    lib.ec_ref_init()
    return lib


def _b(b):
    return (C.c_uint8 * max(len(b), 1)).from_buffer_copy(b if b else b"\0")


class EcRef:
    def __init__(self):
        self.lib = load()

    def exp(self, group: int, point: bytes, scalar: bytes) -> bytes:
        out = (C.c_uint8 * ENC[group])()
        if self.lib.ec_ref_exp(group, _b(point), _b(scalar), out) != 0:
            raise ValueError("invalid encoding")
        return bytes(out)

    def share_work(self, group: int, commitments: bytes, position: int, y: bytes, Y: bytes, r: bytes, c: bytes):
        """X_i, a1_i, a2_i of one share in the reference's operation order (participant.rs:1404-1430 / 1847-1873)."""
        L = ENC[group]
        X, a1, a2 = ((C.c_uint8 * L)() for _ in range(3))
        rc = self.lib.ec_ref_verify_share_work(group, _b(commitments), len(commitments) // L, position, _b(y), _b(Y), _b(r),
                                               _b(c), X, a1, a2)
        if rc != 0:
            raise ValueError("invalid encoding")
        return bytes(X), bytes(a1), bytes(a2)
