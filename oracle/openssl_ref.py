"""OpenSSL-backed CPU baseline for the MODP path (TEST INFRASTRUCTURE / bench.py `cpu_baseline` only -- never imported
by the product).

SURVEY 8(d) asks for a "strong CPU" line beside the plain C port: the reference's per-share operation sequence
(src/participant.rs:408-448 -> src/dleq.rs:66-84) with libcrypto's Montgomery `BN_mod_exp_mont` in the place of
num-bigint's `modpow` (src/groups/modp.rs:122-128) and `BN_mod_mul` for `ModpGroup::mul` (:130-132).  Same order of
operations as oracle/modp_ref.c: t exponentiations with the running exponent i^j mod (q-1), t products, then the two DLEQ
sides (two exponentiations and a product each).  ctypes releases the GIL around every libcrypto call, so a thread pool
scales over the host cores.
"""
import ctypes as C
import ctypes.util

Q_HEX = (
    "ffffffffffffffffc90fdaa22168c234c4c6628b80dc1cd129024e088a67cc74020bbea63b139b22514a08798e3404ddef9519b3cd3a43"
    "1b302b0a6df25f14374fe1356d6d51c245e485b576625e7ec6f44c42e9a637ed6b0bff5cb6f406b7edee386bfb5a899fa5ae9f24117c4b"
    "1fe649286651ece45b3dc2007cb8a163bf0598da48361c55d39a69163fa8fd24cf5f83655d23dca3ad961c62f356208552bb9ed5290770"
    "96966d670c354e4abc9804f1746c08ca18217c32905e462e36ce3be39e772c180e86039b2783a2ec07a28fb5c55df06f4c52c9de2bcbf6"
    "955817183995497cea956ae515d2261898fa051015728e5a8aacaa68ffffffffffffffff")


def available() -> bool:
    try:
        _load()
        return True
    except Exception:
        return False


_lib = None


def _load():
    global _lib
    if _lib is not None:
        return _lib
    name = ctypes.util.find_library("crypto")
    if not name:
        raise OSError("libcrypto not found")
    lib = C.CDLL(name)
    vp = C.c_void_p
    for fn, res, args in (
        ("BN_new", vp, []), ("BN_free", None, [vp]), ("BN_CTX_new", vp, []), ("BN_CTX_free", None, [vp]),
        ("BN_bin2bn", vp, [vp, C.c_int, vp]), ("BN_bn2binpad", C.c_int, [vp, vp, C.c_int]),
        ("BN_MONT_CTX_new", vp, []), ("BN_MONT_CTX_set", C.c_int, [vp, vp, vp]), ("BN_MONT_CTX_free", None, [vp]),
        ("BN_mod_exp_mont", C.c_int, [vp, vp, vp, vp, vp, vp]), ("BN_mod_mul", C.c_int, [vp, vp, vp, vp, vp]),
        ("BN_set_word", C.c_int, [vp, C.c_ulong]), ("BN_sub_word", C.c_int, [vp, C.c_ulong]), ("BN_copy", vp, [vp, vp]),
        ("OpenSSL_version", C.c_char_p, [C.c_int]),
    ):
        f = getattr(lib, fn)
        f.restype = res
        f.argtypes = args
    _lib = lib
    return lib


def version() -> str:
    return _load().OpenSSL_version(0).decode()


class OpenSslRef:
    """One instance per thread (a BN_CTX is not shareable)."""

    def __init__(self):
        self.lib = lib = _load()
        self.ctx = lib.BN_CTX_new()
        qb = bytes.fromhex(Q_HEX)
        self.q = lib.BN_bin2bn(qb, len(qb), None)
        self.order = lib.BN_new()
        lib.BN_copy(self.order, self.q)
        lib.BN_sub_word(self.order, 1)
        self.mont = lib.BN_MONT_CTX_new()
        assert lib.BN_MONT_CTX_set(self.mont, self.q, self.ctx) == 1
        self.tmp = [lib.BN_new() for _ in range(8)]

    def _bn(self, b: bytes):
        return self.lib.BN_bin2bn(b, len(b), None)

    def _out(self, bn) -> bytes:
        buf = (C.c_uint8 * 256)()
        assert self.lib.BN_bn2binpad(bn, buf, 256) == 256
        return bytes(buf)

    def share_work(self, commitments: bytes, position: int, y: bytes, Y: bytes, r: bytes, c: bytes):
        """X_i, a1_i, a2_i of one share, reference operation order (participant.rs:423-447)."""
        lib, ctx, q, mont = self.lib, self.ctx, self.q, self.mont
        t = len(commitments) // 256
        x, e, pw, i_bn, p1, p2, a1, a2 = self.tmp
        lib.BN_set_word(x, 1)
        lib.BN_set_word(e, 1)
        lib.BN_set_word(i_bn, position)
        for j in range(t):
            cj = self._bn(commitments[256 * j:256 * j + 256])
            lib.BN_mod_exp_mont(pw, cj, e, q, ctx, mont)          # exp(C_j, exponent)      :426-428
            lib.BN_mod_mul(x, x, pw, q, ctx)                       # mul(x, ...)             :429
            lib.BN_mod_mul(e, e, i_bn, self.order, ctx)            # exponent * i % (q-1)    :430-433
            lib.BN_free(cj)
        four = lib.BN_new()
        lib.BN_set_word(four, 4)
        bn_y, bn_Y, bn_r, bn_c = (self._bn(v) for v in (y, Y, r, c))
        lib.BN_mod_exp_mont(p1, four, bn_r, q, ctx, mont)
        lib.BN_mod_exp_mont(p2, x, bn_c, q, ctx, mont)
        lib.BN_mod_mul(a1, p1, p2, q, ctx)                         # dleq.rs:75-77
        lib.BN_mod_exp_mont(p1, bn_y, bn_r, q, ctx, mont)
        lib.BN_mod_exp_mont(p2, bn_Y, bn_c, q, ctx, mont)
        lib.BN_mod_mul(a2, p1, p2, q, ctx)                         # dleq.rs:79-81
        out = (self._out(x), self._out(a1), self._out(a2))
        for b in (four, bn_y, bn_Y, bn_r, bn_c):
            lib.BN_free(b)
        return out


# ---- secp256k1 through libcrypto's EC_POINT_mul: the curve groups' "strong CPU" line -----------------------
# The reference's per-share sequence (src/participant.rs:1404-1430 -> src/dleq.rs:66-84 with Secp256k1Group::exp / ::mul,
# src/groups/secp256k1.rs:91-107): t scalar multiplications with the running exponent i^j mod n and t additions for X_i, then
# two multiplications and an addition for each of a1 and a2.  oracle/ec_ref.c does the same with a textbook double-and-add;
# this class with OpenSSL's EC_POINT_mul.  (libcrypto has no ristretto255.)
NID_SECP256K1 = 714
SECP_N_HEX = "FFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141"
_ec_ready = False


def _load_ec():
    global _ec_ready
    lib = _load()
    if _ec_ready:
        return lib
    vp = C.c_void_p
    for fn, res, args in (
        ("EC_GROUP_new_by_curve_name", vp, [C.c_int]), ("EC_GROUP_free", None, [vp]),
        ("EC_POINT_new", vp, [vp]), ("EC_POINT_free", None, [vp]),
        ("EC_POINT_oct2point", C.c_int, [vp, vp, vp, C.c_size_t, vp]),
        ("EC_POINT_point2oct", C.c_size_t, [vp, vp, C.c_int, vp, C.c_size_t, vp]),
        ("EC_POINT_mul", C.c_int, [vp, vp, vp, vp, vp, vp]), ("EC_POINT_add", C.c_int, [vp, vp, vp, vp, vp]),
        ("EC_POINT_set_to_infinity", C.c_int, [vp, vp]), ("EC_POINT_is_at_infinity", C.c_int, [vp, vp]),
    ):
        f = getattr(lib, fn)
        f.restype = res
        f.argtypes = args
    _ec_ready = True
    return lib


def ec_available() -> bool:
    try:
        lib = _load_ec()
        g = lib.EC_GROUP_new_by_curve_name(NID_SECP256K1)
        if not g:
            return False
        lib.EC_GROUP_free(g)
        return True
    except Exception:
        return False


class OpenSslSecpRef:
    """One instance per thread.  Elements are 33-byte SEC1 compressed encodings, the identity 33 zero bytes (k256's GroupEncoding);
    scalars 32 bytes big-endian."""

    def __init__(self):
        self.lib = lib = _load_ec()
        self.ctx = lib.BN_CTX_new()
        self.group = lib.EC_GROUP_new_by_curve_name(NID_SECP256K1)
        if not self.group:
            raise OSError("libcrypto has no secp256k1")
        nb = bytes.fromhex(SECP_N_HEX)
        self.n = lib.BN_bin2bn(nb, len(nb), None)
        self.pts = [lib.EC_POINT_new(self.group) for _ in range(6)]
        self.bns = [lib.BN_new() for _ in range(3)]

    def _point(self, pt, enc: bytes):
        if enc == bytes(33):
            assert self.lib.EC_POINT_set_to_infinity(self.group, pt) == 1
        elif self.lib.EC_POINT_oct2point(self.group, pt, enc, len(enc), self.ctx) != 1:
            raise ValueError("invalid encoding")

    def _enc(self, pt) -> bytes:
        if self.lib.EC_POINT_is_at_infinity(self.group, pt):
            return bytes(33)
        buf = (C.c_uint8 * 33)()
        assert self.lib.EC_POINT_point2oct(self.group, pt, 2, buf, 33, self.ctx) == 33      # POINT_CONVERSION_COMPRESSED
        return bytes(buf)

    def share_work(self, commitments: bytes, position: int, y: bytes, Y: bytes, r: bytes, c: bytes):
        """X_i, a1_i, a2_i of one share, reference operation order (participant.rs:1404-1430)."""
        lib, ctx, grp = self.lib, self.ctx, self.group
        x, cj, pw, p1, p2, acc = self.pts
        e, i_bn, tmp = self.bns
        t = len(commitments) // 33
        assert lib.EC_POINT_set_to_infinity(grp, x) == 1
        lib.BN_set_word(e, 1)
        lib.BN_set_word(i_bn, position % (1 << 64))
        for j in range(t):
            self._point(cj, commitments[33 * j:33 * j + 33])
            lib.EC_POINT_mul(grp, pw, None, cj, e, ctx)            # exp(C_j, exponent)          :1409-1411
            lib.EC_POINT_add(grp, x, x, pw, ctx)                   # mul(x, ...)                 :1412
            lib.BN_mod_mul(e, e, i_bn, self.n, ctx)                # exponent * i                :1413-1416
        bn_r = lib.BN_bin2bn(r, 32, None)
        bn_c = lib.BN_bin2bn(c, 32, None)
        outs = [self._enc(x)]
        for base in (None, y):                                     # a1 = r G + c X, a2 = r y + c Y   dleq.rs:75-81
            if base is None:
                lib.EC_POINT_mul(grp, p1, bn_r, None, None, ctx)
                lib.EC_POINT_mul(grp, p2, None, x, bn_c, ctx)
            else:
                self._point(cj, base)
                lib.EC_POINT_mul(grp, p1, None, cj, bn_r, ctx)
                self._point(cj, Y)
                lib.EC_POINT_mul(grp, p2, None, cj, bn_c, ctx)
            lib.EC_POINT_add(grp, acc, p1, p2, ctx)
            outs.append(self._enc(acc))
        lib.BN_free(bn_r)
        lib.BN_free(bn_c)
        return tuple(outs)
