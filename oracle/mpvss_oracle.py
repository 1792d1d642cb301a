"""CPU oracle for the mpvss-rs group-exponentiation / DLEQ hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product path (mpvss_rs_amd/, the
C-ABI library, bench.py's timed GPU region) may import this module; it is the
checker used by tests/, __graft_entry__.smoke() and tests/golden/make_golden.py.

This is a restatement, in plain Python integers + hashlib, of the reference's
algorithm *as written* (operation order, reductions, byte framing) so that its
outputs are what the Rust crate would produce on the same inputs with the same
randomness.  All citations are file:line in the reference checkout
(AlexiaChen/mpvss-rs v2.0.0).

PARITY PINNING.  The reference cannot be compiled here (no Rust toolchain) and
its arithmetic lives in un-vendored crates (num-bigint 0.2, k256 0.13,
curve25519-dalek 4, sha2 0.10).  The oracle is therefore pinned against
  * every known-answer value the reference's own tests hold for this path
    (polynomial.rs:75-125, util.rs:84-190, dleq.rs:380-403, modp.rs:243-259,
    secp256k1.rs:197-235,270-272, ristretto255.rs:378-401,620-638,680-682) --
    see tests/test_oracle_reference_kats.py,
  * published vectors of the third-party algorithms (SEC2 secp256k1 k*G values,
    RFC 9496 appendix A ristretto255 generator multiples, FIPS 180-4 SHA vectors
    via hashlib),
  * SURVEY.md appendix B values (computed independently of this file).
For exp/mul outputs, DLEQ commitments, transcripts, challenges and verdicts the
reference itself holds no golden vectors: parity there rests on the canonical
encoding argument (every hashed/compared value is a unique canonical byte string
of a mathematically defined group element), i.e. "parity unpinned by reference
goldens" for those values -- stated in DESIGN.md as well.
"""
from __future__ import annotations

import hashlib
from typing import Dict, List, Optional, Sequence, Tuple

# --------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------


def _be_min(v: int) -> bytes:
    """num-bigint BigUint::to_bytes_be: minimal length, zero -> [0]."""
    if v < 0:
        raise ValueError("negative value has no BigUint encoding (reference unwraps -> panic)")
    return v.to_bytes(max(1, (v.bit_length() + 7) // 8), "big")


def framed(b: bytes) -> bytes:
    """dleq.rs:58-61  u64 big-endian length prefix followed by the bytes."""
    return len(b).to_bytes(8, "big") + b


def sha256(b: bytes) -> bytes:
    return hashlib.sha256(b).digest()


# --------------------------------------------------------------------------
# Groups (group.rs:24-124 contract)
# --------------------------------------------------------------------------

MODP_Q_HEX = (
    "ffffffffffffffffc90fdaa22168c234c4c6628b80dc1cd129024e088a67cc74"
    "020bbea63b139b22514a08798e3404ddef9519b3cd3a431b302b0a6df25f1437"
    "4fe1356d6d51c245e485b576625e7ec6f44c42e9a637ed6b0bff5cb6f406b7ed"
    "ee386bfb5a899fa5ae9f24117c4b1fe649286651ece45b3dc2007cb8a163bf05"
    "98da48361c55d39a69163fa8fd24cf5f83655d23dca3ad961c62f356208552bb"
    "9ed529077096966d670c354e4abc9804f1746c08ca18217c32905e462e36ce3b"
    "e39e772c180e86039b2783a2ec07a28fb5c55df06f4c52c9de2bcbf695581718"
    "3995497cea956ae515d2261898fa051015728e5a8aacaa68ffffffffffffffff"
)


class ModpGroup:
    """groups/modp.rs:29-197 -- RFC 3526 group 14."""

    name = "modp2048"
    elem_len = 256
    scalar_len = 256

    def __init__(self):
        self.q = int(MODP_Q_HEX, 16)          # modp.rs:47-58
        self.g = (self.q - 1) // 2              # subgroup order, modp.rs:59
        self.G = 2                              # modp.rs:64
        self.g_gen = pow(2, 2, self.q)          # modp.rs:65-66
        self.q_minus_1 = self.q - 1             # modp.rs:67

    # --- constants
    def order(self):
        return self.q_minus_1                   # modp.rs:101-103

    def subgroup_order(self):
        return self.g                           # modp.rs:105-107

    def generator(self):
        return self.G

    def subgroup_generator(self):
        return self.g_gen

    def identity(self):
        return 1

    # --- element ops
    def exp(self, base: int, scalar: int) -> int:
        if scalar < 0:
            raise ValueError("negative exponent: num-bigint modpow panics")
        return pow(base, scalar, self.q)        # modp.rs:122-128

    def mul(self, a: int, b: int) -> int:
        return (a * b) % self.q                 # modp.rs:130-132

    def element_inverse(self, x: int) -> Optional[int]:
        return mod_inverse(x, self.q)           # modp.rs:138-140

    def elements_equal(self, a, b):
        return a == b

    # --- scalar ops
    def scalar_mul(self, a: int, b: int) -> int:
        return (a * b) % self.q_minus_1         # modp.rs:180-182

    def scalar_sub(self, a: int, b: int) -> int:
        diff = a - b                            # modp.rs:184-192
        return diff + self.q_minus_1 if diff < 0 else diff % self.q_minus_1

    def scalar_inverse(self, x: int) -> Optional[int]:
        return mod_inverse(x, self.q_minus_1)   # modp.rs:134-136

    def scalar_from_u64(self, v: int) -> int:
        return v                                # BigInt::from(position)

    def scalar_from_bigint(self, v: int) -> int:
        return v                                # coefficients used as-is

    def scalar_to_int(self, s: int) -> int:
        return s

    def group_order_int(self) -> int:
        return self.q_minus_1                   # participant.rs:175 (order())

    def hash_to_scalar(self, data: bytes) -> int:
        return int.from_bytes(sha256(data), "big") % self.g   # modp.rs:142-148

    # --- encodings
    def element_to_bytes(self, e: int) -> bytes:
        return _be_min(e)                       # modp.rs:150-152

    def bytes_to_element(self, b: bytes) -> Optional[int]:
        return int.from_bytes(b, "big")         # modp.rs:154-156 (no validation)

    def scalar_to_bytes(self, s: int) -> bytes:
        return _be_min(s)                       # modp.rs:158-160

    def generate_public_key(self, priv: int) -> int:
        return self.exp(self.G, priv)           # modp.rs:176-178

    # U masks (participant.rs:267-272 / 512-517): SHA256(bytes(G^s)) mod q
    def secret_mask(self, g_s: int) -> int:
        return int.from_bytes(sha256(self.element_to_bytes(g_s)), "big") % self.q

    # fixed-width boundary encodings (include/mpvss_hip.h)
    def element_to_fixed(self, e: int) -> bytes:
        return e.to_bytes(256, "big")

    def element_from_fixed(self, b: bytes) -> int:
        return int.from_bytes(b, "big")

    def scalar_to_fixed(self, s: int) -> bytes:
        return s.to_bytes(256, "big")

    def scalar_from_fixed(self, b: bytes) -> int:
        return int.from_bytes(b, "big")


# ---- secp256k1 -----------------------------------------------------------

SECP_P = 2**256 - 2**32 - 977
SECP_N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
SECP_GX = 0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798
SECP_GY = 0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8


class Secp256k1Group:
    """groups/secp256k1.rs:38-189.  Element = affine (x, y) or None (identity)."""

    name = "secp256k1"
    elem_len = 33
    scalar_len = 32

    def __init__(self):
        self.p = SECP_P
        self.n = SECP_N                         # secp256k1.rs:47-51
        self.Gpt = (SECP_GX, SECP_GY)

    def generator(self):
        return self.Gpt

    def subgroup_generator(self):
        return self.Gpt                         # secp256k1.rs:82-85

    def identity(self):
        return None                             # secp256k1.rs:87-89

    def _add(self, P, Q):
        p = self.p
        if P is None:
            return Q
        if Q is None:
            return P
        x1, y1 = P
        x2, y2 = Q
        if x1 == x2:
            if (y1 + y2) % p == 0:
                return None
            lam = (3 * x1 * x1) * pow(2 * y1, -1, p) % p
        else:
            lam = (y2 - y1) * pow(x2 - x1, -1, p) % p
        x3 = (lam * lam - x1 - x2) % p
        y3 = (lam * (x1 - x3) - y1) % p
        return (x3, y3)

    def exp(self, base, scalar: int):
        """secp256k1.rs:91-100: scalar * base, returned affine."""
        k = scalar % self.n
        R = None
        A = base
        while k:
            if k & 1:
                R = self._add(R, A)
            A = self._add(A, A)
            k >>= 1
        return R

    def mul(self, a, b):
        return self._add(a, b)                  # secp256k1.rs:102-107

    def element_inverse(self, x):
        if x is None:
            return None
        return (x[0], (-x[1]) % self.p)         # secp256k1.rs:114-119

    def elements_equal(self, a, b):
        return a == b

    def scalar_mul(self, a, b):
        return (a * b) % self.n                 # secp256k1.rs:173-176

    def scalar_sub(self, a, b):
        return (a - b) % self.n                 # secp256k1.rs:178-181

    def scalar_inverse(self, x):
        if x % self.n == 0:
            return None
        return pow(x, -1, self.n)               # secp256k1.rs:109-112

    def scalar_from_u64(self, v: int) -> int:
        return (v & 0xFFFFFFFFFFFFFFFF) % self.n   # Scalar::from(position as u64)

    def scalar_from_bigint(self, v: int) -> int:
        """participant.rs:1134-1143: BE bytes right-aligned (or first 32 bytes if
        longer) -> Scalar::from_repr(..).unwrap() (panics when >= n)."""
        b = _be_min(v)
        fb = b.rjust(32, b"\0") if len(b) < 32 else b[:32]
        s = int.from_bytes(fb, "big")
        if s >= self.n:
            raise ValueError("Scalar::from_repr rejects non-canonical value (reference panics)")
        return s

    def scalar_to_int(self, s):
        return s

    def group_order_int(self):
        return self.n                           # order_as_bigint, secp256k1.rs:186-188

    def hash_to_scalar(self, data: bytes) -> int:
        return int.from_bytes(sha256(data), "big") % self.n   # secp256k1.rs:121-131

    def element_to_bytes(self, e) -> bytes:
        if e is None:
            return bytes(33)                    # k256 GroupEncoding of identity (spec-derived)
        x, y = e
        return bytes([2 + (y & 1)]) + x.to_bytes(32, "big")   # secp256k1.rs:133-136

    def bytes_to_element(self, b: bytes):
        if len(b) != 33:
            return None                         # secp256k1.rs:139-141
        if b == bytes(33):
            return ("identity",)                # caller must special-case; see decode_element
        if b[0] not in (2, 3):
            return None
        x = int.from_bytes(b[1:], "big")
        if x >= self.p:
            return None
        y2 = (pow(x, 3, self.p) + 7) % self.p
        y = pow(y2, (self.p + 1) // 4, self.p)
        if y * y % self.p != y2:
            return None
        if (y & 1) != (b[0] & 1):
            y = self.p - y
        return (x, y)

    def decode_element(self, b: bytes):
        """(ok, element) wrapper that can represent the identity."""
        r = self.bytes_to_element(b)
        if r is None:
            return False, None
        if r == ("identity",):
            return True, None
        return True, r

    def scalar_to_bytes(self, s: int) -> bytes:
        return s.to_bytes(32, "big")            # secp256k1.rs:154-156

    def generate_public_key(self, priv):
        return self.exp(self.Gpt, priv)         # secp256k1.rs:168-171

    def secret_mask(self, g_s) -> int:
        """participant.rs:1246-1260: digest -> from_repr (panics if >= n) -> % n."""
        h = int.from_bytes(sha256(self.element_to_bytes(g_s)), "big")
        if h >= self.n:
            raise ValueError("Scalar::from_repr(digest).unwrap() panics (p ~ 2^-128)")
        return h % self.n

    element_to_fixed = element_to_bytes

    def element_from_fixed(self, b: bytes):
        ok, e = self.decode_element(b)
        if not ok:
            raise ValueError("invalid SEC1 point")
        return e

    def scalar_to_fixed(self, s):
        return s.to_bytes(32, "big")

    def scalar_from_fixed(self, b):
        return int.from_bytes(b, "big")


# ---- ristretto255 (RFC 9496) ----------------------------------------------

ED_P = 2**255 - 19
ED_L = 2**252 + 27742317777372353535851937790883648493
ED_D = (-121665 * pow(121666, -1, ED_P)) % ED_P
SQRT_M1 = pow(2, (ED_P - 1) // 4, ED_P)
INVSQRT_A_MINUS_D = None  # filled below


def _is_neg(x: int) -> bool:
    return (x % ED_P) & 1 == 1


def _ct_abs(x: int) -> int:
    x %= ED_P
    return ED_P - x if x & 1 else x


def _sqrt_ratio_m1(u: int, v: int) -> Tuple[bool, int]:
    """RFC 9496 section 4.2."""
    p = ED_P
    u %= p
    v %= p
    v3 = v * v % p * v % p
    v7 = v3 * v3 % p * v % p
    r = u * v3 % p * pow(u * v7 % p, (p - 5) // 8, p) % p
    check = v * r % p * r % p
    correct = check == u
    flipped = check == (-u) % p
    flipped_i = check == (-u * SQRT_M1) % p
    if flipped or flipped_i:
        r = r * SQRT_M1 % p
    r = _ct_abs(r)
    return (correct or flipped), r


_ok, INVSQRT_A_MINUS_D = _sqrt_ratio_m1(1, (-1 - ED_D) % ED_P)
assert _ok

# Ed25519 basepoint (RFC 8032); the ristretto255 generator is its coset.
_ED_BY = 4 * pow(5, -1, ED_P) % ED_P
_ED_BX = 15112221349535400772501151409588531511454012693041857206046113283949847762202
ED_BASE = (_ED_BX, _ED_BY, 1, _ED_BX * _ED_BY % ED_P)


class Ristretto255Group:
    """groups/ristretto255.rs:45-253.  Element = extended Edwards (X, Y, Z, T)."""

    name = "ristretto255"
    elem_len = 32
    scalar_len = 32

    def __init__(self):
        self.p = ED_P
        self.l = ED_L                           # ristretto255.rs:55-59

    def generator(self):
        return ED_BASE                          # ristretto255.rs:148-150

    def subgroup_generator(self):
        return ED_BASE                          # ristretto255.rs:152-155

    def identity(self):
        return (0, 1, 1, 0)                     # ristretto255.rs:157-159

    def _add(self, P, Q):
        p = self.p
        X1, Y1, Z1, T1 = P
        X2, Y2, Z2, T2 = Q
        A = (Y1 - X1) * (Y2 - X2) % p
        B = (Y1 + X1) * (Y2 + X2) % p
        C = 2 * ED_D * T1 % p * T2 % p
        D = 2 * Z1 * Z2 % p
        E, F, G, H = (B - A) % p, (D - C) % p, (D + C) % p, (B + A) % p
        return (E * F % p, G * H % p, F * G % p, E * H % p)

    def exp(self, base, scalar: int):
        k = scalar % self.l                     # ristretto255.rs:161-170
        R = self.identity()
        A = base
        while k:
            if k & 1:
                R = self._add(R, A)
            A = self._add(A, A)
            k >>= 1
        return R

    def mul(self, a, b):
        return self._add(a, b)                  # ristretto255.rs:172-177

    def element_inverse(self, x):
        X, Y, Z, T = x
        return ((-X) % self.p, Y, Z, (-T) % self.p)   # ristretto255.rs:189-194

    def elements_equal(self, a, b):
        # RFC 9496 4.3.3
        X1, Y1, _, _ = a
        X2, Y2, _, _ = b
        p = self.p
        return (X1 * Y2 - Y1 * X2) % p == 0 or (Y1 * Y2 - X1 * X2) % p == 0

    def scalar_mul(self, a, b):
        return (a * b) % self.l                 # ristretto255.rs:244-247

    def scalar_sub(self, a, b):
        return (a - b) % self.l                 # ristretto255.rs:249-252

    def scalar_inverse(self, x):
        if x % self.l == 0:
            return None                         # ristretto255.rs:179-187
        return pow(x, -1, self.l)

    def scalar_from_u64(self, v: int) -> int:
        return (v & 0xFFFFFFFFFFFFFFFF) % self.l

    def scalar_from_bigint(self, v: int) -> int:
        """ristretto255.rs:78-105 bigint_to_scalar: BE magnitude, *first* 32 bytes
        if longer, reversed to LE, reduced mod l."""
        b = _be_min(v)
        ln = min(len(b), 32)
        le = bytes(b[ln - 1 - i] for i in range(ln)).ljust(32, b"\0")
        return int.from_bytes(le, "little") % self.l

    def scalar_to_int(self, s):
        return s                                # ristretto255.rs:107-125

    def group_order_int(self):
        return self.l

    def hash_to_scalar(self, data: bytes) -> int:
        return int.from_bytes(hashlib.sha512(data).digest(), "little") % self.l  # :196-205

    def element_to_bytes(self, e) -> bytes:
        """RFC 9496 4.3.2 Encode (ristretto255.rs:207-210)."""
        p = self.p
        X0, Y0, Z0, T0 = e
        u1 = (Z0 + Y0) * (Z0 - Y0) % p
        u2 = X0 * Y0 % p
        _, invsqrt = _sqrt_ratio_m1(1, u1 * u2 % p * u2 % p)
        den1 = invsqrt * u1 % p
        den2 = invsqrt * u2 % p
        z_inv = den1 * den2 % p * T0 % p
        ix0 = X0 * SQRT_M1 % p
        iy0 = Y0 * SQRT_M1 % p
        enchanted = den1 * INVSQRT_A_MINUS_D % p
        rotate = _is_neg(T0 * z_inv % p)
        if rotate:
            x, y, den_inv = iy0, ix0, enchanted
        else:
            x, y, den_inv = X0, Y0, den2
        if _is_neg(x * z_inv % p):
            y = (-y) % p
        s = _ct_abs(den_inv * ((Z0 - y) % p) % p)
        return s.to_bytes(32, "little")

    def bytes_to_element(self, b: bytes):
        """RFC 9496 4.3.1 Decode (ristretto255.rs:212-220)."""
        if len(b) != 32:
            return None
        p = self.p
        s = int.from_bytes(b, "little")
        if s >= p or _is_neg(s):
            return None
        ss = s * s % p
        u1 = (1 - ss) % p
        u2 = (1 + ss) % p
        u2_sqr = u2 * u2 % p
        v = (-(ED_D * u1 % p * u1) - u2_sqr) % p
        was_square, invsqrt = _sqrt_ratio_m1(1, v * u2_sqr % p)
        den_x = invsqrt * u2 % p
        den_y = invsqrt * den_x % p * v % p
        x = _ct_abs(2 * s * den_x % p)
        y = u1 * den_y % p
        t = x * y % p
        if (not was_square) or _is_neg(t) or y == 0:
            return None
        return (x, y, 1, t)

    def scalar_to_bytes(self, s: int) -> bytes:
        return s.to_bytes(32, "little")         # ristretto255.rs:222-225

    def generate_public_key(self, priv):
        return self.exp(ED_BASE, priv)          # ristretto255.rs:239-242

    def secret_mask(self, g_s) -> int:
        """participant.rs:1696-1703: SHA256(compress(G^s)) as BE integer mod l."""
        return int.from_bytes(sha256(self.element_to_bytes(g_s)), "big") % self.l

    element_to_fixed = element_to_bytes

    def element_from_fixed(self, b: bytes):
        e = self.bytes_to_element(b)
        if e is None:
            raise ValueError("invalid ristretto255 encoding")
        return e

    def scalar_to_fixed(self, s):
        return s.to_bytes(32, "little")

    def scalar_from_fixed(self, b):
        return int.from_bytes(b, "little")


GROUPS = {"modp2048": ModpGroup, "secp256k1": Secp256k1Group, "ristretto255": Ristretto255Group}

# --------------------------------------------------------------------------
# util.rs / polynomial.rs
# --------------------------------------------------------------------------


def _tdiv(a: int, b: int) -> Tuple[int, int]:
    """Rust BigInt `/` and `%` truncate toward zero."""
    q = abs(a) // abs(b)
    if (a < 0) != (b < 0):
        q = -q
    return q, a - q * b


def extend_gcd(a: int, b: int) -> Tuple[int, int, int]:
    """util.rs:18-25 (recursive, truncating division)."""
    if a == 0:
        return b, 0, 1
    qt, rem = _tdiv(b, a)
    g, x, y = extend_gcd(rem, a)
    return g, y - qt * x, x


def mod_inverse(a: int, m: int) -> Optional[int]:
    """util.rs:33-41."""
    import sys
    lim = sys.getrecursionlimit()
    sys.setrecursionlimit(max(lim, 20000))
    try:
        g, x, _ = extend_gcd(a, m)
    finally:
        sys.setrecursionlimit(lim)
    if g != 1:
        return None
    r = _tdiv(x, m)[1]
    return _tdiv(r + m, m)[1]


def lagrange_coefficient(i: int, values: Sequence[int]) -> Tuple[int, int]:
    """util.rs:47-64."""
    if i not in values:
        return 0, 1
    num, den = 1, 1
    for j in range(1, max(values) + 1):
        if j != i and j in values:
            num *= j
            den *= j - i
    return num, den


def poly_get_value(coeffs: Sequence[int], x: int) -> int:
    """polynomial.rs:50-58: unreduced integer evaluation."""
    result = coeffs[0]
    for i in range(1, len(coeffs)):
        result += coeffs[i] * x ** i
    return result


# --------------------------------------------------------------------------
# dleq.rs
# --------------------------------------------------------------------------


def dleq_response(group, w, alpha, c):
    """dleq.rs:42-50  r = w - alpha*c."""
    return group.scalar_sub(w, group.scalar_mul(alpha, c))


def dleq_verifier_commitments(group, g1, h1, g2, h2, r, c):
    """dleq.rs:66-84  a1 = g1^r * h1^c, a2 = g2^r * h2^c."""
    a1 = group.mul(group.exp(g1, r), group.exp(h1, c))
    a2 = group.mul(group.exp(g2, r), group.exp(h2, c))
    return a1, a2


def append_transcript(group, h1, h2, a1, a2) -> bytes:
    """dleq.rs:87-99  the bytes fed to the running SHA-256."""
    return b"".join(framed(group.element_to_bytes(e)) for e in (h1, h2, a1, a2))


def dleq_verify(group, g1, h1, g2, h2, c, r) -> bool:
    """dleq.rs:275-302 + 119-126."""
    if c is None or r is None:
        return False
    a1, a2 = dleq_verifier_commitments(group, g1, h1, g2, h2, r, c)
    digest = sha256(append_transcript(group, h1, h2, a1, a2))
    return group.hash_to_scalar(digest) == c


# --------------------------------------------------------------------------
# participant.rs
# --------------------------------------------------------------------------


def commitment_eval(group, commitments, position: int):
    """participant.rs:423-434 (207-215): X_i = prod_j C_j^(i^j), reference order."""
    x_val = group.identity()
    exponent = group.scalar_from_u64(1)
    pos = group.scalar_from_u64(position)
    order = group.group_order_int()
    for c_j in commitments:
        x_val = group.mul(x_val, group.exp(c_j, exponent))
        exponent = group.scalar_mul(exponent, pos) % order
    return x_val


def distribute_secret(group, secret: int, publickeys: Sequence, threshold: int,
                      coefficients: Sequence[int], witnesses: Sequence[int]) -> dict:
    """participant.rs:160-286 (secp :1094-1274, ristretto :1573-1717) with the
    randomness (polynomial coefficients, per-share DLEQ witnesses) as inputs."""
    n = len(publickeys)
    if threshold > n:
        raise AssertionError("threshold <= publickeys.len()")   # participant.rs:166
    assert len(coefficients) == threshold and len(witnesses) == n
    order = group.group_order_int()
    sub_gen = group.subgroup_generator()
    main_gen = group.generator()

    commitments = [group.exp(sub_gen, group.scalar_from_bigint(a)) for a in coefficients]
    positions: Dict[bytes, int] = {}
    shares: Dict[bytes, object] = {}
    sampling: Dict[bytes, int] = {}
    dleq_w: Dict[bytes, int] = {}
    X, A1, A2 = [], [], []
    transcript = hashlib.sha256()
    position = 1
    for pk, w in zip(publickeys, witnesses):
        key = group.element_to_bytes(pk)
        positions[key] = position
        share_scalar = group.scalar_from_bigint(poly_get_value(coefficients, position) % order)
        sampling[key] = share_scalar
        dleq_w[key] = w
        x_val = commitment_eval(group, commitments, position)
        y_enc = group.exp(pk, share_scalar)
        shares[key] = y_enc
        a1 = group.exp(sub_gen, w)               # dleq.rs:207-216
        a2 = group.exp(pk, w)
        transcript.update(append_transcript(group, x_val, y_enc, a1, a2))
        X.append(x_val)
        A1.append(a1)
        A2.append(a2)
        position += 1
    digest = transcript.digest()
    challenge = group.hash_to_scalar(digest)
    responses: Dict[bytes, int] = {}
    for pk in publickeys:
        key = group.element_to_bytes(pk)
        alpha_c = group.scalar_mul(sampling[key], challenge) % order
        responses[key] = group.scalar_sub(dleq_w[key], alpha_c) % order
    s = group.scalar_from_bigint(poly_get_value(coefficients, 0) % order
                                 if group.name == "modp2048" else poly_get_value(coefficients, 0))
    g_s = group.exp(main_gen, s)
    U = secret ^ group.secret_mask(g_s)
    return {
        "group": group.name,
        "commitments": commitments,
        "positions": positions,
        "shares": shares,
        "publickeys": list(publickeys),
        "challenge": challenge,
        "responses": responses,
        "U": U,
        # debug / fixture extras (not part of the reference box)
        "_X": X, "_a1": A1, "_a2": A2, "_digest": digest,
    }


def verify_distribution_shares(group, box: dict, trace: Optional[dict] = None) -> bool:
    """participant.rs:399-455 / mpvss.rs:90-144."""
    sub_gen = group.subgroup_generator()
    h = hashlib.sha256()
    X, A1, A2 = [], [], []
    for pk in box["publickeys"]:
        key = group.element_to_bytes(pk)
        position = box["positions"].get(key)
        response = box["responses"].get(key)
        y_enc = box["shares"].get(key)
        if position is None or response is None or y_enc is None:
            return False
        x_val = commitment_eval(group, box["commitments"], position)
        a1, a2 = dleq_verifier_commitments(group, sub_gen, x_val, pk, y_enc, response, box["challenge"])
        h.update(append_transcript(group, x_val, y_enc, a1, a2))
        X.append(x_val)
        A1.append(a1)
        A2.append(a2)
    digest = h.digest()
    if trace is not None:
        trace.update({"X": X, "a1": A1, "a2": A2, "digest": digest})
    return group.hash_to_scalar(digest) == box["challenge"]


def extract_secret_share(group, box: dict, private_key: int, w: int) -> Optional[dict]:
    """participant.rs:294-353."""
    main_gen = group.generator()
    public_key = group.generate_public_key(private_key)
    key = group.element_to_bytes(public_key)
    y_enc = box["shares"].get(key)
    if y_enc is None:
        return None
    if group.name == "modp2048":
        inv = mod_inverse(private_key, group.group_order_int())
    else:
        inv = group.scalar_inverse(private_key)
    if inv is None:
        return None
    share = group.exp(y_enc, inv)
    a1 = group.exp(main_gen, w)
    a2 = group.exp(share, w)
    digest = sha256(append_transcript(group, public_key, y_enc, a1, a2))
    challenge = group.hash_to_scalar(digest)
    response = dleq_response(group, w, private_key, challenge)
    return {"publickey": public_key, "share": share, "challenge": challenge, "response": response}


def verify_share(group, sharebox: dict, box: dict, publickey) -> bool:
    """participant.rs:361-386."""
    key = group.element_to_bytes(publickey)
    y_enc = box["shares"].get(key)
    if y_enc is None:
        return False
    return dleq_verify(group, group.generator(), publickey, sharebox["share"], y_enc,
                       sharebox["challenge"], sharebox["response"])


def reconstruct(group, shareboxes: Sequence[dict], box: dict) -> Optional[int]:
    """participant.rs:462-561 (MODP), :1452-1557 (secp), :1895-2002 (ristretto)."""
    if len(shareboxes) < len(box["commitments"]):
        return None
    shares: Dict[int, object] = {}
    for sb in shareboxes:
        pos = box["positions"].get(group.element_to_bytes(sb["publickey"]))
        if pos is None:
            return None
        shares[pos] = sb["share"]
    values = sorted(shares.keys()) if group.name == "modp2048" else list(shares.keys())
    secret = group.identity()
    for pos in values:
        share = shares[pos]
        if group.name == "modp2048":
            num, den = lagrange_coefficient(pos, values)
            negative = num * den < 0
            num, den = abs(num), abs(den)
            from math import gcd
            g = gcd(num, den)
            num //= g
            den //= g
            den_inv = mod_inverse(den, group.subgroup_order())
            if den_inv is None:
                return None
            exponent = (num * den_inv) % group.subgroup_order()
            factor = group.exp(share, exponent)
            if negative:
                factor = group.element_inverse(factor)
                if factor is None:
                    return None
        else:
            order = group.group_order_int()
            lam_num, lam_den, sign = 1, 1, 1
            for j in values:
                if j == pos:
                    continue
                lam_num = lam_num * group.scalar_from_u64(j) % order
                diff = j - pos
                if diff < 0:
                    sign = -sign
                    lam_den = lam_den * group.scalar_from_u64(-diff) % order
                else:
                    lam_den = lam_den * group.scalar_from_u64(diff) % order
            lam = lam_num * pow(lam_den, -1, order) % order
            factor = group.exp(share, lam)
            if sign < 0:
                factor = group.element_inverse(factor)
        secret = group.mul(secret, factor)
    return group.secret_mask(secret) ^ box["U"]


# --------------------------------------------------------------------------
# Flat (SoA) view of a box: the layout the C-ABI takes (include/mpvss_hip.h)
# --------------------------------------------------------------------------


def box_to_flat(group, box: dict) -> dict:
    """Positions-ordered fixed-width arrays, as the boundary consumes them."""
    pks = box["publickeys"]
    keys = [group.element_to_bytes(pk) for pk in pks]
    return {
        "n": len(pks),
        "t": len(box["commitments"]),
        "commitments": b"".join(group.element_to_fixed(c) for c in box["commitments"]),
        "positions": [box["positions"][k] for k in keys],
        "publickeys": b"".join(group.element_to_fixed(pk) for pk in pks),
        "shares": b"".join(group.element_to_fixed(box["shares"][k]) for k in keys),
        "responses": b"".join(group.scalar_to_fixed(box["responses"][k]) for k in keys),
        "challenge": group.scalar_to_fixed(box["challenge"]),
    }


def string_to_secret(s: str) -> int:
    return int.from_bytes(s.encode(), "big")     # lib.rs:49-52


def string_from_secret(v: int) -> str:
    return _be_min(v).decode()                   # lib.rs:54-57
