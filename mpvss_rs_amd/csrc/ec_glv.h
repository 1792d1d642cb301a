// secp256k1's efficiently computable endomorphism (GLV, Gallant-Lambert-Vanstone 2001) for the windowed double-scalar
// multiplications of the verifier and the dealer (a1 = r G + c X, a2 = r y + c Y, Y = p y ...):
//   reference: Secp256k1Group::exp = ProjectivePoint * Scalar  src/groups/secp256k1.rs:91-100 -- k256 0.13's mul takes the same
//   route (arithmetic/mul.rs: decompose_scalar + phi), so this is the reference's own algorithm, not a shortcut around it
//
//   phi(x, y) = (beta x, y) = lambda (x, y),  beta^3 = 1 (mod p), lambda^3 = 1 (mod n)
//   k = k1 + k2 lambda (mod n) with |k1|, |k2| < 2^128   =>   k P = k1 P + k2 phi(P): 128 doublings instead of 256, and
//   the table of phi(P) is the table of P with every X multiplied by beta (homogeneous coordinates: beta X / Z = beta x).
// The split is the standard lattice one (c1 = round(k g1 / 2^384), c2 = round(k g2 / 2^384), k2 = c1 (-b1) + c2 (-b2),
// k1 = k - k2 lambda; constants and their derivation: tools/gen_ec_consts.py, which also checks lambda G = (beta Gx, Gy)).
// Plain C++ (host + device): tests/ec_host_shim.cpp runs the same code on the CPU against Python integers.
#pragma once
#include "ec_curves.h"
#include "ec_scalar.h"

namespace ec {

struct GlvHalf {
  u32 kp[5];      // |k_j| + sum_{w < 32} 8 * 16^w (33 nibbles): nibble w minus 8 is the signed digit d_w for w < 32, nibble 32 (0 or 1) the top one
  bool neg;       // k_j < 0: every digit changes its sign at use
};

namespace glv_detail {
// a, b reduced: r = a + b mod n / r = a - b mod n
EC_HD void add_mod(Sc& r, const Sc& a, const Sc& b) {
  u32 t[8];
  u64 c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    c += (u64)a.v[i] + b.v[i];
    t[i] = (u32)c;
    c >>= 32;
  }
  bool ge = c != 0;
  if (!ge) {
    ge = true;
    bool decided = false;
#pragma unroll
    for (int i = 7; i >= 0; --i)
      if (!decided && t[i] != OrderSecp::n(i)) { ge = t[i] > OrderSecp::n(i); decided = true; }
  }
  const u32 mask = ge ? 0xffffffffu : 0u;
  u64 borrow = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const u64 d = (u64)t[i] - (OrderSecp::n(i) & mask) - borrow;
    r.v[i] = (u32)d;
    borrow = (d >> 63) & 1;
  }
}
EC_HD void sub_mod(Sc& r, const Sc& a, const Sc& b) {
  u32 t[8];
  u64 borrow = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const u64 d = (u64)a.v[i] - b.v[i] - borrow;
    t[i] = (u32)d;
    borrow = (d >> 63) & 1;
  }
  const u32 mask = borrow ? 0xffffffffu : 0u;
  u64 c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    c += (u64)t[i] + (OrderSecp::n(i) & mask);
    r.v[i] = (u32)c;
    c >>= 32;
  }
}
// round(k g / 2^384) for a 256-bit g: the top 128 bits of the 512-bit product, rounded at bit 383
template <class G>
EC_HD void mul_shift_384(Sc& r, const Sc& k, G g) {
  u32 prod[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) prod[i] = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    u64 c = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      c += (u64)k.v[i] * g(j) + prod[i + j];
      prod[i + j] = (u32)c;
      c >>= 32;
    }
    prod[i + 8] = (u32)c;
  }
  u64 c = prod[11] >> 31;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    c += prod[12 + i];
    r.v[i] = (u32)c;
    c >>= 32;
  }
  r.v[4] = (u32)c;        // (at most 2^128: one bit)
  r.v[5] = r.v[6] = r.v[7] = 0;
}
// sign and magnitude of a residue that is a small signed number mod n, recoded into signed 4-bit windows
EC_HD void half_from_residue(GlvHalf& h, const Sc& r) {
  bool gt = false, decided = false;
#pragma unroll
  for (int i = 7; i >= 0; --i)
    if (!decided && r.v[i] != GlvSecp::half_n(i)) { gt = r.v[i] > GlvSecp::half_n(i); decided = true; }
  h.neg = gt;
  // magnitude: r, or n - r
  u32 m[5];
  u64 borrow = 0;
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const u64 d = (u64)OrderSecp::n(i) - r.v[i] - borrow;
    m[i] = gt ? (u32)d : r.v[i];
    borrow = (d >> 63) & 1;
  }
  u64 c = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    c += (u64)m[i] + 0x88888888u;
    h.kp[i] = (u32)c;
    c >>= 32;
  }
  h.kp[4] = (u32)(c + m[4]);
}
}  // namespace glv_detail

// k (8 little-endian words, any 256-bit value: it is reduced mod n first) -> k1 (h[0]) and k2 (h[1]), k = k1 + k2 lambda (mod n)
EC_HD void secp_glv_split(GlvHalf (&h)[2], const u32 (&kw)[8]) {
  using namespace glv_detail;
  typedef ScalarField<OrderSecp> F;
  Sc k, zero, c1, c2, t, r1, r2;
#pragma unroll
  for (int i = 0; i < 8; ++i) { k.v[i] = kw[i]; zero.v[i] = 0; }
  add_mod(k, k, zero);                                      // one conditional subtraction: k < 2^256 < 2n
  mul_shift_384(c1, k, [](int i) { return GlvSecp::g1(i); });
  mul_shift_384(c2, k, [](int i) { return GlvSecp::g2(i); });
  Sc mb1, mb2, lam;
#pragma unroll
  for (int i = 0; i < 8; ++i) { mb1.v[i] = GlvSecp::mb1_r(i); mb2.v[i] = GlvSecp::mb2_r(i); lam.v[i] = GlvSecp::lambda_r(i); }
  F::mont_mul(c1, c1, mb1);                                 // c1 (-b1) mod n
  F::mont_mul(c2, c2, mb2);                                 // c2 (-b2) mod n
  add_mod(r2, c1, c2);
  F::mont_mul(t, r2, lam);                                  // k2 lambda mod n
  sub_mod(r1, k, t);
  half_from_residue(h[0], r1);
  half_from_residue(h[1], r2);
}

// signed digit w (0..32) of a recoded half: the sign of the half applied
EC_HD int glv_digit(const GlvHalf& h, int w) {
  const int nib = (int)((h.kp[w >> 3] >> (4 * (w & 7))) & 15u);
  const int d = w == 32 ? nib : nib - 8;
  return h.neg ? -d : d;
}

// the cached table entry of phi(P) from the one of P: X <- beta X
EC_HD void secp_phi_cached(Secp::Cached& e) {
  const Fe beta = {EC_SECP_BETA_INIT};
  Secp::Fp::mul(e.X, e.X, beta);
}

}  // namespace ec
