// Group law, encodings and the per-share routines of the two elliptic-curve groups of the reference:
//   secp256k1     src/groups/secp256k1.rs:38-189   (k256 0.13: complete projective formulas, SEC1 33-byte)
//   ristretto255  src/groups/ristretto255.rs:45-253 (curve25519-dalek 4 / RFC 9496: extended Edwards, 32-byte)
// One group element per lane.  Plain C++ (host + device) so the CPU unit tests compile the same code.
//
// Both curve types expose the same interface:
//   Point, identity(), add(), dbl(), decode(bytes)->ok, encode(bytes), scalar byte order,
// and the generic routines at the bottom (scalar_mul, dual_mul, horner step) are templates over it.
#pragma once
#include <stddef.h>

#include "ec_consts.h"
#include "ec_field.h"

namespace ec {

// canonical field element (limbs < 2^26, value < p) <-> 8 little-endian 32-bit words
EC_HD void pack_fe(u32* w, const Fe& a) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int bit = 32 * i;
    const int j = bit / 26, sft = bit % 26;
    u64 v = (u64)a.v[j] >> sft;
    if (j + 1 < 10) v |= (u64)a.v[j + 1] << (26 - sft);
    if (j + 2 < 10 && 52 - sft < 32) v |= (u64)a.v[j + 2] << (52 - sft);
    w[i] = (u32)v;
  }
}
EC_HD void unpack_fe(Fe& a, const u32* w) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const int bit = 26 * i;
    const int j = bit / 32, sft = bit % 32;
    u64 v = (u64)w[j] >> sft;
    if (j + 1 < 8) v |= (u64)w[j + 1] << (32 - sft);
    a.v[i] = (u32)v & M26;
  }
}

// ------------------------------------------------------------------------------------------------
// secp256k1: y^2 = x^3 + 7, homogeneous projective (X:Y:Z), identity (0:1:0).
// Complete formulas of Renes-Costello-Batina 2016 for a = 0 (Algorithms 7 and 9), b3 = 21 --
// the formulas k256 uses; they need no special cases for doubling, inverses or the identity.
// ------------------------------------------------------------------------------------------------
struct Secp {
  typedef F<PrimeSecp> Fp;
  struct Point {
    Fe X, Y, Z;
  };
  static constexpr int ENC_LEN = 33;
  static constexpr int SCALAR_BIG_ENDIAN = 1;   // secp256k1.rs:154-156
  static constexpr int POINT_WORDS = 30;

  static EC_HD void identity(Point& r) {
    Fp::zero(r.X);
    Fp::one(r.Y);
    Fp::zero(r.Z);
  }
  static EC_HD void generator(Point& r) {
    const Fe gx = {EC_SECP_GX_INIT}, gy = {EC_SECP_GY_INIT};
    r.X = gx;
    r.Y = gy;
    Fp::one(r.Z);
  }
  static EC_HD void add(Point& r, const Point& p, const Point& q) {
    Fe t0, t1, t2, t3, t4, x3, y3, z3;
    Fp::mul(t0, p.X, q.X);
    Fp::mul(t1, p.Y, q.Y);
    Fp::mul(t2, p.Z, q.Z);
    Fp::add(t3, p.X, p.Y);
    Fp::add(t4, q.X, q.Y);
    Fp::mul(t3, t3, t4);
    Fp::add(t4, t0, t1);
    Fp::sub(t3, t3, t4);
    Fp::add(t4, p.Y, p.Z);
    Fp::add(x3, q.Y, q.Z);
    Fp::mul(t4, t4, x3);
    Fp::add(x3, t1, t2);
    Fp::sub(t4, t4, x3);
    Fp::add(x3, p.X, p.Z);
    Fp::add(y3, q.X, q.Z);
    Fp::mul(x3, x3, y3);
    Fp::add(y3, t0, t2);
    Fp::sub(y3, x3, y3);
    Fp::add(x3, t0, t0);
    Fp::addc(t0, x3, t0);            // 3 X1X2
    Fp::mul_small(t2, t2, 21);       // b3 Z1Z2
    Fp::addc(z3, t1, t2);
    Fp::sub(t1, t1, t2);
    Fp::mul_small(y3, y3, 21);
    Fp::mul(x3, t4, y3);
    Fp::mul(t2, t3, t1);
    Fp::sub(x3, t2, x3);
    Fp::mul(y3, y3, t0);
    Fp::mul(t1, t1, z3);
    Fp::addc(y3, t1, y3);
    Fp::mul(t0, t0, t3);
    Fp::mul(z3, z3, t4);
    Fp::addc(z3, z3, t0);
    r.X = x3;
    r.Y = y3;
    r.Z = z3;
  }
  static EC_HD void dbl(Point& r, const Point& p) {
    Fe t0, t1, t2, x3, y3, z3;
    Fp::sqr(t0, p.Y);
    Fp::add(z3, t0, t0);
    Fp::add(z3, z3, z3);
    Fp::addc(z3, z3, z3);            // 8 Y^2
    Fp::mul(t1, p.Y, p.Z);
    Fp::sqr(t2, p.Z);
    Fp::mul_small(t2, t2, 21);
    Fp::mul(x3, t2, z3);
    Fp::addc(y3, t0, t2);
    Fp::mul(z3, t1, z3);
    Fp::add(t1, t2, t2);
    Fp::addc(t2, t1, t2);
    Fp::sub(t0, t0, t2);
    Fp::mul(y3, t0, y3);
    Fp::addc(y3, x3, y3);
    Fp::mul(t1, p.X, p.Y);
    Fp::mul(x3, t0, t1);
    Fp::addc(x3, x3, x3);
    r.X = x3;
    r.Y = y3;
    r.Z = z3;
  }
  static EC_HD void neg(Point& r, const Point& p) {
    r.X = p.X;
    Fp::neg(r.Y, p.Y);
    r.Z = p.Z;
  }
  static EC_HD void cmov(Point& r, const Point& a, bool c) {
    Fp::cmov(r.X, a.X, c);
    Fp::cmov(r.Y, a.Y, c);
    Fp::cmov(r.Z, a.Z, c);
  }
  // a^((p+1)/4) (p = 3 mod 4) and a^(p-2), with the usual addition chain on runs of ones
  static EC_HD void pow_chain(Fe& x223, Fe& x22, Fe& x2, const Fe& a) {
    Fe x3, x6, x9, x11, x44, x88, x176, x220;
    Fp::sqr(x2, a); Fp::mul(x2, x2, a);
    Fp::sqr(x3, x2); Fp::mul(x3, x3, a);
    Fp::sqrn(x6, x3, 3); Fp::mul(x6, x6, x3);
    Fp::sqrn(x9, x6, 3); Fp::mul(x9, x9, x3);
    Fp::sqrn(x11, x9, 2); Fp::mul(x11, x11, x2);
    Fp::sqrn(x22, x11, 11); Fp::mul(x22, x22, x11);
    Fp::sqrn(x44, x22, 22); Fp::mul(x44, x44, x22);
    Fp::sqrn(x88, x44, 44); Fp::mul(x88, x88, x44);
    Fp::sqrn(x176, x88, 88); Fp::mul(x176, x176, x88);
    Fp::sqrn(x220, x176, 44); Fp::mul(x220, x220, x44);
    Fp::sqrn(x223, x220, 3); Fp::mul(x223, x223, x3);
  }
  static EC_HD void sqrt_candidate(Fe& r, const Fe& a) {   // a^((p+1)/4)
    Fe x223, x22, x2, t;
    pow_chain(x223, x22, x2, a);
    Fp::sqrn(t, x223, 23); Fp::mul(t, t, x22);
    Fp::sqrn(t, t, 6); Fp::mul(t, t, x2);
    Fp::sqrn(r, t, 2);
  }
  static EC_HD void invert(Fe& r, const Fe& a) {           // a^(p-2)
    Fe x223, x22, x2, t;
    pow_chain(x223, x22, x2, a);
    Fp::sqrn(t, x223, 23); Fp::mul(t, t, x22);
    Fp::sqrn(t, t, 5); Fp::mul(t, t, a);
    Fp::sqrn(t, t, 3); Fp::mul(t, t, x2);
    Fp::sqrn(t, t, 2); Fp::mul(r, t, a);
  }
  // SEC1 compressed decode (secp256k1.rs:138-152).  33 zero bytes decode to the identity, which is
  // what k256's GroupEncoding produces for it.  Returns false for anything else that is invalid.
  static EC_HD bool decode(Point& r, const uint8_t* b) {
    bool allzero = true;
    for (int i = 0; i < 33; ++i) allzero = allzero && (b[i] == 0);
    if (allzero) {
      identity(r);
      return true;
    }
    uint8_t le[32];
    for (int i = 0; i < 32; ++i) le[i] = b[32 - i];
    Fe x;
    Fp::from_le32_raw(x, le);
    bool ok = (b[0] == 2 || b[0] == 3) && Fp::is_canonical(x);
    Fe y2, y, t;
    Fp::sqr(t, x);
    Fp::mul(y2, t, x);
    Fe seven;
    Fp::zero(seven);
    seven.v[0] = 7;
    Fp::addc(y2, y2, seven);
    sqrt_candidate(y, y2);
    Fp::sqr(t, y);
    ok = ok && Fp::equal(t, y2);
    Fe ny;
    Fp::neg(ny, y);
    Fp::cmov(y, ny, Fp::is_odd(y) != ((b[0] & 1) != 0));
    r.X = x;
    r.Y = y;
    Fp::one(r.Z);
    if (!ok) identity(r);
    return ok;
  }
  static EC_HD void encode(uint8_t* out, const Point& p) {  // secp256k1.rs:133-136
    if (Fp::is_zero(p.Z)) {
      for (int i = 0; i < 33; ++i) out[i] = 0;
      return;
    }
    Fe zi, x, y;
    invert(zi, p.Z);
    Fp::mul(x, p.X, zi);
    Fp::mul(y, p.Y, zi);
    Fp::canon(x);
    Fp::canon(y);
    uint8_t le[32];
    Fp::to_le32(le, x);
    out[0] = 2 + (y.v[0] & 1);
    for (int i = 0; i < 32; ++i) out[1 + i] = le[31 - i];
  }

  // ---- window tables (ec_kernels.hip: signed 4-bit Straus, fixed-base comb for G) ---------------------------------
  // per-share table entry: the projective point itself; the negative digit negates Y
  typedef Point Cached;
  static constexpr int CACHED_WORDS = 30;
  static EC_HD void to_cached(Cached& r, const Point& p) { r = p; }
  static EC_HD void add_cached(Point& r, const Point& p, const Cached& q, bool negate) {
    Point t = q;
    Fe ny;
    Fp::neg(ny, q.Y);
    Fp::cmov(t.Y, ny, negate);
    add(r, p, t);
  }
  // comb entry: an affine point (never the identity), coordinates canonical and packed 8 x 32 bit
  struct Affine {
    Fe x, y;
  };
  static constexpr int AFFINE_PACKED_WORDS = 16;
  // mixed addition, Renes-Costello-Batina Algorithm 8 (a = 0, b3 = 21): complete for every P, Q = (x2, y2) affine
  static EC_HD void add_affine(Point& r, const Point& p, const Affine& q, bool negate) {
    Fe y2 = q.y, ny;
    Fp::neg(ny, q.y);
    Fp::cmov(y2, ny, negate);
    Fe t0, t1, t2, t3, t4, x3, y3, z3;
    Fp::mul(t0, p.X, q.x);
    Fp::mul(t1, p.Y, y2);
    Fp::add(t3, q.x, y2);
    Fp::add(t4, p.X, p.Y);
    Fp::mul(t3, t3, t4);
    Fp::add(t4, t0, t1);
    Fp::sub(t3, t3, t4);               // X1 y2 + x2 Y1
    Fp::mul(t4, y2, p.Z);
    Fp::addc(t4, t4, p.Y);             // Y1 + y2 Z1
    Fp::mul(y3, q.x, p.Z);
    Fp::addc(y3, y3, p.X);             // X1 + x2 Z1
    Fp::add(x3, t0, t0);
    Fp::addc(t0, x3, t0);              // 3 X1 x2
    Fp::mul_small(t2, p.Z, 21);        // b3 Z1
    Fp::addc(z3, t1, t2);
    Fp::sub(t1, t1, t2);
    Fp::mul_small(y3, y3, 21);
    Fp::mul(x3, t4, y3);
    Fp::mul(t2, t3, t1);
    Fp::sub(x3, t2, x3);
    Fp::mul(y3, y3, t0);
    Fp::mul(t1, t1, z3);
    Fp::addc(y3, t1, y3);
    Fp::mul(t0, t0, t3);
    Fp::mul(z3, z3, t4);
    Fp::addc(z3, z3, t0);
    r.X = x3;
    r.Y = y3;
    r.Z = z3;
  }
  // affine form of a point that is not the identity (one inversion): comb construction only
  static EC_HD void to_affine(Affine& r, const Point& p) {
    Fe zi;
    invert(zi, p.Z);
    Fp::mul(r.x, p.X, zi);
    Fp::mul(r.y, p.Y, zi);
    Fp::canon(r.x);
    Fp::canon(r.y);
  }
  static EC_HD void pack_affine(u32* w, const Affine& a) {
    pack_fe(w, a.x);
    pack_fe(w + 8, a.y);
  }
  static EC_HD void unpack_affine(Affine& a, const u32* w) {
    unpack_fe(a.x, w);
    unpack_fe(a.y, w + 8);
  }
  // B points to SEC1 bytes with ONE shared inversion (Montgomery's trick): 5 products per point + 1/B inversion.
  // `load(i, P)` fetches point i (called twice per point, so that only the prefix products stay live),
  // `live(i)` says whether output i is written.
  template <int B, class Load, class Live>
  static EC_HD void encode_batch(uint8_t* out, size_t out_stride, Load load, Live live) {
    Fe pre[B], acc, zi, one;
    Fp::one(one);
    acc = one;
#pragma unroll
    for (int i = 0; i < B; ++i) {
      Point p;
      load(i, p);
      Fe z = p.Z;
      Fp::cmov(z, one, Fp::is_zero(p.Z));   // the identity does not enter the product
      pre[i] = acc;
      Fp::mul(acc, acc, z);
    }
    invert(acc, acc);
#pragma unroll
    for (int i = B - 1; i >= 0; --i) {
      Point p;
      load(i, p);
      const bool inf = Fp::is_zero(p.Z);
      Fe z = p.Z;
      Fp::cmov(z, one, inf);
      Fp::mul(zi, acc, pre[i]);        // 1 / Z_i
      Fp::mul(acc, acc, z);
      Fe x, y;
      Fp::mul(x, p.X, zi);
      Fp::mul(y, p.Y, zi);
      Fp::canon(x);
      Fp::canon(y);
      if (live(i)) {
        uint8_t* o = out + (size_t)i * out_stride;
        if (inf) {
          for (int k = 0; k < 33; ++k) o[k] = 0;
        } else {
          uint8_t le[32];
          Fp::to_le32(le, x);
          o[0] = 2 + (y.v[0] & 1);
          for (int k = 0; k < 32; ++k) o[1 + k] = le[31 - k];
        }
      }
    }
  }
};

// ------------------------------------------------------------------------------------------------
// ristretto255 over the twisted Edwards curve -x^2 + y^2 = 1 + d x^2 y^2, extended coordinates
// (X:Y:Z:T), identity (0:1:1:0).  Unified addition (add-2008-hwcd-3) is complete; encoding and
// decoding follow RFC 9496 section 4.3 step by step.
// ------------------------------------------------------------------------------------------------
struct Ristretto {
  typedef F<PrimeEd> Fp;
  struct Point {
    Fe X, Y, Z, T;
  };
  static constexpr int ENC_LEN = 32;
  static constexpr int SCALAR_BIG_ENDIAN = 0;   // ristretto255.rs:222-225
  static constexpr int POINT_WORDS = 40;

  static EC_HD void identity(Point& r) {
    Fp::zero(r.X);
    Fp::one(r.Y);
    Fp::one(r.Z);
    Fp::zero(r.T);
  }
  static EC_HD void generator(Point& r) {
    const Fe bx = {EC_ED_BX_INIT}, by = {EC_ED_BY_INIT}, bt = {EC_ED_BT_INIT};
    r.X = bx;
    r.Y = by;
    Fp::one(r.Z);
    r.T = bt;
  }
  static EC_HD void add(Point& r, const Point& p, const Point& q) {
    const Fe d2 = {EC_ED_2D_INIT};
    Fe a, b, c, d, e, f, g, h, t;
    Fp::sub(a, p.Y, p.X);
    Fp::sub(t, q.Y, q.X);
    Fp::mul(a, a, t);
    Fp::addc(b, p.Y, p.X);
    Fp::addc(t, q.Y, q.X);
    Fp::mul(b, b, t);
    Fp::mul(c, p.T, q.T);
    Fp::mul(c, c, d2);
    Fp::mul(d, p.Z, q.Z);
    Fp::addc(d, d, d);
    Fp::sub(e, b, a);
    Fp::sub(f, d, c);
    Fp::addc(g, d, c);
    Fp::addc(h, b, a);
    Fp::mul(r.X, e, f);
    Fp::mul(r.Y, g, h);
    Fp::mul(r.T, e, h);
    Fp::mul(r.Z, f, g);
  }
  static EC_HD void dbl(Point& r, const Point& p) {          // dbl-2008-hwcd, a = -1
    Fe a, b, c, e, f, g, h, t;
    Fp::sqr(a, p.X);
    Fp::sqr(b, p.Y);
    Fp::sqr(c, p.Z);
    Fp::addc(c, c, c);
    Fp::addc(t, p.X, p.Y);
    Fp::sqr(e, t);
    Fp::sub(e, e, a);
    Fp::sub(e, e, b);            // E = (X+Y)^2 - A - B
    Fp::sub(g, b, a);            // G = D + B with D = -A
    Fp::sub(f, g, c);            // F = G - C
    Fp::addc(h, a, b);
    Fp::neg(h, h);               // H = D - B = -(A + B)
    Fp::mul(r.X, e, f);
    Fp::mul(r.Y, g, h);
    Fp::mul(r.T, e, h);
    Fp::mul(r.Z, f, g);
  }
  static EC_HD void neg(Point& r, const Point& p) {
    Fp::neg(r.X, p.X);
    r.Y = p.Y;
    r.Z = p.Z;
    Fp::neg(r.T, p.T);
  }
  static EC_HD void cmov(Point& r, const Point& a, bool c) {
    Fp::cmov(r.X, a.X, c);
    Fp::cmov(r.Y, a.Y, c);
    Fp::cmov(r.Z, a.Z, c);
    Fp::cmov(r.T, a.T, c);
  }
  // z^(2^252 - 3) = z^((p-5)/8)
  static EC_HD void pow22523(Fe& r, const Fe& z) {
    Fe t0, t1, t2;
    Fp::sqr(t0, z);
    Fp::sqrn(t1, t0, 2);
    Fp::mul(t1, z, t1);
    Fp::mul(t0, t0, t1);
    Fp::sqr(t0, t0);
    Fp::mul(t0, t1, t0);
    Fp::sqrn(t1, t0, 5);
    Fp::mul(t0, t1, t0);
    Fp::sqrn(t1, t0, 10);
    Fp::mul(t1, t1, t0);
    Fp::sqrn(t2, t1, 20);
    Fp::mul(t1, t2, t1);
    Fp::sqrn(t1, t1, 10);
    Fp::mul(t0, t1, t0);
    Fp::sqrn(t1, t0, 50);
    Fp::mul(t1, t1, t0);
    Fp::sqrn(t2, t1, 100);
    Fp::mul(t1, t2, t1);
    Fp::sqrn(t1, t1, 50);
    Fp::mul(t0, t1, t0);
    Fp::sqrn(t0, t0, 2);
    Fp::mul(r, t0, z);
  }
  static EC_HD void ct_abs(Fe& r) {
    Fe n;
    Fp::neg(n, r);
    Fp::cmov(r, n, Fp::is_odd(r));
  }
  // RFC 9496 4.2: (was_square, r) with r = sqrt(u/v) or sqrt(i*u/v), r non-negative
  static EC_HD bool sqrt_ratio_m1(Fe& r, const Fe& u, const Fe& v) {
    const Fe sqrt_m1 = {EC_ED_SQRT_M1_INIT};
    Fe v3, v7, t, check, nu, nui;
    Fp::sqr(t, v);
    Fp::mul(v3, t, v);
    Fp::sqr(t, v3);
    Fp::mul(v7, t, v);
    Fp::mul(t, u, v7);
    pow22523(t, t);
    Fp::mul(r, u, v3);
    Fp::mul(r, r, t);
    Fp::sqr(t, r);
    Fp::mul(check, v, t);
    Fp::neg(nu, u);
    Fp::mul(nui, nu, sqrt_m1);
    const bool correct = Fp::equal(check, u);
    const bool flipped = Fp::equal(check, nu);
    const bool flipped_i = Fp::equal(check, nui);
    Fe ri;
    Fp::mul(ri, r, sqrt_m1);
    Fp::cmov(r, ri, flipped || flipped_i);
    ct_abs(r);
    return correct || flipped;
  }
  // RFC 9496 4.3.1 (ristretto255.rs:212-220)
  static EC_HD bool decode(Point& r, const uint8_t* b) {
    const Fe dconst = {EC_ED_D_INIT};
    Fe s, ss, u1, u2, u2s, v, t, inv, dx, dy, x, y, one;
    Fp::from_le32_raw(s, b);
    bool ok = ((b[31] & 0x80) == 0) && Fp::is_canonical(s) && ((s.v[0] & 1) == 0);
    Fp::one(one);
    Fp::sqr(ss, s);
    Fp::sub(u1, one, ss);
    Fp::addc(u2, one, ss);
    Fp::sqr(u2s, u2);
    Fp::sqr(t, u1);
    Fp::mul(t, t, dconst);
    Fp::neg(t, t);
    Fp::sub(v, t, u2s);
    Fp::mul(t, v, u2s);
    const bool was_square = sqrt_ratio_m1(inv, one, t);
    Fp::mul(dx, inv, u2);
    Fp::mul(dy, inv, dx);
    Fp::mul(dy, dy, v);
    Fp::mul(x, s, dx);
    Fp::addc(x, x, x);
    ct_abs(x);
    Fp::mul(y, u1, dy);
    Fp::mul(t, x, y);
    ok = ok && was_square && !Fp::is_odd(t) && !Fp::is_zero(y);
    r.X = x;
    r.Y = y;
    Fp::one(r.Z);
    r.T = t;
    if (!ok) identity(r);
    return ok;
  }
  // RFC 9496 4.3.2 (ristretto255.rs:207-210)
  static EC_HD void encode(uint8_t* out, const Point& p) {
    const Fe sqrt_m1 = {EC_ED_SQRT_M1_INIT}, invsqrt_amd = {EC_ED_INVSQRT_A_MINUS_D_INIT};
    Fe u1, u2, t, inv, den1, den2, zinv, ix, iy, ench, x, y, deninv, s, one;
    Fp::addc(u1, p.Z, p.Y);
    Fp::sub(t, p.Z, p.Y);
    Fp::mul(u1, u1, t);
    Fp::mul(u2, p.X, p.Y);
    Fp::sqr(t, u2);
    Fp::mul(t, t, u1);
    Fp::one(one);
    sqrt_ratio_m1(inv, one, t);
    Fp::mul(den1, inv, u1);
    Fp::mul(den2, inv, u2);
    Fp::mul(zinv, den1, den2);
    Fp::mul(zinv, zinv, p.T);
    Fp::mul(ix, p.X, sqrt_m1);
    Fp::mul(iy, p.Y, sqrt_m1);
    Fp::mul(ench, den1, invsqrt_amd);
    Fp::mul(t, p.T, zinv);
    const bool rotate = Fp::is_odd(t);
    x = p.X;
    y = p.Y;
    deninv = den2;
    Fp::cmov(x, iy, rotate);
    Fp::cmov(y, ix, rotate);
    Fp::cmov(deninv, ench, rotate);
    Fp::mul(t, x, zinv);
    Fe ny;
    Fp::neg(ny, y);
    Fp::cmov(y, ny, Fp::is_odd(t));
    Fp::sub(t, p.Z, y);
    Fp::mul(s, deninv, t);
    ct_abs(s);
    Fp::canon(s);
    Fp::to_le32(out, s);
  }

  // ---- window tables (ec_kernels.hip) -----------------------------------------------------------------------------
  // per-share table entry in "cached" form (Y+X, Y-X, Z, 2dT): addition costs 8 products; the negative digit swaps
  // the first two and negates the last
  struct Cached {
    Fe YpX, YmX, Z, T2d;
  };
  static constexpr int CACHED_WORDS = 40;
  static EC_HD void to_cached(Cached& r, const Point& p) {
    const Fe d2 = {EC_ED_2D_INIT};
    Fp::addc(r.YpX, p.Y, p.X);
    Fp::sub(r.YmX, p.Y, p.X);
    r.Z = p.Z;
    Fp::mul(r.T2d, p.T, d2);
  }
  static EC_HD void add_cached(Point& r, const Point& p, const Cached& q, bool negate) {
    Fe qa = q.YmX, qb = q.YpX, qc = q.T2d, nc;
    Fp::cmov(qa, q.YpX, negate);
    Fp::cmov(qb, q.YmX, negate);
    Fp::neg(nc, q.T2d);
    Fp::cmov(qc, nc, negate);
    Fe a, b, c, d, e, f, g, h;
    Fp::sub(a, p.Y, p.X);
    Fp::mul(a, a, qa);
    Fp::addc(b, p.Y, p.X);
    Fp::mul(b, b, qb);
    Fp::mul(c, p.T, qc);
    Fp::mul(d, p.Z, q.Z);
    Fp::addc(d, d, d);
    Fp::sub(e, b, a);
    Fp::sub(f, d, c);
    Fp::addc(g, d, c);
    Fp::addc(h, b, a);
    Fp::mul(r.X, e, f);
    Fp::mul(r.Y, g, h);
    Fp::mul(r.T, e, h);
    Fp::mul(r.Z, f, g);
  }
  // comb entry: affine Niels form (y+x, y-x, 2dxy) of a point with Z = 1, canonical, packed 8 x 32 bit each
  struct Affine {
    Fe ypx, ymx, xy2d;
  };
  static constexpr int AFFINE_PACKED_WORDS = 24;
  static EC_HD void add_affine(Point& r, const Point& p, const Affine& q, bool negate) {   // 7 products
    Fe qa = q.ymx, qb = q.ypx, qc = q.xy2d, nc;
    Fp::cmov(qa, q.ypx, negate);
    Fp::cmov(qb, q.ymx, negate);
    Fp::neg(nc, q.xy2d);
    Fp::cmov(qc, nc, negate);
    Fe a, b, c, d, e, f, g, h;
    Fp::sub(a, p.Y, p.X);
    Fp::mul(a, a, qa);
    Fp::addc(b, p.Y, p.X);
    Fp::mul(b, b, qb);
    Fp::mul(c, p.T, qc);
    Fp::addc(d, p.Z, p.Z);
    Fp::sub(e, b, a);
    Fp::sub(f, d, c);
    Fp::addc(g, d, c);
    Fp::addc(h, b, a);
    Fp::mul(r.X, e, f);
    Fp::mul(r.Y, g, h);
    Fp::mul(r.T, e, h);
    Fp::mul(r.Z, f, g);
  }
  // 1 / z = z^(p-2) = (z^(2^252 - 3))^8 * z^3
  static EC_HD void invert(Fe& r, const Fe& z) {
    Fe t, z2, z3;
    pow22523(t, z);
    Fp::sqrn(t, t, 3);
    Fp::sqr(z2, z);
    Fp::mul(z3, z2, z);
    Fp::mul(r, t, z3);
  }
  static EC_HD void to_affine(Affine& r, const Point& p) {       // comb construction only
    const Fe d2 = {EC_ED_2D_INIT};
    Fe zi, x, y, t;
    invert(zi, p.Z);
    Fp::mul(x, p.X, zi);
    Fp::mul(y, p.Y, zi);
    Fp::addc(r.ypx, y, x);
    Fp::sub(r.ymx, y, x);
    Fp::mul(t, x, y);
    Fp::mul(r.xy2d, t, d2);
    Fp::canon(r.ypx);
    Fp::canon(r.ymx);
    Fp::canon(r.xy2d);
  }
  static EC_HD void pack_affine(u32* w, const Affine& a) {
    pack_fe(w, a.ypx);
    pack_fe(w + 8, a.ymx);
    pack_fe(w + 16, a.xy2d);
  }
  static EC_HD void unpack_affine(Affine& a, const u32* w) {
    unpack_fe(a.ypx, w);
    unpack_fe(a.ymx, w + 8);
    unpack_fe(a.xy2d, w + 16);
  }
};

// ------------------------------------------------------------------------------------------------
// generic routines
// ------------------------------------------------------------------------------------------------

// bit `i` (0 = least significant) of a 32-byte scalar in the curve's byte order
template <class C>
EC_HD u32 scalar_bit(const uint8_t* k, int i) {
  const int byte = C::SCALAR_BIG_ENDIAN ? 31 - (i >> 3) : (i >> 3);
  return (k[byte] >> (i & 7)) & 1;
}

// r = k * p for a 64-bit scalar (the positions i of the commitment polynomial), left to right
template <class C>
EC_HD void small_scalar_mul(typename C::Point& r, const typename C::Point& p, uint64_t k, int nbits) {
  C::identity(r);
  for (int i = nbits - 1; i >= 0; --i) {
    C::dbl(r, r);
    if ((k >> i) & 1) C::add(r, r, p);
  }
}

// the same through the non-adjacent form of k (k < 2^62): a third of the digits are non-zero instead of half, negation is free.
// NAF digit i = bit i+1 of 3k minus bit i+1 of k.  One addition site (the operand is selected).
template <class C>
EC_HD void small_scalar_mul_naf(typename C::Point& r, const typename C::Point& p, uint64_t k) {
  const uint64_t x3 = 3 * k;
  const uint64_t plus = (x3 & ~k) >> 1, minus = (k & ~x3) >> 1, any = plus | minus;
  const int nbits = any == 0 ? 0 : 64 - __builtin_clzll(any);
  typename C::Point np, sel;
  C::neg(np, p);
  C::identity(r);
  for (int i = nbits - 1; i >= 0; --i) {
    C::dbl(r, r);
    if ((any >> i) & 1) {
      sel = p;
      C::cmov(sel, np, (minus >> i) & 1);
      C::add(r, r, sel);
    }
  }
}

// r = k1 * p1 + k2 * p2 (Group::exp twice + Group::mul, dleq.rs:75-81), interleaved bit by bit with
// the joint table {p1, p2, p1 + p2}; k2 may be null (plain Group::exp).
template <class C>
EC_HD void dual_mul(typename C::Point& r, const typename C::Point& p1, const uint8_t* k1, const typename C::Point& p2,
                    const uint8_t* k2) {
  typename C::Point s, id, sel;
  C::add(s, p1, p2);
  C::identity(id);
  C::identity(r);
  for (int i = 255; i >= 0; --i) {
    C::dbl(r, r);
    const u32 b1 = scalar_bit<C>(k1, i);
    const u32 b2 = k2 ? scalar_bit<C>(k2, i) : 0u;
    sel = id;
    C::cmov(sel, p1, b1 && !b2);
    C::cmov(sel, p2, !b1 && b2);
    C::cmov(sel, s, b1 && b2);
    C::add(r, r, sel);
  }
}

// r = k * p for a 256-bit scalar given as 8 little-endian 32-bit words (double and always-add, branch-free)
template <class C>
EC_HD void limb_scalar_mul(typename C::Point& r, const typename C::Point& p, const u32 (&k)[8]) {
  C::identity(r);
  for (int i = 255; i >= 0; --i) {
    C::dbl(r, r);
    typename C::Point sel;
    C::identity(sel);
    C::cmov(sel, p, (k[i >> 5] >> (i & 31)) & 1);
    C::add(r, r, sel);
  }
}

// 32-byte scalar in the curve's byte order -> little-endian words
template <class C>
EC_HD void scalar_words(u32 (&w)[8], const uint8_t* k) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    u32 v = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int byte_le = 4 * i + b;
      v |= (u32)k[C::SCALAR_BIG_ENDIAN ? 31 - byte_le : byte_le] << (8 * b);
    }
    w[i] = v;
  }
}


// ---- signed 4-bit windows ---------------------------------------------------------------------------------------
// k' = k + sum_{w < 64} 8 * 16^w  (257 bits, 9 words): nibble w of k' minus 8 is the signed digit d_w in [-8, 7] for
// w < 64, nibble 64 (0 or 1) is the top digit; sum d_w 16^w = k.  No carry chain between digits at use.
template <class C>
EC_HD void recode_signed4(u32 (&kp)[9], const uint8_t* k) {
  u32 w[8];
  scalar_words<C>(w, k);
  u64 carry = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    carry += (u64)w[i] + 0x88888888u;
    kp[i] = (u32)carry;
    carry >>= 32;
  }
  kp[8] = (u32)carry;
}
// signed digit w (0..64) from the word holding its nibble
EC_HD int signed_digit4(u32 word, int w) {
  const int nib = (int)((word >> (4 * (w & 7))) & 15u);
  return w == 64 ? nib : nib - 8;
}

// table of one point: entries (i + 1) * P, i < 8, in cached form (4 doublings, 3 additions)
template <class C>
EC_HD void build_cached_table(typename C::Cached (&tab)[8], const typename C::Point& p) {
  typename C::Point m[8];
  m[0] = p;
  C::dbl(m[1], m[0]);
  C::add(m[2], m[1], m[0]);
  C::dbl(m[3], m[1]);
  C::add(m[4], m[3], m[0]);
  C::dbl(m[5], m[2]);
  C::add(m[6], m[5], m[0]);
  C::dbl(m[7], m[3]);
#pragma unroll
  for (int i = 0; i < 8; ++i) C::to_cached(tab[i], m[i]);
}

// the same, handing each entry to `store(i, cached)` as soon as it exists (three points live at a time)
template <class C, class Store>
EC_HD void build_cached_table_streamed(const typename C::Point& p, Store store) {
  typename C::Point a, b, c;
  typename C::Cached e;
  C::to_cached(e, p); store(0, e);
  C::dbl(a, p);       C::to_cached(e, a); store(1, e);      // 2P
  C::add(b, a, p);    C::to_cached(e, b); store(2, e);      // 3P
  C::dbl(a, a);       C::to_cached(e, a); store(3, e);      // 4P
  C::add(c, a, p);    C::to_cached(e, c); store(4, e);      // 5P
  C::dbl(b, b);       C::to_cached(e, b); store(5, e);      // 6P
  C::add(c, b, p);    C::to_cached(e, c); store(6, e);      // 7P
  C::dbl(a, a);       C::to_cached(e, a); store(7, e);      // 8P
}

// acc += d * P with P's table entry for |d| supplied by `load(index 0..7)`; d == 0 leaves acc alone
template <class C, class Load>
EC_HD void add_signed_digit(typename C::Point& acc, int d, Load load) {
  const int mag = d < 0 ? -d : d;
  typename C::Cached e;
  load(e, mag > 0 ? mag - 1 : 0);
  typename C::Point sum;
  C::add_cached(sum, acc, e, d < 0);
  C::cmov(acc, sum, mag != 0);
}
template <class C, class Load>
EC_HD void add_signed_digit_affine(typename C::Point& acc, int d, Load load) {
  const int mag = d < 0 ? -d : d;
  typename C::Affine e;
  load(e, mag > 0 ? mag - 1 : 0);
  typename C::Point sum;
  C::add_affine(sum, acc, e, d < 0);
  C::cmov(acc, sum, mag != 0);
}

}  // namespace ec
