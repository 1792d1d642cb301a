/* ec_ref.c -- plain-C CPU restatement of the reference's verification path for its two curve groups.
 *
 * TEST INFRASTRUCTURE ONLY: a checker for tests/ and the `cpu_baseline` leg of bench.py's `ec` objects.
 * Nothing in the product path (mpvss_rs_amd/, libmpvss_hip.so) links or calls this file.
 *
 * It executes the reference's operation sequence for ONE share as written:
 *   verify_distribution_shares            src/participant.rs:1384-1442 (secp256k1), 1827-1885 (ristretto255)
 *     X_i loop: exp, mul, scalar_mul       src/participant.rs:1404-1417, 1847-1860  (t scalar multiplications with the
 *                                          full-width exponents i^j mod order, t point additions)
 *     Verifier::commitments (4 exp, 2 mul) src/dleq.rs:66-84
 *   Secp256k1Group::exp / mul              src/groups/secp256k1.rs:91-107  (every result converted to AFFINE: one field
 *                                          inversion per exp and per mul, as `.into()` / `to_affine()` do)
 *   Ristretto255Group::exp / mul           src/groups/ristretto255.rs:161-177
 *   encodings                              SEC1 compressed (secp256k1.rs:133-152), RFC 9496 4.3 (ristretto255.rs:207-220)
 *
 * The curve arithmetic itself lives in crates the reference does not vendor (k256 = "0.13", curve25519-dalek = "4",
 * Cargo.toml:24,27; no Cargo.lock).  Those crates run constant-time windowed multiplications; restated here is the
 * textbook form of the same maps -- left-to-right double-and-add over all 256 bits on Jacobian (secp256k1) /
 * extended twisted Edwards (curve25519) coordinates -- which yields the same group elements and therefore the same
 * canonical bytes.  A scalar multiplication costs about 2-3x what the crates' tuned code does; bench.py says so
 * next to the number.
 *
 * Independent of the engine by construction: 4 x 64-bit limbs with unsigned __int128 products and "2^256 = c" folding
 * here, 10 x 26-bit limbs on the GPU; Jacobian formulas with explicit special cases here, complete projective
 * formulas there.  Cross-checked against oracle/mpvss_oracle.py (Python integers) in tests/test_oracle_c.py.
 */
#include <stdint.h>
#include <string.h>

typedef uint64_t u64;
typedef unsigned __int128 u128;

/* ---- 256-bit field with p = 2^256 - c (secp256k1: c = 2^32 + 977) or p = 2^255 - 19 (2^256 = 38 mod p) ---------- */
typedef struct { u64 v[4]; } fe;
typedef struct { u64 p[4]; u64 c; } field;   /* c: 2^256 mod p */

static const field F_SECP = {{0xFFFFFFFEFFFFFC2FULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL}, 0x1000003D1ULL};
static const field F_ED = {{0xFFFFFFFFFFFFFFEDULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL, 0x7FFFFFFFFFFFFFFFULL}, 38};

static int ge4(const u64* a, const u64* b) {
  for (int i = 3; i >= 0; --i)
    if (a[i] != b[i]) return a[i] > b[i];
  return 1;
}
static u64 sub4(u64* r, const u64* a, const u64* b) {
  u64 borrow = 0;
  for (int i = 0; i < 4; ++i) {
    u128 d = (u128)a[i] - b[i] - borrow;
    r[i] = (u64)d;
    borrow = (u64)(d >> 64) & 1;
  }
  return borrow;
}
static u64 add4(u64* r, const u64* a, const u64* b) {
  u64 carry = 0;
  for (int i = 0; i < 4; ++i) {
    u128 s = (u128)a[i] + b[i] + carry;
    r[i] = (u64)s;
    carry = (u64)(s >> 64);
  }
  return carry;
}
/* r += k (k < 2^64 * small), folding any carry out of 2^256 again */
static void add_small_fold(const field* f, u64* r, u128 k) {
  while (k) {
    u128 s = (u128)r[0] + (u64)k;
    r[0] = (u64)s;
    u128 c = (s >> 64) + (k >> 64);
    for (int i = 1; i < 4; ++i) {
      s = (u128)r[i] + (u64)c;
      r[i] = (u64)s;
      c = (c >> 64) + (s >> 64);
    }
    k = c * f->c;          /* overflow beyond 2^256 folds back */
  }
}
static void fe_canon(const field* f, fe* a) {
  while (ge4(a->v, f->p)) sub4(a->v, a->v, f->p);
}
static void fe_add(const field* f, fe* r, const fe* a, const fe* b) {
  u64 carry = add4(r->v, a->v, b->v);
  if (carry) add_small_fold(f, r->v, f->c);
}
static void fe_sub(const field* f, fe* r, const fe* a, const fe* b) {
  fe bb = *b;
  fe_canon(f, &bb);
  fe aa = *a;
  fe_canon(f, &aa);
  if (sub4(r->v, aa.v, bb.v)) add4(r->v, r->v, f->p);
}
static void fe_mul(const field* f, fe* r, const fe* a, const fe* b) {
  u64 t[8] = {0};
  for (int i = 0; i < 4; ++i) {
    u64 carry = 0;
    for (int j = 0; j < 4; ++j) {
      u128 s = (u128)a->v[i] * b->v[j] + t[i + j] + carry;
      t[i + j] = (u64)s;
      carry = (u64)(s >> 64);
    }
    t[i + 4] = carry;
  }
  /* lo + hi * c */
  u64 res[4];
  u128 carry = 0;
  for (int i = 0; i < 4; ++i) {
    u128 s = (u128)t[4 + i] * f->c + t[i] + (u64)carry;
    res[i] = (u64)s;
    carry = (s >> 64) + (carry >> 64);
  }
  memcpy(r->v, res, sizeof(res));
  add_small_fold(f, r->v, carry * f->c);
}
static void fe_sqr(const field* f, fe* r, const fe* a) { fe_mul(f, r, a, a); }
static void fe_set(fe* r, u64 x) { r->v[0] = x; r->v[1] = r->v[2] = r->v[3] = 0; }
static int fe_is_zero(const field* f, const fe* a) {
  fe t = *a;
  fe_canon(f, &t);
  return (t.v[0] | t.v[1] | t.v[2] | t.v[3]) == 0;
}
static int fe_eq(const field* f, const fe* a, const fe* b) {
  fe x = *a, y = *b;
  fe_canon(f, &x);
  fe_canon(f, &y);
  return memcmp(x.v, y.v, 32) == 0;
}
static void fe_neg(const field* f, fe* r, const fe* a) {
  fe z;
  fe_set(&z, 0);
  fe_sub(f, r, &z, a);
}
/* r = a^e, e as 4 little-endian 64-bit words */
static void fe_pow(const field* f, fe* r, const fe* a, const u64* e) {
  fe acc;
  fe_set(&acc, 1);
  for (int i = 255; i >= 0; --i) {
    fe_sqr(f, &acc, &acc);
    if ((e[i >> 6] >> (i & 63)) & 1) fe_mul(f, &acc, &acc, a);
  }
  *r = acc;
}
static void fe_inv(const field* f, fe* r, const fe* a) {
  u64 e[4], two[4] = {2, 0, 0, 0};
  sub4(e, f->p, two);
  fe_pow(f, r, a, e);
}
static int fe_is_odd(const field* f, const fe* a) {
  fe t = *a;
  fe_canon(f, &t);
  return (int)(t.v[0] & 1);
}
static void fe_from_le(fe* r, const uint8_t* b) {
  for (int i = 0; i < 4; ++i) {
    u64 w = 0;
    for (int k = 0; k < 8; ++k) w |= (u64)b[8 * i + k] << (8 * k);
    r->v[i] = w;
  }
}
static void fe_to_le(const field* f, uint8_t* b, const fe* a) {
  fe t = *a;
  fe_canon(f, &t);
  for (int i = 0; i < 4; ++i)
    for (int k = 0; k < 8; ++k) b[8 * i + k] = (uint8_t)(t.v[i] >> (8 * k));
}

/* ---- scalars mod the group order (256-bit): products by shift-and-subtract reduction ---------------------------- */
static const u64 N_SECP[4] = {0xBFD25E8CD0364141ULL, 0xBAAEDCE6AF48A03BULL, 0xFFFFFFFFFFFFFFFEULL, 0xFFFFFFFFFFFFFFFFULL};
static const u64 L_ED[4] = {0x5812631A5CF5D3EDULL, 0x14DEF9DEA2F79CD6ULL, 0, 0x1000000000000000ULL};

static void sc_mulmod(u64* r, const u64* a, const u64* b, const u64* n) {
  u64 t[8] = {0};
  for (int i = 0; i < 4; ++i) {
    u64 carry = 0;
    for (int j = 0; j < 4; ++j) {
      u128 s = (u128)a[i] * b[j] + t[i + j] + carry;
      t[i + j] = (u64)s;
      carry = (u64)(s >> 64);
    }
    t[i + 4] = carry;
  }
  u64 rem[5] = {0};
  for (int bit = 511; bit >= 0; --bit) {
    for (int k = 4; k > 0; --k) rem[k] = (rem[k] << 1) | (rem[k - 1] >> 63);
    rem[0] = (rem[0] << 1) | ((t[bit >> 6] >> (bit & 63)) & 1);
    if (rem[4] || ge4(rem, n)) {
      u64 borrow = sub4(rem, rem, n);
      rem[4] -= borrow;
    }
  }
  memcpy(r, rem, 32);
}
static void sc_from_bytes(u64* r, const uint8_t* b, int big_endian) {
  for (int i = 0; i < 4; ++i) {
    u64 w = 0;
    for (int k = 0; k < 8; ++k) w |= (u64)b[big_endian ? 31 - (8 * i + k) : 8 * i + k] << (8 * k);
    r[i] = w;
  }
}

/* ---- secp256k1: y^2 = x^3 + 7, Jacobian coordinates, inf flag --------------------------------------------------- */
typedef struct { fe X, Y, Z; int inf; } jac;
#define FS (&F_SECP)

static void jac_dbl(jac* r, const jac* p) {
  if (p->inf || fe_is_zero(FS, &p->Y)) { r->inf = 1; return; }
  fe a, b, c, d, e, f2, t, x3, y3, z3;
  fe_sqr(FS, &a, &p->X);
  fe_sqr(FS, &b, &p->Y);
  fe_sqr(FS, &c, &b);
  fe_add(FS, &t, &p->X, &b);
  fe_sqr(FS, &t, &t);
  fe_sub(FS, &t, &t, &a);
  fe_sub(FS, &t, &t, &c);
  fe_add(FS, &d, &t, &t);                 /* D = 2((X+B)^2 - A - C) */
  fe_add(FS, &e, &a, &a);
  fe_add(FS, &e, &e, &a);                 /* E = 3A */
  fe_sqr(FS, &f2, &e);
  fe_add(FS, &t, &d, &d);
  fe_sub(FS, &x3, &f2, &t);               /* X3 = F - 2D */
  fe_sub(FS, &t, &d, &x3);
  fe_mul(FS, &y3, &e, &t);
  fe_add(FS, &t, &c, &c);
  fe_add(FS, &t, &t, &t);
  fe_add(FS, &t, &t, &t);
  fe_sub(FS, &y3, &y3, &t);               /* Y3 = E(D - X3) - 8C */
  fe_mul(FS, &z3, &p->Y, &p->Z);
  fe_add(FS, &z3, &z3, &z3);
  r->X = x3; r->Y = y3; r->Z = z3; r->inf = 0;
}
static void jac_add(jac* r, const jac* p, const jac* q) {
  if (p->inf) { *r = *q; return; }
  if (q->inf) { *r = *p; return; }
  fe z1z1, z2z2, u1, u2, s1, s2, h, rr, t, hh, hhh, v, x3, y3, z3;
  fe_sqr(FS, &z1z1, &p->Z);
  fe_sqr(FS, &z2z2, &q->Z);
  fe_mul(FS, &u1, &p->X, &z2z2);
  fe_mul(FS, &u2, &q->X, &z1z1);
  fe_mul(FS, &t, &q->Z, &z2z2);
  fe_mul(FS, &s1, &p->Y, &t);
  fe_mul(FS, &t, &p->Z, &z1z1);
  fe_mul(FS, &s2, &q->Y, &t);
  if (fe_eq(FS, &u1, &u2)) {
    if (fe_eq(FS, &s1, &s2)) { jac_dbl(r, p); return; }
    r->inf = 1;
    return;
  }
  fe_sub(FS, &h, &u2, &u1);
  fe_sub(FS, &rr, &s2, &s1);
  fe_sqr(FS, &hh, &h);
  fe_mul(FS, &hhh, &hh, &h);
  fe_mul(FS, &v, &u1, &hh);
  fe_sqr(FS, &x3, &rr);
  fe_sub(FS, &x3, &x3, &hhh);
  fe_sub(FS, &x3, &x3, &v);
  fe_sub(FS, &x3, &x3, &v);
  fe_sub(FS, &t, &v, &x3);
  fe_mul(FS, &y3, &rr, &t);
  fe_mul(FS, &t, &s1, &hhh);
  fe_sub(FS, &y3, &y3, &t);
  fe_mul(FS, &z3, &p->Z, &q->Z);
  fe_mul(FS, &z3, &z3, &h);
  r->X = x3; r->Y = y3; r->Z = z3; r->inf = 0;
}
/* to affine: Z = 1 (one field inversion), as Secp256k1Group::exp / mul return AffinePoint (secp256k1.rs:99,106) */
static void jac_affine(jac* p) {
  if (p->inf) return;
  fe zi, zi2, zi3;
  fe_inv(FS, &zi, &p->Z);
  fe_sqr(FS, &zi2, &zi);
  fe_mul(FS, &zi3, &zi2, &zi);
  fe_mul(FS, &p->X, &p->X, &zi2);
  fe_mul(FS, &p->Y, &p->Y, &zi3);
  fe_set(&p->Z, 1);
}
static int secp_decode(jac* r, const uint8_t* b) {       /* secp256k1.rs:138-152; 33 zero bytes = identity */
  int allzero = 1;
  for (int i = 0; i < 33; ++i) allzero &= (b[i] == 0);
  if (allzero) { r->inf = 1; return 1; }
  if (b[0] != 2 && b[0] != 3) return 0;
  uint8_t le[32];
  for (int i = 0; i < 32; ++i) le[i] = b[32 - i];
  fe x, y2, y, t, seven;
  fe_from_le(&x, le);
  if (ge4(x.v, F_SECP.p)) return 0;
  fe_sqr(FS, &t, &x);
  fe_mul(FS, &y2, &t, &x);
  fe_set(&seven, 7);
  fe_add(FS, &y2, &y2, &seven);
  u64 e[4], one[4] = {1, 0, 0, 0};
  add4(e, F_SECP.p, one);                                  /* (p + 1) / 4: p + 1 wraps to 2^256 - c + 1, fits */
  for (int i = 0; i < 3; ++i) e[i] = (e[i] >> 2) | (e[i + 1] << 62);
  e[3] >>= 2;
  fe_pow(FS, &y, &y2, e);
  fe_sqr(FS, &t, &y);
  if (!fe_eq(FS, &t, &y2)) return 0;
  if (fe_is_odd(FS, &y) != (b[0] & 1)) fe_neg(FS, &y, &y);
  r->X = x; r->Y = y; fe_set(&r->Z, 1); r->inf = 0;
  return 1;
}
static void secp_encode(uint8_t* out, const jac* p0) {     /* secp256k1.rs:133-136 */
  if (p0->inf) { memset(out, 0, 33); return; }
  jac p = *p0;
  jac_affine(&p);
  uint8_t le[32];
  fe_to_le(FS, le, &p.X);
  out[0] = (uint8_t)(2 + fe_is_odd(FS, &p.Y));
  for (int i = 0; i < 32; ++i) out[1 + i] = le[31 - i];
}
/* Secp256k1Group::exp (secp256k1.rs:91-100): k * P, result affine */
static void secp_exp(jac* r, const jac* p, const u64* k) {
  jac acc;
  acc.inf = 1;
  for (int i = 255; i >= 0; --i) {
    jac_dbl(&acc, &acc);
    if ((k[i >> 6] >> (i & 63)) & 1) jac_add(&acc, &acc, p);
  }
  jac_affine(&acc);
  *r = acc;
}
/* Secp256k1Group::mul (secp256k1.rs:102-107): P + Q, result affine */
static void secp_mul(jac* r, const jac* a, const jac* b) {
  jac t;
  jac_add(&t, a, b);
  jac_affine(&t);
  *r = t;
}

/* ---- ristretto255 over -x^2 + y^2 = 1 + d x^2 y^2, extended coordinates ----------------------------------------- */
typedef struct { fe X, Y, Z, T; } ext;
#define FE (&F_ED)
static fe ED_D, ED_2D, ED_SQRT_M1, ED_INVSQRT_A_MINUS_D;
static ext ED_BASE;
static int ed_ready = 0;

static void ext_identity(ext* r) { fe_set(&r->X, 0); fe_set(&r->Y, 1); fe_set(&r->Z, 1); fe_set(&r->T, 0); }
static void ext_add(ext* r, const ext* p, const ext* q) {  /* add-2008-hwcd-3, complete for a = -1 */
  fe a, b, c, d, e, f, g, h, t;
  fe_sub(FE, &a, &p->Y, &p->X);
  fe_sub(FE, &t, &q->Y, &q->X);
  fe_mul(FE, &a, &a, &t);
  fe_add(FE, &b, &p->Y, &p->X);
  fe_add(FE, &t, &q->Y, &q->X);
  fe_mul(FE, &b, &b, &t);
  fe_mul(FE, &c, &p->T, &q->T);
  fe_mul(FE, &c, &c, &ED_2D);
  fe_mul(FE, &d, &p->Z, &q->Z);
  fe_add(FE, &d, &d, &d);
  fe_sub(FE, &e, &b, &a);
  fe_sub(FE, &f, &d, &c);
  fe_add(FE, &g, &d, &c);
  fe_add(FE, &h, &b, &a);
  fe_mul(FE, &r->X, &e, &f);
  fe_mul(FE, &r->Y, &g, &h);
  fe_mul(FE, &r->T, &e, &h);
  fe_mul(FE, &r->Z, &f, &g);
}
static void ed_abs(fe* r) {
  if (fe_is_odd(FE, r)) fe_neg(FE, r, r);
}
/* RFC 9496 4.2 */
static int sqrt_ratio_m1(fe* r, const fe* u, const fe* v) {
  fe v3, v7, t, check, nu, nui;
  fe_sqr(FE, &t, v);
  fe_mul(FE, &v3, &t, v);
  fe_sqr(FE, &t, &v3);
  fe_mul(FE, &v7, &t, v);
  fe_mul(FE, &t, u, &v7);
  /* (p - 5) / 8 = 2^252 - 3 */
  u64 e[4] = {0xFFFFFFFFFFFFFFFDULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL, 0x0FFFFFFFFFFFFFFFULL};
  fe_pow(FE, &t, &t, e);
  fe_mul(FE, r, u, &v3);
  fe_mul(FE, r, r, &t);
  fe_sqr(FE, &t, r);
  fe_mul(FE, &check, v, &t);
  fe_neg(FE, &nu, u);
  fe_mul(FE, &nui, &nu, &ED_SQRT_M1);
  const int correct = fe_eq(FE, &check, u), flipped = fe_eq(FE, &check, &nu), flipped_i = fe_eq(FE, &check, &nui);
  if (flipped || flipped_i) fe_mul(FE, r, r, &ED_SQRT_M1);
  ed_abs(r);
  return correct || flipped;
}
static void ed_init(void) {
  if (ed_ready) return;
  fe a, b, one;
  fe_set(&a, 121665);
  fe_set(&b, 121666);
  fe_inv(FE, &b, &b);
  fe_mul(FE, &a, &a, &b);
  fe_neg(FE, &ED_D, &a);                                   /* d = -121665/121666 */
  fe_add(FE, &ED_2D, &ED_D, &ED_D);
  u64 e[4] = {0xFFFFFFFFFFFFFFFBULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL, 0x1FFFFFFFFFFFFFFFULL};   /* (p-1)/4 */
  fe two;
  fe_set(&two, 2);
  fe_pow(FE, &ED_SQRT_M1, &two, e);
  fe_set(&one, 1);
  fe_neg(FE, &a, &one);
  fe_sub(FE, &a, &a, &ED_D);                               /* a - d = -1 - d */
  sqrt_ratio_m1(&ED_INVSQRT_A_MINUS_D, &one, &a);
  /* Ed25519 basepoint (RFC 8032): y = 4/5, x positive ("even") */
  static const uint8_t bx[32] = {0x1a, 0xd5, 0x25, 0x8f, 0x60, 0x2d, 0x56, 0xc9, 0xb2, 0xa7, 0x25, 0x95, 0x60, 0xc7, 0x2c, 0x69,
                                 0x5c, 0xdc, 0xd6, 0xfd, 0x31, 0xe2, 0xa4, 0xc0, 0xfe, 0x53, 0x6e, 0xcd, 0xd3, 0x36, 0x69, 0x21};
  fe_from_le(&ED_BASE.X, bx);
  fe_set(&a, 4);
  fe_set(&b, 5);
  fe_inv(FE, &b, &b);
  fe_mul(FE, &ED_BASE.Y, &a, &b);
  fe_set(&ED_BASE.Z, 1);
  fe_mul(FE, &ED_BASE.T, &ED_BASE.X, &ED_BASE.Y);
  ed_ready = 1;
}
static int rist_decode(ext* r, const uint8_t* b) {         /* RFC 9496 4.3.1, ristretto255.rs:212-220 */
  fe s, ss, u1, u2, u2s, v, t, inv, dx, dy, x, y, one;
  fe_from_le(&s, b);
  if ((b[31] & 0x80) || ge4(s.v, F_ED.p) || (s.v[0] & 1)) return 0;
  fe_set(&one, 1);
  fe_sqr(FE, &ss, &s);
  fe_sub(FE, &u1, &one, &ss);
  fe_add(FE, &u2, &one, &ss);
  fe_sqr(FE, &u2s, &u2);
  fe_sqr(FE, &t, &u1);
  fe_mul(FE, &t, &t, &ED_D);
  fe_neg(FE, &t, &t);
  fe_sub(FE, &v, &t, &u2s);
  fe_mul(FE, &t, &v, &u2s);
  const int was_square = sqrt_ratio_m1(&inv, &one, &t);
  fe_mul(FE, &dx, &inv, &u2);
  fe_mul(FE, &dy, &inv, &dx);
  fe_mul(FE, &dy, &dy, &v);
  fe_mul(FE, &x, &s, &dx);
  fe_add(FE, &x, &x, &x);
  ed_abs(&x);
  fe_mul(FE, &y, &u1, &dy);
  fe_mul(FE, &t, &x, &y);
  if (!was_square || fe_is_odd(FE, &t) || fe_is_zero(FE, &y)) return 0;
  r->X = x; r->Y = y; fe_set(&r->Z, 1); r->T = t;
  return 1;
}
static void rist_encode(uint8_t* out, const ext* p) {      /* RFC 9496 4.3.2, ristretto255.rs:207-210 */
  fe u1, u2, t, inv, den1, den2, zinv, ix, iy, ench, x, y, deninv, s, one;
  fe_add(FE, &u1, &p->Z, &p->Y);
  fe_sub(FE, &t, &p->Z, &p->Y);
  fe_mul(FE, &u1, &u1, &t);
  fe_mul(FE, &u2, &p->X, &p->Y);
  fe_sqr(FE, &t, &u2);
  fe_mul(FE, &t, &t, &u1);
  fe_set(&one, 1);
  sqrt_ratio_m1(&inv, &one, &t);
  fe_mul(FE, &den1, &inv, &u1);
  fe_mul(FE, &den2, &inv, &u2);
  fe_mul(FE, &zinv, &den1, &den2);
  fe_mul(FE, &zinv, &zinv, &p->T);
  fe_mul(FE, &ix, &p->X, &ED_SQRT_M1);
  fe_mul(FE, &iy, &p->Y, &ED_SQRT_M1);
  fe_mul(FE, &ench, &den1, &ED_INVSQRT_A_MINUS_D);
  fe_mul(FE, &t, &p->T, &zinv);
  if (fe_is_odd(FE, &t)) { x = iy; y = ix; deninv = ench; } else { x = p->X; y = p->Y; deninv = den2; }
  fe_mul(FE, &t, &x, &zinv);
  if (fe_is_odd(FE, &t)) fe_neg(FE, &y, &y);
  fe_sub(FE, &t, &p->Z, &y);
  fe_mul(FE, &s, &deninv, &t);
  ed_abs(&s);
  fe_to_le(FE, out, &s);
}
static void rist_exp(ext* r, const ext* p, const u64* k) { /* ristretto255.rs:161-170 */
  ext acc;
  ext_identity(&acc);
  for (int i = 255; i >= 0; --i) {
    ext_add(&acc, &acc, &acc);
    if ((k[i >> 6] >> (i & 63)) & 1) ext_add(&acc, &acc, p);
  }
  *r = acc;
}

/* ---- the reference's per-share work ---------------------------------------------------------------------------- */
/* group: 1 = secp256k1, 2 = ristretto255.  commitments: t encodings; y, Y: encodings; r, c: 32-byte scalars in the
 * group's byte order.  Outputs: encodings of X_i, a1_i, a2_i.  Returns 0, or -1 for an invalid encoding. */
int ec_ref_verify_share_work(int group, const uint8_t* commitments, size_t t, int64_t position, const uint8_t* y,
                             const uint8_t* Y, const uint8_t* r, const uint8_t* c, uint8_t* x_out, uint8_t* a1_out,
                             uint8_t* a2_out) {
  u64 rs[4], cs[4];
  const u64* order = group == 1 ? N_SECP : L_ED;
  sc_from_bytes(rs, r, group == 1);
  sc_from_bytes(cs, c, group == 1);
  u64 pos[4] = {(u64)position, 0, 0, 0};                   /* Scalar::from(position as u64), participant.rs:1419,1862 */
  u64 e[4] = {1, 0, 0, 0};
  if (group == 1) {
    jac X, cj, term, g, py, pY, a, b, a1, a2;
    X.inf = 1;                                             /* identity, participant.rs:1411 */
    for (size_t j = 0; j < t; ++j) {
      if (!secp_decode(&cj, commitments + 33 * j)) return -1;
      secp_exp(&term, &cj, e);                             /* exp(C_j, i^j) */
      secp_mul(&X, &X, &term);
      sc_mulmod(e, e, pos, order);
    }
    static const uint8_t G[33] = {0x02, 0x79, 0xBE, 0x66, 0x7E, 0xF9, 0xDC, 0xBB, 0xAC, 0x55, 0xA0, 0x62, 0x95, 0xCE, 0x87, 0x0B, 0x07,
                                  0x02, 0x9B, 0xFC, 0xDB, 0x2D, 0xCE, 0x28, 0xD9, 0x59, 0xF2, 0x81, 0x5B, 0x16, 0xF8, 0x17, 0x98};
    if (!secp_decode(&g, G) || !secp_decode(&py, y) || !secp_decode(&pY, Y)) return -1;
    secp_exp(&a, &g, rs);                                  /* dleq.rs:75-77 */
    secp_exp(&b, &X, cs);
    secp_mul(&a1, &a, &b);
    secp_exp(&a, &py, rs);                                 /* dleq.rs:79-81 */
    secp_exp(&b, &pY, cs);
    secp_mul(&a2, &a, &b);
    secp_encode(x_out, &X);
    secp_encode(a1_out, &a1);
    secp_encode(a2_out, &a2);
    return 0;
  }
  ed_init();
  ext X, cj, term, py, pY, a, b, a1, a2;
  ext_identity(&X);                                        /* participant.rs:1854 */
  for (size_t j = 0; j < t; ++j) {
    if (!rist_decode(&cj, commitments + 32 * j)) return -1;
    rist_exp(&term, &cj, e);
    ext_add(&X, &X, &term);
    sc_mulmod(e, e, pos, order);
  }
  if (!rist_decode(&py, y) || !rist_decode(&pY, Y)) return -1;
  rist_exp(&a, &ED_BASE, rs);
  rist_exp(&b, &X, cs);
  ext_add(&a1, &a, &b);
  rist_exp(&a, &py, rs);
  rist_exp(&b, &pY, cs);
  ext_add(&a2, &a, &b);
  rist_encode(x_out, &X);
  rist_encode(a1_out, &a1);
  rist_encode(a2_out, &a2);
  return 0;
}

/* out = k * P (encodings), for the unit tests */
int ec_ref_exp(int group, const uint8_t* p_enc, const uint8_t* k, uint8_t* out) {
  u64 ks[4];
  sc_from_bytes(ks, k, group == 1);
  if (group == 1) {
    jac p, r;
    if (!secp_decode(&p, p_enc)) return -1;
    secp_exp(&r, &p, ks);
    secp_encode(out, &r);
    return 0;
  }
  ed_init();
  ext p, r;
  if (!rist_decode(&p, p_enc)) return -1;
  rist_exp(&r, &p, ks);
  rist_encode(out, &r);
  return 0;
}

/* must be called once before threads use the ristretto255 functions (fills the curve constants) */
void ec_ref_init(void) { ed_init(); }
