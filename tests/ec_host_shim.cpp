// Host build of the kernels' EC arithmetic (mpvss_rs_amd/csrc/ec_curves.h) for CPU unit tests.
// Test infrastructure: compiled by tests/test_ec_host.py with g++, never shipped.
#include "../mpvss_rs_amd/csrc/ec_curves.h"
#include "../mpvss_rs_amd/csrc/ec_glv.h"

using namespace ec;

template <class C>
static int t_dual(const uint8_t* p1, const uint8_t* k1, const uint8_t* p2, const uint8_t* k2, uint8_t* out) {
  typename C::Point a, b, r;
  if (!C::decode(a, p1)) return -1;
  if (p2) { if (!C::decode(b, p2)) return -2; } else C::identity(b);
  dual_mul<C>(r, a, k1, b, p2 ? k2 : nullptr);
  C::encode(out, r);
  return 0;
}
template <class C>
static int t_add(const uint8_t* p1, const uint8_t* p2, uint8_t* out, int use_dbl) {
  typename C::Point a, b, r;
  if (!C::decode(a, p1)) return -1;
  if (!C::decode(b, p2)) return -2;
  if (use_dbl) C::dbl(r, a); else C::add(r, a, b);
  C::encode(out, r);
  return 0;
}
template <class C>
static int t_small(const uint8_t* p, uint64_t k, uint8_t* out) {
  typename C::Point a, r;
  if (!C::decode(a, p)) return -1;
  int nb = 0;
  while (nb < 64 && (k >> nb)) ++nb;
  small_scalar_mul<C>(r, a, k, nb);
  C::encode(out, r);
  return 0;
}
template <class C>
static int t_small_naf(const uint8_t* p, uint64_t k, uint8_t* out) {
  typename C::Point a, r;
  if (!C::decode(a, p)) return -1;
  small_scalar_mul_naf<C>(r, a, k);
  C::encode(out, r);
  return 0;
}
template <class C>
static int t_gen(uint8_t* out) {
  typename C::Point g;
  C::generator(g);
  C::encode(out, g);
  return 0;
}

// the windowed paths of ec_kernels.hip: signed 4-bit Straus over per-point cached tables, fixed-base comb for the
// generator (65 windows x 8 packed affine entries), batched SEC1 encoding
template <class C>
static int t_dual_win(const uint8_t* p1, const uint8_t* k1, const uint8_t* p2, const uint8_t* k2, uint8_t* out) {
  typename C::Point a, b, r;
  if (!C::decode(a, p1)) return -1;
  if (p2) { if (!C::decode(b, p2)) return -2; } else C::identity(b);
  typename C::Cached t1[8], t2[8];
  build_cached_table<C>(t1, a);
  build_cached_table_streamed<C>(b, [&](int i, const typename C::Cached& e) { t2[i] = e; });
  u32 kp1[9], kp2[9];
  recode_signed4<C>(kp1, k1);
  if (p2) recode_signed4<C>(kp2, k2);
  C::identity(r);
  for (int w = 64; w >= 0; --w) {
    if (w != 64) for (int i = 0; i < 4; ++i) C::dbl(r, r);
    add_signed_digit<C>(r, signed_digit4(kp1[w >> 3], w), [&](typename C::Cached& e, int i) { e = t1[i]; });
    if (p2) add_signed_digit<C>(r, signed_digit4(kp2[w >> 3], w), [&](typename C::Cached& e, int i) { e = t2[i]; });
  }
  C::encode(out, r);
  return 0;
}
// comb[w][i] = (i + 1) * 16^w * G, packed affine; returns k * G
template <class C>
static int t_comb(const uint8_t* k, uint8_t* out) {
  static u32* comb = nullptr;
  constexpr int AW = C::AFFINE_PACKED_WORDS;
  if (!comb) {
    comb = new u32[65 * 8 * AW];
    typename C::Point base, m;
    C::generator(base);
    for (int w = 0; w < 65; ++w) {
      m = base;
      for (int i = 0; i < 8; ++i) {
        typename C::Affine a;
        C::to_affine(a, m);
        C::pack_affine(comb + (w * 8 + i) * AW, a);
        C::add(m, m, base);
      }
      for (int i = 0; i < 4; ++i) C::dbl(base, base);
    }
  }
  u32 kp[9];
  recode_signed4<C>(kp, k);
  typename C::Point r;
  C::identity(r);
  for (int w = 0; w < 65; ++w)
    add_signed_digit_affine<C>(r, signed_digit4(kp[w >> 3], w),
                               [&](typename C::Affine& e, int i) { C::unpack_affine(e, comb + (w * 8 + i) * AW); });
  C::encode(out, r);
  return 0;
}

// secp256k1 with the GLV endomorphism (ec_glv.h, what k_secp_dual_win runs): every scalar split into two 128-bit halves, 33 signed
// windows, the table of phi(P) made from P's on the fly
static int t_dual_win_glv(const uint8_t* p1, const uint8_t* k1, const uint8_t* p2, const uint8_t* k2, uint8_t* out) {
  Secp::Point a, b, r;
  if (!Secp::decode(a, p1)) return -1;
  if (p2) { if (!Secp::decode(b, p2)) return -2; } else Secp::identity(b);
  Secp::Cached tabs[2][8];
  build_cached_table<Secp>(tabs[0], a);
  build_cached_table<Secp>(tabs[1], b);
  GlvHalf h[2][2];
  u32 kw[8];
  scalar_words<Secp>(kw, k1);
  secp_glv_split(h[0], kw);
  if (p2) { scalar_words<Secp>(kw, k2); secp_glv_split(h[1], kw); }
  Secp::identity(r);
  for (int w = 32; w >= 0; --w) {
    if (w != 32) for (int i = 0; i < 4; ++i) Secp::dbl(r, r);
    for (int t = 0; t < (p2 ? 2 : 1); ++t)
      for (int j = 0; j < 2; ++j)
        add_signed_digit<Secp>(r, glv_digit(h[t][j], w), [&](Secp::Cached& e, int i) { e = tabs[t][i]; if (j) secp_phi_cached(e); });
  }
  Secp::encode(out, r);
  return 0;
}

extern "C" {
int ec_dual_win_glv(const uint8_t* p1, const uint8_t* k1, const uint8_t* p2, const uint8_t* k2, uint8_t* out) {
  return t_dual_win_glv(p1, k1, p2, k2, out);
}
// the split alone: out = |k1| (17 bytes LE) | sign | |k2| (17 bytes LE) | sign  (magnitudes from the recoded words)
int secp_glv_split_bytes(const uint8_t* k_be32, uint8_t* out36) {
  u32 kw[8];
  scalar_words<Secp>(kw, k_be32);
  GlvHalf h[2];
  secp_glv_split(h, kw);
  for (int j = 0; j < 2; ++j) {
    // undo the + 0x88..8 of the recoding: magnitude = kp - sum 8 * 16^w
    uint64_t borrow = 0;
    u32 m[5];
    for (int i = 0; i < 5; ++i) {
      const uint64_t d = (uint64_t)h[j].kp[i] - (i < 4 ? 0x88888888u : 0u) - borrow;
      m[i] = (u32)d;
      borrow = (d >> 63) & 1;
    }
    for (int i = 0; i < 17; ++i) out36[18 * j + i] = (uint8_t)(m[i / 4] >> (8 * (i % 4)));
    out36[18 * j + 17] = h[j].neg ? 1 : 0;
  }
  return 0;
}
int ec_dual_win(int curve, const uint8_t* p1, const uint8_t* k1, const uint8_t* p2, const uint8_t* k2, uint8_t* out) {
  return curve == 0 ? t_dual_win<Secp>(p1, k1, p2, k2, out) : t_dual_win<Ristretto>(p1, k1, p2, k2, out);
}
int ec_comb(int curve, const uint8_t* k, uint8_t* out) { return curve == 0 ? t_comb<Secp>(k, out) : t_comb<Ristretto>(k, out); }
// four SEC1 points (33 zero bytes = identity) re-encoded through the shared-inversion path
int secp_encode_batch4(const uint8_t* in, uint8_t* out) {
  Secp::Point p[4];
  bool live[4] = {true, true, true, true};
  for (int i = 0; i < 4; ++i) {
    if (!Secp::decode(p[i], in + 33 * i)) return -1;
    Secp::Point t;                       // de-normalise: multiply by something so that Z != 1
    Secp::dbl(t, p[i]);
    Secp::add(p[i], t, p[i]);            // 3 P
  }
  Secp::encode_batch<4>(out, 33, [&](int i, Secp::Point& q) { q = p[i]; }, [&](int i) { return live[i]; });
  return 0;
}
int ec_dual(int curve, const uint8_t* p1, const uint8_t* k1, const uint8_t* p2, const uint8_t* k2, uint8_t* out) {
  return curve == 0 ? t_dual<Secp>(p1, k1, p2, k2, out) : t_dual<Ristretto>(p1, k1, p2, k2, out);
}
int ec_add(int curve, const uint8_t* p1, const uint8_t* p2, uint8_t* out, int use_dbl) {
  return curve == 0 ? t_add<Secp>(p1, p2, out, use_dbl) : t_add<Ristretto>(p1, p2, out, use_dbl);
}
int ec_small_naf(int curve, const uint8_t* p, uint64_t k, uint8_t* out) {
  return curve == 0 ? t_small_naf<Secp>(p, k, out) : t_small_naf<Ristretto>(p, k, out);
}
int ec_small(int curve, const uint8_t* p, uint64_t k, uint8_t* out) {
  return curve == 0 ? t_small<Secp>(p, k, out) : t_small<Ristretto>(p, k, out);
}
int ec_gen(int curve, uint8_t* out) { return curve == 0 ? t_gen<Secp>(out) : t_gen<Ristretto>(out); }
int ec_decode_ok(int curve, const uint8_t* p) {
  if (curve == 0) { Secp::Point a; return Secp::decode(a, p); }
  Ristretto::Point a; return Ristretto::decode(a, p);
}
// field self-test hooks: r = a*b, a^2, a-b (canonical 32-byte LE)
// mul (op 0) / sqr (op 1) on RAW limbs (10 x u32, any value up to the documented 2^30 input bound); writes the
// canonical result and returns the largest output limb before canonicalisation (the documented output bound is checked
// by the caller)
uint32_t fe_limb_op(int curve, int op, const uint32_t* a10, const uint32_t* b10, uint8_t* out) {
  Fe a, b, r;
  for (int i = 0; i < 10; ++i) { a.v[i] = a10[i]; b.v[i] = b10[i]; }
  uint32_t top = 0;
  if (curve == 0) {
    typedef F<PrimeSecp> Fp;
    if (op == 0) Fp::mul(r, a, b); else Fp::sqr(r, a);
    for (int i = 0; i < 10; ++i) top = r.v[i] > top ? r.v[i] : top;
    Fp::canon(r); Fp::to_le32(out, r);
  } else {
    typedef F<PrimeEd> Fp;
    if (op == 0) Fp::mul(r, a, b); else Fp::sqr(r, a);
    for (int i = 0; i < 10; ++i) top = r.v[i] > top ? r.v[i] : top;
    Fp::canon(r); Fp::to_le32(out, r);
  }
  return top;
}
int fe_op(int curve, int op, const uint8_t* a32, const uint8_t* b32, uint8_t* out) {
  Fe a, b, r;
  if (curve == 0) {
    typedef F<PrimeSecp> Fp;
    Fp::from_le32_raw(a, a32); Fp::from_le32_raw(b, b32);
    if (op == 0) Fp::mul(r, a, b); else if (op == 1) Fp::sqr(r, a); else if (op == 2) Fp::sub(r, a, b);
    else if (op == 3) Secp::invert(r, a); else Fp::addc(r, a, b);
    Fp::canon(r); Fp::to_le32(out, r);
  } else {
    typedef F<PrimeEd> Fp;
    Fp::from_le32_raw(a, a32); Fp::from_le32_raw(b, b32);
    if (op == 0) Fp::mul(r, a, b); else if (op == 1) Fp::sqr(r, a); else if (op == 2) Fp::sub(r, a, b);
    else if (op == 3) Ristretto::pow22523(r, a); else Fp::addc(r, a, b);
    Fp::canon(r); Fp::to_le32(out, r);
  }
  return 0;
}
}
