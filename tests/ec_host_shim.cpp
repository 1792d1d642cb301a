// Host build of the kernels' EC arithmetic (mpvss_rs_amd/csrc/ec_curves.h) for CPU unit tests.
// Test infrastructure: compiled by tests/test_ec_host.py with g++, never shipped.
#include "../mpvss_rs_amd/csrc/ec_curves.h"

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
static int t_gen(uint8_t* out) {
  typename C::Point g;
  C::generator(g);
  C::encode(out, g);
  return 0;
}

extern "C" {
int ec_dual(int curve, const uint8_t* p1, const uint8_t* k1, const uint8_t* p2, const uint8_t* k2, uint8_t* out) {
  return curve == 0 ? t_dual<Secp>(p1, k1, p2, k2, out) : t_dual<Ristretto>(p1, k1, p2, k2, out);
}
int ec_add(int curve, const uint8_t* p1, const uint8_t* p2, uint8_t* out, int use_dbl) {
  return curve == 0 ? t_add<Secp>(p1, p2, out, use_dbl) : t_add<Ristretto>(p1, p2, out, use_dbl);
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
