// Host-side arithmetic modulo the RFC 3526 group-14 prime q (2048 bit), 32 x 64-bit limbs.  Used for the one
// piece of the forward-difference path that is cheaper on the CPU than on the GPU: inverting the t commitments
// (one modular inversion + 3 t products by Montgomery's batch trick, about a millisecond) instead of a
// latency-bound 2048-bit exponentiation on the device (about 20 ms).  Not on the per-share hot path.
#pragma once
#include <stdint.h>
#include <string.h>

#include <vector>

namespace hostq {

constexpr int NL = 32;
typedef unsigned __int128 u128;

struct Num {
  uint64_t v[NL];
};

inline int cmp(const Num& a, const Num& b) {
  for (int i = NL - 1; i >= 0; --i)
    if (a.v[i] != b.v[i]) return a.v[i] < b.v[i] ? -1 : 1;
  return 0;
}
inline bool is_zero(const Num& a) {
  uint64_t o = 0;
  for (int i = 0; i < NL; ++i) o |= a.v[i];
  return o == 0;
}
inline bool is_one(const Num& a) {
  if (a.v[0] != 1) return false;
  for (int i = 1; i < NL; ++i)
    if (a.v[i]) return false;
  return true;
}
inline uint64_t add(Num& r, const Num& a, const Num& b) {
  u128 c = 0;
  for (int i = 0; i < NL; ++i) { c += (u128)a.v[i] + b.v[i]; r.v[i] = (uint64_t)c; c >>= 64; }
  return (uint64_t)c;
}
inline uint64_t sub(Num& r, const Num& a, const Num& b) {
  uint64_t borrow = 0;
  for (int i = 0; i < NL; ++i) {
    const u128 d = (u128)a.v[i] - b.v[i] - borrow;
    r.v[i] = (uint64_t)d;
    borrow = (uint64_t)(d >> 64) & 1;
  }
  return borrow;
}
inline void shr1(Num& a, uint64_t top) {
  for (int i = 0; i < NL - 1; ++i) a.v[i] = (a.v[i] >> 1) | (a.v[i + 1] << 63);
  a.v[NL - 1] = (a.v[NL - 1] >> 1) | (top << 63);
}
inline void from_be(Num& r, const uint8_t* b) {   // 256 bytes big-endian
  for (int i = 0; i < NL; ++i) {
    uint64_t w = 0;
    for (int k = 0; k < 8; ++k) w = (w << 8) | b[255 - 8 * i - 7 + k];
    r.v[i] = w;
  }
}
inline void to_be(uint8_t* b, const Num& a) {
  for (int i = 0; i < NL; ++i)
    for (int k = 0; k < 8; ++k) b[255 - 8 * i - k] = (uint8_t)(a.v[i] >> (8 * k));
}

struct Field {
  Num q, r2, one;
  explicit Field(const uint8_t* q_be) {
    from_be(q, q_be);
    memset(&one, 0, sizeof(one));
    one.v[0] = 1;
    // R^2 mod q with R = 2^2048: start from 1 and double 4096 times
    Num x = one;
    for (int i = 0; i < 2 * 64 * NL; ++i) {
      Num d;
      const uint64_t carry = add(d, x, x);
      if (carry || cmp(d, q) >= 0) sub(d, d, q);
      x = d;
    }
    r2 = x;
  }
  void reduce_once(Num& a) const {          // a < 2^2048 -> a mod q (q > 2^2047)
    if (cmp(a, q) >= 0) sub(a, a, q);
  }
  // a * b * 2^-2048 mod q   (-q^-1 mod 2^64 == 1 because q == -1 mod 2^64)
  void mont_mul(Num& r, const Num& a, const Num& b) const {
    uint64_t t[NL + 2];
    memset(t, 0, sizeof(t));
    for (int i = 0; i < NL; ++i) {
      u128 c = 0;
      for (int j = 0; j < NL; ++j) { c += (u128)a.v[j] * b.v[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
      c += t[NL]; t[NL] = (uint64_t)c; t[NL + 1] = (uint64_t)(c >> 64);
      const uint64_t m = t[0];   // * n0inv (== 1)
      c = (u128)m * q.v[0] + t[0];
      c >>= 64;
      for (int j = 1; j < NL; ++j) { c += (u128)m * q.v[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
      c += t[NL]; t[NL - 1] = (uint64_t)c; t[NL] = t[NL + 1] + (uint64_t)(c >> 64);
    }
    Num x;
    memcpy(x.v, t, sizeof(x.v));
    if (t[NL] || cmp(x, q) >= 0) sub(x, x, q);
    r = x;
  }
  // plain a^-1 mod q for a in [1, q) by the binary extended Euclidean algorithm (q odd)
  void invert(Num& r, const Num& a) const {
    Num u = a, v = q, x1 = one, x2;
    memset(&x2, 0, sizeof(x2));
    auto halve = [&](Num& x) {
      if (x.v[0] & 1) { const uint64_t carry = add(x, x, q); shr1(x, carry); } else shr1(x, 0);
    };
    while (!is_one(u) && !is_one(v)) {
      while (!(u.v[0] & 1)) { shr1(u, 0); halve(x1); }
      while (!(v.v[0] & 1)) { shr1(v, 0); halve(x2); }
      if (cmp(u, v) >= 0) {
        sub(u, u, v);
        if (sub(x1, x1, x2)) add(x1, x1, q);
      } else {
        sub(v, v, u);
        if (sub(x2, x2, x1)) add(x2, x2, q);
      }
    }
    r = is_one(u) ? x1 : x2;
  }
};

// out[j] = in[j]^-1 mod q for t 256-byte big-endian values; false if some value is 0 mod q
inline bool batch_invert(const Field& f, const uint8_t* in_be, size_t t, uint8_t* out_be) {
  std::vector<Num> c(t), pre(t);
  for (size_t j = 0; j < t; ++j) {
    Num x;
    from_be(x, in_be + j * 256);
    f.reduce_once(x);                       // inputs are < 2^2048 < 2q
    if (is_zero(x)) return false;
    f.mont_mul(c[j], x, f.r2);              // Montgomery form
    if (j == 0) pre[0] = c[0]; else f.mont_mul(pre[j], pre[j - 1], c[j]);
  }
  Num total, inv;
  f.mont_mul(total, pre[t - 1], f.one);     // plain product
  f.invert(inv, total);
  f.mont_mul(inv, inv, f.r2);               // back to Montgomery form
  for (size_t j = t; j-- > 0;) {
    Num cj_inv;
    if (j == 0) cj_inv = inv; else f.mont_mul(cj_inv, inv, pre[j - 1]);
    f.mont_mul(inv, inv, c[j]);
    Num plain;
    f.mont_mul(plain, cj_inv, f.one);
    to_be(out_be + j * 256, plain);
  }
  return true;
}

}  // namespace hostq
