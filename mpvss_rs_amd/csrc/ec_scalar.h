// 256-bit arithmetic modulo the group order (secp256k1 n, ristretto255 l): Montgomery form, 8 x 32-bit limbs.
// Only used for the few scalar products a share needs (powers of the position i); plain C++ (host + device).
#pragma once
#include "ec_consts.h"

namespace ec {

struct Sc {
  u32 v[8];
};

template <class O>
struct ScalarField {
  // r = a * b * 2^-256 mod n
  static EC_HD void mont_mul(Sc& r, const Sc& a, const Sc& b) {
    u32 t[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      u64 c = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        c += (u64)a.v[j] * b.v[i] + t[j];
        t[j] = (u32)c;
        c >>= 32;
      }
      c += t[8];
      t[8] = (u32)c;
      t[9] = (u32)(c >> 32);
      const u32 m = t[0] * O::N0INV;
      c = (u64)m * O::n(0) + t[0];
      c >>= 32;
#pragma unroll
      for (int j = 1; j < 8; ++j) {
        c += (u64)m * O::n(j) + t[j];
        t[j - 1] = (u32)c;
        c >>= 32;
      }
      c += t[8];
      t[7] = (u32)c;
      t[8] = t[9] + (u32)(c >> 32);
    }
    // t < 2n: one conditional subtraction
    bool ge = t[8] != 0;
    if (!ge) {
      ge = true;
      bool decided = false;
#pragma unroll
      for (int i = 7; i >= 0; --i) {
        if (!decided && t[i] != O::n(i)) {
          ge = t[i] > O::n(i);
          decided = true;
        }
      }
    }
    const u32 mask = ge ? 0xffffffffu : 0u;
    u64 borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const u64 d = (u64)t[i] - (O::n(i) & mask) - borrow;
      r.v[i] = (u32)d;
      borrow = (d >> 63) & 1;
    }
  }
  // r = base^e mod n as a plain (non-Montgomery) integer; base is a 64-bit value, e a small exponent
  static EC_HD void pow_u64(Sc& r, uint64_t base, u32 e) {
    Sc b, acc, one, r2;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      b.v[i] = 0;
      one.v[i] = 0;
      r2.v[i] = O::r2(i);
    }
    b.v[0] = (u32)base;
    b.v[1] = (u32)(base >> 32);
    one.v[0] = 1;
    mont_mul(b, b, r2);        // base * R
    mont_mul(acc, one, r2);    // R  (Montgomery one)
    int nb = 0;
    while (nb < 32 && (e >> nb)) ++nb;
    for (int i = nb - 1; i >= 0; --i) {
      mont_mul(acc, acc, acc);
      if ((e >> i) & 1) mont_mul(acc, acc, b);
    }
    mont_mul(r, acc, one);     // leave the Montgomery domain
  }
};

}  // namespace ec
