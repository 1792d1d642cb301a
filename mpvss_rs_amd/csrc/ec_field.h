// 256-bit prime-field arithmetic, one field element per lane, radix 2^26 x 10 limbs with 64-bit
// column accumulators (v_mad_u64_u32 on gfx950).  Shared by the secp256k1 and ristretto255 kernels.
// The header is plain C++ so that the same code is compiled for the host by the CPU unit tests
// (tests/ec_host_shim.cpp) and for gfx950 by hipcc.
//
// Representation: value = sum v[i] * 2^(26 i); capacity 260 bits, values are only weakly reduced
// (any representative below ~2^261).  After carry: v[0] < 2^26 + 2^21, v[1] < 2^27, v[2..9] < 2^26; after mul / sqr:
// every limb < 2^27 + 2^12 (reduce_columns).  mul accepts limbs up to 2^30 (10 products of 2^60 fit a 64-bit column).
//
// Folding: 2^260 = R1 * 2^26 + R0 (mod p):  secp256k1 p = 2^256 - 2^32 - 977 -> 2^260 = 16 (2^32 + 977)
// = 2^10 * 2^26 + 15632;   curve25519 p = 2^255 - 19 -> 2^260 = 32 * 19 = 608.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define EC_HD __host__ __device__ __forceinline__
#else
#define EC_HD inline
#endif

namespace ec {

typedef uint32_t u32;
typedef uint64_t u64;
constexpr u32 M26 = (1u << 26) - 1;

struct Fe {
  u32 v[10];
};

struct PrimeSecp {
  static constexpr u32 R0 = 15632, R1 = 1024;
  static constexpr int TOPBITS = 22;          // bits of limb 9 below 2^256
  static constexpr u32 TOP_R0 = 977, TOP_R1 = 64;   // 2^256 = 2^32 + 977 = 64 * 2^26 + 977
};
struct PrimeEd {
  static constexpr u32 R0 = 608, R1 = 0;
  static constexpr int TOPBITS = 21;          // bits of limb 9 below 2^255
  static constexpr u32 TOP_R0 = 19, TOP_R1 = 0;
};

// constants that depend on the prime (canonical p, the "2^29 - m" subtraction pad): ec_consts.h
template <class P> struct PrimeConsts;

template <class P>
struct F {
  static EC_HD void zero(Fe& r) {
#pragma unroll
    for (int i = 0; i < 10; ++i) r.v[i] = 0;
  }
  static EC_HD void one(Fe& r) {
    zero(r);
    r.v[0] = 1;
  }
  static EC_HD void copy(Fe& r, const Fe& a) {
#pragma unroll
    for (int i = 0; i < 10; ++i) r.v[i] = a.v[i];
  }
  // one weak carry pass; limbs in: < 2^31.  out: v[0] < 2^26 + 2^21, v[1] < 2^27, others < 2^26
  static EC_HD void carry(Fe& r) {
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const u32 x = r.v[i] + c;
      r.v[i] = x & M26;
      c = x >> 26;
    }
    const u32 x = r.v[9] + c;
    r.v[9] = x & M26;
    const u32 t = x >> 26;            // < 2^6
    r.v[0] += t * P::R0;
    r.v[1] += t * P::R1;
  }
  static EC_HD void add(Fe& r, const Fe& a, const Fe& b) {     // no carry: limbs add
#pragma unroll
    for (int i = 0; i < 10; ++i) r.v[i] = a.v[i] + b.v[i];
  }
  static EC_HD void addc(Fe& r, const Fe& a, const Fe& b) {    // add + carry
    add(r, a, b);
    carry(r);
  }
  // r = a - b  (b limbs < 2^29 - 2^26), carried
  static EC_HD void sub(Fe& r, const Fe& a, const Fe& b) {
#pragma unroll
    for (int i = 0; i < 10; ++i) r.v[i] = a.v[i] + PrimeConsts<P>::subpad(i) - b.v[i];
    carry(r);
  }
  static EC_HD void neg(Fe& r, const Fe& a) {
#pragma unroll
    for (int i = 0; i < 10; ++i) r.v[i] = PrimeConsts<P>::subpad(i) - a.v[i];
    carry(r);
  }
  // r = a * k for a small constant k (< 2^6), carried
  static EC_HD void mul_small(Fe& r, const Fe& a, u32 k) {
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const u64 x = (u64)a.v[i] * k + c;
      r.v[i] = (u32)x & M26;
      c = x >> 26;
    }
    const u64 x = (u64)a.v[9] * k + c;
    r.v[9] = (u32)x & M26;
    const u32 t = (u32)(x >> 26);     // < 2^12
    r.v[0] += t * P::R0;              // < 2^26 + 2^26: fine as an input limb
    r.v[1] += t * P::R1;
  }

  // the three 26-bit-aligned pieces of a 64-bit column: bits 0..25, 26..51, 52..63 (an and, a funnel shift + and, a shift)
  static EC_HD void pieces(u32& p0, u32& p1, u32& p2, u64 c) {
    const u32 lo = (u32)c, hi = (u32)(c >> 32);
    p0 = lo & M26;
    p1 = ((lo >> 26) | (hi << 6)) & M26;
    p2 = hi >> 20;
  }
  // 19 product columns (each < 2^64) -> 10 limbs.  No carry chain over 64-bit columns: a high column is cut into its
  // three pieces, which are folded (2^260 = R1 2^26 + R0) straight into the low columns with multiply-adds -- top
  // column first, so that what lands in columns 10 and 11 is folded again when their turn comes; the low columns are
  // then cut the same way and limb k = piece0[k] + piece1[k-1] + piece2[k-2].
  // Out: v[0] < 2^26, v[1] < 2^27, v[2..9] < 2^27 + 2^12.
  static EC_HD void reduce_columns(Fe& r, u64 (&c)[20]) {
#pragma unroll
    for (int i = 8; i >= 0; --i) {
      u32 p0, p1, p2;
      pieces(p0, p1, p2, c[10 + i]);
      c[i] += (u64)p0 * P::R0;
      c[i + 1] += (u64)p1 * P::R0;
      c[i + 2] += (u64)p2 * P::R0;
      if (P::R1 != 0) {
        c[i + 1] += (u64)p0 * P::R1;
        c[i + 2] += (u64)p1 * P::R1;
        c[i + 3] += (u64)p2 * P::R1;
      }
    }
    // columns 8 and 9 first: their upper pieces wrap around into columns 0..2
    u32 q0[10], q1[10], q2[10];
    pieces(q0[9], q1[9], q2[9], c[9]);
    pieces(q0[8], q1[8], q2[8], c[8]);
    {
      const u32 w10 = q1[9] + q2[8];        // weight 2^260, < 2^26 + 2^12
      const u32 w11 = q2[9];                // weight 2^286, < 2^12
      c[0] += (u64)w10 * P::R0;
      c[1] += (u64)w11 * P::R0;
      if (P::R1 != 0) {
        c[1] += (u64)w10 * P::R1;
        c[2] += (u64)w11 * P::R1;
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) pieces(q0[k], q1[k], q2[k], c[k]);
    r.v[0] = q0[0];
    r.v[1] = q0[1] + q1[0];
#pragma unroll
    for (int k = 2; k < 10; ++k) r.v[k] = q0[k] + q1[k - 1] + q2[k - 2];
  }

  static EC_HD void mul(Fe& r, const Fe& a, const Fe& b) {
    u64 c[20];
#pragma unroll
    for (int k = 0; k < 20; ++k) c[k] = 0;
#pragma unroll
    for (int i = 0; i < 10; ++i)
#pragma unroll
      for (int j = 0; j < 10; ++j) c[i + j] += (u64)a.v[i] * b.v[j];
    reduce_columns(r, c);
  }

  // r = a * b + c * d with one reduction (limbs up to 2^28.5: 20 products of 2^57 fit a 64-bit column with room for the folds)
  static EC_HD void mul2(Fe& r, const Fe& a, const Fe& b, const Fe& c2, const Fe& d) {
    u64 c[20];
#pragma unroll
    for (int k = 0; k < 20; ++k) c[k] = 0;
#pragma unroll
    for (int i = 0; i < 10; ++i)
#pragma unroll
      for (int j = 0; j < 10; ++j) c[i + j] += (u64)a.v[i] * b.v[j];
#pragma unroll
    for (int i = 0; i < 10; ++i)
#pragma unroll
      for (int j = 0; j < 10; ++j) c[i + j] += (u64)c2.v[i] * d.v[j];
    reduce_columns(r, c);
  }

  static EC_HD void sqr(Fe& r, const Fe& a) {
    u64 c[20];
#pragma unroll
    for (int k = 0; k < 20; ++k) c[k] = 0;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      c[2 * i] += (u64)a.v[i] * a.v[i];
#pragma unroll
      for (int j = i + 1; j < 10; ++j) c[i + j] += (u64)(2 * a.v[i]) * a.v[j];   // inputs < 2^30 -> 2a < 2^31
    }
    reduce_columns(r, c);
  }

  static EC_HD void sqrn(Fe& r, const Fe& a, int n) {
    copy(r, a);
    for (int i = 0; i < n; ++i) sqr(r, r);
  }

  // canonical representative in [0, p): limbs < 2^26, top limb < 2^TOPBITS
  static EC_HD void canon(Fe& r) {
    carry(r);
    // fold the bits above 2^256 / 2^255
    {
      const u32 hi = r.v[9] >> P::TOPBITS;
      r.v[9] &= (1u << P::TOPBITS) - 1;
      r.v[0] += hi * P::TOP_R0;
      r.v[1] += hi * P::TOP_R1;
    }
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      const u32 x = r.v[i] + c;
      r.v[i] = x & M26;
      c = x >> 26;
    }
    // now value < 2^256 (2^255) + small: fold once more, then at most two conditional subtractions
    {
      const u32 hi = r.v[9] >> P::TOPBITS;
      r.v[9] &= (1u << P::TOPBITS) - 1;
      r.v[0] += hi * P::TOP_R0;
      r.v[1] += hi * P::TOP_R1;
      c = 0;
#pragma unroll
      for (int i = 0; i < 10; ++i) {
        const u32 x = r.v[i] + c;
        r.v[i] = x & M26;
        c = x >> 26;
      }
    }
    for (int rep = 0; rep < 2; ++rep) {
      // ge = (r >= p)
      int ge = 1;
      bool decided = false;
#pragma unroll
      for (int i = 9; i >= 0; --i) {
        const u32 pi = PrimeConsts<P>::p(i);
        if (!decided && r.v[i] != pi) {
          ge = r.v[i] > pi;
          decided = true;
        }
      }
      const u32 mask = ge ? 0xffffffffu : 0u;
      u32 borrow = 0;
#pragma unroll
      for (int i = 0; i < 10; ++i) {
        const u32 d = r.v[i] - (PrimeConsts<P>::p(i) & mask) - borrow;
        borrow = (d >> 31) & 1;
        r.v[i] = d & M26;
      }
    }
  }

  static EC_HD bool is_zero(const Fe& a) {
    Fe t;
    copy(t, a);
    canon(t);
    u32 acc = 0;
#pragma unroll
    for (int i = 0; i < 10; ++i) acc |= t.v[i];
    return acc == 0;
  }
  static EC_HD bool equal(const Fe& a, const Fe& b) {
    Fe t;
    sub(t, a, b);
    return is_zero(t);
  }
  static EC_HD bool is_odd(const Fe& a) {   // "negative" in the ristretto255 sense / SEC1 y parity
    Fe t;
    copy(t, a);
    canon(t);
    return t.v[0] & 1;
  }
  // r = cond ? a : r
  static EC_HD void cmov(Fe& r, const Fe& a, bool cond) {
    const u32 m = cond ? 0xffffffffu : 0u;
#pragma unroll
    for (int i = 0; i < 10; ++i) r.v[i] = (r.v[i] & ~m) | (a.v[i] & m);
  }

  // 32-byte big-endian / little-endian <-> limbs.  from_* return false when the value is >= p.
  static EC_HD void from_le32_raw(Fe& r, const uint8_t* b) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      const int bit = 26 * i;
      u64 w = 0;
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const int idx = (bit >> 3) + k;
        if (idx < 32) w |= (u64)b[idx] << (8 * k);
      }
      r.v[i] = (u32)(w >> (bit & 7)) & M26;
    }
  }
  static EC_HD void to_le32(uint8_t* b, const Fe& a_canonical) {
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      const int bit = 8 * k;
      const int i = bit / 26, s = bit % 26;
      u32 w = a_canonical.v[i] >> s;
      if (s > 18 && i + 1 < 10) w |= a_canonical.v[i + 1] << (26 - s);
      b[k] = (uint8_t)w;
    }
  }
  static EC_HD bool is_canonical(const Fe& a) {   // limbs already < 2^26 by construction of from_*_raw
    bool lt = false, decided = false;
#pragma unroll
    for (int i = 9; i >= 0; --i) {
      const u32 pi = PrimeConsts<P>::p(i);
      if (!decided && a.v[i] != pi) {
        lt = a.v[i] < pi;
        decided = true;
      }
    }
    return lt;
  }
};

}  // namespace ec
