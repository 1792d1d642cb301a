// Host-side arithmetic in the reference's SCALAR rings: Z/(q-1) and Z/((q-1)/2) for MODP-2048 (32 x 64-bit limbs),
// Z/n (secp256k1) and Z/l (ristretto255) (4 limbs).  This is the scalar-field work of the protocol --
//   Group::scalar_mul / scalar_sub            src/group.rs:108-113, src/groups/modp.rs:180-192
//   DLEQ response r = w - alpha * c           src/dleq.rs:42-50 (participant.rs:255-264)
//   Polynomial::get_value(i) % order          src/polynomial.rs:50-58 + participant.rs:202
//   Lagrange coefficients of reconstruct      src/participant.rs:526-561, 1518-1557, 1955-2002
// -- O(n) or O(n t) word operations per box against the O(n * 3000) 2048-bit products of the group side, so it stays
// on the host (threaded over shares), behind the C ABI.  Group operations never run here.
#pragma once
#include <stdint.h>
#include <string.h>

#include <thread>
#include <vector>

namespace hsc {

typedef unsigned __int128 u128;

template <int N>
struct Num {
  uint64_t v[N];
};

template <int N>
inline int cmp(const uint64_t* a, const uint64_t* b) {
  for (int i = N - 1; i >= 0; --i)
    if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
  return 0;
}
template <int N>
inline uint64_t add_n(uint64_t* r, const uint64_t* a, const uint64_t* b) {
  u128 c = 0;
  for (int i = 0; i < N; ++i) { c += (u128)a[i] + b[i]; r[i] = (uint64_t)c; c >>= 64; }
  return (uint64_t)c;
}
template <int N>
inline uint64_t sub_n(uint64_t* r, const uint64_t* a, const uint64_t* b) {
  uint64_t borrow = 0;
  for (int i = 0; i < N; ++i) {
    const u128 d = (u128)a[i] - b[i] - borrow;
    r[i] = (uint64_t)d;
    borrow = (uint64_t)(d >> 64) & 1;
  }
  return borrow;
}

// A modulus of exactly N limbs (top limb non-zero) with schoolbook division (Knuth algorithm D, 64-bit digits).
template <int N>
struct Modulus {
  uint64_t m[N];        // the modulus
  uint64_t mn[N];       // normalised: m << shift, top bit set
  int shift;

  void set(const uint64_t* mod) {
    memcpy(m, mod, sizeof(m));
    shift = __builtin_clzll(m[N - 1]);
    for (int i = N - 1; i >= 0; --i) mn[i] = shift ? (m[i] << shift) | (i ? m[i - 1] >> (64 - shift) : 0) : m[i];
  }
  // r = u mod m for u of UL limbs (UL >= N, UL <= 2N + 1)
  void reduce(uint64_t* r, const uint64_t* u_in, int UL) const {
    uint64_t u[2 * N + 3];
    u[UL] = 0;
    for (int i = UL - 1; i >= 0; --i) u[i] = u_in[i];
    if (shift) {
      for (int i = UL; i > 0; --i) u[i] = (u[i] << shift) | (u[i - 1] >> (64 - shift));
      u[0] <<= shift;
    }
    const uint64_t v1 = mn[N - 1], v2 = N > 1 ? mn[N - 2] : 0;
    for (int j = UL - N; j >= 0; --j) {
      // estimate the quotient digit from the top two limbs
      const u128 num = ((u128)u[j + N] << 64) | u[j + N - 1];
      u128 qhat = u[j + N] >= v1 ? (u128)0xFFFFFFFFFFFFFFFFULL : num / v1;
      u128 rhat = num - qhat * v1;
      while (rhat <= 0xFFFFFFFFFFFFFFFFULL && N > 1 && qhat * v2 > ((rhat << 64) | u[j + N - 2])) {
        --qhat;
        rhat += v1;
      }
      // multiply and subtract
      u128 borrow = 0, carry = 0;
      for (int i = 0; i < N; ++i) {
        carry += (u128)(uint64_t)qhat * mn[i];
        const u128 d = (u128)u[j + i] - (uint64_t)carry - (uint64_t)borrow;
        u[j + i] = (uint64_t)d;
        borrow = (d >> 64) & 1;
        carry >>= 64;
      }
      const u128 d = (u128)u[j + N] - (uint64_t)carry - (uint64_t)borrow;
      u[j + N] = (uint64_t)d;
      if ((d >> 64) & 1) {          // qhat was one too large: add back
        u128 c = 0;
        for (int i = 0; i < N; ++i) { c += (u128)u[j + i] + mn[i]; u[j + i] = (uint64_t)c; c >>= 64; }
        u[j + N] += (uint64_t)c;
      }
    }
    // remainder = u[0..N) >> shift
    for (int i = 0; i < N; ++i) r[i] = shift ? (u[i] >> shift) | (u[i + 1] << (64 - shift)) : u[i];
  }
  void reduce1(uint64_t* a) const {                 // N limbs in place
    uint64_t t[N];
    reduce(t, a, N);
    memcpy(a, t, sizeof(t));
  }
  void mulmod(uint64_t* r, const uint64_t* a, const uint64_t* b) const {
    uint64_t t[2 * N];
    memset(t, 0, sizeof(t));
    for (int i = 0; i < N; ++i) {
      u128 c = 0;
      for (int j = 0; j < N; ++j) { c += (u128)a[j] * b[i] + t[i + j]; t[i + j] = (uint64_t)c; c >>= 64; }
      t[i + N] = (uint64_t)c;
    }
    reduce(r, t, 2 * N);
  }
  // r = (a * x + b) mod m for a 64-bit x; a, b < m
  void muladd_small(uint64_t* r, const uint64_t* a, uint64_t x, const uint64_t* b) const {
    uint64_t t[N + 1];
    u128 c = 0;
    for (int j = 0; j < N; ++j) { c += (u128)a[j] * x + b[j]; t[j] = (uint64_t)c; c >>= 64; }
    t[N] = (uint64_t)c;
    reduce(r, t, N + 1);
  }
  // (a - b) mod m for a, b < m
  // r = (a + b) mod m for a, b < m: one add, one subtract, a branch-free choice
  void addmod(uint64_t* r, const uint64_t* a, const uint64_t* b) const {
    uint64_t s[N], d[N];
    const uint64_t carry = add_n<N>(s, a, b);
    const uint64_t borrow = sub_n<N>(d, s, m);
    const uint64_t take_d = (uint64_t)0 - (uint64_t)(carry | (borrow ^ 1));     // a + b >= m
    for (int i = 0; i < N; ++i) r[i] = (d[i] & take_d) | (s[i] & ~take_d);
  }
  void submod(uint64_t* r, const uint64_t* a, const uint64_t* b) const {
    if (sub_n<N>(r, a, b)) add_n<N>(r, r, m);
  }
  // a^-1 mod m for ODD m and gcd(a, m) = 1 (binary extended Euclid); false when a is 0 or not invertible
  bool invert(uint64_t* r, const uint64_t* a_in) const {
    uint64_t u[N], v[N], x1[N], x2[N];
    memcpy(u, a_in, sizeof(u));
    memcpy(v, m, sizeof(v));
    memset(x1, 0, sizeof(x1));
    memset(x2, 0, sizeof(x2));
    x1[0] = 1;
    auto is_zero = [](const uint64_t* x) { uint64_t o = 0; for (int i = 0; i < N; ++i) o |= x[i]; return o == 0; };
    auto is_one = [](const uint64_t* x) { if (x[0] != 1) return false; for (int i = 1; i < N; ++i) if (x[i]) return false; return true; };
    auto shr1 = [](uint64_t* x, uint64_t top) {
      for (int i = 0; i < N - 1; ++i) x[i] = (x[i] >> 1) | (x[i + 1] << 63);
      x[N - 1] = (x[N - 1] >> 1) | (top << 63);
    };
    auto halve = [&](uint64_t* x) {
      if (x[0] & 1) { const uint64_t carry = add_n<N>(x, x, m); shr1(x, carry); } else shr1(x, 0);
    };
    if (is_zero(u)) return false;
    while (!is_one(u) && !is_one(v)) {
      if (is_zero(u) || is_zero(v)) return false;             // gcd > 1
      while (!(u[0] & 1)) { shr1(u, 0); halve(x1); }
      while (!(v[0] & 1)) { shr1(v, 0); halve(x2); }
      if (cmp<N>(u, v) >= 0) {
        sub_n<N>(u, u, v);
        if (sub_n<N>(x1, x1, x2)) add_n<N>(x1, x1, m);
      } else {
        sub_n<N>(v, v, u);
        if (sub_n<N>(x2, x2, x1)) add_n<N>(x2, x2, m);
      }
    }
    memcpy(r, is_one(u) ? x1 : x2, sizeof(x1));
    return true;
  }
};

// byte conversions: `bytes` is N*8 bytes wide, big- or little-endian
template <int N>
inline void from_bytes(uint64_t* r, const uint8_t* b, bool big_endian) {
  for (int i = 0; i < N; ++i) {
    uint64_t w = 0;
    for (int k = 0; k < 8; ++k) w |= (uint64_t)b[big_endian ? N * 8 - 1 - (8 * i + k) : 8 * i + k] << (8 * k);
    r[i] = w;
  }
}
template <int N>
inline void to_bytes(uint8_t* b, const uint64_t* a, bool big_endian) {
  for (int i = 0; i < N; ++i)
    for (int k = 0; k < 8; ++k) b[big_endian ? N * 8 - 1 - (8 * i + k) : 8 * i + k] = (uint8_t)(a[i] >> (8 * k));
}

// run fn(k) for k in [0, nt): nt - 1 helper threads and the caller.  Every caller of this sits under an extern "C" entry
// point, so a thread that cannot be created (EAGAIN, a cgroup pids limit: std::system_error) must not unwind through the
// C ABI or past a joinable std::thread (std::terminate) -- the indices that found no thread run on the caller instead.
template <class Fn>
inline void parallel_indices(unsigned nt, Fn fn) {
  if (nt < 1) nt = 1;
  std::vector<std::thread> pool;
  unsigned started = 0;
  try {
    pool.reserve(nt);
    for (; started + 1 < nt; ++started) pool.emplace_back(fn, started);
  } catch (...) {
  }
  for (unsigned k = started; k < nt; ++k) fn(k);
  for (auto& t : pool) t.join();
}

// run fn(lo, hi) over [0, n) on up to `threads` host threads
template <class Fn>
inline void parallel_for(size_t n, int threads, Fn fn) {
  if (threads < 1) threads = 1;
  const size_t per = (n + (size_t)threads - 1) / (size_t)threads;
  if (threads == 1 || n < 256) { fn((size_t)0, n); return; }
  const unsigned parts = (unsigned)((n + per - 1) / per);
  parallel_indices(parts, [&](unsigned k) { const size_t lo = (size_t)k * per; fn(lo, lo + per < n ? lo + per : n); });
}

}  // namespace hsc
