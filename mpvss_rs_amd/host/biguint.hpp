// Minimal arbitrary-precision unsigned integer for the HOST side of the engine (the stand-in for the
// num-bigint values the reference's Rust host code holds: polynomial coefficients, scalars mod q-1,
// Lagrange factors, the secret/U encoding).  Nothing here is on the hot path: group exponentiations go
// to the GPU through the C ABI.  Header-only, no dependencies.
#pragma once
#include <stdint.h>

#include <algorithm>
#include <stdexcept>
#include <string>
#include <vector>

namespace mpvss_host {

class BigUint {
 public:
  std::vector<uint32_t> d;   // little-endian 32-bit digits, no leading zero digits

  BigUint() {}
  BigUint(uint64_t v) {
    while (v) { d.push_back((uint32_t)v); v >>= 32; }
  }
  static BigUint from_bytes_be(const uint8_t* b, size_t n) {
    BigUint r;
    r.d.assign((n + 3) / 4, 0);
    for (size_t i = 0; i < n; ++i) r.d[i / 4] |= (uint32_t)b[n - 1 - i] << (8 * (i % 4));
    r.trim();
    return r;
  }
  static BigUint from_bytes_be(const std::vector<uint8_t>& b) { return from_bytes_be(b.data(), b.size()); }
  static BigUint from_hex(const std::string& h) {
    BigUint r;
    for (char c : h) {
      int v = (c >= '0' && c <= '9') ? c - '0' : (c >= 'a' && c <= 'f') ? c - 'a' + 10 : (c >= 'A' && c <= 'F') ? c - 'A' + 10 : -1;
      if (v < 0) continue;
      r = r.shl(4) + BigUint((uint64_t)v);
    }
    return r;
  }
  // BigUint::to_bytes_be of num-bigint: minimal length, zero -> [0]
  std::vector<uint8_t> to_bytes_be() const {
    if (d.empty()) return {0};
    std::vector<uint8_t> out;
    const size_t nbytes = (bits() + 7) / 8;
    out.resize(nbytes);
    for (size_t i = 0; i < nbytes; ++i) out[nbytes - 1 - i] = (uint8_t)(d[i / 4] >> (8 * (i % 4)));
    return out;
  }
  // fixed width big-endian (the C ABI's 256-byte encoding); throws if it does not fit
  std::vector<uint8_t> to_fixed_be(size_t width) const {
    if (bits() > 8 * width) throw std::range_error("BigUint does not fit the fixed-width encoding");
    std::vector<uint8_t> out(width, 0);
    for (size_t i = 0; i < width && i / 4 < d.size(); ++i) out[width - 1 - i] = (uint8_t)(d[i / 4] >> (8 * (i % 4)));
    return out;
  }
  size_t bits() const {
    if (d.empty()) return 0;
    return 32 * (d.size() - 1) + (32 - __builtin_clz(d.back()));
  }
  bool is_zero() const { return d.empty(); }
  bool is_odd() const { return !d.empty() && (d[0] & 1); }
  bool bit(size_t i) const { return i / 32 < d.size() && ((d[i / 32] >> (i % 32)) & 1); }
  uint64_t low64() const { return (d.size() > 0 ? d[0] : 0) | ((uint64_t)(d.size() > 1 ? d[1] : 0) << 32); }

  static int cmp(const BigUint& a, const BigUint& b) {
    if (a.d.size() != b.d.size()) return a.d.size() < b.d.size() ? -1 : 1;
    for (size_t i = a.d.size(); i-- > 0;)
      if (a.d[i] != b.d[i]) return a.d[i] < b.d[i] ? -1 : 1;
    return 0;
  }
  bool operator==(const BigUint& o) const { return cmp(*this, o) == 0; }
  bool operator!=(const BigUint& o) const { return cmp(*this, o) != 0; }
  bool operator<(const BigUint& o) const { return cmp(*this, o) < 0; }
  bool operator<=(const BigUint& o) const { return cmp(*this, o) <= 0; }
  bool operator>(const BigUint& o) const { return cmp(*this, o) > 0; }
  bool operator>=(const BigUint& o) const { return cmp(*this, o) >= 0; }

  BigUint operator+(const BigUint& o) const {
    BigUint r;
    const size_t n = std::max(d.size(), o.d.size());
    r.d.resize(n + 1);
    uint64_t c = 0;
    for (size_t i = 0; i < n; ++i) {
      c += (uint64_t)(i < d.size() ? d[i] : 0) + (i < o.d.size() ? o.d[i] : 0);
      r.d[i] = (uint32_t)c;
      c >>= 32;
    }
    r.d[n] = (uint32_t)c;
    r.trim();
    return r;
  }
  // requires *this >= o
  BigUint operator-(const BigUint& o) const {
    if (*this < o) throw std::range_error("BigUint subtraction underflow");
    BigUint r;
    r.d.resize(d.size());
    int64_t b = 0;
    for (size_t i = 0; i < d.size(); ++i) {
      int64_t t = (int64_t)d[i] - (i < o.d.size() ? o.d[i] : 0) + b;
      r.d[i] = (uint32_t)t;
      b = t >> 32;
    }
    r.trim();
    return r;
  }
  BigUint operator*(const BigUint& o) const {
    BigUint r;
    if (d.empty() || o.d.empty()) return r;
    r.d.assign(d.size() + o.d.size(), 0);
    for (size_t i = 0; i < d.size(); ++i) {
      uint64_t c = 0;
      for (size_t j = 0; j < o.d.size(); ++j) {
        const uint64_t t = (uint64_t)d[i] * o.d[j] + r.d[i + j] + c;
        r.d[i + j] = (uint32_t)t;
        c = t >> 32;
      }
      r.d[i + o.d.size()] = (uint32_t)c;
    }
    r.trim();
    return r;
  }
  BigUint shl(size_t n) const {
    if (d.empty()) return *this;
    BigUint r;
    const size_t w = n / 32, s = n % 32;
    r.d.assign(d.size() + w + 1, 0);
    for (size_t i = 0; i < d.size(); ++i) {
      r.d[i + w] |= d[i] << s;
      if (s) r.d[i + w + 1] |= d[i] >> (32 - s);
    }
    r.trim();
    return r;
  }
  BigUint shr(size_t n) const {
    BigUint r;
    const size_t w = n / 32, s = n % 32;
    if (w >= d.size()) return r;
    r.d.assign(d.size() - w, 0);
    for (size_t i = w; i < d.size(); ++i) {
      r.d[i - w] = d[i] >> s;
      if (s && i + 1 < d.size()) r.d[i - w] |= d[i + 1] << (32 - s);
    }
    r.trim();
    return r;
  }
  BigUint operator^(const BigUint& o) const {
    BigUint r;
    r.d.assign(std::max(d.size(), o.d.size()), 0);
    for (size_t i = 0; i < r.d.size(); ++i) r.d[i] = (i < d.size() ? d[i] : 0) ^ (i < o.d.size() ? o.d[i] : 0);
    r.trim();
    return r;
  }
  // schoolbook long division, bit by bit on the quotient digits (host-side sizes are small)
  static void divmod(const BigUint& a, const BigUint& m, BigUint& q, BigUint& r) {
    if (m.is_zero()) throw std::domain_error("division by zero");
    q = BigUint();
    r = BigUint();
    if (a < m) { r = a; return; }
    q.d.assign(a.d.size(), 0);
    for (size_t i = a.bits(); i-- > 0;) {
      r = r.shl(1);
      if (a.bit(i)) { if (r.d.empty()) r.d.push_back(1); else r.d[0] |= 1; }
      if (r >= m) { r = r - m; q.d[i / 32] |= 1u << (i % 32); }
    }
    q.trim();
  }
  BigUint operator%(const BigUint& m) const { BigUint q, r; divmod(*this, m, q, r); return r; }
  BigUint operator/(const BigUint& m) const { BigUint q, r; divmod(*this, m, q, r); return q; }

  static BigUint gcd(BigUint a, BigUint b) {
    while (!b.is_zero()) { BigUint t = a % b; a = b; b = t; }
    return a;
  }
  // a^-1 mod m, or false when gcd(a, m) != 1      (util.rs:33-41)
  static bool mod_inverse(const BigUint& a, const BigUint& m, BigUint& out) {
    // extended Euclid with the Bezout coefficient tracked modulo m (kept non-negative)
    BigUint r0 = m, r1 = a % m, t0 = BigUint(), t1 = BigUint(1);
    while (!r1.is_zero()) {
      BigUint q, r2;
      divmod(r0, r1, q, r2);
      BigUint qt = (q * t1) % m;
      BigUint t2 = (t0 >= qt) ? t0 - qt : (t0 + m) - qt;
      r0 = r1; r1 = r2; t0 = t1; t1 = t2;
    }
    if (r0 != BigUint(1)) return false;
    out = t0 % m;
    return true;
  }
  // host modpow (square and multiply); only used for tiny jobs and self-checks
  static BigUint modpow(const BigUint& b, const BigUint& e, const BigUint& m) {
    BigUint r(1), x = b % m;
    for (size_t i = 0; i < e.bits(); ++i) {
      if (e.bit(i)) r = (r * x) % m;
      x = (x * x) % m;
    }
    return r % m;
  }
  std::string to_hex() const {
    if (d.empty()) return "0";
    static const char* hx = "0123456789abcdef";
    std::string s;
    for (size_t i = bits(); i > 0;) {
      const size_t nib = (i - 1) / 4;
      s.push_back(hx[(d[nib / 8] >> (4 * (nib % 8))) & 15]);
      i = nib * 4;
    }
    return s;
  }

 private:
  void trim() { while (!d.empty() && d.back() == 0) d.pop_back(); }
};

}  // namespace mpvss_host
