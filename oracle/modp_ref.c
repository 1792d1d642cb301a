/* modp_ref.c -- plain-C CPU restatement of the reference's MODP-2048 verification path.
 *
 * TEST INFRASTRUCTURE ONLY: the checker for tests/ and the `cpu_baseline` leg of bench.py.
 * Nothing in the product path (mpvss_rs_amd/, libmpvss_hip.so) links or calls this file.
 *
 * It executes the reference's operation sequence as written:
 *   verify_distribution_shares            src/participant.rs:399-455
 *     X_i loop: exp, mul, scalar_mul       src/participant.rs:423-434
 *     Verifier::commitments (4 exp, 2 mul) src/dleq.rs:66-84
 *     framed transcript                    src/dleq.rs:58-61,87-99
 *     hash_to_scalar / check               src/groups/modp.rs:142-148, src/dleq.rs:119-126
 *   ModpGroup::exp  = BigInt::modpow       src/groups/modp.rs:122-128
 *   ModpGroup::mul  = (a*b) % q            src/groups/modp.rs:130-132
 *   scalar_mul      = (a*b) % (q-1)        src/groups/modp.rs:180-182
 *
 * The big-integer arithmetic itself lives in a crate the reference does not vendor
 * (num-bigint = "0.2", Cargo.toml:15; Cargo.lock is git-ignored so no exact pin exists).  Its
 * published algorithm for an odd modulus is restated here: 32-bit digits, Montgomery domain,
 * right-to-left binary exponentiation (one Montgomery squaring per exponent bit, one Montgomery
 * product per set bit, each a full schoolbook product followed by a word-serial REDC), and
 * `%` as schoolbook long division (Knuth algorithm D).  Results are canonical residues, so any
 * correct implementation yields the same bytes; the structure is kept so that the measured CPU
 * time is representative of the reference's.
 *
 * Parity pinning: see oracle/mpvss_oracle.py header; this file is additionally cross-checked
 * against that Python restatement in tests/test_oracle_c.py.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define NW 64            /* 32-bit digits in 2048 bits */
typedef uint32_t u32;
typedef uint64_t u64;

static const char Q_HEX[] =
    "ffffffffffffffffc90fdaa22168c234c4c6628b80dc1cd129024e088a67cc74"
    "020bbea63b139b22514a08798e3404ddef9519b3cd3a431b302b0a6df25f1437"
    "4fe1356d6d51c245e485b576625e7ec6f44c42e9a637ed6b0bff5cb6f406b7ed"
    "ee386bfb5a899fa5ae9f24117c4b1fe649286651ece45b3dc2007cb8a163bf05"
    "98da48361c55d39a69163fa8fd24cf5f83655d23dca3ad961c62f356208552bb"
    "9ed529077096966d670c354e4abc9804f1746c08ca18217c32905e462e36ce3b"
    "e39e772c180e86039b2783a2ec07a28fb5c55df06f4c52c9de2bcbf695581718"
    "3995497cea956ae515d2261898fa051015728e5a8aacaa68ffffffffffffffff";   /* modp.rs:47-58 */

static u32 Q[NW], QM1[NW], RMODQ[NW];
static u32 N0INV;
static int g_init = 0;

/* ---------------------------------------------------------------- SHA-256 (FIPS 180-4) */
typedef struct { u32 h[8]; uint8_t buf[64]; u64 total; size_t fill; } sha256_t;
static const u32 SK[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
#define ROR(x, n) (((x) >> (n)) | ((x) << (32 - (n))))
static void sha_block(u32 st[8], const uint8_t* p) {
  u32 w[64];
  for (int i = 0; i < 16; ++i) w[i] = ((u32)p[4 * i] << 24) | ((u32)p[4 * i + 1] << 16) | ((u32)p[4 * i + 2] << 8) | p[4 * i + 3];
  for (int i = 16; i < 64; ++i) {
    u32 s0 = ROR(w[i - 15], 7) ^ ROR(w[i - 15], 18) ^ (w[i - 15] >> 3);
    u32 s1 = ROR(w[i - 2], 17) ^ ROR(w[i - 2], 19) ^ (w[i - 2] >> 10);
    w[i] = w[i - 16] + s0 + w[i - 7] + s1;
  }
  u32 a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
  for (int i = 0; i < 64; ++i) {
    u32 t1 = h + (ROR(e, 6) ^ ROR(e, 11) ^ ROR(e, 25)) + ((e & f) ^ (~e & g)) + SK[i] + w[i];
    u32 t2 = (ROR(a, 2) ^ ROR(a, 13) ^ ROR(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
    h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
  }
  st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}
static void sha_init(sha256_t* s) {
  static const u32 iv[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  memcpy(s->h, iv, sizeof(iv)); s->total = 0; s->fill = 0;
}
static void sha_update(sha256_t* s, const uint8_t* p, size_t len) {
  s->total += len;
  while (len) {
    size_t take = 64 - s->fill; if (take > len) take = len;
    memcpy(s->buf + s->fill, p, take); s->fill += take; p += take; len -= take;
    if (s->fill == 64) { sha_block(s->h, s->buf); s->fill = 0; }
  }
}
static void sha_final(sha256_t* s, uint8_t out[32]) {
  u64 bits = s->total * 8; uint8_t pad = 0x80, z = 0, lenb[8];
  sha_update(s, &pad, 1);
  while (s->fill != 56) sha_update(s, &z, 1);
  for (int i = 0; i < 8; ++i) lenb[i] = (uint8_t)(bits >> (56 - 8 * i));
  sha_update(s, lenb, 8);
  for (int i = 0; i < 8; ++i) { out[4 * i] = s->h[i] >> 24; out[4 * i + 1] = s->h[i] >> 16; out[4 * i + 2] = s->h[i] >> 8; out[4 * i + 3] = s->h[i]; }
}

/* ---------------------------------------------------------------- big integers, 32-bit digits */
static void from_be(u32* r, const uint8_t* b, size_t len) { /* len <= 256 */
  memset(r, 0, NW * 4);
  for (size_t i = 0; i < len; ++i) { size_t p = len - 1 - i; r[i >> 2] |= (u32)b[p] << (8 * (i & 3)); }
}
static void to_be256(uint8_t* b, const u32* a) {
  for (int i = 0; i < 256; ++i) b[255 - i] = (uint8_t)(a[i >> 2] >> (8 * (i & 3)));
}
static int cmp_n(const u32* a, const u32* b, int n) {
  for (int i = n - 1; i >= 0; --i) if (a[i] != b[i]) return a[i] > b[i] ? 1 : -1;
  return 0;
}

static int bitlen(const u32* a, int n) {
  for (int i = n - 1; i >= 0; --i) if (a[i]) return 32 * i + 32 - __builtin_clz(a[i]);
  return 0;
}
static void mul_full(u32* r /*na+nb*/, const u32* a, int na, const u32* b, int nb) {
  memset(r, 0, (size_t)(na + nb) * 4);
  for (int i = 0; i < na; ++i) {
    u64 c = 0; const u64 ai = a[i];
    if (!ai) continue;
    for (int j = 0; j < nb; ++j) { u64 t = ai * b[j] + r[i + j] + c; r[i + j] = (u32)t; c = t >> 32; }
    r[i + nb] = (u32)c;
  }
}
/* r = u mod v  (Knuth D, v has nv digits with top digit non-zero, u has nu >= nv digits) */
static void mod_knuth(u32* r /*nv*/, const u32* u_in, int nu, const u32* v_in, int nv) {
  u32 u[2 * NW + 2], v[NW];
  const int s = __builtin_clz(v_in[nv - 1]);
  for (int i = nv - 1; i > 0; --i) v[i] = s ? (v_in[i] << s) | (v_in[i - 1] >> (32 - s)) : v_in[i];
  v[0] = v_in[0] << s;
  u[nu] = s ? u_in[nu - 1] >> (32 - s) : 0;
  for (int i = nu - 1; i > 0; --i) u[i] = s ? (u_in[i] << s) | (u_in[i - 1] >> (32 - s)) : u_in[i];
  u[0] = u_in[0] << s;
  for (int j = nu - nv; j >= 0; --j) {
    u64 num = ((u64)u[j + nv] << 32) | u[j + nv - 1];
    u64 qhat = num / v[nv - 1], rhat = num % v[nv - 1];
    while (qhat >> 32 || (nv > 1 && qhat * v[nv - 2] > ((rhat << 32) | u[j + nv - 2]))) {
      --qhat; rhat += v[nv - 1];
      if (rhat >> 32) break;
    }
    int64_t borrow = 0; u64 carry = 0;
    for (int i = 0; i < nv; ++i) {
      u64 p = qhat * v[i] + carry; carry = p >> 32;
      int64_t t = (int64_t)u[i + j] - (int64_t)(u32)p + borrow;
      u[i + j] = (u32)t; borrow = t >> 32;
    }
    int64_t t = (int64_t)u[j + nv] - (int64_t)carry + borrow;
    u[j + nv] = (u32)t;
    if (t < 0) { /* add back */
      u64 c = 0;
      for (int i = 0; i < nv; ++i) { u64 x = (u64)u[i + j] + v[i] + c; u[i + j] = (u32)x; c = x >> 32; }
      u[j + nv] += (u32)c;
    }
  }
  for (int i = 0; i < nv; ++i) r[i] = s ? (u[i] >> s) | ((u64)u[i + 1] << (32 - s)) : u[i];
}
/* Montgomery REDC of a 2*NW-digit value, modulus Q */
static void redc(u32* r, u32* t /* 2NW+1, destroyed */) {
  for (int i = 0; i < NW; ++i) {
    const u64 m = (u32)(t[i] * N0INV);
    u64 c = 0;
    for (int j = 0; j < NW; ++j) { u64 x = m * Q[j] + t[i + j] + c; t[i + j] = (u32)x; c = x >> 32; }
    for (int k = i + NW; c && k <= 2 * NW; ++k) { u64 x = (u64)t[k] + c; t[k] = (u32)x; c = x >> 32; }
  }
  u32* hi = t + NW;
  if (hi[NW] || cmp_n(hi, Q, NW) >= 0) {
    int64_t b = 0;
    for (int i = 0; i < NW; ++i) { int64_t x = (int64_t)hi[i] - Q[i] + b; hi[i] = (u32)x; b = x >> 32; }
  }
  memcpy(r, hi, NW * 4);
}
static void monty_mul(u32* r, const u32* a, const u32* b) {
  u32 t[2 * NW + 1];
  mul_full(t, a, NW, b, NW); t[2 * NW] = 0;
  redc(r, t);
}
/* ModpGroup::exp (modp.rs:122-128): base^e mod q; e has ne digits */
static void modpow(u32* out, const u32* base, const u32* e, int ne) {
  u32 apri[NW], ans[NW], wide[2 * NW];
  /* a * R mod q */
  memset(wide, 0, sizeof(wide));
  memcpy(wide + NW, base, NW * 4);
  mod_knuth(apri, wide, 2 * NW, Q, NW);
  memcpy(ans, RMODQ, NW * 4);
  const int nb = bitlen(e, ne);
  for (int i = 0; i < nb; ++i) {
    if ((e[i >> 5] >> (i & 31)) & 1) monty_mul(ans, ans, apri);
    if (i + 1 < nb) monty_mul(apri, apri, apri);
  }
  u32 t[2 * NW + 1];
  memset(t, 0, sizeof(t)); memcpy(t, ans, NW * 4);
  redc(out, t);
}
/* (a*b) % m */
static void mulmod(u32* r, const u32* a, const u32* b, const u32* m) {
  u32 t[2 * NW];
  mul_full(t, a, NW, b, NW);
  mod_knuth(r, t, 2 * NW, m, NW);
}

static void init_once(void) {
  if (g_init) return;
  uint8_t qb[256];
  for (int i = 0; i < 256; ++i) {
    unsigned hi = Q_HEX[2 * i], lo = Q_HEX[2 * i + 1];
    hi = hi <= '9' ? hi - '0' : hi - 'a' + 10; lo = lo <= '9' ? lo - '0' : lo - 'a' + 10;
    qb[i] = (uint8_t)(hi << 4 | lo);
  }
  from_be(Q, qb, 256);
  memcpy(QM1, Q, sizeof(Q)); QM1[0] -= 1;            /* q is odd */
  u32 inv = 1;                                        /* -q^-1 mod 2^32 by Newton */
  for (int i = 0; i < 5; ++i) inv *= 2 - Q[0] * inv;
  N0INV = (u32)(0 - inv);
  u32 wide[2 * NW]; memset(wide, 0, sizeof(wide)); wide[NW] = 1;   /* R = 2^2048 */
  /* R mod q: R has 2NW+... digits; use nu = NW+1 */
  mod_knuth(RMODQ, wide, NW + 1, Q, NW);
  g_init = 1;
}

/* framed minimal-length big-endian bytes (modp.rs:150-152, dleq.rs:58-61) */
static void frame(sha256_t* h, const u32* a) {
  uint8_t b[256], pre[8]; to_be256(b, a);
  size_t skip = 0; while (skip < 255 && b[skip] == 0) ++skip;
  u64 len = 256 - skip;
  for (int i = 0; i < 8; ++i) pre[i] = (uint8_t)(len >> (56 - 8 * i));
  sha_update(h, pre, 8); sha_update(h, b + skip, (size_t)len);
}

/* ---------------------------------------------------------------- exported API */
void ref_modpow(const uint8_t* base256, const uint8_t* exp256, uint8_t* out256) {
  init_once();
  u32 b[NW], e[NW], r[NW];
  from_be(b, base256, 256); from_be(e, exp256, 256);
  modpow(r, b, e, NW); to_be256(out256, r);
}
void ref_mulmod_q(const uint8_t* a256, const uint8_t* b256, uint8_t* out256) {
  init_once();
  u32 a[NW], b[NW], r[NW];
  from_be(a, a256, 256); from_be(b, b256, 256);
  mulmod(r, a, b, Q); to_be256(out256, r);
}

/* One share of the verifier loop (participant.rs:408-448): X_i, a1_i, a2_i as 256-byte BE. */
void ref_verify_share_work(const uint8_t* commitments, size_t t, int64_t position, const uint8_t* y256,
                           const uint8_t* Y256, const uint8_t* r256, const uint8_t* c256, uint8_t* X_out,
                           uint8_t* a1_out, uint8_t* a2_out) {
  init_once();
  u32 x[NW], e[NW], pos[NW], cj[NW], tmp[NW];
  memset(x, 0, sizeof(x)); x[0] = 1;                 /* identity, participant.rs:424 */
  memset(e, 0, sizeof(e)); e[0] = 1;                 /* exponent = 1 */
  memset(pos, 0, sizeof(pos)); pos[0] = (u32)position; pos[1] = (u32)((u64)position >> 32);
  for (size_t j = 0; j < t; ++j) {
    from_be(cj, commitments + j * 256, 256);
    modpow(tmp, cj, e, NW);                          /* exp(C_j, exponent)      :426-428 */
    mulmod(x, x, tmp, Q);                            /* x_val = mul(x_val, ..)  :429 */
    mulmod(e, e, pos, QM1);                          /* scalar_mul(..) % order  :430-433 */
  }
  u32 g[NW], y[NW], Y[NW], r[NW], c[NW], p1[NW], p2[NW], a1[NW], a2[NW];
  memset(g, 0, sizeof(g)); g[0] = 4;                 /* subgroup generator, modp.rs:65-66 */
  from_be(y, y256, 256); from_be(Y, Y256, 256); from_be(r, r256, 256); from_be(c, c256, 256);
  modpow(p1, g, r, NW); modpow(p2, x, c, NW); mulmod(a1, p1, p2, Q);   /* dleq.rs:75-77 */
  modpow(p1, y, r, NW); modpow(p2, Y, c, NW); mulmod(a2, p1, p2, Q);   /* dleq.rs:79-81 */
  to_be256(X_out, x); to_be256(a1_out, a1); to_be256(a2_out, a2);
}

/* Whole box, single thread, reference order.  Returns the verdict; digest32 gets SHA256(transcript). */
int ref_verify_distribution(const uint8_t* commitments, size_t t, const int64_t* positions, const uint8_t* pubkeys,
                            const uint8_t* shares, const uint8_t* responses, size_t n, const uint8_t* challenge256,
                            uint8_t* digest32, uint8_t* X_out, uint8_t* a1_out, uint8_t* a2_out) {
  init_once();
  sha256_t h; sha_init(&h);
  uint8_t X[256], a1[256], a2[256];
  u32 w[NW];
  for (size_t i = 0; i < n; ++i) {
    ref_verify_share_work(commitments, t, positions[i], pubkeys + i * 256, shares + i * 256, responses + i * 256,
                          challenge256, X, a1, a2);
    from_be(w, X, 256); frame(&h, w);                /* dleq.rs:95-98 order: h1, h2, a1, a2 */
    from_be(w, shares + i * 256, 256);
    { u32 red[NW]; memcpy(red, w, sizeof(w)); frame(&h, red); }
    from_be(w, a1, 256); frame(&h, w);
    from_be(w, a2, 256); frame(&h, w);
    if (X_out) memcpy(X_out + i * 256, X, 256);
    if (a1_out) memcpy(a1_out + i * 256, a1, 256);
    if (a2_out) memcpy(a2_out + i * 256, a2, 256);
  }
  uint8_t d[32], hh[32];
  sha_final(&h, d);
  if (digest32) memcpy(digest32, d, 32);
  sha256_t h2; sha_init(&h2); sha_update(&h2, d, 32); sha_final(&h2, hh);  /* hash_to_scalar, modp.rs:142-148 */
  for (int i = 0; i < 224; ++i) if (challenge256[i]) return 0;
  return memcmp(hh, challenge256 + 224, 32) == 0;
}

void ref_sha256(const uint8_t* data, size_t len, uint8_t* out32) {
  sha256_t h; sha_init(&h); sha_update(&h, data, len); sha_final(&h, out32);
}
