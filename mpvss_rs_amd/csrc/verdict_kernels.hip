// K7: the per-share Fiat-Shamir verdict of `verify_share` on the device (gfx950).
//   reference: Participant::verify_share src/participant.rs:361-386 (secp256k1 :1346-1371, ristretto255 :1789-1814)
//              -> DLEQ::verify src/dleq.rs:275-302: a fresh SHA-256 over framed(h1) framed(h2) framed(a1) framed(a2)
//              (framed(b) = u64-BE(len(b)) || b, src/dleq.rs:58-61,87-99), then
//              check: hash_to_scalar(digest) == c   (src/dleq.rs:119-126)
//   hash_to_scalar: MODP  int_BE(SHA256(digest)) mod (q-1)/2  (src/groups/modp.rs:142-148; the reduction is the identity)
//                   secp  int_BE(SHA256(digest)) mod n         (src/groups/secp256k1.rs:121-131)
//                   rist  int_LE(SHA512(digest)) mod l         (src/groups/ristretto255.rs:196-205)
// Every share box carries its own challenge, so the verdicts are independent: ONE LANE PER SHARE, the 64-byte block
// buffer of the lane's running hash in LDS (word-major across the 64 lanes of the wave: conflict-free), the message
// schedule and state in registers.  MODP elements are framed with their minimal-length big-endian bytes
// (src/groups/modp.rs:150-152: leading zero bytes are stripped, zero is one 0x00 byte).
// Work per share: 17 SHA-256 blocks (MODP) / 3 (curves) + 1-2 for hash_to_scalar; traffic 5 x 256 B in, 1 B out.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ec_consts.h"
#include "ec_scalar.h"
#include "verdict_kernels.h"

namespace {

typedef uint32_t u32;
typedef uint64_t u64;

__constant__ u32 K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

__constant__ u64 K512[80] = {
    0x428a2f98d728ae22ULL, 0x7137449123ef65cdULL, 0xb5c0fbcfec4d3b2fULL, 0xe9b5dba58189dbbcULL, 0x3956c25bf348b538ULL,
    0x59f111f1b605d019ULL, 0x923f82a4af194f9bULL, 0xab1c5ed5da6d8118ULL, 0xd807aa98a3030242ULL, 0x12835b0145706fbeULL,
    0x243185be4ee4b28cULL, 0x550c7dc3d5ffb4e2ULL, 0x72be5d74f27b896fULL, 0x80deb1fe3b1696b1ULL, 0x9bdc06a725c71235ULL,
    0xc19bf174cf692694ULL, 0xe49b69c19ef14ad2ULL, 0xefbe4786384f25e3ULL, 0x0fc19dc68b8cd5b5ULL, 0x240ca1cc77ac9c65ULL,
    0x2de92c6f592b0275ULL, 0x4a7484aa6ea6e483ULL, 0x5cb0a9dcbd41fbd4ULL, 0x76f988da831153b5ULL, 0x983e5152ee66dfabULL,
    0xa831c66d2db43210ULL, 0xb00327c898fb213fULL, 0xbf597fc7beef0ee4ULL, 0xc6e00bf33da88fc2ULL, 0xd5a79147930aa725ULL,
    0x06ca6351e003826fULL, 0x142929670a0e6e70ULL, 0x27b70a8546d22ffcULL, 0x2e1b21385c26c926ULL, 0x4d2c6dfc5ac42aedULL,
    0x53380d139d95b3dfULL, 0x650a73548baf63deULL, 0x766a0abb3c77b2a8ULL, 0x81c2c92e47edaee6ULL, 0x92722c851482353bULL,
    0xa2bfe8a14cf10364ULL, 0xa81a664bbc423001ULL, 0xc24b8b70d0f89791ULL, 0xc76c51a30654be30ULL, 0xd192e819d6ef5218ULL,
    0xd69906245565a910ULL, 0xf40e35855771202aULL, 0x106aa07032bbd1b8ULL, 0x19a4c116b8d2d0c8ULL, 0x1e376c085141ab53ULL,
    0x2748774cdf8eeb99ULL, 0x34b0bcb5e19b48a8ULL, 0x391c0cb3c5c95a63ULL, 0x4ed8aa4ae3418acbULL, 0x5b9cca4f7763e373ULL,
    0x682e6ff3d6b2b8a3ULL, 0x748f82ee5defb2fcULL, 0x78a5636f43172f60ULL, 0x84c87814a1f0ab72ULL, 0x8cc702081a6439ecULL,
    0x90befffa23631e28ULL, 0xa4506cebde82bde9ULL, 0xbef9a3f7b2c67915ULL, 0xc67178f2e372532bULL, 0xca273eceea26619cULL,
    0xd186b8c721c0c207ULL, 0xeada7dd6cde0eb1eULL, 0xf57d4f7fee6ed178ULL, 0x06f067aa72176fbaULL, 0x0a637dc5a2c898a6ULL,
    0x113f9804bef90daeULL, 0x1b710b35131c471bULL, 0x28db77f523047d84ULL, 0x32caab7b40c72493ULL, 0x3c9ebe0a15c9bebcULL,
    0x431d67c49c100d4cULL, 0x4cc5d4becb3e42b6ULL, 0x597f299cfc657e2aULL, 0x5fcb6fab3ad6faecULL, 0x6c44198c4a475817ULL};

__device__ __forceinline__ u32 rotr32(u32 x, int n) { return __builtin_amdgcn_alignbit(x, x, n); }
__device__ __forceinline__ u64 rotr64(u64 x, int n) { return (x >> n) | (x << (64 - n)); }

// one SHA-256 compression; w[16] = the block as big-endian words
__device__ __forceinline__ void sha256_compress(u32 (&st)[8], u32 (&w)[16]) {
  u32 a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
#pragma unroll
  for (int i = 0; i < 64; ++i) {
    if (i >= 16) {
      const u32 w15 = w[(i + 1) & 15], w2 = w[(i + 14) & 15];
      const u32 s0 = rotr32(w15, 7) ^ rotr32(w15, 18) ^ (w15 >> 3);
      const u32 s1 = rotr32(w2, 17) ^ rotr32(w2, 19) ^ (w2 >> 10);
      w[i & 15] = w[i & 15] + s0 + w[(i + 9) & 15] + s1;
    }
    const u32 t1 = h + (rotr32(e, 6) ^ rotr32(e, 11) ^ rotr32(e, 25)) + ((e & f) ^ (~e & g)) + K256[i] + w[i & 15];
    const u32 t2 = (rotr32(a, 2) ^ rotr32(a, 13) ^ rotr32(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
    h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
  }
  st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}

// One block from the lane's LDS column, out of line: the streaming code below reaches it from many places and the
// 64 unrolled rounds must exist once.  State travels in registers (struct by value).
struct Sha256State {
  u32 v[8];
};
__device__ __noinline__ Sha256State sha256_block_from_lds(Sha256State s, const u32* buf) {
  u32 w[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) w[i] = buf[i * 64];
  sha256_compress(s.v, w);
  return s;
}

// Streaming SHA-256 of one lane.  The 64-byte block buffer lives in LDS: word w of the lane's block is buf[w * 64]
// (buf already points at the lane's column), bytes are stored so that the word reads back big-endian.
struct LaneSha256 {
  u32 st[8];
  u32 fill;      // bytes in the block buffer
  u32 total;     // bytes absorbed so far
  u32* buf;

  __device__ __forceinline__ void init(u32* lane_column) {
    st[0] = 0x6a09e667; st[1] = 0xbb67ae85; st[2] = 0x3c6ef372; st[3] = 0xa54ff53a;
    st[4] = 0x510e527f; st[5] = 0x9b05688c; st[6] = 0x1f83d9ab; st[7] = 0x5be0cd19;
    fill = 0;
    total = 0;
    buf = lane_column;
  }
  __device__ __forceinline__ void flush() {
    Sha256State s;
#pragma unroll
    for (int i = 0; i < 8; ++i) s.v[i] = st[i];
    s = sha256_block_from_lds(s, buf);
#pragma unroll
    for (int i = 0; i < 8; ++i) st[i] = s.v[i];
    fill = 0;
  }
  __device__ __forceinline__ void push(u32 byte) {
    reinterpret_cast<uint8_t*>(buf + (fill >> 2) * 64)[3 - (fill & 3)] = (uint8_t)byte;
    ++fill;
    ++total;
    if (fill == 64) flush();
  }
  // four message bytes given as one big-endian word; fast when the stream is word-aligned
  __device__ __forceinline__ void push_be32(u32 word) {
    if ((fill & 3) == 0) {
      buf[(fill >> 2) * 64] = word;
      fill += 4;
      total += 4;
      if (fill == 64) flush();
    } else {
      push(word >> 24); push((word >> 16) & 0xff); push((word >> 8) & 0xff); push(word & 0xff);
    }
  }
  __device__ __forceinline__ void push_len(u32 len) {       // u64::to_be_bytes(len), src/dleq.rs:58-61
    push_be32(0);
    push_be32(len);
  }
  __device__ __forceinline__ void finish(u32 (&digest)[8]) {
    const u32 bits = total * 8;
    push(0x80);
    while (fill != 56) push(0);
    push_be32(0);
    push_be32(bits);                                        // flushes
#pragma unroll
    for (int i = 0; i < 8; ++i) digest[i] = st[i];
  }
};

// SHA-256 of a 32-byte digest given as 8 big-endian words (the inner hash of hash_to_scalar): one block
__device__ __forceinline__ void sha256_of_digest(u32 (&out)[8], const u32 (&d)[8]) {
  u32 st[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  u32 w[16];
#pragma unroll
  for (int i = 0; i < 8; ++i) w[i] = d[i];
  w[8] = 0x80000000u;
#pragma unroll
  for (int i = 9; i < 15; ++i) w[i] = 0;
  w[15] = 256;
  sha256_compress(st, w);
#pragma unroll
  for (int i = 0; i < 8; ++i) out[i] = st[i];
}

// SHA-512 of a 32-byte digest (8 big-endian 32-bit words): one block; out = 8 big-endian 64-bit words
__device__ __forceinline__ void sha512_of_digest(u64 (&out)[8], const u32 (&d)[8]) {
  u64 st[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
               0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
  u64 w[16];
#pragma unroll
  for (int i = 0; i < 4; ++i) w[i] = ((u64)d[2 * i] << 32) | d[2 * i + 1];
  w[4] = 0x8000000000000000ULL;
#pragma unroll
  for (int i = 5; i < 15; ++i) w[i] = 0;
  w[15] = 256;
  u64 a = st[0], b = st[1], c = st[2], dd = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
#pragma unroll 8
  for (int i = 0; i < 80; ++i) {
    if (i >= 16) {
      const u64 w15 = w[(i + 1) & 15], w2 = w[(i + 14) & 15];
      const u64 s0 = rotr64(w15, 1) ^ rotr64(w15, 8) ^ (w15 >> 7);
      const u64 s1 = rotr64(w2, 19) ^ rotr64(w2, 61) ^ (w2 >> 6);
      w[i & 15] = w[i & 15] + s0 + w[(i + 9) & 15] + s1;
    }
    const u64 t1 = h + (rotr64(e, 14) ^ rotr64(e, 18) ^ rotr64(e, 41)) + ((e & f) ^ (~e & g)) + K512[i] + w[i & 15];
    const u64 t2 = (rotr64(a, 28) ^ rotr64(a, 34) ^ rotr64(a, 39)) + ((a & b) ^ (a & c) ^ (b & c));
    h = g; g = f; f = e; e = dd + t1; dd = c; c = b; b = a; a = t1 + t2;
  }
  out[0] = st[0] + a; out[1] = st[1] + b; out[2] = st[2] + c; out[3] = st[3] + dd;
  out[4] = st[4] + e; out[5] = st[5] + f; out[6] = st[6] + g; out[7] = st[7] + h;
}

// frame one MODP element: minimal-length big-endian bytes of a 256-byte value (modp.rs:150-152)
__device__ __forceinline__ void frame_modp(LaneSha256& h, const uint8_t* __restrict__ e) {
  const u32* e32 = reinterpret_cast<const u32*>(e);
  int skip = 255;                                        // zero is hashed as one 0x00 byte
  for (int k = 0; k < 64; ++k) {
    const u32 w = e32[k];                                // little-endian load: byte 4k is the low byte
    if (w != 0) {
      skip = 4 * k + (__builtin_ctz(w) >> 3);
      break;
    }
  }
  h.push_len(256u - (u32)skip);
  int i = skip;
  while ((i & 3) != 0 && i < 256) h.push(e[i++]);
  for (; i < 256; i += 4) h.push_be32(__builtin_bswap32(e32[i >> 2]));
}

template <int LEN>
__device__ __forceinline__ void frame_fixed(LaneSha256& h, const uint8_t* __restrict__ e) {
  h.push_len(LEN);
#pragma unroll 1
  for (int i = 0; i < LEN; ++i) h.push(e[i]);
}

// little-endian words of a 32-byte scalar in the curve's byte order
template <bool BE>
__device__ __forceinline__ void scalar_words(u32 (&w)[8], const uint8_t* __restrict__ s) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    u32 v = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int le = 4 * i + b;
      v |= (u32)s[BE ? 31 - le : le] << (8 * b);
    }
    w[i] = v;
  }
}

template <class O>
__device__ __forceinline__ bool below_order(const u32 (&w)[8]) {
  bool lt = false, decided = false;
#pragma unroll
  for (int i = 7; i >= 0; --i) {
    if (!decided && w[i] != O::n(i)) {
      lt = w[i] < O::n(i);
      decided = true;
    }
  }
  return lt;
}


// Per-share well-formedness of a MODP distribution block (the verdict bytes a sharded verification
// all-gathers for W_A): out[i] = 1 iff 0 < y_i < q, 0 < Y_i < q and r_i < q - 1.  The reference does not validate its
// inputs (src/groups/modp.rs:154-156 decodes anything, src/participant.rs:408-448 hashes whatever it computed), so this
// byte never changes the box verdict; it tells an operator which rank holds a share that is not a canonical encoding.
// 16 lanes per number: a lane compares its 16-byte piece, the pieces are combined through a ballot (piece 0 = the most
// significant bytes of the big-endian encoding).  bounds: [2][256] = q, q - 1 as big-endian bytes.
__device__ __forceinline__ void cmp16(const uint8_t* v, const uint8_t* b, int& c, bool& nz) {
  c = 0;
  nz = false;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int x = v[k], y = b[k];
    nz = nz || x != 0;
    if (c == 0) c = x < y ? -1 : (x > y ? 1 : 0);
  }
}
__global__ void __launch_bounds__(256) k_modp_wellformed(const uint8_t* __restrict__ y, const uint8_t* __restrict__ Y,
                                                         const uint8_t* __restrict__ r, const uint8_t* __restrict__ bounds,
                                                         int n, uint8_t* __restrict__ out) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int share = gid >> 4, part = gid & 15;
  const int s = share < n ? share : n - 1;
  const int sh = (threadIdx.x & 63) & 48;
  bool good = true;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const uint8_t* arr = a == 0 ? y : (a == 1 ? Y : r);
    int c;
    bool nz;
    cmp16(arr + (size_t)s * 256 + 16 * part, bounds + (a == 2 ? 256 : 0) + 16 * part, c, nz);
    const u32 ne = (u32)(__ballot(c != 0) >> sh) & 0xffffu;
    const u32 lt = (u32)(__ballot(c < 0) >> sh) & 0xffffu;
    const u32 any = (u32)(__ballot(nz) >> sh) & 0xffffu;
    const bool below = ne != 0 && ((lt >> (__ffs((int)ne) - 1)) & 1u);
    good = good && below && (a == 2 || any != 0);
  }
  if (share < n && part == 0) out[share] = good ? 1 : 0;
}
}  // namespace

// ---- MODP-2048 ---------------------------------------------------------------------------------------------
// verdict[i] = ( int_BE(SHA256(SHA256(framed(h1_i) framed(h2_i) framed(a1_i) framed(a2_i)))) == c_i )
// h1 = pk_i, h2 = Y_i (participant.rs:376-385 builds DLEQ(G, pk, S, Y)); a1, a2 from the dual-exponentiation kernels.
extern "C" __global__ void __launch_bounds__(64)
k_modp_share_verdict(const uint8_t* __restrict__ h1, const uint8_t* __restrict__ h2, const uint8_t* __restrict__ a1,
                     const uint8_t* __restrict__ a2, const uint8_t* __restrict__ c, int count,
                     uint8_t* __restrict__ verdict, uint8_t* __restrict__ c_out32) {
  __shared__ u32 lds[16 * 64];
  const int x = blockIdx.x * 64 + threadIdx.x;
  if (x >= count) return;
  LaneSha256 h;
  h.init(lds + threadIdx.x);
  const size_t off = (size_t)x * 256;
  frame_modp(h, h1 + off);
  frame_modp(h, h2 + off);
  frame_modp(h, a1 + off);
  frame_modp(h, a2 + off);
  u32 digest[8], hs[8];
  h.finish(digest);
  sha256_of_digest(hs, digest);                         // modp.rs:142-148 hashes the digest again
  // c as a 256-byte big-endian integer must equal the 256-bit hash: the top 224 bytes are zero, the rest matches.
  // (mod (q-1)/2 is the identity on 256-bit values; a challenge >= 2^256 can never match.)
  if (c_out32 != nullptr) {      // the prover's side (extract_secret_share, participant.rs:329-343): hand the challenge out
    u32* o = reinterpret_cast<u32*>(c_out32 + (size_t)x * 32);
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = __builtin_bswap32(hs[k]);
    return;
  }
  const u32* c32 = reinterpret_cast<const u32*>(c + off);
  u32 diff = 0;
  for (int k = 0; k < 56; ++k) diff |= c32[k];
#pragma unroll
  for (int k = 0; k < 8; ++k) diff |= __builtin_bswap32(c32[56 + k]) ^ hs[k];
  verdict[x] = diff == 0 ? 1 : 0;
}

// ---- secp256k1 / ristretto255 ------------------------------------------------------------------------------
// GROUP 1: 33-byte elements, hash_to_scalar = int_BE(SHA256(digest)) mod n, scalars big-endian.
// GROUP 2: 32-byte elements, hash_to_scalar = int_LE(SHA512(digest)) mod l, scalars little-endian.
// ok[i] = 0 when c_i or r_i is not below the group order (the reference's scalar types cannot hold such a value).
template <int GROUP>
__device__ __forceinline__ void ec_share_verdict_body(const uint8_t* __restrict__ h1, const uint8_t* __restrict__ h2,
                                                      const uint8_t* __restrict__ a1, const uint8_t* __restrict__ a2,
                                                      const uint8_t* __restrict__ c, const uint8_t* __restrict__ r,
                                                      int count, uint8_t* __restrict__ verdict, uint8_t* __restrict__ ok,
                                                      u32* lds) {
  constexpr int LEN = GROUP == 1 ? 33 : 32;
  const int x = blockIdx.x * 64 + threadIdx.x;
  if (x >= count) return;
  LaneSha256 h;
  h.init(lds + threadIdx.x);
  const size_t off = (size_t)x * LEN;
  frame_fixed<LEN>(h, h1 + off);
  frame_fixed<LEN>(h, h2 + off);
  frame_fixed<LEN>(h, a1 + off);
  frame_fixed<LEN>(h, a2 + off);
  u32 digest[8];
  h.finish(digest);
  u32 cw[8], rw[8], want[8];
  scalar_words<GROUP == 1>(cw, c + (size_t)x * 32);
  scalar_words<GROUP == 1>(rw, r + (size_t)x * 32);
  if (GROUP == 1) {
    u32 hs[8];
    sha256_of_digest(hs, digest);
#pragma unroll
    for (int i = 0; i < 8; ++i) want[i] = hs[7 - i];          // big-endian words -> little-endian limbs
    if (!below_order<ec::OrderSecp>(want)) {                  // 2^256 < 2n: one subtraction reduces
      u64 borrow = 0;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const u64 d = (u64)want[i] - ec::OrderSecp::n(i) - borrow;
        want[i] = (u32)d;
        borrow = (d >> 63) & 1;
      }
    }
    ok[x] = (below_order<ec::OrderSecp>(cw) && below_order<ec::OrderSecp>(rw)) ? 1 : 0;
  } else {
    u64 wide[8];
    sha512_of_digest(wide, digest);
    // the 64 output bytes as a little-endian integer: byte k of the output is byte (7 - k % 8) of word k / 8
    ec::Sc lo, hi, one, r2, t1, t2;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const u64 wl = __builtin_bswap64(wide[i / 2]), wh = __builtin_bswap64(wide[4 + i / 2]);
      lo.v[i] = (u32)(wl >> (32 * (i & 1)));
      hi.v[i] = (u32)(wh >> (32 * (i & 1)));
      one.v[i] = i == 0 ? 1u : 0u;
      r2.v[i] = ec::OrderEd::r2(i);
    }
    typedef ec::ScalarField<ec::OrderEd> SF;
    SF::mont_mul(t1, lo, r2);        // lo * R mod l
    SF::mont_mul(t1, t1, one);       // lo mod l
    SF::mont_mul(t2, hi, r2);        // hi * 2^256 mod l
    // (t1 + t2) mod l
    u64 carry = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      carry += (u64)t1.v[i] + t2.v[i];
      want[i] = (u32)carry;
      carry >>= 32;
    }
    if (!below_order<ec::OrderEd>(want)) {
      u64 borrow = 0;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const u64 d = (u64)want[i] - ec::OrderEd::n(i) - borrow;
        want[i] = (u32)d;
        borrow = (d >> 63) & 1;
      }
    }
    ok[x] = (below_order<ec::OrderEd>(cw) && below_order<ec::OrderEd>(rw)) ? 1 : 0;
  }
  u32 diff = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) diff |= want[i] ^ cw[i];
  verdict[x] = diff == 0 ? 1 : 0;
}

extern "C" __global__ void __launch_bounds__(64)
k_secp_share_verdict(const uint8_t* h1, const uint8_t* h2, const uint8_t* a1, const uint8_t* a2, const uint8_t* c,
                     const uint8_t* r, int count, uint8_t* verdict, uint8_t* ok) {
  __shared__ u32 lds[16 * 64];
  ec_share_verdict_body<1>(h1, h2, a1, a2, c, r, count, verdict, ok, lds);
}
extern "C" __global__ void __launch_bounds__(64)
k_rist_share_verdict(const uint8_t* h1, const uint8_t* h2, const uint8_t* a1, const uint8_t* a2, const uint8_t* c,
                     const uint8_t* r, int count, uint8_t* verdict, uint8_t* ok) {
  __shared__ u32 lds[16 * 64];
  ec_share_verdict_body<2>(h1, h2, a1, a2, c, r, count, verdict, ok, lds);
}

// first_bad = min index of a scalar that is not below the group order (INT_MAX when all are canonical); stride 32
template <int GROUP>
__device__ __forceinline__ void check_scalars_body(const uint8_t* __restrict__ s, int count, int* __restrict__ first_bad) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= count) return;
  u32 w[8];
  scalar_words<GROUP == 1>(w, s + (size_t)x * 32);
  const bool ok = GROUP == 1 ? below_order<ec::OrderSecp>(w) : below_order<ec::OrderEd>(w);
  if (!ok) atomicMin(first_bad, x);
}
extern "C" __global__ void k_secp_check_scalars(const uint8_t* s, int count, int* first_bad) { check_scalars_body<1>(s, count, first_bad); }
extern "C" __global__ void k_rist_check_scalars(const uint8_t* s, int count, int* first_bad) { check_scalars_body<2>(s, count, first_bad); }

// ---- launchers ------------------------------------------------------------------------------------------------
extern "C" int verdict_launch_modp_wellformed(const uint8_t* y, const uint8_t* Y, const uint8_t* r, const uint8_t* bounds,
                                              int count, uint8_t* out, hipStream_t s) {
  if (count <= 0) return 0;
  hipLaunchKernelGGL(k_modp_wellformed, dim3((count + 15) / 16), dim3(256), 0, s, y, Y, r, bounds, count, out);
  return (int)hipGetLastError();
}
extern "C" int verdict_launch_modp(const uint8_t* h1, const uint8_t* h2, const uint8_t* a1, const uint8_t* a2,
                                   const uint8_t* c, int count, uint8_t* verdict, hipStream_t s) {
  if (count <= 0) return 0;
  hipLaunchKernelGGL(k_modp_share_verdict, dim3((count + 63) / 64), dim3(64), 0, s, h1, h2, a1, a2, c, count, verdict,
                     (uint8_t*)nullptr);
  return (int)hipGetLastError();
}
extern "C" int verdict_launch_modp_challenge(const uint8_t* h1, const uint8_t* h2, const uint8_t* a1, const uint8_t* a2, int count,
                                             uint8_t* c_out32, hipStream_t s) {
  if (count <= 0) return 0;
  hipLaunchKernelGGL(k_modp_share_verdict, dim3((count + 63) / 64), dim3(64), 0, s, h1, h2, a1, a2, (const uint8_t*)nullptr, count,
                     (uint8_t*)nullptr, c_out32);
  return (int)hipGetLastError();
}
extern "C" int verdict_launch_ec(int group, const uint8_t* h1, const uint8_t* h2, const uint8_t* a1, const uint8_t* a2,
                                 const uint8_t* c, const uint8_t* r, int count, uint8_t* verdict, uint8_t* ok,
                                 hipStream_t s) {
  if (count <= 0) return 0;
  if (group == 1)
    hipLaunchKernelGGL(k_secp_share_verdict, dim3((count + 63) / 64), dim3(64), 0, s, h1, h2, a1, a2, c, r, count, verdict, ok);
  else
    hipLaunchKernelGGL(k_rist_share_verdict, dim3((count + 63) / 64), dim3(64), 0, s, h1, h2, a1, a2, c, r, count, verdict, ok);
  return (int)hipGetLastError();
}
extern "C" int verdict_launch_check_scalars(int group, const uint8_t* scalars, int count, int* first_bad, hipStream_t s) {
  if (count <= 0) return 0;
  if (group == 1) hipLaunchKernelGGL(k_secp_check_scalars, dim3((count + 255) / 256), dim3(256), 0, s, scalars, count, first_bad);
  else hipLaunchKernelGGL(k_rist_check_scalars, dim3((count + 255) / 256), dim3(256), 0, s, scalars, count, first_bad);
  return (int)hipGetLastError();
}


// ---------------------------------------------------------------------------------------------------------------------
// Measurement aid (mpvss_issue_probe, bench.py): what one SIMD sustains of the instruction the engine's kernels are made of,
// on THIS device, now -- every wave issues `iters` x 64 independent-enough VALU instructions (8 accumulator chains) and stamps
// its shader-clock and wall-clock (100 MHz) time; 4 waves per SIMD.  kind 0: v_mad_u64_u32 (the limb product of all three
// groups' kernels); kind 1: 32-bit integer work (v_add3_u32 / v_and_b32 / v_lshl_add_u32).
// ---------------------------------------------------------------------------------------------------------------------
#define PROBE_REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define PROBE_REP64(X) PROBE_REP8(X) PROBE_REP8(X) PROBE_REP8(X) PROBE_REP8(X) PROBE_REP8(X) PROBE_REP8(X) PROBE_REP8(X) PROBE_REP8(X)
extern "C" __global__ void __launch_bounds__(256) k_issue_probe(uint32_t* out, unsigned long long* stamps, int iters, int kind, uint32_t seed) {
  uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9e3779b9u;
  uint64_t p[8];
  uint32_t q[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { p[k] = a + k; q[k] = b + k; }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  if (kind == 0) {
    for (int it = 0; it < iters; ++it) {
#define X(k) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(p[k]) : "v"(a), "v"(b) : "vcc");
      PROBE_REP64(X)
#undef X
    }
  } else if (kind == 1) {
    for (int it = 0; it < iters; ++it) {
#define X(k) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(q[k]) : "v"(a), "v"(b));
      PROBE_REP8(X) PROBE_REP8(X) PROBE_REP8(X)
#undef X
#define X(k) asm volatile("v_and_b32 %0, %0, %1" : "+v"(q[k]) : "v"(b));
      PROBE_REP8(X) PROBE_REP8(X)
#undef X
#define X(k) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(q[k]) : "v"(a));
      PROBE_REP8(X) PROBE_REP8(X) PROBE_REP8(X)
#undef X
    }
  } else if (kind == 2) {          // the 64-bit shifts and shift-adds of a retire step (two passes each)
    const uint64_t c64 = ((uint64_t)b << 32) | a;
    for (int it = 0; it < iters; ++it) {
#define X(k) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(p[k]) : "v"(c64));
      PROBE_REP8(X) PROBE_REP8(X) PROBE_REP8(X) PROBE_REP8(X)
#undef X
#define X(k) asm volatile("v_lshrrev_b64 %0, 3, %0" : "+v"(p[k]));
      PROBE_REP8(X) PROBE_REP8(X) PROBE_REP8(X) PROBE_REP8(X)
#undef X
    }
  } else if (kind == 3) {          // the lane-half exchange of the pair layout
    for (int it = 0; it < iters; ++it) {
#define X(k) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(q[k]), "+v"(q[(k + 1) & 7]));
      PROBE_REP64(X)
#undef X
    }
  } else if (kind == 4) {
    for (int it = 0; it < iters; ++it) {
#define X(k) asm volatile("v_mov_b64 %0, %1" : "=v"(p[k]) : "v"(p[(k + 1) & 7]));
      PROBE_REP64(X)
#undef X
    }
  } else {                         // VOP2 32-bit add
    for (int it = 0; it < iters; ++it) {
#define X(k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(q[k]) : "v"(a));
      PROBE_REP64(X)
#undef X
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) acc ^= p[k] ^ q[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(acc ^ (acc >> 32));
  if ((threadIdx.x & 63) == 0) {
    const size_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    stamps[2 * w] = t1 - t0;
    stamps[2 * w + 1] = r1 - r0;
  }
}

extern "C" int issue_probe_launch(uint32_t* out, unsigned long long* stamps, int blocks, int iters, int kind, hipStream_t s) {
  hipLaunchKernelGGL(k_issue_probe, dim3(blocks), dim3(256), 0, s, out, stamps, iters, kind, 12345u);
  return (int)hipGetLastError();
}


// ---------------------------------------------------------------------------------------------------------------------
// The curve groups' scalar side of the dealer on the device (one share per lane, 8 x 32-bit Montgomery arithmetic mod the
// group order, ec_scalar.h):
//   P(i) mod n     Polynomial::get_value src/polynomial.rs:50-58 followed by the caller's reduction (`% n`
//                  src/participant.rs:1155-1157, Scalar arithmetic :1619-1621): Horner's rule, the position as `position as u64`
//   r_i = w_i - P(i) c mod n     src/dleq.rs:42-50 through Group::scalar_mul / scalar_sub (secp256k1.rs:173-181, ristretto255.rs:244-252)
// Scalars cross as 32 bytes in the group's byte order (secp256k1 big-endian, ristretto255 little-endian); any 256-bit input is
// reduced, as the host functions of the same ABI do (mpvss_ec_poly_eval, mpvss_ec_dleq_responses).
// ---------------------------------------------------------------------------------------------------------------------
namespace {
template <bool BE>
__device__ __forceinline__ void sc_load(ec::Sc& a, const uint8_t* __restrict__ b) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const uint8_t* q = BE ? b + 28 - 4 * i : b + 4 * i;
    a.v[i] = BE ? (((u32)q[0] << 24) | ((u32)q[1] << 16) | ((u32)q[2] << 8) | q[3])
                : (((u32)q[3] << 24) | ((u32)q[2] << 16) | ((u32)q[1] << 8) | q[0]);
  }
}
template <bool BE>
__device__ __forceinline__ void sc_store(uint8_t* __restrict__ b, const ec::Sc& a) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint8_t* q = BE ? b + 28 - 4 * i : b + 4 * i;
    const u32 v = a.v[i];
    if (BE) { q[0] = (uint8_t)(v >> 24); q[1] = (uint8_t)(v >> 16); q[2] = (uint8_t)(v >> 8); q[3] = (uint8_t)v; }
    else { q[0] = (uint8_t)v; q[1] = (uint8_t)(v >> 8); q[2] = (uint8_t)(v >> 16); q[3] = (uint8_t)(v >> 24); }
  }
}
// r = a + b mod n for reduced a, b
template <class O>
__device__ __forceinline__ void sc_add(ec::Sc& r, const ec::Sc& a, const ec::Sc& b) {
  u32 t[8];
  u64 c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    c += (u64)a.v[i] + b.v[i];
    t[i] = (u32)c;
    c >>= 32;
  }
  bool ge = c != 0;
  if (!ge) {
    ge = true;
    bool decided = false;
#pragma unroll
    for (int i = 7; i >= 0; --i)
      if (!decided && t[i] != O::n(i)) { ge = t[i] > O::n(i); decided = true; }
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
// r = a - b mod n for reduced a, b
template <class O>
__device__ __forceinline__ void sc_sub(ec::Sc& r, const ec::Sc& a, const ec::Sc& b) {
  u32 t[8];
  u64 borrow = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const u64 d = (u64)a.v[i] - b.v[i] - borrow;
    t[i] = (u32)d;
    borrow = (d >> 63) & 1;
  }
  const u32 mask = borrow ? 0xffffffffu : 0u;
  u64 c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    c += (u64)t[i] + (O::n(i) & mask);
    r.v[i] = (u32)c;
    c >>= 32;
  }
}
template <class O>
__device__ __forceinline__ void sc_consts(ec::Sc& r2, ec::Sc& one) {
#pragma unroll
  for (int i = 0; i < 8; ++i) { r2.v[i] = O::r2(i); one.v[i] = i == 0 ? 1u : 0u; }
}

// coef_m[j] = coeffs[j] * R mod n (reduced): t lanes
template <class O, bool BE>
__global__ void k_ec_coeffs_to_mont(const uint8_t* __restrict__ coeffs, int t, u32* __restrict__ coef_m) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= t) return;
  ec::Sc a, r2, one;
  sc_consts<O>(r2, one);
  sc_load<BE>(a, coeffs + (size_t)j * 32);
  ec::ScalarField<O>::mont_mul(a, a, r2);
#pragma unroll
  for (int i = 0; i < 8; ++i) coef_m[(size_t)j * 8 + i] = a.v[i];
}

template <class O, bool BE>
__global__ void __launch_bounds__(64) k_ec_poly_eval(const u32* __restrict__ coef_m, int t, const int64_t* __restrict__ positions,
                                                    int count, uint8_t* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  typedef ec::ScalarField<O> F;
  ec::Sc x, acc, c, r2, one;
  sc_consts<O>(r2, one);
  const uint64_t pos = (uint64_t)positions[i];          // `position as u64` (participant.rs:1419, 1862)
#pragma unroll
  for (int k = 0; k < 8; ++k) x.v[k] = 0;
  x.v[0] = (u32)pos;
  x.v[1] = (u32)(pos >> 32);
  F::mont_mul(x, x, r2);                                // x R
#pragma unroll
  for (int k = 0; k < 8; ++k) acc.v[k] = coef_m[(size_t)(t - 1) * 8 + k];
  for (int j = t - 2; j >= 0; --j) {
    F::mont_mul(acc, acc, x);
#pragma unroll
    for (int k = 0; k < 8; ++k) c.v[k] = coef_m[(size_t)j * 8 + k];
    sc_add<O>(acc, acc, c);
  }
  F::mont_mul(acc, acc, one);                           // out of the Montgomery domain, reduced
  sc_store<BE>(out + (size_t)i * 32, acc);
}

template <class O, bool BE>
__global__ void __launch_bounds__(64) k_ec_responses(const uint8_t* __restrict__ w, const uint8_t* __restrict__ alpha,
                                                    const uint8_t* __restrict__ c, size_t c_stride, int count, uint8_t* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  typedef ec::ScalarField<O> F;
  ec::Sc ww, aa, cc, r2, one, prod;
  sc_consts<O>(r2, one);
  sc_load<BE>(ww, w + (size_t)i * 32);
  sc_load<BE>(aa, alpha + (size_t)i * 32);
  sc_load<BE>(cc, c + (size_t)i * c_stride);
  F::mont_mul(aa, aa, r2);                              // alpha R
  F::mont_mul(prod, aa, cc);                            // alpha c mod n
  F::mont_mul(ww, ww, r2);
  F::mont_mul(ww, ww, one);                             // w mod n
  sc_sub<O>(ww, ww, prod);
  sc_store<BE>(out + (size_t)i * 32, ww);
}
}  // namespace

extern "C" int ec_scalar_launch_poly_eval(int group, const uint8_t* coeffs_dev, int t, const int64_t* positions, int count,
                                          uint32_t* coef_m_scratch, uint8_t* out, hipStream_t s) {
  if (count <= 0 || t <= 0) return 0;
  const dim3 gt((t + 63) / 64), gn((count + 63) / 64), b(64);
  if (group == 1) {
    hipLaunchKernelGGL((k_ec_coeffs_to_mont<ec::OrderSecp, true>), gt, b, 0, s, coeffs_dev, t, coef_m_scratch);
    hipLaunchKernelGGL((k_ec_poly_eval<ec::OrderSecp, true>), gn, b, 0, s, coef_m_scratch, t, positions, count, out);
  } else {
    hipLaunchKernelGGL((k_ec_coeffs_to_mont<ec::OrderEd, false>), gt, b, 0, s, coeffs_dev, t, coef_m_scratch);
    hipLaunchKernelGGL((k_ec_poly_eval<ec::OrderEd, false>), gn, b, 0, s, coef_m_scratch, t, positions, count, out);
  }
  return (int)hipGetLastError();
}
extern "C" int ec_scalar_launch_responses(int group, const uint8_t* w, const uint8_t* alpha, const uint8_t* c, size_t c_stride,
                                          int count, uint8_t* out, hipStream_t s) {
  if (count <= 0) return 0;
  const dim3 gn((count + 63) / 64), b(64);
  if (group == 1)
    hipLaunchKernelGGL((k_ec_responses<ec::OrderSecp, true>), gn, b, 0, s, w, alpha, c, c_stride, count, out);
  else
    hipLaunchKernelGGL((k_ec_responses<ec::OrderEd, false>), gn, b, 0, s, w, alpha, c, c_stride, count, out);
  return (int)hipGetLastError();
}
