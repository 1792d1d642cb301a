// secp256k1 / ristretto255 batch kernels for gfx950: one share per lane, 64-lane workgroups.
//   reference: src/groups/secp256k1.rs:91-107, src/groups/ristretto255.rs:161-177 (exp / mul),
//              src/participant.rs:1404-1417, 1847-1860 (X_i = sum_j i^j C_j), src/dleq.rs:66-84.
// All group elements cross the kernel boundary in the reference's canonical encodings (33-byte SEC1
// compressed / 32-byte ristretto255), scalars as 32 bytes (big-endian / little-endian).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "ec_curves.h"
#include "ec_scalar.h"
#include "ec_glv.h"
#include "ec_quad.h"
#include "ec_kernels.h"

using namespace ec;

namespace {

template <class C>
__device__ __forceinline__ void load_point_aos(typename C::Point& p, const u32* __restrict__ src) {
  u32* w = reinterpret_cast<u32*>(&p);
#pragma unroll
  for (int i = 0; i < C::POINT_WORDS; ++i) w[i] = src[i];
}
template <class C>
__device__ __forceinline__ void store_point_aos(u32* __restrict__ dst, const typename C::Point& p) {
  const u32* w = reinterpret_cast<const u32*>(&p);
#pragma unroll
  for (int i = 0; i < C::POINT_WORDS; ++i) dst[i] = w[i];
}

// encoded points -> decoded points (array of structs), ok flags
template <class C>
__device__ __forceinline__ void decode_body(const uint8_t* __restrict__ enc, int count, u32* __restrict__ pts,
                                            uint8_t* __restrict__ ok) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= count) return;
  typename C::Point p;
  const bool good = C::decode(p, enc + (size_t)x * C::ENC_LEN);
  store_point_aos<C>(pts + (size_t)x * C::POINT_WORDS, p);
  ok[x] = good ? 1 : 0;
}

// X_i = sum_j (i^j) C_j by Horner's rule:  X = ((C_{t-1} * i + C_{t-2}) * i + ...) * i + C_0.
// Same group element as the reference's loop (participant.rs:1411-1417 reduces i^j mod the group
// order, which is the order of every element of these prime-order groups).
// gate / want: the forward-difference kernels and Horner's rule exclude each other through a device flag when the
// positions live in device memory (the choice then needs no host synchronisation); gate == null: always run
#define EC_GATE_CHECK(gate, want) \
  if ((gate) != nullptr && *(gate) != (want)) return
// Several boxes in one launch (blockIdx.y = box): the X path of a box is a chain of narrow launches, and the device runs
// only as many launches side by side as it has hardware queues -- batching the boxes of one mpvss_ec_verify_many call
// makes each launch B times wider instead.  Per-box strides of the operands (0 for a single box):
struct BoxStride {
  size_t cm;      // decoded commitments, u32 words
  size_t pos;     // positions, elements
  size_t pts;     // internal points, u32 words
  size_t state;   // difference tables, u32 words
  size_t enc;     // encodings, bytes
};
#define EC_BOX_GATE_CHECK(gate, want) \
  if ((gate) != nullptr && (gate)[blockIdx.y] != (want)) return

template <class C>
__device__ __forceinline__ void commit_eval_body(const u32* __restrict__ cm, int t, const int64_t* __restrict__ positions,
                                                 int count, uint8_t* __restrict__ x_enc) {
  const int xi = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = xi < count;
  const int x = live ? xi : count - 1;
  const uint64_t pos = (uint64_t)positions[x];
  int nb = (pos == 0) ? 0 : 64 - __builtin_clzll(pos);
  typename C::Point acc, cj, r;
  load_point_aos<C>(acc, cm + (size_t)(t - 1) * C::POINT_WORDS);
  for (int j = t - 2; j >= 0; --j) {
    small_scalar_mul<C>(r, acc, pos, nb);
    load_point_aos<C>(cj, cm + (size_t)j * C::POINT_WORDS);
    C::add(acc, r, cj);
  }
  if (live) C::encode(x_enc + (size_t)x * C::ENC_LEN, acc);
}

// out = k1 * P1 + k2 * P2.  p1_stride / k2_stride 0 = shared operand; p2 == nullptr: out = k1 * P1.
template <class C>
__device__ __forceinline__ void dual_mul_body(const uint8_t* __restrict__ p1_enc, size_t p1_stride,
                                              const uint8_t* __restrict__ k1, const uint8_t* __restrict__ p2_enc,
                                              const uint8_t* __restrict__ k2, size_t k2_stride, int count,
                                              uint8_t* __restrict__ out_enc, uint8_t* __restrict__ ok) {
  const int xi = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = xi < count;
  const int x = live ? xi : count - 1;
  typename C::Point a, b, r;
  bool good = C::decode(a, p1_enc + (size_t)x * p1_stride);
  if (p2_enc) good = C::decode(b, p2_enc + (size_t)x * C::ENC_LEN) && good; else C::identity(b);
  dual_mul<C>(r, a, k1 + (size_t)x * 32, b, p2_enc ? k2 + (size_t)x * k2_stride : nullptr);
  if (live) {
    C::encode(out_enc + (size_t)x * C::ENC_LEN, r);
    ok[x] = good ? 1 : 0;
  }
}

template <class C>
__device__ __forceinline__ void add_body(const uint8_t* __restrict__ a_enc, const uint8_t* __restrict__ b_enc, int count,
                                         uint8_t* __restrict__ out_enc, uint8_t* __restrict__ ok) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= count) return;
  typename C::Point a, b, r;
  const bool g1 = C::decode(a, a_enc + (size_t)x * C::ENC_LEN);
  const bool g2 = C::decode(b, b_enc + (size_t)x * C::ENC_LEN);
  C::add(r, a, b);
  C::encode(out_enc + (size_t)x * C::ENC_LEN, r);
  ok[x] = (g1 && g2) ? 1 : 0;
}

// ---- forward differences for consecutive positions (additive version of modp_kernels.hip's) ---------------------
// X(i) = sum_j i^j C_j is a polynomial of degree t-1 in i, so with D_0 = X, D_k(i) = D_{k-1}(i+1) - D_{k-1}(i) the level
// D_{t-1} is constant and D_k(i+1) = D_k(i) + D_{k+1}(i): one point addition per share and coefficient instead of
// ~35 point operations for Horner's rule.  Negation is free, so there is no inversion step.  Chain c of `chains`
// owns the positions c, c+S, c+2S, ..; its seeds are t consecutive members in its middle (Horner), from which it is
// stepped both ways (forward with D_l = E_l[0], backward with H_l = (-1)^l E_l[t-1-l], same recurrence).
// One workgroup = one chain, one lane = one level; neighbours talk through LDS.  Results are the same group
// elements, hence the same canonical encodings.

// seeds: Horner as above, but the points stay in internal coordinates (array of structs)
template <class C>
__device__ __forceinline__ void seeds_body(const u32* __restrict__ cm, int t, const int64_t* __restrict__ positions,
                                           int count, u32* __restrict__ pts) {
  const int xi = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = xi < count;
  const int x = live ? xi : count - 1;
  const uint64_t pos = (uint64_t)positions[x];
  int nb = (pos == 0) ? 0 : 64 - __builtin_clzll(pos);
  typename C::Point acc, cj, r;
  load_point_aos<C>(acc, cm + (size_t)(t - 1) * C::POINT_WORDS);
  for (int j = t - 2; j >= 0; --j) {
    small_scalar_mul<C>(r, acc, pos, nb);
    load_point_aos<C>(cj, cm + (size_t)j * C::POINT_WORDS);
    C::add(acc, r, cj);
  }
  if (live) store_point_aos<C>(pts + (size_t)x * C::POINT_WORDS, acc);
}

template <class C>
__device__ __forceinline__ void lds_put_raw(u32* lds, int k, const typename C::Point& p) {
  const u32* w = reinterpret_cast<const u32*>(&p);
#pragma unroll
  for (int i = 0; i < C::POINT_WORDS; ++i) lds[i * blockDim.x + k] = w[i];
}
template <class C>
__device__ __forceinline__ void lds_get_raw(typename C::Point& p, const u32* lds, int k) {
  u32* w = reinterpret_cast<u32*>(&p);
#pragma unroll
  for (int i = 0; i < C::POINT_WORDS; ++i) w[i] = lds[i * blockDim.x + k];
}

// The same seeds with the polynomial cut into PARTS pieces evaluated by PARTS adjacent lanes:
//   X(x) = sum_p x^(lo_p) * Piece_p(x),   Piece_p(x) = sum_{j < len_p} x^j C_{lo_p + j}   (Horner, len_p = t/PARTS),
// the factor x^(lo_p) mod the group order by 256-bit Montgomery arithmetic, applied as one full-width scalar
// multiplication; the pieces are added up through LDS.  The sequential depth of a seed drops from (t-1) Horner steps
// to t/PARTS steps + one scalar multiplication (about five-fold for t = 256): the seed launch bounds a box's latency.
template <class C, class O, int PARTS>
__device__ __forceinline__ void seeds_split_body(const u32* __restrict__ cm, int t, const int64_t* __restrict__ positions,
                                                 int count, u32* __restrict__ pts) {
  extern __shared__ u32 lds[];
  const int gi = blockIdx.x * blockDim.x + threadIdx.x;
  const int xi = gi / PARTS, part = gi % PARTS;
  const bool live = xi < count;
  const int x = live ? xi : count - 1;
  const uint64_t pos = (uint64_t)positions[x];
  const int nb = (pos == 0) ? 0 : 64 - __builtin_clzll(pos);
  const int len = (t + PARTS - 1) / PARTS;
  const int lo = part * len;
  const int hi = (lo + len < t) ? lo + len : t;            // coefficients lo .. hi-1 (empty when lo >= t)
  typename C::Point acc, cj, r;
  C::identity(acc);
  for (int j = hi - 1; j >= lo; --j) {
    small_scalar_mul<C>(r, acc, pos, nb);
    load_point_aos<C>(cj, cm + (size_t)j * C::POINT_WORDS);
    C::add(acc, r, cj);
  }
  if (part > 0) {                                          // times x^lo
    Sc s;
    ScalarField<O>::pow_u64(s, pos, (u32)lo);
    limb_scalar_mul<C>(r, acc, s.v);
    acc = r;
  }
  // sum of the PARTS pieces: tree through LDS (pieces of one seed sit in adjacent lanes)
  const int k = threadIdx.x;
  for (int d = PARTS / 2; d >= 1; d >>= 1) {
    lds_put_raw<C>(lds, k, acc);
    __syncthreads();
    if (part < d) {
      lds_get_raw<C>(cj, lds, k + d);
      C::add(acc, acc, cj);
    }
    __syncthreads();
  }
  if (live && part == 0) store_point_aos<C>(pts + (size_t)x * C::POINT_WORDS, acc);
}

template <class C>
__device__ __forceinline__ void lds_put(u32* lds, int k, const typename C::Point& p) {
  const u32* w = reinterpret_cast<const u32*>(&p);
#pragma unroll
  for (int i = 0; i < C::POINT_WORDS; ++i) lds[i * blockDim.x + k] = w[i];     // word-major: conflict-free
}
template <class C>
__device__ __forceinline__ void lds_get(typename C::Point& p, const u32* lds, int k) {
  u32* w = reinterpret_cast<u32*>(&p);
#pragma unroll
  for (int i = 0; i < C::POINT_WORDS; ++i) w[i] = lds[i * blockDim.x + k];
}

// difference tables: E_l[k] = E_{l-1}[k+1] - E_{l-1}[k];  seeds of chain c are pts[c + chains k]
template <class C>
__device__ __forceinline__ void fd_table_body(const u32* __restrict__ seeds, int chains, int t, u32* __restrict__ fwd,
                                              u32* __restrict__ bwd) {
  extern __shared__ u32 lds[];
  const int chain = blockIdx.x, k = threadIdx.x;
  typename C::Point E, nb, neg;
  if (k < t) load_point_aos<C>(E, seeds + ((size_t)chain + (size_t)chains * k) * C::POINT_WORDS); else C::identity(E);
  u32* f = fwd + (size_t)chain * t * C::POINT_WORDS;
  u32* b = bwd + (size_t)chain * t * C::POINT_WORDS;
  if (k == 0) store_point_aos<C>(f, E);
  if (k == t - 1) store_point_aos<C>(b, E);
  for (int lvl = 1; lvl < t; ++lvl) {
    lds_put<C>(lds, k, E);
    __syncthreads();
    if (k <= t - 1 - lvl) {
      lds_get<C>(nb, lds, k + 1);
      C::neg(neg, E);
      C::add(E, nb, neg);
    }
    __syncthreads();
    if (k == 0) store_point_aos<C>(f + (size_t)lvl * C::POINT_WORDS, E);
    if (k == t - 1 - lvl) {
      if (lvl & 1) { C::neg(neg, E); store_point_aos<C>(b + (size_t)lvl * C::POINT_WORDS, neg); }
      else store_point_aos<C>(b + (size_t)lvl * C::POINT_WORDS, E);
    }
  }
}

// stepping: D_k <- D_k + D_{k+1}; block = (direction, chain); pts[c + chains j] = X at index j of chain c
template <class C>
__device__ __forceinline__ void fd_step_body(const u32* __restrict__ fwd, const u32* __restrict__ bwd, int chains, int t,
                                             int w0, int chain_len, int count, u32* __restrict__ pts) {
  extern __shared__ u32 lds[];
  const int dir = blockIdx.x / chains, chain = blockIdx.x % chains, k = threadIdx.x;
  if (dir == 1 && w0 == 0) return;
  const int steps = dir == 0 ? chain_len - 1 - w0 : w0 + t - 1;
  const u32* st = (dir == 0 ? fwd : bwd) + (size_t)chain * t * C::POINT_WORDS;
  typename C::Point D, nb;
  if (k < t) load_point_aos<C>(D, st + (size_t)k * C::POINT_WORDS); else C::identity(D);
  for (int step = 1; step <= steps; ++step) {
    lds_put<C>(lds, k, D);
    __syncthreads();
    if (k + 1 < t) {
      lds_get<C>(nb, lds, k + 1);
      C::add(D, D, nb);
    }
    __syncthreads();
    const int j = dir == 0 ? w0 + step : w0 + t - 1 - step;
    const size_t idx = (size_t)chain + (size_t)chains * j;
    if (k == 0 && step >= t && idx < (size_t)count) store_point_aos<C>(pts + idx * C::POINT_WORDS, D);
  }
}

// ---- stepping by a pipeline of single-wave stages, four lanes per level (the lone box's latency) ---------------------
// fd_step_body above is one workgroup per chain with one lane per level: every step is one complete addition in
// EVERY lane -- 12 (9) field products one after the other -- between two workgroup barriers, 6-7 us per step, and a
// lone box waits for 2 x 2300 of them.  Here a level is a QUAD of lanes (ec_quad.h: the independent products of the
// addition side by side), a wave holds 16 consecutive levels, and the t levels of a chain are a pipeline of t/16
// single-wave workgroups on different SIMDs, exactly as modp_kernels.hip's stepping kernels are: level k needs level
// k+1 of the previous step only, so a stage needs one point per step from the stage above and nothing from below.
// The point travels through HBM as self-validating words (bit 31 set on every limb word of a buffer zeroed before the
// launch; limbs are < 2^29), written and polled with agent-scope relaxed atomics; the entry of the next step is
// requested before this step's addition and checked after it.  Stages take their place in the chain from a ticket
// (the top stage first), so a stage only ever waits for a workgroup that has started.  A stage whose wait times out
// (or that is told to fail by the test hook) clears the box's gate -- the gated Horner launch then recomputes every
// X -- and poisons its output so that the stages below give up at once.
#ifndef EC_LONE_SETPRIO
#define EC_LONE_SETPRIO 3       // wave priority of the lone box's latency-bound launches (seeds, table and stepping pipelines)
#endif
constexpr u32 EC_HAND_VALID = 0x80000000u;
constexpr u32 EC_HAND_POISON = 0x40000000u;
constexpr u32 EC_HAND_LIMB = 0x3fffffffu;
constexpr long long EC_FD_TIMEOUT_TICKS = 200000000LL;      // 2 s of the 100 MHz wall clock
// levels per wave and words per handed point come from the lane layout: Q::LEVELS = 64 / Q::LANES, 10 words per lane

__device__ __forceinline__ void ec_hand_publish(u32* __restrict__ dst, const Fe& a, u32 tag) {
#pragma unroll
  for (int i = 0; i < 10; ++i) __hip_atomic_store(dst + i, a.v[i] | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void ec_hand_load(u32 (&v)[10], const u32* __restrict__ src) {
#pragma unroll
  for (int i = 0; i < 10; ++i) v[i] = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u32 ec_hand_state(const u32 (&v)[10]) {
  u32 all = 0xffffffffu, any = 0;
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    all &= v[i];
    any |= v[i];
  }
  return (any & EC_HAND_POISON) ? EC_HAND_POISON : (all & EC_HAND_VALID);
}
// Every quad of the wave asks for the same entry (lane role i: words 10 i .. 10 i + 9) -- no lane-dependent control flow
// around the DPP moves that follow (ec_quad.h); waits until every lane holds valid words.  Wave-uniform result.
__device__ __forceinline__ bool ec_hand_wait(u32 (&v)[10], const u32* __restrict__ src) {
  const long long t0 = wall_clock64();
  while (true) {
    const u32 st = ec_hand_state(v);
    if (__builtin_amdgcn_ballot_w64(st == EC_HAND_VALID) == ~0ull) return true;
    if (__builtin_amdgcn_ballot_w64(st == EC_HAND_POISON) != 0 || wall_clock64() - t0 > EC_FD_TIMEOUT_TICKS) return false;
    __builtin_amdgcn_s_sleep(2);
    ec_hand_load(v, src);
  }
}

// words of handoff space one box needs for a stepping launch (tickets first)
__host__ __device__ inline size_t ec_quad_hand_words(int t, int chains, int max_steps, int lanes) {
  const int levels = 64 / lanes;
  const size_t nst = (size_t)(t + levels - 1) / levels;
  const size_t tickets = ((size_t)2 * chains + 63) / 64 * 64;
  return tickets + (size_t)2 * chains * nst * (size_t)(max_steps + 1) * (size_t)(lanes * 10);
}

template <class Q>
__device__ __forceinline__ void fd_quad_step_body(const u32* __restrict__ fwd, const u32* __restrict__ bwd, int chains, int t,
                                                  int w0, int chain_len, int count, u32* __restrict__ pts, u32* __restrict__ hand,
                                                  int max_steps, int* __restrict__ gate, int inject_fault) {
  typedef typename Q::C C;
  extern __shared__ u32 lds[];
  constexpr int PW = C::POINT_WORDS;
  __builtin_amdgcn_s_setprio(EC_LONE_SETPRIO);    // latency-critical and few: issue ahead of the wide kernels sharing the SIMD (a2's)
  constexpr int EC_QUAD_LEVELS = Q::LEVELS, EC_HAND_ENTRY = Q::LANES * 10;
  const int nst = (t + EC_QUAD_LEVELS - 1) / EC_QUAD_LEVELS;
  const int cd = blockIdx.x / nst;
  const int dir = cd / chains, chain = cd % chains;
  if (dir == 1 && w0 == 0) return;
  int* tickets = reinterpret_cast<int*>(hand);
  u32* entries = hand + ((size_t)2 * chains + 63) / 64 * 64;
  int ticket = 0;
  if (threadIdx.x == 0) ticket = atomicAdd(tickets + cd, 1);
  const int sidx = __builtin_amdgcn_readfirstlane(ticket);           // 0 = the top levels
  const int kbase = (nst - 1 - sidx) * EC_QUAD_LEVELS;
  const int lane = threadIdx.x, quad = lane / Q::LANES, role = lane % Q::LANES;
  const int k = kbase + quad;
  const bool has_up = sidx > 0, has_down = kbase > 0;
  const bool reader = quad == EC_QUAD_LEVELS - 1;
  const int steps = dir == 0 ? chain_len - 1 - w0 : w0 + t - 1;
  const u32* st = (dir == 0 ? fwd : bwd) + (size_t)chain * t * PW;
  const size_t stage_words = (size_t)(max_steps + 1) * EC_HAND_ENTRY;
  u32* mine = entries + ((size_t)cd * nst + sidx) * stage_words + role * 10;
  const u32* up = entries + ((size_t)cd * nst + (has_up ? sidx - 1 : 0)) * stage_words + role * 10;
  typename Q::St D, top;
  if (k < t) Q::load(D, st + (size_t)k * PW, role); else Q::identity(D, role);
  // the level above this wave's: its initial value comes from the table, the later ones from the stage above
  if (has_up && kbase + EC_QUAD_LEVELS < t) Q::load(top, st + (size_t)(kbase + EC_QUAD_LEVELS) * PW, role); else Q::identity(top, role);
  u32 pre[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) pre[i] = 0;
  if (has_up && steps >= 2) ec_hand_load(pre, up + (size_t)1 * EC_HAND_ENTRY);
  for (int step = 1; step <= steps; ++step) {
    typename Q::St nb;
    Fe prim;
    if (has_up && step >= 2) {                                       // entry step-1 of the stage above
      bool ok = ec_hand_wait(pre, up + (size_t)(step - 1) * EC_HAND_ENTRY);
      if (inject_fault == 1 && cd == 0 && sidx == 1 && step == 3) ok = false;
      if (!ok) {
        if (threadIdx.x == 0) __hip_atomic_store(gate + blockIdx.y, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (has_down && quad == 0) {
          Fe z;
          Secp::Fp::zero(z);
          ec_hand_publish(mine + (size_t)step * EC_HAND_ENTRY, z, EC_HAND_POISON);
        }
        return;
      }
#pragma unroll
      for (int i = 0; i < 10; ++i) prim.v[i] = pre[i] & EC_HAND_LIMB;
      Q::nb_from_primary(top, prim, role);
      if (step < steps) ec_hand_load(pre, up + (size_t)step * EC_HAND_ENTRY);     // for the next step, under this one's addition
    }
    fe_from_next_group<Q::LANES>(prim, Q::primary(D));
    Q::nb_from_primary(nb, prim, role);
    Q::select(nb, top, reader);
    Q::add(D, nb, role, lds);
    if (has_down) {
      if (quad == 0) ec_hand_publish(mine + (size_t)step * EC_HAND_ENTRY, Q::primary(D), EC_HAND_VALID);
    } else if (step >= t) {
      const int j = dir == 0 ? w0 + step : w0 + t - 1 - step;
      const size_t idx = (size_t)chain + (size_t)chains * j;
      Fe o;
      Q::out_words(o, D, role);
      if (quad == 0 && Q::out_lane(role) && idx < (size_t)count) {
        u32* dst = pts + idx * PW + Q::out_offset(role);
#pragma unroll
        for (int i = 0; i < 10; ++i) dst[i] = o.v[i];
      }
    }
  }
}

// The difference tables by the same pipeline: E_l[k] = E_{l-1}[k+1] - E_{l-1}[k] has the neighbour structure of the
// stepping recurrence (element k needs element k+1 of the previous level), so a stage of 16 elements needs one point
// per level from the stage above while its top element is still in the triangle (k <= t-1-l).  fwd[l] = E_l[0] leaves
// from the bottom stage, bwd[l] = (-1)^l E_l[t-1-l] from whichever stage holds element t-1-l.
template <class Q>
__device__ __forceinline__ void fd_quad_table_body(const u32* __restrict__ seeds, int chains, int t, u32* __restrict__ fwd,
                                                   u32* __restrict__ bwd, u32* __restrict__ hand, int* __restrict__ gate,
                                                   int inject_fault) {
  typedef typename Q::C C;
  extern __shared__ u32 lds[];
  constexpr int PW = C::POINT_WORDS;
  __builtin_amdgcn_s_setprio(EC_LONE_SETPRIO);
  constexpr int EC_QUAD_LEVELS = Q::LEVELS, EC_HAND_ENTRY = Q::LANES * 10;
  const int nst = (t + EC_QUAD_LEVELS - 1) / EC_QUAD_LEVELS;
  const int chain = blockIdx.x / nst;
  int* tickets = reinterpret_cast<int*>(hand);
  u32* entries = hand + ((size_t)2 * chains + 63) / 64 * 64;
  int ticket = 0;
  if (threadIdx.x == 0) ticket = atomicAdd(tickets + chain, 1);
  const int sidx = __builtin_amdgcn_readfirstlane(ticket);           // 0 = the top elements
  const int kbase = (nst - 1 - sidx) * EC_QUAD_LEVELS;
  const int lane = threadIdx.x, quad = lane / Q::LANES, role = lane % Q::LANES;
  const int k = kbase + quad;
  const bool has_up = sidx > 0, has_down = kbase > 0;
  const bool reader = quad == EC_QUAD_LEVELS - 1;
  const size_t stage_words = (size_t)(t + 1) * EC_HAND_ENTRY;
  u32* mine = entries + ((size_t)chain * nst + sidx) * stage_words + role * 10;
  const u32* up = entries + ((size_t)chain * nst + (has_up ? sidx - 1 : 0)) * stage_words + role * 10;
  u32* f = fwd + (size_t)chain * t * PW;
  u32* b = bwd + (size_t)chain * t * PW;
  typename Q::St E, top;
  if (k < t) Q::load(E, seeds + ((size_t)chain + (size_t)chains * k) * PW, role); else Q::identity(E, role);
  if (has_up && kbase + EC_QUAD_LEVELS < t) Q::load(top, seeds + ((size_t)chain + (size_t)chains * (kbase + EC_QUAD_LEVELS)) * PW, role);
  else Q::identity(top, role);
  auto emit = [&](int lvl) {                                          // level lvl is in E
    const int kb = t - 1 - lvl;                                       // the element that leaves backward
    if (kbase == 0 || (kb >= kbase && kb < kbase + EC_QUAD_LEVELS)) {
      Fe o;
      Q::out_words(o, E, role);
      if (kbase == 0 && quad == 0 && Q::out_lane(role)) {
        u32* dst = f + (size_t)lvl * PW + Q::out_offset(role);
#pragma unroll
        for (int i = 0; i < 10; ++i) dst[i] = o.v[i];
      }
      if (lvl & 1) {
        typename Q::St n = E;
        Q::neg(n, role);
        Q::out_words(o, n, role);
      }
      if (k == kb && Q::out_lane(role)) {
        u32* dst = b + (size_t)lvl * PW + Q::out_offset(role);
#pragma unroll
        for (int i = 0; i < 10; ++i) dst[i] = o.v[i];
      }
    }
  };
  emit(0);
  const int last_lvl = t - 1 - kbase;                                 // element k is in level l while k <= t-1-l
  u32 pre[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) pre[i] = 0;
  const auto need_up = [&](int lvl) { return has_up && lvl >= 2 && kbase + EC_QUAD_LEVELS - 1 <= t - 1 - lvl; };
  if (need_up(2)) ec_hand_load(pre, up + (size_t)1 * EC_HAND_ENTRY);
  for (int lvl = 1; lvl <= last_lvl; ++lvl) {
    typename Q::St nb, neg;
    Fe prim;
    if (need_up(lvl)) {                                               // level lvl-1 of the first element of the stage above
      bool ok = ec_hand_wait(pre, up + (size_t)(lvl - 1) * EC_HAND_ENTRY);
      if (inject_fault == 2 && chain == 0 && sidx == 1 && lvl == 3) ok = false;
      if (!ok) {
        if (threadIdx.x == 0) __hip_atomic_store(gate + blockIdx.y, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (has_down && quad == 0) {
          Fe z;
          Secp::Fp::zero(z);
          ec_hand_publish(mine + (size_t)lvl * EC_HAND_ENTRY, z, EC_HAND_POISON);
        }
        return;
      }
#pragma unroll
      for (int i = 0; i < 10; ++i) prim.v[i] = pre[i] & EC_HAND_LIMB;
      Q::nb_from_primary(top, prim, role);
      if (need_up(lvl + 1)) ec_hand_load(pre, up + (size_t)lvl * EC_HAND_ENTRY);
    }
    fe_from_next_group<Q::LANES>(prim, Q::primary(E));
    Q::nb_from_primary(nb, prim, role);
    Q::select(nb, top, reader);
    neg = E;
    Q::neg(neg, role);
    Q::add(neg, nb, role, lds);                                       // E_{l-1}[k+1] - E_{l-1}[k]
    Q::select(E, neg, k <= t - 1 - lvl);
    if (has_down && quad == 0) ec_hand_publish(mine + (size_t)lvl * EC_HAND_ENTRY, Q::primary(E), EC_HAND_VALID);
    emit(lvl);
  }
}

template <class C>
__device__ __forceinline__ void encode_body(const u32* __restrict__ pts, int count, uint8_t* __restrict__ enc) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= count) return;
  typename C::Point p;
  load_point_aos<C>(p, pts + (size_t)x * C::POINT_WORDS);
  C::encode(enc + (size_t)x * C::ENC_LEN, p);
}


// ---- windowed double-scalar multiplication (signed 4-bit Straus) ---------------------------------------------------
// out = k1 * P1 + k2 * P2   (dleq.rs:66-84; Group::exp is the k2-less case).  k256 and curve25519-dalek, which the
// reference delegates to (secp256k1.rs:91-100, ristretto255.rs:161-170), use 4-bit signed windows as well.
//   * a per-share base comes with a table of its multiples P .. 8P in cached form, built once per base by
//     k_*_build_tables into HBM (960 / 1280 B per share, array of structs: a lane reads ITS entry |d| contiguously);
//   * the group generator comes as a fixed-base comb  comb[w][i] = (i+1) 16^w G  (65 windows x 8 affine entries,
//     packed canonical coordinates, 33 KB / 49 KB) staged in LDS by every workgroup: r G costs 64 mixed additions
//     and NO doublings;
//   * signed digits d_w = nibble_w(k + 0x88..8) - 8 need no carry between windows; the recoded scalars sit in LDS
//     (word-major across the lanes), the loop fetches one word per eight windows.
// One share per lane, 256 lanes per workgroup; 256 doublings + 128 (two tables) or 64 + 64 (comb + table) additions
// instead of 256 + 256.  The result leaves in internal coordinates (array of structs) for the encoding kernels.
constexpr int DW_THREADS = 256;

template <class C>
__device__ __forceinline__ void store_cached(u32* __restrict__ dst, const typename C::Cached& e) {
  const u32* w = reinterpret_cast<const u32*>(&e);
#pragma unroll
  for (int i = 0; i < C::CACHED_WORDS; ++i) dst[i] = w[i];
}
template <class C>
__device__ __forceinline__ void load_cached(typename C::Cached& e, const u32* __restrict__ src) {
  u32* w = reinterpret_cast<u32*>(&e);
#pragma unroll
  for (int i = 0; i < C::CACHED_WORDS; ++i) w[i] = src[i];
}

// tab[x][i] = (i + 1) * P_x, i < 8.  P_x from its encoding (enc != null: decoded here, ok[x] = validity) or from
// internal coordinates (pts).
template <class C>
__device__ __forceinline__ void build_tables_body(const uint8_t* __restrict__ enc, size_t enc_stride,
                                                  const u32* __restrict__ pts, int count, u32* __restrict__ tab,
                                                  uint8_t* __restrict__ ok, const int* __restrict__ gate) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= count) return;
  typename C::Point p;
  // both sources given: the internal points are valid when the forward-difference path ran (*gate == 1)
  const bool from_enc = enc != nullptr && (pts == nullptr || (gate != nullptr && *gate != 1));
  if (from_enc) {
    const bool good = C::decode(p, enc + (size_t)x * enc_stride);
    if (ok != nullptr) ok[x] = good ? 1 : 0;
  } else {
    load_point_aos<C>(p, pts + (size_t)x * C::POINT_WORDS);
  }
  u32* mine = tab + (size_t)x * 8 * C::CACHED_WORDS;
  build_cached_table_streamed<C>(p, [&](int i, const typename C::Cached& e) { store_cached<C>(mine + i * C::CACHED_WORDS, e); });
}

// comb[w][i] = (i + 1) * 16^w * G, packed affine -- once per context and group
template <class C>
__device__ __forceinline__ void comb_build_body(u32* __restrict__ comb) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= 65) return;
  typename C::Point base, m;
  C::generator(base);
  for (int i = 0; i < 4 * w; ++i) C::dbl(base, base);
  m = base;
  for (int i = 0; i < 8; ++i) {
    typename C::Affine a;
    C::to_affine(a, m);
    C::pack_affine(comb + (size_t)(w * 8 + i) * C::AFFINE_PACKED_WORDS, a);
    C::add(m, m, base);
  }
}

#ifndef EC_GLV
#define EC_GLV 1          // secp256k1: table-based scalars are split by the GLV endomorphism (ec_glv.h): 128 doublings instead of 256
#endif
[[maybe_unused]] constexpr int DW_K_ROWS = 20;      // LDS rows of recoded scalars per lane: 9 (65 windows, the comb's scalar) + 10 (two GLV halves), or 2 x 10

// secp256k1 with the endomorphism: every scalar that multiplies a per-share table is split k = k1 + k2 lambda, |k1|, |k2| < 2^128,
// and the loop runs 33 signed 4-bit windows over up to four (table, half) pairs -- the entry of phi(P) is P's with X times beta.
// The generator's scalar (comb != null) keeps its 65 windows: the comb has no doublings to save.
__device__ __forceinline__ void dual_win_glv_body(const u32* __restrict__ comb, const u32* __restrict__ tab1,
                                                  const uint8_t* __restrict__ k1, size_t k1_stride, const u32* __restrict__ tab2,
                                                  const uint8_t* __restrict__ k2, size_t k2_stride, int count,
                                                  u32* __restrict__ out_pts, u32* lds) {
  typedef Secp C;
  constexpr int AW = C::AFFINE_PACKED_WORDS;
  u32* lds_comb = lds;
  u32* lds_k = lds + (comb != nullptr ? 65 * 8 * AW : 0);        // [DW_K_ROWS][DW_THREADS]
  const int xi = blockIdx.x * DW_THREADS + threadIdx.x;
  const bool live = xi < count;
  const int x = live ? xi : count - 1;
  if (comb != nullptr) {
    for (int i = threadIdx.x; i < 65 * 8 * AW; i += DW_THREADS) lds_comb[i] = comb[i];
  }
  // tables and their scalars: with a comb for the first scalar the (single) table belongs to the SECOND scalar
  const u32* t1 = tab1 != nullptr ? tab1 + (size_t)x * 8 * C::CACHED_WORDS : nullptr;
  const u32* t2 = tab2 != nullptr ? tab2 + (size_t)x * 8 * C::CACHED_WORDS : nullptr;
  const u32* tabs[2] = {t1 != nullptr ? t1 : t2, t2};
  const uint8_t* tks[2] = {t1 != nullptr ? k1 + (size_t)x * k1_stride : k2 + (size_t)x * k2_stride, k2 + (size_t)x * k2_stride};
  const int ntab = (t1 != nullptr && t2 != nullptr) ? 2 : ((t1 != nullptr || t2 != nullptr) ? 1 : 0);
  const int glv_row0 = comb != nullptr ? 9 : 0;
  {
    if (comb != nullptr) {
      u32 kp[9];
      recode_signed4<C>(kp, k1 + (size_t)x * k1_stride);
#pragma unroll
      for (int j = 0; j < 9; ++j) lds_k[j * DW_THREADS + threadIdx.x] = kp[j];
    }
#pragma unroll 1
    for (int b = 0; b < ntab; ++b) {
      u32 kw[8];
      scalar_words<C>(kw, tks[b]);
      GlvHalf h[2];
      secp_glv_split(h, kw);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int i = 0; i < 5; ++i)
          lds_k[(glv_row0 + 10 * b + 5 * j + i) * DW_THREADS + threadIdx.x] = h[j].kp[i] | ((i == 4 && h[j].neg) ? 0x80000000u : 0u);
      }
    }
  }
  __syncthreads();
  typename C::Point acc;
  C::identity(acc);
  if (ntab > 0) {
    u32 words[4] = {0, 0, 0, 0};
    u32 negs = 0;                      // bit s: half s = 2 b + j is negative
    for (int s = 0; s < 2 * ntab; ++s)
      negs |= (lds_k[(glv_row0 + 5 * s + 4) * DW_THREADS + threadIdx.x] >> 31) << s;
    for (int w = 32; w >= 0; --w) {
      if ((w & 7) == 7 || w == 32) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
          if (s < 2 * ntab) words[s] = lds_k[(glv_row0 + 5 * s + (w >> 3)) * DW_THREADS + threadIdx.x];
      }
      if (w != 32) {
#pragma unroll 1
        for (int i = 0; i < 4; ++i) C::dbl(acc, acc);
      }
      // ONE addition site for the (up to) four table halves: the loop body has to stay inside the instruction cache
#pragma unroll 1
      for (int s = 0; s < 2 * ntab; ++s) {
        const u32* tb = tabs[s >> 1];
        const int nib = (int)((words[s] >> (4 * (w & 7))) & 15u);
        int d = w == 32 ? nib : nib - 8;
        if ((negs >> s) & 1u) d = -d;
        add_signed_digit<C>(acc, d, [&](typename C::Cached& e, int i) {
          load_cached<C>(e, tb + i * C::CACHED_WORDS);
          if (s & 1) secp_phi_cached(e);
        });
      }
    }
  }
  if (comb != nullptr) {     // k1 * G from the comb in LDS: one mixed addition per window, no doublings
    typename C::Point g;
    C::identity(g);
    u32 w1 = 0;
    for (int w = 0; w < 65; ++w) {
      if ((w & 7) == 0) w1 = lds_k[(w >> 3) * DW_THREADS + threadIdx.x];
      add_signed_digit_affine<C>(g, signed_digit4(w1, w),
                                 [&](typename C::Affine& e, int i) { C::unpack_affine(e, lds_comb + (w * 8 + i) * AW); });
    }
    C::add(acc, acc, g);
  }
  if (live) store_point_aos<C>(out_pts + (size_t)x * C::POINT_WORDS, acc);
}

template <class C>
__device__ __forceinline__ void dual_win_body(const u32* __restrict__ comb, const u32* __restrict__ tab1,
                                              const uint8_t* __restrict__ k1, size_t k1_stride, const u32* __restrict__ tab2,
                                              const uint8_t* __restrict__ k2, size_t k2_stride, int count,
                                              u32* __restrict__ out_pts, u32* lds) {
#if EC_GLV
  if constexpr (C::SCALAR_BIG_ENDIAN) {                          // (secp256k1)
    dual_win_glv_body(comb, tab1, k1, k1_stride, tab2, k2, k2_stride, count, out_pts, lds);
    return;
  }
#endif
  constexpr int AW = C::AFFINE_PACKED_WORDS;
  u32* lds_comb = lds;                                           // [65 * 8 * AW]        (only when comb != null)
  u32* lds_k = lds + (comb != nullptr ? 65 * 8 * AW : 0);        // [2][9][DW_THREADS]   recoded scalars
  const int xi = blockIdx.x * DW_THREADS + threadIdx.x;
  const bool live = xi < count;
  const int x = live ? xi : count - 1;
  if (comb != nullptr) {
    for (int i = threadIdx.x; i < 65 * 8 * AW; i += DW_THREADS) lds_comb[i] = comb[i];
  }
  {
    u32 kp[9];
    recode_signed4<C>(kp, k1 + (size_t)x * k1_stride);
#pragma unroll
    for (int j = 0; j < 9; ++j) lds_k[j * DW_THREADS + threadIdx.x] = kp[j];
    if (tab2 != nullptr) {
      recode_signed4<C>(kp, k2 + (size_t)x * k2_stride);
#pragma unroll
      for (int j = 0; j < 9; ++j) lds_k[(9 + j) * DW_THREADS + threadIdx.x] = kp[j];
    }
  }
  __syncthreads();
  const u32* t1 = tab1 != nullptr ? tab1 + (size_t)x * 8 * C::CACHED_WORDS : nullptr;
  const u32* t2 = tab2 != nullptr ? tab2 + (size_t)x * 8 * C::CACHED_WORDS : nullptr;
  typename C::Point acc;
  C::identity(acc);
  if (t1 != nullptr || t2 != nullptr) {
    // the table(s) and the LDS rows of their recoded scalars; with a comb for the first scalar the (single) table
    // belongs to the SECOND scalar
    const u32* tabs[2] = {t1 != nullptr ? t1 : t2, t2};
    const int rows[2] = {(t1 != nullptr) ? 0 : 9, 9};
    const int ntab = (t1 != nullptr && t2 != nullptr) ? 2 : 1;
    u32 words[2] = {0, 0};
    for (int w = 64; w >= 0; --w) {
      if ((w & 7) == 7 || w == 64) {
        words[0] = lds_k[(rows[0] + (w >> 3)) * DW_THREADS + threadIdx.x];
        words[1] = lds_k[(rows[1] + (w >> 3)) * DW_THREADS + threadIdx.x];
      }
      if (w != 64) {
#pragma unroll 1
        for (int i = 0; i < 4; ++i) C::dbl(acc, acc);
      }
      // ONE addition site for both tables: the loop body (a doubling and an addition, about 50 KB of code for
      // secp256k1) has to stay inside the 64 KB instruction cache
#pragma unroll 1
      for (int b = 0; b < ntab; ++b) {
        const u32* tb = tabs[b];
        add_signed_digit<C>(acc, signed_digit4(words[b], w), [&](typename C::Cached& e, int i) { load_cached<C>(e, tb + i * C::CACHED_WORDS); });
      }
    }
  }
  if (comb != nullptr) {     // k1 * G from the comb in LDS: one mixed addition per window, no doublings
    typename C::Point g;
    C::identity(g);
    u32 w1 = 0;
    for (int w = 0; w < 65; ++w) {
      if ((w & 7) == 0) w1 = lds_k[(w >> 3) * DW_THREADS + threadIdx.x];
      add_signed_digit_affine<C>(g, signed_digit4(w1, w),
                                 [&](typename C::Affine& e, int i) { C::unpack_affine(e, lds_comb + (w * 8 + i) * AW); });
    }
    C::add(acc, acc, g);
  }
  if (live) store_point_aos<C>(out_pts + (size_t)x * C::POINT_WORDS, acc);
}

// ---- seeds for a box that has the chip to itself --------------------------------------------------------------------
// seeds_split_body with the piece's factor x^lo applied through signed 4-bit windows (and secp256k1's endomorphism)
// instead of 256 doublings and 256 additions: the lane's table P .. 8P goes to its own 8 entries of `wtab` (HBM, read
// back through L2: 65 or 2 x 33 look-ups against 128 / 256 doublings), the recoded scalar to LDS.  With PARTS = 16 the
// sequential depth of a seed is t/16 Horner steps + one windowed multiplication: about 0.45 of seeds_split_body's.
[[maybe_unused]] constexpr int SEEDS_WIN_K_ROWS = 10;
template <class C, class O, int PARTS>
__device__ __forceinline__ void seeds_win_body(const u32* __restrict__ cm, int t, const int64_t* __restrict__ positions,
                                               int count, u32* __restrict__ pts, u32* __restrict__ wtab) {
  extern __shared__ u32 lds[];
  constexpr int CW = C::CACHED_WORDS;
  __builtin_amdgcn_s_setprio(EC_LONE_SETPRIO);
  u32* lds_k = lds + 64 * C::POINT_WORDS;                  // [SEEDS_WIN_K_ROWS][64]
  const int gi = blockIdx.x * blockDim.x + threadIdx.x;
  const int xi = gi / PARTS, part = gi % PARTS;
  const bool live = xi < count;
  const int x = live ? xi : count - 1;
  const uint64_t pos = (uint64_t)positions[x];
  const int len = (t + PARTS - 1) / PARTS;
  const int lo = part * len;
  const int hi = (lo + len < t) ? lo + len : t;
  typename C::Point acc, cj, r;
  C::identity(acc);
  for (int j = hi - 1; j >= lo; --j) {
    small_scalar_mul_naf<C>(r, acc, pos);
    load_point_aos<C>(cj, cm + (size_t)j * C::POINT_WORDS);
    C::add(acc, r, cj);
  }
  {                                                        // times x^lo (1 for the first piece: the same code, a few wasted additions of the identity)
    Sc sc;
    ScalarField<O>::pow_u64(sc, pos, (u32)lo);
    u32* mine = wtab + (size_t)gi * 8 * CW;
    typename C::Cached e;
    cj = acc;
#pragma unroll 1
    for (int i = 0; i < 8; ++i) {
      C::to_cached(e, cj);
      store_cached<C>(mine + i * CW, e);
      if (i < 7) C::add(cj, cj, acc);
    }
    __threadfence_block();
    const int k = threadIdx.x;
    C::identity(r);
#if EC_GLV
    if constexpr (C::SCALAR_BIG_ENDIAN) {
      GlvHalf h[2];
      secp_glv_split(h, sc.v);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 5; ++i) lds_k[(5 * j + i) * 64 + k] = h[j].kp[i];
      u32 words[2] = {0, 0};
      const u32 negs = (h[0].neg ? 1u : 0u) | (h[1].neg ? 2u : 0u);
      for (int w = 32; w >= 0; --w) {
        if ((w & 7) == 7 || w == 32) {
          words[0] = lds_k[(w >> 3) * 64 + k];
          words[1] = lds_k[(5 + (w >> 3)) * 64 + k];
        }
        if (w != 32) {
#pragma unroll 1
          for (int i = 0; i < 4; ++i) C::dbl(r, r);
        }
#pragma unroll 1
        for (int s2 = 0; s2 < 2; ++s2) {
          const int nib = (int)(((s2 ? words[1] : words[0]) >> (4 * (w & 7))) & 15u);
          int d = w == 32 ? nib : nib - 8;
          if ((negs >> s2) & 1u) d = -d;
          add_signed_digit<C>(r, d, [&](typename C::Cached& ce, int i) {
            load_cached<C>(ce, mine + i * CW);
            if (s2 & 1) secp_phi_cached(ce);
          });
        }
      }
    } else
#endif
    {
      u64 carry = 0;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        carry += (u64)sc.v[i] + 0x88888888u;
        lds_k[i * 64 + k] = (u32)carry;
        carry >>= 32;
      }
      lds_k[8 * 64 + k] = (u32)carry;
      u32 word = 0;
      for (int w = 64; w >= 0; --w) {
        if ((w & 7) == 7 || w == 64) word = lds_k[(w >> 3) * 64 + k];
        if (w != 64) {
#pragma unroll 1
          for (int i = 0; i < 4; ++i) C::dbl(r, r);
        }
        add_signed_digit<C>(r, signed_digit4(word, w), [&](typename C::Cached& ce, int i) { load_cached<C>(ce, mine + i * CW); });
      }
    }
    acc = r;
  }
  // sum of the PARTS pieces: tree through LDS (pieces of one seed sit in adjacent lanes)
  const int k = threadIdx.x;
  for (int d = PARTS / 2; d >= 1; d >>= 1) {
    lds_put_raw<C>(lds, k, acc);
    __syncthreads();
    if (part < d) {
      lds_get_raw<C>(cj, lds, k + d);
      C::add(acc, acc, cj);
    }
    __syncthreads();
  }
  if (live && part == 0) store_point_aos<C>(pts + (size_t)x * C::POINT_WORDS, acc);
}

// out = sum of m points (internal coordinates): the fold of reconstruct's Lagrange factors (participant.rs:1489-1492).
// One workgroup: every lane adds up a strided subset, then a tree through LDS.
template <class C>
__device__ __forceinline__ void sum_points_body(const u32* __restrict__ pts, int m, u32* __restrict__ out, u32* lds) {
  const int k = threadIdx.x;
  typename C::Point acc, p;
  C::identity(acc);
  for (int i = k; i < m; i += blockDim.x) {
    load_point_aos<C>(p, pts + (size_t)i * C::POINT_WORDS);
    C::add(acc, acc, p);
  }
  for (int d = blockDim.x / 2; d >= 1; d >>= 1) {
    lds_put_raw<C>(lds, k, acc);
    __syncthreads();
    if (k < d) {
      lds_get_raw<C>(p, lds, k + d);
      C::add(acc, acc, p);
    }
    __syncthreads();
  }
  if (k == 0) store_point_aos<C>(out, acc);
}

// secp256k1: eight points per lane share one field inversion (Montgomery's trick)
__device__ __forceinline__ void secp_encode_batch_body(const u32* __restrict__ pts, int count, uint8_t* __restrict__ enc) {
  constexpr int B = 8;
  const int first = (blockIdx.x * blockDim.x + threadIdx.x) * B;
  if (first >= count) return;
  Secp::encode_batch<B>(
      enc + (size_t)first * 33, 33,
      [&](int i, Secp::Point& p) {
        const int idx = first + i < count ? first + i : count - 1;
        load_point_aos<Secp>(p, pts + (size_t)idx * Secp::POINT_WORDS);
      },
      [&](int i) { return first + i < count; });
}

}  // namespace

// The file is compiled twice (Makefile): EC_PART 1 = the secp256k1 kernels and every launcher, EC_PART 2 = the
// ristretto255 kernels.  The launchers only need the other part's kernels declared (BODY = EC_DECL).
#ifndef EC_PART
#define EC_PART 0
#endif
#define EC_DEF(...) { __VA_ARGS__ }
#define EC_DECL(...) ;
#define EC_KERNELS(NAME, CURVE, ORDER, QUAD, BODY)                                                                     \
  extern "C" __global__ void __launch_bounds__(64) k_##NAME##_decode(const uint8_t* enc, int count, u32* pts,          \
                                                                     uint8_t* ok)                                      \
      BODY(decode_body<CURVE>(enc, count, pts, ok);)                                                                   \
  extern "C" __global__ void __launch_bounds__(64) k_##NAME##_commit_eval(const u32* cm, int t, const int64_t* pos,    \
                                                                          int count, uint8_t* x_enc, const int* gate,  \
                                                                          int want, BoxStride bs)                      \
      BODY(EC_BOX_GATE_CHECK(gate, want);                                                                              \
           commit_eval_body<CURVE>(cm + blockIdx.y * bs.cm, t, pos + blockIdx.y * bs.pos, count,                       \
                                   x_enc + blockIdx.y * bs.enc);)                                                      \
  extern "C" __global__ void __launch_bounds__(64) k_##NAME##_dual_mul(                                                \
      const uint8_t* p1, size_t p1_stride, const uint8_t* k1, const uint8_t* p2, const uint8_t* k2, size_t k2_stride,  \
      int count, uint8_t* out, uint8_t* ok)                                                                            \
      BODY(dual_mul_body<CURVE>(p1, p1_stride, k1, p2, k2, k2_stride, count, out, ok);)                                \
  extern "C" __global__ void __launch_bounds__(64) k_##NAME##_add(const uint8_t* a, const uint8_t* b, int count,       \
                                                                  uint8_t* out, uint8_t* ok)                           \
      BODY(add_body<CURVE>(a, b, count, out, ok);)                                                                     \
  extern "C" __global__ void __launch_bounds__(64) k_##NAME##_fd_seeds(const u32* cm, int t, const int64_t* pos,       \
                                                                       int count, u32* pts, const int* gate,           \
                                                                       BoxStride bs)                                   \
      BODY(EC_BOX_GATE_CHECK(gate, 1);                                                                                 \
           seeds_body<CURVE>(cm + blockIdx.y * bs.cm, t, pos + blockIdx.y * bs.pos, count, pts + blockIdx.y * bs.pts);) \
  extern "C" __global__ void __launch_bounds__(64) k_##NAME##_fd_seeds_split(const u32* cm, int t, const int64_t* pos, \
                                                                             int count, u32* pts, const int* gate,     \
                                                                             BoxStride bs)                             \
      BODY(EC_BOX_GATE_CHECK(gate, 1);                                                                                 \
           seeds_split_body<CURVE, ORDER, 8>(cm + blockIdx.y * bs.cm, t, pos + blockIdx.y * bs.pos, count,             \
                                             pts + blockIdx.y * bs.pts);)                                              \
  extern "C" __global__ void __launch_bounds__(64) k_##NAME##_fd_seeds_win(const u32* cm, int t, const int64_t* pos,   \
                                                                           int count, u32* pts, u32* wtab,             \
                                                                           size_t wtab_box, const int* gate,           \
                                                                           BoxStride bs)                               \
      BODY(EC_BOX_GATE_CHECK(gate, 1);                                                                                 \
           seeds_win_body<CURVE, ORDER, 16>(cm + blockIdx.y * bs.cm, t, pos + blockIdx.y * bs.pos, count,              \
                                            pts + blockIdx.y * bs.pts, wtab + blockIdx.y * wtab_box);)                 \
  extern "C" __global__ void __launch_bounds__(512) k_##NAME##_fd_table(const u32* seeds, int chains, int t, u32* fwd, \
                                                                        u32* bwd, const int* gate, BoxStride bs)       \
      BODY(EC_BOX_GATE_CHECK(gate, 1);                                                                                 \
           fd_table_body<CURVE>(seeds + blockIdx.y * bs.pts, chains, t, fwd + blockIdx.y * bs.state,                   \
                                bwd + blockIdx.y * bs.state);)                                                         \
  extern "C" __global__ void __launch_bounds__(512) k_##NAME##_fd_step(const u32* fwd, const u32* bwd, int chains,     \
                                                                       int t, int w0, int chain_len, int count,        \
                                                                       u32* pts, const int* gate, BoxStride bs)        \
      BODY(EC_BOX_GATE_CHECK(gate, 1);                                                                                 \
           fd_step_body<CURVE>(fwd + blockIdx.y * bs.state, bwd + blockIdx.y * bs.state, chains, t, w0, chain_len,     \
                               count, pts + blockIdx.y * bs.pts);)                                                     \
  extern "C" __global__ void __launch_bounds__(64) k_##NAME##_fd_step_quad(                                            \
      const u32* fwd, const u32* bwd, int chains, int t, int w0, int chain_len, int count, u32* pts, u32* hand,        \
      size_t hand_box, int max_steps, int* gate, int inject_fault, BoxStride bs)                                       \
      BODY(EC_BOX_GATE_CHECK(gate, 1);                                                                                 \
           fd_quad_step_body<QUAD>(fwd + blockIdx.y * bs.state, bwd + blockIdx.y * bs.state, chains, t, w0, chain_len, \
                                   count, pts + blockIdx.y * bs.pts, hand + blockIdx.y * hand_box, max_steps, gate,    \
                                   inject_fault);)                                                                     \
  extern "C" __global__ void __launch_bounds__(64) k_##NAME##_fd_table_quad(const u32* seeds, int chains, int t,       \
                                                                            u32* fwd, u32* bwd, u32* hand,             \
                                                                            size_t hand_box, int* gate,                \
                                                                            int inject_fault, BoxStride bs)            \
      BODY(EC_BOX_GATE_CHECK(gate, 1);                                                                                 \
           fd_quad_table_body<QUAD>(seeds + blockIdx.y * bs.pts, chains, t, fwd + blockIdx.y * bs.state,               \
                                    bwd + blockIdx.y * bs.state, hand + blockIdx.y * hand_box, gate, inject_fault);)   \
  extern "C" __global__ void __launch_bounds__(64) k_##NAME##_encode(const u32* pts, int count, uint8_t* enc,          \
                                                                     const int* gate, BoxStride bs)                    \
      BODY(EC_BOX_GATE_CHECK(gate, 1);                                                                                 \
           encode_body<CURVE>(pts + blockIdx.y * bs.pts, count, enc + blockIdx.y * bs.enc);)

#define EC_WIN_KERNELS(NAME, CURVE, BODY)                                                                                    \
  extern "C" __global__ void __launch_bounds__(64) k_##NAME##_build_tables(                                                  \
      const uint8_t* enc, size_t enc_stride, const u32* pts, int count, u32* tab, uint8_t* ok, const int* gate)              \
      BODY(build_tables_body<CURVE>(enc, enc_stride, pts, count, tab, ok, gate);)                                            \
  extern "C" __global__ void __launch_bounds__(128) k_##NAME##_comb_build(u32* comb) BODY(comb_build_body<CURVE>(comb);)     \
  extern "C" __global__ void __launch_bounds__(DW_THREADS) k_##NAME##_dual_win(                                              \
      const u32* comb, const u32* tab1, const uint8_t* k1, size_t k1_stride, const u32* tab2, const uint8_t* k2,             \
      size_t k2_stride, int count, u32* out_pts)                                                                             \
      BODY(extern __shared__ u32 lds[];                                                                                      \
           dual_win_body<CURVE>(comb, tab1, k1, k1_stride, tab2, k2, k2_stride, count, out_pts, lds);)                       \
  extern "C" __global__ void __launch_bounds__(64) k_##NAME##_add_pts(const u32* a, const u32* b, int count, u32* out)       \
      BODY(const int x = blockIdx.x * blockDim.x + threadIdx.x; if (x >= count) return;                                      \
           typename CURVE::Point p, q, r;                                                                                    \
           load_point_aos<CURVE>(p, a + (size_t)x * CURVE::POINT_WORDS);                                                     \
           load_point_aos<CURVE>(q, b + (size_t)x * CURVE::POINT_WORDS);                                                     \
           CURVE::add(r, p, q);                                                                                              \
           store_point_aos<CURVE>(out + (size_t)x * CURVE::POINT_WORDS, r);)                                                 \
  extern "C" __global__ void __launch_bounds__(256) k_##NAME##_sum_points(const u32* pts, int m, u32* out)                   \
      BODY(extern __shared__ u32 lds[]; sum_points_body<CURVE>(pts, m, out, lds);)

#if EC_PART != 2
EC_KERNELS(secp, Secp, OrderSecp, QuadSecp, EC_DEF)
EC_WIN_KERNELS(secp, Secp, EC_DEF)
extern "C" __global__ void __launch_bounds__(64) k_secp_fd_step_oct(const u32* fwd, const u32* bwd, int chains, int t, int w0, int chain_len,
                                                                    int count, u32* pts, u32* hand, size_t hand_box, int max_steps,
                                                                    int* gate, int inject_fault, BoxStride bs) {
  EC_BOX_GATE_CHECK(gate, 1);
  fd_quad_step_body<OctSecp>(fwd + blockIdx.y * bs.state, bwd + blockIdx.y * bs.state, chains, t, w0, chain_len, count,
                             pts + blockIdx.y * bs.pts, hand + blockIdx.y * hand_box, max_steps, gate, inject_fault);
}
extern "C" __global__ void __launch_bounds__(64) k_secp_fd_table_oct(const u32* seeds, int chains, int t, u32* fwd, u32* bwd, u32* hand,
                                                                     size_t hand_box, int* gate, int inject_fault, BoxStride bs) {
  EC_BOX_GATE_CHECK(gate, 1);
  fd_quad_table_body<OctSecp>(seeds + blockIdx.y * bs.pts, chains, t, fwd + blockIdx.y * bs.state, bwd + blockIdx.y * bs.state,
                              hand + blockIdx.y * hand_box, gate, inject_fault);
}
extern "C" __global__ void __launch_bounds__(64) k_secp_encode_batch(const u32* pts, int count, uint8_t* enc, const int* gate,
                                                                     BoxStride bs) {
  EC_BOX_GATE_CHECK(gate, 1);
  secp_encode_batch_body(pts + blockIdx.y * bs.pts, count, enc + blockIdx.y * bs.enc);
}
#endif
#if EC_PART == 1
EC_KERNELS(rist, Ristretto, OrderEd, QuadRist, EC_DECL)
EC_WIN_KERNELS(rist, Ristretto, EC_DECL)
#else
EC_KERNELS(rist, Ristretto, OrderEd, QuadRist, EC_DEF)
EC_WIN_KERNELS(rist, Ristretto, EC_DEF)
#endif

#if EC_PART != 2
// ---- launchers ---------------------------------------------------------------------------------------
static inline int blocks_for(int count) { return (count + 63) / 64; }

extern "C" int ec_point_words(int group) { return group == 1 ? Secp::POINT_WORDS : Ristretto::POINT_WORDS; }

extern "C" int ec_launch_decode(int group, const uint8_t* enc, int count, uint32_t* pts, uint8_t* ok, hipStream_t s) {
  if (count <= 0) return 0;
  if (group == 1) hipLaunchKernelGGL(k_secp_decode, dim3(blocks_for(count)), dim3(64), 0, s, enc, count, pts, ok);
  else hipLaunchKernelGGL(k_rist_decode, dim3(blocks_for(count)), dim3(64), 0, s, enc, count, pts, ok);
  return (int)hipGetLastError();
}
// boxes > 1: the same launch for `boxes` boxes laid out one after the other (strides in the units of BoxStride; gate[b])
extern "C" int ec_launch_commit_eval_boxes(int group, const uint32_t* cm, int t, const int64_t* positions, int count,
                                           uint8_t* x_enc, const int* gate, int want, int boxes, size_t cm_stride,
                                           size_t pos_stride, size_t enc_stride, hipStream_t s) {
  if (count <= 0 || boxes <= 0) return 0;
  const BoxStride bs{cm_stride, pos_stride, 0, 0, enc_stride};
  if (group == 1)
    hipLaunchKernelGGL(k_secp_commit_eval, dim3(blocks_for(count), boxes), dim3(64), 0, s, cm, t, positions, count, x_enc, gate, want, bs);
  else
    hipLaunchKernelGGL(k_rist_commit_eval, dim3(blocks_for(count), boxes), dim3(64), 0, s, cm, t, positions, count, x_enc, gate, want, bs);
  return (int)hipGetLastError();
}
extern "C" int ec_launch_commit_eval(int group, const uint32_t* cm, int t, const int64_t* positions, int count,
                                     uint8_t* x_enc, const int* gate, int want, hipStream_t s) {
  return ec_launch_commit_eval_boxes(group, cm, t, positions, count, x_enc, gate, want, 1, 0, 0, 0, s);
}
extern "C" int ec_launch_dual_mul(int group, const uint8_t* p1, size_t p1_stride, const uint8_t* k1, const uint8_t* p2,
                                  const uint8_t* k2, size_t k2_stride, int count, uint8_t* out, uint8_t* ok,
                                  hipStream_t s) {
  if (count <= 0) return 0;
  if (group == 1)
    hipLaunchKernelGGL(k_secp_dual_mul, dim3(blocks_for(count)), dim3(64), 0, s, p1, p1_stride, k1, p2, k2, k2_stride, count, out, ok);
  else
    hipLaunchKernelGGL(k_rist_dual_mul, dim3(blocks_for(count)), dim3(64), 0, s, p1, p1_stride, k1, p2, k2, k2_stride, count, out, ok);
  return (int)hipGetLastError();
}
extern "C" int ec_launch_add(int group, const uint8_t* a, const uint8_t* b, int count, uint8_t* out, uint8_t* ok,
                             hipStream_t s) {
  if (count <= 0) return 0;
  if (group == 1) hipLaunchKernelGGL(k_secp_add, dim3(blocks_for(count)), dim3(64), 0, s, a, b, count, out, ok);
  else hipLaunchKernelGGL(k_rist_add, dim3(blocks_for(count)), dim3(64), 0, s, a, b, count, out, ok);
  return (int)hipGetLastError();
}

extern "C" int ec_launch_encode_gated(int group, const uint32_t* pts, int count, uint8_t* enc, const int* gate, hipStream_t s);

// ---- forward differences: seeds (m0 positions from `positions`), tables, stepping, encoding -------------------------
// pts: [count][point words] internal points, index 0 = first position of the batch; seeds go to pts + seed0.
// Two-level seeding (state_l1 != null, chains > 1): Horner's rule -- (t-1) small scalar multiplications per seed, most of
// a box's instructions when all chains*t seeds are computed that way -- only for the t positions in the MIDDLE of the seed
// window; one stride-1 forward-difference chain built from those t values steps both ways through the rest of the
// window at t additions per seed.  Then the strided chains as before.
extern "C" int ec_launch_encode_boxes(int group, const uint32_t* pts, int count, uint8_t* enc, const int* gate, int boxes,
                                      size_t pts_stride, size_t enc_stride, hipStream_t s);
// boxes > 1: `boxes` boxes of the same shape in every launch.  Per box: cm_stride words of commitments, pos_stride
// positions, pts_stride words of points, enc_stride bytes of encodings, state_stride words of difference tables
// (state_fwd, state_bwd and state_l1 all advance by it), gate[b].
// quad != null: the stepping launches are pipelines of quad-lane stages (fd_quad_step_body) -- for boxes that have the
// chip to themselves; quad->hand holds ec_fd_quad_hand_words() words per box and is zeroed here before each launch,
// quad->gate is the boxes' (writable) gate.
// lanes per point of the stepping / table pipelines: secp256k1 8 (OctSecp; 4 = QuadSecp with quad->oct == 0), ristretto255 4
static inline int ec_quad_lanes(int group, int oct) { return group == 1 && oct ? 8 : 4; }
extern "C" size_t ec_fd_quad_hand_words(int group, int oct, int t, int chains, int w0, int chain_len) {
  const int m0 = chains * t, w1 = (m0 - t) / 2, lanes = ec_quad_lanes(group, oct);
  const int steps_l1 = std::max(m0 - 1 - w1, w1 + t - 1), steps = std::max(chain_len - 1 - w0, w0 + t - 1);
  return std::max(std::max(ec_quad_hand_words(t, 1, steps_l1, lanes), ec_quad_hand_words(t, chains, steps, lanes)),
                  ec_quad_hand_words(t, chains, t, lanes));
}
extern "C" size_t ec_fd_seed_tab_words(int group, int seeds) {
  return (size_t)blocks_for(16 * seeds) * 64 * 8 * (size_t)(group == 1 ? Secp::CACHED_WORDS : Ristretto::CACHED_WORDS);
}
extern "C" int ec_launch_fd_boxes_q(int group, const uint32_t* cm, int t, const int64_t* positions, int count, int chains, int w0,
                                    int chain_len, uint32_t* pts, uint32_t* state_fwd, uint32_t* state_bwd, uint32_t* state_l1,
                                    uint8_t* x_enc, int split_seeds, const int* gate, int boxes, size_t cm_stride,
                                    size_t pos_stride, size_t pts_stride, size_t state_stride, size_t enc_stride,
                                    const EcQuadStepping* quad, hipStream_t s) {
  const int m0 = chains * t;
  const size_t seed0 = (size_t)chains * w0;
  const int pw = ec_point_words(group);
  uint32_t* seeds = pts + seed0 * pw;
  const int lanes = ((t + 63) / 64) * 64;                       // one lane per level
  const size_t lds = (size_t)lanes * pw * 4;
  const bool split = split_seeds && t >= 64;                    // 8 lanes per seed
  const size_t lds_seed = (size_t)64 * pw * 4;
  const bool two_level = state_l1 != nullptr && chains > 1;
  const int w1 = two_level ? (m0 - t) / 2 : 0;                  // first Horner seed inside the window
  const int horner = two_level ? t : m0;
  const int64_t* hpos = positions + seed0 + w1;
  uint32_t* hseeds = seeds + (size_t)w1 * pw;
  uint32_t* l1_fwd = state_l1;
  uint32_t* l1_bwd = two_level ? state_l1 + (size_t)t * pw : nullptr;
  const BoxStride bs{cm_stride, pos_stride, pts_stride, state_stride, enc_stride};
  const unsigned B = (unsigned)boxes;
  auto step_quad = [&](uint32_t* fwd, uint32_t* bwd, int nchains, int first, int len, int cnt, uint32_t* out) -> int {
    const int lanes_q = ec_quad_lanes(group, quad->oct), levels = 64 / lanes_q;
    const int nst = (t + levels - 1) / levels;
    const int max_steps = std::max(len - 1 - first, first + t - 1);
    const size_t need = ec_quad_hand_words(t, nchains, max_steps, lanes_q);
    if (need > quad->hand_box_words) return (int)hipErrorInvalidValue;
    // (only what this launch uses is zeroed: the space is sized for the largest launch of the box)
    for (int b = 0; b < boxes; ++b) {
      hipError_t e = hipMemsetAsync(quad->hand + (size_t)b * quad->hand_box_words, 0, need * 4, s);
      if (e != hipSuccess) return (int)e;
    }
    if (group == 1 && quad->oct)
      hipLaunchKernelGGL(k_secp_fd_step_oct, dim3(2 * nchains * nst, B), dim3(64), OctSecp::LDS_WORDS * 4, s, fwd, bwd, nchains, t,
                         first, len, cnt, out, quad->hand, quad->hand_box_words, max_steps, quad->gate, quad->fault, bs);
    else if (group == 1)
      hipLaunchKernelGGL(k_secp_fd_step_quad, dim3(2 * nchains * nst, B), dim3(64), QuadSecp::LDS_WORDS * 4, s, fwd, bwd, nchains, t,
                         first, len, cnt, out, quad->hand, quad->hand_box_words, max_steps, quad->gate, quad->fault, bs);
    else
      hipLaunchKernelGGL(k_rist_fd_step_quad, dim3(2 * nchains * nst, B), dim3(64), QuadRist::LDS_WORDS * 4, s, fwd, bwd, nchains, t,
                         first, len, cnt, out, quad->hand, quad->hand_box_words, max_steps, quad->gate, quad->fault, bs);
    return 0;
  };
  const bool win = split && quad != nullptr && quad->wtab != nullptr;     // 16 lanes per seed, windowed x^lo
  const size_t lds_win = ((size_t)64 * pw + SEEDS_WIN_K_ROWS * 64) * 4;
  if (win) {
    if (group == 1)
      hipLaunchKernelGGL(k_secp_fd_seeds_win, dim3(blocks_for(16 * horner), B), dim3(64), lds_win, s, cm, t, hpos, horner, hseeds,
                         quad->wtab, quad->wtab_box_words, gate, bs);
    else
      hipLaunchKernelGGL(k_rist_fd_seeds_win, dim3(blocks_for(16 * horner), B), dim3(64), lds_win, s, cm, t, hpos, horner, hseeds,
                         quad->wtab, quad->wtab_box_words, gate, bs);
  }
  auto table_quad = [&](const uint32_t* sd, int nchains, uint32_t* fw, uint32_t* bw) -> int {
    const int lanes_q = ec_quad_lanes(group, quad->oct), levels = 64 / lanes_q;
    const int nst = (t + levels - 1) / levels;
    const size_t need = ec_quad_hand_words(t, nchains, t, lanes_q);
    if (need > quad->hand_box_words) return (int)hipErrorInvalidValue;
    for (int b = 0; b < boxes; ++b) {
      hipError_t e = hipMemsetAsync(quad->hand + (size_t)b * quad->hand_box_words, 0, need * 4, s);
      if (e != hipSuccess) return (int)e;
    }
    if (group == 1 && quad->oct)
      hipLaunchKernelGGL(k_secp_fd_table_oct, dim3(nchains * nst, B), dim3(64), OctSecp::LDS_WORDS * 4, s, sd, nchains, t, fw, bw,
                         quad->hand, quad->hand_box_words, quad->gate, quad->fault, bs);
    else if (group == 1)
      hipLaunchKernelGGL(k_secp_fd_table_quad, dim3(nchains * nst, B), dim3(64), QuadSecp::LDS_WORDS * 4, s, sd, nchains, t, fw, bw,
                         quad->hand, quad->hand_box_words, quad->gate, quad->fault, bs);
    else
      hipLaunchKernelGGL(k_rist_fd_table_quad, dim3(nchains * nst, B), dim3(64), QuadRist::LDS_WORDS * 4, s, sd, nchains, t, fw, bw,
                         quad->hand, quad->hand_box_words, quad->gate, quad->fault, bs);
    return 0;
  };
  const bool qtab = quad != nullptr && quad->table;
  if (group == 1) {
    if (win) {}
    else if (split)
      hipLaunchKernelGGL(k_secp_fd_seeds_split, dim3(blocks_for(8 * horner), B), dim3(64), lds_seed, s, cm, t, hpos, horner, hseeds, gate, bs);
    else
      hipLaunchKernelGGL(k_secp_fd_seeds, dim3(blocks_for(horner), B), dim3(64), 0, s, cm, t, hpos, horner, hseeds, gate, bs);
    if (two_level) {
      if (qtab) { if (int rc = table_quad(hseeds, 1, l1_fwd, l1_bwd)) return rc; }
      else hipLaunchKernelGGL(k_secp_fd_table, dim3(1, B), dim3(lanes), lds, s, hseeds, 1, t, l1_fwd, l1_bwd, gate, bs);
      if (quad) { if (int rc = step_quad(l1_fwd, l1_bwd, 1, w1, m0, m0, seeds)) return rc; }
      else hipLaunchKernelGGL(k_secp_fd_step, dim3(2, B), dim3(lanes), lds, s, l1_fwd, l1_bwd, 1, t, w1, m0, m0, seeds, gate, bs);
    }
    if (qtab) { if (int rc = table_quad(seeds, chains, state_fwd, state_bwd)) return rc; }
    else hipLaunchKernelGGL(k_secp_fd_table, dim3(chains, B), dim3(lanes), lds, s, seeds, chains, t, state_fwd, state_bwd, gate, bs);
    if (quad) { if (int rc = step_quad(state_fwd, state_bwd, chains, w0, chain_len, count, pts)) return rc; }
    else hipLaunchKernelGGL(k_secp_fd_step, dim3(2 * chains, B), dim3(lanes), lds, s, state_fwd, state_bwd, chains, t, w0, chain_len,
                            count, pts, gate, bs);
  } else {
    if (win) {}
    else if (split)
      hipLaunchKernelGGL(k_rist_fd_seeds_split, dim3(blocks_for(8 * horner), B), dim3(64), lds_seed, s, cm, t, hpos, horner, hseeds, gate, bs);
    else
      hipLaunchKernelGGL(k_rist_fd_seeds, dim3(blocks_for(horner), B), dim3(64), 0, s, cm, t, hpos, horner, hseeds, gate, bs);
    if (two_level) {
      if (qtab) { if (int rc = table_quad(hseeds, 1, l1_fwd, l1_bwd)) return rc; }
      else hipLaunchKernelGGL(k_rist_fd_table, dim3(1, B), dim3(lanes), lds, s, hseeds, 1, t, l1_fwd, l1_bwd, gate, bs);
      if (quad) { if (int rc = step_quad(l1_fwd, l1_bwd, 1, w1, m0, m0, seeds)) return rc; }
      else hipLaunchKernelGGL(k_rist_fd_step, dim3(2, B), dim3(lanes), lds, s, l1_fwd, l1_bwd, 1, t, w1, m0, m0, seeds, gate, bs);
    }
    if (qtab) { if (int rc = table_quad(seeds, chains, state_fwd, state_bwd)) return rc; }
    else hipLaunchKernelGGL(k_rist_fd_table, dim3(chains, B), dim3(lanes), lds, s, seeds, chains, t, state_fwd, state_bwd, gate, bs);
    if (quad) { if (int rc = step_quad(state_fwd, state_bwd, chains, w0, chain_len, count, pts)) return rc; }
    else hipLaunchKernelGGL(k_rist_fd_step, dim3(2 * chains, B), dim3(lanes), lds, s, state_fwd, state_bwd, chains, t, w0, chain_len,
                            count, pts, gate, bs);
  }
  if (x_enc != nullptr) {
    const int rc = ec_launch_encode_boxes(group, pts, count, x_enc, gate, boxes, pts_stride, enc_stride, s);
    if (rc != 0) return rc;
  }
  return (int)hipGetLastError();
}
extern "C" int ec_launch_fd_boxes(int group, const uint32_t* cm, int t, const int64_t* positions, int count, int chains, int w0,
                                  int chain_len, uint32_t* pts, uint32_t* state_fwd, uint32_t* state_bwd, uint32_t* state_l1,
                                  uint8_t* x_enc, int split_seeds, const int* gate, int boxes, size_t cm_stride,
                                  size_t pos_stride, size_t pts_stride, size_t state_stride, size_t enc_stride, hipStream_t s) {
  return ec_launch_fd_boxes_q(group, cm, t, positions, count, chains, w0, chain_len, pts, state_fwd, state_bwd, state_l1, x_enc,
                              split_seeds, gate, boxes, cm_stride, pos_stride, pts_stride, state_stride, enc_stride, nullptr, s);
}
extern "C" int ec_launch_fd(int group, const uint32_t* cm, int t, const int64_t* positions, int count, int chains, int w0,
                            int chain_len, uint32_t* pts, uint32_t* state_fwd, uint32_t* state_bwd, uint32_t* state_l1,
                            uint8_t* x_enc, int split_seeds, const int* gate, hipStream_t s) {
  return ec_launch_fd_boxes(group, cm, t, positions, count, chains, w0, chain_len, pts, state_fwd, state_bwd, state_l1, x_enc,
                            split_seeds, gate, 1, 0, 0, 0, 0, 0, s);
}

// ---- windowed path ---------------------------------------------------------------------------------------------------
extern "C" int ec_cached_words(int group) { return group == 1 ? Secp::CACHED_WORDS : Ristretto::CACHED_WORDS; }
extern "C" int ec_comb_words(int group) { return 65 * 8 * (group == 1 ? Secp::AFFINE_PACKED_WORDS : Ristretto::AFFINE_PACKED_WORDS); }

extern "C" int ec_launch_comb_build(int group, uint32_t* comb, hipStream_t s) {
  if (group == 1) hipLaunchKernelGGL(k_secp_comb_build, dim3(1), dim3(128), 0, s, comb);
  else hipLaunchKernelGGL(k_rist_comb_build, dim3(1), dim3(128), 0, s, comb);
  return (int)hipGetLastError();
}
// tables of `count` bases: from encodings (enc, stride 0 = one shared base; ok receives the validity flags) or from
// internal points (pts)
extern "C" int ec_launch_build_tables(int group, const uint8_t* enc, size_t enc_stride, const uint32_t* pts, int count,
                                      uint32_t* tab, uint8_t* ok, const int* gate, hipStream_t s) {
  if (count <= 0) return 0;
  if (group == 1)
    hipLaunchKernelGGL(k_secp_build_tables, dim3(blocks_for(count)), dim3(64), 0, s, enc, enc_stride, pts, count, tab, ok, gate);
  else
    hipLaunchKernelGGL(k_rist_build_tables, dim3(blocks_for(count)), dim3(64), 0, s, enc, enc_stride, pts, count, tab, ok, gate);
  return (int)hipGetLastError();
}
// out_pts[x] = k1[x] * (G if comb else P1[x]) + k2[x] * P2[x]; tab2 may be null (single multiplication)
extern "C" int ec_launch_dual_win(int group, const uint32_t* comb, const uint32_t* tab1, const uint8_t* k1, size_t k1_stride,
                                  const uint32_t* tab2, const uint8_t* k2, size_t k2_stride, int count, uint32_t* out_pts,
                                  hipStream_t s) {
  if (count <= 0) return 0;
  // with a comb the generator takes the first table's place: LDS holds the comb's 9 rows of recoded scalar + one table's 10
  // (DW_K_ROWS = 20); a comb AND two tables would write rows 9..28
  if (comb != nullptr && tab1 != nullptr) return (int)hipErrorInvalidValue;
  const size_t lds = ((comb != nullptr ? (size_t)ec_comb_words(group) : 0) + (size_t)DW_K_ROWS * DW_THREADS) * 4;
  const dim3 grid((count + DW_THREADS - 1) / DW_THREADS);
  if (group == 1)
    hipLaunchKernelGGL(k_secp_dual_win, grid, dim3(DW_THREADS), lds, s, comb, tab1, k1, k1_stride, tab2, k2, k2_stride, count, out_pts);
  else
    hipLaunchKernelGGL(k_rist_dual_win, grid, dim3(DW_THREADS), lds, s, comb, tab1, k1, k1_stride, tab2, k2, k2_stride, count, out_pts);
  return (int)hipGetLastError();
}
// out[x] = a[x] + b[x] in internal coordinates (out may alias a or b)
extern "C" int ec_launch_add_pts(int group, const uint32_t* a, const uint32_t* b, int count, uint32_t* out, hipStream_t s) {
  if (count <= 0) return 0;
  if (group == 1) hipLaunchKernelGGL(k_secp_add_pts, dim3(blocks_for(count)), dim3(64), 0, s, a, b, count, out);
  else hipLaunchKernelGGL(k_rist_add_pts, dim3(blocks_for(count)), dim3(64), 0, s, a, b, count, out);
  return (int)hipGetLastError();
}
// internal points -> canonical encodings (secp256k1: eight points per lane share an inversion); gate: run only if gate[b] == 1
extern "C" int ec_launch_encode_boxes(int group, const uint32_t* pts, int count, uint8_t* enc, const int* gate, int boxes,
                                      size_t pts_stride, size_t enc_stride, hipStream_t s) {
  if (count <= 0 || boxes <= 0) return 0;
  const BoxStride bs{0, 0, pts_stride, 0, enc_stride};
  if (group == 1)
    hipLaunchKernelGGL(k_secp_encode_batch, dim3(blocks_for((count + 7) / 8), boxes), dim3(64), 0, s, pts, count, enc, gate, bs);
  else
    hipLaunchKernelGGL(k_rist_encode, dim3(blocks_for(count), boxes), dim3(64), 0, s, pts, count, enc, gate, bs);
  return (int)hipGetLastError();
}
extern "C" int ec_launch_encode_gated(int group, const uint32_t* pts, int count, uint8_t* enc, const int* gate, hipStream_t s) {
  return ec_launch_encode_boxes(group, pts, count, enc, gate, 1, 0, 0, s);
}
extern "C" int ec_launch_encode(int group, const uint32_t* pts, int count, uint8_t* enc, hipStream_t s) {
  return ec_launch_encode_gated(group, pts, count, enc, nullptr, s);
}
// out (one point, may alias pts) = sum of the m points at pts
extern "C" int ec_launch_sum(int group, const uint32_t* pts, int m, uint32_t* out, hipStream_t s) {
  if (m <= 0) return 0;
  const size_t lds = (size_t)256 * ec_point_words(group) * 4;
  if (group == 1) hipLaunchKernelGGL(k_secp_sum_points, dim3(1), dim3(256), lds, s, pts, m, out);
  else hipLaunchKernelGGL(k_rist_sum_points, dim3(1), dim3(256), lds, s, pts, m, out);
  return (int)hipGetLastError();
}
#endif  // EC_PART != 2
