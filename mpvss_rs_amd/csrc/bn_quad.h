// 2048-bit Montgomery arithmetic for gfx950, "quad" layout: one number is spread over the
// four lanes of a DPP quad (16 numbers per 64-lane wavefront).
//
// Representation
//   radix 2^29, L = 72 limbs (capacity 2088 bits, R = 2^2088), lane q of the quad owns limbs
//   18q .. 18q+17 in registers.  Limbs are "almost normalised": every limb <= 2^29 - 1 + 2^9.
//   Values are kept in [0, 2N) (R > 4N, so no conditional subtraction inside a chain).
//
// Why this shape (measured on MI355X, profiles/r01_ubench_valu_issue_rates.txt):
//   * v_mad_u64_u32 issues at one wave-instruction per ~2.07 ns per SIMD, but a carry-producing add
//     (v_add_co/v_addc) costs the same, so a radix-2^32 schoolbook row (1 mad + 1 addc per product) is
//     ~1.85x the cost of a row whose products are accumulated carry-free in 64-bit column accumulators
//     (1 mad per product).  The radix is the largest for which that works: every column passes the lowest
//     position of a lane every 18 rows and is carried there, so it collects at most 36 products < 2^58 in
//     between (< 2^63.2, tests/test_limb_model.py); 2^30 would overflow.
//   * a whole 2048-bit operand per lane (64+ VGPRs per operand, 3-4 live) does not fit 3 waves per SIMD,
//     a quarter of one (18 limbs) does.
//   * quad_perm DPP moves data between the four lanes at full VALU rate (no LDS round trip).
//
// Montgomery product (CIOS, one b-limb per step, accumulators shift one limb per step):
//   the second operand is read limb by limb from LDS (all four lanes read the same word ->
//   broadcast, 16 distinct banks per half-wave -> conflict free).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bn {

typedef uint32_t u32;
typedef uint64_t u64;

constexpr int W = 29;
constexpr int L = 72;
constexpr int LPL = 18;              // limbs per lane
constexpr u32 MASK = (1u << W) - 1;
constexpr int NUMS_PER_WAVE = 16;
constexpr int SLOT_WORDS = 72;       // LDS words per number operand slot (288 B = 18 x 16 B)

// quad_perm selectors: dpp_ctrl = p0 | p1<<2 | p2<<4 | p3<<6
constexpr int QP_BCAST0 = 0x00;                          // every lane reads lane 0
constexpr int QP_DOWN = 1 | (2 << 2) | (3 << 4) | (3 << 6);  // lane q reads lane q+1 (lane 3 reads itself)
constexpr int QP_UP = 0 | (0 << 2) | (1 << 4) | (2 << 6);    // lane q reads lane q-1 (lane 0 reads itself)

__device__ __forceinline__ u32 quad_bcast0(u32 v) {
  return (u32)__builtin_amdgcn_mov_dpp((int)v, QP_BCAST0, 0xf, 0xf, true);
}
__device__ __forceinline__ u32 quad_from_next(u32 v) {
  return (u32)__builtin_amdgcn_mov_dpp((int)v, QP_DOWN, 0xf, 0xf, true);
}
__device__ __forceinline__ u32 quad_from_prev(u32 v) {
  return (u32)__builtin_amdgcn_mov_dpp((int)v, QP_UP, 0xf, 0xf, true);
}

struct Lane {
  u32 q;         // lane index inside the quad, 0..3
  u32 not_top;   // all ones unless q == 3
  u32 top28;     // 2^W - 1 unless q == 3 (then 0): "take the low limb of the lane above"
  u32 low01;     // 1 iff q == 0, else 0
  u32 not_low;   // all ones unless q == 0
  u32 mask28;    // 2^W - 1 in a VGPR (lets the compiler fuse "broadcast & mask" into one v_and_b32_dpp)
  u32 one;       // 1 in a VGPR, opaque: "x * one" stays a v_mad_u64_u32 (a 64-bit add would be two carry instructions)
  u32 eight;     // 8 likewise: "hi * eight" must not become a 64-bit shift-and-add
};

__device__ __forceinline__ Lane make_lane() {
  Lane ln;
  ln.q = threadIdx.x & 3;
  ln.not_top = (ln.q != 3) ? 0xffffffffu : 0u;
  ln.top28 = (ln.q != 3) ? MASK : 0u;
  ln.low01 = (ln.q == 0) ? 1u : 0u;
  ln.not_low = (ln.q != 0) ? 0xffffffffu : 0u;
  // Opaque to the optimiser: otherwise `x & mask` becomes v_cndmask_b32 on an SGPR-pair condition,
  // which issues ~4x slower than v_and_b32 on gfx950 (profiles/r01_ubench_valu_issue_rates.txt).
  ln.mask28 = MASK;
  ln.one = 1u;
  ln.eight = 8u;
  asm volatile("" : "+v"(ln.not_top), "+v"(ln.top28), "+v"(ln.low01), "+v"(ln.not_low), "+v"(ln.mask28), "+v"(ln.one),
               "+v"(ln.eight));
  return ln;
}

// r = a * b * R^-1 (mod N), result almost normalised and < 2N when a, b < 2N.
//   a   : this lane's 18 limbs of the first operand (registers)
//   b   : LDS pointer to the 72 limbs of the second operand of THIS number
//   n   : this lane's 18 limbs of the modulus (registers)
// N0INV == 1 for the RFC 3526 prime (N = -1 mod 2^64), the multiply folds away.
//
// One step (one limb b_i of b), per lane: 18 mads a[k]*b_i, m from lane 0's lowest column,
// 18 mads m*n[k]; lane 0's lowest column is then 0 mod 2^29 and retires.  Every lane carries the upper bits
// of its lowest column into its next column and hands the low 29 bits to the lane below (lane 3 starts a
// fresh zero column).  Column accumulators stay below 2^64 (at most 18 steps x 2 products < 2^58.01 between two
// carries of a column, checked for worst-case limbs in tests/test_limb_model.py).
//
// SQ = true: b must be (an LDS copy of) a itself.  Row r = 18 o + rr then only visits the local positions k >= rr:
// k > rr with the doubled limb 2 a_r, k == rr with a_r itself.  Every pair {r, j}, r != j, is then counted exactly
// twice -- once doubled in the row of the limb with the smaller local index, or once in each of the two rows when the
// local indices are equal -- and every square once, in all four lanes by the same instructions: 9.5 instead of 18
// mads per row for the a*a half, 24 % fewer mads per squaring.  (Needs an even number of limbs per lane only in
// the sense that the rule is lane-independent; bounds: tests/test_limb_model.py.)
// OUTER < L / LPL: only the low 18 OUTER limbs of b take part (b < 2^(522 OUTER)) and the result is a b 2^(-522 OUTER) mod N --
// the scalar-ring kernels multiply by small numbers this way (a quarter of a product per Horner step).
template <u32 N0INV, bool SQ = false, int OUTER = L / LPL>
__device__ __forceinline__ void mont_mul(u32 (&r)[LPL], const u32 (&a)[LPL], const u32* __restrict__ b,
                                         const u32 (&n)[LPL], const Lane& ln) {
  u64 T[LPL];
#pragma unroll
  for (int k = 0; k < LPL; ++k) T[k] = 0;
  u32 bnext = b[0];   // software prefetch of the next b limb (one LDS read in flight)
#pragma nounroll
  for (int o = 0; o < OUTER; ++o) {
#pragma unroll
    for (int rr = 0; rr < LPL; ++rr) {
      // local position k lives in T[(k + rr) % LPL]
      const u32 bi = bnext;
      {
        const int nxt = o * LPL + rr + 1;
        bnext = b[nxt < L ? nxt : L - 1];
      }
      if (SQ) {
        const u32 bi2 = bi << 1;
#pragma unroll
        for (int k = rr; k < LPL; ++k) T[(k + rr) % LPL] += (u64)a[k] * (k > rr ? bi2 : bi);
      } else {
#pragma unroll
        for (int k = 0; k < LPL; ++k) T[(k + rr) % LPL] += (u64)a[k] * bi;
      }
      const u32 m = quad_bcast0((u32)T[rr] * N0INV) & ln.mask28;
#pragma unroll
      for (int k = 0; k < LPL; ++k) T[(k + rr) % LPL] += (u64)m * n[k];
      // Every lane moves the upper bits of its lowest column into its next column (same weight) and hands the
      // low 28 bits to the lane below, whose fresh top column they become; lane 0's lowest column is 0 mod 2^W
      // by construction and retires.  No lane-dependent arithmetic: one shift, one 64-bit add, one v_and_b32_dpp.
      {
        const u64 ret = T[rr];
        T[(rr + 1) % LPL] += ret >> W;
        T[rr] = (u64)(quad_from_next((u32)ret) & ln.top28);
      }
      // Pin the row-wise order: without this LLVM reassociates the 19-fold unrolled body into a
      // column-wise (product-scanning) form that keeps every b_i and m_i of the block live and
      // no longer fits 128 VGPRs (4 waves/SIMD).  The empty asm makes each accumulator opaque.
#ifndef MODP_NO_PIN
#pragma unroll
      for (int k = 0; k < LPL; ++k) asm volatile("" : "+v"(T[k]));
#endif
    }
  }
  // L is a multiple of LPL, so local position k is back in T[k].
  // pass 1: carry-propagate inside the lane
  u64 c = 0;
#pragma unroll
  for (int k = 0; k < LPL; ++k) {
    const u64 v = T[k] + c;
    r[k] = (u32)v & MASK;
    c = v >> W;
  }
  // pass 2: hand the lane's carry-out (< 2^37) to the next lane, one more local step
  const u32 cl = quad_from_prev((u32)c) & ln.not_low;
  const u32 ch = quad_from_prev((u32)(c >> 32)) & ln.not_low;
  const u64 v = (u64)r[0] + (((u64)ch << 32) | cl);
  r[0] = (u32)v & MASK;
  r[1] += (u32)(v >> W);
}

// r = a^2 R^-1 (mod N); `self` = LDS slot holding a copy of a
template <u32 N0INV>
__device__ __forceinline__ void mont_sqr(u32 (&r)[LPL], const u32 (&a)[LPL], const u32* __restrict__ self,
                                         const u32 (&n)[LPL], const Lane& ln) {
  mont_mul<N0INV, true>(r, a, self, n, ln);
}

// ---- operand slot helpers (one wave = 16 numbers, slot = 76 words per number) -------------

// store this lane's 19 limbs into the number's LDS slot
__device__ __forceinline__ void slot_store(u32* slot, const u32 (&a)[LPL], const Lane& ln) {
#pragma unroll
  for (int k = 0; k < LPL; ++k) slot[ln.q * LPL + k] = a[k];
}

__device__ __forceinline__ void slot_load(u32 (&a)[LPL], const u32* slot, const Lane& ln) {
#pragma unroll
  for (int k = 0; k < LPL; ++k) a[k] = slot[ln.q * LPL + k];
}

// copy 76 words global -> LDS slot with 16-byte accesses: chunk c (0..18) is handled by lane c & 3
__device__ __forceinline__ void slot_fill_from_global(u32* slot, const u32* __restrict__ g, const Lane& ln) {
  const uint4* g4 = reinterpret_cast<const uint4*>(g);
  uint4* s4 = reinterpret_cast<uint4*>(slot);
#pragma unroll
  for (int c = 0; c < 5; ++c) {
    const int idx = c * 4 + (int)ln.q;
    if (idx < SLOT_WORDS / 4) s4[idx] = g4[idx];
  }
}

__device__ __forceinline__ void slot_spill_to_global(u32* __restrict__ g, const u32* slot, const Lane& ln) {
  uint4* g4 = reinterpret_cast<uint4*>(g);
  const uint4* s4 = reinterpret_cast<const uint4*>(slot);
#pragma unroll
  for (int c = 0; c < 5; ++c) {
    const int idx = c * 4 + (int)ln.q;
    if (idx < SLOT_WORDS / 4) g4[idx] = s4[idx];
  }
}

}  // namespace bn
