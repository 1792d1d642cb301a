// 2048-bit Montgomery arithmetic for gfx950, "row" layout: one number is spread over the 16 lanes of a DPP row
// (4 numbers per 64-lane wavefront).  For a LATENCY-bound launch that has the chip to itself.
//
// Why (round 6): the Horner seeds of a lone X path are 4096 numbers x ~7 900 strictly sequential Montgomery operations.  In the
// quad layout (bn_quad.h: 4 lanes per number, 18 limbs per lane) that is 256 waves on 1 024 SIMDs and 52 ms -- a wave alone on
// its SIMD issues one VALU instruction every ~5 cycles whatever its instruction-level parallelism, so the launch's time follows
// the instruction count of ONE number's chain: 41 instructions per row (36 multiply-adds + 5) x 72 rows per product.  Sixteen
// lanes per number cut the row to 5 + 5 multiply-adds + 5 = 15 instructions (a squaring: 3 + 5 + 5 = 13) for 1.3x the issue slots
// per number, on 1 024 waves -- one per SIMD -- instead of 256.  Measured alone on the chip: the seed launch 52 -> 28 ms
// (0.55, not the 0.37 of the instruction count: the DPP and LDS round trips of a row do not shrink with it), the X path of a
// stand-alone commit_eval 68.5 -> 47.7 ms (profiles/r06_commit_eval_alone.txt).  Inside a verifier's box it LOSES: the box is bound
// by the sum of its work, not by this chain (mpvss_capi.cpp::eval_x).
//
// Representation: radix 2^29, 72 limbs, R = 2^2088 -- the SAME Montgomery domain and the same 72-word limb form in HBM as the
// quad and pair layouts; lane l of the row owns limb slots 5l .. 5l+4 of 80 (slots 72..79 are zero: lane 14 holds two limbs,
// lane 15 none).  Same CIOS row as bn_quad.h::mont_mul: a[k] b_i, m from lane 0's lowest column, m n[k], every lane carries the
// upper bits of its lowest column into its next one and hands the low 29 bits to the lane below.  DPP: row_newbcast:0 (m),
// row_shl:1 / row_shr:1 with bound_ctrl (the row's end lanes read 0: no lane masks needed).
// Bounds: a column is carried every 5 rows, so it collects at most 10 products < 2^58 between two carries (2^61.4 in all) --
// tests/test_limb_model.py::test_row_layout_model runs the same integer pipeline with worst-case limbs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bnrow {

typedef uint32_t u32;
typedef uint64_t u64;

constexpr int W = 29;
constexpr int L = 72;                // limbs of a number (rows of a product)
constexpr int LPL = 5;               // limb slots per lane
constexpr int LANES = 16;
constexpr u32 MASK = (1u << W) - 1;
constexpr int NUMS_PER_WAVE = 4;
constexpr int SLOT_WORDS = 80;       // LDS words per number operand slot (16 lanes x 5; words 72..79 are zero)
constexpr int FULL_GROUPS = L / LPL; // 14 groups of 5 rows, then TAIL_ROWS
constexpr int TAIL_ROWS = L % LPL;   // 2

constexpr int DPP_ROW_BCAST0 = 0x150;   // row_newbcast:0
constexpr int DPP_ROW_SHL1 = 0x101;     // lane l reads lane l+1 of its row (lane 15: 0)
constexpr int DPP_ROW_SHR1 = 0x111;     // lane l reads lane l-1 of its row (lane 0: 0)

__device__ __forceinline__ u32 row_bcast0(u32 v) { return (u32)__builtin_amdgcn_mov_dpp((int)v, DPP_ROW_BCAST0, 0xf, 0xf, true); }
__device__ __forceinline__ u32 row_from_next(u32 v) { return (u32)__builtin_amdgcn_mov_dpp((int)v, DPP_ROW_SHL1, 0xf, 0xf, true); }
__device__ __forceinline__ u32 row_from_prev(u32 v) { return (u32)__builtin_amdgcn_mov_dpp((int)v, DPP_ROW_SHR1, 0xf, 0xf, true); }

struct Lane {
  u32 l;         // lane index inside the row, 0..15
  u32 mask28;    // 2^W - 1 in a VGPR, opaque (keeps "broadcast & mask" one v_and_b32_dpp; see bn_quad.h)
};

__device__ __forceinline__ Lane make_lane() {
  Lane ln;
  ln.l = threadIdx.x & 15;
  ln.mask28 = MASK;
  asm volatile("" : "+v"(ln.mask28));
  return ln;
}

// one row of the product: T (rotated by RR: local position k lives in T[(k + RR) % LPL]) += a * bi + m * n, then retire
template <u32 N0INV, bool SQ, int RR>
__device__ __forceinline__ void row_step(u64 (&T)[LPL], const u32 (&a)[LPL], const u32 (&n)[LPL], u32 bi, const Lane& ln) {
  if (SQ) {
    // b is a itself: row r = 5 o + RR visits only the local positions k >= RR, k > RR with the doubled limb (bn_quad.h: every
    // pair of limbs is then counted exactly twice, every square once, by the same instructions in all lanes)
    const u32 bi2 = bi << 1;
#pragma unroll
    for (int k = RR; k < LPL; ++k) T[(k + RR) % LPL] += (u64)a[k] * (k > RR ? bi2 : bi);
  } else {
#pragma unroll
    for (int k = 0; k < LPL; ++k) T[(k + RR) % LPL] += (u64)a[k] * bi;
  }
  const u32 m = row_bcast0((u32)T[RR] * N0INV) & ln.mask28;
#pragma unroll
  for (int k = 0; k < LPL; ++k) T[(k + RR) % LPL] += (u64)m * n[k];
  const u64 ret = T[RR];
  T[(RR + 1) % LPL] += ret >> W;
  T[RR] = (u64)(row_from_next((u32)ret) & ln.mask28);      // lane 15 reads 0 (bound_ctrl): a fresh zero column at the top
#pragma unroll
  for (int k = 0; k < LPL; ++k) asm volatile("" : "+v"(T[k]));      // pin the row-wise order (bn_quad.h)
}

// r = a * b * R^-1 (mod N), almost normalised and < 2N when a, b < 2N.
//   a, n : this lane's 5 limb slots (registers);  b : LDS pointer to the 72 limbs of the second operand of THIS number
// SQ: b must be (an LDS copy of) a itself.
template <u32 N0INV, bool SQ = false>
__device__ __forceinline__ void mont_mul(u32 (&r)[LPL], const u32 (&a)[LPL], const u32* __restrict__ b, const u32 (&n)[LPL],
                                         const Lane& ln) {
  u64 T[LPL];
#pragma unroll
  for (int k = 0; k < LPL; ++k) T[k] = 0;
  u32 bnext = b[0];
#pragma nounroll
  for (int o = 0; o < FULL_GROUPS; ++o) {
    const u32* bo = b + o * LPL;
    u32 bi;
    bi = bnext; bnext = bo[1]; row_step<N0INV, SQ, 0>(T, a, n, bi, ln);
    bi = bnext; bnext = bo[2]; row_step<N0INV, SQ, 1>(T, a, n, bi, ln);
    bi = bnext; bnext = bo[3]; row_step<N0INV, SQ, 2>(T, a, n, bi, ln);
    bi = bnext; bnext = bo[4]; row_step<N0INV, SQ, 3>(T, a, n, bi, ln);
    bi = bnext; bnext = bo[5]; row_step<N0INV, SQ, 4>(T, a, n, bi, ln);      // (bo[5] of the last group is limb 70: read below)
  }
  {
    const u32 b71 = b[L - 1];
    row_step<N0INV, SQ, 0>(T, a, n, bnext, ln);      // row 70
    row_step<N0INV, SQ, 1>(T, a, n, b71, ln);        // row 71
  }
  // 72 = 14 x 5 + 2 rows: local position k now lives in T[(k + 2) % 5].  Pass 1: carry-propagate inside the lane
  u64 c = 0;
#pragma unroll
  for (int k = 0; k < LPL; ++k) {
    const u64 v = T[(k + TAIL_ROWS) % LPL] + c;
    r[k] = (u32)v & MASK;
    c = v >> W;
  }
  // pass 2: the lane's carry-out (< 2^36) goes to the next lane, one more local step
  const u32 cl = row_from_prev((u32)c);
  const u32 ch = row_from_prev((u32)(c >> 32));
  const u64 v = (u64)r[0] + (((u64)ch << 32) | cl);
  r[0] = (u32)v & MASK;
  r[1] += (u32)(v >> W);
}

template <u32 N0INV>
__device__ __forceinline__ void mont_sqr(u32 (&r)[LPL], const u32 (&a)[LPL], const u32* __restrict__ self, const u32 (&n)[LPL],
                                         const Lane& ln) {
  mont_mul<N0INV, true>(r, a, self, n, ln);
}

// ---- operand slots (80 words per number) and the 72-word limb form in HBM ----------------------------------------
__device__ __forceinline__ void slot_store(u32* slot, const u32 (&a)[LPL], const Lane& ln) {
#pragma unroll
  for (int k = 0; k < LPL; ++k) slot[ln.l * LPL + k] = a[k];
}

// this lane's limb slots of a number in limb form (72 words; slots beyond are zero)
__device__ __forceinline__ void load_lane_limbs(u32 (&a)[LPL], const u32* __restrict__ g, const Lane& ln) {
#pragma unroll
  for (int k = 0; k < LPL; ++k) {
    const int j = (int)ln.l * LPL + k;
    a[k] = j < L ? g[j < L ? j : 0] : 0u;
  }
}

__device__ __forceinline__ void store_lane_limbs(u32* __restrict__ g, const u32 (&a)[LPL], const Lane& ln) {
#pragma unroll
  for (int k = 0; k < LPL; ++k) {
    const int j = (int)ln.l * LPL + k;
    if (j < L) g[j] = a[k];
  }
}

// 72 words global -> the number's LDS slot (words 72..79 zeroed): 16 lanes x 5 words
__device__ __forceinline__ void slot_fill_from_global(u32* slot, const u32* __restrict__ g, const Lane& ln) {
#pragma unroll
  for (int k = 0; k < LPL; ++k) {
    const int j = (int)ln.l * LPL + k;
    slot[j] = j < L ? g[j < L ? j : 0] : 0u;
  }
}

}  // namespace bnrow
