// Row-layout kernels (bn_row.h: 16 lanes per number) for a latency-bound launch that has the chip to itself.
//
//   k_modp_commit_eval_row : the Horner seeds of the forward-difference X path of a stand-alone mpvss_modp_commit_eval call
//                            (participant.rs:423-434 at the seed positions; src/mpvss.rs:110-123):
//                            the same program as k_modp_commit_eval (modp_kernels.hip) -- X_i = (..(C_{t-1}^i C_{t-2})^i ..)^i C_0 by
//                            one Montgomery-product site that only chooses its LDS operand -- with a third of the instructions on
//                            a number's sequential chain.  Output: Montgomery limb form (what the inversion tree, the difference
//                            tables and the stepping take); same values as the quad kernel bit for bit (a Montgomery product of
//                            the same operands in [0, 2N) is the same residue, and both layouts finish with the same two carry
//                            passes -- tests/test_gpu_fd.py compares the resulting X with Horner's, tests/test_gpu_row.py the limbs).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "bn_row.h"
#include "modp_kernels.h"

using namespace bnrow;

namespace {
struct ModpConsts {             // same layout as in modp_kernels.hip: N, R^2 mod N, R mod N, plain 1
  u32 n[L];
  u32 r2[L];
  u32 one_m[L];
  u32 one[L];
};
}  // namespace

#ifndef ROW_SETPRIO
#define ROW_SETPRIO 3
#endif

// One wave = 4 numbers.  LDS per wave: operand slot + saved-base slot per number, one shared slot holding one_m.
extern "C" __global__ void __launch_bounds__(64)
k_modp_commit_eval_row(const u32* __restrict__ cm, int t, const int64_t* __restrict__ positions, int count, u32* __restrict__ x_m,
                       const int* __restrict__ gate, int gate_want, const ModpConsts* __restrict__ cs, size_t box_cm_words,
                       size_t box_positions, size_t box_out, int prio) {
  __shared__ __attribute__((aligned(16))) u32 lds[(2 * NUMS_PER_WAVE + 1) * SLOT_WORDS];
  cm += blockIdx.y * box_cm_words;
  positions += blockIdx.y * box_positions;
  x_m += blockIdx.y * box_out * L;
  if (gate != nullptr && *gate != gate_want) return;
  if (prio) __builtin_amdgcn_s_setprio(ROW_SETPRIO);
  const Lane ln = make_lane();
  const int num = threadIdx.x >> 4;
  const int xi = blockIdx.x * NUMS_PER_WAVE + num;
  const bool live = xi < count;
  const int x = live ? xi : count - 1;
  u32* slot = lds + num * SLOT_WORDS;
  u32* bslot = lds + (NUMS_PER_WAVE + num) * SLOT_WORDS;
  u32* oneslot = lds + 2 * NUMS_PER_WAVE * SLOT_WORDS;
  u32 n[LPL], acc[LPL];
  load_lane_limbs(n, cs->n, ln);
  if (threadIdx.x < 16) slot_fill_from_global(oneslot, cs->one_m, ln);
  const u64 pos = (u64)positions[x];
  int nb = (pos == 0) ? 0 : 64 - __builtin_clzll(pos);       // wave-wide maximum bit length of the positions
#pragma unroll
  for (int off = 32; off >= 16; off >>= 1) {
    const int other = __shfl_xor(nb, off);
    nb = other > nb ? other : nb;
  }
  nb = __builtin_amdgcn_readfirstlane(nb);
  __builtin_amdgcn_wave_barrier();

  load_lane_limbs(acc, cm + (size_t)(t - 1) * L, ln);
  // for j = t-2 .. 0:  base = acc; acc = topbit ? base : one;  for bit = nb-2 .. 0: SQUARE; CONDMUL (by base or one, skipped when no
  // number of the wave needs it);  CMUL (by C_j)
  enum { K_SQUARE, K_CONDMUL, K_CMUL, K_DONE };
  int j = t - 2, bit = 0, kind = K_DONE;
  auto begin_coefficient = [&]() {
    if (nb == 0) {   // every position of the wave is 0: acc^0 = 1
      load_lane_limbs(acc, cs->one_m, ln);
      kind = K_CMUL;
      return;
    }
    slot_store(bslot, acc, ln);
    if (!((pos >> (nb - 1)) & 1)) load_lane_limbs(acc, cs->one_m, ln);
    bit = nb - 2;
    kind = (bit >= 0) ? K_SQUARE : K_CMUL;
  };
  if (j >= 0) begin_coefficient();
  while (kind != K_DONE) {
    const u32* bptr = slot;
    bool skip = false;
    if (kind == K_SQUARE) {
      slot_store(slot, acc, ln);
    } else if (kind == K_CONDMUL) {
      const bool mine = (pos >> bit) & 1;
      skip = __builtin_amdgcn_ballot_w64(mine) == 0;
      bptr = mine ? bslot : oneslot;
    } else {
      slot_fill_from_global(slot, cm + (size_t)j * L, ln);
    }
    if (!skip) {
      __builtin_amdgcn_wave_barrier();
      if (kind == K_SQUARE) mont_sqr<MODP_N0INV_C>(acc, acc, slot, n, ln);
      else mont_mul<MODP_N0INV_C>(acc, acc, bptr, n, ln);
      __builtin_amdgcn_wave_barrier();
    }
    if (kind == K_SQUARE) {
      kind = K_CONDMUL;
    } else if (kind == K_CONDMUL) {
      --bit;
      kind = (bit >= 0) ? K_SQUARE : K_CMUL;
    } else {
      --j;
      if (j >= 0) begin_coefficient(); else kind = K_DONE;
    }
  }
  if (live) store_lane_limbs(x_m + (size_t)x * L, acc, ln);
}

// ---------------------------------------------------------------------------------------
// out = B1^e1 * B2^e2 for SMALL batches (the sizes of the reference's own tests and examples: a handful to a few thousand numbers):
// the program of k_modp_dual_exp (modp_kernels.hip: 4-bit fixed windows over the numbers' 16-entry tables in HBM, 2 044 squarings + 511
// + e2_windows products on ONE number's sequential chain) on the row layout -- a call that small is the latency of that chain, and an
// operation of a wave that has its SIMD to itself takes 3.5 instead of 5.7 us (ModpGroup::exp, modp.rs:122-128; dleq.rs:66-84).
// blockIdx.y = 1: a second exponent set over the same tables (the dealer's Y = y^p and a2 = y^w, participant.rs:219 / dleq.rs:213-216).
// Output: Montgomery limb form, converted by k_modp_from_mont (one more product, the exact normalisation).
// ---------------------------------------------------------------------------------------
extern "C" __global__ void __launch_bounds__(64)
k_modp_dual_exp_row(const u32* __restrict__ tab1, size_t tab1_stride, const u32* __restrict__ tab2, size_t tab2_stride,
                    const uint8_t* __restrict__ e1_be, const uint8_t* __restrict__ e2_be, size_t e2_stride, int e2_windows,
                    const uint8_t* __restrict__ e1b_be, const uint8_t* __restrict__ e2b_be, int count, u32* __restrict__ out_m,
                    const ModpConsts* __restrict__ cs) {
  __shared__ __attribute__((aligned(16))) u32 lds[NUMS_PER_WAVE * SLOT_WORDS];
  const Lane ln = make_lane();
  const int num = threadIdx.x >> 4;
  const int xi = blockIdx.x * NUMS_PER_WAVE + num;
  const bool live = xi < count;
  const int x = live ? xi : count - 1;
  u32* slot = lds + num * SLOT_WORDS;
  u32 n[LPL], acc[LPL];
  load_lane_limbs(n, cs->n, ln);
  const u32* t1 = tab1 + (size_t)x * tab1_stride;
  const u32* t2 = tab2 + (size_t)x * tab2_stride;
  const uint8_t* e1 = (blockIdx.y ? e1b_be : e1_be) + (size_t)x * 256;
  const uint8_t* e2 = (blockIdx.y ? e2b_be : e2_be) + (size_t)x * e2_stride;
  const int first_e2 = 512 - e2_windows;
  load_lane_limbs(acc, t1 + (size_t)(e1[0] >> 4) * L, ln);      // the first window loads tab1[d1] instead of multiplying into one
  //   window w = 0 .. 511, most significant first; step s inside it: 0 .. 3 square, 4 tab1, 5 tab2 (the low e2_windows windows), 6 next
  int w = 0, s = (first_e2 == 0) ? 5 : 6;
  while (true) {
    if (s == 6) { ++w; s = 0; }
    if (w == 512) break;
    const bool sq = s < 4;
    if (sq) {
      slot_store(slot, acc, ln);
    } else {
      const uint8_t* e = (s == 4) ? e1 : e2;
      const u32 byte = e[w >> 1];
      const u32 d = (w & 1) ? (byte & 15) : (byte >> 4);
      slot_fill_from_global(slot, ((s == 4) ? t1 : t2) + (size_t)d * L, ln);
    }
    __builtin_amdgcn_wave_barrier();
    if (sq) mont_sqr<MODP_N0INV_C>(acc, acc, slot, n, ln); else mont_mul<MODP_N0INV_C>(acc, acc, slot, n, ln);
    __builtin_amdgcn_wave_barrier();
    ++s;
    if (s == 5 && w < first_e2) s = 6;
  }
  if (live) store_lane_limbs(out_m + ((size_t)blockIdx.y * count + x) * L, acc, ln);
}

// test hook (tests/test_gpu_row.py): out[x] = a[x] * b[x] R^-1 (sq == 0) or a[x]^2 R^-1 (sq != 0), limb form in and out
extern "C" __global__ void __launch_bounds__(64)
k_modp_row_unit(const u32* __restrict__ a_m, const u32* __restrict__ b_m, int count, int sq, u32* __restrict__ out_m,
                const ModpConsts* __restrict__ cs) {
  __shared__ __attribute__((aligned(16))) u32 lds[NUMS_PER_WAVE * SLOT_WORDS];
  const Lane ln = make_lane();
  const int num = threadIdx.x >> 4;
  const int xi = blockIdx.x * NUMS_PER_WAVE + num;
  const bool live = xi < count;
  const int x = live ? xi : count - 1;
  u32* slot = lds + num * SLOT_WORDS;
  u32 n[LPL], a[LPL];
  load_lane_limbs(n, cs->n, ln);
  load_lane_limbs(a, a_m + (size_t)x * L, ln);
  if (sq) slot_store(slot, a, ln); else slot_fill_from_global(slot, b_m + (size_t)x * L, ln);
  __builtin_amdgcn_wave_barrier();
  if (sq) mont_sqr<MODP_N0INV_C>(a, a, slot, n, ln); else mont_mul<MODP_N0INV_C>(a, a, slot, n, ln);
  __builtin_amdgcn_wave_barrier();
  if (live) store_lane_limbs(out_m + (size_t)x * L, a, ln);
}

// out_m: (e1b ? 2 : 1) x count x 72 limbs, set 1 behind set 0
extern "C" int modp_launch_dual_exp_row(const uint32_t* tab1, size_t tab1_stride, const uint32_t* tab2, size_t tab2_stride, const uint8_t* e1,
                                        const uint8_t* e2, size_t e2_stride, int e2_windows, const uint8_t* e1b, const uint8_t* e2b, int count,
                                        uint32_t* out_m, const void* cs, hipStream_t s) {
  if (count <= 0) return 0;
  hipLaunchKernelGGL(k_modp_dual_exp_row, dim3((count + NUMS_PER_WAVE - 1) / NUMS_PER_WAVE, e1b ? 2 : 1), dim3(64), 0, s, tab1, tab1_stride,
                     tab2, tab2_stride, e1, e2, e2_stride, e2_windows, e1b, e2b, count, out_m, (const ModpConsts*)cs);
  return (int)hipGetLastError();
}

extern "C" int modp_launch_commit_eval_row_boxes(const uint32_t* cm, int t, const int64_t* positions, size_t box_positions, int count,
                                                 int boxes, uint32_t* x_m, size_t box_out, const int* gate, int gate_want, const void* cs,
                                                 hipStream_t s, int prio) {
  if (count <= 0 || boxes <= 0) return 0;
  hipLaunchKernelGGL(k_modp_commit_eval_row, dim3((count + NUMS_PER_WAVE - 1) / NUMS_PER_WAVE, boxes), dim3(64), 0, s, cm, t, positions,
                     count, x_m, gate, gate_want, (const ModpConsts*)cs, (size_t)t * L, box_positions, box_out, prio);
  return (int)hipGetLastError();
}

extern "C" int modp_launch_row_unit(const uint32_t* a_m, const uint32_t* b_m, int count, int sq, uint32_t* out_m, const void* cs,
                                    hipStream_t s) {
  if (count <= 0) return 0;
  hipLaunchKernelGGL(k_modp_row_unit, dim3((count + NUMS_PER_WAVE - 1) / NUMS_PER_WAVE), dim3(64), 0, s, a_m, b_m, count, sq, out_m,
                     (const ModpConsts*)cs);
  return (int)hipGetLastError();
}
