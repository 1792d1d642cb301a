// 2048-bit Montgomery product for gfx950 with the REDUCTION on the matrix cores ("pair" layout; the VALU-only "quad"
// layout is bn_quad.h).  Microbenchmark and exact integer model: tools/mfma_mont.
//
// Today's product (mpvss_rs_amd/csrc/bn_quad.h) interleaves a*b and m*N on the VALU: 36 v_mad_u64_u32 per row and lane, half
// of them (two thirds for a squaring) for m*N.  N is the same for every number, so "m * N" for many numbers at once IS a
// matrix product against a constant matrix -- the one part of this arithmetic the MFMA units can take:
//
//   phase A (VALU)  T = a * b in radix-2^29 columns, row by row; every row retires one exact limb of T_lo = T mod R
//                   (R = 2^2088) into LDS, the 72 high columns stay in 64-bit accumulators.
//   GEMM 1 (MFMA)   m' = T_lo * N' (mod R):  digits of T_lo (4 bytes per 29-bit limb: 8,8,8,5 bits -- the limb IS the packed
//                   operand word) against the constant matrix "signed digit (k,e) of (N' << (29 i + 8 f)) mod R".
//                   v_mfma_i32_32x32x32_i8, 32 numbers per wave on the N side, digit positions on the M side.
//   epilogue 1      four column sums per limb -> V_k (64-bit); v_k = (V_k mod 2^29) + (V_{k-1} >> 29) is again a valid
//                   operand word (top byte signed, |v_k| < 2^30): NO carry chain.  m'' = sum v_k 2^(29k) = m' (mod R), < R(1+2^-8).
//   GEMM 2 (MFMA)   the high limbs of m'' * N, limb-aligned by the constant matrix "digit (k,e) of N << (29 i + 8 f)", plus one
//                   guard limb (k = 71) that yields the carry out of the low half (which is = 0 mod R by construction).
//   epilogue 2      T_hi[k] += V2_k, one carry pass -> result a*b*R^-1 mod N in [0, 2N), almost normalised.
//
// Layout: a number is spread over lanes j and j + 32 of a wave ("pair"): lane half h holds limbs 36h .. 36h+35.  That is the
// MFMA's own split (operand and result halves live in lanes l and l+32), so the GEMM results land where the next product needs
// them; the rows of both constant matrices are ordered for that (tools/mfma_mont/model.py, which also proves every bound).
// Exactness: model.py runs the same integer pipeline (signed digits, C-init corrections, v_k, guard, bias) with assertions.
#pragma once
#ifndef MM_REDUCE_PRIO
#define MM_REDUCE_PRIO 1      // wave priority while a wave is in reduce(): measured +1-2 % in the headline pipeline (profiles/r03_pair_occupancy_ab.txt), inside the noise
#endif
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "modp_mfma_tables.h"


namespace mm {

typedef uint32_t u32;
typedef uint64_t u64;
typedef int64_t i64;
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int W = 29, L = 72, LP = 36;
constexpr int SLOTW = 76;                 // LDS words per number slot (72 limbs + pad: 16-byte aligned rows)
constexpr u32 MASK = (1u << W) - 1;
constexpr u32 XM = 0x00808080u;           // bytes 0..2 of an operand word are unsigned digits (minus 128 for the MFMA), byte 3 is signed

struct __attribute__((aligned(64))) Tables {   // constant, one copy in LDS per workgroup (17 KB)
  // 16-byte A-operand records of the two Toeplitz digit matrices: gt[e][x], x = (row limb) - (8 C + 4 h') + 4, bytes
  // [4 j' + f] = digit e of limb x - 4 - j' of (N' or N) << 8 f  (tools/mfma_mont/model.py checks every tile and lane)
  v4i gt1[4 * MM_GT1_X];
  v4i gt2[4 * MM_GT2_X];
  v16i c1[9 * 2];                         // C-init of tile R for lane half h: [2 R + h], register 4 e + g
  v16i c2[9 * 2];
};

struct PairLane {
  u32 h;        // lane half: 0 holds limbs 0..35, 1 holds 36..71
  u32 m0, m1;   // all ones in lane half 0 / 1
  u32 lane;     // 0..63
  int k16, one; // 65536 and 1 in VGPRs, opaque to the optimiser: "x * k16 + y" stays one v_mad_i64_i32
  // this lane's A-operand record of tile (R, C): GEMM 1 at rec1 + 8 (R - C), GEMM 2 at rec2 + (4 R - 8 C + 64) (tile 8: rec2g,
  // whose row rho = 71 is the guard limb 71 instead of limb 143)
  u32 rec1, rec2, rec2g;
};

__device__ __forceinline__ PairLane make_pair_lane() {
  PairLane pl;
  pl.lane = threadIdx.x & 63;
  pl.h = pl.lane >> 5;
  pl.m1 = pl.h ? 0xffffffffu : 0u;
  pl.m0 = ~pl.m1;
  pl.k16 = 65536;
  pl.one = 1;
  {
    const u32 row = pl.lane & 31, g = row & 3, hr = (row >> 2) & 1, e = row >> 3, hp = pl.lane >> 5;
    pl.rec1 = e * MM_GT1_X + 4 * hr + g - 4 * hp + 4;
    pl.rec2 = e * MM_GT2_X + 76 + 36 * hr + g - 4 * hp - 64;
    pl.rec2g = pl.rec2 - ((hr == 1 && g == 3) ? 72u : 0u);
  }
  asm volatile("" : "+v"(pl.m0), "+v"(pl.m1), "+v"(pl.k16), "+v"(pl.one));
  return pl;
}

// v_permlane32_swap: lanes 32..63 of `a` <-> lanes 0..31 of `b`
__device__ __forceinline__ void swap32(u32& a, u32& b) {
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  a = r[0];
  b = r[1];
}

// r = a * b * R^-1 mod N (almost normalised, < 2N), in two steps so that a kernel which both squares and multiplies
// holds ONE copy of the reduction (two copies of everything are 47 KB of straight-line code: measured 12x slower, the
// instruction cache thrashes):
//   phase_a<SQ>(T, a, slot, junk, pl)   T = a * b (b in the LDS slot; SQ: b is a copy of a); T_lo goes to the slot
//   reduce(r, T, slot, tb, pl)          the two GEMMs, their epilogues and the final carry pass
//   a     : this lane's 36 limbs of the first operand
//   slot  : LDS slot of THIS number holding the 72 limbs of b (SQ: a copy of a); it is overwritten with T_lo's operand words
//   junk  : 72 LDS words any lane of half 1 may scribble on
//   bsrc  : where b is read from -- the number's own slot, or (forward-difference stepping) the slot of the NEXT number of the
//           wave: every lane reads b two 16-byte chunks ahead of the row whose retired limb it writes, and the lanes of a wave
//           run in step, so a neighbour's slot is read before its owner overwrites it
template <bool SQ>
__device__ __forceinline__ void phase_a(u64 (&T)[LP], const u32 (&a)[LP], u32* slot, u32* junk, const PairLane& pl, const u32* bsrc) {
#pragma unroll
  for (int k = 0; k < LP; ++k) T[k] = 0;
  u32* xs = pl.h ? junk : slot;           // where this lane's retired limb goes (only half 0 retires limbs of T_lo)
  // b is read four limbs at a time, two chunks (8 rows) ahead: the late rows of a squaring are short (a handful of mads),
  // a single-limb prefetch one row ahead would expose the LDS latency there
  v4i bq[3];
  bq[0] = *reinterpret_cast<const v4i*>(bsrc);
  bq[1] = *reinterpret_cast<const v4i*>(bsrc + 4);
  // ---------------- phase A: T = a * b, one limb of b per row --------------------------------------------------------
  // Retiring a column is a chain of dependent instructions; it is written BETWEEN the products of the next row (all but
  // the one into the fresh column), so that the compiler can interleave the two: the late rows of a squaring have only a
  // handful of products and two waves per SIMD do not hide a serial chain per row.
  auto retire = [&](int rr, int i) {
    // half 0's lowest column is a finished limb of T_lo (exact 29 bits), half 1's moves down to half 0
    const u64 ret = T[rr];
    T[(rr + 1) % LP] += ret >> W;
    u32 low = (u32)ret & MASK;
    xs[i] = low ^ XM;
    // half 1's fresh top column is column 72 + i: it starts at the bias beta_i (2^47 on limb 0, 2^47 - 2^18 on limbs
    // 1..70: the total is 2^18 on limb 71, which the final pass drops) so that the signed GEMM-2 sums never take a column
    // below zero
    const u32 s_lo = (i == 0 || i == L - 1) ? 0u : 0xFFFC0000u;
    const u32 s_hi = i == 0 ? 0x8000u : (i == L - 1 ? 0u : 0x7FFFu);
    u32 z = s_lo & pl.m1;
    swap32(low, z);                                  // z = [half 0: half 1's low | half 1: beta_lo]
    T[rr] = ((u64)(s_hi & pl.m1) << 32) | z;
  };
#pragma nounroll
  for (int o = 0; o < 2; ++o) {
#pragma unroll
    for (int rr = 0; rr < LP; ++rr) {
      const int i = o * LP + rr;
      if ((rr & 3) == 0) {
        const int nxt = i / 4 + 2;
        bq[(rr / 4 + 2) % 3] = *reinterpret_cast<const v4i*>(bsrc + 4 * (nxt < L / 4 ? nxt : L / 4 - 1));
      }
      const u32 bi = (u32)bq[(rr / 4) % 3][rr & 3];
      const u32 bi2 = bi << 1;
      // SQ: row r = 36 o + rr visits the local positions k >= rr, doubled above the diagonal (see bn_quad.h)
#pragma unroll
      for (int k = SQ ? rr : 0; k < LP - 1; ++k) T[(k + rr) % LP] += (u64)a[k] * ((SQ && k > rr) ? bi2 : bi);
      if (rr > 0) retire(rr - 1, i - 1);
      T[(LP - 1 + rr) % LP] += (u64)a[LP - 1] * ((SQ && LP - 1 > rr) ? bi2 : bi);
#pragma unroll
      for (int k = 0; k < LP; ++k) asm volatile("" : "+v"(T[k]));   // keep the row-wise order (see bn_quad.h)
    }
    retire(LP - 1, o * LP + LP - 1);
#pragma unroll
    for (int k = 0; k < LP; ++k) asm volatile("" : "+v"(T[k]));
  }
  asm volatile("" ::: "memory");
}

// N groups of [1 MFMA, V VALU instructions] for the scheduling region that ends here
template <int N, int V, bool FIRST = true>
__device__ __forceinline__ void sgb_mfma_first() {
}
constexpr int mm_row_mfmas2(int R) {
  int n = 0;
  for (int C = 0; C < 9; ++C) n += MM_SKIP2[9 * R + C] ? 0 : 1;
  return n;
}

struct NoHook { __device__ __forceinline__ void operator()() const {} };

// slot_free(): called once the reduction holds T_lo in registers, i.e. from the moment the number's LDS slot is free again --
// a kernel that knows its NEXT operand starts the LDS-DMA of it there, under the two GEMMs (modp_pair_kernels.hip)
// PRIO_IN / PRIO_OUT: wave priority inside the two GEMMs / afterwards (the latency-bound stepping waves keep theirs throughout)
template <class Hook = NoHook, int PRIO_IN = MM_REDUCE_PRIO, int PRIO_OUT = 0>
__device__ __forceinline__ void reduce(u32 (&r)[LP], u64 (&T)[LP], const u32* slot, const Tables* tb, const PairLane& pl,
                                       Hook&& slot_free = NoHook()) {
  __builtin_amdgcn_sched_barrier(0);
#if MM_REDUCE_PRIO > 0
  __builtin_amdgcn_s_setprio(PRIO_IN);          // the wave that is in its MFMA chains issues ahead of the one in its VALU rows
#endif
  // ---------------- GEMM 1: m'' = T_lo * N' (mod R) --------------------------------------------------------------------
  // Software pipeline, one stage per row tile: the (dependent) MFMA chain of tile R runs on the matrix pipe while the VALU
  // does the epilogue of tile R-1; a scheduling barrier between the stages keeps the compiler from unrolling the whole GEMM
  // into eight accumulators at once (it does, and then spills).
  // T_lo and m'' stay in registers (2 x 36): taking the data operand of every MFMA from the LDS slot instead was built and
  // measured (170 instead of 197 VGPRs, but 4.2 instead of 4.8 G squarings/s, and 0.96 instead of 1.03 M share
  // verifications/s in the pipeline: profiles/r03_pair_ab.txt).
  v4i xw[9], mw[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) xw[c] = *reinterpret_cast<const v4i*>(slot + 8 * c + 4 * pl.h);
  if constexpr (!std::is_same<typename std::decay<Hook>::type, NoHook>::value) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the slot has been read: it may be overwritten from here on
    slot_free();
    __builtin_amdgcn_sched_barrier(0);
  }
  const v16i* c1 = tb->c1 + pl.h;
  const v16i* c2 = tb->c2 + pl.h;
  u32 saved = 0;                                       // half 0: hi of limb 8R-1 (from half 1, previous tile)
  v16i acc[2];
  auto chain1 = [&](int R, v16i& ac) {
    ac = c1[2 * R];
#pragma unroll
    for (int C = 0; C <= R; ++C) {
      const v4i A = tb->gt1[pl.rec1 + 8 * (R - C)];
      ac = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, xw[C], ac, 0, 0, 0);
    }
  };
  auto epi1 = [&](int R, const v16i& ac) {
    u32 lo[4], hi[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int p01 = ac[g] + (ac[4 + g] << 8), p23 = ac[8 + g] + (ac[12 + g] << 8);     // register 4 e + g: byte e of limb g
      const i64 V = (i64)p23 * pl.k16 + (i64)p01;          // v_mad_i64_i32 (the constant is opaque: no 64-bit shift-and-add)
      lo[g] = (u32)V & MASK;
      hi[g] = __builtin_amdgcn_alignbit((u32)((u64)V >> 32), (u32)V, W);     // low 32 bits of V >> 29 (|V| < 2^47)
    }
    u32 A3 = hi[3], B3 = hi[3];
    swap32(A3, B3);                                    // A3 = half 0's hi[3] in both halves, B3 = half 1's
    const u32 prev = (A3 & pl.m1) | (saved & pl.m0);
    saved = B3;
    v4i m;
    m[0] = (int)((lo[0] + prev) ^ XM);
    m[1] = (int)((lo[1] + hi[0]) ^ XM);
    m[2] = (int)((lo[2] + hi[1]) ^ XM);
    m[3] = (int)((lo[3] + hi[2]) ^ XM);
    mw[R] = m;
  };
  constexpr int EPI1_VALU = 38, EPI2_VALU = 22;         // VALU instructions of one epilogue (ISA count), spread over the stage's MFMAs
#define MM_FETCH1(R)
#define MM_FETCH2(R)
  chain1(0, acc[0]);
  __builtin_amdgcn_sched_barrier(0);
#define MM_STAGE1(R)                                                  \
  chain1(R, acc[(R) & 1]);                                            \
  epi1((R) - 1, acc[((R) - 1) & 1]);                                  \
  MM_FETCH1(R)                                                        \
  sgb_mfma_first<(R) + 1, (EPI1_VALU + (R)) / ((R) + 1)>();           \
  __builtin_amdgcn_sched_barrier(0);
  MM_STAGE1(1) MM_STAGE1(2) MM_STAGE1(3) MM_STAGE1(4) MM_STAGE1(5) MM_STAGE1(6) MM_STAGE1(7) MM_STAGE1(8)
#undef MM_STAGE1
  epi1(8, acc[0]);
  __builtin_amdgcn_sched_barrier(0);
  // ---------------- GEMM 2: high limbs of m'' * N, and the carry out of the low half -----------------------------------
  auto chain2 = [&](int R, v16i& ac) {
    ac = c2[2 * R];
#pragma unroll
    for (int C = 0; C < 9; ++C) {
      if (MM_SKIP2[9 * R + C]) continue;
      const v4i A = tb->gt2[(R == 8 ? pl.rec2g : pl.rec2) + (4 * R - 8 * C + 64)];
      ac = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, mw[C], ac, 0, 0, 0);
    }
  };
  auto epi2 = [&](int R, const v16i& ac) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int p01 = ac[g] + (ac[4 + g] << 8), p23 = ac[8 + g] + (ac[12 + g] << 8);
      if (R == 8 && g == 3) {
        // half 0: result limb 35 as usual; half 1: the guard limb (absolute limb 71) -> carry of the low half into limb 0
        const i64 V = (i64)p23 * pl.k16 + (i64)p01;
        T[35] += ((u64)((u32)((u64)V >> 32) & pl.m0) << 32) | ((u32)V & pl.m0);
        const u32 x71 = (u32)xw[8][3] ^ XM;
        const i64 cs = V + (i64)x71 + ((i64)1 << 28);
        u32 c32 = __builtin_amdgcn_alignbit((u32)((u64)cs >> 32), (u32)cs, W);
        u32 y = 0;
        swap32(c32, y);                                // y (half 0) = half 1's carry
        T[0] += (u64)(y & pl.m0);
      } else {
        i64 t = (i64)T[4 * R + g];
        t = (i64)p23 * pl.k16 + t;                     // v_mad_i64_i32
        asm volatile("" : "+v"(t));
        t = (i64)p01 * pl.one + t;                     // v_mad_i64_i32 (a sign-extending 64-bit add would be three instructions)
        T[4 * R + g] = (u64)t;
      }
    }
  };
  // tile 8 first: it carries the guard limb, whose carry goes into limb 0
  chain2(8, acc[0]);
  __builtin_amdgcn_sched_barrier(0);
#define MM_STAGE2(R)                                                                   \
  chain2(R, acc[((R) + 1) & 1]);                                                       \
  epi2((R) == 0 ? 8 : (R) - 1, acc[(R) & 1]);                                          \
  MM_FETCH2(R)                                                                         \
  sgb_mfma_first<mm_row_mfmas2(R), (EPI2_VALU + mm_row_mfmas2(R) - 1) / mm_row_mfmas2(R)>(); \
  __builtin_amdgcn_sched_barrier(0);
  MM_STAGE2(0) MM_STAGE2(1) MM_STAGE2(2) MM_STAGE2(3) MM_STAGE2(4) MM_STAGE2(5) MM_STAGE2(6) MM_STAGE2(7)
#undef MM_STAGE2
#undef MM_FETCH1
#undef MM_FETCH2
  epi2(7, acc[0]);
  __builtin_amdgcn_sched_barrier(0);
#if MM_REDUCE_PRIO > 0
  __builtin_amdgcn_s_setprio(PRIO_OUT);
#endif
  // ---------------- final pass: carry-propagate inside the half, hand half 0's carry to half 1 -----------------------
  u64 c = 0;
#pragma unroll
  for (int k = 0; k < LP; ++k) {
    const u64 v = T[k] + c;
    r[k] = (u32)v & MASK;
    c = v >> W;
  }
  r[LP - 1] &= pl.m0;                                  // limb 71 is zero (what sits there in half 1 is the bias total)
  u32 cl = (u32)c, ch = (u32)(c >> 32), zl = 0, zh = 0;
  swap32(zl, cl);                                      // zl (half 1) = half 0's carry-out (low word)
  swap32(zh, ch);
  const u64 cin = ((u64)(zh & pl.m1) << 32) | (zl & pl.m1);
  const u64 v0 = (u64)r[0] + cin;
  r[0] = (u32)v0 & MASK;
  r[1] += (u32)(v0 >> W);
}

template <bool SQ>
__device__ __forceinline__ void phase_a(u64 (&T)[LP], const u32 (&a)[LP], u32* slot, u32* junk, const PairLane& pl) {
  phase_a<SQ>(T, a, slot, junk, pl, slot);
}
template <bool SQ>
__device__ __forceinline__ void mont_pair(u32 (&r)[LP], const u32 (&a)[LP], u32* slot, u32* junk, const Tables* tb,
                                          const PairLane& pl) {
  u64 T[LP];
  phase_a<SQ>(T, a, slot, junk, pl);
  reduce(r, T, slot, tb, pl);
}

}  // namespace mm
