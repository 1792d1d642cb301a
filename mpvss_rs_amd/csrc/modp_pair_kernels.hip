// MODP-2048 kernels in the "pair" layout: the Montgomery REDUCTION runs on the matrix cores (bn_pair.h).
//
//   reference:  DLEQ verifier commitment a2 = y^r * Y^c   src/dleq.rs:79-81 (two ModpGroup::exp, one ::mul,
//               src/groups/modp.rs:122-132), the dominant kernel of verify_distribution_shares (src/participant.rs:399-455)
//
// Why a second layout.  bn_quad.h spends 36 v_mad_u64_u32 per row and lane, 18 of them (of 27.5 for a squaring) on m*N.
// N never changes, so m*N over the 32 numbers of a wave is a matrix product against a constant matrix: two int8 GEMMs on
// v_mfma_i32_32x32x32_i8 (T_lo*N' mod R, then m*N), 114 MFMAs per product, beside 18.5 (squaring) / 36 (product) VALU
// mads per row for a*b.  Measured on MI355X (tools/mfma_mont/ubench): 5.0 G squarings/s against 3.7-3.8 G/s for the
// VALU-only chain, bit-exact.  The instruction mix, the value bounds and the constant matrices come from the exact integer
// model tools/mfma_mont/model.py.
//
// Layout: one wave = 32 numbers; a number lives in lanes j and j+32 (limbs 36h .. 36h+35 in lane half h).  A workgroup is
// PAIR_WAVES waves that share one copy of the constant digit records in LDS (17 KB) and own one operand slot per number each.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "bn_pair.h"
#include "modp_kernels.h"

#ifndef PAIR_WAVES
#define PAIR_WAVES 2            // waves per workgroup: they share one copy of the constant records (17 KB) and own 9 KB of operand
                                // slots each, so two waves make 36 KB and a CU holds four such workgroups = TWO waves per SIMD
                                // (one-wave workgroups of 27 KB: five per CU).  Measured in the headline pipeline with the a2
                                // kernel's registers allocated for two waves per SIMD (PAIR_WAVES_PER_EU below; interleaved runs,
                                // profiles/r03_pair_occupancy_ab.txt): 1.03 / 1.08 / 1.02 M share verifications/s for 1 / 2 / 4
                                // waves, 1.02 M with one wave per SIMD.  (At one wave per SIMD bigger workgroups LOSE, 0.91 /
                                // 0.98 / 1.04 M for 4 / 2 / 1: they only start on CUs the single-wave workgroups of the other
                                // kernels have left empty, profiles/r03_pair_ab.txt.)
#endif

// Waves per SIMD the register allocator aims for.  The straight-line product keeps 230-256 arch VGPRs live and the MFMA
// accumulators take 32 AGPRs more: left alone the kernels end up at 260-294 registers, ONE wave per SIMD (the unified file
// holds 512 per lane).  Two waves -- 256 registers in all -- let one wave's VALU rows run under the other's MFMA chains and
// LDS round trips.  Applied where it costs no spills (a2 = y^r Y^c: 240 registers, headline pipeline 1.02 -> 1.08 M share
// verifications/s although the launch alone on the chip gets slower, 43 -> 54 ms); the table builder and the twin
// exponentiation would spill (the dealer loses 8 %) and keep their one wave (profiles/r03_pair_occupancy_ab.txt).
#ifndef PAIR_WAVES_PER_EU
#define PAIR_WAVES_PER_EU 2
#endif
#if PAIR_WAVES_PER_EU > 0
#define PAIR_OCC_ATTR __attribute__((amdgpu_waves_per_eu(PAIR_WAVES_PER_EU, PAIR_WAVES_PER_EU)))
#else
#define PAIR_OCC_ATTR
#endif

namespace {

using namespace mm;

struct ModpConsts {             // same layout as in modp_kernels.hip: N, R^2 mod N, R mod N, plain 1
  u32 n[L];
  u32 r2[L];
  u32 one_m[L];
  u32 one[L];
};

#ifndef PAIR_LDS_PAD
#define PAIR_LDS_PAD 0          // bytes of unused LDS per workgroup: lowers the number of workgroups a CU holds (tuning)
#endif

struct PairShared {
  Tables tb;
  __attribute__((aligned(16))) u32 slots[PAIR_WAVES][32 * SLOTW];
  u32 junk[PAIR_WAVES][L];
#if PAIR_LDS_PAD > 0
  u32 pad[PAIR_LDS_PAD / 4];
#endif
};

__device__ __forceinline__ void tables_to_lds(Tables* dst, const Tables* __restrict__ src) {
  const uint4* s4 = reinterpret_cast<const uint4*>(src);
  uint4* d4 = reinterpret_cast<uint4*>(dst);
  for (int i = threadIdx.x; i < (int)(sizeof(Tables) / 16); i += blockDim.x) d4[i] = s4[i];
  __syncthreads();
}

// this lane's half (36 limbs = 9 x 16 bytes) of a 72-limb number in global memory -> the number's LDS slot
__device__ __forceinline__ void slot_fill_pair(u32* slot, const u32* __restrict__ g, const PairLane& pl) {
  const uint4* g4 = reinterpret_cast<const uint4*>(g + LP * pl.h);
  uint4* s4 = reinterpret_cast<uint4*>(slot + LP * pl.h);
#pragma unroll
  for (int c = 0; c < LP / 4; ++c) s4[c] = g4[c];
}

__device__ __forceinline__ void slot_store_pair(u32* slot, const u32 (&a)[LP], const PairLane& pl) {
  uint4* s4 = reinterpret_cast<uint4*>(slot + LP * pl.h);
#pragma unroll
  for (int c = 0; c < LP / 4; ++c) s4[c] = make_uint4(a[4 * c], a[4 * c + 1], a[4 * c + 2], a[4 * c + 3]);
}

__device__ __forceinline__ void load_pair_limbs(u32 (&a)[LP], const u32* __restrict__ g, const PairLane& pl) {
  const uint4* g4 = reinterpret_cast<const uint4*>(g + LP * pl.h);
#pragma unroll
  for (int c = 0; c < LP / 4; ++c) {
    const uint4 v = g4[c];
    a[4 * c] = v.x; a[4 * c + 1] = v.y; a[4 * c + 2] = v.z; a[4 * c + 3] = v.w;
  }
}

// plain almost-normalised value < 2N in `a` -> canonical residue in [0, N) as 256 big-endian bytes (as
// store_canonical_be256 of modp_kernels.hip: exact carry propagation, one conditional subtraction, done by lane half 0)
__device__ __forceinline__ void store_canonical_pair(uint8_t* __restrict__ out, const u32 (&a)[LP], u32* slot,
                                                     const ModpConsts* __restrict__ cs, const PairLane& pl, bool write) {
  slot_store_pair(slot, a, pl);
  __builtin_amdgcn_wave_barrier();
  if (pl.h == 0) {
    u32 c = 0;
#pragma nounroll
    for (int j = 0; j < L; ++j) {
      const u32 v = slot[j] + c;
      slot[j] = v & MASK;
      c = v >> W;
    }
    int ge = 1;
#pragma nounroll
    for (int j = L - 1; j >= 0; --j) {
      const u32 x = slot[j], y = cs->n[j];
      if (x != y) { ge = x > y; break; }
    }
    if (ge) {
      u32 borrow = 0;
#pragma nounroll
      for (int j = 0; j < L; ++j) {
        const u32 d = slot[j] - cs->n[j] - borrow;
        borrow = (d >> 31) & 1;
        slot[j] = d & MASK;
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
  if (write) {
    // lane half h emits the little-endian 32-bit words 32h .. 32h+31 (byte-swapped, mirrored position)
    u32* out32 = reinterpret_cast<u32*>(out);
#pragma nounroll
    for (int i = 0; i < 32; ++i) {
      const int wd = (int)pl.h * 32 + i;
      const int bit = 32 * wd;
      const int j = bit / W, s = bit % W;
      u64 two = (u64)slot[j] | ((u64)(j + 1 < L ? slot[j + 1] : 0u) << W);
      two >>= s;
      if (2 * W - s < 32) two |= (u64)(j + 2 < L ? slot[j + 2] : 0u) << (2 * W - s);
      out32[63 - wd] = __builtin_bswap32((u32)two);
    }
  }
  __builtin_amdgcn_wave_barrier();
}

}  // namespace

// ---------------------------------------------------------------------------------------
// a2 = y^r * Y^c (dleq.rs:79-81): the same left-to-right schedule as k_modp_dual_exp_w6 (modp_kernels.hip) -- 6-bit
// windows of r against the 64-entry table of y, the windows of c (fixed 4-bit, or the host's sliding-window schedule when the
// box has ONE challenge) against the table of Y -- on the pair layout.  tab1 [count][64][72], tab2 [count][16][72] in
// Montgomery limb form as the quad kernels build them (the limb order in HBM does not depend on the layout).
// ---------------------------------------------------------------------------------------

namespace {
// The next product's operands of the wave's 32 numbers, HBM -> LDS slots, without registers (global_load_lds_dwordx4).  One
// DMA instruction moves 64 x 16 bytes to CONSECUTIVE LDS addresses, and the wave's 32 slots of SLOTW = 76 words are 608
// consecutive 16-byte pieces (19 per number: 18 of data, one of padding), so lane l of instruction i fetches piece 64 i + l:
// piece q = (64 i + l) % 19 of number n = (64 i + l) / 19, whose table entry it learns from lane n by a shuffle.
//   src of number n = base + (x0 + n, clamped to count - 1) * num_stride + entry_n * 72 words;   entry: this lane's OWN number's
//   entry == PREFETCH_ALT: that number takes the 72 words at `alt` instead (a constant operand: the same for every number)
constexpr u32 PREFETCH_ALT = 0xffffffffu;
__device__ __forceinline__ void slots_prefetch_pair(u32* wave_slots, const u32* __restrict__ base, size_t num_stride, int x0, int count,
                                                    u32 entry, const PairLane& pl, const u32* __restrict__ alt = nullptr) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    int p = (int)pl.lane + 64 * i;
    asm volatile("" : "+v"(p));          // (recomputed per call: hoisted out of the main loop these addresses would be 20 registers, i.e. spills)
    const bool valid = p < 32 * (SLOTW / 4);
    const int n = valid ? (int)(((u32)p * 3450u) >> 16) : 31;        // p / 19 for p < 608
    const int q = p - 19 * n;
    const u32 e = (u32)__shfl((int)entry, n);
    const int xj = (x0 + n < count) ? x0 + n : count - 1;
    const u32* src = base + (size_t)xj * num_stride + (size_t)e * L + (q < 18 ? q : 17) * 4;
    if (alt != nullptr && e == PREFETCH_ALT) src = alt + (q < 18 ? q : 17) * 4;
    if (valid)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(uintptr_t)src,
                                       (__attribute__((address_space(3))) void*)(uintptr_t)(wave_slots + i * 256), 16, 0, 0);
  }
}
// where the operand of a kernel's NEXT product lives: number n of the wave reads base + (x0 + n) * stride + ent_n * 72 words
struct PairNext {
  const u32* base = nullptr;     // null: the next operation takes nothing from HBM (a squaring, a register operand)
  size_t stride = 0;
  u32 ent = 0;
};
}  // namespace

// (A register cap on this kernel -- 224 / 216 / 208 instead of the 240 two waves per SIMD allow, so that a wave of a box's own X
// path fits beside two a2 waves on a SIMD -- was measured in the pipeline in round 5 (nothing, profiles/r05_a2_vgpr_cap_ab.txt) and
// for a box that has the chip to itself in round 6 (a twin of the kernel capped at 224 / 208: the call gets SLOWER, 128 against 122 ms:
// co-resident X-path waves take issue slots the a2 waves would have used -- the box is bound by the sum of its work, not by who
// waits for whom; profiles/r06_lone_box_schedule_ab.txt).  Not in the source any more.)
extern "C" __global__ void __launch_bounds__(64 * PAIR_WAVES) PAIR_OCC_ATTR
k_modp_dual_exp_w6_pair(const u32* __restrict__ tab1, const u32* __restrict__ tab2, const uint8_t* __restrict__ e1_be,
                        const uint8_t* __restrict__ c_all, size_t c_stride, int count, uint8_t* __restrict__ out_be,
                        const ModpConsts* __restrict__ cs, const uint16_t* __restrict__ c_sched,
                        const Tables* __restrict__ gtab) {
  __shared__ PairShared sh;
  tables_to_lds(&sh.tb, gtab);
  const PairLane pl = make_pair_lane();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // (wave-uniform values stay in scalar registers)
  const int x0 = (blockIdx.x * PAIR_WAVES + wave) * 32;
  const int xi = x0 + (int)(pl.lane & 31);
  const bool live = xi < count;
  const int x = live ? xi : count - 1;
  u32* wslots = sh.slots[wave];
  u32* slot = &wslots[(pl.lane & 31) * SLOTW];
  u32* junk = sh.junk[wave];
  const Tables* tb = &sh.tb;
  u32 acc[LP];
  // (per-number pointers are formed from x where they are used: the product keeps 230 registers live, and every 64-bit
  // pointer carried across it is two more)
  auto digit6 = [&](int w) -> u32 {
    const uint8_t* e1 = e1_be + (size_t)x * 256;
    const int o = 6 * w, k = o >> 3;
    const u32 lo = e1[255 - k];
    const u32 hi = (k + 1 < 256) ? e1[254 - k] : 0u;
    return ((lo | (hi << 8)) >> (o & 7)) & 63u;
  };
  load_pair_limbs(acc, tab1 + ((size_t)x * 64 + digit6(341)) * L, pl);
  // The schedule: from weight cur = 2046 down, per bit one squaring, then (cur % 6 == 0) the product with y^digit from tab1, then
  // the product with Y^digit from tab2 where the schedule of c (or its fixed 4-bit windows) has one; at the end the product with
  // plain 1 that leaves the Montgomery domain.  next_op() steps that state machine to the next operation that is not skipped:
  //   kind 0 squaring; 1 product with entry `ent` of tab1; 2 with entry `ent` of tab2; 3 the closing product with cs->one
  int cur = 2046, s = 0, si = 0;
  const int sn = c_sched ? (int)c_sched[0] : 0;
  struct Op { int kind; u32 ent; };
  auto next_op = [&]() -> Op {
    while (true) {
      Op op{-1, 0u};
      if (s == 0) {
        --cur;
        op.kind = 0;
      } else if (s == 1) {
        if (cur % 6 == 0) { op.kind = 1; op.ent = digit6(cur / 6); }
      } else if (s == 2) {
        if (c_sched != nullptr) {
          if (si < sn && cur == (int)c_sched[1 + 2 * si]) {
            op.kind = 2;
            op.ent = c_sched[2 + 2 * si];
            ++si;
          }
        } else if ((cur & 3) == 0 && cur < 256 && c_all != nullptr) {
          const u32 byte = c_all[(size_t)x * c_stride + 255 - (cur >> 3)];
          op.kind = 2;
          op.ent = (cur & 4) ? (byte >> 4) : (byte & 15);
        }
      } else {
        op.kind = 3;
      }
      if (s != 3) s = (s == 2) ? (cur == 0 ? 3 : 0) : s + 1;
      if (op.kind >= 0) return op;
    }
  };
  auto source = [&](const Op& op, const u32*& base, size_t& stride) {       // table of a product: base of number 0, words per number
    if (op.kind == 1) { base = tab1; stride = (size_t)64 * L; }
    else if (op.kind == 2) { base = tab2; stride = (size_t)16 * L; }
    else { base = cs->one; stride = 0; }
  };
  Op op = next_op();
  bool fetched = false;              // the operand of `op` is already on its way into the slots
  while (true) {
    Op nx{-1, 0u};
    if (op.kind != 3) nx = next_op();
    u64 T[LP];
    if (op.kind == 0) {
      slot_store_pair(slot, acc, pl);
      __builtin_amdgcn_wave_barrier();
      phase_a<true>(T, acc, slot, junk, pl);
    } else {
      const u32* base;
      size_t stride;
      source(op, base, stride);
      if (fetched) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        slot_fill_pair(slot, base + (size_t)x * stride + (size_t)op.ent * L, pl);
      }
      __builtin_amdgcn_wave_barrier();
      phase_a<false>(T, acc, slot, junk, pl);
    }
    u32 r[LP];
    reduce(r, T, slot, tb, pl);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < LP; ++k) acc[k] = r[k];
    if (op.kind == 3) break;
    op = nx;
  }
  store_canonical_pair(out_be + (size_t)x * 256, acc, slot, cs, pl, live);
}

// ---------------------------------------------------------------------------------------
// One Montgomery operation of a kernel's main loop.  Every kernel below calls this from ONE place: the straight-line code of
// a product (12 KB), of a squaring (9 KB) and of the reduction (10 KB) must exist once per kernel, or the loop outgrows the
// instruction cache.  sq: acc = acc^2; else acc = acc * b with b = `fill` (72 limbs in global memory) or, if fill is null,
// the 36 limbs per lane in `breg`.
// ---------------------------------------------------------------------------------------
namespace {
template <bool HAS_SQ>
__device__ __forceinline__ void pair_step(u32 (&acc)[LP], bool sq, const u32* fill, const u32 (&breg)[LP], u32* slot, u32* junk,
                                          const Tables* tb, const PairLane& pl) {
  u64 T[LP];
  if (HAS_SQ && sq) {
    slot_store_pair(slot, acc, pl);
    __builtin_amdgcn_wave_barrier();
    phase_a<true>(T, acc, slot, junk, pl);
  } else {
    if (fill != nullptr) slot_fill_pair(slot, fill, pl); else slot_store_pair(slot, breg, pl);
    __builtin_amdgcn_wave_barrier();
    phase_a<false>(T, acc, slot, junk, pl);
  }
  u32 r[LP];
  reduce(r, T, slot, tb, pl);
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int k = 0; k < LP; ++k) acc[k] = r[k];
}

// The same for kernels whose products take their second operand from HBM: `fetched` says that this operation's operand is
// already on its way into the slots (an LDS-DMA started under the previous reduction), `nx` names the next one's, whose DMA
// starts as soon as this reduction has the slot's contents in registers.  A squaring (sq) needs neither.
template <bool HAS_SQ>
__device__ __forceinline__ void pair_step_pf(u32 (&acc)[LP], bool sq, const u32* fill, bool fetched, const PairNext& nx, u32* wslots,
                                             int x0, int count, u32* slot, u32* junk, const Tables* tb, const PairLane& pl) {
  u64 T[LP];
  if (HAS_SQ && sq) {
    slot_store_pair(slot, acc, pl);
    __builtin_amdgcn_wave_barrier();
    phase_a<true>(T, acc, slot, junk, pl);
  } else {
    if (fetched) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else slot_fill_pair(slot, fill, pl);
    __builtin_amdgcn_wave_barrier();
    phase_a<false>(T, acc, slot, junk, pl);
  }
  u32 r[LP];
  reduce(r, T, slot, tb, pl);
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int k = 0; k < LP; ++k) acc[k] = r[k];
}

// limb j (W bits at bit offset W j) of a 256-byte big-endian integer (as be256_limb of modp_kernels.hip)
__device__ __forceinline__ u32 be256_limb_pair(const uint8_t* __restrict__ be, int j) {
  const int o = W * j;
  const int p = o >> 3, sft = o & 7;
  u64 w = 0;
#pragma unroll
  for (int t = 0; t < 5; ++t) {
    const int idx = 255 - (p + t);
    if (idx >= 0) w |= (u64)be[idx] << (8 * t);
  }
  return (u32)(w >> sft) & MASK;
}

__device__ __forceinline__ void load_be256_pair(u32 (&a)[LP], const uint8_t* __restrict__ be, const PairLane& pl) {
#pragma unroll
  for (int k = 0; k < LP; ++k) a[k] = be256_limb_pair(be, (int)pl.h * LP + k);
}

__device__ __forceinline__ void store_pair_limbs(u32* __restrict__ g, const u32 (&a)[LP], const PairLane& pl) {
  uint4* g4 = reinterpret_cast<uint4*>(g + LP * pl.h);
#pragma unroll
  for (int c = 0; c < LP / 4; ++c) g4[c] = make_uint4(a[4 * c], a[4 * c + 1], a[4 * c + 2], a[4 * c + 3]);
}

struct PairCtx {          // what every pair kernel sets up the same way
  PairLane pl;
  int x, x0;              // this lane's number (clamped to count - 1), the first number of its wave
  bool live;
  u32* slot;
  u32* wslots;            // the wave's 32 slots
  u32* junk;
  const Tables* tb;
};
}  // namespace

#define PAIR_KERNEL_PROLOGUE(gtab, count)                                                  \
  __shared__ PairShared sh;                                                                \
  tables_to_lds(&sh.tb, gtab);                                                             \
  PairCtx pc;                                                                              \
  pc.pl = make_pair_lane();                                                                \
  {                                                                                        \
    const int wave_ = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);                    \
    pc.x0 = (blockIdx.x * PAIR_WAVES + wave_) * 32;                                        \
    const int xi_ = pc.x0 + (int)(pc.pl.lane & 31);                                        \
    pc.live = xi_ < (count);                                                               \
    pc.x = pc.live ? xi_ : (count)-1;                                                      \
    pc.wslots = sh.slots[wave_];                                                           \
    pc.slot = &pc.wslots[(pc.pl.lane & 31) * SLOTW];                                       \
    pc.junk = sh.junk[wave_];                                                              \
    pc.tb = &sh.tb;                                                                        \
  }

// ---------------------------------------------------------------------------------------
// Window tables of per-share bases (part of ModpGroup::exp, modp.rs:122-128), pair layout.  Same HBM format as the quad
// builders: tab[x][entries][72] Montgomery limbs, entry e = base^e.
//   entries == 64: all powers 0..63 (k_modp_build_table64: the 6-bit windows of y^r)
//   entries == 16, odd_only: 0, 1, 3, 5, .. 15 (k_modp_build_table(odd): what a sliding-window schedule asks for)
// ---------------------------------------------------------------------------------------
extern "C" __global__ void __launch_bounds__(64 * PAIR_WAVES)
k_modp_build_table_pair(const uint8_t* __restrict__ base_be, int count, u32* __restrict__ tab, int entries, int odd_only,
                        const ModpConsts* __restrict__ cs, const Tables* __restrict__ gtab) {
  PAIR_KERNEL_PROLOGUE(gtab, count)
  const PairLane& pl = pc.pl;
  u32 acc[LP], b[LP];
  load_be256_pair(acc, base_be + (size_t)pc.x * 256, pl);
  u32* my = tab + (size_t)pc.x * entries * L;
  // it 0: base -> Montgomery form (times R^2); it 1 (odd_only): b <- b^2 after saving entry 1; then acc *= b, store
  int e = 1;
  for (int it = 0;; ++it) {
    const u32* fill = it == 0 ? cs->r2 : nullptr;
    pair_step<false>(acc, false, fill, b, pc.slot, pc.junk, pc.tb, pl);
    if (it == 0) {
#pragma unroll
      for (int k = 0; k < LP; ++k) b[k] = acc[k];
      if (pc.live) {
        u32 one[LP];
        load_pair_limbs(one, cs->one_m, pl);
        store_pair_limbs(my, one, pl);
        store_pair_limbs(my + L, acc, pl);
      }
      if (!odd_only) continue;
    } else if (odd_only && it == 1) {        // acc = base^2: becomes the multiplier; the chain restarts from base
      u32 t[LP];
#pragma unroll
      for (int k = 0; k < LP; ++k) { t[k] = acc[k]; acc[k] = b[k]; b[k] = t[k]; }
      continue;
    }
    if (it > 0) {
      e += odd_only ? 2 : 1;
      if (pc.live) store_pair_limbs(my + (size_t)e * L, acc, pl);
      if (e >= entries - 1) break;
    }
  }
}

// ---------------------------------------------------------------------------------------
// p_m[x] = g^e1[x] in Montgomery form through the wide fixed-base comb (comb16[k][d] = g^(d 65536^k), 128 rows): 127
// products, no squarings -- mode 1 of k_modp_comb_dual_exp.  The g^r half of a1 = g^r X^c (dleq.rs:75-77).
// ---------------------------------------------------------------------------------------
extern "C" __global__ void __launch_bounds__(64 * PAIR_WAVES) PAIR_OCC_ATTR
k_modp_comb16_exp_pair(const u32* __restrict__ comb16, const uint8_t* __restrict__ e1_be, int count, u32* __restrict__ p_m,
                       const ModpConsts* __restrict__ cs, const Tables* __restrict__ gtab) {
  PAIR_KERNEL_PROLOGUE(gtab, count)
  const PairLane& pl = pc.pl;
  auto digit16 = [&](int k) -> u32 {
    const uint8_t* e1 = e1_be + (size_t)pc.x * 256;
    return ((u32)e1[254 - 2 * k] << 8) | e1[255 - 2 * k];
  };
  auto row = [&](int k) { return comb16 + (size_t)k * 65536 * L; };      // row k of the comb: the same for every number
  u32 acc[LP];
  load_pair_limbs(acc, row(0) + (size_t)digit16(0) * L, pl);
  // every operation is a product with an entry of the next row: its DMA runs under the reduction before it
  u32 d = digit16(1);
  for (int k = 1; k < 128; ++k) {
    PairNext nx;
    if (k + 1 < 128) { nx.base = row(k + 1); nx.ent = digit16(k + 1); }
    pair_step_pf<false>(acc, false, row(k) + (size_t)d * L, false, nx, pc.wslots, pc.x0, count, pc.slot, pc.junk, pc.tb, pl);
    d = nx.ent;
  }
  if (pc.live) store_pair_limbs(p_m + (size_t)pc.x * L, acc, pl);
}

// ---------------------------------------------------------------------------------------
// The dealer's two fixed-base powers, X_i = g^P(i) and a1_i = g^w_i (participant.rs:207-215 with C_j = g^a_j, dleq.rs:207-211), through the
// same comb as canonical bytes: 127 products + the one that leaves the Montgomery domain; blockIdx.y picks the exponent set.  Beside
// the key-table kernel (no chain of squarings next to it) the pair layout's 122 instead of 191 issue slots per product count.
// ---------------------------------------------------------------------------------------
extern "C" __global__ void __launch_bounds__(64 * PAIR_WAVES) PAIR_OCC_ATTR
k_modp_comb16_twin_exp_pair(const u32* __restrict__ comb16, const uint8_t* __restrict__ e1_be, const uint8_t* __restrict__ e2_be, int count,
                            uint8_t* __restrict__ out1_be, uint8_t* __restrict__ out2_be, const ModpConsts* __restrict__ cs,
                            const Tables* __restrict__ gtab) {
  PAIR_KERNEL_PROLOGUE(gtab, count)
  const PairLane& pl = pc.pl;
  const uint8_t* e = (blockIdx.y ? e2_be : e1_be) + (size_t)pc.x * 256;
  uint8_t* out_be = blockIdx.y ? out2_be : out1_be;
  auto digit16 = [&](int k) -> u32 { return ((u32)e[254 - 2 * k] << 8) | e[255 - 2 * k]; };
  auto row = [&](int k) { return comb16 + (size_t)k * 65536 * L; };
  u32 acc[LP];
  load_pair_limbs(acc, row(0) + (size_t)digit16(0) * L, pl);
  for (int k = 1; k <= 128; ++k) {
    const u32* fill = k < 128 ? row(k) + (size_t)digit16(k) * L : cs->one;
    pair_step<false>(acc, false, fill, acc, pc.slot, pc.junk, pc.tb, pl);
  }
  store_canonical_pair(out_be + (size_t)pc.x * 256, acc, pc.slot, cs, pl, pc.live);
}

// ---------------------------------------------------------------------------------------
// out[x] = g^r[x] * B2[x]^c[x] with a 256-bit c of the share's own (verify_share: a1 = G^r pk^c, dleq.rs:66-77 via participant.rs:376-385):
// B2^c by fixed 4-bit windows of the low 256 bits of c against B2's 16-entry table (252 squarings, 64 products), then the 128 comb
// products of g^r on the same accumulator -- mode 0 of k_modp_comb_dual_exp on the pair layout (85 / 122 instead of 153 / 191 slots).
// ---------------------------------------------------------------------------------------
extern "C" __global__ void __launch_bounds__(64 * PAIR_WAVES) PAIR_OCC_ATTR
k_modp_comb16_dual_exp_pair(const u32* __restrict__ comb16, const u32* __restrict__ tab2, const uint8_t* __restrict__ r_be,
                            const uint8_t* __restrict__ c_be, size_t c_stride, int count, uint8_t* __restrict__ out_be,
                            const ModpConsts* __restrict__ cs, const Tables* __restrict__ gtab) {
  PAIR_KERNEL_PROLOGUE(gtab, count)
  const PairLane& pl = pc.pl;
  const uint8_t* r = r_be + (size_t)pc.x * 256;
  const uint8_t* c = c_be + (size_t)pc.x * c_stride;
  const u32* t2 = tab2 + (size_t)pc.x * 16 * L;
  auto nibble = [&](int w) -> u32 { const u32 b = c[255 - (w >> 1)]; return (w & 1) ? (b >> 4) : (b & 15); };
  auto digit16 = [&](int k) -> u32 { return ((u32)r[254 - 2 * k] << 8) | r[255 - 2 * k]; };
  u32 acc[LP];
  load_pair_limbs(acc, t2 + (size_t)nibble(63) * L, pl);        // top window: load instead of multiply
  //   s = 0 .. 3 the squarings in front of window w, 4 its product; then k = 0 .. 127 the comb rows, then the closing product
  int w = 62, s = 0, k = -1;
  while (true) {
    const u32* fill = nullptr;
    bool sq = false;
    if (w >= 0) {
      if (s < 4) sq = true; else fill = t2 + (size_t)nibble(w) * L;
    } else if (k < 128) {
      fill = comb16 + ((size_t)k * 65536 + digit16(k)) * L;
    } else {
      fill = cs->one;
    }
    pair_step<true>(acc, sq, fill, acc, pc.slot, pc.junk, pc.tb, pl);
    if (w >= 0) {
      if (s < 4) ++s; else { s = 0; --w; if (w < 0) k = 0; }
    } else if (k < 128) {
      ++k;
    } else {
      break;
    }
  }
  store_canonical_pair(out_be + (size_t)pc.x * 256, acc, pc.slot, cs, pl, pc.live);
}

// ---------------------------------------------------------------------------------------
// out[x] = p_m[x] * B2[x]^c for ONE shared exponent c given as a sliding-window schedule (mode 2 of
// k_modp_comb_dual_exp with c_sched): a1 = g^r * X^c (dleq.rs:75-77) once X is known.  tab2: odd-power tables of X.
// ---------------------------------------------------------------------------------------
extern "C" __global__ void __launch_bounds__(64 * PAIR_WAVES) PAIR_OCC_ATTR
k_modp_sched_exp_mul_pair(const u32* __restrict__ tab2, size_t tab2_stride, const uint16_t* __restrict__ c_sched,
                          const u32* __restrict__ p_m, int count, uint8_t* __restrict__ out_be,
                          const ModpConsts* __restrict__ cs, const Tables* __restrict__ gtab) {
  PAIR_KERNEL_PROLOGUE(gtab, count)
  const PairLane& pl = pc.pl;
  const int sn = (int)c_sched[0];
  u32 acc[LP];
  load_pair_limbs(acc, tab2 + (size_t)pc.x * tab2_stride + (size_t)c_sched[2] * L, pl);     // top window: load instead of multiply
  // operations from the top window's weight down: a squaring per bit, a product with X^digit where the schedule has a window,
  // at weight 0 the product with the stored g^r and the one with plain 1 that leaves the Montgomery domain.
  //   kind 0 squaring; 1 product with entry `ent` of tab2; 2 with p_m; 3 with cs->one (the last)
  int cur = (int)c_sched[1], si = 1, stage = 0, tail = 0;
  struct Op { int kind; u32 ent; };
  auto next_op = [&]() -> Op {
    while (true) {
      if (tail == 1) { tail = 2; return Op{3, 0u}; }
      if (stage == 0) {
        if (cur == 0) { tail = 1; return Op{2, 0u}; }
        --cur;
        stage = 1;
        return Op{0, 0u};
      }
      stage = 0;
      if (si < sn && cur == (int)c_sched[1 + 2 * si]) {
        const u32 e = c_sched[2 + 2 * si];
        ++si;
        return Op{1, e};
      }
    }
  };
  auto source = [&](const Op& op, PairNext& nx) {
    if (op.kind == 1) { nx.base = tab2; nx.stride = tab2_stride; nx.ent = op.ent; }
    else if (op.kind == 2) { nx.base = p_m; nx.stride = L; nx.ent = 0; }
    else if (op.kind == 3) { nx.base = cs->one; nx.stride = 0; nx.ent = 0; }
  };
  Op op = next_op();
  bool fetched = false;
  while (true) {
    PairNext me, nx;
    source(op, me);
    Op nop{-1, 0u};
    if (op.kind != 3) { nop = next_op(); source(nop, nx); }
    const u32* fill = me.base ? me.base + (size_t)pc.x * me.stride + (size_t)me.ent * L : nullptr;
    pair_step_pf<true>(acc, op.kind == 0, fill, fetched, nx, pc.wslots, pc.x0, count, pc.slot, pc.junk, pc.tb, pl);
    fetched = false;
    if (op.kind == 3) break;
    op = nop;
  }
  store_canonical_pair(out_be + (size_t)pc.x * 256, acc, pc.slot, cs, pl, pc.live);
}

// ---------------------------------------------------------------------------------------
// a2 = y^r * Y^c against a REGISTERED key's table (k_modp_keyset_dual_exp of modp_kernels.hip on the pair layout: the same table
// ks[key][j][d] = y^(d 2^(256 j)), d < 128, the same program -- 252 squarings, 296 products with table entries (7-bit windows of the
// eight 256-bit rows of r), 64 with Y^(nibble of c) -- at 85 / 122 instead of 153 / 191 issue slots).  dleq.rs:79-81 with y a
// long-lived participant key.
// ---------------------------------------------------------------------------------------
extern "C" __global__ void __launch_bounds__(64 * PAIR_WAVES) PAIR_OCC_ATTR
k_modp_keyset_dual_exp_pair(const u32* __restrict__ ks, size_t key_words, const u32* __restrict__ tab2, const uint8_t* __restrict__ r_be,
                            const uint8_t* __restrict__ c_be, int count, uint8_t* __restrict__ out_be,
                            const ModpConsts* __restrict__ cs, const Tables* __restrict__ gtab) {
  PAIR_KERNEL_PROLOGUE(gtab, count)
  const PairLane& pl = pc.pl;
  // the program of k_modp_keyset_dual_exp: from bit 252 down, a squaring per bit (not at the first), at bits divisible by 7 the eight
  // products with ks[j][window of r_j], at bits divisible by 4 the product with Y^(nibble of c), at the end the product with plain 1
  //   s = 0 squaring, 1 .. 8 key table j = s - 1, 9 nibble of c, 10 closing
  auto digit = [&](int j, int w) -> u32 {
    const uint8_t* r = r_be + (size_t)pc.x * 256;
    const int g = 256 * j + 7 * w, b = g >> 3;
    const u32 lo = r[255 - b];
    const u32 hi = (b + 1 < 256) ? r[254 - b] : 0u;
    const int top = 256 - 7 * w;
    return ((lo | (hi << 8)) >> (g & 7)) & (u32)((1 << (top < 7 ? top : 7)) - 1);
  };
  u32 acc[LP];
  load_pair_limbs(acc, cs->one_m, pl);
  int cur = 7 * 36, s = 1;
  while (true) {
    const u32* fill = nullptr;
    bool skip = false;
    if (s >= 1 && s <= 8) {
      if (cur % 7 == 0) fill = ks + (size_t)pc.x * key_words + ((size_t)(s - 1) * 128 + digit(s - 1, cur / 7)) * L; else skip = true;
    } else if (s == 9) {
      if ((cur & 3) == 0) {
        const u32 byte = c_be[255 - (cur >> 3)];
        fill = tab2 + ((size_t)pc.x * 16 + ((cur & 4) ? (byte >> 4) : (byte & 15))) * L;
      } else {
        skip = true;
      }
    } else if (s == 10) {
      fill = cs->one;                                    // leave the Montgomery domain
    }
    if (!skip) pair_step<true>(acc, s == 0, fill, acc, pc.slot, pc.junk, pc.tb, pl);
    if (s == 10) break;
    if (s == 9) {
      if (cur == 0) { s = 10; continue; }
      --cur;
      s = 0;
    } else {
      ++s;
    }
  }
  store_canonical_pair(out_be + (size_t)pc.x * 256, acc, pc.slot, cs, pl, pc.live);
}

// ---------------------------------------------------------------------------------------
// The dealer against REGISTERED keys: Y_i = y_i^P(i) and a2_i = y_i^w_i (participant.rs:219, dleq.rs:213-216) from the per-key tables of
// k_modp_keyset_dual_exp_pair -- 252 squarings and 296 table products per exponent instead of the bucket kernels' shared 2 045 squarings
// and 820 products for both: 115 K instead of 274 K issue slots per share.  blockIdx.y picks the exponent set; full-width exponents
// (eight 256-bit rows, P(i) and w_i are residues mod q - 1).
// ---------------------------------------------------------------------------------------
extern "C" __global__ void __launch_bounds__(64 * PAIR_WAVES) PAIR_OCC_ATTR
k_modp_keyset_twin_exp_pair(const u32* __restrict__ ks, size_t key_words, const uint8_t* __restrict__ e1_be, const uint8_t* __restrict__ e2_be,
                            int count, uint8_t* __restrict__ out1_be, uint8_t* __restrict__ out2_be,
                            const ModpConsts* __restrict__ cs, const Tables* __restrict__ gtab) {
  PAIR_KERNEL_PROLOGUE(gtab, count)
  const PairLane& pl = pc.pl;
  const uint8_t* e_be = blockIdx.y ? e2_be : e1_be;
  uint8_t* out_be = blockIdx.y ? out2_be : out1_be;
  const uint8_t* e = e_be + (size_t)pc.x * 256;
  auto digit = [&](int j, int w) -> u32 {
    const int g = 256 * j + 7 * w, b = g >> 3;
    const u32 lo = e[255 - b];
    const u32 hi = (b + 1 < 256) ? e[254 - b] : 0u;
    const int top = 256 - 7 * w;
    return ((lo | (hi << 8)) >> (g & 7)) & (u32)((1 << (top < 7 ? top : 7)) - 1);
  };
  const u32* kt = ks + (size_t)pc.x * key_words;
  u32 acc[LP];
  load_pair_limbs(acc, cs->one_m, pl);
  //   s = 0 squaring (bit cur), 1 .. 8 key table j = s - 1 (cur divisible by 7), 9 closing: leave the Montgomery domain
  int cur = 7 * 36, s = 1;
  while (true) {
    const u32* fill = nullptr;
    if (s >= 1 && s <= 8) fill = kt + ((size_t)(s - 1) * 128 + digit(s - 1, cur / 7)) * L;
    else if (s == 9) fill = cs->one;
    pair_step<true>(acc, s == 0, fill, acc, pc.slot, pc.junk, pc.tb, pl);
    if (s == 9) break;
    if (s == 0) {
      if (cur % 7 == 0) s = 1; else --cur;
    } else if (s == 8) {
      if (cur == 0) s = 9; else { --cur; s = 0; }
    } else {
      ++s;
    }
  }
  store_canonical_pair(out_be + (size_t)pc.x * 256, acc, pc.slot, cs, pl, pc.live);
}

// ---------------------------------------------------------------------------------------
// One base, two exponents (the dealer: Y_i = y_i^P(i) and a2_i = y_i^w_i, participant.rs:219, dleq.rs:213-216; the
// participant: S_i = Y_i^(1/x_i) and a2_i = S_i^w_i, participant.rs:310-314): the right-to-left bucket phase of
// k_modp_twin_exp_buckets (modp_kernels.hip -- same windows, same buckets and occupancy masks in HBM, the same combine kernel
// afterwards) on the pair layout: 2 045 squarings and 820 bucket products per share at 85 / 122 instead of 153 / 191 issue
// slots.  A number whose digit is 0, or whose bucket is still empty, multiplies by a harmless operand and drops the result
// (a bucket's first factor is stored, not multiplied), so that the wave stays in step.
// ---------------------------------------------------------------------------------------
extern "C" __global__ void __launch_bounds__(64 * PAIR_WAVES) PAIR_OCC_ATTR
k_modp_twin_exp_buckets_pair(const uint8_t* __restrict__ base_be, const uint8_t* __restrict__ e1_be, const uint8_t* __restrict__ e2_be,
                             int count, u32* __restrict__ buckets, u32* __restrict__ occupancy, u32* __restrict__ curbuf,
                             const ModpConsts* __restrict__ cs, const Tables* __restrict__ gtab) {
  constexpr int BW = MODP_BUCKET_W, BENT = (1 << BW) - 1, BWIN = (2048 + BW - 1) / BW;
  PAIR_KERNEL_PROLOGUE(gtab, count)
  const PairLane& pl = pc.pl;
  // The running power cur = y^(2^(5k)) is an operand of both bucket products of a window and of the next squaring: kept in
  // registers across a product it would be live beside T and the result (36 + 72 + 36 + the rows' temporaries: 284
  // registers, one wave per SIMD).  It lives in HBM instead (288 bytes per number, L2-resident: one store and three loads of
  // a window against its seven Montgomery operations), every operation starts from a fresh copy, and the kernel fits the 256
  // registers of two waves per SIMD without spilling.
  // (indexed by the lane's own number, also for the padding lanes past `count`: those repeat the last share's work in step with
  // nothing, and a slot of their own keeps them from writing into the live one's)
  u32* mycur = curbuf + (size_t)((blockIdx.x * PAIR_WAVES + (threadIdx.x >> 6)) * 32 + (int)(pl.lane & 31)) * L;
  u32 acc[LP];
  load_be256_pair(acc, base_be + (size_t)pc.x * 256, pl);
  u32* mine = buckets + (size_t)pc.x * 2 * BENT * L;
  const uint8_t* ex[2] = {e1_be + (size_t)pc.x * 256, e2_be + (size_t)pc.x * 256};
  auto digit = [&](const uint8_t* e, int k) -> u32 {
    const int o = BW * k, b = o >> 3;
    const u32 lo = e[255 - b];
    const u32 hi = (b + 1 < 256) ? e[254 - b] : 0u;
    return ((lo | (hi << 8)) >> (o & 7)) & (u32)BENT;
  };
  u32 occ[2] = {0, 0};
  // op 0: the base into Montgomery form; per window k: ops 1, 2 = the bucket products of the two exponents, ops 3 .. 2+BW the
  // squarings.  ONE product site, one squaring site and one reduction in the loop (instruction cache).
  // TWIN_PREFETCH: the bucket (or the harmless operand) of the NEXT bucket product comes in by LDS-DMA while the operation
  // before it reduces -- the last squaring of a window fetches for op 1 of the next, op 1 for op 2 --, as the VALU-only kernel
  // does (modp_kernels.hip::bucket_prefetch).  A bucket is only ever written by its own number's earlier operations, which
  // the s_waitcnt vmcnt(0) in front of the prefetch has seen complete.
  int k = 0, op = 0;
  bool fetched = false;
  while (true) {
    const bool sq = op >= 3;
    const u32* fill = cs->r2;
    u32* bk = nullptr;
    bool has = false;
    if (op == 1 || op == 2) {
      const u32 d = digit(ex[op - 1], k);
      has = (occ[op - 1] >> d) & 1u;
      if (d != 0) {
        bk = mine + ((size_t)(op - 1) * BENT + (d - 1)) * L;
        occ[op - 1] |= 1u << d;
      }
      fill = (bk != nullptr && has) ? bk : cs->one_m;
      load_pair_limbs(acc, mycur, pl);
      if (bk != nullptr && !has && pc.live) store_pair_limbs(bk, acc, pl);        // a bucket's first factor is stored, not multiplied
    }
    u64 T[LP];
    if (sq) {
      slot_store_pair(pc.slot, acc, pl);
      __builtin_amdgcn_wave_barrier();
      phase_a<true>(T, acc, pc.slot, pc.junk, pl);
    } else {
      if (fetched) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else slot_fill_pair(pc.slot, fill, pl);
      __builtin_amdgcn_wave_barrier();
      phase_a<false>(T, acc, pc.slot, pc.junk, pl);
    }
    u32 r[LP];
    reduce(r, T, pc.slot, pc.tb, pl);
    __builtin_amdgcn_wave_barrier();
    if (sq || op == 0) {
#pragma unroll
      for (int i = 0; i < LP; ++i) acc[i] = r[i];
      if (op == 0 || op == 2 + BW) store_pair_limbs(mycur, acc, pl);               // the window's cur
    } else if (bk != nullptr && has && pc.live) {
      store_pair_limbs(bk, r, pl);
    }
    if (op == 2 && k == BWIN - 1) break;
    if (op == 2) load_pair_limbs(acc, mycur, pl);                                  // back to the chain of squarings
    if (op == 2 + BW) { op = 1; ++k; } else ++op;
  }
  if (pc.live && pl.h == 0) {
    occupancy[(size_t)pc.x * 2] = occ[0];
    occupancy[(size_t)pc.x * 2 + 1] = occ[1];
  }
}

// ---------------------------------------------------------------------------------------
// Forward-difference stepping (k_modp_fd_step of modp_kernels.hip: D_k <- D_k * D_(k+1), output D_0) on the pair layout: a
// pipeline stage is one wave = 32 consecutive levels of a chain, level k multiplies by level k + 1 -- the NEXT number of the
// wave, whose LDS slot phase A reads directly (bn_pair.h) -- and the top level of a stage by the number the stage above hands
// down through HBM (same self-validating words, tags and time-out as the quad kernel).  Same state arrays and outputs; the
// stages of a chain are counted in 32s here (tpad = t rounded up to a power of two >= 32).  One wave per workgroup, as the
// quad pipelines: a waiting stage only ever waits for workgroups dispatched before it.
// ---------------------------------------------------------------------------------------
namespace {
constexpr u32 FD_VALID = 0x80000000u, FD_POISON = 0x40000000u;
constexpr long long FD_TIMEOUT = 200000000LL;          // 2 s of the 100 MHz wall clock
struct FdShared {
  Tables tb;
  __attribute__((aligned(16))) u32 slots[32 * SLOTW];
  __attribute__((aligned(16))) u32 inslot[SLOTW];
  __attribute__((aligned(16))) u32 oneslot[SLOTW];
  u32 junk[L];
};
__device__ __forceinline__ void fd_publish(u32* __restrict__ dst, const u32 (&a)[LP], const PairLane& pl, u32 tag) {
#pragma unroll
  for (int i = 0; i < LP; ++i) __hip_atomic_store(dst + LP * pl.h + i, a[i] | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// number 31 of the wave (lanes 31 and 63) waits for one number from the stage above and puts it into `dst`; false
// (wave-uniform) when the input was poisoned or the wait timed out
__device__ __forceinline__ bool fd_receive(const u32* __restrict__ src, u32* dst, bool reader, const PairLane& pl) {
  bool ok = true;
  if (reader) {
    u32 v[LP];
    const long long t0 = wall_clock64();
    while (true) {
      u32 all = 0xffffffffu, any = 0;
#pragma unroll
      for (int i = 0; i < LP; ++i) {
        v[i] = __hip_atomic_load(src + LP * pl.h + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        all &= v[i];
        any |= v[i];
      }
      const u32 st = (any & FD_POISON) ? FD_POISON : (all & FD_VALID);
      const uint64_t both = (1ull << 31) | (1ull << 63);
      const uint64_t good = __builtin_amdgcn_ballot_w64(st == FD_VALID) & both;
      const uint64_t bad = __builtin_amdgcn_ballot_w64(st == FD_POISON) & both;
      if (good == both) break;
      if (bad != 0 || wall_clock64() - t0 > FD_TIMEOUT) { ok = false; break; }
      __builtin_amdgcn_s_sleep(4);
    }
#pragma unroll
    for (int i = 0; i < LP; ++i) dst[LP * pl.h + i] = v[i] & 0x3fffffffu;
  }
  return __builtin_amdgcn_ballot_w64(!ok) == 0;
}
}  // namespace

extern "C" __global__ void __launch_bounds__(64) PAIR_OCC_ATTR
k_modp_fd_step_pair(const u32* __restrict__ state, const u32* __restrict__ state_back, int chains, int t, int tpad, int w0,
                    int chain_len, int count, u32* __restrict__ x_m, u32* __restrict__ hand, int* __restrict__ gate,
                    int inject_fault, const ModpConsts* __restrict__ cs, const Tables* __restrict__ gtab, size_t box_state,
                    size_t box_xm, size_t box_hand) {
  __shared__ FdShared sh;
  if (*gate != 1) return;
  state += blockIdx.y * box_state;
  state_back += blockIdx.y * box_state;
  x_m += blockIdx.y * box_xm;
  hand += blockIdx.y * box_hand;
  const int stages = tpad / 32;
  const int sidx = blockIdx.x / (2 * chains);                       // 0 = the top levels
  const int dir = (blockIdx.x / chains) & 1;
  const int chain = blockIdx.x % chains;
  const int kbase = tpad - 32 * (sidx + 1);
  if (kbase >= t || (dir == 1 && w0 == 0)) return;
  tables_to_lds(&sh.tb, gtab);
  __builtin_amdgcn_s_setprio(3);       // latency-critical and few: issue ahead of the wide kernels sharing the SIMD
  const PairLane pl = make_pair_lane();
  const int j = (int)(pl.lane & 31);
  const int steps = dir == 0 ? chain_len - 1 - w0 : w0 + t - 1;
  const int hand_len = chain_len + t;
  const int k = kbase + j;
  const bool has_up = kbase + 32 < t;
  const bool has_down = kbase > 0;
  u32* slot = sh.slots + j * SLOTW;
  const bool reader = j == 31;
  const u32* bsrc = (k + 1 < t) ? (reader ? sh.inslot : slot + SLOTW) : sh.oneslot;
  const size_t lane_area = ((size_t)dir * chains + chain) * stages;
  u32* mine = hand + (lane_area + sidx) * (size_t)hand_len * L;
  const u32* up = hand + (lane_area + sidx - 1) * (size_t)hand_len * L;
  const u32* st = (dir == 0 ? state : state_back) + (size_t)chain * t * L;
  u32 D[LP];
  if (k < t) load_pair_limbs(D, st + (size_t)k * L, pl); else load_pair_limbs(D, cs->one_m, pl);
  if (j == 0) slot_fill_pair(sh.oneslot, cs->one_m, pl);
  if (has_up && reader) slot_fill_pair(sh.inslot, st + (size_t)(kbase + 32) * L, pl);        // step 0 of the stage above
  const bool writer = kbase == 0 && j == 0;
  for (int step = 1; step <= steps; ++step) {
    slot_store_pair(slot, D, pl);
    const bool faulty = (inject_fault == 1 && !has_up && dir == 0 && chain == 0 && step == 5) ||
                        (inject_fault == 2 && dir == 1 && chain == chains - 1 && sidx == (stages > 1 ? 1 : 0) && step == 40);
    if (faulty) {
      if (threadIdx.x == 0) *gate = 0;
      if (has_down && j == 0) fd_publish(mine + (size_t)step * L, D, pl, FD_POISON);
      return;
    }
    if (has_up && step > 1) {
      if (!fd_receive(up + (size_t)(step - 1) * L, sh.inslot, reader, pl)) {
        if (threadIdx.x == 0) *gate = 0;
        if (has_down && j == 0) fd_publish(mine + (size_t)step * L, D, pl, FD_POISON);
        return;
      }
    }
    __builtin_amdgcn_wave_barrier();
    u64 T[LP];
    phase_a<false>(T, D, slot, sh.junk, pl, bsrc);
    u32 r[LP];
    reduce(r, T, slot, &sh.tb, pl);
    __builtin_amdgcn_s_setprio(3);       // (reduce() leaves the wave at priority 0)
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < LP; ++i) D[i] = r[i];
    if (has_down && j == 0) fd_publish(mine + (size_t)step * L, D, pl, FD_VALID);
    const int jj = dir == 0 ? w0 + step : w0 + t - 1 - step;
    const size_t idx = (size_t)chain + (size_t)chains * jj;
    if (writer && step >= t && idx < (size_t)count) store_pair_limbs(x_m + idx * L, D, pl);
  }
}

// ---------------------------------------------------------------------------------------
// The same stepping as a sequence of WIDE launches in which no wave ever waits for another (round 5).  The cell (level k,
// step s) needs (k, s-1) and (k+1, s-1) only, so the (stage, block of `tile_steps` steps) grid can be walked by anti-diagonals:
// tile (stage sidx, block b) runs in launch number sidx + b, after tile (sidx - 1, b) -- the stage above, which handed down the
// values of its steps [b M, (b + 1) M) -- and tile (sidx, b - 1), its own state.  The kernel boundary orders everything: plain
// stores, no validity tags, no spinning, no time-out, no residency requirement; a wave occupies its slot exactly for the
// tile_steps products it computes, where a persistent stage occupies it for the whole chain and spins whenever the stage above is
// late (which, with ten boxes in flight, is most of the time: 512 waves of 252 registers per box for 93-240 ms).
// The handed value of the NEXT step comes in by LDS-DMA under the current product (two in-slots), so the wave never sees the
// load's latency either.  State lives in the `state` arrays between tiles (level k of a chain at its step b M); entry 0 of a
// stage's hand-over area holds its bottom level's INITIAL value (the persistent kernel reads that from `state`, which is
// overwritten here).  Same hand-over layout and outputs as k_modp_fd_step_pair.
// ---------------------------------------------------------------------------------------
namespace {
struct FdTileShared {
  Tables tb;
  __attribute__((aligned(16))) u32 slots[32 * SLOTW];
  __attribute__((aligned(16))) u32 inslot[2][SLOTW];
  __attribute__((aligned(16))) u32 oneslot[SLOTW];
  u32 junk[L];
};
// 72 words at `src` -> `dst` (LDS) without registers: lanes 0 .. 17 move 16 bytes each
__device__ __forceinline__ void fd_tile_fetch(u32* dst, const u32* __restrict__ src, const PairLane& pl) {
  if (pl.lane < 18)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(uintptr_t)(src + 4 * pl.lane),
                                     (__attribute__((address_space(3))) void*)(uintptr_t)dst, 16, 0, 0);
}
}  // namespace

extern "C" __global__ void __launch_bounds__(64) PAIR_OCC_ATTR
k_modp_fd_step_pair_tile(u32* __restrict__ state, u32* __restrict__ state_back, int chains, int t, int tpad, int w0, int chain_len,
                         int count, u32* __restrict__ x_m, u32* __restrict__ hand, const int* __restrict__ gate, int diag,
                         int tile_steps, const ModpConsts* __restrict__ cs, const Tables* __restrict__ gtab, size_t box_state,
                         size_t box_xm, size_t box_hand) {
  __shared__ FdTileShared sh;
  if (*gate != 1) return;
  const int stages = tpad / 32;
  const int sidx = blockIdx.x / (2 * chains);                       // 0 = the top levels
  const int dir = (blockIdx.x / chains) & 1;
  const int chain = blockIdx.x % chains;
  const int kbase = tpad - 32 * (sidx + 1);
  if (kbase >= t || (dir == 1 && w0 == 0)) return;
  const int s_first = (tpad - t) / 32;                              // stages above this one hold no level
  const int b = diag - (sidx - s_first);
  const int steps = dir == 0 ? chain_len - 1 - w0 : w0 + t - 1;
  if (b < 0 || b * tile_steps + 1 > steps) return;
  const int step_lo = b * tile_steps + 1;
  const int step_hi = step_lo + tile_steps - 1 < steps ? step_lo + tile_steps - 1 : steps;
  state += blockIdx.y * box_state;
  state_back += blockIdx.y * box_state;
  x_m += blockIdx.y * box_xm;
  hand += blockIdx.y * box_hand;
  tables_to_lds(&sh.tb, gtab);
  const PairLane pl = make_pair_lane();
  const int j = (int)(pl.lane & 31);
  const int hand_len = chain_len + t;
  const int k = kbase + j;
  const bool has_up = kbase + 32 < t;
  const bool has_down = kbase > 0;
  u32* slot = sh.slots + j * SLOTW;
  const bool reader = j == 31;
  const size_t lane_area = ((size_t)dir * chains + chain) * stages;
  u32* mine = hand + (lane_area + sidx) * (size_t)hand_len * L;
  const u32* up = hand + (lane_area + sidx - 1) * (size_t)hand_len * L;
  u32* st = (dir == 0 ? state : state_back) + (size_t)chain * t * L;
  u32 D[LP];
  if (k < t) load_pair_limbs(D, st + (size_t)k * L, pl); else load_pair_limbs(D, cs->one_m, pl);
  if (j == 0) slot_fill_pair(sh.oneslot, cs->one_m, pl);
  if (b == 0 && has_down && j == 0) fd_publish(mine, D, pl, 0);                        // entry 0: the initial value
  if (has_up) fd_tile_fetch(sh.inslot[step_lo & 1], up + (size_t)(step_lo - 1) * L, pl);
  const bool writer = kbase == 0 && j == 0;
  for (int step = step_lo; step <= step_hi; ++step) {
    slot_store_pair(slot, D, pl);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // this step's handed value has landed
    __builtin_amdgcn_wave_barrier();
    if (has_up && step < step_hi) fd_tile_fetch(sh.inslot[(step + 1) & 1], up + (size_t)step * L, pl);     // the next one under this product
    const u32* bsrc = (k + 1 < t) ? (reader ? sh.inslot[step & 1] : slot + SLOTW) : sh.oneslot;
    u64 T[LP];
    phase_a<false>(T, D, slot, sh.junk, pl, bsrc);
    u32 r[LP];
    reduce(r, T, slot, &sh.tb, pl);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < LP; ++i) D[i] = r[i];
    if (has_down && j == 0) fd_publish(mine + (size_t)step * L, D, pl, 0);
    const int jj = dir == 0 ? w0 + step : w0 + t - 1 - step;
    const size_t idx = (size_t)chain + (size_t)chains * jj;
    if (writer && step >= t && idx < (size_t)count) store_pair_limbs(x_m + idx * L, D, pl);
  }
  if (k < t) store_pair_limbs(st + (size_t)k * L, D, pl);            // this level at step step_hi: where the next tile starts
}

// ---------------------------------------------------------------------------------------
extern "C" int modp_pair_tables_upload(void** dev_tables) {
  static_assert(sizeof(MM_GT1) == sizeof(Tables::gt1) && sizeof(MM_GT2) == sizeof(Tables::gt2) && sizeof(MM_C1) == sizeof(Tables::c1) &&
                    sizeof(MM_C2) == sizeof(Tables::c2), "generated tables do not match bn_pair.h");
  Tables* h = new Tables;
  memcpy(h->gt1, MM_GT1, sizeof(MM_GT1));
  memcpy(h->gt2, MM_GT2, sizeof(MM_GT2));
  memcpy(h->c1, MM_C1, sizeof(MM_C1));       // [R][h][16] ints = v16i [2 R + h]
  memcpy(h->c2, MM_C2, sizeof(MM_C2));
  void* d = nullptr;
  hipError_t e = hipMalloc(&d, sizeof(Tables));
  if (e == hipSuccess) e = hipMemcpy(d, h, sizeof(Tables), hipMemcpyHostToDevice);
  delete h;
  if (e != hipSuccess) return (int)e;
  *dev_tables = d;
  return 0;
}

static inline int pair_grid(int count) { return (count + 32 * PAIR_WAVES - 1) / (32 * PAIR_WAVES); }

extern "C" int modp_launch_dual_exp_w6_pair(const uint32_t* tab1, const uint32_t* tab2, const uint8_t* e1, const uint8_t* c,
                                            size_t c_stride, const uint16_t* c_sched, int count, uint8_t* out, const void* cs,
                                            const void* pair_tables, hipStream_t s) {
  if (count <= 0) return 0;
  hipLaunchKernelGGL(k_modp_dual_exp_w6_pair, dim3(pair_grid(count)), dim3(64 * PAIR_WAVES), 0, s, tab1, tab2, e1, c, c_stride, count,
                     out, (const ModpConsts*)cs, c_sched, (const Tables*)pair_tables);
  return (int)hipGetLastError();
}

extern "C" int modp_launch_build_table_pair(const uint8_t* base_be, int count, uint32_t* tab, int entries, int odd_only, const void* cs,
                                            const void* pair_tables, hipStream_t s) {
  if (count <= 0) return 0;
  hipLaunchKernelGGL(k_modp_build_table_pair, dim3(pair_grid(count)), dim3(64 * PAIR_WAVES), 0, s, base_be, count, tab, entries, odd_only,
                     (const ModpConsts*)cs, (const Tables*)pair_tables);
  return (int)hipGetLastError();
}

extern "C" int modp_launch_comb16_exp_pair(const uint32_t* comb16, const uint8_t* e1, int count, uint32_t* p_m, const void* cs,
                                           const void* pair_tables, hipStream_t s) {
  if (count <= 0) return 0;
  hipLaunchKernelGGL(k_modp_comb16_exp_pair, dim3(pair_grid(count)), dim3(64 * PAIR_WAVES), 0, s, comb16, e1, count, p_m,
                     (const ModpConsts*)cs, (const Tables*)pair_tables);
  return (int)hipGetLastError();
}

extern "C" int modp_launch_comb16_twin_exp_pair(const uint32_t* comb16, const uint8_t* e1, const uint8_t* e2, int count, uint8_t* out1,
                                                uint8_t* out2, const void* cs, const void* pair_tables, hipStream_t s) {
  if (count <= 0) return 0;
  // e2 == nullptr: one exponent set (the participant's a1 = G^w)
  hipLaunchKernelGGL(k_modp_comb16_twin_exp_pair, dim3(pair_grid(count), e2 ? 2 : 1), dim3(64 * PAIR_WAVES), 0, s, comb16, e1, e2, count, out1, out2,
                     (const ModpConsts*)cs, (const Tables*)pair_tables);
  return (int)hipGetLastError();
}

extern "C" int modp_launch_comb16_dual_exp_pair(const uint32_t* comb16, const uint32_t* tab2, const uint8_t* r, const uint8_t* c, size_t c_stride,
                                                int count, uint8_t* out, const void* cs, const void* pair_tables, hipStream_t s) {
  if (count <= 0) return 0;
  hipLaunchKernelGGL(k_modp_comb16_dual_exp_pair, dim3(pair_grid(count)), dim3(64 * PAIR_WAVES), 0, s, comb16, tab2, r, c, c_stride, count, out,
                     (const ModpConsts*)cs, (const Tables*)pair_tables);
  return (int)hipGetLastError();
}

extern "C" int modp_launch_sched_exp_mul_pair(const uint32_t* tab2, size_t tab2_stride, const uint16_t* c_sched, const uint32_t* p_m,
                                              int count, uint8_t* out, const void* cs, const void* pair_tables, hipStream_t s) {
  if (count <= 0) return 0;
  hipLaunchKernelGGL(k_modp_sched_exp_mul_pair, dim3(pair_grid(count)), dim3(64 * PAIR_WAVES), 0, s, tab2, tab2_stride, c_sched, p_m,
                     count, out, (const ModpConsts*)cs, (const Tables*)pair_tables);
  return (int)hipGetLastError();
}

extern "C" int modp_launch_twin_exp_pair(const uint8_t* base_be, const uint8_t* e1, const uint8_t* e2, int count, uint32_t* buckets,
                                         uint32_t* occupancy, uint8_t* out1, uint8_t* out2, const void* cs, const void* pair_tables,
                                         hipStream_t s) {
  if (count <= 0) return 0;
  // the running powers (72 words per share) sit behind the occupancy masks: callers size the buffer with
  // modp_twin_exp_bucket_words() + MODP_TWIN_EXTRA_WORDS words per share
  // (one per lane of the grid: up to 32 PAIR_WAVES - 1 more than `count`, MODP_TWIN_SLACK_BYTES at the end of the buffer)
  uint32_t* curbuf = occupancy + (((size_t)2 * count + 3) & ~(size_t)3);
  hipLaunchKernelGGL(k_modp_twin_exp_buckets_pair, dim3(pair_grid(count)), dim3(64 * PAIR_WAVES), 0, s, base_be, e1, e2, count, buckets,
                     occupancy, curbuf, (const ModpConsts*)cs, (const Tables*)pair_tables);
  if (hipGetLastError() != hipSuccess) return 1;
  return modp_launch_bucket_combine(buckets, occupancy, count, out1, out2, cs, s);
}

extern "C" int modp_fd_tpad_pair(int t) {
  int p = 32;
  while (p < t) p <<= 1;
  return p;
}
extern "C" int modp_launch_fd_step_pair_boxes(const uint32_t* state, const uint32_t* state_back, size_t box_state, int chains, int t,
                                              int w0, int chain_len, int count, uint32_t* x_m, size_t box_xm, uint32_t* hand,
                                              size_t box_hand, int boxes, int* gate, int inject_fault, const void* cs,
                                              const void* pair_tables, hipStream_t s) {
  const int tpad = modp_fd_tpad_pair(t);
  hipLaunchKernelGGL(k_modp_fd_step_pair, dim3(2 * chains * (tpad / 32), boxes), dim3(64), 0, s, state, state_back, chains, t, tpad, w0,
                     chain_len, count, x_m, hand, gate, inject_fault, (const ModpConsts*)cs, (const Tables*)pair_tables, box_state,
                     box_xm, box_hand);
  return (int)hipGetLastError();
}

// the tiled form of the same stepping: one launch per anti-diagonal of the (stage, block of tile_steps steps) grid
extern "C" int modp_launch_fd_step_pair_tiled_boxes(uint32_t* state, uint32_t* state_back, size_t box_state, int chains, int t, int w0,
                                                    int chain_len, int count, uint32_t* x_m, size_t box_xm, uint32_t* hand,
                                                    size_t box_hand, int boxes, const int* gate, int tile_steps, const void* cs,
                                                    const void* pair_tables, hipStream_t s) {
  const int tpad = modp_fd_tpad_pair(t);
  const int stages_used = tpad / 32 - (tpad - t) / 32;
  const int steps_f = chain_len - 1 - w0, steps_b = w0 > 0 ? w0 + t - 1 : 0;
  const int steps = steps_f > steps_b ? steps_f : steps_b;
  if (steps <= 0) return 0;
  if (tile_steps < 1) tile_steps = 1;
  const int nblk = (steps + tile_steps - 1) / tile_steps;
  for (int diag = 0; diag < stages_used + nblk - 1; ++diag)
    hipLaunchKernelGGL(k_modp_fd_step_pair_tile, dim3(2 * chains * (tpad / 32), boxes), dim3(64), 0, s, state, state_back, chains, t, tpad,
                       w0, chain_len, count, x_m, hand, gate, diag, tile_steps, (const ModpConsts*)cs, (const Tables*)pair_tables, box_state,
                       box_xm, box_hand);
  return (int)hipGetLastError();
}

extern "C" int modp_launch_keyset_twin_exp_pair(const uint32_t* ks, const uint8_t* e1, const uint8_t* e2, int count, uint8_t* out1,
                                                uint8_t* out2, const void* cs, const void* pair_tables, hipStream_t s) {
  if (count <= 0) return 0;
  hipLaunchKernelGGL(k_modp_keyset_twin_exp_pair, dim3(pair_grid(count), 2), dim3(64 * PAIR_WAVES), 0, s, ks, modp_keyset_words_per_key(), e1,
                     e2, count, out1, out2, (const ModpConsts*)cs, (const Tables*)pair_tables);
  return (int)hipGetLastError();
}

extern "C" int modp_launch_keyset_dual_exp_pair(const uint32_t* ks, const uint32_t* tab2, const uint8_t* r, const uint8_t* c, int count,
                                                uint8_t* out, const void* cs, const void* pair_tables, hipStream_t s) {
  if (count <= 0) return 0;
  hipLaunchKernelGGL(k_modp_keyset_dual_exp_pair, dim3(pair_grid(count)), dim3(64 * PAIR_WAVES), 0, s, ks, modp_keyset_words_per_key(), tab2, r,
                     c, count, out, (const ModpConsts*)cs, (const Tables*)pair_tables);
  return (int)hipGetLastError();
}
