// MODP-2048 batch group-exponentiation kernels for gfx950 (MI355X).
//
// These are the device side of the hot path of the reference's
//   Participant<ModpGroup>::verify_distribution_shares  (src/participant.rs:399-455)
//   DLEQ verifier commitments                            (src/dleq.rs:66-84)
//   ModpGroup::exp / ModpGroup::mul                      (src/groups/modp.rs:122-132)
// re-designed for CDNA4: one number per DPP quad, radix-2^29 carry-free column
// accumulation (see bn_quad.h), one wavefront (16 numbers) per workgroup so that waves
// never synchronise with each other, window tables in an HBM workspace, second operand of
// every product staged in LDS.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include "bn_quad.h"
#include "modp_kernels.h"

// Occupancy target.  Measured on MI355X (n=65536, t=256): 3 waves/SIMD without spills (135 VGPRs) beats
// 4 waves/SIMD with a few spilled values (128 VGPRs); v_mad_u64_u32 already issues at its full rate from
// 2 waves/SIMD (profiles/r01_ubench_sustained_mad_clock.txt).
#ifndef MODP_WAVES_PER_EU
#define MODP_WAVES_PER_EU 3
#endif
#ifndef MODP_WAVES_PER_EU_MAX
#define MODP_WAVES_PER_EU_MAX MODP_WAVES_PER_EU
#endif
#ifndef FD_STEP_PREFETCH
#define FD_STEP_PREFETCH 1      // stepping stages request the next step's handed number under this step's product
#endif
#ifndef MODP_SETPRIO
#define MODP_SETPRIO 3          // wave priority of the latency-bound launches (seeds, inversion tree, pipeline stages)
#endif
#define WAVES_ATTR __attribute__((amdgpu_waves_per_eu(MODP_WAVES_PER_EU, MODP_WAVES_PER_EU_MAX)))
// Waves per workgroup.  Waves never talk to each other, so a workgroup is ONE wave: a single-wave workgroup fits any
// free wave slot, whereas a 4-wave workgroup needs a free slot on all four SIMDs of a CU at once -- and the
// single-wave stages of the forward-difference pipelines, which sit on their SIMDs for tens of milliseconds, leave
// the slots of a CU unevenly filled (measured: +6 % for the whole verification with 1 instead of 4).
#ifndef MODP_WPB
#define MODP_WPB 1
#endif
#define BLOCK_THREADS (64 * MODP_WPB)
#define NUMS_PER_BLOCK (NUMS_PER_WAVE * MODP_WPB)

using namespace bn;

namespace {

// device-resident constants: [0]=N, [1]=R^2 mod N, [2]=R mod N (Montgomery one), [3]=plain 1
struct ModpConsts {
  u32 n[L];
  u32 r2[L];
  u32 one_m[L];
  u32 one[L];
};

__device__ __forceinline__ void load_lane_limbs(u32 (&a)[LPL], const u32* __restrict__ g, const Lane& ln) {
#pragma unroll
  for (int k = 0; k < LPL; ++k) a[k] = g[ln.q * LPL + k];
}

__device__ __forceinline__ void store_lane_limbs(u32* __restrict__ g, const u32 (&a)[LPL], const Lane& ln) {
#pragma unroll
  for (int k = 0; k < LPL; ++k) g[ln.q * LPL + k] = a[k];
}

// limb j (W bits at bit offset W j) of a 256-byte big-endian integer
__device__ __forceinline__ u32 be256_limb(const uint8_t* __restrict__ be, int j) {
  const int o = W * j;
  const int p = o >> 3, s = o & 7;
  u64 w = 0;
#pragma unroll
  for (int t = 0; t < 5; ++t) {                // W + 7 <= 40 bits
    const int idx = 255 - (p + t);
    if (idx >= 0) w |= (u64)be[idx] << (8 * t);
  }
  return (u32)(w >> s) & MASK;
}

__device__ __forceinline__ void load_be256(u32 (&a)[LPL], const uint8_t* __restrict__ be, const Lane& ln) {
#pragma unroll
  for (int k = 0; k < LPL; ++k) a[k] = be256_limb(be, (int)ln.q * LPL + k);
}

// plain integer (any value < 2^2048) -> Montgomery form, via the LDS slot
__device__ __forceinline__ void to_mont(u32 (&a)[LPL], u32* slot, const ModpConsts* __restrict__ cs,
                                        const u32 (&n)[LPL], const Lane& ln) {
  slot_fill_from_global(slot, cs->r2, ln);
  __builtin_amdgcn_wave_barrier();
  mont_mul<MODP_N0INV_C>(a, a, slot, n, ln);
  __builtin_amdgcn_wave_barrier();
}

// Montgomery (or plain, if `plain`) almost-normalised value in `a` -> canonical residue in
// [0, N) written as 256 big-endian bytes.  Uses the LDS slot as scratch.
__device__ __forceinline__ void store_canonical_be256(uint8_t* __restrict__ out, u32 (&a)[LPL], bool from_montgomery,
                                                      u32* slot, const ModpConsts* __restrict__ cs,
                                                      const u32 (&n)[LPL], const Lane& ln, bool write, int lift_parity = -1) {
  // lift_parity 0 / 1 (scalar ring, cs = the constants of q'): the canonical residue v in [0, q') is lifted to the
  // number in [0, 2q') = [0, q-1) of that parity, v or v + q' (Chinese remainders; q' is odd)
  if (from_montgomery) {
    slot_fill_from_global(slot, cs->one, ln);
    __builtin_amdgcn_wave_barrier();
    mont_mul<MODP_N0INV_C>(a, a, slot, n, ln);
    __builtin_amdgcn_wave_barrier();
  }
  slot_store(slot, a, ln);
  __builtin_amdgcn_wave_barrier();
  if (ln.q == 0) {
    // exact carry propagation
    u32 c = 0;
#pragma nounroll
    for (int j = 0; j < L; ++j) {
      const u32 v = slot[j] + c;
      slot[j] = v & MASK;
      c = v >> W;
    }
    // value < 2N: subtract N once if value >= N
    int ge = 1;  // value >= N ?  (decided by the most significant differing limb)
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
        borrow = (d >> 31) & 1;  // operands < 2^W, so a wrap sets the top bit
        slot[j] = d & MASK;
      }
    }
    if (lift_parity >= 0 && (int)(slot[0] & 1u) != lift_parity) {
      u32 carry = 0;
#pragma nounroll
      for (int j = 0; j < L; ++j) {
        const u32 v = slot[j] + cs->n[j] + carry;
        slot[j] = v & MASK;
        carry = v >> W;
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
  if (write) {
    // lane q emits little-endian 32-bit words 16q .. 16q+15 (byte-swapped, mirrored position)
    u32* out32 = reinterpret_cast<u32*>(out);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int wd = (int)ln.q * 16 + i;
      const int bit = 32 * wd;
      const int j = bit / W, s = bit % W;
      // 32 bits starting at bit s of limb j: up to three limbs (s + 32 can exceed 2 W)
      u64 two = (u64)slot[j] | ((u64)(j + 1 < L ? slot[j + 1] : 0u) << W);
      two >>= s;
      if (2 * W - s < 32) two |= (u64)(j + 2 < L ? slot[j + 2] : 0u) << (2 * W - s);
      const u32 v = (u32)two;
      out32[63 - wd] = __builtin_bswap32(v);
    }
  }
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ void square_into(u32 (&a)[LPL], u32* slot, const u32 (&n)[LPL], const Lane& ln) {
  slot_store(slot, a, ln);
  __builtin_amdgcn_wave_barrier();
  mont_sqr<MODP_N0INV_C>(a, a, slot, n, ln);
  __builtin_amdgcn_wave_barrier();
}

}  // namespace

// ---------------------------------------------------------------------------------------
// out[x] = a[x] * b[x] mod q          (Group::mul, modp.rs:130-132)
// ---------------------------------------------------------------------------------------
extern "C" __global__ void __launch_bounds__(BLOCK_THREADS) WAVES_ATTR
k_modp_mul(const uint8_t* __restrict__ a_be, const uint8_t* __restrict__ b_be, uint8_t* __restrict__ out_be,
           int count, const ModpConsts* __restrict__ cs) {
  __shared__ __attribute__((aligned(16))) u32 lds[NUMS_PER_BLOCK * SLOT_WORDS];
  const Lane ln = make_lane();
  const int xi = blockIdx.x * NUMS_PER_BLOCK + (threadIdx.x >> 2);
  const bool live = xi < count;
  const int x = live ? xi : count - 1;
  u32* slot = lds + (threadIdx.x >> 2) * SLOT_WORDS;
  u32 n[LPL], a[LPL], b[LPL];
  load_lane_limbs(n, cs->n, ln);
  load_be256(a, a_be + (size_t)x * 256, ln);
  load_be256(b, b_be + (size_t)x * 256, ln);
  to_mont(a, slot, cs, n, ln);          // aR
  slot_store(slot, b, ln);              // plain b
  __builtin_amdgcn_wave_barrier();
  mont_mul<MODP_N0INV_C>(a, a, slot, n, ln);  // a*b (plain, < 2N)
  __builtin_amdgcn_wave_barrier();
  store_canonical_be256(out_be + (size_t)x * 256, a, false, slot, cs, n, ln, live);
}

// ---------------------------------------------------------------------------------------
// plain 256-byte big-endian integers -> Montgomery limb form ([count][76] words)
// ---------------------------------------------------------------------------------------
extern "C" __global__ void __launch_bounds__(BLOCK_THREADS) WAVES_ATTR
k_modp_to_mont(const uint8_t* __restrict__ in_be, u32* __restrict__ out_m, int count,
               const ModpConsts* __restrict__ cs) {
  __shared__ __attribute__((aligned(16))) u32 lds[NUMS_PER_BLOCK * SLOT_WORDS];
  const Lane ln = make_lane();
  const int xi = blockIdx.x * NUMS_PER_BLOCK + (threadIdx.x >> 2);
  const bool live = xi < count;
  const int x = live ? xi : count - 1;
  u32* slot = lds + (threadIdx.x >> 2) * SLOT_WORDS;
  u32 n[LPL], a[LPL];
  load_lane_limbs(n, cs->n, ln);
  load_be256(a, in_be + (size_t)x * 256, ln);
  to_mont(a, slot, cs, n, ln);
  if (live) store_lane_limbs(out_m + (size_t)x * L, a, ln);
}

// =======================================================================================
// The scalar ring Z/(q-1) of the MODP group on the device (the dealer's P(i) and responses; src/polynomial.rs:50-58,
// src/dleq.rs:42-50).  q - 1 = 2 q' with q' = (q-1)/2 an odd prime: a scalar is kept as its residue mod q' -- in the
// Montgomery machinery above with the constants of q' (also -1 mod 2^29) -- and its parity, and lifted to [0, q-1) when
// it is written (store_canonical_be256, lift_parity).
// =======================================================================================

// a += b limb by limb, then carries inside the lane and one hand-over to the next lane: almost normalised again
__device__ __forceinline__ void add_limbs(u32 (&a)[LPL], const u32 (&b)[LPL], const Lane& ln) {
  u32 c = 0;
#pragma unroll
  for (int k = 0; k < LPL; ++k) {
    const u32 v = a[k] + b[k] + c;
    a[k] = v & MASK;
    c = v >> W;
  }
  a[0] += quad_from_prev(c) & ln.not_low;       // values stay far below 2^2088: the top lane has no carry-out
}

// out[x] = P(positions[x]) mod (q-1), P = sum_j a_j X^j.  Horner's rule with the position as a SMALL multiplier: 18 rows of
// the Montgomery product instead of 72 (positions < 2^63 are three limbs), which leaves a factor 2^-522 per step -- the
// caller hands the coefficients over as a'_j = a_j 2^(522 j) mod q' (plain limbs), so the factors cancel.
//   coef        : [t][72] limbs of a'_j
//   par_even/odd: parity of P at even / odd positions (a_0 resp. the sum of all a_j, mod 2)
extern "C" __global__ void __launch_bounds__(BLOCK_THREADS) WAVES_ATTR
k_modq_poly_eval(const u32* __restrict__ coef, int t, const int64_t* __restrict__ positions, int count, int par_even, int par_odd,
                 uint8_t* __restrict__ out_be, const ModpConsts* __restrict__ cs) {
  __shared__ __attribute__((aligned(16))) u32 lds[NUMS_PER_BLOCK * SLOT_WORDS];
  const Lane ln = make_lane();
  const int xi = blockIdx.x * NUMS_PER_BLOCK + (threadIdx.x >> 2);
  const bool live = xi < count;
  const int x = live ? xi : count - 1;
  u32* slot = lds + (threadIdx.x >> 2) * SLOT_WORDS;
  const u64 pos = (u64)positions[x];
  u32 n[LPL], acc[LPL], cj[LPL];
  load_lane_limbs(n, cs->n, ln);
#pragma unroll
  for (int k = 0; k < LPL; ++k) {
    const int j = (int)ln.q * LPL + k;
    slot[j] = j < 3 ? (u32)(pos >> (W * j)) & MASK : 0u;
  }
  __builtin_amdgcn_wave_barrier();
  load_lane_limbs(acc, coef + (size_t)(t - 1) * L, ln);
  for (int j = t - 2; j >= 0; --j) {
    load_lane_limbs(cj, coef + (size_t)j * L, ln);
    mont_mul<MODP_N0INV_C, false, 1>(acc, acc, slot, n, ln);
    add_limbs(acc, cj, ln);
  }
  __builtin_amdgcn_wave_barrier();
  store_canonical_be256(out_be + (size_t)x * 256, acc, false, slot, cs, n, ln, live, (pos & 1) ? par_odd : par_even);
}

// r[x] = w[x] - alpha[x] c mod (q-1) for one shared c (dleq.rs:42-50, participant.rs:255-264).
//   cneg_be : (q' - c mod q') mod q' as 256 big-endian bytes, c_parity = c mod 2
extern "C" __global__ void __launch_bounds__(BLOCK_THREADS) WAVES_ATTR
k_modq_responses(const uint8_t* __restrict__ w_be, const uint8_t* __restrict__ alpha_be, const uint8_t* __restrict__ cneg_be,
                 int c_parity, int count, uint8_t* __restrict__ out_be, const ModpConsts* __restrict__ cs) {
  __shared__ __attribute__((aligned(16))) u32 lds[NUMS_PER_BLOCK * SLOT_WORDS];
  const Lane ln = make_lane();
  const int xi = blockIdx.x * NUMS_PER_BLOCK + (threadIdx.x >> 2);
  const bool live = xi < count;
  const int x = live ? xi : count - 1;
  u32* slot = lds + (threadIdx.x >> 2) * SLOT_WORDS;
  u32 n[LPL], a[LPL], b[LPL];
  load_lane_limbs(n, cs->n, ln);
  load_be256(a, alpha_be + (size_t)x * 256, ln);
  const int parity = (int)((w_be[(size_t)x * 256 + 255] ^ (alpha_be[(size_t)x * 256 + 255] & (uint8_t)c_parity)) & 1u);
  to_mont(a, slot, cs, n, ln);                       // alpha R mod q'
  load_be256(b, cneg_be, ln);
  slot_store(slot, b, ln);
  __builtin_amdgcn_wave_barrier();
  mont_mul<MODP_N0INV_C>(a, a, slot, n, ln);         // -alpha c mod q' (plain, < 2q')
  __builtin_amdgcn_wave_barrier();
  load_be256(b, w_be + (size_t)x * 256, ln);
  add_limbs(a, b, ln);                               // w - alpha c, < 2^2050
  slot_fill_from_global(slot, cs->one_m, ln);
  __builtin_amdgcn_wave_barrier();
  mont_mul<MODP_N0INV_C>(a, a, slot, n, ln);         // times R R^-1: the same residue, below 2q'
  __builtin_amdgcn_wave_barrier();
  store_canonical_be256(out_be + (size_t)x * 256, a, false, slot, cs, n, ln, live, parity);
}

// out[x] = a[x] * b[x] mod (q-1): the scalar ring's product per share (Group::scalar_mul, modp.rs:180-182) -- the
// participant's second exponent w_i / x_i of a batched extract_secret_share (participant.rs:310-323) without host work.
// Residue mod q' by one Montgomery product of a R with the plain b, parity = that of the integer product (q - 1 is even).
extern "C" __global__ void __launch_bounds__(BLOCK_THREADS) WAVES_ATTR
k_modq_mul(const uint8_t* __restrict__ a_be, const uint8_t* __restrict__ b_be, int count, uint8_t* __restrict__ out_be,
           const ModpConsts* __restrict__ cs) {
  __shared__ __attribute__((aligned(16))) u32 lds[NUMS_PER_BLOCK * SLOT_WORDS];
  const Lane ln = make_lane();
  const int xi = blockIdx.x * NUMS_PER_BLOCK + (threadIdx.x >> 2);
  const bool live = xi < count;
  const int x = live ? xi : count - 1;
  u32* slot = lds + (threadIdx.x >> 2) * SLOT_WORDS;
  u32 n[LPL], a[LPL], b[LPL];
  load_lane_limbs(n, cs->n, ln);
  load_be256(a, a_be + (size_t)x * 256, ln);
  const int parity = (int)(a_be[(size_t)x * 256 + 255] & b_be[(size_t)x * 256 + 255] & 1u);
  to_mont(a, slot, cs, n, ln);                       // a R mod q'
  load_be256(b, b_be + (size_t)x * 256, ln);
  slot_store(slot, b, ln);
  __builtin_amdgcn_wave_barrier();
  mont_mul<MODP_N0INV_C>(a, a, slot, n, ln);         // a b mod q' (plain, < 2q')
  __builtin_amdgcn_wave_barrier();
  store_canonical_be256(out_be + (size_t)x * 256, a, false, slot, cs, n, ln, live, parity);
}

// ---------------------------------------------------------------------------------------
// X_i = prod_j C_j^(i^j)   (participant.rs:423-434) evaluated by Horner's rule in the
// exponent:  X_i = (..((C_{t-1})^i * C_{t-2})^i .. )^i * C_0 .  Identical group element for
// every C_j (C_j^(i^j mod (q-1)) == C_j^(i^j) for units, and both sides are 0 when some
// C_j == 0 mod q because every i^j >= 1), hence identical canonical bytes.
//   cm        : commitments in Montgomery limb form [t][76]
//   positions : i (>= 0) per share
//   x_be      : X_i canonical 256-byte big-endian
// The whole evaluation is one loop around a single Montgomery-product site: every step only
// chooses which LDS operand (own copy = square, saved base, Montgomery one, C_j) it multiplies by.
// LDS per wave: operand slot + saved-base slot per number, one shared slot holding one_m.
// ---------------------------------------------------------------------------------------
extern "C" __global__ void __launch_bounds__(BLOCK_THREADS) WAVES_ATTR
k_modp_commit_eval(const u32* __restrict__ cm_a, const u32* __restrict__ cm_b, int split, int t,
                   const int64_t* __restrict__ positions, int count, uint8_t* __restrict__ x_be,
                   u32* __restrict__ x_m, const int* __restrict__ gate, int gate_want,
                   const ModpConsts* __restrict__ cs, size_t box_cm_words, size_t box_positions, size_t box_out) {
  __shared__ __attribute__((aligned(16))) u32 lds[(2 * NUMS_PER_BLOCK + MODP_WPB) * SLOT_WORDS];
  // blockIdx.y: box of a group of same-shaped boxes evaluated by one launch (own commitments, positions and output rows)
  cm_a += blockIdx.y * box_cm_words;
  cm_b += blockIdx.y * box_cm_words;
  positions += blockIdx.y * box_positions;
  if (x_be != nullptr) x_be += blockIdx.y * box_out * 256;
  if (x_m != nullptr) x_m += blockIdx.y * box_out * L;
  // gate: the forward-difference path (below) and this kernel exclude each other through a device flag,
  // so that the choice needs no host synchronisation
  if (gate != nullptr && *gate != gate_want) return;
  if (gate != nullptr && x_m != nullptr) __builtin_amdgcn_s_setprio(MODP_SETPRIO);   // seed launch of the forward-difference path
  const Lane ln = make_lane();
  const int xi = blockIdx.x * NUMS_PER_BLOCK + (threadIdx.x >> 2);
  // numbers >= split evaluate the second commitment set (the inverted commitments of the seed phase)
  const u32* __restrict__ cm = (xi >= split) ? cm_b : cm_a;
  const bool live = xi < count;
  const int x = live ? xi : count - 1;
  u32* slot = lds + (threadIdx.x >> 2) * SLOT_WORDS;
  u32* bslot = lds + (NUMS_PER_BLOCK + (threadIdx.x >> 2)) * SLOT_WORDS;
  u32* oneslot = lds + (2 * NUMS_PER_BLOCK + (threadIdx.x >> 6)) * SLOT_WORDS;
  u32 n[LPL], acc[LPL];
  load_lane_limbs(n, cs->n, ln);
  if ((threadIdx.x & 63) < 4) slot_fill_from_global(oneslot, cs->one_m, ln);
  const u64 pos = (u64)positions[x];
  // wave-wide maximum bit length of the positions
  int nb = (pos == 0) ? 0 : 64 - __builtin_clzll(pos);
#pragma unroll
  for (int off = 32; off >= 4; off >>= 1) {
    const int other = __shfl_xor(nb, off);
    nb = other > nb ? other : nb;
  }
  nb = __builtin_amdgcn_readfirstlane(nb);
  __builtin_amdgcn_wave_barrier();

  load_lane_limbs(acc, cm + (size_t)(t - 1) * L, ln);
  // Program (one Montgomery-product site):
  //   for j = t-2 .. 0:   base = acc; acc = topbit ? base : one
  //                       for bit = nb-2 .. 0: SQUARE; CONDMUL (by base or one, skipped if no lane needs it)
  //                       CMUL (by C_j)
  //   FINAL (by plain 1: leave the Montgomery domain)
  enum { K_SQUARE, K_CONDMUL, K_CMUL, K_FINAL };
  int j = t - 2, bit = 0, kind = K_FINAL;
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
  while (true) {
    const u32* bptr = slot;
    bool skip = false;
    if (kind == K_SQUARE) {
      slot_store(slot, acc, ln);
    } else if (kind == K_CONDMUL) {
      const bool mine = (pos >> bit) & 1;
      skip = __builtin_amdgcn_ballot_w64(mine) == 0;
      bptr = mine ? bslot : oneslot;
    } else if (kind == K_CMUL) {
      slot_fill_from_global(slot, cm + (size_t)j * L, ln);
    } else {
      if (x_m != nullptr) break;                      // keep the Montgomery form (seed values)
      slot_fill_from_global(slot, cs->one, ln);
    }
    if (!skip) {
      __builtin_amdgcn_wave_barrier();
      if (kind == K_SQUARE) mont_sqr<MODP_N0INV_C>(acc, acc, slot, n, ln);
      else mont_mul<MODP_N0INV_C>(acc, acc, bptr, n, ln);
      __builtin_amdgcn_wave_barrier();
    }
    if (kind == K_FINAL) break;
    if (kind == K_SQUARE) {
      kind = K_CONDMUL;
    } else if (kind == K_CONDMUL) {
      --bit;
      kind = (bit >= 0) ? K_SQUARE : K_CMUL;
    } else {
      --j;
      if (j >= 0) begin_coefficient(); else kind = K_FINAL;
    }
  }
  if (x_m != nullptr) {
    if (live) store_lane_limbs(x_m + (size_t)x * L, acc, ln);
    return;
  }
  store_canonical_be256(x_be + (size_t)x * 256, acc, false, slot, cs, n, ln, live);
}

// =======================================================================================
// Forward differences in the exponent (consecutive positions only).
// X(i) = g^P(i) with deg P = t-1, so the t-th multiplicative difference of the sequence X(c), X(c+1), ..
// is constant: with D_k(c) = "Delta^k X (c)" (D_0 = X, D_k(c) = D_{k-1}(c+1) / D_{k-1}(c)),
//   D_k(c+1) = D_k(c) * D_{k+1}(c)   for k < t-1,   D_{t-1} constant,
// i.e. ONE Montgomery product per share and coefficient instead of ~24 for Horner's rule.  A chain of
// consecutive positions is served by t numbers (one per k) that step in lock-step inside one workgroup.
//   seeds  : X(c+k) and X(c+k)^-1, k < t, for each chain start c -- by the Horner kernel above, the inverses as
//            the same polynomial over the inverted commitments C_j^-1
//   table  : E_l[k] = E_{l-1}[k+1] F_{l-1}[k],  F_l[k] = F_{l-1}[k+1] E_{l-1}[k]  (E_0 = X, F_0 = X^-1); D_l = E_l[0]
//   step   : D_k <- D_k * D_{k+1}, output D_0
// Canonical results are identical to Horner's by uniqueness of the group element.
// tpad = t rounded up to a power of two >= 16 (whole waves of 16 levels).
// =======================================================================================

// flag = 1 iff positions[i] == positions[0] + i for all i (and no negative / overflowing value)
extern "C" __global__ void k_modp_fd_check_positions(const int64_t* __restrict__ positions, int count,
                                                      int* __restrict__ flag, size_t box_positions) {
  positions += blockIdx.y * box_positions;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const int64_t p0 = positions[0];
  const bool ok = p0 >= 0 && p0 < ((int64_t)1 << 61) && positions[i] == p0 + i;
  if (!ok) atomicAnd(flag, 0);
}

// ---- simultaneous inversion (Montgomery's trick) of m numbers in Montgomery form ----------------------
// One quad per group of G consecutive numbers.  up: prefix[i] = a[first] * .. * a[i] inside the group, the
// group total goes to totals[g] (the next level's input).  down: given the inverse of the group total,
//   inv(a[i]) = running * prefix[i-1],  running *= a[i]   for i = last .. first+1,   inv(a[first]) = running.
// Three products per number; the single real inversion (of the root) is done by the host in stream order.
extern "C" __global__ void __launch_bounds__(BLOCK_THREADS) WAVES_ATTR
k_modp_binv_up(const u32* __restrict__ a, int m, int G, u32* __restrict__ prefix, u32* __restrict__ totals,
               const int* __restrict__ gate, const ModpConsts* __restrict__ cs) {
  __shared__ __attribute__((aligned(16))) u32 lds[NUMS_PER_BLOCK * SLOT_WORDS];
  if (*gate != 1) return;
  __builtin_amdgcn_s_setprio(MODP_SETPRIO);   // latency-critical and few: issue ahead of the wide kernels sharing the SIMD
  const Lane ln = make_lane();
  const int groups = (m + G - 1) / G;
  const int gi = blockIdx.x * NUMS_PER_BLOCK + (threadIdx.x >> 2);
  const bool live = gi < groups;
  const int g = live ? gi : groups - 1;
  const int first = g * G;
  const int cnt = (m - first < G) ? m - first : G;
  u32* slot = lds + (threadIdx.x >> 2) * SLOT_WORDS;
  u32 n[LPL], acc[LPL], tmp[LPL];
  load_lane_limbs(n, cs->n, ln);
  load_lane_limbs(acc, a + (size_t)first * L, ln);
  if (live) store_lane_limbs(prefix + (size_t)first * L, acc, ln);
  for (int i = 1; i < G; ++i) {
    const bool act = i < cnt;
    const int idx = first + (act ? i : cnt - 1);
    slot_fill_from_global(slot, a + (size_t)idx * L, ln);
    __builtin_amdgcn_wave_barrier();
    mont_mul<MODP_N0INV_C>(tmp, acc, slot, n, ln);
    __builtin_amdgcn_wave_barrier();
    if (act) {
#pragma unroll
      for (int k = 0; k < LPL; ++k) acc[k] = tmp[k];
      if (live) store_lane_limbs(prefix + (size_t)idx * L, acc, ln);
    }
  }
  if (live) store_lane_limbs(totals + (size_t)g * L, acc, ln);
}

extern "C" __global__ void __launch_bounds__(BLOCK_THREADS) WAVES_ATTR
k_modp_binv_down(const u32* __restrict__ a, const u32* __restrict__ prefix, const u32* __restrict__ tot_inv, int m,
                 int G, u32* __restrict__ a_inv, const int* __restrict__ gate, const ModpConsts* __restrict__ cs) {
  __shared__ __attribute__((aligned(16))) u32 lds[NUMS_PER_BLOCK * SLOT_WORDS];
  if (*gate != 1) return;
  __builtin_amdgcn_s_setprio(MODP_SETPRIO);   // latency-critical and few: issue ahead of the wide kernels sharing the SIMD
  const Lane ln = make_lane();
  const int groups = (m + G - 1) / G;
  const int gi = blockIdx.x * NUMS_PER_BLOCK + (threadIdx.x >> 2);
  const bool live = gi < groups;
  const int g = live ? gi : groups - 1;
  const int first = g * G;
  const int cnt = (m - first < G) ? m - first : G;
  u32* slot = lds + (threadIdx.x >> 2) * SLOT_WORDS;
  u32 n[LPL], run[LPL], tmp[LPL];
  load_lane_limbs(n, cs->n, ln);
  load_lane_limbs(run, tot_inv + (size_t)g * L, ln);
  // 2 (G - 1) products through one site: odd steps emit inv(a[i]) = run * prefix[i-1], even steps do run *= a[i]
  for (int s = 2 * (G - 1) - 1; s >= 0; --s) {
    const int i = (s >> 1) + 1;
    const bool emit = (s & 1) != 0;
    const bool act = i < cnt;
    const int ii = act ? i : (cnt > 1 ? cnt - 1 : 1);
    const u32* src = emit ? prefix + (size_t)(first + ii - 1) * L : a + (size_t)(first + (cnt > 1 ? ii : 0)) * L;
    slot_fill_from_global(slot, src, ln);
    __builtin_amdgcn_wave_barrier();
    mont_mul<MODP_N0INV_C>(tmp, run, slot, n, ln);
    __builtin_amdgcn_wave_barrier();
    if (act) {
      if (emit) {
        if (live) store_lane_limbs(a_inv + (size_t)(first + i) * L, tmp, ln);
      } else {
#pragma unroll
        for (int k = 0; k < LPL; ++k) run[k] = tmp[k];
      }
    }
  }
  if (live) store_lane_limbs(a_inv + (size_t)first * L, run, ln);
}

// flag <- 0 when the host found the root of the inversion tree to be 0 mod q (some X is 0: no inverses)
extern "C" __global__ void k_modp_fd_apply_ok(const int* __restrict__ ok, int* __restrict__ flag) {
  if (threadIdx.x == 0 && blockIdx.x == 0 && *ok != 1) *flag = 0;
}

// Chain c (of `chains`) owns the positions c, c + chains, c + 2 chains, ..: its seeds are x[c + chains k], k < t,
// i.e. the first chains*t values of X are the seeds of all chains and are outputs at the same time.
//
// Both kernels below are PIPELINES OF SINGLE-WAVE WORKGROUPS.  The t numbers of a chain (one per level k) are cut
// into stages of 16 consecutive levels = one wave = one workgroup; level k only ever needs level k+1 of the previous
// step, so a stage needs exactly one number per step from the stage above it and nothing from below.  That number
// travels through HBM as self-validating words (bit 31 set on every limb word of a buffer zeroed before the
// launch; limbs are < 2^29), written and polled with agent-scope relaxed atomics: no fences, no barriers, no
// cross-workgroup synchronisation other than the data itself.  Stages above are dispatched first (lower block
// index), so a waiting stage only ever waits for workgroups that are already resident or done.  A stage that
// waits longer than FD_TIMEOUT_TICKS clears the device flag -- the gated Horner kernel then recomputes everything --
// and poisons its own output so that the stages below give up at once.
// One step then costs one Montgomery product of ONE wave per SIMD instead of four waves sharing a SIMD behind a
// workgroup barrier, and the 128-register ceiling of 1024-thread workgroups is gone.
constexpr u32 HAND_VALID = 0x80000000u;
constexpr u32 HAND_POISON = 0x40000000u;
constexpr long long FD_TIMEOUT_TICKS = 200000000LL;   // 2 s of the 100 MHz wall clock

__device__ __forceinline__ void hand_publish(u32* __restrict__ dst, const u32 (&a)[LPL], const Lane& ln, u32 tag) {
#pragma unroll
  for (int k = 0; k < LPL; ++k)
    __hip_atomic_store(dst + ln.q * LPL + k, a[k] | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// one polling round: this lane's 19 words; returns HAND_VALID if all are valid, HAND_POISON if poisoned, else 0
__device__ __forceinline__ u32 hand_poll(const u32* __restrict__ src, u32 (&v)[LPL], const Lane& ln) {
  u32 all = 0xffffffffu, any = 0;
#pragma unroll
  for (int k = 0; k < LPL; ++k) {
    v[k] = __hip_atomic_load(src + ln.q * LPL + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    all &= v[k];
    any |= v[k];
  }
  return (any & HAND_POISON) ? HAND_POISON : (all & HAND_VALID);
}

// The quad `is_reader` of the wave waits for one number from the stage above and puts it into the LDS slot `dst`.
// Returns false (wave-uniform) when the wait failed: poisoned input or timeout.
__device__ __forceinline__ bool hand_receive(const u32* __restrict__ src, u32* dst, bool is_reader, const Lane& ln) {
  u32 v[LPL];
  bool ok = true;
  if (is_reader) {
    const long long t0 = wall_clock64();
    while (true) {
      const u32 st = hand_poll(src, v, ln);
      // all four lanes of the quad must agree (the reader quad is lanes 60..63)
      const uint64_t good = __builtin_amdgcn_ballot_w64(st == HAND_VALID) >> 60;
      const uint64_t bad = __builtin_amdgcn_ballot_w64(st == HAND_POISON) >> 60;
      if (good == 0xf) break;
      if (bad != 0 || wall_clock64() - t0 > FD_TIMEOUT_TICKS) { ok = false; break; }
      __builtin_amdgcn_s_sleep(4);
    }
#pragma unroll
    for (int k = 0; k < LPL; ++k) dst[ln.q * LPL + k] = v[k] & 0x3fffffffu;
  }
  return __builtin_amdgcn_ballot_w64(!ok) == 0;
}

// The same for a number whose words were REQUESTED EARLIER (v holds what the loads returned: the stepping kernel asks for the
// entry of the next step under this step's product, so that the hand-over's memory latency -- several microseconds when the
// wide launches keep HBM busy -- is not part of the step); polls as hand_receive does when the words are not there yet.
__device__ __forceinline__ bool hand_receive_requested(u32 (&v)[LPL], const u32* __restrict__ src, u32* dst, bool is_reader,
                                                       const Lane& ln) {
  bool ok = true;
  if (is_reader) {
    const long long t0 = wall_clock64();
    u32 all = 0xffffffffu, any = 0;
#pragma unroll
    for (int k = 0; k < LPL; ++k) {
      all &= v[k];
      any |= v[k];
    }
    u32 st = (any & HAND_POISON) ? HAND_POISON : (all & HAND_VALID);
    while (true) {
      const uint64_t good = __builtin_amdgcn_ballot_w64(st == HAND_VALID) >> 60;
      const uint64_t bad = __builtin_amdgcn_ballot_w64(st == HAND_POISON) >> 60;
      if (good == 0xf) break;
      if (bad != 0 || wall_clock64() - t0 > FD_TIMEOUT_TICKS) { ok = false; break; }
      __builtin_amdgcn_s_sleep(4);
      st = hand_poll(src, v, ln);
    }
#pragma unroll
    for (int k = 0; k < LPL; ++k) dst[ln.q * LPL + k] = v[k] & 0x3fffffffu;
  }
  return __builtin_amdgcn_ballot_w64(!ok) == 0;
}

// (two products per level keep E, F, T1 and the modulus live: 2 waves per SIMD's worth of registers instead of 3 --
// the launch is a few dozen latency-bound waves, occupancy is not its problem, spilling would be)
extern "C" __global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 2)))
k_modp_fd_table(const u32* __restrict__ x, const u32* __restrict__ x_inv, int chains, int t, int tpad,
                u32* __restrict__ state, u32* __restrict__ state_back, u32* __restrict__ hand, int* __restrict__ gate,
                int inject_fault, const ModpConsts* __restrict__ cs, size_t box_x, size_t box_xinv, size_t box_state,
                size_t box_hand) {
  // blockIdx.y: box of a group (word strides; the stages of one box keep their dispatch order, x runs fastest)
  x += blockIdx.y * box_x;
  x_inv += blockIdx.y * box_xinv;
  state += blockIdx.y * box_state;
  state_back += blockIdx.y * box_state;
  hand += blockIdx.y * box_hand;
  // state[c][l]      = E_l[0]                       = D_l at the first seed: the chain that steps forward
  // state_back[c][l] = (l even ? E_l : F_l)[t-1-l]  = g^((-1)^l nabla^l P) at the last seed: with H_l(i-1) = H_l(i) H_{l+1}(i)
  //                    the same recurrence steps BACKWARD from the last seed, H_0 being X (one seed window serves
  //                    both directions, so half the seeds cover the same positions)
  __shared__ __attribute__((aligned(16))) u32 lds[(2 * NUMS_PER_WAVE + 2) * SLOT_WORDS];
  if (*gate != 1) return;
  __builtin_amdgcn_s_setprio(MODP_SETPRIO);   // latency-critical and few: issue ahead of the wide kernels sharing the SIMD
  const Lane ln = make_lane();
  const int quad = threadIdx.x >> 2;
  const int stages = tpad / NUMS_PER_WAVE;
  const int sidx = blockIdx.x / chains;                       // 0 = the top levels
  const int chain = blockIdx.x % chains;
  const int kbase = tpad - NUMS_PER_WAVE * (sidx + 1);
  if (kbase >= t) return;                                     // nothing but padding in this stage
  const int k = kbase + quad;
  const int kk = k < t ? k : t - 1;
  const bool has_up = kbase + NUMS_PER_WAVE < t;
  const bool has_down = kbase > 0;
  const int last_lvl = t - 1 - kbase;                         // level l needs k <= t-1-l
  u32* eslot = lds + quad * SLOT_WORDS;
  u32* fslot = lds + (NUMS_PER_WAVE + quad) * SLOT_WORDS;
  u32* in_e = lds + 2 * NUMS_PER_WAVE * SLOT_WORDS;
  u32* in_f = in_e + SLOT_WORDS;
  const bool reader = quad == NUMS_PER_WAVE - 1;
  const u32* enb = reader ? in_e : eslot + SLOT_WORDS;        // E_{l-1}[k+1]
  const u32* fnb = reader ? in_f : fslot + SLOT_WORDS;        // F_{l-1}[k+1]
  // handoff areas: [chain][stage][level][E,F]
  u32* mine = hand + ((size_t)chain * stages + sidx) * (size_t)t * 2 * L;
  const u32* up = hand + ((size_t)chain * stages + (sidx - 1)) * (size_t)t * 2 * L;
  u32 n[LPL], E[LPL], F[LPL], T1[LPL];
  load_lane_limbs(n, cs->n, ln);
  load_lane_limbs(E, x + ((size_t)chain + (size_t)chains * kk) * L, ln);
  load_lane_limbs(F, x_inv + ((size_t)chain + (size_t)chains * kk) * L, ln);
  u32* st = state + (size_t)chain * t * L;
  u32* sb = state_back + (size_t)chain * t * L;
  if (kbase == 0 && quad == 0) store_lane_limbs(st, E, ln);
  if (k == t - 1) store_lane_limbs(sb, E, ln);
  if (has_up && reader) {                                     // level 0 of the stage above: its seeds
    const size_t src = ((size_t)chain + (size_t)chains * (kbase + NUMS_PER_WAVE)) * L;
#pragma unroll
    for (int i = 0; i < LPL; ++i) {
      in_e[ln.q * LPL + i] = x[src + ln.q * LPL + i];
      in_f[ln.q * LPL + i] = x_inv[src + ln.q * LPL + i];
    }
  }
  for (int lvl = 1; lvl <= last_lvl; ++lvl) {
    slot_store(eslot, E, ln);
    slot_store(fslot, F, ln);
    const bool need_up = has_up && (kbase + NUMS_PER_WAVE - 1 <= t - 1 - lvl);
    if (inject_fault == 4 && sidx == 0 && chain == 0 && lvl == 3) {      // test hook: the top stage gives up
      if (threadIdx.x == 0) *gate = 0;
      if (has_down && quad == 0) hand_publish(mine + (size_t)lvl * 2 * L, E, ln, HAND_POISON);
      return;
    }
    if (need_up && lvl > 1) {
      const u32* src = up + (size_t)(lvl - 1) * 2 * L;
      const bool ok = hand_receive(src, in_e, reader, ln) && hand_receive(src + L, in_f, reader, ln);
      if (!ok) {
        if (threadIdx.x == 0) *gate = 0;
        if (has_down && quad == 0) hand_publish(mine + (size_t)lvl * 2 * L, E, ln, HAND_POISON);
        return;
      }
    }
    __builtin_amdgcn_wave_barrier();
    mont_mul<MODP_N0INV_C>(T1, F, enb, n, ln);          // E_l[k] = E_{l-1}[k+1] * F_{l-1}[k]
    mont_mul<MODP_N0INV_C>(F, E, fnb, n, ln);           // F_l[k] = F_{l-1}[k+1] * E_{l-1}[k]
#pragma unroll
    for (int i = 0; i < LPL; ++i) E[i] = T1[i];
    __builtin_amdgcn_wave_barrier();
    if (has_down && quad == 0) {
      hand_publish(mine + (size_t)lvl * 2 * L, E, ln, HAND_VALID);
      hand_publish(mine + (size_t)lvl * 2 * L + L, F, ln, HAND_VALID);
    }
    if (kbase == 0 && quad == 0) store_lane_limbs(st + (size_t)lvl * L, E, ln);   // D_l = E_l[0]
    if (k == t - 1 - lvl) {                                                        // the other diagonal
      if (lvl & 1) store_lane_limbs(sb + (size_t)lvl * L, F, ln); else store_lane_limbs(sb + (size_t)lvl * L, E, ln);
    }
  }
}

// Chain c owns x_m[c + chains j], j < chain_len; its seeds are j = w0 .. w0+t-1 (already in x_m).  Two pipelines per
// chain: direction 0 steps forward from the first seed (after s steps its level 0 is X at j = w0 + s), direction 1
// backward from the last seed (X at j = w0 + t - 1 - s).  The first t-1 steps of either only advance the table.
extern "C" __global__ void __launch_bounds__(64) WAVES_ATTR
k_modp_fd_step(const u32* __restrict__ state, const u32* __restrict__ state_back, int chains, int t, int tpad, int w0,
               int chain_len, int count, u32* __restrict__ x_m, u32* __restrict__ hand, int* __restrict__ gate,
               int inject_fault, const ModpConsts* __restrict__ cs, size_t box_state, size_t box_xm, size_t box_hand) {
  state += blockIdx.y * box_state;
  state_back += blockIdx.y * box_state;
  x_m += blockIdx.y * box_xm;
  hand += blockIdx.y * box_hand;
  __shared__ __attribute__((aligned(16))) u32 lds[(NUMS_PER_WAVE + 2) * SLOT_WORDS];
  if (*gate != 1) return;
  __builtin_amdgcn_s_setprio(MODP_SETPRIO);   // latency-critical and few: issue ahead of the wide kernels sharing the SIMD
  const Lane ln = make_lane();
  const int quad = threadIdx.x >> 2;
  const int stages = tpad / NUMS_PER_WAVE;
  // block order: all stages of level group 0 (the top levels) of both directions first, then the next group, ..
  const int sidx = blockIdx.x / (2 * chains);
  const int dir = (blockIdx.x / chains) & 1;
  const int chain = blockIdx.x % chains;
  const int kbase = tpad - NUMS_PER_WAVE * (sidx + 1);
  if (kbase >= t || (dir == 1 && w0 == 0)) return;                  // no positions before the seeds: nothing to do
  const int steps = dir == 0 ? chain_len - 1 - w0 : w0 + t - 1;     // last forward j = chain_len-1, last backward j = 0
  const int hand_len = chain_len + t;                               // room for either direction
  const int k = kbase + quad;
  const bool has_up = kbase + NUMS_PER_WAVE < t;
  const bool has_down = kbase > 0;
  u32* slot = lds + quad * SLOT_WORDS;
  u32* inslot = lds + NUMS_PER_WAVE * SLOT_WORDS;
  u32* oneslot = inslot + SLOT_WORDS;
  const bool reader = quad == NUMS_PER_WAVE - 1;
  const u32* bptr = (k + 1 < t) ? (reader ? inslot : slot + SLOT_WORDS) : oneslot;
  const size_t lane_area = ((size_t)dir * chains + chain) * stages;
  u32* mine = hand + (lane_area + sidx) * (size_t)hand_len * L;
  const u32* up = hand + (lane_area + sidx - 1) * (size_t)hand_len * L;
  const u32* st = (dir == 0 ? state : state_back) + (size_t)chain * t * L;
  u32 n[LPL], D[LPL];
  load_lane_limbs(n, cs->n, ln);
  if (threadIdx.x < 4) slot_fill_from_global(oneslot, cs->one_m, ln);
  if (k < t) load_lane_limbs(D, st + (size_t)k * L, ln); else load_lane_limbs(D, cs->one_m, ln);
  if (has_up && reader) {                                     // step 0 of the stage above: its table entry
    const u32* src = st + (size_t)(kbase + NUMS_PER_WAVE) * L;
#pragma unroll
    for (int i = 0; i < LPL; ++i) inslot[ln.q * LPL + i] = src[ln.q * LPL + i];
  }
  const bool writer = kbase == 0 && quad == 0;
#if FD_STEP_PREFETCH
  u32 pre[LPL];
#pragma unroll
  for (int i = 0; i < LPL; ++i) pre[i] = 0;
  if (has_up && reader && steps >= 2) hand_poll(up + (size_t)1 * L, pre, ln);
#endif
  for (int step = 1; step <= steps; ++step) {
    slot_store(slot, D, ln);
    // test hooks (MPVSS_FD_TEST_FAULT): behave like a stage whose wait timed out -- 1: the top stage of the first forward
    // chain early on; 2: a middle stage of the last backward chain, later
    const bool faulty = (inject_fault == 1 && !has_up && dir == 0 && chain == 0 && step == 5) ||
                        (inject_fault == 2 && dir == 1 && chain == chains - 1 && sidx == (stages > 1 ? 1 : 0) && step == 40);
    if (faulty) {
      if (threadIdx.x == 0) *gate = 0;
      if (has_down && quad == 0) hand_publish(mine + (size_t)step * L, D, ln, HAND_POISON);
      return;
    }
    if (has_up && step > 1) {
#if FD_STEP_PREFETCH
      const bool got = hand_receive_requested(pre, up + (size_t)(step - 1) * L, inslot, reader, ln);
#else
      const bool got = hand_receive(up + (size_t)(step - 1) * L, inslot, reader, ln);
#endif
      if (!got) {
        if (threadIdx.x == 0) *gate = 0;
        if (has_down && quad == 0) hand_publish(mine + (size_t)step * L, D, ln, HAND_POISON);
        return;
      }
#if FD_STEP_PREFETCH
      if (reader && step < steps) hand_poll(up + (size_t)step * L, pre, ln);      // the next step's, under this step's product
#endif
    }
    __builtin_amdgcn_wave_barrier();
    mont_mul<MODP_N0INV_C>(D, D, bptr, n, ln);
    __builtin_amdgcn_wave_barrier();
    if (has_down && quad == 0) hand_publish(mine + (size_t)step * L, D, ln, HAND_VALID);
    const int j = dir == 0 ? w0 + step : w0 + t - 1 - step;
    const size_t idx = (size_t)chain + (size_t)chains * j;
    if (writer && step >= t && idx < (size_t)count) store_lane_limbs(x_m + idx * L, D, ln);
  }
}

// Montgomery limb form -> canonical 256-byte big-endian
extern "C" __global__ void __launch_bounds__(BLOCK_THREADS) WAVES_ATTR
k_modp_from_mont(const u32* __restrict__ x_m, int count, uint8_t* __restrict__ out_be, const int* __restrict__ gate,
                 const ModpConsts* __restrict__ cs) {
  __shared__ __attribute__((aligned(16))) u32 lds[NUMS_PER_BLOCK * SLOT_WORDS];
  if (gate != nullptr && *gate != 1) return;
  const Lane ln = make_lane();
  const int xi = blockIdx.x * NUMS_PER_BLOCK + (threadIdx.x >> 2);
  const bool live = xi < count;
  const int x = live ? xi : count - 1;
  u32* slot = lds + (threadIdx.x >> 2) * SLOT_WORDS;
  u32 n[LPL], a[LPL];
  load_lane_limbs(n, cs->n, ln);
  load_lane_limbs(a, x_m + (size_t)x * L, ln);
  store_canonical_be256(out_be + (size_t)x * 256, a, true, slot, cs, n, ln, live);
}

// ---------------------------------------------------------------------------------------
// Window table: tab[x][e] = base_x^e (Montgomery form), e = 0..15.
//   base_be : [count][256] big-endian (stride 0 when `count` == 1 shared base)
// ---------------------------------------------------------------------------------------
extern "C" __global__ void __launch_bounds__(BLOCK_THREADS) WAVES_ATTR
k_modp_build_table(const uint8_t* __restrict__ base_be, int count, u32* __restrict__ tab,
                   const ModpConsts* __restrict__ cs, int odd_only) {
  __shared__ __attribute__((aligned(16))) u32 lds[NUMS_PER_BLOCK * SLOT_WORDS];
  const Lane ln = make_lane();
  const int xi = blockIdx.x * NUMS_PER_BLOCK + (threadIdx.x >> 2);
  const bool live = xi < count;
  const int x = live ? xi : count - 1;
  u32* slot = lds + (threadIdx.x >> 2) * SLOT_WORDS;
  u32 n[LPL], b[LPL], acc[LPL];
  load_lane_limbs(n, cs->n, ln);
  load_be256(b, base_be + (size_t)x * 256, ln);
  to_mont(b, slot, cs, n, ln);
  u32* my = tab + (size_t)x * 16 * L;
  load_lane_limbs(acc, cs->one_m, ln);
  if (live) store_lane_limbs(my, acc, ln);
  if (live) store_lane_limbs(my + L, b, ln);
  slot_store(slot, b, ln);
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int k = 0; k < LPL; ++k) acc[k] = b[k];
  if (odd_only) {
    // a sliding window over an exponent that every share has in common only ever asks for the odd powers:
    // b^2 once, then b^3, b^5, .. b^15 -- 1 squaring + 7 products instead of 14 products (even entries stay unwritten)
    u32 b2[LPL];
    mont_sqr<MODP_N0INV_C>(b2, b, slot, n, ln);
    __builtin_amdgcn_wave_barrier();
    slot_store(slot, b2, ln);
    __builtin_amdgcn_wave_barrier();
    for (int e = 3; e < 16; e += 2) {
      mont_mul<MODP_N0INV_C>(acc, acc, slot, n, ln);
      if (live) store_lane_limbs(my + (size_t)e * L, acc, ln);
    }
    return;
  }
  for (int e = 2; e < 16; ++e) {
    mont_mul<MODP_N0INV_C>(acc, acc, slot, n, ln);
    if (live) store_lane_limbs(my + (size_t)e * L, acc, ln);
  }
}

// ---------------------------------------------------------------------------------------
// out[x] = B1[x]^e1[x] * B2[x]^e2[x]  mod q    (dleq.rs:66-84: a = g^r * h^c)
// Straus interleaving with 4-bit fixed windows over precomputed tables.
//   tab1, tab2     : window tables; *_stride = 16*76 words per number, or 0 for a shared base
//   e1_be          : [count][256] exponents (2048 bit)
//   e2_be          : [count][256] or, when e2_stride == 0, one shared [256] exponent
//   e2_windows     : number of low 4-bit windows of e2 that may be non-zero (64 for a 256-bit
//                    challenge, 512 for a full-width exponent)
// ---------------------------------------------------------------------------------------
extern "C" __global__ void __launch_bounds__(BLOCK_THREADS) WAVES_ATTR
k_modp_dual_exp(const u32* __restrict__ tab1, size_t tab1_stride, const u32* __restrict__ tab2, size_t tab2_stride,
                const uint8_t* __restrict__ e1_be, const uint8_t* __restrict__ e2_be, size_t e2_stride,
                int e2_windows, int count, uint8_t* __restrict__ out_be, const ModpConsts* __restrict__ cs) {
  __shared__ __attribute__((aligned(16))) u32 lds[NUMS_PER_BLOCK * SLOT_WORDS];
  const Lane ln = make_lane();
  const int xi = blockIdx.x * NUMS_PER_BLOCK + (threadIdx.x >> 2);
  const bool live = xi < count;
  const int x = live ? xi : count - 1;
  u32* slot = lds + (threadIdx.x >> 2) * SLOT_WORDS;
  u32 n[LPL], acc[LPL];
  load_lane_limbs(n, cs->n, ln);
  const u32* t1 = tab1 + (size_t)x * tab1_stride;
  const u32* t2 = tab2 + (size_t)x * tab2_stride;
  const uint8_t* e1 = e1_be + (size_t)x * 256;
  const uint8_t* e2 = e2_be + (size_t)x * e2_stride;
  const int first_e2 = 512 - e2_windows;

  // Program (one Montgomery-product site): window w = 0..511 of the exponents, most significant
  // first: 4 x SQUARE (w > 0), MUL by tab1[d1], MUL by tab2[d2] (only the low e2_windows windows),
  // then FINAL (by plain 1).  The first window loads tab1[d1] instead of multiplying into one.
  {
    const u32 byte1 = e1[0];
    load_lane_limbs(acc, t1 + (size_t)(byte1 >> 4) * L, ln);
  }
  int w = 0, s = (first_e2 == 0) ? 5 : 6;     // step inside the window: 0..3 square, 4 tab1, 5 tab2, 6 next
  while (true) {
    if (s == 6) { ++w; s = 0; }
    const bool final_step = (w == 512);
    const bool sq = !final_step && s < 4;
    if (final_step) {
      slot_fill_from_global(slot, cs->one, ln);
    } else if (s < 4) {
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
    if (final_step) break;
    ++s;
    if (s == 5 && w < first_e2) s = 6;
  }
  store_canonical_be256(out_be + (size_t)x * 256, acc, false, slot, cs, n, ln, live);
}

// ---------------------------------------------------------------------------------------
// Fixed-base comb table of one base g:  comb[k][d] = g^(d * 16^k)  (Montgomery form), k = 0..511,
// d = 0..15  (2.5 MB, built once per context and base, L2/MALL resident afterwards).
//   step 1 (one quad, sequential): comb[k][1] = comb[k-1][1]^16
//   step 2 (one number per k):     comb[k][d] = comb[k][d-1] * comb[k][1]
// ---------------------------------------------------------------------------------------
extern "C" __global__ void __launch_bounds__(BLOCK_THREADS) WAVES_ATTR
k_modp_comb_bases(const uint8_t* __restrict__ base_be, u32* __restrict__ comb, const ModpConsts* __restrict__ cs) {
  __shared__ __attribute__((aligned(16))) u32 lds[NUMS_PER_BLOCK * SLOT_WORDS];
  const Lane ln = make_lane();
  u32* slot = lds + (threadIdx.x >> 2) * SLOT_WORDS;
  const bool writer = (blockIdx.x == 0) && (threadIdx.x < 4);
  u32 n[LPL], acc[LPL];
  load_lane_limbs(n, cs->n, ln);
  load_be256(acc, base_be, ln);
  // op 0: to Montgomery form (by R^2); then 4 squarings per k
  for (int op = 0; op <= 511 * 4; ++op) {
    if (op == 0) slot_fill_from_global(slot, cs->r2, ln); else slot_store(slot, acc, ln);
    __builtin_amdgcn_wave_barrier();
    if (op == 0) mont_mul<MODP_N0INV_C>(acc, acc, slot, n, ln); else mont_sqr<MODP_N0INV_C>(acc, acc, slot, n, ln);
    __builtin_amdgcn_wave_barrier();
    if (writer && (op % 4) == 0) store_lane_limbs(comb + ((size_t)(op / 4) * 16 + 1) * L, acc, ln);
  }
}

extern "C" __global__ void __launch_bounds__(BLOCK_THREADS) WAVES_ATTR
k_modp_comb_rows(u32* __restrict__ comb, const ModpConsts* __restrict__ cs) {
  __shared__ __attribute__((aligned(16))) u32 lds[NUMS_PER_BLOCK * SLOT_WORDS];
  const Lane ln = make_lane();
  const int xi = blockIdx.x * NUMS_PER_BLOCK + (threadIdx.x >> 2);
  const bool live = xi < 512;
  const int k = live ? xi : 511;
  u32* slot = lds + (threadIdx.x >> 2) * SLOT_WORDS;
  u32 n[LPL], acc[LPL];
  load_lane_limbs(n, cs->n, ln);
  u32* row = comb + (size_t)k * 16 * L;
  load_lane_limbs(acc, cs->one_m, ln);
  if (live) store_lane_limbs(row, acc, ln);
  load_lane_limbs(acc, row + L, ln);
  slot_store(slot, acc, ln);
  __builtin_amdgcn_wave_barrier();
  for (int d = 2; d < 16; ++d) {
    mont_mul<MODP_N0INV_C>(acc, acc, slot, n, ln);
    if (live) store_lane_limbs(row + (size_t)d * L, acc, ln);
  }
}

// Wide comb (16-bit windows): comb16[k][d] = g^(d * 2^(16 k)), k < 128, d < 65536 -- 2.5 GB, sized for 288 GB of HBM.
// Row bases come from the 4-bit comb (comb4[4k][1] = g^(2^(16k))); pass j = 1..15 doubles every row:
//   p = entry[2^(j-1)]^2 = entry[2^j],   entry[2^j + i] = entry[i] * p   for i < 2^j.
extern "C" __global__ void k_modp_comb16_init(const u32* __restrict__ comb4, u32* __restrict__ comb16,
                                              const ModpConsts* __restrict__ cs) {
  const int k = blockIdx.x;                    // 128 blocks, one thread per limb
  const int j = threadIdx.x;
  if (j >= L) return;
  comb16[((size_t)k * 65536 + 0) * L + j] = cs->one_m[j];
  comb16[((size_t)k * 65536 + 1) * L + j] = comb4[((size_t)(4 * k) * 16 + 1) * L + j];
}

extern "C" __global__ void __launch_bounds__(BLOCK_THREADS) WAVES_ATTR
k_modp_comb16_pass(u32* __restrict__ comb16, int j, const ModpConsts* __restrict__ cs) {
  __shared__ __attribute__((aligned(16))) u32 lds[NUMS_PER_BLOCK * SLOT_WORDS];
  const Lane ln = make_lane();
  const int half = 1 << j;
  const int items = 128 * half;
  const int xi = blockIdx.x * NUMS_PER_BLOCK + (threadIdx.x >> 2);
  const bool live = xi < items;
  const int x = live ? xi : items - 1;
  const int k = x >> j, i = x & (half - 1);
  u32* slot = lds + (threadIdx.x >> 2) * SLOT_WORDS;
  u32* row = comb16 + (size_t)k * 65536 * L;
  u32 n[LPL], p[LPL], a[LPL];
  load_lane_limbs(n, cs->n, ln);
  load_lane_limbs(p, row + (size_t)(half >> 1) * L, ln);
  slot_store(slot, p, ln);
  __builtin_amdgcn_wave_barrier();
  mont_mul<MODP_N0INV_C>(p, p, slot, n, ln);           // p = entry[2^j]
  __builtin_amdgcn_wave_barrier();
  slot_store(slot, p, ln);
  load_lane_limbs(a, row + (size_t)i * L, ln);
  __builtin_amdgcn_wave_barrier();
  mont_mul<MODP_N0INV_C>(a, a, slot, n, ln);
  if (live) store_lane_limbs(row + (size_t)(half + i) * L, a, ln);
}

// ---------------------------------------------------------------------------------------
// out[x] = g^e1[x] * B2[x]^e2[x] with g given as a comb table (no squarings for the g part):
//   phase A: acc = B2^e2 by 4-bit windows over tab2 (only the low e2_windows windows)
//   phase B: acc = prod_k comb[k][digit_k(e1)], then times the saved phase-A value
// This is a1 = g^r * X^c (dleq.rs:75-77) with g the subgroup generator, and G^r * pk^c of verify_share.
// ---------------------------------------------------------------------------------------
extern "C" __global__ void __launch_bounds__(BLOCK_THREADS) WAVES_ATTR
k_modp_comb_dual_exp(const u32* __restrict__ comb, const u32* __restrict__ tab2, size_t tab2_stride,
                     const uint8_t* __restrict__ e1_be, const uint8_t* __restrict__ e2_be, size_t e2_stride,
                     int e2_windows, int count, uint8_t* __restrict__ out_be, int mode, u32* __restrict__ p_m,
                     int comb_bits, const ModpConsts* __restrict__ cs, const uint16_t* __restrict__ c_sched) {
  // c_sched: sliding-window schedule of a shared e2 (see k_modp_dual_exp_w6); tab2 then needs its odd entries only
  // comb_bits: 4 = comb[k][d] = g^(d 16^k), 512 rows of 16; 16 = comb[k][d] = g^(d 65536^k), 128 rows of 65536
  // (2.5 GB in HBM, a quarter of the products).
  // mode 0: the whole product.  mode 1: only g^e1 (needs nothing but the exponent, so it can run before B2 is
  // known), left in Montgomery form in p_m.  mode 2: B2^e2 times the stored p_m.
  __shared__ __attribute__((aligned(16))) u32 lds[2 * NUMS_PER_BLOCK * SLOT_WORDS];
  const Lane ln = make_lane();
  const int xi = blockIdx.x * NUMS_PER_BLOCK + (threadIdx.x >> 2);
  const bool live = xi < count;
  const int x = live ? xi : count - 1;
  u32* slot = lds + (threadIdx.x >> 2) * SLOT_WORDS;
  u32* pslot = lds + (NUMS_PER_BLOCK + (threadIdx.x >> 2)) * SLOT_WORDS;
  u32 n[LPL], acc[LPL];
  load_lane_limbs(n, cs->n, ln);
  const u32* t2 = tab2 + (size_t)x * tab2_stride;
  const uint8_t* e1 = e1_be + (size_t)x * 256;
  const uint8_t* e2 = e2_be + (size_t)x * e2_stride;
  const int first_e2 = 512 - e2_windows;
  const int comb_rows = comb_bits == 16 ? 128 : 512;

  // steps: A(w, s): w = first_e2..511, s = 0..3 square (w > first_e2), 4 = table product
  //        B(k):    k = 0..511 comb product (k = 0 loads), then P = product with phase-A value, F = final
  enum { PH_A, PH_B, PH_P, PH_F };
  int phase = (e2_windows > 0 && mode != 1) ? PH_A : PH_B;
  int w = first_e2, s = 4, k = 0;
  int cur = 0, si = 1, a_stage = 0;                      // schedule mode: weight of the accumulator's unit, next window
  const int sn = c_sched ? (int)c_sched[0] : 0;
  if (phase == PH_A) {
    if (c_sched != nullptr) {
      load_lane_limbs(acc, t2 + (size_t)c_sched[2] * L, ln);       // top window: load instead of multiply
      cur = (int)c_sched[1];
    } else {
      const u32 byte = e2[w >> 1];
      const u32 d = (w & 1) ? (byte & 15) : (byte >> 4);
      load_lane_limbs(acc, t2 + (size_t)d * L, ln);       // first window: load instead of multiply
      s = 5;
    }
  }
  while (true) {
    const u32* bptr = slot;
    const u32* fill = nullptr;     // global operand staged into `fill_to` (one staging site keeps the register use low)
    u32* fill_to = slot;
    bool skip = false, sq = false;
    if (phase == PH_A) {
      bool a_done = false;
      if (c_sched != nullptr) {
        if (a_stage == 0) {                              // square down one bit, or finish at weight 0
          if (cur == 0) a_done = true;
          else { slot_store(slot, acc, ln); sq = true; --cur; a_stage = 1; }
        } else {                                         // a window that ends at this bit?
          a_stage = 0;
          if (si < sn && cur == (int)c_sched[1 + 2 * si]) { fill = t2 + (size_t)c_sched[2 + 2 * si] * L; ++si; }
          else skip = true;
        }
      } else {
        if (s == 5) { ++w; s = 0; }
        if (w == 512) {
          a_done = true;
        } else {
          if (s < 4) {
            slot_store(slot, acc, ln);
            sq = true;
          } else {
            const u32 byte = e2[w >> 1];
            const u32 d = (w & 1) ? (byte & 15) : (byte >> 4);
            fill = t2 + (size_t)d * L;
          }
          ++s;
        }
      }
      if (a_done) {
        if (mode == 2) {                                 // multiply by the stored g^e1
          fill = p_m + (size_t)x * L;
          fill_to = pslot;
          bptr = pslot;
          phase = PH_F;
        } else {
          slot_store(pslot, acc, ln);                    // save B2^e2
          phase = PH_B; k = 0;
          continue;
        }
      }
    } else if (phase == PH_B) {
      size_t ent;
      if (comb_bits == 16) {
        ent = (size_t)k * 65536 + (((u32)e1[254 - 2 * k] << 8) | e1[255 - 2 * k]);
      } else {
        const u32 byte = e1[255 - (k >> 1)];
        ent = (size_t)k * 16 + ((k & 1) ? (byte >> 4) : (byte & 15));
      }
      const u32* entry = comb + ent * L;
      if (k == 0) { load_lane_limbs(acc, entry, ln); skip = true; }
      else fill = entry;
      ++k;
      if (k == comb_rows) phase = (mode == 1) ? PH_F + 2 : (e2_windows > 0) ? PH_P : PH_F;
    } else if (phase == PH_P) {
      bptr = pslot;
      phase = PH_F;
    } else {
      fill = cs->one;
      phase = PH_F + 1;
    }
    if (fill != nullptr) slot_fill_from_global(fill_to, fill, ln);
    if (!skip) {
      __builtin_amdgcn_wave_barrier();
      if (sq) mont_sqr<MODP_N0INV_C>(acc, acc, slot, n, ln); else mont_mul<MODP_N0INV_C>(acc, acc, bptr, n, ln);
      __builtin_amdgcn_wave_barrier();
    }
    if (phase >= PH_F + 1) break;
  }
  if (mode == 1) {
    if (live) store_lane_limbs(p_m + (size_t)x * L, acc, ln);
    return;
  }
  store_canonical_be256(out_be + (size_t)x * 256, acc, false, slot, cs, n, ln, live);
}

// ---------------------------------------------------------------------------------------
// a2 = y^r * Y^c of the distribution check with a 6-bit window table for y (64 entries, 62 products to build,
// 341 products instead of 511 for the 2047-bit r) and the 4-bit table for Y (c < 2^256): 2 046 squarings +
// 341 + 64 products.  Windows of r are aligned at multiples of 6 bits, windows of c at multiples of 4.
// ---------------------------------------------------------------------------------------
extern "C" __global__ void __launch_bounds__(BLOCK_THREADS) WAVES_ATTR
k_modp_build_table64(const uint8_t* __restrict__ base_be, int count, u32* __restrict__ tab,
                     const ModpConsts* __restrict__ cs) {
  __shared__ __attribute__((aligned(16))) u32 lds[NUMS_PER_BLOCK * SLOT_WORDS];
  const Lane ln = make_lane();
  const int xi = blockIdx.x * NUMS_PER_BLOCK + (threadIdx.x >> 2);
  const bool live = xi < count;
  const int x = live ? xi : count - 1;
  u32* slot = lds + (threadIdx.x >> 2) * SLOT_WORDS;
  u32 n[LPL], b[LPL], acc[LPL];
  load_lane_limbs(n, cs->n, ln);
  load_be256(b, base_be + (size_t)x * 256, ln);
  to_mont(b, slot, cs, n, ln);
  u32* my = tab + (size_t)x * 64 * L;
  load_lane_limbs(acc, cs->one_m, ln);
  if (live) store_lane_limbs(my, acc, ln);
  if (live) store_lane_limbs(my + L, b, ln);
  slot_store(slot, b, ln);
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int k = 0; k < LPL; ++k) acc[k] = b[k];
  for (int e = 2; e < 64; ++e) {
    mont_mul<MODP_N0INV_C>(acc, acc, slot, n, ln);
    if (live) store_lane_limbs(my + (size_t)e * L, acc, ln);
  }
}

extern "C" __global__ void __launch_bounds__(BLOCK_THREADS) WAVES_ATTR
k_modp_dual_exp_w6(const u32* __restrict__ tab1, const u32* __restrict__ tab2, const uint8_t* __restrict__ e1_be,
                   const uint8_t* __restrict__ c_all, size_t c_stride, int count, uint8_t* __restrict__ out_be,
                   const ModpConsts* __restrict__ cs, const uint16_t* __restrict__ c_sched) {
  // c_sched (one challenge for all shares): a sliding-window schedule made by the host -- [0] = number of windows, then
  // (bit position of the window's lowest bit, odd digit) in descending position -- instead of 64 fixed 4-bit windows:
  // about 51 products for Y^c and a table of the odd powers only (dleq.rs:79-81 has the same c for every share of a box)
  __shared__ __attribute__((aligned(16))) u32 lds[NUMS_PER_BLOCK * SLOT_WORDS];
  const Lane ln = make_lane();
  const int xi = blockIdx.x * NUMS_PER_BLOCK + (threadIdx.x >> 2);
  const bool live = xi < count;
  const int x = live ? xi : count - 1;
  u32* slot = lds + (threadIdx.x >> 2) * SLOT_WORDS;
  u32 n[LPL], acc[LPL];
  load_lane_limbs(n, cs->n, ln);
  const u32* t1 = tab1 + (size_t)x * 64 * L;
  const u32* t2 = tab2 + (size_t)x * 16 * L;
  const uint8_t* e1 = e1_be + (size_t)x * 256;
  const uint8_t* c_be = c_all + (size_t)x * c_stride;      // stride 0: one challenge for all shares; c_all null: y^r only
  // 6-bit window w of the 2048-bit exponent: bits 6w .. 6w+5 (little-endian byte k is e1[255 - k])
  auto digit6 = [&](int w) -> u32 {
    const int o = 6 * w, k = o >> 3;
    const u32 lo = e1[255 - k];
    const u32 hi = (k + 1 < 256) ? e1[254 - k] : 0u;
    return ((lo | (hi << 8)) >> (o & 7)) & 63u;
  };
  // cur = weight (bit index) of the accumulator's unit.  Start with the top window (bits 2046, 2047).
  load_lane_limbs(acc, t1 + (size_t)digit6(341) * L, ln);
  int cur = 2046;
  int s = 0;                       // 0: square (cur decreases), 1: product with tab1, 2: product with tab2, 3: final
  int si = 0;
  const int sn = c_sched ? (int)c_sched[0] : 0;
  while (true) {
    const u32* fill = nullptr;
    bool skip = false;
    if (s == 0) {
      slot_store(slot, acc, ln);
      --cur;
    } else if (s == 1) {
      if (cur % 6 == 0) fill = t1 + (size_t)digit6(cur / 6) * L; else skip = true;
    } else if (s == 2) {
      if (c_sched != nullptr) {
        if (si < sn && cur == (int)c_sched[1 + 2 * si]) {
          fill = t2 + (size_t)c_sched[2 + 2 * si] * L;
          ++si;
        } else {
          skip = true;
        }
      } else if ((cur & 3) == 0 && cur < 256 && c_all != nullptr) {
        const u32 byte = c_be[255 - (cur >> 3)];
        fill = t2 + (size_t)((cur & 4) ? (byte >> 4) : (byte & 15)) * L;
      } else {
        skip = true;
      }
    } else {
      fill = cs->one;              // leave the Montgomery domain
    }
    if (!skip) {
      if (fill != nullptr) slot_fill_from_global(slot, fill, ln);
      __builtin_amdgcn_wave_barrier();
      if (s == 0) mont_sqr<MODP_N0INV_C>(acc, acc, slot, n, ln); else mont_mul<MODP_N0INV_C>(acc, acc, slot, n, ln);
      __builtin_amdgcn_wave_barrier();
    }
    if (s == 3) break;
    s = (s == 2) ? (cur == 0 ? 3 : 0) : s + 1;
  }
  store_canonical_be256(out_be + (size_t)x * 256, acc, false, slot, cs, n, ln, live);
}

// ---------------------------------------------------------------------------------------
// One base, two exponents, two results (the dealer: Y_i = y_i^P(i), participant.rs:219, and a2_i = y_i^w_i,
// dleq.rs:213-216).  Left-to-right windows cannot share the squarings of two exponentiations; right-to-left can:
//   cur_k = y^(2^(5k)) is computed once (2 045 squarings), and for each exponent e = sum_k d_k 2^(5k)
//   y^e = prod_{d=1..31} B[d]^d  with the buckets  B[d] = prod_{k : d_k = d} cur_k        (Yao 1976 / BGMW)
// 410 bucket products per exponent, then at most 60 products for prod B[d]^d (running products, k_modp_bucket_combine).
// The 2 x 31 buckets of a share live in HBM (17.9 KB per share; a quad reads and writes ITS 288 bytes, every bucket
// is private to its quad), a 32-bit occupancy mask per exponent stays in a register: a bucket's first factor is stored,
// not multiplied.  ~2 500 product equivalents for both results instead of ~3 870 for two 6-bit-window ladders.
// ---------------------------------------------------------------------------------------
constexpr int BK_W = MODP_BUCKET_W;            // window bits (5; modp_kernels.h -- the pair-layout bucket kernel uses the same)
constexpr int BK_ENT = (1 << BK_W) - 1;       // buckets per exponent (digit 0 has none)
constexpr int BK_WINDOWS = (2048 + BK_W - 1) / BK_W;

// The bucket a product needs is fetched while the wave is still squaring: an LDS-DMA (global_load_lds_dwordx4, no
// registers) of the 16 numbers' buckets into a second operand slot, issued before the five squarings for the first
// exponent's bucket and before the first bucket product for the second one (whose slot the squarings were using).
// One DMA instruction moves 64 x 16 bytes to consecutive LDS addresses; a number's 288-byte slot is 18 such pieces, so
// lane l of instruction i fetches piece (l + 64 i) % 18 of number (l + 64 i) / 18 -- whose digit it gets by a shuffle.
__device__ __forceinline__ void bucket_prefetch(u32* wave_slots, const u32* __restrict__ buckets, int first_x, int count,
                                                int e, u32 d_eff) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int c = lane + 64 * i;
    const int jn = (c < NUMS_PER_WAVE * 18) ? c / 18 : NUMS_PER_WAVE - 1;
    const int part = c - 18 * jn;
    const u32 dj = (u32)__shfl((int)d_eff, jn * 4);
    const int xj = (first_x + jn < count) ? first_x + jn : count - 1;
    const u32* src = buckets + (((size_t)xj * 2 + (size_t)e) * BK_ENT + (dj - 1)) * L + part * 4;
    if (c < NUMS_PER_WAVE * 18)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(uintptr_t)src,
                                       (__attribute__((address_space(3))) void*)(uintptr_t)(wave_slots + i * 256), 16, 0, 0);
  }
}

extern "C" __global__ void __launch_bounds__(BLOCK_THREADS) WAVES_ATTR
k_modp_twin_exp_buckets(const uint8_t* __restrict__ base_be, const uint8_t* __restrict__ e1_be,
                        const uint8_t* __restrict__ e2_be, int count, u32* __restrict__ buckets,
                        u32* __restrict__ occupancy, const ModpConsts* __restrict__ cs) {
  __shared__ __attribute__((aligned(16))) u32 lds[2 * NUMS_PER_BLOCK * SLOT_WORDS];
  const Lane ln = make_lane();
  const int wave = threadIdx.x >> 6;
  const int first_x = blockIdx.x * NUMS_PER_BLOCK + wave * NUMS_PER_WAVE;
  const int xi = first_x + ((threadIdx.x & 63) >> 2);
  const bool live = xi < count;
  const int x = live ? xi : count - 1;
  u32* wslot1 = lds + wave * 2 * NUMS_PER_WAVE * SLOT_WORDS;     // squarings (a copy of cur), then the second exponent's bucket
  u32* wslot2 = wslot1 + NUMS_PER_WAVE * SLOT_WORDS;             // the first exponent's bucket
  u32* slot1 = wslot1 + ((threadIdx.x & 63) >> 2) * SLOT_WORDS;
  u32* slot2 = wslot2 + ((threadIdx.x & 63) >> 2) * SLOT_WORDS;
  u32 n[LPL], cur[LPL], acc[LPL];
  load_lane_limbs(n, cs->n, ln);
  load_be256(cur, base_be + (size_t)x * 256, ln);
  u32* mine = buckets + (size_t)x * 2 * BK_ENT * L;
  const uint8_t* ex0 = e1_be + (size_t)x * 256;
  const uint8_t* ex1 = e2_be + (size_t)x * 256;
  auto digit = [&](const uint8_t* ex, int k) -> u32 {
    const int o = BK_W * k, b = o >> 3;
    const u32 lo = ex[255 - b];
    const u32 hi = (b + 1 < 256) ? ex[254 - b] : 0u;
    return ((lo | (hi << 8)) >> (o & 7)) & (u32)BK_ENT;
  };
  u32 occ0 = 0, occ1 = 0;
  // base to Montgomery form
  slot_fill_from_global(slot1, cs->r2, ln);
  __builtin_amdgcn_wave_barrier();
  mont_mul<MODP_N0INV_C>(cur, cur, slot1, n, ln);
  __builtin_amdgcn_wave_barrier();
  u32 d0 = digit(ex0, 0), d1 = digit(ex1, 0);
  // per window k: op 1 = bucket of the first exponent, op 2 = of the second, ops 3..7 = five squarings.
  // bit 0 of an occupancy mask is never set: digit 0 touches nothing.
  int k = 0, op = 1;
  while (true) {
    if (op <= 2) {
      const u32 d = (op == 1) ? d0 : d1;
      const u32 occ = (op == 1) ? occ0 : occ1;
      const bool has = (occ >> d) & 1u;
      u32* bk = mine + ((size_t)(op - 1) * BK_ENT + (d ? d - 1 : 0)) * L;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this op's bucket has landed (and earlier stores are done)
      if (op == 1) {
        // the squarings are over: their slot takes the second exponent's bucket while the first product runs
        const bool has1 = (occ1 >> d1) & 1u;
        bucket_prefetch(wslot1, buckets, first_x, count, 1, (d1 != 0 && has1) ? d1 : 1u);
      }
      __builtin_amdgcn_wave_barrier();
      // the bucket is the LDS operand, cur stays in registers; quads with nothing to multiply compute and drop a product
      mont_mul<MODP_N0INV_C>(acc, cur, (op == 1) ? slot2 : slot1, n, ln);
      __builtin_amdgcn_wave_barrier();
      if (d != 0 && live) {
        if (has) store_lane_limbs(bk, acc, ln); else store_lane_limbs(bk, cur, ln);
      }
      const u32 bit = d ? (1u << d) : 0u;
      if (op == 1) occ0 |= bit; else occ1 |= bit;
      if (op == 2) {
        if (k == BK_WINDOWS - 1) break;
        d0 = digit(ex0, k + 1);
        d1 = digit(ex1, k + 1);
        const bool has0 = (occ0 >> d0) & 1u;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the bucket stores above are done before the next fetch
        bucket_prefetch(wslot2, buckets, first_x, count, 0, (d0 != 0 && has0) ? d0 : 1u);
      }
    } else {
      slot_store(slot1, cur, ln);
      __builtin_amdgcn_wave_barrier();
      mont_sqr<MODP_N0INV_C>(cur, cur, slot1, n, ln);
      __builtin_amdgcn_wave_barrier();
    }
    if (op == 2 + BK_W) { op = 1; ++k; } else ++op;
  }
  if (live && ln.q == 0) {
    occupancy[(size_t)x * 2] = occ0;
    occupancy[(size_t)x * 2 + 1] = occ1;
  }
}

// y^e = prod_d B[d]^d: run = B[31]; res = run; then for d = 30..1: run *= B[d]; res *= run.  One quad per (share, exponent);
// empty buckets and a not-yet-started run / res skip their product (the result for e = 0 is 1).
extern "C" __global__ void __launch_bounds__(BLOCK_THREADS) WAVES_ATTR
k_modp_bucket_combine(const u32* __restrict__ buckets, const u32* __restrict__ occupancy, int count,
                      uint8_t* __restrict__ out1_be, uint8_t* __restrict__ out2_be, const ModpConsts* __restrict__ cs) {
  __shared__ __attribute__((aligned(16))) u32 lds[NUMS_PER_BLOCK * SLOT_WORDS];
  const Lane ln = make_lane();
  const int xi = blockIdx.x * NUMS_PER_BLOCK + (threadIdx.x >> 2);
  const bool live = xi < 2 * count;
  const int x = live ? xi : 2 * count - 1;      // (share, exponent) pair: share x >> 1, exponent x & 1
  u32* slot = lds + (threadIdx.x >> 2) * SLOT_WORDS;
  u32 n[LPL], run[LPL], res[LPL], tmp[LPL];
  load_lane_limbs(n, cs->n, ln);
  load_lane_limbs(run, cs->one_m, ln);
  load_lane_limbs(res, cs->one_m, ln);
  const u32 occ = occupancy[x];
  const u32* mine = buckets + (size_t)x * BK_ENT * L;
  bool run_set = false, res_set = false;
  // two ops per digit: op 0 folds B[d] into run, op 1 folds run into res
  for (int step = 0; step < 2 * BK_ENT; ++step) {
    const int d = BK_ENT - (step >> 1);
    const bool has = (occ >> d) & 1u;
    if ((step & 1) == 0) {
      if (has) slot_fill_from_global(slot, mine + (size_t)(d - 1) * L, ln);
      __builtin_amdgcn_wave_barrier();
      mont_mul<MODP_N0INV_C>(tmp, run, slot, n, ln);
      if (has) {
        if (run_set) {
#pragma unroll
          for (int j = 0; j < LPL; ++j) run[j] = tmp[j];
        } else {
          slot_load(run, slot, ln);
        }
        run_set = true;
      }
      __builtin_amdgcn_wave_barrier();
    } else {
      slot_store(slot, run, ln);
      __builtin_amdgcn_wave_barrier();
      mont_mul<MODP_N0INV_C>(tmp, res, slot, n, ln);
      __builtin_amdgcn_wave_barrier();
      if (run_set) {
        if (res_set) {
#pragma unroll
          for (int j = 0; j < LPL; ++j) res[j] = tmp[j];
        } else {
#pragma unroll
          for (int j = 0; j < LPL; ++j) res[j] = run[j];
        }
        res_set = true;
      }
    }
  }
  uint8_t* out = ((x & 1) ? out2_be : out1_be) + (size_t)(x >> 1) * 256;
  store_canonical_be256(out, res, true, slot, cs, n, ln, live);
}

// ---------------------------------------------------------------------------------------
// Registered public keys (opt-in): per-key tables for y^r, built once per key set and reused by every box that is
// verified against it.  ks[key][j][d] = y^(d * 2^(256 j)), j < 8, d < 128 (Montgomery form; 8 x 128 x 288 B =
// 295 KB per key, 19.3 GB for 65536 keys; round 4 had 256 entries per sub-base: 38.7 GB).  With r = sum_j r_j 2^(256 j):
//   y^r * Y^c = prod_j (y^(2^(256 j)))^(r_j) * Y^c    -- Straus over 256-bit rows:
// 252 squarings shared by all nine bases, 37 x 8 = 296 products from the key table (7-bit windows; signed digits would
// halve the table in a curve group, here an inverse is not free), 64 from Y's 4-bit table: 613 products instead of 2 620.
// ---------------------------------------------------------------------------------------
constexpr int KS_SUB = 8;          // sub-bases per key
constexpr int KS_WIN = 7;          // window width of the 256-bit rows of r
constexpr int KS_ENT = 1 << KS_WIN;                  // entries per sub-base
constexpr int KS_NWIN = (256 + KS_WIN - 1) / KS_WIN; // windows per row (37; the top one holds 4 bits)
// window w of row j of the 256-byte big-endian r: bits [256 j + 7 w, 256 j + 7 w + 7) of r, without the bits of row j + 1
__device__ __forceinline__ u32 ks_digit(const uint8_t* __restrict__ r, int j, int w) {
  const int g = 256 * j + KS_WIN * w, b = g >> 3;
  const u32 lo = r[255 - b];
  const u32 hi = (b + 1 < 256) ? r[254 - b] : 0u;
  const int top = 256 - KS_WIN * w;                  // bits of this window that belong to the row
  return ((lo | (hi << 8)) >> (g & 7)) & (u32)((1 << (top < KS_WIN ? top : KS_WIN)) - 1);
}

// sub-bases: ks[key][j][1] = y^(2^(256 j)), ks[key][j][0] = 1   (one quad per key, 7 x 256 squarings)
extern "C" __global__ void __launch_bounds__(BLOCK_THREADS) WAVES_ATTR
k_modp_keyset_bases(const uint8_t* __restrict__ pk_be, int count, u32* __restrict__ ks,
                    const ModpConsts* __restrict__ cs) {
  __shared__ __attribute__((aligned(16))) u32 lds[NUMS_PER_BLOCK * SLOT_WORDS];
  const Lane ln = make_lane();
  const int xi = blockIdx.x * NUMS_PER_BLOCK + (threadIdx.x >> 2);
  const bool live = xi < count;
  const int x = live ? xi : count - 1;
  u32* slot = lds + (threadIdx.x >> 2) * SLOT_WORDS;
  u32 n[LPL], acc[LPL], one[LPL];
  load_lane_limbs(n, cs->n, ln);
  load_lane_limbs(one, cs->one_m, ln);
  load_be256(acc, pk_be + (size_t)x * 256, ln);
  u32* mine = ks + (size_t)x * KS_SUB * KS_ENT * L;
  // op 0: to Montgomery form; then 256 squarings per further sub-base
  for (int op = 0; op <= (KS_SUB - 1) * 256; ++op) {
    if (op == 0) slot_fill_from_global(slot, cs->r2, ln); else slot_store(slot, acc, ln);
    __builtin_amdgcn_wave_barrier();
    if (op == 0) mont_mul<MODP_N0INV_C>(acc, acc, slot, n, ln); else mont_sqr<MODP_N0INV_C>(acc, acc, slot, n, ln);
    __builtin_amdgcn_wave_barrier();
    if (live && (op % 256) == 0) {
      u32* row = mine + (size_t)(op / 256) * KS_ENT * L;
      store_lane_limbs(row, one, ln);
      store_lane_limbs(row + L, acc, ln);
    }
  }
}

// rows: ks[key][j][d] = ks[key][j][d-1] * ks[key][j][1], d = 2..255   (one quad per (key, j))
extern "C" __global__ void __launch_bounds__(BLOCK_THREADS) WAVES_ATTR
k_modp_keyset_rows(u32* __restrict__ ks, int rows, const ModpConsts* __restrict__ cs) {
  __shared__ __attribute__((aligned(16))) u32 lds[NUMS_PER_BLOCK * SLOT_WORDS];
  const Lane ln = make_lane();
  const int xi = blockIdx.x * NUMS_PER_BLOCK + (threadIdx.x >> 2);
  const bool live = xi < rows;
  const int x = live ? xi : rows - 1;
  u32* slot = lds + (threadIdx.x >> 2) * SLOT_WORDS;
  u32* row = ks + (size_t)x * KS_ENT * L;
  u32 n[LPL], acc[LPL];
  load_lane_limbs(n, cs->n, ln);
  load_lane_limbs(acc, row + L, ln);
  slot_store(slot, acc, ln);
  __builtin_amdgcn_wave_barrier();
  for (int d = 2; d < KS_ENT; ++d) {
    mont_mul<MODP_N0INV_C>(acc, acc, slot, n, ln);
    if (live) store_lane_limbs(row + (size_t)d * L, acc, ln);
  }
}

// out[x] = y_x^r_x * Y_x^c with y_x's registered table ks[x] and Y's 4-bit table tab2; c < 2^256 shared.
extern "C" __global__ void __launch_bounds__(BLOCK_THREADS) WAVES_ATTR
k_modp_keyset_dual_exp(const u32* __restrict__ ks, const u32* __restrict__ tab2, const uint8_t* __restrict__ r_be,
                       const uint8_t* __restrict__ c_be, int count, uint8_t* __restrict__ out_be,
                       const ModpConsts* __restrict__ cs) {
  __shared__ __attribute__((aligned(16))) u32 lds[NUMS_PER_BLOCK * SLOT_WORDS];
  const Lane ln = make_lane();
  const int xi = blockIdx.x * NUMS_PER_BLOCK + (threadIdx.x >> 2);
  const bool live = xi < count;
  const int x = live ? xi : count - 1;
  u32* slot = lds + (threadIdx.x >> 2) * SLOT_WORDS;
  u32 n[LPL], acc[LPL];
  load_lane_limbs(n, cs->n, ln);
  const u32* kt = ks + (size_t)x * KS_SUB * KS_ENT * L;
  const u32* t2 = tab2 + (size_t)x * 16 * L;
  const uint8_t* r = r_be + (size_t)x * 256;
  // Program: from bit 252 down; at every bit a squaring (not at the first), then -- bit divisible by 7 -- the eight products with
  // ks[j][window of r_j], then -- bit divisible by 4 -- the product with Y^(nibble of c); at the end the product with plain 1.
  //   s = 0 squaring, 1 .. 8 key table j = s - 1, 9 nibble of c, 10 closing
  load_lane_limbs(acc, cs->one_m, ln);
  int cur = KS_WIN * (KS_NWIN - 1), s = 1;
  while (true) {
    const u32* fill = nullptr;
    bool skip = false;
    if (s == 0) {
      slot_store(slot, acc, ln);                       // squaring
    } else if (s <= KS_SUB) {
      if (cur % KS_WIN == 0) fill = kt + ((size_t)(s - 1) * KS_ENT + ks_digit(r, s - 1, cur / KS_WIN)) * L; else skip = true;
    } else if (s == KS_SUB + 1) {
      if ((cur & 3) == 0) {
        const u32 byte = c_be[255 - (cur >> 3)];
        fill = t2 + (size_t)((cur & 4) ? (byte >> 4) : (byte & 15)) * L;
      } else {
        skip = true;
      }
    } else {
      fill = cs->one;                                  // leave the Montgomery domain
    }
    if (!skip) {
      if (fill != nullptr) slot_fill_from_global(slot, fill, ln);
      __builtin_amdgcn_wave_barrier();
      if (s == 0) mont_sqr<MODP_N0INV_C>(acc, acc, slot, n, ln); else mont_mul<MODP_N0INV_C>(acc, acc, slot, n, ln);
      __builtin_amdgcn_wave_barrier();
    }
    if (s == KS_SUB + 2) break;
    if (s == KS_SUB + 1) {
      if (cur == 0) { s = KS_SUB + 2; continue; }
      --cur;
      s = 0;
    } else {
      ++s;
    }
  }
  store_canonical_be256(out_be + (size_t)x * 256, acc, false, slot, cs, n, ln, live);
}

// ---------------------------------------------------------------------------------------
// host-callable launchers (plain C linkage, used by mpvss_capi.cpp)
// ---------------------------------------------------------------------------------------


extern "C" int modp_consts_upload(void** dev_consts) {
  ModpConsts h;
  for (int j = 0; j < L; ++j) {
    h.n[j] = MODP_N_LIMBS[j];
    h.r2[j] = MODP_R2_LIMBS[j];
    h.one_m[j] = MODP_ONE_M_LIMBS[j];
    h.one[j] = (j == 0) ? 1u : 0u;
  }
  void* d = nullptr;
  hipError_t e = hipMalloc(&d, sizeof(ModpConsts));
  if (e != hipSuccess) return (int)e;
  e = hipMemcpy(d, &h, sizeof(ModpConsts), hipMemcpyHostToDevice);
  if (e != hipSuccess) return (int)e;
  *dev_consts = d;
  return 0;
}

static inline int grid_for(int count);
// the same four rows for q' = (q-1)/2, the modulus of the scalar-ring kernels
extern "C" int modq_consts_upload(void** dev_consts) {
  ModpConsts h;
  for (int j = 0; j < L; ++j) {
    h.n[j] = MODQH_N_LIMBS[j];
    h.r2[j] = MODQH_R2_LIMBS[j];
    h.one_m[j] = MODQH_ONE_M_LIMBS[j];
    h.one[j] = (j == 0) ? 1u : 0u;
  }
  void* d = nullptr;
  hipError_t e = hipMalloc(&d, sizeof(ModpConsts));
  if (e != hipSuccess) return (int)e;
  e = hipMemcpy(d, &h, sizeof(ModpConsts), hipMemcpyHostToDevice);
  if (e != hipSuccess) return (int)e;
  *dev_consts = d;
  return 0;
}
extern "C" int modq_launch_poly_eval(const uint32_t* coef, int t, const int64_t* positions, int count, int par_even, int par_odd,
                                     uint8_t* out_be, const void* cs_q, hipStream_t s) {
  if (count <= 0) return 0;
  hipLaunchKernelGGL(k_modq_poly_eval, dim3(grid_for(count)), dim3(BLOCK_THREADS), 0, s, coef, t, positions, count, par_even, par_odd,
                     out_be, (const ModpConsts*)cs_q);
  return (int)hipGetLastError();
}
extern "C" int modq_launch_responses(const uint8_t* w_be, const uint8_t* alpha_be, const uint8_t* cneg_be, int c_parity, int count,
                                     uint8_t* out_be, const void* cs_q, hipStream_t s) {
  if (count <= 0) return 0;
  hipLaunchKernelGGL(k_modq_responses, dim3(grid_for(count)), dim3(BLOCK_THREADS), 0, s, w_be, alpha_be, cneg_be, c_parity, count,
                     out_be, (const ModpConsts*)cs_q);
  return (int)hipGetLastError();
}

extern "C" int modq_launch_mul(const uint8_t* a_be, const uint8_t* b_be, int count, uint8_t* out_be, const void* cs_q, hipStream_t s) {
  if (count <= 0) return 0;
  hipLaunchKernelGGL(k_modq_mul, dim3(grid_for(count)), dim3(BLOCK_THREADS), 0, s, a_be, b_be, count, out_be, (const ModpConsts*)cs_q);
  return (int)hipGetLastError();
}

// Residency report for tuning: max resident workgroups per CU the runtime computes for each kernel.
extern "C" int modp_occupancy_report(int* out5) {
  const void* ks[5] = {(const void*)k_modp_commit_eval, (const void*)k_modp_dual_exp, (const void*)k_modp_build_table,
                       (const void*)k_modp_to_mont, (const void*)k_modp_mul};
  for (int i = 0; i < 5; ++i) {
    int nb = -1;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, ks[i], BLOCK_THREADS, 0);
    out5[i] = (e == hipSuccess) ? nb * MODP_WPB : -(int)e;
  }
  return 0;
}

static inline int grid_for(int count) { return (count + NUMS_PER_BLOCK - 1) / NUMS_PER_BLOCK; }

extern "C" int modp_launch_mul(const uint8_t* a, const uint8_t* b, uint8_t* out, int count, const void* cs,
                               hipStream_t s) {
  if (count <= 0) return 0;
  hipLaunchKernelGGL(k_modp_mul, dim3(grid_for(count)), dim3(BLOCK_THREADS), 0, s, a, b, out, count, (const ModpConsts*)cs);
  return (int)hipGetLastError();
}

extern "C" int modp_launch_to_mont(const uint8_t* in, uint32_t* out_m, int count, const void* cs, hipStream_t s) {
  if (count <= 0) return 0;
  hipLaunchKernelGGL(k_modp_to_mont, dim3(grid_for(count)), dim3(BLOCK_THREADS), 0, s, in, out_m, count,
                     (const ModpConsts*)cs);
  return (int)hipGetLastError();
}

extern "C" int modp_launch_commit_eval(const uint32_t* cm, int t, const int64_t* positions, int count,
                                       uint32_t* x_m, uint8_t* x_be, const void* cs, hipStream_t s) {
  if (count <= 0) return 0;
  hipLaunchKernelGGL(k_modp_commit_eval, dim3(grid_for(count)), dim3(BLOCK_THREADS), 0, s, cm, cm, 0x7fffffff, t,
                     positions, count, x_be, x_m, (const int*)nullptr, 0, (const ModpConsts*)cs, (size_t)0, (size_t)0, (size_t)0);
  return (int)hipGetLastError();
}

// Horner kernel gated on the device flag (runs only when *gate == want)
extern "C" int modp_launch_commit_eval_gated(const uint32_t* cm_a, const uint32_t* cm_b, int split, int t,
                                             const int64_t* positions, int count, uint32_t* x_m, uint8_t* x_be,
                                             const int* gate, int want, const void* cs, hipStream_t s) {
  if (count <= 0) return 0;
  hipLaunchKernelGGL(k_modp_commit_eval, dim3(grid_for(count)), dim3(BLOCK_THREADS), 0, s, cm_a, cm_b, split, t,
                     positions, count, x_be, x_m, gate, want, (const ModpConsts*)cs, (size_t)0, (size_t)0, (size_t)0);
  return (int)hipGetLastError();
}
// the same for `boxes` same-shaped boxes in one launch: box b reads cm + b * t rows, positions + b * box_positions, and
// writes its `count` results from row b * box_out of x_m / x_be
extern "C" int modp_launch_commit_eval_boxes(const uint32_t* cm, int t, const int64_t* positions, size_t box_positions, int count,
                                             int boxes, uint32_t* x_m, uint8_t* x_be, size_t box_out, const int* gate, int want,
                                             const void* cs, hipStream_t s) {
  if (count <= 0 || boxes <= 0) return 0;
  hipLaunchKernelGGL(k_modp_commit_eval, dim3(grid_for(count), boxes), dim3(BLOCK_THREADS), 0, s, cm, cm, 0x7fffffff, t,
                     positions, count, x_be, x_m, gate, want, (const ModpConsts*)cs, (size_t)t * L, box_positions, box_out);
  return (int)hipGetLastError();
}
// dst[b][0 .. row_words) = src[b * src_stride_words ..): the seeds of a group's boxes side by side for one simultaneous inversion
__global__ void k_modp_gather_rows(const u32* __restrict__ src, size_t src_stride_words, size_t row_words, u32* __restrict__ dst) {
  const u32* from = src + blockIdx.y * src_stride_words;
  u32* to = dst + blockIdx.y * row_words;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < row_words; i += (size_t)gridDim.x * blockDim.x) to[i] = from[i];
}
extern "C" int modp_launch_gather_rows(const uint32_t* src, size_t src_stride_words, size_t row_words, int boxes, uint32_t* dst,
                                       hipStream_t s) {
  if (boxes <= 0 || row_words == 0) return 0;
  const unsigned gx = (unsigned)std::min<size_t>((row_words + 255) / 256, 64);
  hipLaunchKernelGGL(k_modp_gather_rows, dim3(gx, boxes), dim3(256), 0, s, src, src_stride_words, row_words, dst);
  return (int)hipGetLastError();
}
// out[x] = rows[x / group]: one 256-byte challenge per box spread to one per share
__global__ void k_modp_spread_rows(const uint8_t* __restrict__ rows, int group, int count, uint8_t* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // one 16-byte piece each
  if (i >= (size_t)count * 16) return;
  const size_t x = i / 16, piece = i % 16;
  ((uint4*)out)[i] = ((const uint4*)rows)[(x / group) * 16 + piece];
}
extern "C" int modp_launch_spread_rows(const uint8_t* rows, int group, int count, uint8_t* out, hipStream_t s) {
  if (count <= 0) return 0;
  hipLaunchKernelGGL(k_modp_spread_rows, dim3(((size_t)count * 16 + 255) / 256), dim3(256), 0, s, rows, group, count, out);
  return (int)hipGetLastError();
}

extern "C" int modp_fd_tpad(int t) {
  int p = 16;
  while (p < t) p <<= 1;
  return p;
}

extern "C" int modp_launch_fd_check_positions(const int64_t* positions, int count, int* flag, hipStream_t s) {
  hipLaunchKernelGGL(k_modp_fd_check_positions, dim3((count + 255) / 256), dim3(256), 0, s, positions, count, flag, (size_t)0);
  return (int)hipGetLastError();
}
extern "C" int modp_launch_fd_check_positions_boxes(const int64_t* positions, size_t box_positions, int count, int boxes, int* flag,
                                                    hipStream_t s) {
  hipLaunchKernelGGL(k_modp_fd_check_positions, dim3((count + 255) / 256, boxes), dim3(256), 0, s, positions, count, flag,
                     box_positions);
  return (int)hipGetLastError();
}
extern "C" int modp_launch_binv_up(const uint32_t* a, int m, int G, uint32_t* prefix, uint32_t* totals, const int* gate,
                                   const void* cs, hipStream_t s) {
  const int groups = (m + G - 1) / G;
  hipLaunchKernelGGL(k_modp_binv_up, dim3(grid_for(groups)), dim3(BLOCK_THREADS), 0, s, a, m, G, prefix, totals, gate,
                     (const ModpConsts*)cs);
  return (int)hipGetLastError();
}
extern "C" int modp_launch_binv_down(const uint32_t* a, const uint32_t* prefix, const uint32_t* tot_inv, int m, int G,
                                     uint32_t* a_inv, const int* gate, const void* cs, hipStream_t s) {
  const int groups = (m + G - 1) / G;
  hipLaunchKernelGGL(k_modp_binv_down, dim3(grid_for(groups)), dim3(BLOCK_THREADS), 0, s, a, prefix, tot_inv, m, G,
                     a_inv, gate, (const ModpConsts*)cs);
  return (int)hipGetLastError();
}
extern "C" int modp_launch_fd_apply_ok(const int* ok, int* flag, hipStream_t s) {
  hipLaunchKernelGGL(k_modp_fd_apply_ok, dim3(1), dim3(64), 0, s, ok, flag);
  return (int)hipGetLastError();
}
// words of the handoff buffers (zeroed by the caller before every launch)
extern "C" size_t modp_fd_table_hand_words(int chains, int t) {
  return (size_t)chains * (modp_fd_tpad(t) / NUMS_PER_WAVE) * (size_t)t * 2 * L;
}
extern "C" size_t modp_fd_step_hand_words(int chains, int t, int chain_len) {
  return (size_t)2 * chains * (modp_fd_tpad(t) / NUMS_PER_WAVE) * (size_t)(chain_len + t) * L;
}
extern "C" int modp_launch_fd_table(const uint32_t* x, const uint32_t* x_inv, int chains, int t, uint32_t* state,
                                    uint32_t* state_back, uint32_t* hand, int* gate, int inject_fault, const void* cs,
                                    hipStream_t s) {
  const int tpad = modp_fd_tpad(t);
  hipLaunchKernelGGL(k_modp_fd_table, dim3(chains * (tpad / NUMS_PER_WAVE)), dim3(64), 0, s, x, x_inv, chains, t, tpad,
                     state, state_back, hand, gate, inject_fault, (const ModpConsts*)cs, (size_t)0, (size_t)0, (size_t)0, (size_t)0);
  return (int)hipGetLastError();
}
// `boxes` boxes in one launch; box_*: word strides from one box to the next (state and state_back share box_state)
extern "C" int modp_launch_fd_table_boxes(const uint32_t* x, size_t box_x, const uint32_t* x_inv, size_t box_xinv, int chains, int t,
                                          uint32_t* state, uint32_t* state_back, size_t box_state, uint32_t* hand, size_t box_hand,
                                          int boxes, int* gate, int inject_fault, const void* cs, hipStream_t s) {
  const int tpad = modp_fd_tpad(t);
  hipLaunchKernelGGL(k_modp_fd_table, dim3(chains * (tpad / NUMS_PER_WAVE), boxes), dim3(64), 0, s, x, x_inv, chains, t, tpad,
                     state, state_back, hand, gate, inject_fault, (const ModpConsts*)cs, box_x, box_xinv, box_state, box_hand);
  return (int)hipGetLastError();
}
// x_m: base of the chains (position index 0); w0: index of the first seed inside every chain
extern "C" int modp_launch_fd_step(const uint32_t* state, const uint32_t* state_back, int chains, int t, int w0,
                                   int chain_len, int count, uint32_t* x_m, uint32_t* hand, int* gate, int inject_fault,
                                   const void* cs, hipStream_t s) {
  const int tpad = modp_fd_tpad(t);
  hipLaunchKernelGGL(k_modp_fd_step, dim3(2 * chains * (tpad / NUMS_PER_WAVE)), dim3(64), 0, s, state, state_back, chains,
                     t, tpad, w0, chain_len, count, x_m, hand, gate, inject_fault, (const ModpConsts*)cs, (size_t)0, (size_t)0,
                     (size_t)0);
  return (int)hipGetLastError();
}
extern "C" int modp_launch_fd_step_boxes(const uint32_t* state, const uint32_t* state_back, size_t box_state, int chains, int t,
                                         int w0, int chain_len, int count, uint32_t* x_m, size_t box_xm, uint32_t* hand,
                                         size_t box_hand, int boxes, int* gate, int inject_fault, const void* cs, hipStream_t s) {
  const int tpad = modp_fd_tpad(t);
  hipLaunchKernelGGL(k_modp_fd_step, dim3(2 * chains * (tpad / NUMS_PER_WAVE), boxes), dim3(64), 0, s, state, state_back, chains,
                     t, tpad, w0, chain_len, count, x_m, hand, gate, inject_fault, (const ModpConsts*)cs, box_state, box_xm,
                     box_hand);
  return (int)hipGetLastError();
}
extern "C" int modp_launch_from_mont(const uint32_t* x_m, int count, uint8_t* out_be, const int* gate, const void* cs,
                                     hipStream_t s) {
  if (count <= 0) return 0;
  hipLaunchKernelGGL(k_modp_from_mont, dim3(grid_for(count)), dim3(BLOCK_THREADS), 0, s, x_m, count, out_be, gate,
                     (const ModpConsts*)cs);
  return (int)hipGetLastError();
}

extern "C" int modp_launch_build_table(const uint8_t* base_be, int count, uint32_t* tab, const void* cs,
                                       hipStream_t s) {
  if (count <= 0) return 0;
  hipLaunchKernelGGL(k_modp_build_table, dim3(grid_for(count)), dim3(BLOCK_THREADS), 0, s, base_be, count, tab,
                     (const ModpConsts*)cs, 0);
  return (int)hipGetLastError();
}
// entries 0, 1, 3, 5, .. 15 only (what a sliding-window schedule asks for)
extern "C" int modp_launch_build_table_odd(const uint8_t* base_be, int count, uint32_t* tab, const void* cs, hipStream_t s) {
  if (count <= 0) return 0;
  hipLaunchKernelGGL(k_modp_build_table, dim3(grid_for(count)), dim3(BLOCK_THREADS), 0, s, base_be, count, tab,
                     (const ModpConsts*)cs, 1);
  return (int)hipGetLastError();
}

extern "C" int modp_launch_dual_exp(const uint32_t* tab1, size_t tab1_stride, const uint32_t* tab2,
                                    size_t tab2_stride, const uint8_t* e1, const uint8_t* e2, size_t e2_stride,
                                    int e2_windows, int count, uint8_t* out, const void* cs, hipStream_t s) {
  if (count <= 0) return 0;
  hipLaunchKernelGGL(k_modp_dual_exp, dim3(grid_for(count)), dim3(BLOCK_THREADS), 0, s, tab1, tab1_stride, tab2,
                     tab2_stride, e1, e2, e2_stride, e2_windows, count, out, (const ModpConsts*)cs);
  return (int)hipGetLastError();
}

extern "C" int modp_launch_comb_build(const uint8_t* base_be_dev, uint32_t* comb, const void* cs, hipStream_t s) {
  hipLaunchKernelGGL(k_modp_comb_bases, dim3(1), dim3(BLOCK_THREADS), 0, s, base_be_dev, comb, (const ModpConsts*)cs);
  hipLaunchKernelGGL(k_modp_comb_rows, dim3(grid_for(512)), dim3(BLOCK_THREADS), 0, s, comb, (const ModpConsts*)cs);
  return (int)hipGetLastError();
}

extern "C" int modp_launch_build_table64(const uint8_t* base_be, int count, uint32_t* tab, const void* cs, hipStream_t s) {
  if (count <= 0) return 0;
  hipLaunchKernelGGL(k_modp_build_table64, dim3(grid_for(count)), dim3(BLOCK_THREADS), 0, s, base_be, count, tab,
                     (const ModpConsts*)cs);
  return (int)hipGetLastError();
}
extern "C" int modp_launch_dual_exp_w6(const uint32_t* tab1, const uint32_t* tab2, const uint8_t* e1, const uint8_t* c,
                                       size_t c_stride, int count, uint8_t* out, const void* cs, hipStream_t s) {
  if (count <= 0) return 0;
  hipLaunchKernelGGL(k_modp_dual_exp_w6, dim3(grid_for(count)), dim3(BLOCK_THREADS), 0, s, tab1, tab2, e1, c, c_stride, count,
                     out, (const ModpConsts*)cs, (const uint16_t*)nullptr);
  return (int)hipGetLastError();
}
// one c for all shares, given as a sliding-window schedule (device memory; tab2: odd entries)
extern "C" int modp_launch_dual_exp_w6_sched(const uint32_t* tab1, const uint32_t* tab2, const uint8_t* e1,
                                             const uint16_t* c_sched, int count, uint8_t* out, const void* cs, hipStream_t s) {
  if (count <= 0) return 0;
  hipLaunchKernelGGL(k_modp_dual_exp_w6, dim3(grid_for(count)), dim3(BLOCK_THREADS), 0, s, tab1, tab2, e1,
                     (const uint8_t*)nullptr, (size_t)0, count, out, (const ModpConsts*)cs, c_sched);
  return (int)hipGetLastError();
}
extern "C" size_t modp_twin_exp_bucket_words() { return (size_t)2 * BK_ENT * L; }
// Y = base^e1, A = base^e2 (same base): buckets [count][2][31][72] u32 and occupancy [count][2] u32 are scratch
extern "C" int modp_launch_twin_exp(const uint8_t* base_be, const uint8_t* e1, const uint8_t* e2, int count, uint32_t* buckets,
                                    uint32_t* occupancy, uint8_t* out1, uint8_t* out2, const void* cs, hipStream_t s) {
  if (count <= 0) return 0;
  hipLaunchKernelGGL(k_modp_twin_exp_buckets, dim3(grid_for(count)), dim3(BLOCK_THREADS), 0, s, base_be, e1, e2, count,
                     buckets, occupancy, (const ModpConsts*)cs);
  hipLaunchKernelGGL(k_modp_bucket_combine, dim3(grid_for(2 * count)), dim3(BLOCK_THREADS), 0, s, buckets, occupancy, count,
                     out1, out2, (const ModpConsts*)cs);
  return (int)hipGetLastError();
}
// the second phase alone (after the pair-layout bucket kernel, modp_pair_kernels.hip)
extern "C" int modp_launch_bucket_combine(const uint32_t* buckets, const uint32_t* occupancy, int count, uint8_t* out1, uint8_t* out2,
                                          const void* cs, hipStream_t s) {
  if (count <= 0) return 0;
  hipLaunchKernelGGL(k_modp_bucket_combine, dim3(grid_for(2 * count)), dim3(BLOCK_THREADS), 0, s, buckets, occupancy, count,
                     out1, out2, (const ModpConsts*)cs);
  return (int)hipGetLastError();
}
extern "C" size_t modp_keyset_words_per_key() { return (size_t)KS_SUB * KS_ENT * L; }
extern "C" int modp_launch_keyset_build(const uint8_t* pk_be, int count, uint32_t* ks, const void* cs, hipStream_t s) {
  if (count <= 0) return 0;
  hipLaunchKernelGGL(k_modp_keyset_bases, dim3(grid_for(count)), dim3(BLOCK_THREADS), 0, s, pk_be, count, ks,
                     (const ModpConsts*)cs);
  const long long rows = (long long)count * KS_SUB;
  hipLaunchKernelGGL(k_modp_keyset_rows, dim3((unsigned)((rows + NUMS_PER_BLOCK - 1) / NUMS_PER_BLOCK)), dim3(BLOCK_THREADS),
                     0, s, ks, (int)rows, (const ModpConsts*)cs);
  return (int)hipGetLastError();
}
extern "C" int modp_launch_keyset_dual_exp(const uint32_t* ks, const uint32_t* tab2, const uint8_t* r, const uint8_t* c,
                                           int count, uint8_t* out, const void* cs, hipStream_t s) {
  if (count <= 0) return 0;
  hipLaunchKernelGGL(k_modp_keyset_dual_exp, dim3(grid_for(count)), dim3(BLOCK_THREADS), 0, s, ks, tab2, r, c, count, out,
                     (const ModpConsts*)cs);
  return (int)hipGetLastError();
}

// wide comb from the 4-bit comb of the same base (16 launches)
extern "C" int modp_launch_comb16_build(const uint32_t* comb4, uint32_t* comb16, const void* cs, hipStream_t s) {
  hipLaunchKernelGGL(k_modp_comb16_init, dim3(128), dim3(128), 0, s, comb4, comb16, (const ModpConsts*)cs);
  for (int j = 1; j < 16; ++j)
    hipLaunchKernelGGL(k_modp_comb16_pass, dim3(grid_for(128 << j)), dim3(BLOCK_THREADS), 0, s, comb16, j,
                       (const ModpConsts*)cs);
  return (int)hipGetLastError();
}

extern "C" int modp_launch_comb_dual_exp(const uint32_t* comb, const uint32_t* tab2, size_t tab2_stride,
                                         const uint8_t* e1, const uint8_t* e2, size_t e2_stride, int e2_windows,
                                         int count, uint8_t* out, int comb_bits, const void* cs, hipStream_t s) {
  if (count <= 0) return 0;
  hipLaunchKernelGGL(k_modp_comb_dual_exp, dim3(grid_for(count)), dim3(BLOCK_THREADS), 0, s, comb, tab2, tab2_stride,
                     e1, e2, e2_stride, e2_windows, count, out, 0, (uint32_t*)nullptr, comb_bits, (const ModpConsts*)cs,
                     (const uint16_t*)nullptr);
  return (int)hipGetLastError();
}

// the two halves of the same product: mode 1 = g^e1 into p_m (Montgomery form), mode 2 = B2^e2 * p_m
extern "C" int modp_launch_comb_dual_exp_split(const uint32_t* comb, const uint32_t* tab2, size_t tab2_stride,
                                               const uint8_t* e1, const uint8_t* e2, size_t e2_stride, int e2_windows,
                                               int count, uint8_t* out, int mode, uint32_t* p_m, int comb_bits,
                                               const void* cs, hipStream_t s) {
  if (count <= 0) return 0;
  hipLaunchKernelGGL(k_modp_comb_dual_exp, dim3(grid_for(count)), dim3(BLOCK_THREADS), 0, s, comb, tab2, tab2_stride,
                     e1, e2, e2_stride, e2_windows, count, out, mode, p_m, comb_bits, (const ModpConsts*)cs,
                     (const uint16_t*)nullptr);
  return (int)hipGetLastError();
}
// mode 2 with the shared second exponent given as a sliding-window schedule (tab2: odd entries)
extern "C" int modp_launch_comb_dual_exp_sched(const uint32_t* comb, const uint32_t* tab2, size_t tab2_stride,
                                               const uint16_t* c_sched, int count, uint8_t* out, uint32_t* p_m, int comb_bits,
                                               const void* cs, hipStream_t s) {
  if (count <= 0) return 0;
  hipLaunchKernelGGL(k_modp_comb_dual_exp, dim3(grid_for(count)), dim3(BLOCK_THREADS), 0, s, comb, tab2, tab2_stride,
                     (const uint8_t*)nullptr, (const uint8_t*)nullptr, (size_t)0, 64, count, out, 2, p_m, comb_bits,
                     (const ModpConsts*)cs, c_sched);
  return (int)hipGetLastError();
}
