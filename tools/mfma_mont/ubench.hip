// Microbenchmark + exactness check of the MFMA-assisted Montgomery product (mont_pair.h) against today's VALU-only
// product (mpvss_rs_amd/csrc/bn_quad.h): a chain of S Montgomery squarings of n numbers with each.
//   ubench <in.bin> <out_prefix> <n> <S> [reps]
// in.bin: n x 72 u32 limbs (radix 2^29, values < 2N).  Writes <out_prefix>.pair.bin / .quad.bin (n x 72 limbs) and prints
// kernel times.  tools/mfma_mont/run.py generates the inputs and checks both outputs against Python integers.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../mpvss_rs_amd/csrc/bn_quad.h"
#include "../../mpvss_rs_amd/csrc/modp2048_consts.h"
#include "../../mpvss_rs_amd/csrc/bn_pair.h"

#define CHECK(x)                                                                       \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } \
  } while (0)

#ifndef UB_WAVES
#define UB_WAVES 8
#endif
constexpr int WAVES = UB_WAVES;   // waves per workgroup of the pair kernel (one workgroup per CU: the tables take 39 KB of LDS)

#ifndef UB_WPE
#define UB_WPE 2
#endif
#ifndef UB_STAGGER
#define UB_STAGGER 0     // odd waves of a workgroup start this many s_sleep units late (the two waves of a SIMD run the same program)
#endif
// UB_STAMPS: a diagnostic build -- every wave adds up the shader-clock cycles it spends in phase A and in the reduction
// (s_memtime around each) and notes the wall clock (s_memrealtime, 100 MHz) around its loop; the sums go to a buffer of their
// own that nothing else reads: stamps[wave][4] = {cycles in phase A, cycles in reduce, loop cycles, loop wall ticks}
template <bool MUL>
__global__ void __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(UB_WPE, UB_WPE))) k_chain_pair(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int n,
                                                          int S, const mm::Tables* __restrict__ gt, unsigned long long* __restrict__ stamps) {
  __shared__ mm::Tables tb;
  __shared__ __attribute__((aligned(16))) uint32_t slots[WAVES][32 * mm::SLOTW];
  __shared__ uint32_t junk[WAVES][mm::L];
  {
    const uint4* src = reinterpret_cast<const uint4*>(gt);
    uint4* dst = reinterpret_cast<uint4*>(&tb);
    for (int i = threadIdx.x; i < (int)(sizeof(mm::Tables) / 16); i += blockDim.x) dst[i] = src[i];
  }
  __syncthreads();
  const mm::PairLane pl = mm::make_pair_lane();
  const int wave = threadIdx.x >> 6, j = pl.lane & 31;
  int num = (blockIdx.x * WAVES + wave) * 32 + j;
  const bool live = num < n;
  if (!live) num = n - 1;
  uint32_t* slot = &slots[wave][j * mm::SLOTW];
  uint32_t a[mm::LP];
#pragma unroll
  for (int k = 0; k < mm::LP; ++k) {
    a[k] = in[(size_t)num * mm::L + mm::LP * pl.h + k];
    slot[mm::LP * pl.h + k] = a[k];
  }
#if UB_STAGGER > 0
  if (wave & 1) __builtin_amdgcn_s_sleep(UB_STAGGER);
#endif
#ifdef UB_STAMPS
  unsigned long long cyc_a = 0, cyc_r = 0;
  const unsigned long long loop_c0 = __builtin_amdgcn_s_memtime(), loop_w0 = __builtin_amdgcn_s_memrealtime();
#endif
#pragma nounroll
  for (int s = 0; s < S; ++s) {
    uint32_t r[mm::LP];
#ifdef UB_STAMPS
    {
      if (MUL) {
        const uint4* g4 = reinterpret_cast<const uint4*>(in + (size_t)num * mm::L + mm::LP * pl.h);
        uint4* s4 = reinterpret_cast<uint4*>(slot + mm::LP * pl.h);
#pragma unroll
        for (int c = 0; c < mm::LP / 4; ++c) s4[c] = g4[c];
        __builtin_amdgcn_wave_barrier();
      }
      uint64_t T[mm::LP];
      const unsigned long long t0 = __builtin_amdgcn_s_memtime();
      mm::phase_a<!MUL>(T, a, slot, junk[wave], pl);
      const unsigned long long t1 = __builtin_amdgcn_s_memtime();
      mm::reduce(r, T, slot, &tb, pl);
      const unsigned long long t2 = __builtin_amdgcn_s_memtime();
      cyc_a += t1 - t0;
      cyc_r += t2 - t1;
#pragma unroll
      for (int k = 0; k < mm::LP; ++k) {
        a[k] = r[k];
        if (!MUL) slot[mm::LP * pl.h + k] = r[k];
      }
      asm volatile("" ::: "memory");
      continue;
    }
#endif
    if (MUL) {
      const uint4* g4 = reinterpret_cast<const uint4*>(in + (size_t)num * mm::L + mm::LP * pl.h);
      uint4* s4 = reinterpret_cast<uint4*>(slot + mm::LP * pl.h);
#pragma unroll
      for (int c = 0; c < mm::LP / 4; ++c) s4[c] = g4[c];
      __builtin_amdgcn_wave_barrier();
      mm::mont_pair<false>(r, a, slot, junk[wave], &tb, pl);
#pragma unroll
      for (int k = 0; k < mm::LP; ++k) a[k] = r[k];
    } else {
      mm::mont_pair<true>(r, a, slot, junk[wave], &tb, pl);
#pragma unroll
      for (int k = 0; k < mm::LP; ++k) {
        a[k] = r[k];
        slot[mm::LP * pl.h + k] = r[k];
      }
    }
    asm volatile("" ::: "memory");
  }
#ifdef UB_STAMPS
  if (pl.lane == 0) {
    unsigned long long* st = stamps + (size_t)(blockIdx.x * WAVES + wave) * 4;
    st[0] = cyc_a;
    st[1] = cyc_r;
    st[2] = __builtin_amdgcn_s_memtime() - loop_c0;
    st[3] = __builtin_amdgcn_s_memrealtime() - loop_w0;
  }
#endif
  if (live)
#pragma unroll
    for (int k = 0; k < mm::LP; ++k) out[(size_t)num * mm::L + mm::LP * pl.h + k] = a[k];
}

// today's product: one number per DPP quad, 16 numbers per single-wave workgroup
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3)))
k_chain_quad(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int n, int S, const uint32_t* __restrict__ nl) {
  __shared__ __attribute__((aligned(16))) uint32_t slots[16 * bn::SLOT_WORDS];
  const bn::Lane ln = bn::make_lane();
  int num = blockIdx.x * 16 + (threadIdx.x >> 2);
  const bool live = num < n;
  if (!live) num = n - 1;
  uint32_t* slot = slots + (threadIdx.x >> 2) * bn::SLOT_WORDS;
  uint32_t a[bn::LPL], nn[bn::LPL];
#pragma unroll
  for (int k = 0; k < bn::LPL; ++k) {
    a[k] = in[(size_t)num * bn::L + ln.q * bn::LPL + k];
    nn[k] = nl[ln.q * bn::LPL + k];
  }
#pragma nounroll
  for (int s = 0; s < S; ++s) {
    bn::slot_store(slot, a, ln);
    __builtin_amdgcn_wave_barrier();
    uint32_t r[bn::LPL];
    bn::mont_sqr<MODP_N0INV>(r, a, slot, nn, ln);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < bn::LPL; ++k) a[k] = r[k];
  }
  if (live)
#pragma unroll
    for (int k = 0; k < bn::LPL; ++k) out[(size_t)num * bn::L + ln.q * bn::LPL + k] = a[k];
}

// exact-integer check of the operand / result lane maps this code assumes for v_mfma_i32_32x32x32_i8 and of v_permlane32_swap
__global__ void k_selftest(int* bad) {
  const int l = threadIdx.x, row = l & 31, h = l >> 5;
  mm::v4i A, B;
  // A[row][k] = row - 16 + (k % 5), B[k][col] = (col % 7) - 3 + (k % 3): asymmetric, small
  for (int w = 0; w < 4; ++w) {
    unsigned pa = 0, pb = 0;
    for (int b = 0; b < 4; ++b) {
      const int k = 16 * h + 4 * w + b;
      pa |= (unsigned)((row - 16 + (k % 5)) & 0xff) << (8 * b);
      pb |= (unsigned)(((row % 7) - 3 + (k % 3)) & 0xff) << (8 * b);
    }
    A[w] = (int)pa;
    B[w] = (int)pb;
  }
  mm::v16i acc;
  for (int r = 0; r < 16; ++r) acc[r] = 1000 * r + l;
  acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B, acc, 0, 0, 0);
  int errs = 0;
  for (int r = 0; r < 16; ++r) {
    const int drow = (r & 3) + 8 * (r >> 2) + 4 * h, dcol = l & 31;
    int want = 1000 * r + l;
    for (int k = 0; k < 32; ++k) want += (drow - 16 + (k % 5)) * ((dcol % 7) - 3 + (k % 3));
    errs += acc[r] != want;
  }
  unsigned x = 100 + l, y = 200 + l;
  mm::swap32(x, y);     // lanes 32..63 of x <-> lanes 0..31 of y
  const unsigned wx = l < 32 ? 100u + l : 200u + (l - 32), wy = l < 32 ? 100u + (l + 32) : 200u + l;
  errs += (x != wx) + (y != wy);
  if (errs) atomicAdd(bad, errs);
}

int main(int argc, char** argv) {
  if (argc < 5) { fprintf(stderr, "usage: ubench in.bin out_prefix n S [reps]\n"); return 1; }
  const int n = atoi(argv[3]), S = atoi(argv[4]), reps = argc > 5 ? atoi(argv[5]) : 3;
  std::vector<uint32_t> hin((size_t)n * 72);
  FILE* f = fopen(argv[1], "rb");
  if (!f || fread(hin.data(), 4, hin.size(), f) != hin.size()) { fprintf(stderr, "cannot read %s\n", argv[1]); return 1; }
  fclose(f);
  int* dbad;
  CHECK(hipMalloc(&dbad, 4));
  CHECK(hipMemset(dbad, 0, 4));
  hipLaunchKernelGGL(k_selftest, dim3(1), dim3(64), 0, 0, dbad);
  int hbad = -1;
  CHECK(hipMemcpy(&hbad, dbad, 4, hipMemcpyDeviceToHost));
  printf("selftest (MFMA i8 32x32x32 lane maps, permlane32_swap): %s (%d)\n", hbad == 0 ? "ok" : "MISMATCH", hbad);
  if (hbad != 0) return 3;
  mm::Tables* ht = new mm::Tables;
  static_assert(sizeof(MM_GT1) == sizeof(ht->gt1) && sizeof(MM_GT2) == sizeof(ht->gt2) && sizeof(MM_C1) == sizeof(ht->c1), "tables");
  memcpy(ht->gt1, MM_GT1, sizeof(MM_GT1));
  memcpy(ht->gt2, MM_GT2, sizeof(MM_GT2));
  memcpy(ht->c1, MM_C1, sizeof(MM_C1));
  memcpy(ht->c2, MM_C2, sizeof(MM_C2));
  mm::Tables* dt;
  uint32_t *din, *dout, *dn;
  CHECK(hipMalloc(&dt, sizeof(mm::Tables)));
  CHECK(hipMemcpy(dt, ht, sizeof(mm::Tables), hipMemcpyHostToDevice));
  CHECK(hipMalloc(&din, hin.size() * 4));
  CHECK(hipMalloc(&dout, hin.size() * 4));
  CHECK(hipMalloc(&dn, 72 * 4));
  CHECK(hipMemcpy(din, hin.data(), hin.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dn, MODP_N_LIMBS, 72 * 4, hipMemcpyHostToDevice));
  {
    int nb = -1;
    hipError_t oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)k_chain_pair<false>, 64 * WAVES, 0);
    printf("pair kernel: %d waves per workgroup, occupancy API: %d workgroups per CU (%s)\n", WAVES, nb, hipGetErrorString(oe));
  }
  const int nwaves = ((n + 32 * WAVES - 1) / (32 * WAVES)) * WAVES;
  unsigned long long* dstamps;
  CHECK(hipMalloc(&dstamps, (size_t)nwaves * 4 * 8));
  CHECK(hipMemset(dstamps, 0, (size_t)nwaves * 4 * 8));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  std::vector<uint32_t> hout(hin.size());
  for (int which = 0; which < 3; ++which) {
    float best = 1e30f;
    for (int rep = 0; rep < reps; ++rep) {
      CHECK(hipMemset(dout, 0, hin.size() * 4));
      CHECK(hipEventRecord(e0, 0));
      if (which == 0)
        hipLaunchKernelGGL(k_chain_pair<false>, dim3((n + 32 * WAVES - 1) / (32 * WAVES)), dim3(64 * WAVES), 0, 0, din, dout, n, S, dt, dstamps);
      else if (which == 2)
        hipLaunchKernelGGL(k_chain_pair<true>, dim3((n + 32 * WAVES - 1) / (32 * WAVES)), dim3(64 * WAVES), 0, 0, din, dout, n, S, dt, dstamps);
      else
        hipLaunchKernelGGL(k_chain_quad, dim3((n + 15) / 16), dim3(64), 0, 0, din, dout, n, S, dn);
      CHECK(hipGetLastError());
      CHECK(hipEventRecord(e1, 0));
      CHECK(hipEventSynchronize(e1));
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    CHECK(hipMemcpy(hout.data(), dout, hout.size() * 4, hipMemcpyDeviceToHost));
    char name[512];
    snprintf(name, sizeof(name), "%s.%s.bin", argv[2], which == 0 ? "pair" : (which == 1 ? "quad" : "pairmul"));
    FILE* g = fopen(name, "wb");
    fwrite(hout.data(), 4, hout.size(), g);
    fclose(g);
    printf("%s: n=%d S=%d best of %d: %.3f ms -> %.3f G %s/s\n",
           which == 0 ? "pair (MFMA reduction)" : (which == 1 ? "quad (VALU only)   " : "pair, products       "), n, S, reps, best,
           (double)n * S / (best * 1e-3) / 1e9, which == 2 ? "products" : "squarings");
#ifdef UB_STAMPS
    if (which != 1) {
      std::vector<unsigned long long> hs((size_t)nwaves * 4);
      CHECK(hipMemcpy(hs.data(), dstamps, hs.size() * 8, hipMemcpyDeviceToHost));
      double a = 0, r = 0, c = 0, w = 0;
      for (int i = 0; i < nwaves; ++i) { a += hs[4 * i]; r += hs[4 * i + 1]; c += hs[4 * i + 2]; w += hs[4 * i + 3]; }
      printf("  stamps (mean over %d waves, per operation): phase A %.0f cycles, reduce %.0f cycles, whole iteration %.0f cycles; "
             "shader clock %.3f GHz\n", nwaves, a / nwaves / S, r / nwaves / S, c / nwaves / S, c / w * 0.1);
    }
#endif
  }
  return 0;
}
