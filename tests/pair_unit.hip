// Test helper (built by `make examples` into tests/_build/pair_unit, run by tests/test_gpu_pair_edges.py): ONE Montgomery
// operation of the pair layout (mpvss_rs_amd/csrc/bn_pair.h -- phase A on the VALU, the reduction on the matrix cores) per
// operand pair, so that the test can compare every result with Python integers on operands chosen to sit at the bounds the
// integer model (tools/mfma_mont/model.py) proves: 0, 1, N-1, N, N+1, 2N-1, every limb at the almost-normalised maximum,
// T_lo = 0, T_lo = R-1.
//   pair_unit <in.bin> <out.bin> <n>
// in.bin : n x 2 x 72 u32 limbs (radix 2^29): a, b        out.bin: n x 3 x 72 limbs: a*b, a*a, a*b with b read from the
// NEXT number's slot (the forward-difference stepping form of phase A: number j multiplies by b of number j+1 of its wave;
// the last number of a wave by its own b)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../mpvss_rs_amd/csrc/bn_pair.h"

#define CHECK(x)                                                                       \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } \
  } while (0)

constexpr int WAVES = 2;

__global__ void __launch_bounds__(64 * WAVES)
k_unit(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int n, const mm::Tables* __restrict__ gt) {
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
  uint32_t a[mm::LP], b[mm::LP], r[mm::LP];
#pragma unroll
  for (int k = 0; k < mm::LP; ++k) {
    a[k] = in[((size_t)num * 2) * mm::L + mm::LP * pl.h + k];
    b[k] = in[((size_t)num * 2 + 1) * mm::L + mm::LP * pl.h + k];
  }
#pragma nounroll
  for (int which = 0; which < 3; ++which) {
#pragma unroll
    for (int k = 0; k < mm::LP; ++k) slot[mm::LP * pl.h + k] = which == 1 ? a[k] : b[k];
    __builtin_amdgcn_wave_barrier();
    uint64_t T[mm::LP];
    if (which == 1)
      mm::phase_a<true>(T, a, slot, junk[wave], pl, slot);
    else
      mm::phase_a<false>(T, a, slot, junk[wave], pl, (which == 2 && j != 31) ? slot + mm::SLOTW : slot);
    mm::reduce(r, T, slot, &tb, pl);
    __builtin_amdgcn_wave_barrier();
    if (live)
#pragma unroll
      for (int k = 0; k < mm::LP; ++k) out[((size_t)num * 3 + which) * mm::L + mm::LP * pl.h + k] = r[k];
    asm volatile("" ::: "memory");
  }
}

int main(int argc, char** argv) {
  if (argc < 4) { fprintf(stderr, "usage: pair_unit in.bin out.bin n\n"); return 1; }
  const int n = atoi(argv[3]);
  std::vector<uint32_t> hin((size_t)n * 2 * 72), hout((size_t)n * 3 * 72);
  FILE* f = fopen(argv[1], "rb");
  if (!f || fread(hin.data(), 4, hin.size(), f) != hin.size()) { fprintf(stderr, "cannot read %s\n", argv[1]); return 1; }
  fclose(f);
  mm::Tables* ht = new mm::Tables;
  static_assert(sizeof(MM_GT1) == sizeof(ht->gt1) && sizeof(MM_GT2) == sizeof(ht->gt2) && sizeof(MM_C1) == sizeof(ht->c1) &&
                    sizeof(MM_C2) == sizeof(ht->c2), "tables");
  memcpy(ht->gt1, MM_GT1, sizeof(MM_GT1));
  memcpy(ht->gt2, MM_GT2, sizeof(MM_GT2));
  memcpy(ht->c1, MM_C1, sizeof(MM_C1));
  memcpy(ht->c2, MM_C2, sizeof(MM_C2));
  mm::Tables* dt;
  uint32_t *din, *dout;
  CHECK(hipMalloc(&dt, sizeof(mm::Tables)));
  CHECK(hipMemcpy(dt, ht, sizeof(mm::Tables), hipMemcpyHostToDevice));
  CHECK(hipMalloc(&din, hin.size() * 4));
  CHECK(hipMalloc(&dout, hout.size() * 4));
  CHECK(hipMemcpy(din, hin.data(), hin.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemset(dout, 0xff, hout.size() * 4));
  hipLaunchKernelGGL(k_unit, dim3((n + 32 * WAVES - 1) / (32 * WAVES)), dim3(64 * WAVES), 0, 0, din, dout, n, dt);
  CHECK(hipGetLastError());
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemcpy(hout.data(), dout, hout.size() * 4, hipMemcpyDeviceToHost));
  FILE* g = fopen(argv[2], "wb");
  if (!g || fwrite(hout.data(), 4, hout.size(), g) != hout.size()) { fprintf(stderr, "cannot write %s\n", argv[2]); return 1; }
  fclose(g);
  printf("pair_unit: %d operand pairs done\n", n);
  return 0;
}
