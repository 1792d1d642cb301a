// Test helper (built by `make examples` into tests/_build/row_unit, run by tests/test_gpu_row.py): ONE Montgomery operation of the
// row layout (mpvss_rs_amd/csrc/bn_row.h: 16 lanes per number) per operand pair, compared by the test with Python integers on
// operands at the bounds of the integer model (tests/test_limb_model.py::test_row_layout_model): 0, 1, N-1, N, N+1, 2N-1 and
// every limb at the almost-normalised maximum.
//   row_unit <in.bin> <out.bin> <n>
// in.bin : 72 u32 limbs of the modulus N (radix 2^29), then n x 2 x 72 limbs: a, b      out.bin: n x 2 x 72 limbs: a*b R^-1, a*a R^-1
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "../mpvss_rs_amd/csrc/bn_row.h"

#define CHECK(x)                                                                       \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } \
  } while (0)

using namespace bnrow;

__global__ void __launch_bounds__(64) k_unit(const u32* __restrict__ nmod, const u32* __restrict__ in, u32* __restrict__ out, int count) {
  __shared__ __attribute__((aligned(16))) u32 lds[NUMS_PER_WAVE * SLOT_WORDS];
  const Lane ln = make_lane();
  const int num = threadIdx.x >> 4;
  const int xi = blockIdx.x * NUMS_PER_WAVE + num;
  const bool live = xi < count;
  const int x = live ? xi : count - 1;
  u32* slot = lds + num * SLOT_WORDS;
  u32 n[LPL], a[LPL], r[LPL];
  load_lane_limbs(n, nmod, ln);
  load_lane_limbs(a, in + ((size_t)x * 2) * L, ln);
  slot_fill_from_global(slot, in + ((size_t)x * 2 + 1) * L, ln);
  __builtin_amdgcn_wave_barrier();
  mont_mul<1>(r, a, slot, n, ln);
  __builtin_amdgcn_wave_barrier();
  if (live) store_lane_limbs(out + ((size_t)x * 2) * L, r, ln);
  slot_store(slot, a, ln);
  __builtin_amdgcn_wave_barrier();
  mont_sqr<1>(r, a, slot, n, ln);
  __builtin_amdgcn_wave_barrier();
  if (live) store_lane_limbs(out + ((size_t)x * 2 + 1) * L, r, ln);
}

int main(int argc, char** argv) {
  if (argc < 4) { fprintf(stderr, "usage: row_unit in.bin out.bin n\n"); return 1; }
  const int n = atoi(argv[3]);
  std::vector<uint32_t> hin(72 + (size_t)n * 2 * 72), hout((size_t)n * 2 * 72);
  FILE* f = fopen(argv[1], "rb");
  if (!f || fread(hin.data(), 4, hin.size(), f) != hin.size()) { fprintf(stderr, "cannot read %s\n", argv[1]); return 1; }
  fclose(f);
  uint32_t *din, *dout;
  CHECK(hipMalloc(&din, hin.size() * 4));
  CHECK(hipMalloc(&dout, hout.size() * 4));
  CHECK(hipMemcpy(din, hin.data(), hin.size() * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_unit, dim3((n + NUMS_PER_WAVE - 1) / NUMS_PER_WAVE), dim3(64), 0, 0, din, din + 72, dout, n);
  CHECK(hipGetLastError());
  CHECK(hipMemcpy(hout.data(), dout, hout.size() * 4, hipMemcpyDeviceToHost));
  f = fopen(argv[2], "wb");
  if (!f || fwrite(hout.data(), 4, hout.size(), f) != hout.size()) { fprintf(stderr, "cannot write %s\n", argv[2]); return 1; }
  fclose(f);
  return 0;
}
