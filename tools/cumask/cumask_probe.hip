// Where do the workgroups of a stream created with hipExtStreamCreateWithCUMask run?  Every workgroup records
// (XCC_ID, HW_ID: SE / CU) and the probe prints how many distinct (xcc, se, cu) triples each mask reached.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <set>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
__global__ void k_where(unsigned* out, int spin) {
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  // keep the CU busy for a while so that the workgroups spread over every CU the stream may use
  unsigned long long t0 = clock64();
  while (clock64() - t0 < (unsigned long long)spin) {}
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}
int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  printf("multiProcessorCount %d\n", prop.multiProcessorCount);
  const int nwg = 4096;
  unsigned* d;
  CHECK(hipMalloc(&d, nwg * 8));
  std::vector<unsigned> h(nwg * 2);
  // contiguous ranges [lo, hi) of the 256 mask bits, and a few strided patterns
  struct Pat { const char* name; int lo, hi, stride_keep, stride_of; };
  const Pat pats[] = {{"all", 0, 256, 1, 1},       {"[0,32)", 0, 32, 1, 1},     {"[0,64)", 0, 64, 1, 1},     {"[0,96)", 0, 96, 1, 1},
                      {"[0,128)", 0, 128, 1, 1},   {"[0,160)", 0, 160, 1, 1},   {"[0,192)", 0, 192, 1, 1},   {"[192,256)", 192, 256, 1, 1},
                      {"[64,256)", 64, 256, 1, 1}, {"3 of 4", 0, 256, 3, 4},    {"1 of 4", 0, 256, 1, 4},    {"6 of 8", 0, 256, 6, 8},
                      {"12 of 16", 0, 256, 12, 16}, {"24 of 32", 0, 256, 24, 32}, {"[0,8)", 0, 8, 1, 1}};
  for (const Pat& pt : pats) {
    uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int bits = 0;
    for (int b = pt.lo; b < pt.hi; ++b)
      if ((b % pt.stride_of) < pt.stride_keep) { mask[b / 32] |= 1u << (b % 32); ++bits; }
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, mask);
    if (e != hipSuccess) { printf("%s: hipExtStreamCreateWithCUMask failed: %s\n", pt.name, hipGetErrorString(e)); continue; }
    CHECK(hipMemsetAsync(d, 0xff, nwg * 8, s));
    hipLaunchKernelGGL(k_where, dim3(nwg), dim3(256), 0, s, d, 200000);
    CHECK(hipStreamSynchronize(s));
    CHECK(hipMemcpy(h.data(), d, nwg * 8, hipMemcpyDeviceToHost));
    std::set<unsigned> cus;
    int per_xcc[16] = {0};
    std::set<unsigned> seen_x[16];
    for (int i = 0; i < nwg; ++i) {
      const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
      const unsigned key = (xcc << 16) | (hw & 0xff00);       // bits 8..15 of HW_ID: cu, sh, se
      cus.insert(key);
      seen_x[xcc].insert(key);
    }
    printf("%-10s bits %3d -> %3zu distinct CUs; per xcc:", pt.name, bits, cus.size());
    for (int x = 0; x < 8; ++x) printf(" %zu", seen_x[x].size());
    printf("\n");
    (void)per_xcc;
    CHECK(hipStreamDestroy(s));
  }
  return 0;
}
