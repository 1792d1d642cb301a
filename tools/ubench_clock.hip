// In-kernel clock under sustained v_mad_u64_u32 load: delta(s_memtime) / delta(s_memrealtime) * 100 MHz,
// and the sustained mad issue rate per SIMD at 1..8 waves per SIMD over ~100 ms.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP16(X) REP8(X) REP8(X)
__global__ void __launch_bounds__(256) k_mad(uint32_t* out, uint64_t* stamps, int iters, uint32_t seed) {
  uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9e3779b9u;
  uint64_t p[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) p[k] = a + k;
  uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#define X(k) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(p[k]) : "v"(a), "v"(b) : "vcc");
    REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
  }
  uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) acc ^= p[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(acc ^ (acc >> 32));
  if ((threadIdx.x & 63) == 0) { size_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; stamps[2 * w] = t1 - t0; stamps[2 * w + 1] = r1 - r0; }
}
int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  uint32_t* d_out; uint64_t* d_st;
  int maxw = cus * 4 * 8;
  CHECK(hipMalloc(&d_out, (size_t)maxw * 64 * 4)); CHECK(hipMalloc(&d_st, (size_t)maxw * 16));
  std::vector<uint64_t> st(maxw * 2);
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int iters = 200000;  // x64 mads per wave
  for (int wps : {1, 2, 3, 4, 6, 8}) {
    int blocks = cus * wps;
    hipLaunchKernelGGL(k_mad, dim3(blocks), dim3(256), 0, 0, d_out, d_st, 1000, 1u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_mad, dim3(blocks), dim3(256), 0, 0, d_out, d_st, iters / wps, 2u);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipMemcpy(st.data(), d_st, (size_t)blocks * 4 * 16, hipMemcpyDeviceToHost));
    double sm = 0, sr = 0; int nw = blocks * 4;
    for (int i = 0; i < nw; ++i) { sm += st[2 * i]; sr += st[2 * i + 1]; }
    double insts = (double)(iters / wps) * 64;
    double clk_ghz = (sm / sr) * 0.1;                 // memrealtime = 100 MHz
    double ns_per_inst_simd = ms * 1e6 / (insts * wps);
    printf("waves/SIMD=%d  wall=%.1f ms  in-kernel clock=%.3f GHz  mad: %.2f ns per SIMD-issue = %.2f cycles at that clock (memtime ticks/inst/wave %.2f)\n",
           wps, ms, clk_ghz, ns_per_inst_simd, ns_per_inst_simd * clk_ghz, sm / nw / insts);
  }
  return 0;
}
