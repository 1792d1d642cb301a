// Issue-rate microbenchmark for the gfx950 VALU / cross-lane instructions that a
// wide-integer (2048-bit Montgomery, 256-bit field) kernel is built from.
// Reports cycles per wave-instruction per SIMD at 1, 2 and 4 waves per SIMD.
//
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_valu.hip -o tools/ubench_valu
// Run on the GPU box: ./tools/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int ITERS = 512;
constexpr int UNROLL = 16;   // instructions per loop body

// Each body is UNROLL instructions over 8 independent destination sets.
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP16(X) REP8(X) REP8(X)

#define KERNEL_BEGIN(name) \
__global__ void __launch_bounds__(256) name(uint32_t* out, uint64_t* cyc, uint32_t seed) { \
  uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9e3779b9u; \
  uint32_t r0[8], r1[8]; \
  _Pragma("unroll") for (int k = 0; k < 8; ++k) { r0[k] = a + k; r1[k] = b + k; } \
  double d[8], da = 1.0 + a * 1e-9, db = 1.0 + b * 1e-10; \
  _Pragma("unroll") for (int k = 0; k < 8; ++k) d[k] = k + 0.5; \
  uint64_t t0 = __builtin_amdgcn_s_memtime(); \
  for (int it = 0; it < ITERS; ++it) {

#define KERNEL_END \
  } \
  uint64_t t1 = __builtin_amdgcn_s_memtime(); \
  uint32_t acc = 0; double dacc = 0; \
  _Pragma("unroll") for (int k = 0; k < 8; ++k) { acc ^= r0[k] ^ r1[k]; dacc += d[k]; } \
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc ^ (uint32_t)__double2ll_rn(dacc); \
  if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0; \
}

// ---- integer multiply family
#define I_MAD64(k) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0" : "+v"(*(uint64_t*)&pr[k]) , "=v"(dummy) : "v"(a), "v"(b) : "vcc");

__global__ void __launch_bounds__(256) k_mad_u64_u32(uint32_t* out, uint64_t* cyc, uint32_t seed) {
  uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9e3779b9u;
  uint64_t p[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) p[k] = a + k;
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITERS; ++it) {
#define X(k) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(p[k]) : "v"(a), "v"(b) : "vcc");
    REP16(X)
#undef X
  }
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) acc ^= p[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(acc ^ (acc >> 32));
  if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

// mad_u64_u32 with an SGPR carry-out (not vcc) and 32-bit zero-extended addend: the form a
// row of a*b[i]+t[j] uses
__global__ void __launch_bounds__(256) k_mad_u64_u32_sgprcy(uint32_t* out, uint64_t* cyc, uint32_t seed) {
  uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9e3779b9u;
  uint64_t p[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) p[k] = a + k;
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITERS; ++it) {
#define X(k) { uint64_t cy; asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(p[k]), "=s"(cy) : "v"(a), "v"(b)); }
    REP16(X)
#undef X
  }
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) acc ^= p[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(acc ^ (acc >> 32));
  if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

KERNEL_BEGIN(k_mul_lo_u32)
#define X(k) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r0[k]) : "v"(a));
  REP16(X)
#undef X
KERNEL_END

KERNEL_BEGIN(k_mul_hi_u32)
#define X(k) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(r0[k]) : "v"(a));
  REP16(X)
#undef X
KERNEL_END

KERNEL_BEGIN(k_mad_u32_u24)
#define X(k) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(r0[k]) : "v"(a), "v"(b));
  REP16(X)
#undef X
KERNEL_END

KERNEL_BEGIN(k_mul_hi_u32_u24)
#define X(k) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(r0[k]) : "v"(a));
  REP16(X)
#undef X
KERNEL_END

KERNEL_BEGIN(k_mul_u32_u24)
#define X(k) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(r0[k]) : "v"(a));
  REP16(X)
#undef X
KERNEL_END

KERNEL_BEGIN(k_add_co_u32)
#define X(k) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(r0[k]) : "v"(a) : "vcc");
  REP16(X)
#undef X
KERNEL_END

KERNEL_BEGIN(k_addc_co_u32)
#define X(k) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(r0[k]) : "v"(a) : "vcc");
  REP16(X)
#undef X
KERNEL_END

// carry chain through an explicit SGPR pair (VOP3 form)
KERNEL_BEGIN(k_addc_co_u32_sgpr)
  uint64_t cy = 0;
#define X(k) asm volatile("v_addc_co_u32 %0, %1, %0, %2, %1" : "+v"(r0[k]), "+s"(cy) : "v"(a));
  REP16(X)
#undef X
KERNEL_END

KERNEL_BEGIN(k_add3_u32)
#define X(k) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r0[k]) : "v"(a), "v"(b));
  REP16(X)
#undef X
KERNEL_END

KERNEL_BEGIN(k_add_u32)
#define X(k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r0[k]) : "v"(a));
  REP16(X)
#undef X
KERNEL_END

KERNEL_BEGIN(k_lshl_add_u64)
  uint64_t q[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) q[k] = ((uint64_t)r0[k] << 32) | r1[k];
  uint64_t ab = ((uint64_t)a << 32) | b;
#define X(k) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(q[k]) : "v"(ab));
  REP16(X)
#undef X
#pragma unroll
  for (int k = 0; k < 8; ++k) r0[k] ^= (uint32_t)(q[k] ^ (q[k] >> 32));
KERNEL_END

KERNEL_BEGIN(k_alignbit_b32)
#define X(k) asm volatile("v_alignbit_b32 %0, %0, %1, 13" : "+v"(r0[k]) : "v"(a));
  REP16(X)
#undef X
KERNEL_END

KERNEL_BEGIN(k_and_or_b32)
#define X(k) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(r0[k]) : "v"(a), "v"(b));
  REP16(X)
#undef X
KERNEL_END

KERNEL_BEGIN(k_cndmask_b32)
#define X(k) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r0[k]) : "v"(a) : );
  REP16(X)
#undef X
KERNEL_END

// ---- floating point
KERNEL_BEGIN(k_fma_f64)
#define X(k) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[k]) : "v"(da), "v"(db));
  REP16(X)
#undef X
KERNEL_END

KERNEL_BEGIN(k_add_f64)
#define X(k) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[k]) : "v"(da));
  REP16(X)
#undef X
KERNEL_END

KERNEL_BEGIN(k_mul_f64)
#define X(k) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[k]) : "v"(da));
  REP16(X)
#undef X
KERNEL_END

KERNEL_BEGIN(k_fma_f32)
  float f[8]; float fa = 1.0f + a * 1e-9f, fb = 1.0f + b * 1e-10f;
#pragma unroll
  for (int k = 0; k < 8; ++k) f[k] = k + 0.5f;
#define X(k) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f[k]) : "v"(fa), "v"(fb));
  REP16(X)
#undef X
#pragma unroll
  for (int k = 0; k < 8; ++k) r0[k] ^= __float_as_uint(f[k]);
KERNEL_END

KERNEL_BEGIN(k_pk_fma_f32)
#define X(k) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(d[k]) : "v"(da), "v"(db));
  REP16(X)
#undef X
KERNEL_END

KERNEL_BEGIN(k_cvt_f64_u32)
#define X(k) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d[k]) : "v"(r0[k]));
  REP16(X)
#undef X
KERNEL_END

KERNEL_BEGIN(k_cvt_u32_f64)
#define X(k) asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(r0[k]) : "v"(d[k]));
  REP16(X)
#undef X
KERNEL_END

// ---- dot products (integer)
KERNEL_BEGIN(k_dot2_u32_u16)
#define X(k) asm volatile("v_dot2_u32_u16 %0, %1, %2, %0" : "+v"(r0[k]) : "v"(a), "v"(b));
  REP16(X)
#undef X
KERNEL_END

KERNEL_BEGIN(k_dot4_u32_u8)
#define X(k) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(r0[k]) : "v"(a), "v"(b));
  REP16(X)
#undef X
KERNEL_END

KERNEL_BEGIN(k_mad_u16_pk)
#define X(k) asm volatile("v_pk_mad_u16 %0, %1, %2, %0" : "+v"(r0[k]) : "v"(a), "v"(b));
  REP16(X)
#undef X
KERNEL_END

// ---- cross-lane
KERNEL_BEGIN(k_mov_dpp_row_shr)
#define X(k) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1" : "+v"(r0[k]));
  REP16(X)
#undef X
KERNEL_END

KERNEL_BEGIN(k_add_dpp_row_shr)
#define X(k) asm volatile("v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r0[k]) : "v"(r1[k]));
  REP16(X)
#undef X
KERNEL_END

KERNEL_BEGIN(k_mov_dpp_wave_shr)
#define X(k) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r0[k]) : "v"(r1[k]));
  REP16(X)
#undef X
KERNEL_END

KERNEL_BEGIN(k_mov_dpp_row_bcast15)
#define X(k) asm volatile("v_mov_b32_dpp %0, %1 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(r0[k]) : "v"(r1[k]));
  REP16(X)
#undef X
KERNEL_END

KERNEL_BEGIN(k_ds_bpermute)
  uint32_t addr = ((threadIdx.x + 1) & 63) * 4;
#define X(k) asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(r0[k]) : "v"(addr));
  REP16(X)
#undef X
KERNEL_END

KERNEL_BEGIN(k_ds_bpermute_pipelined)
  uint32_t addr = ((threadIdx.x + 1) & 63) * 4;
#define X(k) asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(r0[k]) : "v"(addr), "v"(r1[k]));
  REP16(X)
#undef X
  asm volatile("s_waitcnt lgkmcnt(0)");
KERNEL_END

KERNEL_BEGIN(k_ds_swizzle)
#define X(k) asm volatile("ds_swizzle_b32 %0, %1 offset:swizzle(BROADCAST,8,0)" : "=v"(r0[k]) : "v"(r1[k]));
  REP16(X)
#undef X
  asm volatile("s_waitcnt lgkmcnt(0)");
KERNEL_END

KERNEL_BEGIN(k_readlane)
  uint32_t s[8];
#define X(k) asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(s[k]) : "v"(r1[k]));
  REP16(X)
#undef X
#pragma unroll
  for (int k = 0; k < 8; ++k) r0[k] ^= s[k];
KERNEL_END

KERNEL_BEGIN(k_readlane_then_valu)
  uint32_t s[8];
#define X(k) asm volatile("v_readlane_b32 %0, %1, 5\n v_add_u32 %1, %0, %1" : "=&s"(s[k]), "+v"(r1[k]));
  REP16(X)
#undef X
KERNEL_END

KERNEL_BEGIN(k_permlane32_swap)
#define X(k) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(r0[k]), "+v"(r1[k]));
  REP16(X)
#undef X
KERNEL_END

// ---- LDS
__global__ void __launch_bounds__(256) k_ds_read_b32(uint32_t* out, uint64_t* cyc, uint32_t seed) {
  __shared__ uint32_t lds[256 * 8];
  for (int k = 0; k < 8; ++k) lds[k * 256 + threadIdx.x] = seed + k;
  __syncthreads();
  uint32_t r0[8];
  uint32_t addr = threadIdx.x * 4;
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITERS; ++it) {
#define X(k) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r0[k]) : "v"(addr), "i"(k * 1024));
    REP16(X)
#undef X
    asm volatile("s_waitcnt lgkmcnt(0)");
  }
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  uint32_t acc = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) acc ^= r0[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

// ---- a dependent chain of mad_u64_u32 (latency)
__global__ void __launch_bounds__(256) k_mad_u64_u32_dep(uint32_t* out, uint64_t* cyc, uint32_t seed) {
  uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9e3779b9u;
  uint64_t p = a;
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITERS; ++it) {
#define X(k) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(p) : "v"(a), "v"(b) : "vcc");
    REP16(X)
#undef X
  }
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(p ^ (p >> 32));
  if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

__global__ void __launch_bounds__(256) k_fma_f64_dep(uint32_t* out, uint64_t* cyc, uint32_t seed) {
  uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9e3779b9u;
  double da = 1.0 + a * 1e-9, db = 1.0 + b * 1e-10, d = 0.5;
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITERS; ++it) {
#define X(k) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d) : "v"(da), "v"(db));
    REP16(X)
#undef X
  }
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)__double2ll_rn(d);
  if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

// mixed: mad_u64_u32 interleaved 1:1 with v_addc (the CIOS row shape)
__global__ void __launch_bounds__(256) k_mix_mad_addc(uint32_t* out, uint64_t* cyc, uint32_t seed) {
  uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9e3779b9u;
  uint64_t p[8]; uint32_t r0[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { p[k] = a + k; r0[k] = b + k; }
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITERS; ++it) {
#define X(k) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_addc_co_u32 %1, vcc, %1, %2, vcc" : "+v"(p[k]), "+v"(r0[k]) : "v"(a), "v"(b) : "vcc");
    REP16(X)
#undef X
  }
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) acc ^= p[k] ^ r0[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(acc ^ (acc >> 32));
  if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

// mixed: fma_f64 interleaved with integer 64-bit adds (the DFMA-limb shape: 2 fma + 1 add_f64 + 2 u64 adds)
__global__ void __launch_bounds__(256) k_mix_dfma_limb(uint32_t* out, uint64_t* cyc, uint32_t seed) {
  uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9e3779b9u;
  double da = 1.0 + a * 1e-9, db = 1.0 + b * 1e-10;
  double hi[8], lo[8]; uint64_t s0[8], s1[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { hi[k] = k; lo[k] = k; s0[k] = k; s1[k] = k; }
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITERS; ++it) {
#define X(k) asm volatile( \
      "v_fma_f64 %0, %4, %5, %0\n" \
      "v_add_f64 %1, %0, %1\n" \
      "v_fma_f64 %1, %4, %5, %1\n" \
      "v_lshl_add_u64 %2, %0, 0, %2\n" \
      "v_lshl_add_u64 %3, %1, 0, %3\n" \
      : "+v"(hi[k]), "+v"(lo[k]), "+v"(s0[k]), "+v"(s1[k]) : "v"(da), "v"(db));
    REP16(X)
#undef X
  }
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) acc ^= s0[k] ^ s1[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(acc ^ (acc >> 32));
  if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

typedef void (*kern_t)(uint32_t*, uint64_t*, uint32_t);
struct Bench { const char* name; kern_t k; int insts_per_body; };

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  printf("device: %s  CUs=%d  clock=%d kHz  wallclock=%d kHz\n", prop.gcnArchName, cus, prop.clockRate, 0);
  std::vector<Bench> benches = {
    {"v_mad_u64_u32 (vcc)", k_mad_u64_u32, 16},
    {"v_mad_u64_u32 (sgpr cy)", k_mad_u64_u32_sgprcy, 16},
    {"v_mad_u64_u32 dependent", k_mad_u64_u32_dep, 16},
    {"v_mul_lo_u32", k_mul_lo_u32, 16},
    {"v_mul_hi_u32", k_mul_hi_u32, 16},
    {"v_mad_u32_u24", k_mad_u32_u24, 16},
    {"v_mul_u32_u24", k_mul_u32_u24, 16},
    {"v_mul_hi_u32_u24", k_mul_hi_u32_u24, 16},
    {"v_add_u32", k_add_u32, 16},
    {"v_add_co_u32", k_add_co_u32, 16},
    {"v_addc_co_u32 (vcc)", k_addc_co_u32, 16},
    {"v_addc_co_u32 (sgpr)", k_addc_co_u32_sgpr, 16},
    {"v_add3_u32", k_add3_u32, 16},
    {"v_lshl_add_u64", k_lshl_add_u64, 16},
    {"v_alignbit_b32", k_alignbit_b32, 16},
    {"v_and_or_b32", k_and_or_b32, 16},
    {"v_cndmask_b32", k_cndmask_b32, 16},
    {"v_fma_f64", k_fma_f64, 16},
    {"v_fma_f64 dependent", k_fma_f64_dep, 16},
    {"v_add_f64", k_add_f64, 16},
    {"v_mul_f64", k_mul_f64, 16},
    {"v_fma_f32", k_fma_f32, 16},
    {"v_pk_fma_f32", k_pk_fma_f32, 16},
    {"v_cvt_f64_u32", k_cvt_f64_u32, 16},
    {"v_cvt_u32_f64", k_cvt_u32_f64, 16},
    {"v_dot2_u32_u16", k_dot2_u32_u16, 16},
    {"v_dot4_u32_u8", k_dot4_u32_u8, 16},
    {"v_pk_mad_u16", k_mad_u16_pk, 16},
    {"v_mov_b32_dpp row_shr (+s_nop1)", k_mov_dpp_row_shr, 16},
    {"v_add_u32_dpp row_shr", k_add_dpp_row_shr, 16},
    {"v_mov_b32_dpp wave_shr", k_mov_dpp_wave_shr, 16},
    {"v_mov_b32_dpp row_bcast15", k_mov_dpp_row_bcast15, 16},
    {"ds_bpermute_b32 (serialized)", k_ds_bpermute, 16},
    {"ds_bpermute_b32 (pipelined)", k_ds_bpermute_pipelined, 16},
    {"ds_swizzle_b32", k_ds_swizzle, 16},
    {"ds_read_b32", k_ds_read_b32, 16},
    {"v_readlane_b32", k_readlane, 16},
    {"v_readlane_b32+v_add", k_readlane_then_valu, 32},
    {"v_permlane32_swap", k_permlane32_swap, 16},
    {"mix: mad_u64_u32 + addc", k_mix_mad_addc, 32},
    {"mix: 2 fma_f64 + add_f64 + 2 lshl_add_u64", k_mix_dfma_limb, 80},
  };
  uint32_t* d_out; uint64_t* d_cyc;
  int max_threads = cus * 4 * 4 * 64;
  CHECK(hipMalloc(&d_out, max_threads * sizeof(uint32_t)));
  CHECK(hipMalloc(&d_cyc, (max_threads / 64) * sizeof(uint64_t)));
  std::vector<uint64_t> h_cyc(max_threads / 64);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  printf("%-44s %10s %10s %10s   (cycles per wave-instruction per SIMD; s_memtime ticks; wall-derived GHz)\n",
         "instruction", "1 w/SIMD", "2 w/SIMD", "4 w/SIMD");
  for (auto& b : benches) {
    printf("%-44s", b.name);
    for (int wps : {1, 2, 4}) {
      int blocks = cus * wps;  // 256 threads = 4 waves = one per SIMD
      hipLaunchKernelGGL(b.k, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, 1u);  // warm
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(b.k, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, 2u);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      CHECK(hipMemcpy(h_cyc.data(), d_cyc, blocks * 4 * sizeof(uint64_t), hipMemcpyDeviceToHost));
      double sum = 0; for (int i = 0; i < blocks * 4; ++i) sum += (double)h_cyc[i];
      double avg_ticks = sum / (blocks * 4);
      double insts = (double)ITERS * b.insts_per_body;
      // s_memtime counts at a fixed 100 MHz on gfx9? report both tick-based and wall-based
      double per_inst_ticks = avg_ticks / insts * wps;   // wps waves share a SIMD
      double ns_per_inst = (ms * 1e6) / insts * 1.0;       // wall ns for one wave's instruction stream
      printf("  %6.2f/%5.2fns", per_inst_ticks, ns_per_inst / wps);
    }
    printf("\n");
  }
  return 0;
}
