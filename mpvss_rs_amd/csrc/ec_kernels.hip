// secp256k1 / ristretto255 batch kernels for gfx950: one share per lane, 64-lane workgroups.
//   reference: src/groups/secp256k1.rs:91-107, src/groups/ristretto255.rs:161-177 (exp / mul),
//              src/participant.rs:1404-1417, 1847-1860 (X_i = sum_j i^j C_j), src/dleq.rs:66-84.
// All group elements cross the kernel boundary in the reference's canonical encodings (33-byte SEC1
// compressed / 32-byte ristretto255), scalars as 32 bytes (big-endian / little-endian).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ec_curves.h"
#include "ec_kernels.h"

using namespace ec;

namespace {

template <class C>
__device__ __forceinline__ void load_point_aos(typename C::Point& p, const u32* __restrict__ src) {
  u32* w = reinterpret_cast<u32*>(&p);
#pragma unroll
  for (int i = 0; i < C::POINT_WORDS; ++i) w[i] = src[i];
}
template <class C>
__device__ __forceinline__ void store_point_aos(u32* __restrict__ dst, const typename C::Point& p) {
  const u32* w = reinterpret_cast<const u32*>(&p);
#pragma unroll
  for (int i = 0; i < C::POINT_WORDS; ++i) dst[i] = w[i];
}

// encoded points -> decoded points (array of structs), ok flags
template <class C>
__device__ __forceinline__ void decode_body(const uint8_t* __restrict__ enc, int count, u32* __restrict__ pts,
                                            uint8_t* __restrict__ ok) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= count) return;
  typename C::Point p;
  const bool good = C::decode(p, enc + (size_t)x * C::ENC_LEN);
  store_point_aos<C>(pts + (size_t)x * C::POINT_WORDS, p);
  ok[x] = good ? 1 : 0;
}

// X_i = sum_j (i^j) C_j by Horner's rule:  X = ((C_{t-1} * i + C_{t-2}) * i + ...) * i + C_0.
// Same group element as the reference's loop (participant.rs:1411-1417 reduces i^j mod the group
// order, which is the order of every element of these prime-order groups).
template <class C>
__device__ __forceinline__ void commit_eval_body(const u32* __restrict__ cm, int t, const int64_t* __restrict__ positions,
                                                 int count, uint8_t* __restrict__ x_enc) {
  const int xi = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = xi < count;
  const int x = live ? xi : count - 1;
  const uint64_t pos = (uint64_t)positions[x];
  int nb = (pos == 0) ? 0 : 64 - __builtin_clzll(pos);
  typename C::Point acc, cj, r;
  load_point_aos<C>(acc, cm + (size_t)(t - 1) * C::POINT_WORDS);
  for (int j = t - 2; j >= 0; --j) {
    small_scalar_mul<C>(r, acc, pos, nb);
    load_point_aos<C>(cj, cm + (size_t)j * C::POINT_WORDS);
    C::add(acc, r, cj);
  }
  if (live) C::encode(x_enc + (size_t)x * C::ENC_LEN, acc);
}

// out = k1 * P1 + k2 * P2.  p1_stride / k2_stride 0 = shared operand; p2 == nullptr: out = k1 * P1.
template <class C>
__device__ __forceinline__ void dual_mul_body(const uint8_t* __restrict__ p1_enc, size_t p1_stride,
                                              const uint8_t* __restrict__ k1, const uint8_t* __restrict__ p2_enc,
                                              const uint8_t* __restrict__ k2, size_t k2_stride, int count,
                                              uint8_t* __restrict__ out_enc, uint8_t* __restrict__ ok) {
  const int xi = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = xi < count;
  const int x = live ? xi : count - 1;
  typename C::Point a, b, r;
  bool good = C::decode(a, p1_enc + (size_t)x * p1_stride);
  if (p2_enc) good = C::decode(b, p2_enc + (size_t)x * C::ENC_LEN) && good; else C::identity(b);
  dual_mul<C>(r, a, k1 + (size_t)x * 32, b, p2_enc ? k2 + (size_t)x * k2_stride : nullptr);
  if (live) {
    C::encode(out_enc + (size_t)x * C::ENC_LEN, r);
    ok[x] = good ? 1 : 0;
  }
}

template <class C>
__device__ __forceinline__ void add_body(const uint8_t* __restrict__ a_enc, const uint8_t* __restrict__ b_enc, int count,
                                         uint8_t* __restrict__ out_enc, uint8_t* __restrict__ ok) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= count) return;
  typename C::Point a, b, r;
  const bool g1 = C::decode(a, a_enc + (size_t)x * C::ENC_LEN);
  const bool g2 = C::decode(b, b_enc + (size_t)x * C::ENC_LEN);
  C::add(r, a, b);
  C::encode(out_enc + (size_t)x * C::ENC_LEN, r);
  ok[x] = (g1 && g2) ? 1 : 0;
}

}  // namespace

#define EC_KERNELS(NAME, CURVE)                                                                                        \
  extern "C" __global__ void __launch_bounds__(64) k_##NAME##_decode(const uint8_t* enc, int count, u32* pts,          \
                                                                     uint8_t* ok) {                                    \
    decode_body<CURVE>(enc, count, pts, ok);                                                                           \
  }                                                                                                                    \
  extern "C" __global__ void __launch_bounds__(64) k_##NAME##_commit_eval(const u32* cm, int t, const int64_t* pos,    \
                                                                          int count, uint8_t* x_enc) {                 \
    commit_eval_body<CURVE>(cm, t, pos, count, x_enc);                                                                 \
  }                                                                                                                    \
  extern "C" __global__ void __launch_bounds__(64) k_##NAME##_dual_mul(                                                \
      const uint8_t* p1, size_t p1_stride, const uint8_t* k1, const uint8_t* p2, const uint8_t* k2, size_t k2_stride,  \
      int count, uint8_t* out, uint8_t* ok) {                                                                          \
    dual_mul_body<CURVE>(p1, p1_stride, k1, p2, k2, k2_stride, count, out, ok);                                        \
  }                                                                                                                    \
  extern "C" __global__ void __launch_bounds__(64) k_##NAME##_add(const uint8_t* a, const uint8_t* b, int count,       \
                                                                  uint8_t* out, uint8_t* ok) {                         \
    add_body<CURVE>(a, b, count, out, ok);                                                                             \
  }

EC_KERNELS(secp, Secp)
EC_KERNELS(rist, Ristretto)

// ---- launchers ---------------------------------------------------------------------------------------
static inline int blocks_for(int count) { return (count + 63) / 64; }

extern "C" int ec_point_words(int group) { return group == 1 ? Secp::POINT_WORDS : Ristretto::POINT_WORDS; }

extern "C" int ec_launch_decode(int group, const uint8_t* enc, int count, uint32_t* pts, uint8_t* ok, hipStream_t s) {
  if (count <= 0) return 0;
  if (group == 1) hipLaunchKernelGGL(k_secp_decode, dim3(blocks_for(count)), dim3(64), 0, s, enc, count, pts, ok);
  else hipLaunchKernelGGL(k_rist_decode, dim3(blocks_for(count)), dim3(64), 0, s, enc, count, pts, ok);
  return (int)hipGetLastError();
}
extern "C" int ec_launch_commit_eval(int group, const uint32_t* cm, int t, const int64_t* positions, int count,
                                     uint8_t* x_enc, hipStream_t s) {
  if (count <= 0) return 0;
  if (group == 1) hipLaunchKernelGGL(k_secp_commit_eval, dim3(blocks_for(count)), dim3(64), 0, s, cm, t, positions, count, x_enc);
  else hipLaunchKernelGGL(k_rist_commit_eval, dim3(blocks_for(count)), dim3(64), 0, s, cm, t, positions, count, x_enc);
  return (int)hipGetLastError();
}
extern "C" int ec_launch_dual_mul(int group, const uint8_t* p1, size_t p1_stride, const uint8_t* k1, const uint8_t* p2,
                                  const uint8_t* k2, size_t k2_stride, int count, uint8_t* out, uint8_t* ok,
                                  hipStream_t s) {
  if (count <= 0) return 0;
  if (group == 1)
    hipLaunchKernelGGL(k_secp_dual_mul, dim3(blocks_for(count)), dim3(64), 0, s, p1, p1_stride, k1, p2, k2, k2_stride, count, out, ok);
  else
    hipLaunchKernelGGL(k_rist_dual_mul, dim3(blocks_for(count)), dim3(64), 0, s, p1, p1_stride, k1, p2, k2, k2_stride, count, out, ok);
  return (int)hipGetLastError();
}
extern "C" int ec_launch_add(int group, const uint8_t* a, const uint8_t* b, int count, uint8_t* out, uint8_t* ok,
                             hipStream_t s) {
  if (count <= 0) return 0;
  if (group == 1) hipLaunchKernelGGL(k_secp_add, dim3(blocks_for(count)), dim3(64), 0, s, a, b, count, out, ok);
  else hipLaunchKernelGGL(k_rist_add, dim3(blocks_for(count)), dim3(64), 0, s, a, b, count, out, ok);
  return (int)hipGetLastError();
}
