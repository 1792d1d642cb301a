// Test helper (built by `make examples` into tests/_build/ec_quad_unit, run by tests/test_gpu_ec_fd.py): the quad-lane point
// addition and negation of mpvss_rs_amd/csrc/ec_quad.h against the one-lane complete formulas of ec_curves.h, one pair of
// points per quad:  P = a G, Q = +-b G for 64-bit a, b from the input (a = 0 / b = 0: the identity; b = a: a doubling;
// b = -a: P + (-P)), results compared projectively in the kernel.
//   ec_quad_unit <1 secp256k1 quads | 2 ristretto255 quads | 3 secp256k1 eight lanes> <in.bin> <n>        in.bin: n x (a, b, flags) u64; flags bit 0: negate Q
// prints "bad <count>" and the first failing indices; exit code 0 when every pair agrees
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "../mpvss_rs_amd/csrc/ec_quad.h"

using namespace ec;

#define CHECK(x)                                                                       \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } \
  } while (0)

template <class C>
__device__ bool same_point(const typename C::Point& a, const typename C::Point& b) {
  typedef typename C::Fp Fp;
  Fe l, r;
  Fp::mul(l, a.X, b.Z);
  Fp::mul(r, b.X, a.Z);
  bool ok = Fp::equal(l, r);
  Fp::mul(l, a.Y, b.Z);
  Fp::mul(r, b.Y, a.Z);
  ok = ok && Fp::equal(l, r);
  ok = ok && (Fp::is_zero(a.Z) == Fp::is_zero(b.Z));
  if constexpr (C::POINT_WORDS == 40) {
    Fp::mul(l, a.T, b.Z);
    Fp::mul(r, b.T, a.Z);
    ok = ok && Fp::equal(l, r);
  }
  return ok;
}

template <class C>
__device__ void store_pt(u32* dst, const typename C::Point& p) {
  const u32* w = reinterpret_cast<const u32*>(&p);
  for (int i = 0; i < C::POINT_WORDS; ++i) dst[i] = w[i];
}
template <class C>
__device__ void load_pt(typename C::Point& p, const u32* src) {
  u32* w = reinterpret_cast<u32*>(&p);
  for (int i = 0; i < C::POINT_WORDS; ++i) w[i] = src[i];
}

template <class Q>
__global__ void __launch_bounds__(64) k_unit(const uint64_t* __restrict__ in, int n, int* __restrict__ bad) {
  typedef typename Q::C C;
  extern __shared__ u32 lds[];
  constexpr int PW = C::POINT_WORDS;
  const int gl = blockIdx.x * 64 + threadIdx.x, qi = gl / Q::LANES, role = gl % Q::LANES;
  const int i = qi < n ? qi : n - 1;
  const uint64_t a = in[3 * i], b = in[3 * i + 1], flags = in[3 * i + 2];
  typename C::Point g, P, Qp, t;
  C::generator(g);
  small_scalar_mul<C>(P, g, a, 64);
  small_scalar_mul<C>(t, g, b, 64);
  if (flags & 1) C::neg(Qp, t); else Qp = t;
  // the four points of a quad (P, Q, P + Q, -P + Q) travel between its lanes through LDS
  u32* mine = lds + Q::LDS_WORDS + (threadIdx.x / Q::LANES) * 6 * PW;
  if (role == 0) {
    store_pt<C>(mine, P);
    store_pt<C>(mine + PW, Qp);
  }
  __syncthreads();
  typename Q::St D, nb;
  Q::load(D, mine, role);
  Q::load(nb, mine + PW, role);
  Q::add(D, nb, role, lds);
  Fe o;
  Q::out_words(o, D, role);
  if (Q::out_lane(role))
    for (int k = 0; k < 10; ++k) mine[2 * PW + Q::out_offset(role) + k] = o.v[k];
  Q::load(D, mine, role);
  Q::neg(D, role);
  Q::add(D, nb, role, lds);
  Q::out_words(o, D, role);
  if (Q::out_lane(role))
    for (int k = 0; k < 10; ++k) mine[3 * PW + Q::out_offset(role) + k] = o.v[k];
  Q::load(D, mine, role);
  Q::add(D, nb, role, lds);
  Q::out_words(o, D, role);
  if (Q::out_lane(role))
    for (int k = 0; k < 10; ++k) mine[4 * PW + Q::out_offset(role) + k] = o.v[k];
  Q::load(D, mine, role);
  Q::neg(D, role);
  Q::out_words(o, D, role);
  if (Q::out_lane(role))
    for (int k = 0; k < 10; ++k) mine[5 * PW + Q::out_offset(role) + k] = o.v[k];
  __syncthreads();
  // -P through the one-lane negation and LDS, then the quad addition
  if (role == 0) { typename C::Point np2; C::neg(np2, P); store_pt<C>(mine, np2); }
  __syncthreads();
  Q::load(D, mine, role);
  Q::add(D, nb, role, lds);
  Q::out_words(o, D, role);
  __syncthreads();
  if (Q::out_lane(role))
    for (int k = 0; k < 10; ++k) mine[Q::out_offset(role) + k] = o.v[k];
  __syncthreads();
  if (role == 0 && qi < n) {
    typename C::Point r1, r2, e1, e2, np;
    load_pt<C>(r1, mine + 2 * PW);
    load_pt<C>(r2, mine + 3 * PW);
    C::add(e1, P, Qp);
    C::neg(np, P);
    C::add(e2, np, Qp);
    int code = (same_point<C>(r1, e1) ? 0 : 1) | (same_point<C>(r2, e2) ? 0 : 2);
    load_pt<C>(r1, mine + 4 * PW);
    load_pt<C>(r2, mine + 5 * PW);
    code |= (same_point<C>(r1, e1) ? 0 : 4) | (same_point<C>(r2, np) ? 0 : 8);
    load_pt<C>(r1, mine);
    code |= same_point<C>(r1, e2) ? 0 : 16;
    if (code) {
      const int slot = atomicAdd(bad, 1);
      if (slot < 15) { bad[1 + 2 * slot] = qi; bad[2 + 2 * slot] = code; }
    }
  }
}

int main(int argc, char** argv) {
  if (argc != 4) { fprintf(stderr, "usage: ec_quad_unit <group> <in.bin> <n>\n"); return 2; }
  const int group = atoi(argv[1]), n = atoi(argv[3]);
  std::vector<uint64_t> in((size_t)n * 3);
  FILE* f = fopen(argv[2], "rb");
  if (!f || fread(in.data(), 8, in.size(), f) != in.size()) { fprintf(stderr, "cannot read %s\n", argv[2]); return 2; }
  fclose(f);
  uint64_t* din;
  int* bad;
  CHECK(hipMalloc(&din, in.size() * 8));
  CHECK(hipMalloc(&bad, 32 * 4));
  CHECK(hipMemcpy(din, in.data(), in.size() * 8, hipMemcpyHostToDevice));
  CHECK(hipMemset(bad, 0, 32 * 4));
  // group 1: secp256k1 by quads, 2: ristretto255 by quads, 3: secp256k1 by eight lanes (six products side by side)
  if (group == 1)
    hipLaunchKernelGGL(k_unit<QuadSecp>, dim3((n + 15) / 16), dim3(64), (QuadSecp::LDS_WORDS + 16 * 6 * 30) * 4, 0, din, n, bad);
  else if (group == 2)
    hipLaunchKernelGGL(k_unit<QuadRist>, dim3((n + 15) / 16), dim3(64), (QuadRist::LDS_WORDS + 16 * 6 * 40) * 4, 0, din, n, bad);
  else
    hipLaunchKernelGGL(k_unit<OctSecp>, dim3((n + 7) / 8), dim3(64), (OctSecp::LDS_WORDS + 8 * 6 * 30) * 4, 0, din, n, bad);
  CHECK(hipDeviceSynchronize());
  int h[32];
  CHECK(hipMemcpy(h, bad, sizeof h, hipMemcpyDeviceToHost));
  printf("bad %d\n", h[0]);
  for (int i = 0; i < h[0] && i < 15; ++i) printf("  pair %d code %d\n", h[1 + 2 * i], h[2 + 2 * i]);
  return h[0] == 0 ? 0 : 1;
}
