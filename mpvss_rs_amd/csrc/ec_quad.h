// Point addition of the curve groups by the FOUR LANES OF A QUAD (device only): the stepping recurrence of the forward
// differences, D_k <- D_k + D_{k+1}, is a serial chain of complete additions -- 12 field products for secp256k1, 9 for
// ristretto255, one after the other in one lane -- and a lone box's latency is that chain (tools/ec_lone_box_trace.sh).
// Here the independent products of one addition run side by side in the lanes of a quad and the lanes trade their
// results by DPP quad permutations / a few LDS words; the same group elements come out (canonical encodings are what
// is compared), in a third of the sequential depth.
//   reference: src/participant.rs:1404-1430 (X_i = sum_j i^j C_j), src/groups/secp256k1.rs:86-100, ristretto255.rs:150-170
//
// ristretto255 (unified extended Edwards addition, a = -1; the 4-way split of the vectorised dalek backends):
//   lane i keeps the FORM u_i of its point:  u = (Y - X, Y + X, T, Z)
//   v = u(P) * u(Q);  w = v * kappa,  kappa = (1, 1, 2d, 2)               -> (A, B, C, D)
//   e = even lane ? partner - w : partner + w   (partner = lane ^ 1)        -> (E, H, F, G)
//   r = a * b,  a = quad_perm[0,1,0,2](e), b = quad_perm[2,3,1,3](e)       -> (X3, Y3, T3, Z3)
//   u' = lanes 0, 1: partner(r) -/+ r;  lanes 2, 3: r                      -> (Y3 - X3, Y3 + X3, T3, Z3)
//   three products deep instead of nine.
// secp256k1 (Renes-Costello-Batina complete addition, a = 0, b3 = 21), lanes 0..2 active, lane 3 mirrors lane 0:
//   lane i keeps a = c_i and b = c_{i+1} of (c_0, c_1, c_2) = (X, Y, Z)
//   t = a(P) a(Q);  s = (a + b)(P) (a + b)(Q);  cr = s - t - t_{i+1}       -> (X1Y2+X2Y1, Y1Z2+Y2Z1, Z1X2+Z2X1)
//   m = (3, 1, 21) t;  crk = (1, 1, 21) cr;  zz = m + m_{i+1};  mm = m - m_{i+1}     (lane 1: z3 = t1 + 21 t2, t1' = t1 - 21 t2)
//   published to LDS: S0 = crk, S1 = (m, mm, -crk)[lane], S2 = zz
//   r = A B + C D with one reduction:  lane 0: X3 = S0[0] S1[1] + S0[1] S1[2]
//                                      lane 1: Y3 = S1[1] S2[1] + S0[2] S1[0]
//                                      lane 2: Z3 = S2[1] S0[1] + S1[0] S0[0]
//   two products and a double product deep instead of twelve products.
// CROSS-LANE MOVES AND THE OPTIMISER.  hipcc treats __builtin_amdgcn_mov_dpp as if it were lane-local: with a lane-dependent
// `if (lane ...) x = y` ahead of it the moves were sunk into the divergent region (inactive lanes read as 0), and with a
// lane-dependent select feeding it the move was distributed over the select's arms (each lane then picks by ITS OWN condition
// what another lane computed) -- both came out as points off the curve in tests/ec_quad_unit.hip, the second only when two
// additions were inlined one after the other.  So (a) everything here is branch-free, lane-dependent choices are selects, and
// (b) the operand of every cross-lane move passes through an empty `asm volatile` first, which makes it opaque.
// tests: tests/ec_quad_unit.hip through tests/test_gpu_ec_fd.py (addition, negation + addition against the one-lane formulas:
// doublings, P + (-P), the identity on either side); the stepping and table pipelines against Horner's rule and the oracle.
#pragma once
#include "ec_curves.h"

namespace ec {

// quad_perm selectors: dpp_ctrl = p0 | p1<<2 | p2<<4 | p3<<6
template <int P0, int P1, int P2, int P3>
__device__ __forceinline__ void fe_quad_perm(Fe& r, const Fe& a) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    u32 x = a.v[i];
    asm volatile("" : "+v"(x));        // opaque to the optimiser: see the note on lane-dependent selects above
    r.v[i] = (u32)__builtin_amdgcn_mov_dpp((int)x, P0 | (P1 << 2) | (P2 << 4) | (P3 << 6), 0xf, 0xf, true);
  }
}
__device__ __forceinline__ void fe_select(Fe& r, const Fe& a, const Fe& b, bool take_a) {
#pragma unroll
  for (int i = 0; i < 10; ++i) r.v[i] = take_a ? a.v[i] : b.v[i];
}
// the same word of the lane four places up (the quad of the next level); lanes 60..63 get their own
__device__ __forceinline__ void fe_from_next_quad(Fe& r, const Fe& a) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    u32 x = a.v[i];
    asm volatile("" : "+v"(x));
    r.v[i] = (u32)__shfl_down((int)x, 4);
  }
}

struct QuadRist {
  typedef Ristretto C;
  typedef Ristretto::Fp Fp;
  static constexpr int LDS_WORDS = 0;
  struct St {
    Fe u;
  };
  typedef St Nb;
  // word offset of this lane's coordinate inside a Point {X, Y, Z, T}; forms (Y-X, Y+X, T, Z)
  __device__ static void load(St& s, const u32* __restrict__ pt, int role) {
    Fe x, y;
    const int own = role == 2 ? 30 : 20;                   // T / Z for lanes 2, 3
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      x.v[i] = pt[(role < 2 ? 0 : own) + i];
      y.v[i] = pt[(role < 2 ? 10 : own) + i];
    }
    Fe d, sum;
    Fp::sub(d, y, x);
    Fp::addc(sum, y, x);
    fe_select(s.u, role == 0 ? d : sum, x, role < 2);
  }
  __device__ static void identity(St& s, int role) {        // (0, 1, 1, 0): forms (1, 1, 0, 1)
    Fp::zero(s.u);
    s.u.v[0] = role == 2 ? 0u : 1u;
  }
  __device__ static const Fe& primary(const St& s) { return s.u; }
  __device__ static void select(St& r, const St& a, bool take_a) { fe_select(r.u, a.u, r.u, take_a); }
  __device__ static void nb_from_primary(Nb& q, const Fe& prim, int) { q.u = prim; }
  __device__ static void neg(St& s, int role) {            // (-X, Y, Z, -T): lanes 0, 1 trade places, lane 2 changes sign
    Fe sw, n;
    fe_quad_perm<1, 0, 2, 3>(sw, s.u);
    Fp::neg(n, s.u);
    fe_select(s.u, n, sw, role == 2);
  }
  __device__ static void add(St& s, const Nb& q, int role, u32*) {
    const Fe d2 = {EC_ED_2D_INIT};
    Fe kappa, v, w, partner, np, e, a, b, r;
    Fp::zero(kappa);
    kappa.v[0] = role == 3 ? 2u : 1u;
    fe_select(kappa, d2, kappa, role == 2);
    Fp::mul(v, s.u, q.u);
    Fp::mul(w, v, kappa);
    fe_quad_perm<1, 0, 3, 2>(partner, w);
    // even lanes: partner - w; odd lanes: partner + w   (one carry for both)
#pragma unroll
    for (int i = 0; i < 10; ++i) np.v[i] = (role & 1) ? w.v[i] : PrimeConsts<PrimeEd>::subpad(i) - w.v[i];
    Fp::addc(e, partner, np);
    fe_quad_perm<0, 1, 0, 2>(a, e);
    fe_quad_perm<2, 3, 1, 3>(b, e);
    Fp::mul(r, a, b);
    fe_quad_perm<1, 0, 3, 2>(partner, r);
#pragma unroll
    for (int i = 0; i < 10; ++i) np.v[i] = (role & 1) ? r.v[i] : PrimeConsts<PrimeEd>::subpad(i) - r.v[i];
    Fp::addc(e, partner, np);
    fe_select(s.u, e, r, role < 2);
  }
  // this lane's ten words of the point in internal coordinates {X, Y, Z, T} -- twice every coordinate, the same point
  __device__ static int out_offset(int role) { return role == 0 ? 0 : role == 1 ? 10 : role == 2 ? 30 : 20; }
  __device__ static bool out_lane(int) { return true; }
  __device__ static void out_words(Fe& o, const St& s, int role) {
    Fe partner, np, e, twice;
    fe_quad_perm<1, 0, 3, 2>(partner, s.u);
    // lane 0: u1 - u0 = 2X; lane 1: u0 + u1 = 2Y
#pragma unroll
    for (int i = 0; i < 10; ++i) np.v[i] = (role & 1) ? s.u.v[i] : PrimeConsts<PrimeEd>::subpad(i) - s.u.v[i];
    Fp::addc(e, partner, np);
    Fp::addc(twice, s.u, s.u);
    fe_select(o, e, twice, role < 2);
  }
};

struct QuadSecp {
  typedef Secp C;
  typedef Secp::Fp Fp;
  static constexpr int LDS_WORDS = 3 * 10 * 64;             // [slot][word][lane]
  struct St {
    Fe a, b;
  };
  typedef St Nb;
  __device__ static int coord(int role) { return role == 3 ? 0 : role; }
  __device__ static void load(St& s, const u32* __restrict__ pt, int role) {
    const int c = coord(role), cn = c == 2 ? 0 : c + 1;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      s.a.v[i] = pt[c * 10 + i];
      s.b.v[i] = pt[cn * 10 + i];
    }
  }
  __device__ static void identity(St& s, int role) {        // (0 : 1 : 0)
    const int c = coord(role);
    Fp::zero(s.a);
    Fp::zero(s.b);
    s.a.v[0] = c == 1 ? 1u : 0u;
    s.b.v[0] = c == 0 ? 1u : 0u;
  }
  __device__ static const Fe& primary(const St& s) { return s.a; }
  __device__ static void select(St& r, const St& a, bool take_a) {
    fe_select(r.a, a.a, r.a, take_a);
    fe_select(r.b, a.b, r.b, take_a);
  }
  __device__ static void nb_from_primary(Nb& q, const Fe& prim, int) {
    q.a = prim;
    fe_quad_perm<1, 2, 0, 1>(q.b, prim);
  }
  __device__ static void neg(St& s, int role) {            // (X, -Y, Z)
    const int c = coord(role);
    Fe n;
    Fp::neg(n, s.a);
    fe_select(s.a, n, s.a, c == 1);
    fe_quad_perm<1, 2, 0, 1>(s.b, s.a);
  }
  __device__ static void add(St& s, const Nb& q, int role, u32* lds) {
    const int c = coord(role), lane = threadIdx.x & 63;
    Fe t, sp, sq, sm, tn, u, cr, m, crk, mn, zz, mm, ncrk, s1;
    Fp::mul(t, s.a, q.a);
    Fp::add(sp, s.a, s.b);
    Fp::add(sq, q.a, q.b);
    Fp::mul(sm, sp, sq);
    fe_quad_perm<1, 2, 0, 1>(tn, t);
    Fp::add(u, t, tn);
#pragma unroll
    for (int i = 0; i < 10; ++i) cr.v[i] = sm.v[i] + PrimeConsts<PrimeSecp>::subpad(i) - u.v[i];    // (carried by mul_small below; < 2^30)
    Fp::mul_small(m, t, c == 0 ? 3u : c == 1 ? 1u : 21u);
    Fp::mul_small(crk, cr, c == 2 ? 21u : 1u);
    fe_quad_perm<1, 2, 0, 1>(mn, m);
    Fp::add(zz, m, mn);
#pragma unroll
    for (int i = 0; i < 10; ++i) mm.v[i] = m.v[i] + PrimeConsts<PrimeSecp>::subpad(i) - mn.v[i];     // (a factor of mul2: no carry, < 2^30)
#pragma unroll
    for (int i = 0; i < 10; ++i) ncrk.v[i] = PrimeConsts<PrimeSecp>::subpad(i) - crk.v[i];      // (no carry: a factor of mul2, limbs < 2^30)
    fe_select(s1, mm, m, c == 1);
    fe_select(s1, ncrk, s1, c == 2);
    // publish S0 = crk, S1, S2 = zz; fetch A, B, C, D by (lane of the quad, slot)
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      lds[(0 * 10 + i) * 64 + lane] = crk.v[i];
      lds[(1 * 10 + i) * 64 + lane] = s1.v[i];
      lds[(2 * 10 + i) * 64 + lane] = zz.v[i];
    }
    const int q0 = lane & ~3;
    // (lane, slot) of A, B, C, D per coordinate
    const int la = c == 0 ? 0 : 1, sa = c == 0 ? 0 : c == 1 ? 1 : 2;
    const int lb = 1, sb = c == 0 ? 1 : c == 1 ? 2 : 0;
    const int lc = c == 0 ? 1 : c == 1 ? 2 : 0, sc = c == 2 ? 1 : 0;
    const int ld = c == 0 ? 2 : 0, sd = c == 2 ? 0 : 1;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    Fe A, B, Cc, D;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      A.v[i] = lds[(sa * 10 + i) * 64 + q0 + la];
      B.v[i] = lds[(sb * 10 + i) * 64 + q0 + lb];
      Cc.v[i] = lds[(sc * 10 + i) * 64 + q0 + lc];
      D.v[i] = lds[(sd * 10 + i) * 64 + q0 + ld];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    Fp::mul2(s.a, A, B, Cc, D);
    fe_quad_perm<1, 2, 0, 1>(s.b, s.a);
  }
  __device__ static int out_offset(int role) { return role * 10; }
  __device__ static bool out_lane(int role) { return role < 3; }
  __device__ static void out_words(Fe& o, const St& s, int) { o = s.a; }
};

}  // namespace ec
