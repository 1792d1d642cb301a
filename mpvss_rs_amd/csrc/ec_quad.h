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
    asm("" : "+v"(x));                 // opaque to the optimiser: see the note on lane-dependent selects above
    r.v[i] = (u32)__builtin_amdgcn_mov_dpp((int)x, P0 | (P1 << 2) | (P2 << 4) | (P3 << 6), 0xf, 0xf, true);
  }
}
__device__ __forceinline__ void fe_select(Fe& r, const Fe& a, const Fe& b, bool take_a) {
#pragma unroll
  for (int i = 0; i < 10; ++i) r.v[i] = take_a ? a.v[i] : b.v[i];      // (v_cndmask_b32; the same through v_bfi_b32 and a VGPR mask measured 6 % slower)
}
// the same word of the lane LANES places up (the lanes of the next level); the last group gets its own
template <int LANES>
__device__ __forceinline__ void fe_from_next_group(Fe& r, const Fe& a) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    u32 x = a.v[i];
    asm("" : "+v"(x));
    r.v[i] = (u32)__shfl_down((int)x, LANES);
  }
}

struct QuadRist {
  typedef Ristretto C;
  typedef Ristretto::Fp Fp;
  static constexpr int LANES = 4, LEVELS = 16;             // lanes per point, points (levels) per wave
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
  static constexpr int LANES = 4, LEVELS = 16;
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

// secp256k1 with EIGHT lanes per point, six of them at work: every product of the complete addition in a lane of its own --
// ONE product and one more deep instead of two and a double one (QuadSecp), at twice the waves per chain; what a lone wave
// can issue is what bounds a step, so the products per lane are the step time.
//   lane m keeps the form u_m of its point:  u = (X, Y, Z, X + Y, Y + Z, X + Z)        (lanes 6, 7 mirror lanes 0, 1)
//   t_m = u_m(P) u_m(Q)                                                  -> X1X2, Y1Y2, Z1Z2 and the three Karatsuba products
//   mid_m = K (t_A -+ kB t_B - t_C) from LDS:  t3', t4', y3 = 21 t5', t1' = t1 - 21 t2, z3 = t1 + 21 t2, t0' = 3 t0
//   p_m = mid_a mid_b:  t4' y3, t3' t1', y3 t0', t1' z3, t0' t3', z3 t4'
//   u'_m = signed sums of the p:  X3 = p1 - p0, Y3 = p3 + p2, Z3 = p5 + p4 and their pairwise sums
// Checked in integers first (doublings, P + (-P), the identity); tests/ec_quad_unit.hip runs it beside QuadSecp.
struct OctSecp {
  typedef Secp C;
  typedef Secp::Fp Fp;
  static constexpr int LANES = 8, LEVELS = 8;
  static constexpr int LDS_WORDS = 3 * 10 * 64;             // three rounds x [word][lane]
  struct St {
    Fe u;
  };
  typedef St Nb;
  __device__ static int form(int role) { return role >= 6 ? role - 6 : role; }
  // u_m = cx X + cy Y + cz Z with coefficients 0 / 1; sy: Y enters negated
  __device__ static void from_xyz(Fe& u, const Fe& X, const Fe& Y, const Fe& Z, int role, bool negate_y) {
    const int m = form(role);
    const u32 mx = (m == 0 || m == 3 || m == 5) ? 0xffffffffu : 0u, my = (m == 1 || m == 3 || m == 4) ? 0xffffffffu : 0u,
              mz = (m == 2 || m == 4 || m == 5) ? 0xffffffffu : 0u;
    Fe yy;
    Fp::neg(yy, Y);
    fe_select(yy, yy, Y, negate_y);
#pragma unroll
    for (int i = 0; i < 10; ++i) u.v[i] = (X.v[i] & mx) + (yy.v[i] & my) + (Z.v[i] & mz);
  }
  __device__ static void load(St& s, const u32* __restrict__ pt, int role) {
    Fe X, Y, Z;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      X.v[i] = pt[i];
      Y.v[i] = pt[10 + i];
      Z.v[i] = pt[20 + i];
    }
    from_xyz(s.u, X, Y, Z, role, false);
  }
  __device__ static void identity(St& s, int role) {        // (0 : 1 : 0)
    const int m = form(role);
    Fp::zero(s.u);
    s.u.v[0] = (m == 1 || m == 3 || m == 4) ? 1u : 0u;
  }
  __device__ static const Fe& primary(const St& s) { return s.u; }
  __device__ static void select(St& r, const St& a, bool take_a) { fe_select(r.u, a.u, r.u, take_a); }
  __device__ static void nb_from_primary(Nb& q, const Fe& prim, int) { q.u = prim; }
  // one word of lane `src` of this lane's group
  __device__ static void gather(Fe& r, const Fe& a, int src) {
    const int lane = (threadIdx.x & 63 & ~7) + src;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      u32 x = a.v[i];
      asm("" : "+v"(x));
      r.v[i] = (u32)__shfl((int)x, lane);
    }
  }
  __device__ static void neg(St& s, int role) {            // (X, -Y, Z): the forms again from the coordinates in lanes 0, 1, 2
    Fe X, Y, Z;
    gather(X, s.u, 0);
    gather(Y, s.u, 1);
    gather(Z, s.u, 2);
    from_xyz(s.u, X, Y, Z, role, true);
  }
  __device__ static void add(St& s, const Nb& q, int role, u32* lds) {
    const int m = form(role), lane = threadIdx.x & 63, g0 = lane & ~7;
    // mid_m = K (t_A + sB kB t_B - [useC] t_C):   A, B, C: lanes of the group; sB: B enters negated
    const int iA = m == 0 ? 3 : m == 1 ? 4 : m == 2 ? 5 : m == 5 ? 0 : 1;
    const int iB = m == 1 ? 1 : (m == 3 || m == 4) ? 2 : 0;
    const int iC = m == 0 ? 1 : 2;
    const u32 kB = (m == 3 || m == 4) ? 21u : 1u, K = m == 2 ? 21u : m == 5 ? 3u : 1u;
    const bool negB = m != 4, useB = m != 5, useC = m < 3;
    // p_m = mid_a mid_b
    const int ia = m == 0 ? 1 : m == 1 ? 0 : m == 2 ? 2 : m == 3 ? 3 : m == 4 ? 5 : 4;
    const int ib = m == 0 ? 2 : m == 1 ? 3 : m == 2 ? 5 : m == 3 ? 4 : m == 4 ? 0 : 1;
    // u'_m = (X3 = p1 - p0 if wx) + (Y3 = p3 + p2 if wy) + (Z3 = p5 + p4 if wz)
    const bool wx = m == 0 || m == 3 || m == 5, wy = m == 1 || m == 3 || m == 4, wz = m == 2 || m == 4 || m == 5;
    u32* r0 = lds;
    u32* r1 = lds + 10 * 64;
    u32* r2 = lds + 20 * 64;
    Fe t, A, B, Cc, x, mid, ma, mb, pr;
    Fp::mul(t, s.u, q.u);
#pragma unroll
    for (int i = 0; i < 10; ++i) r0[i * 64 + lane] = t.v[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      A.v[i] = r0[i * 64 + g0 + iA];
      B.v[i] = r0[i * 64 + g0 + iB];
      Cc.v[i] = r0[i * 64 + g0 + iC];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    Fp::mul_small(B, B, kB);
    {
      const u32 mB = useB ? 0xffffffffu : 0u, mC = useC ? 0xffffffffu : 0u;
#pragma unroll
      for (int i = 0; i < 10; ++i) {
        const u32 pad = PrimeConsts<PrimeSecp>::subpad(i);
        const u32 bt = negB ? pad - B.v[i] : B.v[i];
        x.v[i] = A.v[i] + (bt & mB) + ((pad - Cc.v[i]) & mC);      // < 2^27.1 + 2^29 + 2^29
      }
    }
    Fp::mul_small(mid, x, K);
#pragma unroll
    for (int i = 0; i < 10; ++i) r1[i * 64 + lane] = mid.v[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      ma.v[i] = r1[i * 64 + g0 + ia];
      mb.v[i] = r1[i * 64 + g0 + ib];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    Fp::mul(pr, ma, mb);
#pragma unroll
    for (int i = 0; i < 10; ++i) r2[i * 64 + lane] = pr.v[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    {
      const u32 mX = wx ? 0xffffffffu : 0u, mY = wy ? 0xffffffffu : 0u, mZ = wz ? 0xffffffffu : 0u;
#pragma unroll
      for (int i = 0; i < 10; ++i) {
        const u32 p0 = r2[i * 64 + g0 + 0], p1 = r2[i * 64 + g0 + 1], p2 = r2[i * 64 + g0 + 2], p3 = r2[i * 64 + g0 + 3],
                  p4 = r2[i * 64 + g0 + 4], p5 = r2[i * 64 + g0 + 5];
        const u32 x3 = p1 + (PrimeConsts<PrimeSecp>::subpad(i) - p0), y3 = p3 + p2, z3 = p5 + p4;     // < 2^29.3, 2^28.1, 2^28.1
        s.u.v[i] = (x3 & mX) + (y3 & mY) + (z3 & mZ);                                                  // < 2^30
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    Fp::carry(s.u);
  }
  __device__ static int out_offset(int role) { return role * 10; }
  __device__ static bool out_lane(int role) { return role < 3; }
  __device__ static void out_words(Fe& o, const St& s, int) { o = s.u; }
};

}  // namespace ec
