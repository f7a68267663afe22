// models.hpp -- model policies (cart + single pole, cart + double pole) and the model-generic RK4
// steps with and without sensitivities.  A model is a second-order mechanical system
//     x = [q (NQ positions: base, then pole angles); q' (NQ velocities)],   x' = [q'; a(x, u)]
// and provides the accelerations a with their partials Ja = da/dx (NQ x NX) and Jua = da/du (NQ).
// The full stage Jacobian is [[0 I],[Ja]]; only Ja is ever materialised.
//
// Reference: runge_kutta_4th_order<D> / _no_jacobians (optimization/integration.hpp:13-62), which is
// templated on the state dimension D exactly for this generalisation (optimization.cc:197-199).
#pragma once
#include <hip/hip_runtime.h>

#include "cartpole_device.hpp"
#include "double_pendulum_gen.hpp"
#include "single_pendulum_gen.hpp"

// 1: the single-pole kernels are built on the code tools/gen_dynamics.py emits (single_pendulum_gen.hpp) instead of the
// hand-written cartpole_accel.  Same results to 1e-12 (tests/test_generated_dynamics.py); the generated form evaluates
// the closed-form 2x2 inverse and its symbolic derivatives like the reference's generator does and costs more
// instructions, so the hand-written one stays the default.
#ifndef CPMPC_GENERATED_SINGLE
#define CPMPC_GENERATED_SINGLE 0
#endif

namespace cpmpc {

// ------------------------------------------------------------------------------------------------
// cart + single pole: gen::single_pendulum_dynamics (single_pendulum_dynamics.hpp:13-186)
// ------------------------------------------------------------------------------------------------
template <typename R>
struct SingleModelHandWritten {
  static constexpr int NX = 4, NQ = 2, NP = 9;
  using Consts = CartPoleConsts<R>;
  template <typename P>
  __host__ __device__ static Consts make(const P* p) {
    return make_consts<R, P>(p);
  }
  template <bool WITH_J, bool HAS_EXT>
  __device__ __forceinline__ static void accel(const Consts& k, const R (&x)[NX], const R u,
                                               const ExtForce<R>& fe, R (&a)[NQ], R (&Ja)[NQ][NX],
                                               R (&Jua)[NQ]) {
    cartpole_accel<R, WITH_J, HAS_EXT>(k, x[0], x[1], x[2], x[3], u, fe, a[0], a[1], Ja, Jua);
  }
  // the same at stage STAGE (1..4) of an RK4 step, with what the stages can share (the pole angle's sine and cosine)
  using StepCache = TrigBase<R>;
  // before the first step of a rollout from x (CPMPC_F64_TRIG_CHAIN = 2: the base pair in full, once per rollout)
  __device__ __forceinline__ static void chain_begin(StepCache& sc, const R (&x)[NX]) {
    if constexpr (Math<R>::kIncrementalTrig && CPMPC_F64_TRIG_CHAIN == 2) {
      Math<R>::sincos(x[1], sc.s0, sc.c0);
      sc.th0 = x[1];
      sc.valid = true;
    }
  }
  template <bool WITH_J, bool HAS_EXT, int STAGE>
  __device__ __forceinline__ static void accel_stage(const Consts& k, const R (&x)[NX], const R u,
                                                     const ExtForce<R>& fe, R (&a)[NQ], R (&Ja)[NQ][NX],
                                                     R (&Jua)[NQ], StepCache& sc) {
    R s, c;
    stage_sincos<R, STAGE>(sc, x[1], s, c);
    cartpole_accel_sc<R, WITH_J, HAS_EXT>(k, x[0], s, c, x[2], x[3], u, fe, a[0], a[1], Ja, Jua);
  }
};

// models without anything to share between the stages of a step
struct NoStepCache {
  bool valid = false;
};
// the same model on the generated code (README.md:60-71 "Changing the dynamics": edit the Lagrangian in
// tools/gen_dynamics.py, run it, rebuild with -DCPMPC_GENERATED_SINGLE=1)
template <typename R>
struct SingleModelGenerated {
  static constexpr int NX = 4, NQ = 2, NP = 9;
  using Consts = SinglePendulumGenConsts<R>;
  template <typename P>
  __host__ __device__ static Consts make(const P* p) {
    return single_pendulum_gen_consts<R, P>(p);
  }
  template <bool WITH_J, bool HAS_EXT>
  __device__ __forceinline__ static void accel(const Consts& k, const R (&x)[NX], const R u,
                                               const ExtForce<R>& fe, R (&a)[NQ], R (&Ja)[NQ][NX],
                                               R (&Jua)[NQ]) {
    if constexpr (HAS_EXT)
      single_pendulum_gen_accel_ext<R, WITH_J>(k, x[0], x[1], x[2], x[3], u, fe.fbx, fe.fmx, fe.fmy, a, Ja, Jua);
    else
      single_pendulum_gen_accel_noext<R, WITH_J>(k, x[0], x[1], x[2], x[3], u, R(0), R(0), R(0), a, Ja, Jua);
  }
  using StepCache = NoStepCache;
  __device__ __forceinline__ static void chain_begin(StepCache&, const R (&)[NX]) {}
  template <bool WITH_J, bool HAS_EXT, int STAGE>
  __device__ __forceinline__ static void accel_stage(const Consts& k, const R (&x)[NX], const R u,
                                                     const ExtForce<R>& fe, R (&a)[NQ], R (&Ja)[NQ][NX],
                                                     R (&Jua)[NQ], StepCache&) {
    accel<WITH_J, HAS_EXT>(k, x, u, fe, a, Ja, Jua);
  }
};

#if CPMPC_GENERATED_SINGLE
template <typename R>
using SingleModel = SingleModelGenerated<R>;
#else
template <typename R>
using SingleModel = SingleModelHandWritten<R>;
#endif

// ------------------------------------------------------------------------------------------------
// cart + double pole: symbolic/dynamics_double.py:25-148 (generated terms, numeric 3x3 solve)
//   a = M^-1 F,  da/dx_c = M^-1 (dF/dx_c - dM/dx_c a),  da/du = M^-1 e_0.  No dissipation, no
//   external forces (HAS_EXT is ignored).
// ------------------------------------------------------------------------------------------------
template <typename R>
struct DoubleConsts {
  R p[6];  // m_b, m_1, m_2, l_1, l_2, g
};

template <typename R>
struct DoubleModel {
  static constexpr int NX = 6, NQ = 3, NP = 6;
  using Consts = DoubleConsts<R>;
  template <typename P>
  __host__ __device__ static Consts make(const P* p) {
    Consts k;
    for (int i = 0; i < 6; ++i) k.p[i] = R(p[i]);
    return k;
  }
  // LDL^T of the symmetric positive definite 3x3 mass matrix
  __device__ __forceinline__ static void factor(const R* M, R (&L)[3], R (&id)[3]) {
    const R d0 = M[0];
    id[0] = Math<R>::rcp(d0);
    L[0] = M[3] * id[0];
    L[1] = M[6] * id[0];
    const R d1 = M[4] - L[0] * L[0] * d0;
    id[1] = Math<R>::rcp(d1);
    L[2] = (M[7] - L[1] * L[0] * d0) * id[1];
    const R d2 = M[8] - L[1] * L[1] * d0 - L[2] * L[2] * d1;
    id[2] = Math<R>::rcp(d2);
  }
  __device__ __forceinline__ static void solve(const R (&L)[3], const R (&id)[3], const R b0, const R b1,
                                               const R b2, R (&y)[3]) {
    const R z0 = b0;
    const R z1 = b1 - L[0] * z0;
    const R z2 = b2 - L[1] * z0 - L[2] * z1;
    y[2] = z2 * id[2];
    y[1] = z1 * id[1] - L[2] * y[2];
    y[0] = z0 * id[0] - L[0] * y[1] - L[1] * y[2];
  }
  template <bool WITH_J, bool HAS_EXT>
  __device__ __forceinline__ static void accel(const Consts& k, const R (&x)[NX], const R u,
                                               const ExtForce<R>&, R (&a)[NQ], R (&Ja)[NQ][NX],
                                               R (&Jua)[NQ]) {
    R M[9], F[3], dFdx[18], dM1[9], dM2[9], L[3], id[3];
    double_pendulum_terms<R>(k.p, x, u, M, F, dFdx, dM1, dM2);
    factor(M, L, id);
    solve(L, id, F[0], F[1], F[2], a);
    if (WITH_J) {
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        R r0 = dFdx[0 * 6 + c], r1 = dFdx[1 * 6 + c], r2 = dFdx[2 * 6 + c];
        if (c == 1 || c == 2) {
          const R* dM = (c == 1) ? dM1 : dM2;
          r0 -= dM[0] * a[0] + dM[1] * a[1] + dM[2] * a[2];
          r1 -= dM[3] * a[0] + dM[4] * a[1] + dM[5] * a[2];
          r2 -= dM[6] * a[0] + dM[7] * a[1] + dM[8] * a[2];
        }
        R y[3];
        solve(L, id, r0, r1, r2, y);
        Ja[0][c] = y[0];
        Ja[1][c] = y[1];
        Ja[2][c] = y[2];
      }
      solve(L, id, R(1), R(0), R(0), Jua);
    }
  }
  using StepCache = NoStepCache;
  __device__ __forceinline__ static void chain_begin(StepCache&, const R (&)[NX]) {}
  template <bool WITH_J, bool HAS_EXT, int STAGE>
  __device__ __forceinline__ static void accel_stage(const Consts& k, const R (&x)[NX], const R u,
                                                     const ExtForce<R>& fe, R (&a)[NQ], R (&Ja)[NQ][NX],
                                                     R (&Jua)[NQ], StepCache&) {
    accel<WITH_J, HAS_EXT>(k, x, u, fe, a, Ja, Jua);
  }
};

// pole angles are components 1..NQ-1 (wrapped to (-pi, pi]); component 0 is the base position (clamped)
template <typename M>
__host__ __device__ constexpr bool is_angle(int t) {
  return t >= 1 && t < M::NQ;
}

// ------------------------------------------------------------------------------------------------
// RK4 without sensitivities (integration.hpp:52-62).  x updated in place.
// ------------------------------------------------------------------------------------------------
// `sc` carries what consecutive steps of one rollout can share (models.hpp: StepCache): pass the same object to every
// step of the rollout, start a new rollout with sc.valid = false.
template <typename R, typename M, bool HAS_EXT>
__device__ __forceinline__ void rk4_step_m(const typename M::Consts& k, const R h, R (&x)[M::NX], const R u,
                                           const ExtForce<R>& fe, typename M::StepCache& sc) {
  constexpr int NX = M::NX, NQ = M::NQ;
  R Ja[NQ][NX], Jua[NQ];
  R a1[NQ], a2[NQ], a3[NQ], a4[NQ], v2[NQ], v3[NQ], v4[NQ], xt[NX];
  const R hh = h / R(2);
  M::template accel_stage<false, HAS_EXT, 1>(k, x, u, fe, a1, Ja, Jua, sc);  // k1 = [x_v; a1]
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    v2[i] = x[NQ + i] + a1[i] * hh;
    xt[i] = x[i] + x[NQ + i] * hh;
    xt[NQ + i] = v2[i];
  }
  M::template accel_stage<false, HAS_EXT, 2>(k, xt, u, fe, a2, Ja, Jua, sc);  // k2 = [v2; a2]
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    v3[i] = x[NQ + i] + a2[i] * hh;
    xt[i] = x[i] + v2[i] * hh;
    xt[NQ + i] = v3[i];
  }
  M::template accel_stage<false, HAS_EXT, 3>(k, xt, u, fe, a3, Ja, Jua, sc);  // k3 = [v3; a3]
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    v4[i] = x[NQ + i] + a3[i] * h;
    xt[i] = x[i] + v3[i] * h;
    xt[NQ + i] = v4[i];
  }
  M::template accel_stage<false, HAS_EXT, 4>(k, xt, u, fe, a4, Ja, Jua, sc);  // k4 = [v4; a4]
  const R h6 = h / R(6);
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const R v1 = x[NQ + i];
    x[i] += h6 * (v1 + v2[i] * R(2) + v3[i] * R(2) + v4[i]);
    x[NQ + i] += h6 * (a1[i] + a2[i] * R(2) + a3[i] * R(2) + a4[i]);
  }
}

template <typename R, typename M, bool HAS_EXT>
__device__ __forceinline__ void rk4_step_m(const typename M::Consts& k, const R h, R (&x)[M::NX], const R u,
                                           const ExtForce<R>& fe) {
  typename M::StepCache sc;  // a single step: nothing to chain from
  rk4_step_m<R, M, HAS_EXT>(k, h, x, u, fe, sc);
}

// ------------------------------------------------------------------------------------------------
// RK4 with sensitivities A = dx+/dx (NX x NX), Bv = dx+/du (NX)   (integration.hpp:13-49).
//   D_1 = K_1,  D_{j+1} = K_{j+1} (I + a_j D_j),  a = {h/2, h/2, h},  A = I + h/6 (D_1 + 2 D_2 + 2 D_3 + D_4)
// With K = [[0 I],[Ja]] the top NQ rows of K X are the bottom NQ rows of X and the bottom NQ rows are
// Ja X, so each stage product costs NQ*NX*NX multiply-adds instead of NX^3.
// ------------------------------------------------------------------------------------------------
template <typename R, int NX, int NQ>
__device__ __forceinline__ void stage_chain_m(const R (&Ja)[NQ][NX], const R (&Jua)[NQ], const R a,
                                              const R (&D)[NX][NX], const R (&d)[NX], R (&Dn)[NX][NX],
                                              R (&dn)[NX]) {
#pragma unroll
  for (int c = 0; c < NX; ++c) {
#pragma unroll
    for (int r = 0; r < NQ; ++r) Dn[r][c] = a * D[NQ + r][c] + (c == NQ + r ? R(1) : R(0));
#pragma unroll
    for (int r = 0; r < NQ; ++r) {
      R acc = Ja[r][0] * D[0][c];
#pragma unroll
      for (int kk = 1; kk < NX; ++kk) acc += Ja[r][kk] * D[kk][c];
      Dn[NQ + r][c] = Ja[r][c] + a * acc;
    }
  }
#pragma unroll
  for (int r = 0; r < NQ; ++r) {
    dn[r] = a * d[NQ + r];
    R acc = Ja[r][0] * d[0];
#pragma unroll
    for (int kk = 1; kk < NX; ++kk) acc += Ja[r][kk] * d[kk];
    dn[NQ + r] = a * acc + Jua[r];
  }
}

// The same product for stage 2, where D = D_1 = [[0 I],[Ja1]] and d = [0; Jua1] are known by their structure: the products
// with the zeros and ones of the top rows are not issued (the compiler may not drop a multiplication by a literal 0.0 on
// its own: 0 * inf).  Same operations in the same order on everything that is not such a constant, so a finite result has
// the bits stage_chain_m gives.
template <typename R, int NX, int NQ>
__device__ __forceinline__ void stage_chain_first_m(const R (&Ja)[NQ][NX], const R (&Jua)[NQ], const R a,
                                                    const R (&Ja1)[NQ][NX], const R (&Jua1)[NQ], R (&Dn)[NX][NX],
                                                    R (&dn)[NX]) {
#pragma unroll
  for (int c = 0; c < NX; ++c) {
#pragma unroll
    for (int r = 0; r < NQ; ++r) Dn[r][c] = a * Ja1[r][c] + (c == NQ + r ? R(1) : R(0));
#pragma unroll
    for (int r = 0; r < NQ; ++r) {
      R acc;
      if (c >= NQ) {
        acc = Ja[r][c - NQ];  // the one of the identity block
#pragma unroll
        for (int kk = 0; kk < NQ; ++kk) acc += Ja[r][NQ + kk] * Ja1[kk][c];
      } else {
        acc = Ja[r][NQ] * Ja1[0][c];
#pragma unroll
        for (int kk = 1; kk < NQ; ++kk) acc += Ja[r][NQ + kk] * Ja1[kk][c];
      }
      Dn[NQ + r][c] = Ja[r][c] + a * acc;
    }
  }
#pragma unroll
  for (int r = 0; r < NQ; ++r) {
    dn[r] = a * Jua1[r];
    R acc = Ja[r][NQ] * Jua1[0];
#pragma unroll
    for (int kk = 1; kk < NQ; ++kk) acc += Ja[r][NQ + kk] * Jua1[kk];
    dn[NQ + r] = a * acc + Jua[r];
  }
}

#ifndef CPMPC_RK4_STAGE2_STRUCTURED
#define CPMPC_RK4_STAGE2_STRUCTURED 1
#endif

template <typename R, typename M, bool HAS_EXT>
__device__ __forceinline__ void rk4_step_jac_m(const typename M::Consts& k, const R h, R (&x)[M::NX],
                                               const R u, const ExtForce<R>& fe, R (&A)[M::NX][M::NX],
                                               R (&Bv)[M::NX], typename M::StepCache& sc) {
  constexpr int NX = M::NX, NQ = M::NQ;
  const R hh = h / R(2);
  R Ja[NQ][NX], Jua[NQ];
  R D[NX][NX], d[NX], Dn[NX][NX], dn[NX];
  R As[NX][NX], bs[NX];  // running sums D_1 + 2 D_2 + 2 D_3 + D_4
  R a1[NQ], a2[NQ], a3[NQ], a4[NQ], v2[NQ], v3[NQ], v4[NQ], xt[NX];

  // stage 1
  M::template accel_stage<true, HAS_EXT, 1>(k, x, u, fe, a1, Ja, Jua, sc);
#pragma unroll
  for (int r = 0; r < NQ; ++r) {
#pragma unroll
    for (int c = 0; c < NX; ++c) {
      D[r][c] = (c == NQ + r ? R(1) : R(0));
      D[NQ + r][c] = Ja[r][c];
    }
    d[r] = R(0);
    d[NQ + r] = Jua[r];
  }
#pragma unroll
  for (int r = 0; r < NX; ++r) {
#pragma unroll
    for (int c = 0; c < NX; ++c) As[r][c] = D[r][c];
    bs[r] = d[r];
  }

  // stage 2
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    v2[i] = x[NQ + i] + a1[i] * hh;
    xt[i] = x[i] + x[NQ + i] * hh;
    xt[NQ + i] = v2[i];
  }
#if CPMPC_RK4_STAGE2_STRUCTURED
  {
    R Ja1[NQ][NX], Jua1[NQ];
#pragma unroll
    for (int r = 0; r < NQ; ++r) {
#pragma unroll
      for (int c = 0; c < NX; ++c) Ja1[r][c] = Ja[r][c];
      Jua1[r] = Jua[r];
    }
    M::template accel_stage<true, HAS_EXT, 2>(k, xt, u, fe, a2, Ja, Jua, sc);
    stage_chain_first_m<R, NX, NQ>(Ja, Jua, hh, Ja1, Jua1, Dn, dn);
  }
#else
  M::template accel_stage<true, HAS_EXT, 2>(k, xt, u, fe, a2, Ja, Jua, sc);
  stage_chain_m<R, NX, NQ>(Ja, Jua, hh, D, d, Dn, dn);
#endif
#pragma unroll
  for (int r = 0; r < NX; ++r) {
#pragma unroll
    for (int c = 0; c < NX; ++c) {
      As[r][c] += Dn[r][c] * R(2);
      D[r][c] = Dn[r][c];
    }
    bs[r] += dn[r] * R(2);
    d[r] = dn[r];
  }

  // stage 3
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    v3[i] = x[NQ + i] + a2[i] * hh;
    xt[i] = x[i] + v2[i] * hh;
    xt[NQ + i] = v3[i];
  }
  M::template accel_stage<true, HAS_EXT, 3>(k, xt, u, fe, a3, Ja, Jua, sc);
  stage_chain_m<R, NX, NQ>(Ja, Jua, hh, D, d, Dn, dn);
#pragma unroll
  for (int r = 0; r < NX; ++r) {
#pragma unroll
    for (int c = 0; c < NX; ++c) {
      As[r][c] += Dn[r][c] * R(2);
      D[r][c] = Dn[r][c];
    }
    bs[r] += dn[r] * R(2);
    d[r] = dn[r];
  }

  // stage 4
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    v4[i] = x[NQ + i] + a3[i] * h;
    xt[i] = x[i] + v3[i] * h;
    xt[NQ + i] = v4[i];
  }
  M::template accel_stage<true, HAS_EXT, 4>(k, xt, u, fe, a4, Ja, Jua, sc);
  stage_chain_m<R, NX, NQ>(Ja, Jua, h, D, d, Dn, dn);

  const R h6 = h / R(6);
#pragma unroll
  for (int r = 0; r < NX; ++r) {
#pragma unroll
    for (int c = 0; c < NX; ++c) A[r][c] = (r == c ? R(1) : R(0)) + h6 * (As[r][c] + Dn[r][c]);
    Bv[r] = h6 * (bs[r] + dn[r]);
  }
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const R v1 = x[NQ + i];
    x[i] += h6 * (v1 + v2[i] * R(2) + v3[i] * R(2) + v4[i]);
    x[NQ + i] += h6 * (a1[i] + a2[i] * R(2) + a3[i] * R(2) + a4[i]);
  }
}

// ------------------------------------------------------------------------------------------------
// The same sensitivities WITHOUT forming A (round 5, the 6-state model in the fused kernel).  rk4_step_jac_m holds three
// NX x NX blocks (D_j, D_{j+1}, their weighted sum) next to the stage Jacobian while it builds A, and the caller a fourth
// and fifth (Phi and A Phi): with NX = 6 that is ~180 values live at the peak, more than the float kernel's 256-register
// budget for two waves per SIMD allows and, in double, the source of a third of the step's instructions
// (accumulation-register traffic of the spills).  Here the step keeps only what the chain rule needs -- the four stage
// Jacobians Ja_j (NQ x NX) and Jua_j (NQ) -- and every direction is then pushed through the four stages ON ITS OWN:
//     w_j = Ja_j z_{j-1} (+ Jua_j),   z_0 = v,   z_j = v + a_j [ (z_{j-1})_bottom ; w_j ],   a = {h/2, h/2, h}
//     v+_top    = v_top + h v_bottom + h^2/6 (w_1 + w_2 + w_3)
//     v+_bottom = v_bottom + h/6 (w_1 + 2 w_2 + 2 w_3 + w_4)
// (the top NQ rows of K z are the bottom rows of z, so only the bottoms w_j are ever computed).  The columns of Phi, the
// earlier columns of Gamma and the new column B (v = 0, w_j += Jua_j) are such directions: ~110 multiply-adds each instead
// of 36 for a product with a formed A, 84 + 21 live values instead of ~180.  Same mathematics as integration.hpp:36-46,
// another order of the sums.
// ------------------------------------------------------------------------------------------------
template <typename R, typename M, bool HAS_EXT>
__device__ __forceinline__ void rk4_step_stages_m(const typename M::Consts& k, const R h, R (&x)[M::NX], const R u,
                                                  const ExtForce<R>& fe, R (&JaS)[4][M::NQ][M::NX], R (&JuaS)[4][M::NQ],
                                                  typename M::StepCache& sc) {
  constexpr int NX = M::NX, NQ = M::NQ;
  R a1[NQ], a2[NQ], a3[NQ], a4[NQ], v2[NQ], v3[NQ], v4[NQ], xt[NX];
  const R hh = h / R(2);
  M::template accel_stage<true, HAS_EXT, 1>(k, x, u, fe, a1, JaS[0], JuaS[0], sc);
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    v2[i] = x[NQ + i] + a1[i] * hh;
    xt[i] = x[i] + x[NQ + i] * hh;
    xt[NQ + i] = v2[i];
  }
  M::template accel_stage<true, HAS_EXT, 2>(k, xt, u, fe, a2, JaS[1], JuaS[1], sc);
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    v3[i] = x[NQ + i] + a2[i] * hh;
    xt[i] = x[i] + v2[i] * hh;
    xt[NQ + i] = v3[i];
  }
  M::template accel_stage<true, HAS_EXT, 3>(k, xt, u, fe, a3, JaS[2], JuaS[2], sc);
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    v4[i] = x[NQ + i] + a3[i] * h;
    xt[i] = x[i] + v3[i] * h;
    xt[NQ + i] = v4[i];
  }
  M::template accel_stage<true, HAS_EXT, 4>(k, xt, u, fe, a4, JaS[3], JuaS[3], sc);
  const R h6 = h / R(6);
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const R v1 = x[NQ + i];
    x[i] += h6 * (v1 + v2[i] * R(2) + v3[i] * R(2) + v4[i]);
    x[NQ + i] += h6 * (a1[i] + a2[i] * R(2) + a3[i] * R(2) + a4[i]);
  }
}

// one direction through the four stages, in place.  WITH_U: the direction of the control (v = 0 on entry is assumed and v
// is overwritten with B = dx+/du).
template <typename R, int NX, int NQ, bool WITH_U>
__device__ __forceinline__ void rk4_push_direction(const R (&JaS)[4][NQ][NX], const R (&JuaS)[4][NQ], const R h, R (&v)[NX]) {
  const R hh = h / R(2);
  R zt[NQ], zb[NQ], w[NQ], s3[NQ], s6[NQ];  // z_{j-1} (top, bottom), w_j, w_1 + w_2 + w_3, w_1 + 2 w_2 + 2 w_3 + w_4
  // stage 1: z_0 = v
#pragma unroll
  for (int r = 0; r < NQ; ++r) {
    if constexpr (WITH_U) {
      w[r] = JuaS[0][r];
    } else {
      R acc = JaS[0][r][0] * v[0];
#pragma unroll
      for (int c = 1; c < NX; ++c) acc += JaS[0][r][c] * v[c];
      w[r] = acc;
    }
    s3[r] = w[r];
    s6[r] = w[r];
  }
  // stages 2 .. 4
#pragma unroll
  for (int j = 1; j < 4; ++j) {
    const R a = (j == 3) ? h : hh;
#pragma unroll
    for (int r = 0; r < NQ; ++r) {
      if constexpr (WITH_U) {
        zt[r] = (j == 1) ? R(0) : a * zb[r];  // (z_0)_bottom = 0
      } else {
        zt[r] = v[r] + a * ((j == 1) ? v[NQ + r] : zb[r]);
      }
    }
#pragma unroll
    for (int r = 0; r < NQ; ++r) zb[r] = WITH_U ? a * w[r] : v[NQ + r] + a * w[r];
#pragma unroll
    for (int r = 0; r < NQ; ++r) {
      R acc = JaS[j][r][NQ] * zb[0];
#pragma unroll
      for (int c = 1; c < NQ; ++c) acc += JaS[j][r][NQ + c] * zb[c];
      if (!(WITH_U && j == 1)) {  // (z_1)_top = 0 for the control's direction
#pragma unroll
        for (int c = 0; c < NQ; ++c) acc += JaS[j][r][c] * zt[c];
      }
      if constexpr (WITH_U) acc += JuaS[j][r];
      w[r] = acc;
    }
#pragma unroll
    for (int r = 0; r < NQ; ++r) {
      if (j < 3) s3[r] += w[r];
      s6[r] += (j < 3) ? w[r] * R(2) : w[r];
    }
  }
  const R h6 = h / R(6), hh6 = h * h6;
#pragma unroll
  for (int r = 0; r < NQ; ++r) {
    if constexpr (WITH_U) {
      v[r] = hh6 * s3[r];
      v[NQ + r] = h6 * s6[r];
    } else {
      v[r] = (v[r] + h * v[NQ + r]) + hh6 * s3[r];
      v[NQ + r] = v[NQ + r] + h6 * s6[r];
    }
  }
}

template <typename R, typename M, bool HAS_EXT>
__device__ __forceinline__ void rk4_step_jac_m(const typename M::Consts& k, const R h, R (&x)[M::NX],
                                               const R u, const ExtForce<R>& fe, R (&A)[M::NX][M::NX],
                                               R (&Bv)[M::NX]) {
  typename M::StepCache sc;
  rk4_step_jac_m<R, M, HAS_EXT>(k, h, x, u, fe, A, Bv, sc);
}

}  // namespace cpmpc
