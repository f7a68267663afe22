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

#include <type_traits>

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

// the same model on the generated code (README.md:60-71 "Changing the dynamics": edit the Lagrangian in
// tools/gen_dynamics.py, run it, rebuild with -DCPMPC_GENERATED_SINGLE=1).  Round 6: generated in the form the 6-state model
// has -- M(q) q'' = F terms with folded constants, sine / cosine and the helper terms as inputs, a hand-written 2 x 2 solve --
// so that it shares the hand-written model's instruction diet: the RK4 stages rotate the pole's sine / cosine from stage 1's
// pair (StepCache = TrigBase), the first pivot of the mass matrix and its reciprocal are parameters, one reciprocal per stage.
template <typename R>
struct SingleGenConsts {
  SinglePendulumMFConsts<R> g;
  R inv_m00;  // 1 / (m_b + m_1)
};
template <typename R>
struct SingleModelGenerated {
  static constexpr int NX = 4, NQ = 2, NP = 9;
  using Consts = SingleGenConsts<R>;
  using Sp = SinglePendulumMFSparsity;
  template <typename P>
  __host__ __device__ static Consts make(const P* p) {
    Consts k;
    k.g = single_pendulum_mf_consts<R, P>(p);
    k.inv_m00 = R(P(1) / (p[0] + p[1]));
    return k;
  }
  template <bool WITH_J, bool HAS_EXT>
  __device__ __forceinline__ static void accel_sc(const Consts& k, const R s, const R c, const R (&x)[NX], const R u,
                                                  const ExtForce<R>& fe, R (&a)[NQ], R (&Ja)[NQ][NX], R (&Jua)[NQ]) {
    R tv, sech2, n, inv_n, sr, sl, on_r, on_l, vx, vy;
    single_pendulum_mf_helpers<R, WITH_J>(k.g, s, c, x[0], x[2], x[3], tv, sech2, n, inv_n, sr, sl, on_r, on_l, vx, vy);
    R M[4], F[2], dFdx[8], dM1[4];
    if constexpr (HAS_EXT)
      single_pendulum_mf_terms_ext<R, WITH_J>(k.g, s, c, tv, sech2, n, inv_n, sr, sl, on_r, on_l, vx, vy, x[2], x[3], u, fe.fbx, fe.fmx, fe.fmy,
                                              M, F, dFdx, dM1);
    else
      single_pendulum_mf_terms_noext<R, WITH_J>(k.g, s, c, tv, sech2, n, inv_n, sr, sl, on_r, on_l, vx, vy, x[2], x[3], u, R(0), R(0), R(0), M,
                                                F, dFdx, dM1);
    // LDL^T of the 2 x 2 mass matrix: d0 = M_00 and 1 / d0 are parameters
    const R L = M[2] * k.inv_m00;
    const R id1 = Math<R>::rcp(M[3] - L * M[2]);
    {
      const R y1 = (F[1] - L * F[0]) * id1;
      a[1] = y1;
      a[0] = F[0] * k.inv_m00 - L * y1;
    }
    if (WITH_J) {
#pragma unroll
      for (int cc = 0; cc < NX; ++cc) {
        R r[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          bool have = Sp::dFdx[i * 4 + cc];
          R v = have ? dFdx[i * 4 + cc] : R(0);
          if (cc == 1) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
              if (Sp::dM1[i * 2 + j]) {
                v = have ? v - dM1[i * 2 + j] * a[j] : -(dM1[i * 2 + j] * a[j]);
                have = true;
              }
          }
          r[i] = v;
        }
        const R y1 = (r[1] - L * r[0]) * id1;
        Ja[1][cc] = y1;
        Ja[0][cc] = r[0] * k.inv_m00 - L * y1;
      }
      const R y1 = -L * id1;  // M^-1 e_0: the control acts on the base only (checked by the generator)
      Jua[1] = y1;
      Jua[0] = k.inv_m00 - L * y1;
    }
  }
  template <bool WITH_J, bool HAS_EXT>
  __device__ __forceinline__ static void accel(const Consts& k, const R (&x)[NX], const R u,
                                               const ExtForce<R>& fe, R (&a)[NQ], R (&Ja)[NQ][NX],
                                               R (&Jua)[NQ]) {
    R s, c;
    Math<R>::sincos(x[1], s, c);
    accel_sc<WITH_J, HAS_EXT>(k, s, c, x, u, fe, a, Ja, Jua);
  }
  using StepCache = TrigBase<R>;
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
    accel_sc<WITH_J, HAS_EXT>(k, s, c, x, u, fe, a, Ja, Jua);
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
// Round 6 (the 4-state model's instruction diet, through the generator): the parameter-only coefficients of the generated
// terms are folded once per parameter set (DoublePendulumGenConsts, host / kernel-argument segment) together with 1 / M_00,
// which is a parameter too; the generated code takes the poles' sines and cosines as INPUTS, so stages 2-4 of an RK4 step
// get both pairs by one rotation each from stage 1's (StepCache = two TrigBase: 8 full fp64 sincos per step -> 2); entries
// of dF/dx and dM/dth that vanish identically are neither written nor read (DoublePendulumGenSparsity), and the columns of
// da/dx that vanish identically -- b_x and b_x' do not enter these dynamics -- are known to the RK4 sensitivity chain
// (kJaZeroCols: a third of its multiply-adds were products with those zeros).
// ------------------------------------------------------------------------------------------------
#ifndef CPMPC_DOUBLE_TRIG_ROTATE
#define CPMPC_DOUBLE_TRIG_ROTATE 1   // 0: a full sincos of both angles at every RK4 stage (A/B)
#endif
// as CPMPC_F64_TRIG_CHAIN (cartpole_device.hpp) for this model: 2 = the rollouts of the fused kernel evaluate both base pairs in
// full once (chain_begin) and stage 1 of every step rotates from the previous step's pair
#ifndef CPMPC_DOUBLE_TRIG_CHAIN
#define CPMPC_DOUBLE_TRIG_CHAIN 0
#endif
#ifndef CPMPC_JA_ZERO_COLS
#define CPMPC_JA_ZERO_COLS 1         // 0: the sensitivity chain multiplies through the identically-zero columns (A/B)
#endif
template <typename R>
struct DoubleConsts {
  DoublePendulumGenConsts<R> g;  // generated: coefficients of the terms
  R inv_m00;                     // 1 / (m_b + m_1 + m_2): the first pivot of the mass matrix is a parameter
};
template <typename R>
struct DoubleTrig {
  TrigBase<R> t1, t2;
  __device__ __forceinline__ void invalidate() { t1.valid = t2.valid = false; }
};

template <typename R>
struct DoubleModel {
  static constexpr int NX = 6, NQ = 3, NP = 6;
  using Consts = DoubleConsts<R>;
  using Sp = DoublePendulumGenSparsity;
  static constexpr unsigned kJaZeroCols = CPMPC_JA_ZERO_COLS ? Sp::ja_zero_cols : 0u;
  template <typename P>
  __host__ __device__ static Consts make(const P* p) {
    Consts k;
    k.g = double_pendulum_gen_consts<R, P>(p);
    k.inv_m00 = R(P(1) / (p[0] + p[1] + p[2]));
    return k;
  }
  // LDL^T of the symmetric positive definite 3x3 mass matrix (its first pivot and that pivot's reciprocal are parameters)
  __device__ __forceinline__ static void factor(const R* M, const R id0, R (&L)[3], R (&id)[3]) {
    const R d0 = M[0];
    id[0] = id0;
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
  // the accelerations and their partials from the poles' sines and cosines
  template <bool WITH_J>
  __device__ __forceinline__ static void accel_sc(const Consts& k, const R s1, const R c1, const R s2, const R c2,
                                                  const R (&x)[NX], const R u, R (&a)[NQ], R (&Ja)[NQ][NX], R (&Jua)[NQ]) {
    R M[9], F[3], dFdx[18], dM1[9], dM2[9], L[3], id[3];
    double_pendulum_terms_sc<R>(k.g, s1, c1, s2, c2, x, u, M, F, dFdx, dM1, dM2);
    factor(M, k.inv_m00, L, id);
    solve(L, id, F[0], F[1], F[2], a);
    if (WITH_J) {
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        if ((Sp::ja_zero_cols >> c) & 1u) {  // the state component does not enter the dynamics: da/dx_c = 0
          Ja[0][c] = R(0);
          Ja[1][c] = R(0);
          Ja[2][c] = R(0);
          continue;
        }
        // right-hand side dF/dx_c - (dM/dx_c) a, identically-zero entries left out
        R r[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          bool have = Sp::dFdx[i * 6 + c];
          R v = have ? dFdx[i * 6 + c] : R(0);
          if (c == 1 || c == 2) {
            const R* dM = (c == 1) ? dM1 : dM2;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
              const bool nz = (c == 1) ? Sp::dM1[i * 3 + j] : Sp::dM2[i * 3 + j];
              if (nz) {
                v = have ? v - dM[i * 3 + j] * a[j] : -(dM[i * 3 + j] * a[j]);
                have = true;
              }
            }
          }
          r[i] = v;
        }
        R y[3];
        solve(L, id, r[0], r[1], r[2], y);
        Ja[0][c] = y[0];
        Ja[1][c] = y[1];
        Ja[2][c] = y[2];
      }
      // M^-1 e_0: z = (1, -L0, -L1 + L2 L0)
      const R z1 = -L[0];
      const R z2 = -L[1] - L[2] * z1;
      Jua[2] = z2 * id[2];
      Jua[1] = z1 * id[1] - L[2] * Jua[2];
      Jua[0] = id[0] - L[0] * Jua[1] - L[1] * Jua[2];
    }
  }
  template <bool WITH_J, bool HAS_EXT>
  __device__ __forceinline__ static void accel(const Consts& k, const R (&x)[NX], const R u,
                                               const ExtForce<R>&, R (&a)[NQ], R (&Ja)[NQ][NX],
                                               R (&Jua)[NQ]) {
    R s1, c1, s2, c2;
    Math<R>::sincos(x[1], s1, c1);
    Math<R>::sincos(x[2], s2, c2);
    accel_sc<WITH_J>(k, s1, c1, s2, c2, x, u, a, Ja, Jua);
  }
  // what the stages of an RK4 step (and consecutive steps of a rollout) share: both poles' sine / cosine pairs
  using StepCache = DoubleTrig<R>;
  __device__ __forceinline__ static void chain_begin(StepCache& sc, const R (&x)[NX]) {
    if constexpr (Math<R>::kIncrementalTrig && CPMPC_DOUBLE_TRIG_ROTATE && CPMPC_DOUBLE_TRIG_CHAIN == 2) {
      Math<R>::sincos(x[1], sc.t1.s0, sc.t1.c0);
      Math<R>::sincos(x[2], sc.t2.s0, sc.t2.c0);
      sc.t1.th0 = x[1];
      sc.t2.th0 = x[2];
      sc.t1.valid = sc.t2.valid = true;
    }
  }
  template <bool WITH_J, bool HAS_EXT, int STAGE>
  __device__ __forceinline__ static void accel_stage(const Consts& k, const R (&x)[NX], const R u,
                                                     const ExtForce<R>& fe, R (&a)[NQ], R (&Ja)[NQ][NX],
                                                     R (&Jua)[NQ], StepCache& sc) {
    if constexpr (CPMPC_DOUBLE_TRIG_ROTATE) {
      R s1, c1, s2, c2;
      stage_sincos<R, STAGE, CPMPC_DOUBLE_TRIG_CHAIN>(sc.t1, x[1], s1, c1);
      stage_sincos<R, STAGE, CPMPC_DOUBLE_TRIG_CHAIN>(sc.t2, x[2], s2, c2);
      accel_sc<WITH_J>(k, s1, c1, s2, c2, x, u, a, Ja, Jua);
    } else {
      accel<WITH_J, HAS_EXT>(k, x, u, fe, a, Ja, Jua);
    }
  }
};

// columns of da/dx that vanish identically for the model (a bit mask; models without the knowledge: none)
template <typename M, typename = void>
struct JaZeroCols {
  static constexpr unsigned value = 0u;
};
template <typename M>
struct JaZeroCols<M, std::void_t<decltype(M::kJaZeroCols)>> {
  static constexpr unsigned value = M::kJaZeroCols;
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
// step of the rollout, start a new rollout with a fresh object (or sc.invalidate()).
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

template <int NX>
__host__ __device__ constexpr int first_nonzero_col(unsigned zmask) {
  for (int c = 0; c < NX; ++c)
    if (!((zmask >> c) & 1u)) return c;
  return 0;
}

// Columns of the step Jacobian A = dx+/dx that are known in closed form.  A state component c that does not enter the
// dynamics (zmask bit c: Ja[:, c] = 0 at every stage) gives, by D_1 = K_1, D_{j+1} = K_{j+1} (I + a_j D_j):
//   a position c < NQ:                                   D_j[:, c] = 0       for every stage,  A[:, c] = e_c
//   a velocity c = NQ + r whose position r is such too:  D_j[:, c] = e_r     for every stage,  A[:, c] = e_c + h e_r
// so the chain neither computes nor stores those columns, and products with A take them as the 0, 1 and h they are.
template <int NX, int NQ>
__host__ __device__ constexpr unsigned trivial_cols(unsigned zmask) {
  unsigned t = 0;
  for (int c = 0; c < NX; ++c) {
    if (!((zmask >> c) & 1u)) continue;
    if (c < NQ || ((zmask >> (c - NQ)) & 1u)) t |= 1u << c;
  }
  return t;
}
#ifndef CPMPC_JA_TRIVIAL_COLS
#define CPMPC_JA_TRIVIAL_COLS 1   // 0: the chain carries the identically-known columns like any other (A/B)
#endif
// The pattern those columns leave in every product of step Jacobians (Phi of an interval, Psi = diag(w) Phi ... Phi across
// intervals): in a trivial column c only row c and, for a velocity, row c - NQ can be non-zero -- A[:, c] = e_c (+ h e_{c-NQ}),
// and the pattern is closed under the product (its c - NQ is a trivial position column, whose only row is itself).  Entry
// (r, c) with struct_zero(triv, r, c) is an exact zero: the fused kernel neither computes, stores nor multiplies by it.
template <int NX, int NQ>
__host__ __device__ constexpr bool struct_zero(unsigned triv, int r, int c) {
  if (!((triv >> c) & 1u)) return false;
  return !(r == c || (c >= NQ && r == c - NQ));
}

// ------------------------------------------------------------------------------------------------
// RK4 with sensitivities A = dx+/dx (NX x NX), Bv = dx+/du (NX)   (integration.hpp:13-49).
//   D_1 = K_1,  D_{j+1} = K_{j+1} (I + a_j D_j),  a = {h/2, h/2, h},  A = I + h/6 (D_1 + 2 D_2 + 2 D_3 + D_4)
// With K = [[0 I],[Ja]] the top NQ rows of K X are the bottom NQ rows of X and the bottom NQ rows are
// Ja X, so each stage product costs NQ*NX*NX multiply-adds instead of NX^3.
// ------------------------------------------------------------------------------------------------
// ZMASK: bit c set = column c of Ja vanishes identically for the model (JaZeroCols): its products are not issued and its
// entry is not added (the compiler may not drop a multiplication by a value it cannot prove finite).
template <typename R, int NX, int NQ, unsigned ZMASK = 0u>
__device__ __forceinline__ void stage_chain_m(const R (&Ja)[NQ][NX], const R (&Jua)[NQ], const R a,
                                              const R (&D)[NX][NX], const R (&d)[NX], R (&Dn)[NX][NX],
                                              R (&dn)[NX]) {
  constexpr int k0 = first_nonzero_col<NX>(ZMASK);
  constexpr unsigned TRIV = CPMPC_JA_TRIVIAL_COLS ? trivial_cols<NX, NQ>(ZMASK) : 0u;
#pragma unroll
  for (int c = 0; c < NX; ++c) {
    if ((TRIV >> c) & 1u) continue;  // known in closed form: neither computed nor stored
#pragma unroll
    for (int r = 0; r < NQ; ++r) Dn[r][c] = a * D[NQ + r][c] + (c == NQ + r ? R(1) : R(0));
#pragma unroll
    for (int r = 0; r < NQ; ++r) {
      R acc = Ja[r][k0] * D[k0][c];
#pragma unroll
      for (int kk = k0 + 1; kk < NX; ++kk)
        if (!((ZMASK >> kk) & 1u)) acc += Ja[r][kk] * D[kk][c];
      Dn[NQ + r][c] = ((ZMASK >> c) & 1u) ? a * acc : Ja[r][c] + a * acc;
    }
  }
#pragma unroll
  for (int r = 0; r < NQ; ++r) {
    dn[r] = a * d[NQ + r];
    R acc = Ja[r][k0] * d[k0];
#pragma unroll
    for (int kk = k0 + 1; kk < NX; ++kk)
      if (!((ZMASK >> kk) & 1u)) acc += Ja[r][kk] * d[kk];
    dn[NQ + r] = a * acc + Jua[r];
  }
}

// The same product for stage 2, where D = D_1 = [[0 I],[Ja1]] and d = [0; Jua1] are known by their structure: the products
// with the zeros and ones of the top rows are not issued (the compiler may not drop a multiplication by a literal 0.0 on
// its own: 0 * inf).  Same operations in the same order on everything that is not such a constant, so a finite result has
// the bits stage_chain_m gives.
template <typename R, int NX, int NQ, unsigned ZMASK = 0u>
__device__ __forceinline__ void stage_chain_first_m(const R (&Ja)[NQ][NX], const R (&Jua)[NQ], const R a,
                                                    const R (&Ja1)[NQ][NX], const R (&Jua1)[NQ], R (&Dn)[NX][NX],
                                                    R (&dn)[NX]) {
  // D_1's rows: top r = e_{NQ + r}, bottom r = Ja1[r].  (Ja D_1)[r][c] = sum_kk<NQ Ja[r][kk] [c == NQ + kk] + sum_kk Ja[r][NQ + kk] Ja1[kk][c]
  constexpr unsigned TRIV = CPMPC_JA_TRIVIAL_COLS ? trivial_cols<NX, NQ>(ZMASK) : 0u;
#pragma unroll
  for (int c = 0; c < NX; ++c) {
    if ((TRIV >> c) & 1u) continue;
    const bool czero = (ZMASK >> c) & 1u;   // then Ja1[.][c] = Ja[.][c] = 0
#pragma unroll
    for (int r = 0; r < NQ; ++r) Dn[r][c] = czero ? (c == NQ + r ? R(1) : R(0)) : a * Ja1[r][c] + (c == NQ + r ? R(1) : R(0));
#pragma unroll
    for (int r = 0; r < NQ; ++r) {
      bool have = false;
      R acc = R(0);
      if (c >= NQ && !((ZMASK >> (c - NQ)) & 1u)) {
        acc = Ja[r][c - NQ];  // the one of the identity block
        have = true;
      }
      if (!czero) {
#pragma unroll
        for (int kk = 0; kk < NQ; ++kk) {
          if ((ZMASK >> (NQ + kk)) & 1u) continue;
          acc = have ? acc + Ja[r][NQ + kk] * Ja1[kk][c] : Ja[r][NQ + kk] * Ja1[kk][c];
          have = true;
        }
      }
      Dn[NQ + r][c] = czero ? (have ? a * acc : R(0)) : (have ? Ja[r][c] + a * acc : Ja[r][c]);
    }
  }
#pragma unroll
  for (int r = 0; r < NQ; ++r) {
    dn[r] = a * Jua1[r];
    bool have = false;
    R acc = R(0);
#pragma unroll
    for (int kk = 0; kk < NQ; ++kk) {
      if ((ZMASK >> (NQ + kk)) & 1u) continue;
      acc = have ? acc + Ja[r][NQ + kk] * Jua1[kk] : Ja[r][NQ + kk] * Jua1[kk];
      have = true;
    }
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
  constexpr unsigned ZM = JaZeroCols<M>::value;
  constexpr unsigned TRIV = CPMPC_JA_TRIVIAL_COLS ? trivial_cols<NX, NQ>(ZM) : 0u;
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
      if ((TRIV >> c) & 1u) continue;
      D[r][c] = (c == NQ + r ? R(1) : R(0));
      D[NQ + r][c] = Ja[r][c];
    }
    d[r] = R(0);
    d[NQ + r] = Jua[r];
  }
#pragma unroll
  for (int r = 0; r < NX; ++r) {
#pragma unroll
    for (int c = 0; c < NX; ++c)
      if (!((TRIV >> c) & 1u)) As[r][c] = D[r][c];
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
    stage_chain_first_m<R, NX, NQ, ZM>(Ja, Jua, hh, Ja1, Jua1, Dn, dn);
  }
#else
  M::template accel_stage<true, HAS_EXT, 2>(k, xt, u, fe, a2, Ja, Jua, sc);
  stage_chain_m<R, NX, NQ, ZM>(Ja, Jua, hh, D, d, Dn, dn);
#endif
#pragma unroll
  for (int r = 0; r < NX; ++r) {
#pragma unroll
    for (int c = 0; c < NX; ++c) {
      if ((TRIV >> c) & 1u) continue;
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
  stage_chain_m<R, NX, NQ, ZM>(Ja, Jua, hh, D, d, Dn, dn);
#pragma unroll
  for (int r = 0; r < NX; ++r) {
#pragma unroll
    for (int c = 0; c < NX; ++c) {
      if ((TRIV >> c) & 1u) continue;
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
  const R h6 = h / R(6);
  stage_chain_m<R, NX, NQ, ZM>(Ja, Jua, h, D, d, Dn, dn);
#pragma unroll
  for (int r = 0; r < NX; ++r) {
#pragma unroll
    for (int c = 0; c < NX; ++c) {
      if ((TRIV >> c) & 1u) A[r][c] = (r == c) ? R(1) : ((c >= NQ && r == c - NQ) ? h : R(0));  // e_c (+ h e_{c-NQ}): see trivial_cols
      else A[r][c] = (r == c ? R(1) : R(0)) + h6 * (As[r][c] + Dn[r][c]);
    }
    Bv[r] = h6 * (bs[r] + dn[r]);
  }
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const R v1 = x[NQ + i];
    x[i] += h6 * (v1 + v2[i] * R(2) + v3[i] * R(2) + v4[i]);
    x[NQ + i] += h6 * (a1[i] + a2[i] * R(2) + a3[i] * R(2) + a4[i]);
  }
}

// y = A v for the step Jacobian A of rk4_step_jac_m<R, M> with step h: the columns of A that are known in closed form
// (trivial_cols) enter as what they are -- v_c to row c, h v_c to row c - NQ -- and not as products with stored 0, 1, h.
template <typename R, typename M>
__device__ __forceinline__ void step_jac_apply(const R (&A)[M::NX][M::NX], const R h, const R (&v)[M::NX], R (&y)[M::NX]) {
  constexpr int NX = M::NX, NQ = M::NQ;
  constexpr unsigned TRIV = CPMPC_JA_TRIVIAL_COLS ? trivial_cols<NX, NQ>(JaZeroCols<M>::value) : 0u;
#pragma unroll
  for (int r = 0; r < NX; ++r) {
    bool have = false;
    R acc = R(0);
#pragma unroll
    for (int m = 0; m < NX; ++m) {
      if ((TRIV >> m) & 1u) continue;
      acc = have ? acc + A[r][m] * v[m] : A[r][m] * v[m];
      have = true;
    }
#pragma unroll
    for (int m = 0; m < NX; ++m) {
      if (!((TRIV >> m) & 1u)) continue;
      if (r == m) acc = have ? acc + v[m] : v[m], have = true;
      if (m >= NQ && r == m - NQ) acc = have ? acc + h * v[m] : h * v[m], have = true;
    }
    y[r] = acc;
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
