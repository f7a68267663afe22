// mpc_kernels.hpp -- the batched MPC inner loop as gfx950 kernels, one problem per lane.
//
// Pipeline of one batched re-plan (host loop in cpmpc_api.hip; reference: Optimization::Step,
// optimization/optimization.cc:39-97):
//
//   prepare_kernel      guess (warm shift / sinusoid) + FillInitialGuess      optimization.cc:46-71,333-351
//   repeat max_iterations times:
//     linearize_kernel  every shooting interval: RK4 + sensitivities          optimization.cc:99-160
//     qp_ls_kernel      structured equality-constrained QP + merit line search (role of mini_opt)
//   finalize_kernel     ComputePredictedStates + outputs                      optimization.cc:85-96,353-371
//
// Workspace layout in HBM: field-major with the problem index fastest.  Every 4-vector of the problem
// (a shooting node, a column of Gamma, a row of Phi, a defect) is ONE 16-byte (fp32) / 32-byte (fp64)
// element, so lane i of a wave reads `field_base + i` as a single global_load_dwordx4 and the wave
// moves 1 KiB fully coalesced; scalars (controls, per-problem solver state) are 4/8-byte elements,
// 256/512 B per wave.  The field base is wave-uniform (scalar registers), the lane offset is the
// 32-bit problem index.  No LDS: per-lane state lives in VGPRs, the per-interval sensitivities
// stream through the workspace once per SQP iteration.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cartpole_device.hpp"

namespace cpmpc {

constexpr int kTermNone = 0;
constexpr int kTermMaxIterations = 1;
constexpr int kTermRelTol = 3;
constexpr int kTermFirstOrder = 4;
constexpr int kTermQpIndefinite = 5;
constexpr int kTermMaxLambda = 7;
constexpr int kTermNonFinite = 8;

// per-problem real scalars kept in the workspace (index into `sc`)
enum { SC_LAMBDA = 0, SC_MU, SC_F_LAST, SC_CN_LAST, SC_UPREV, SC_ALPHA, SC_COUNT };
// per-problem int scalars (index into `ist`)
enum { IS_STATUS = 0, IS_ITERS, IS_LS_EVALS, IS_FAILED, IS_COUNT };

template <typename R>
struct VecT;
template <>
struct VecT<float> {
  using V4 = float4;
};
template <>
struct VecT<double> {
  using V4 = double4;
};

template <typename R>
__device__ __forceinline__ typename VecT<R>::V4 mk4(R a, R b, R c, R d) {
  typename VecT<R>::V4 v;
  v.x = a;
  v.y = b;
  v.z = c;
  v.w = d;
  return v;
}

template <typename R>
struct SolverArgs {
  using V4 = typename VecT<R>::V4;
  // sizes
  int64_t B;       // problems in this call
  int64_t stride;  // elements between consecutive fields (capacity of the workspace)
  int N, S, SP;
  R dt;
  // cost weights (optimization.hpp:40-48)
  R wu, wd;          // u_cost_weight, u_derivative_cost_weight (0 disables the rows)
  R term_w[4];       // residual weight of terminal row t (1 for an equality row)
  R term_tgt[4];     // targets; [0] is the shared set-point unless set_point != nullptr
  int term_is_cost;  // bit t set: terminal row t is a cost (weight >= 0), else an equality
  // solver options
  int max_ls;
  R c1, shrink_max, shrink_min, alpha_growth, rho;
  R lam_init, lam_fail_init, lam_up, lam_down, lam_min, lam_max;
  R bx_lim, u_lim;
  R rel_tol, fo_tol, mu_init;
  int has_prev;
  // workspace (device)
  V4* zx;    // [S]       shooting nodes of the iterate          } persist between calls:
  R* zu;     // [N]       controls of the iterate                } the warm start
  V4* dzx;   // [S]       QP step, nodes
  R* dzu;    // [N]       QP step, controls
  V4* Phi;   // [4(S-1)]  row r of Phi_s at field 4s+r
  V4* Gam;   // [N]       column k of Gamma = d x_end / d u_k
  V4* cs;    // [S-1]     shooting defects
  V4* Wk;    // [N]       row k of U^-1 R^T
  V4* Tk;    // [N]       {(U^-1 g)_k, upsilon_k, 1/d_k, g_k}
  R* sc;     // [SC_COUNT]
  int32_t* ist;        // [IS_COUNT]
  const R* sin_table;  // [N] device: u_guess_sinusoid_amplitude * sin(2 pi k / N), from the host
  // inputs, packed [field][B]
  const R* x0;         // [4]
  const R* dyn;        // [9] per-problem, or nullptr
  const R* set_point;  // [1] per-problem, or nullptr
  CartPoleConsts<R> consts;  // shared model constants (used when dyn == nullptr)
  // outputs, packed [field][B] (nullable)
  R* u_out;
  R* pred_out;
  int32_t* status_out;
  int32_t* iters_out;
  int32_t* ls_out;
  R* cost_out;
  R* eq_out;
  R* guess_out;
};

template <typename R>
__device__ __forceinline__ CartPoleConsts<R> load_consts(const SolverArgs<R>& a, unsigned p) {
  if (a.dyn == nullptr) return a.consts;
  R prm[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) prm[i] = a.dyn[i * a.B + p];
  return make_consts<R, R>(prm);
}

template <typename R>
__device__ __forceinline__ R clampr(R v, R lo, R hi) {
  return v < lo ? lo : (v > hi ? hi : v);
}

template <typename R>
__device__ __forceinline__ R dot4(const typename VecT<R>::V4& a, const R (&b)[4]) {
  return a.x * b[0] + a.y * b[1] + a.z * b[2] + a.w * b[3];
}

// ------------------------------------------------------------------------------------------------
// prepare: initial guess.  One thread per problem.
// ------------------------------------------------------------------------------------------------
template <typename R>
__global__ __launch_bounds__(64) void prepare_kernel(const SolverArgs<R> a) {
  const unsigned p = blockIdx.x * 64u + threadIdx.x;
  if (p >= a.B) return;
  const int64_t st = a.stride;
  const CartPoleConsts<R> k = load_consts(a, p);
  const ExtForce<R> fe{R(0), R(0), R(0)};

  // BuildProblem reads u_prev before the previous solution is overwritten (optimization.cc:288-291)
  R u_prev = R(0);
  if (a.has_prev) {
    u_prev = a.zu[p];
    // shift the controls left by one, duplicate the last (optimization.cc:54-57)
    for (int kk = 0; kk + 1 < a.N; ++kk) a.zu[(int64_t)kk * st + p] = a.zu[(int64_t)(kk + 1) * st + p];
  } else {
    for (int kk = 0; kk < a.N; ++kk) a.zu[(int64_t)kk * st + p] = a.sin_table[kk];
  }
  a.sc[SC_UPREV * st + p] = u_prev;
  a.sc[SC_LAMBDA * st + p] = a.lam_init;
  a.sc[SC_MU * st + p] = a.mu_init;
  a.sc[SC_F_LAST * st + p] = R(0);
  a.sc[SC_CN_LAST * st + p] = R(0);
  a.sc[SC_ALPHA * st + p] = R(1);
  a.ist[IS_STATUS * st + p] = kTermNone;
  a.ist[IS_ITERS * st + p] = 0;
  a.ist[IS_LS_EVALS * st + p] = 0;
  a.ist[IS_FAILED * st + p] = 0;

  // FillInitialGuess (optimization.cc:333-351): roll the states, wrapping after every step
  R x[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) x[t] = a.x0[t * a.B + p];
  a.zx[p] = mk4<R>(x[0], x[1], x[2], x[3]);
  if (a.guess_out) {
#pragma unroll
    for (int t = 0; t < 4; ++t) a.guess_out[(int64_t)t * a.B + p] = x[t];
  }
  int kk = 0;
  for (int s = 1; s < a.S; ++s) {
    for (int i = 0; i < a.SP; ++i, ++kk) {
      const R u = a.zu[(int64_t)kk * st + p];
      rk4_step<R, false>(k, a.dt, x, u, fe);
      x[1] = mod_pi(x[1]);
    }
    a.zx[(int64_t)s * st + p] = mk4<R>(x[0], x[1], x[2], x[3]);
    if (a.guess_out) {
#pragma unroll
      for (int t = 0; t < 4; ++t) a.guess_out[(int64_t)(4 * s + t) * a.B + p] = x[t];
    }
  }
  if (a.guess_out)
    for (int i = 0; i < a.N; ++i) a.guess_out[(int64_t)(4 * a.S + i) * a.B + p] = a.zu[(int64_t)i * st + p];
}

// ------------------------------------------------------------------------------------------------
// linearize: one thread per (problem, shooting interval).
// Integrates x_s through SP controls with RK4, accumulating Phi = dx_end/dx_s and
// Gamma = dx_end/du FORWARD (Phi <- A Phi, Gamma_j <- A Gamma_j, Gamma_i = B) so that nothing per
// step has to be stored: the 4 x SP block lives in registers with static indices.  Algebraically
// equal to the reference's backward accumulation (optimization.cc:145-154).
// ------------------------------------------------------------------------------------------------
template <typename R, int SP>
__global__ __launch_bounds__(64) void linearize_kernel(const SolverArgs<R> a,
                                                        const typename VecT<R>::V4* zx_in,
                                                        const R* zu_in, const int32_t* status) {
  using V4 = typename VecT<R>::V4;
  const int64_t gid = (int64_t)blockIdx.x * 64 + threadIdx.x;
  const int s = (int)(gid / a.B);
  const unsigned p = (unsigned)(gid - (int64_t)s * a.B);
  if (s >= a.S - 1) return;
  const int64_t st = a.stride;
  if (status != nullptr && status[IS_STATUS * st + p] != kTermNone) return;
  const CartPoleConsts<R> k = load_consts(a, p);
  const ExtForce<R> fe{R(0), R(0), R(0)};

  const V4 xs = zx_in[(int64_t)s * st + p];
  const V4 xe = zx_in[(int64_t)(s + 1) * st + p];
  R x[4] = {xs.x, xs.y, xs.z, xs.w};
  R Phi[4][4];
  R Gam[SP][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) Phi[r][c] = (r == c) ? R(1) : R(0);
#pragma unroll
  for (int j = 0; j < SP; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) Gam[j][r] = R(0);

  const R* zu = zu_in + (int64_t)(s * SP) * st;
  R u_next = zu[p];
#pragma unroll 1
  for (int i = 0; i < SP; ++i) {
    const R u = u_next;
    if (i + 1 < SP) u_next = zu[(int64_t)(i + 1) * st + p];  // prefetch the next control
    R A[4][4], Bv[4];
    rk4_step_jac<R, false>(k, a.dt, x, u, fe, A, Bv);
    // Phi <- A Phi
    R T[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c)
        T[r][c] = A[r][0] * Phi[0][c] + A[r][1] * Phi[1][c] + A[r][2] * Phi[2][c] + A[r][3] * Phi[3][c];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) Phi[r][c] = T[r][c];
      // Gamma columns: j < i propagate, j == i is the new control's column.  `i` is uniform over
      // the wave, so these are scalar branches and the register indices stay static.
#pragma unroll
    for (int j = 0; j < SP; ++j) {
      if (j < i) {
        const R g0 = Gam[j][0], g1 = Gam[j][1], g2 = Gam[j][2], g3 = Gam[j][3];
#pragma unroll
        for (int r = 0; r < 4; ++r) Gam[j][r] = A[r][0] * g0 + A[r][1] * g1 + A[r][2] * g2 + A[r][3] * g3;
      } else if (j == i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) Gam[j][r] = Bv[r];
      }
    }
  }
  // wrap the angle once at the end of the interval, then the defect (optimization.cc:139,156-157)
  x[1] = mod_pi(x[1]);
  const R c1 = mod_pi(x[1] - xe.y);
  a.cs[(int64_t)s * st + p] = mk4<R>(x[0] - xe.x, c1, x[2] - xe.z, x[3] - xe.w);
#pragma unroll
  for (int r = 0; r < 4; ++r)
    a.Phi[(int64_t)(4 * s + r) * st + p] = mk4<R>(Phi[r][0], Phi[r][1], Phi[r][2], Phi[r][3]);
#pragma unroll
  for (int j = 0; j < SP; ++j)
    a.Gam[(int64_t)(s * SP + j) * st + p] = mk4<R>(Gam[j][0], Gam[j][1], Gam[j][2], Gam[j][3]);
}

// ------------------------------------------------------------------------------------------------
// merit evaluation at z (+) alpha dz: 1/2 |r|^2 and |c|_1 through the retraction
// (optimization.cc:309-329) and a no-Jacobian rollout of every interval (optimization.cc:130-139).
// ------------------------------------------------------------------------------------------------
template <typename R>
__device__ __forceinline__ void trial_node(const SolverArgs<R>& a, const int s, const unsigned p,
                                           const R alpha, R (&xs)[4]) {
  using V4 = typename VecT<R>::V4;
  const V4 zv = a.zx[(int64_t)s * a.stride + p];
  const V4 dv = a.dzx[(int64_t)s * a.stride + p];
  xs[0] = clampr(zv.x + alpha * dv.x, -a.bx_lim, a.bx_lim);
  xs[1] = mod_pi(zv.y + alpha * dv.y);
  xs[2] = zv.z + alpha * dv.z;
  xs[3] = zv.w + alpha * dv.w;
}

template <typename R>
__device__ __forceinline__ void merit_eval(const SolverArgs<R>& a, const CartPoleConsts<R>& k,
                                           const unsigned p, const R alpha, const R (&xm)[4],
                                           const R (&tgt)[4], const R u_prev, R& f_out, R& cn_out) {
  const int64_t st = a.stride;
  const ExtForce<R> fe{R(0), R(0), R(0)};
  R f = R(0), cn = R(0);

  // node 0 and the initial-state equality rows (optimization.cc:228-232)
  R xs[4];
  trial_node<R>(a, 0, p, alpha, xs);
  cn += Math<R>::fabs(xs[0] - xm[0]) + Math<R>::fabs(mod_pi(xs[1] - xm[1])) + Math<R>::fabs(xs[2] - xm[2]) +
        Math<R>::fabs(xs[3] - xm[3]);

  R u_before = u_prev;  // u_{k-1} of the trial point, for the du rows
  int kk = 0;
  R u_raw = a.zu[p] + alpha * a.dzu[p];
  for (int s = 0; s + 1 < a.S; ++s) {
    R x[4] = {xs[0], xs[1], xs[2], xs[3]};
    for (int i = 0; i < a.SP; ++i, ++kk) {
      const R u = clampr(u_raw, -a.u_lim, a.u_lim);
      if (kk + 1 < a.N)  // prefetch the next control of the trial point
        u_raw = a.zu[(int64_t)(kk + 1) * st + p] + alpha * a.dzu[(int64_t)(kk + 1) * st + p];
      // control cost rows (optimization.cc:270-301)
      const R ru = a.wu * u;
      const R rd = a.wd * (u_before - u);  // (u_{k-1} - u_k) w; for k = 0 it is -(u_0 - u_prev) w
      f += ru * ru + rd * rd;
      u_before = u;
      rk4_step<R, false>(k, a.dt, x, u, fe);
    }
    x[1] = mod_pi(x[1]);
    trial_node<R>(a, s + 1, p, alpha, xs);  // next node of the trial point
    cn += Math<R>::fabs(x[0] - xs[0]) + Math<R>::fabs(mod_pi(x[1] - xs[1])) + Math<R>::fabs(x[2] - xs[2]) +
          Math<R>::fabs(x[3] - xs[3]);
  }
  // terminal rows on the last node (optimization.cc:236-267)
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    R d = xs[t] - tgt[t];
    if (t == 1) d = mod_pi(d);
    if ((a.term_is_cost >> t) & 1) {
      const R r = a.term_w[t] * d;
      f += r * r;
    } else {
      cn += Math<R>::fabs(d);
    }
  }
  f_out = R(0.5) * f;
  cn_out = cn;
}

// ------------------------------------------------------------------------------------------------
// qp_ls: one thread per problem.  Solves
//     min 1/2 |J dz + r|^2 + 1/2 lambda |du|^2   s.t.  A dz + c = 0
// exactly, without forming it: the states are eliminated through the shooting recursion
//     dx_0 = -c_init,  dx_{s+1} = Phi_s dx_s + Gamma_s du_s + c_s,
// which leaves a QP in du whose Hessian is T + R^T R with T tridiagonal (control costs) and R the
// <= 4 terminal rows (cost or equality).
//   sweep 1 (k descending): T = U D U^T by a scalar recurrence; W = U^-1 R^T row by row from
//           m_k = Psi Gamma_k, Psi = diag(w) Phi_{S-2} ... Phi_{s+1}; gw = U^-1 g; accumulate
//           S = W^T D^-1 W (4x4) and rho = W^T D^-1 gw; rows of W and {gw, upsilon, 1/d, g} are stored.
//   4x4 LDL^T of S + diag(1 for cost rows, 0 for equality rows) in registers -> multipliers q.
//   sweep 2 (k ascending): y = -(gw + W q), U^T du = D^-1 y, state recovery through Phi/Gamma, and the
//           directional quantities g.du and |J dz|^2.
// Then the l1-merit penalty update and the Armijo line search with quadratic-interpolation
// backtracking, started from the remembered step length.
// ------------------------------------------------------------------------------------------------
template <typename R>
__global__ __launch_bounds__(64) void qp_ls_kernel(const SolverArgs<R> a) {
  using V4 = typename VecT<R>::V4;
  const unsigned p = blockIdx.x * 64u + threadIdx.x;
  if (p >= a.B) return;
  const int64_t st = a.stride;
  if (a.ist[IS_STATUS * st + p] != kTermNone) return;
  const CartPoleConsts<R> k = load_consts(a, p);
  const int N = a.N, S = a.S, SP = a.SP;

  R lam = a.sc[SC_LAMBDA * st + p];
  R mu = a.sc[SC_MU * st + p];
  const R u_prev = a.sc[SC_UPREV * st + p];
  R a_start = a.sc[SC_ALPHA * st + p];
  R tgt[4] = {a.term_tgt[0], a.term_tgt[1], a.term_tgt[2], a.term_tgt[3]};
  if (a.set_point) tgt[0] = a.set_point[p];
  R xm[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) xm[t] = a.x0[t * a.B + p];

  const R wu2 = a.wu * a.wu, wd2 = a.wd * a.wd;

  // ---- residuals at z: constraint l1 norm, a = dx_{S-1} for du = 0, terminal rows ---------------
  R f = R(0), cn = R(0);
  R ci[4];
  {
    const V4 z0 = a.zx[p];
    ci[0] = z0.x - xm[0];
    ci[1] = mod_pi(z0.y - xm[1]);
    ci[2] = z0.z - xm[2];
    ci[3] = z0.w - xm[3];
#pragma unroll
    for (int t = 0; t < 4; ++t) cn += Math<R>::fabs(ci[t]);
  }
  // The weighted free response  ha = diag(w) dx_{S-1}|_{du=0} = sum_s Psi_s c_s - Psi_{-1} c_init  is
  // accumulated inside sweep 1, where Psi_s = diag(w) Phi_{S-2}...Phi_{s+1} is available anyway.
  R hv[4], Rw[4], Dg[4], e_term[4];
  {
    const V4 zT = a.zx[(int64_t)(S - 1) * st + p];
    const R zt[4] = {zT.x, zT.y, zT.z, zT.w};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      R d = zt[t] - tgt[t];
      if (t == 1) d = mod_pi(d);
      e_term[t] = d;
      const bool is_cost = (a.term_is_cost >> t) & 1;
      Rw[t] = a.term_w[t];
      Dg[t] = is_cost ? R(1) : R(0);
      if (is_cost) {
        const R r = a.term_w[t] * d;
        f += r * r;
      } else {
        cn += Math<R>::fabs(d);
      }
      hv[t] = Rw[t] * d;  // + ha[t], added after sweep 1
    }
  }

  // ---- sweep 1 (k descending) -------------------------------------------------------------------
  R Sm[4][4], rho[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    rho[i] = R(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) Sm[i][j] = R(0);
  }
  bool pd_ok = true;
  {
    R Psi[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) Psi[r][c] = (r == c) ? Rw[r] : R(0);
    R wprev[4] = {R(0), R(0), R(0), R(0)};
    R ha[4] = {R(0), R(0), R(0), R(0)};
    R gwprev = R(0);
    R d_next = R(1);
    const V4* __restrict__ gam_p = a.Gam + p;
    const R* __restrict__ zu_p = a.zu + p;
    R u_hi = R(0);                              // u_{k+1}
    R u_cur = zu_p[(int64_t)(N - 1) * st];      // u_k
    // software pipeline: the loads of column k-1 are issued before column k is consumed
    V4 G_nx = gam_p[(int64_t)(N - 1) * st];
    R u_nx = (N > 1) ? zu_p[(int64_t)(N - 2) * st] : u_prev;
    int kk = N - 1;
    for (int s = S - 2; s >= 0; --s) {
      for (int i = SP - 1; i >= 0; --i, --kk) {
        const V4 G = G_nx;
        const R u_lo = u_nx;  // u_{k-1} (u_prev for k = 0)
        if (kk > 0) {
          G_nx = gam_p[(int64_t)(kk - 1) * st];
          u_nx = (kk > 1) ? zu_p[(int64_t)(kk - 2) * st] : u_prev;
        }
        // control cost rows at z, tridiagonal entries and the control-cost gradient g_k
        const R ru = a.wu * u_cur, rd = a.wd * (u_lo - u_cur);
        f += ru * ru + rd * rd;
        const R nd = (kk < N - 1 ? R(1) : R(0)) + R(1);  // du rows touching u_k
        const R diag = wu2 + lam + wd2 * nd;
        R g = wu2 * u_cur + wd2 * (u_cur - u_lo);
        if (kk < N - 1) g += wd2 * (u_cur - u_hi);
        // U D U^T recurrence (off-diagonal of T is -wd2)
        const R ups = (kk < N - 1) ? (-wd2 / d_next) : R(0);
        const R dk = diag + wd2 * ups;
        if (!(dk > R(0))) pd_ok = false;
        const R inv_d = R(1) / dk;
        d_next = dk;
        // m_k = Psi Gamma_k ; w_k = m_k - ups w_{k+1}
        const R gk[4] = {G.x, G.y, G.z, G.w};
        R wk[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const R m = Psi[r][0] * gk[0] + Psi[r][1] * gk[1] + Psi[r][2] * gk[2] + Psi[r][3] * gk[3];
          wk[r] = m - ups * wprev[r];
        }
        const R gw = g - ups * gwprev;
        a.Wk[(int64_t)kk * st + p] = mk4<R>(wk[0], wk[1], wk[2], wk[3]);
        a.Tk[(int64_t)kk * st + p] = mk4<R>(gw, ups, inv_d, g);
#pragma unroll
        for (int i2 = 0; i2 < 4; ++i2) {
          const R wi = wk[i2] * inv_d;
          rho[i2] += wi * gw;
#pragma unroll
          for (int j2 = 0; j2 <= i2; ++j2) Sm[i2][j2] += wi * wk[j2];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) wprev[r] = wk[r];
        gwprev = gw;
        u_hi = u_cur;
        u_cur = u_lo;
      }
      // defect of this interval: |c|_1 and its weighted propagation to the last node, Psi_s c_s
      {
        const V4 c = a.cs[(int64_t)s * st + p];
        cn += Math<R>::fabs(c.x) + Math<R>::fabs(c.y) + Math<R>::fabs(c.z) + Math<R>::fabs(c.w);
#pragma unroll
        for (int r = 0; r < 4; ++r) ha[r] += Psi[r][0] * c.x + Psi[r][1] * c.y + Psi[r][2] * c.z + Psi[r][3] * c.w;
      }
      // Psi <- Psi Phi_s
      R Ph[4][4], T[4][4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const V4 row = a.Phi[(int64_t)(4 * s + r) * st + p];
        Ph[r][0] = row.x;
        Ph[r][1] = row.y;
        Ph[r][2] = row.z;
        Ph[r][3] = row.w;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c)
          T[r][c] = Psi[r][0] * Ph[0][c] + Psi[r][1] * Ph[1][c] + Psi[r][2] * Ph[2][c] + Psi[r][3] * Ph[3][c];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) Psi[r][c] = T[r][c];
    }
    // Psi is now diag(w) Phi_{S-2}...Phi_0: contribution of dx_0 = -c_init
#pragma unroll
    for (int r = 0; r < 4; ++r)
      hv[r] += ha[r] - (Psi[r][0] * ci[0] + Psi[r][1] * ci[1] + Psi[r][2] * ci[2] + Psi[r][3] * ci[3]);
  }
  f *= R(0.5);

  int status = kTermNone;
  if (!Math<R>::finite(f) || !Math<R>::finite(cn)) status = kTermNonFinite;

  // ---- (S + Dg) q = h - rho by LDL^T on the lower triangle, in registers ------------------------
  R q[4];
  {
    R Lm[4][4], dv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) Sm[i][i] += Dg[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      R dj = Sm[j][j];
#pragma unroll
      for (int m = 0; m < j; ++m) dj -= Lm[j][m] * Lm[j][m] * dv[m];
      if (!(dj > R(0))) pd_ok = false;
      dv[j] = dj;
      const R inv = R(1) / dj;
#pragma unroll
      for (int i = j + 1; i < 4; ++i) {
        R v = Sm[i][j];
#pragma unroll
        for (int m = 0; m < j; ++m) v -= Lm[i][m] * Lm[j][m] * dv[m];
        Lm[i][j] = v * inv;
      }
    }
    R y[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      R v = hv[i] - rho[i];
#pragma unroll
      for (int m = 0; m < i; ++m) v -= Lm[i][m] * y[m];
      y[i] = v;
    }
#pragma unroll
    for (int i = 3; i >= 0; --i) {
      R v = y[i] / dv[i];
#pragma unroll
      for (int m = i + 1; m < 4; ++m) v -= Lm[m][i] * q[m];
      q[i] = v;
    }
  }
  if (status == kTermNone && !pd_ok) status = kTermQpIndefinite;

  // ---- sweep 2 (k ascending): U^T du = D^-1 y, state recovery, directional quantities -----------
  R gd = R(0), curv = R(0);
  {
    R dx[4] = {-ci[0], -ci[1], -ci[2], -ci[3]};
    a.dzx[p] = mk4<R>(dx[0], dx[1], dx[2], dx[3]);
    R du_prev = R(0);   // du_{k-1}; the (u_0 - u_prev) row sees only du_0
    R ups_prev = R(0);  // upsilon_{k-1}
    const V4* __restrict__ w_p = a.Wk + p;
    const V4* __restrict__ t_p = a.Tk + p;
    const V4* __restrict__ g_p = a.Gam + p;
    V4 W_nx = w_p[0], T_nx = t_p[0], G_nx = g_p[0];  // software pipeline, one column ahead
    int kk = 0;
    for (int s = 0; s + 1 < S; ++s) {
      const V4 c = a.cs[(int64_t)s * st + p];
      R acc[4];
      acc[0] = c.x + dot4<R>(a.Phi[(int64_t)(4 * s + 0) * st + p], dx);
      acc[1] = c.y + dot4<R>(a.Phi[(int64_t)(4 * s + 1) * st + p], dx);
      acc[2] = c.z + dot4<R>(a.Phi[(int64_t)(4 * s + 2) * st + p], dx);
      acc[3] = c.w + dot4<R>(a.Phi[(int64_t)(4 * s + 3) * st + p], dx);
      for (int i = 0; i < SP; ++i, ++kk) {
        const V4 W = W_nx, T = T_nx, G = G_nx;
        if (kk + 1 < N) {
          W_nx = w_p[(int64_t)(kk + 1) * st];
          T_nx = t_p[(int64_t)(kk + 1) * st];
          G_nx = g_p[(int64_t)(kk + 1) * st];
        }
        const R y = -(T.x + dot4<R>(W, q));
        const R du = y * T.z - ups_prev * du_prev;
        a.dzu[(int64_t)kk * st + p] = du;
        acc[0] += G.x * du;
        acc[1] += G.y * du;
        acc[2] += G.z * du;
        acc[3] += G.w * du;
        gd += T.w * du;
        const R jd = a.wd * (du_prev - du);  // rows (u_{k-1} - u_k) w and, for k = 0, (u_0 - u_prev) w
        curv += wu2 * du * du + jd * jd + lam * du * du;
        du_prev = du;
        ups_prev = T.y;
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) dx[t] = acc[t];
      a.dzx[(int64_t)(s + 1) * st + p] = mk4<R>(dx[0], dx[1], dx[2], dx[3]);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if ((a.term_is_cost >> t) & 1) {
        const R jd = a.term_w[t] * dx[t];
        gd += (a.term_w[t] * e_term[t]) * jd;
        curv += jd * jd;
      }
    }
  }
  if (status == kTermNone && (!Math<R>::finite(gd) || !Math<R>::finite(curv))) status = kTermQpIndefinite;

  // ---- penalty update (Nocedal & Wright 18.36, sigma = 1), merit slope --------------------------
  if (cn > R(0)) {
    const R mu_req = (gd + R(0.5) * curv) / ((R(1) - a.rho) * cn);
    if (mu < mu_req) mu = mu_req;
  }
  const R D = gd - mu * cn;
  const R phi0 = f + mu * cn;
  const bool first_order = Math<R>::fabs(D) < a.fo_tol;

  // ---- Armijo line search, lock-step over the wave ----------------------------------------------
  bool active = (status == kTermNone);
  bool accepted = false;
  R alpha = a_start, phi_t = R(0), f_t = f, cn_t = cn;
  int evals = 0;
  for (int t = 0; t < a.max_ls; ++t) {
    if (!__any(active)) break;
    if (active) {
      R ft, ct;
      merit_eval<R>(a, k, p, alpha, xm, tgt, u_prev, ft, ct);
      ++evals;
      phi_t = ft + mu * ct;
      if (phi_t <= phi0 + a.c1 * alpha * D) {
        accepted = true;
        active = false;
        f_t = ft;
        cn_t = ct;
      } else {
        const R denom = R(2) * (phi_t - phi0 - D * alpha);
        R a_new = (denom > R(0)) ? (-D * alpha * alpha / denom) : (a.shrink_max * alpha);
        if (!(a_new >= a.shrink_min * alpha)) a_new = a.shrink_min * alpha;
        if (a_new > a.shrink_max * alpha) a_new = a.shrink_max * alpha;
        alpha = a_new;
      }
    }
  }

  // ---- accept / reject, step-length memory, damping schedule, termination -----------------------
  int failed = a.ist[IS_FAILED * st + p];
  if (status == kTermNone) {
    a_start = R(1);
    if (accepted && a.alpha_growth > R(0)) {
      a_start = a.alpha_growth * alpha;
      if (!(a_start < R(1))) a_start = R(1);
    }
    if (accepted) {
      for (int s = 0; s < S; ++s) {
        const int64_t idx = (int64_t)s * st + p;
        const V4 zv = a.zx[idx];
        const V4 dv = a.dzx[idx];
        a.zx[idx] = mk4<R>(clampr(zv.x + alpha * dv.x, -a.bx_lim, a.bx_lim), mod_pi(zv.y + alpha * dv.y),
                           zv.z + alpha * dv.z, zv.w + alpha * dv.w);
      }
      for (int i = 0; i < N; ++i) {
        const int64_t idx = (int64_t)i * st + p;
        a.zu[idx] = clampr(a.zu[idx] + alpha * a.dzu[idx], -a.u_lim, a.u_lim);
      }
      lam *= a.lam_down;
      if (lam < a.lam_min) lam = R(0);
    }
    if (first_order) {
      status = kTermFirstOrder;
    } else if (accepted) {
      if ((phi0 - phi_t) < a.rel_tol * phi0) status = kTermRelTol;
    } else {
      ++failed;
      lam = (lam > R(0)) ? lam * a.lam_up : a.lam_fail_init;
      if (lam > a.lam_max) status = kTermMaxLambda;
    }
  }
  if (status != kTermNonFinite) a.ist[IS_ITERS * st + p] += 1;
  a.sc[SC_LAMBDA * st + p] = lam;
  a.sc[SC_MU * st + p] = mu;
  a.sc[SC_F_LAST * st + p] = f_t;
  a.sc[SC_CN_LAST * st + p] = cn_t;
  a.sc[SC_ALPHA * st + p] = a_start;
  a.ist[IS_STATUS * st + p] = status;
  a.ist[IS_LS_EVALS * st + p] += evals;
  a.ist[IS_FAILED * st + p] = failed;
}

// ------------------------------------------------------------------------------------------------
// finalize: predicted states (optimization.cc:353-371) and outputs.  One thread per problem.
// ------------------------------------------------------------------------------------------------
template <typename R>
__global__ __launch_bounds__(64) void finalize_kernel(const SolverArgs<R> a) {
  const unsigned p = blockIdx.x * 64u + threadIdx.x;
  if (p >= a.B) return;
  const int64_t st = a.stride;
  const int64_t ob = a.B;  // outputs are packed [field][B]
  int status = a.ist[IS_STATUS * st + p];
  if (status == kTermNone) status = kTermMaxIterations;
  if (a.status_out) a.status_out[p] = status;
  if (a.iters_out) a.iters_out[p] = a.ist[IS_ITERS * st + p];
  if (a.ls_out) a.ls_out[p] = a.ist[IS_LS_EVALS * st + p];
  if (a.cost_out) a.cost_out[p] = a.sc[SC_F_LAST * st + p];
  if (a.eq_out) a.eq_out[p] = a.sc[SC_CN_LAST * st + p];
  if (a.pred_out) {
    const CartPoleConsts<R> k = load_consts(a, p);
    const ExtForce<R> fe{R(0), R(0), R(0)};
    R x[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) x[t] = a.x0[t * ob + p];
    R u_next = a.zu[p];
    for (int kk = 0; kk < a.N; ++kk) {
      const R u = u_next;
      if (kk + 1 < a.N) u_next = a.zu[(int64_t)(kk + 1) * st + p];
      if (a.u_out) a.u_out[(int64_t)kk * ob + p] = u;
      rk4_step<R, false>(k, a.dt, x, u, fe);
      x[1] = mod_pi(x[1]);
#pragma unroll
      for (int t = 0; t < 4; ++t) a.pred_out[((int64_t)kk * 4 + t) * ob + p] = x[t];
    }
  } else if (a.u_out) {
    for (int kk = 0; kk < a.N; ++kk) a.u_out[(int64_t)kk * ob + p] = a.zu[(int64_t)kk * st + p];
  }
}

// ------------------------------------------------------------------------------------------------
// layout conversion between the packed external z [4S+N][B] (MapKey order) and the workspace
// ------------------------------------------------------------------------------------------------
template <typename R>
__global__ __launch_bounds__(64) void pack_z_kernel(int64_t B, int64_t st, int S, int N, const R* z_ext,
                                                     typename VecT<R>::V4* zx, R* zu) {
  const unsigned p = blockIdx.x * 64u + threadIdx.x;
  if (p >= B) return;
  for (int s = 0; s < S; ++s)
    zx[(int64_t)s * st + p] = mk4<R>(z_ext[(int64_t)(4 * s + 0) * B + p], z_ext[(int64_t)(4 * s + 1) * B + p],
                                     z_ext[(int64_t)(4 * s + 2) * B + p], z_ext[(int64_t)(4 * s + 3) * B + p]);
  for (int i = 0; i < N; ++i) zu[(int64_t)i * st + p] = z_ext[(int64_t)(4 * S + i) * B + p];
}

template <typename R>
__global__ __launch_bounds__(64) void unpack_z_kernel(int64_t B, int64_t st, int S, int N,
                                                       const typename VecT<R>::V4* zx, const R* zu,
                                                       R* z_ext) {
  const unsigned p = blockIdx.x * 64u + threadIdx.x;
  if (p >= B) return;
  for (int s = 0; s < S; ++s) {
    const typename VecT<R>::V4 v = zx[(int64_t)s * st + p];
    z_ext[(int64_t)(4 * s + 0) * B + p] = v.x;
    z_ext[(int64_t)(4 * s + 1) * B + p] = v.y;
    z_ext[(int64_t)(4 * s + 2) * B + p] = v.z;
    z_ext[(int64_t)(4 * s + 3) * B + p] = v.w;
  }
  for (int i = 0; i < N; ++i) z_ext[(int64_t)(4 * S + i) * B + p] = zu[(int64_t)i * st + p];
}

// workspace linearisation -> packed c [4(S-1)][B], Phi [16(S-1)][B] (row-major), Gamma [4N][B] (4k+r)
template <typename R>
__global__ __launch_bounds__(64) void unpack_lin_kernel(const SolverArgs<R> a, R* c, R* Phi, R* Gam) {
  const unsigned p = blockIdx.x * 64u + threadIdx.x;
  if (p >= a.B) return;
  const int64_t st = a.stride, B = a.B;
  for (int s = 0; s + 1 < a.S; ++s) {
    const typename VecT<R>::V4 v = a.cs[(int64_t)s * st + p];
    c[(int64_t)(4 * s + 0) * B + p] = v.x;
    c[(int64_t)(4 * s + 1) * B + p] = v.y;
    c[(int64_t)(4 * s + 2) * B + p] = v.z;
    c[(int64_t)(4 * s + 3) * B + p] = v.w;
    for (int r = 0; r < 4; ++r) {
      const typename VecT<R>::V4 row = a.Phi[(int64_t)(4 * s + r) * st + p];
      Phi[(int64_t)(16 * s + 4 * r + 0) * B + p] = row.x;
      Phi[(int64_t)(16 * s + 4 * r + 1) * B + p] = row.y;
      Phi[(int64_t)(16 * s + 4 * r + 2) * B + p] = row.z;
      Phi[(int64_t)(16 * s + 4 * r + 3) * B + p] = row.w;
    }
  }
  for (int i = 0; i < a.N; ++i) {
    const typename VecT<R>::V4 g = a.Gam[(int64_t)i * st + p];
    Gam[(int64_t)(4 * i + 0) * B + p] = g.x;
    Gam[(int64_t)(4 * i + 1) * B + p] = g.y;
    Gam[(int64_t)(4 * i + 2) * B + p] = g.z;
    Gam[(int64_t)(4 * i + 3) * B + p] = g.w;
  }
}

// ------------------------------------------------------------------------------------------------
// stand-alone pieces (parity tests, callers that want them)
// ------------------------------------------------------------------------------------------------
template <typename R>
__global__ __launch_bounds__(64) void dynamics_kernel(int64_t B, CartPoleConsts<R> k,
                                                       ExtForce<R> fe, const R* x, const R* u,
                                                       R* f, R* Jx, R* Ju) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= B) return;
  R ax, at, Ja[2][4], Jua[2];
  const R x0 = x[p], x1 = x[B + p], x2 = x[2 * B + p], x3 = x[3 * B + p];
  cartpole_accel<R, true, true>(k, x0, x1, x2, x3, u[p], fe, ax, at, Ja, Jua);
  f[p] = x2;
  f[B + p] = x3;
  f[2 * B + p] = ax;
  f[3 * B + p] = at;
  if (Jx) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      Jx[(0 * 4 + c) * B + p] = (c == 2) ? R(1) : R(0);
      Jx[(1 * 4 + c) * B + p] = (c == 3) ? R(1) : R(0);
      Jx[(2 * 4 + c) * B + p] = Ja[0][c];
      Jx[(3 * 4 + c) * B + p] = Ja[1][c];
    }
  }
  if (Ju) {
    Ju[p] = R(0);
    Ju[B + p] = R(0);
    Ju[2 * B + p] = Jua[0];
    Ju[3 * B + p] = Jua[1];
  }
}

template <typename R>
__global__ __launch_bounds__(64) void rk4_kernel(int64_t B, CartPoleConsts<R> k, ExtForce<R> fe,
                                                  R h, const R* x, const R* u, R* xn, R* A, R* Bm) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= B) return;
  R xs[4] = {x[p], x[B + p], x[2 * B + p], x[3 * B + p]};
  if (A != nullptr || Bm != nullptr) {
    R Am[4][4], Bv[4];
    rk4_step_jac<R, true>(k, h, xs, u[p], fe, Am, Bv);
    if (A)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) A[(r * 4 + c) * B + p] = Am[r][c];
    if (Bm)
#pragma unroll
      for (int r = 0; r < 4; ++r) Bm[r * B + p] = Bv[r];
  } else {
    rk4_step<R, true>(k, h, xs, u[p], fe);
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) xn[t * B + p] = xs[t];
}

// Simulator::Step (simulator.cc:11-36): fixed 1 ms sub-steps, angle wrapped after each.
template <typename R>
__global__ __launch_bounds__(64) void sim_kernel(int64_t B, CartPoleConsts<R> k, ExtForce<R> fe_shared,
                                                  const R* fext, int n_sub, R h_last, const R* u,
                                                  R* state) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= B) return;
  ExtForce<R> fe = fe_shared;
  if (fext) {
    fe.fbx = fext[p];
    fe.fmx = fext[2 * B + p];
    fe.fmy = fext[3 * B + p];
  }
  R xs[4] = {state[p], state[B + p], state[2 * B + p], state[3 * B + p]};
  const R uu = u[p];
  // the host evaluates the reference's `while (dt > 0) { SubStep(min(dt, 0.001)); dt -= 0.001; }`
  // in double and passes the count and the last step, so f32 and f64 take the same sub-steps
  const R internal_dt = R(0.001);
  for (int i = 0; i < n_sub; ++i) {
    const R h = (i + 1 == n_sub) ? h_last : internal_dt;
    rk4_step<R, true>(k, h, xs, uu, fe);
    xs[1] = mod_pi(xs[1]);
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) state[t * B + p] = xs[t];
}

}  // namespace cpmpc
