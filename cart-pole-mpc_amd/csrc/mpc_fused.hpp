// mpc_fused.hpp -- all SQP iterations of a batch in ONE kernel, a problem spread over L = S-1 lanes.
//
// The split pipeline (mpc_kernels.hpp) streams the sensitivities Gamma/Phi and the QP factors W/T through
// HBM between its kernels (measured 6.6 KB per problem and iteration, 40 % of the HBM peak, profiles/).
// Here lane s of a group of L consecutive lanes owns shooting interval s of one problem for the whole
// solve: its node x_s and Phi_s live in registers, its SP controls, the step and Gamma_s in lane-private LDS,
// the tridiagonal factors in registers; nothing of it reaches memory.  Per iteration:
//   linearise      every lane integrates its own interval (RK4 + forward sensitivities), in parallel
//   sweep 1        T = U D U^T and W = U^-1 R^T run DOWN the intervals: lane s works in turn s and hands
//                  {Psi, w_{k+1}, gw_{k+1}, d_{k+1}} to lane s-1 through wave shuffles; S, rho, the free
//                  response, f and |c|_1 are per-lane partial sums, reduced over the group afterwards
//   NX x NX LDL^T  every lane of the group solves the same small system (bitwise identical inputs)
//   sweep 1b       with q known, W q is a scalar recurrence  omega_k = psi . Gamma_k - ups_k omega_{k+1},
//                  psi = Psi^T q, again DOWN the intervals (W itself is never stored)
//   sweep 2        U^T du = D^-1 y and the state recovery run UP the intervals the same way
//   line search    every lane rolls out its own interval of the trial point; the merit is a group sum
// 64/L problems share a wave, so the lock-step line search pays the maximum trial count over 16 (L = 4)
// problems instead of 64.  Only the sequential sweeps leave lanes idle (about a fifth of the instructions).
// prepare_kernel / finalize_kernel of mpc_kernels.hpp run before / after, unchanged: the workspace fields
// zx, zu (iterate = warm start), sc and ist are this kernel's only global traffic.
//
// The arithmetic is the split pipeline's, operation for operation; group sums use a butterfly whose
// result is bitwise identical in every lane of the group.
#pragma once
#include "mpc_kernels.hpp"

namespace cpmpc {

template <typename R, int L>
__device__ __forceinline__ R group_sum(R v) {
#pragma unroll
  for (int off = 1; off < L; off <<= 1) v += __shfl_xor(v, off);
  return v;
}
// value held by the lane `delta` positions to the right (+) / left (-); callers ignore it at the group edge
template <typename R>
__device__ __forceinline__ R from_lane(R v, int src) {
  return __shfl(v, src);
}

#ifndef CPMPC_FUSED_WAVES
#define CPMPC_FUSED_WAVES 0
#endif
#if CPMPC_FUSED_WAVES > 0
#define CPMPC_FUSED_BOUNDS __launch_bounds__(64, CPMPC_FUSED_WAVES)
#else
#define CPMPC_FUSED_BOUNDS __launch_bounds__(64)
#endif

// SHARED: the batch shares one parameter set -> the model constants stay wave-uniform (scalar registers)
template <typename R, typename M, int SP, int L, bool SHARED>
__global__ CPMPC_FUSED_BOUNDS void fused_sqp_kernel(const SolverArgs<R, M> a, const int max_iters) {
  constexpr int NX = M::NX;
  constexpr int PPW = 64 / L;  // problems per wave
  __shared__ R lds_u[SP * 64];
  __shared__ R lds_du[SP * 64];
  __shared__ XV<R, NX> lds_G[SP * 64];  // column i of my Gamma_s at [i*64 + lane]
  __shared__ R lds_gw[SP * 64];         // (U^-1 g)_k of my controls
  __shared__ R lds_id[SP * 64];         // 1/d_k of my controls
  const int lane = threadIdx.x;
  const int s = lane % L;            // my shooting interval
  const int gbase = lane - s;        // first lane of my group
  int64_t pp = (int64_t)blockIdx.x * PPW + lane / L;
  const bool valid = pp < a.B;
  if (!valid) pp = a.B - 1;          // compute redundantly, never store: keeps the shuffles well defined
  const unsigned p = (unsigned)pp;
  const int64_t st = a.stride;
  const int N = a.N;
  typename M::Consts k_lane;
  if constexpr (!SHARED) k_lane = load_consts(a, p);
  const typename M::Consts& k = SHARED ? a.consts : k_lane;
  const ExtForce<R> fe{R(0), R(0), R(0)};
  const R wu2 = a.wu * a.wu, wd2 = a.wd * a.wd;
  const int right = (s + 1 < L) ? lane + 1 : lane;
  const int left = (s > 0) ? lane - 1 : lane;

  // ---- per-problem state, replicated in the L lanes of the group -------------------------------------
  int status = a.ist[IS_STATUS * st + p];
  int iters = a.ist[IS_ITERS * st + p], evals_tot = a.ist[IS_LS_EVALS * st + p], failed = a.ist[IS_FAILED * st + p];
  R lam = a.sc[SC_LAMBDA * st + p], mu = a.sc[SC_MU * st + p], a_start = a.sc[SC_ALPHA * st + p];
  R f_last = a.sc[SC_F_LAST * st + p], cn_last = a.sc[SC_CN_LAST * st + p];
  const R u_prev = a.sc[SC_UPREV * st + p];
  R tgt[NX], xm[NX], Rw[NX], Dg[NX];
#pragma unroll
  for (int t = 0; t < NX; ++t) {
    tgt[t] = a.term_tgt[t];
    xm[t] = a.x0[t * a.B + p];
    Rw[t] = a.term_w[t];
    Dg[t] = ((a.term_is_cost >> t) & 1) ? R(1) : R(0);
  }
  if (a.set_point) tgt[0] = a.set_point[p];

  // ---- my piece of the iterate --------------------------------------------------------------------------
  R xs[NX], xe[NX];  // node s, node s+1 (a copy, or the terminal node itself for s = L-1)
  unpack<R, NX>(a.zx[(int64_t)s * st + p], xs);
  unpack<R, NX>(a.zx[(int64_t)(s + 1) * st + p], xe);
#pragma unroll
  for (int i = 0; i < SP; ++i) lds_u[i * 64 + lane] = a.zu[(int64_t)(s * SP + i) * st + p];

  for (int it = 0; it < max_iters; ++it) {
    if (!__any(status == kTermNone)) break;
    const bool live = (status == kTermNone);

    // ================= linearise my interval (optimization.cc:99-160) ==================================
    R Phi[NX][NX], cdef[NX];
    {
      R x[NX];
#pragma unroll
      for (int r = 0; r < NX; ++r) {
        x[r] = xs[r];
#pragma unroll
        for (int c = 0; c < NX; ++c) Phi[r][c] = (r == c) ? R(1) : R(0);
      }
#pragma unroll 1
      for (int i = 0; i < SP; ++i) {
        const R u = lds_u[i * 64 + lane];
        R A[NX][NX], Bv[NX];
        rk4_step_jac_m<R, M, false>(k, a.dt, x, u, fe, A, Bv);
        R T[NX][NX];
#pragma unroll
        for (int r = 0; r < NX; ++r)
#pragma unroll
          for (int c = 0; c < NX; ++c) {
            R acc = A[r][0] * Phi[0][c];
#pragma unroll
            for (int m = 1; m < NX; ++m) acc += A[r][m] * Phi[m][c];
            T[r][c] = acc;
          }
#pragma unroll
        for (int r = 0; r < NX; ++r)
#pragma unroll
          for (int c = 0; c < NX; ++c) Phi[r][c] = T[r][c];
        // Gamma_j <- A Gamma_j for the earlier controls, Gamma_i = B (forward accumulation, in LDS)
#pragma unroll 1
        for (int j = 0; j < i; ++j) {
          R g[NX], gn[NX];
          unpack<R, NX>(lds_G[j * 64 + lane], g);
#pragma unroll
          for (int r = 0; r < NX; ++r) {
            R acc = A[r][0] * g[0];
#pragma unroll
            for (int m = 1; m < NX; ++m) acc += A[r][m] * g[m];
            gn[r] = acc;
          }
          lds_G[j * 64 + lane] = pack<R, NX>(gn);
        }
        lds_G[i * 64 + lane] = pack<R, NX>(Bv);
      }
      wrap_angles<R, M>(x);
#pragma unroll
      for (int t = 0; t < NX; ++t) cdef[t] = x[t] - xe[t];
      wrap_angles<R, M>(cdef);
    }

    // neighbours' boundary controls (u_{k-1} of my first control, u_{k+1} of my last one)
    const R u_first = lds_u[0 * 64 + lane], u_last = lds_u[(SP - 1) * 64 + lane];
    R u_left = from_lane(u_last, left);
    if (s == 0) u_left = u_prev;
    const R u_right = from_lane(u_first, right);  // unused for s = L-1

    // ================= sweep 1, down the intervals =====================================================
    // Systolic: in each of the L passes EVERY lane runs its block from its carry-in {Psi, w_{k+1}, gw_{k+1},
    // d_{k+1}} and then takes its right neighbour's carry-out.  Lane L-1's carry-in is the constant start, so
    // after pass j the lanes L-1 ... L-j hold their final block (recomputing with an unchanged carry-in is
    // idempotent).  Per-control results (gw_k, 1/d_k) go to lane-private LDS, so the loops stay rolled and
    // there are no lane-conditional register writes.
    R Sm[NX][NX], rho[NX], ha[NX];
    R f_part = R(0), cn_part = R(0);
    bool pd_ok = true;
    R Psi_in[NX][NX], wprev_in[NX], gw_in = R(0), d_in = R(1);    // carry-in  (from interval s+1)
    R Psi[NX][NX], wprev[NX], gwprev = R(0), d_next = R(1);       // carry-out (to interval s-1)
#pragma unroll
    for (int r = 0; r < NX; ++r) {
      wprev_in[r] = R(0);
#pragma unroll
      for (int c = 0; c < NX; ++c) Psi_in[r][c] = (r == c) ? Rw[r] : R(0);
    }
#pragma unroll 1
    for (int pass = 0; pass < L; ++pass) {
      // block-local accumulators: the last pass is the one that counts
      f_part = R(0);
      cn_part = R(0);
      pd_ok = true;
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        rho[i] = R(0);
        wprev[i] = wprev_in[i];
#pragma unroll
        for (int j = 0; j < NX; ++j) {
          Sm[i][j] = R(0);
          Psi[i][j] = Psi_in[i][j];
        }
      }
      gwprev = gw_in;
      d_next = d_in;
      R u_hi = u_right;
      R u_cur = lds_u[(SP - 1) * 64 + lane];
#pragma unroll 1
      for (int i = SP - 1; i >= 0; --i) {
        const int kk = s * SP + i;
        const R u_lo = (i > 0) ? lds_u[(i - 1) * 64 + lane] : u_left;
        const R ru = a.wu * u_cur, rd = a.wd * (u_lo - u_cur);
        f_part += ru * ru + rd * rd;
        const R nd = (kk < N - 1 ? R(1) : R(0)) + R(1);
        const R diag = wu2 + lam + wd2 * nd;
        R g = wu2 * u_cur + wd2 * (u_cur - u_lo);
        if (kk < N - 1) g += wd2 * (u_cur - u_hi);
        const R ups = (kk < N - 1) ? (-wd2 / d_next) : R(0);
        const R dk = diag + wd2 * ups;
        if (!(dk > R(0))) pd_ok = false;
        const R inv_d = R(1) / dk;
        d_next = dk;
        R wk[NX], Gi[NX];
        unpack<R, NX>(lds_G[i * 64 + lane], Gi);
#pragma unroll
        for (int r = 0; r < NX; ++r) {
          R m = Psi[r][0] * Gi[0];
#pragma unroll
          for (int c = 1; c < NX; ++c) m += Psi[r][c] * Gi[c];
          wk[r] = m - ups * wprev[r];
        }
        const R gw = g - ups * gwprev;
        lds_gw[i * 64 + lane] = gw;
        lds_id[i * 64 + lane] = inv_d;
#pragma unroll
        for (int i2 = 0; i2 < NX; ++i2) {
          const R wi = wk[i2] * inv_d;
          rho[i2] += wi * gw;
#pragma unroll
          for (int j2 = 0; j2 <= i2; ++j2) Sm[i2][j2] += wi * wk[j2];
        }
#pragma unroll
        for (int r = 0; r < NX; ++r) wprev[r] = wk[r];
        gwprev = gw;
        u_hi = u_cur;
        u_cur = u_lo;
      }
      // my defect: |c|_1 and Psi_s c_s; then Psi <- Psi Phi_s
#pragma unroll
      for (int t = 0; t < NX; ++t) cn_part += Math<R>::fabs(cdef[t]);
#pragma unroll
      for (int r = 0; r < NX; ++r) {
        R acc = Psi[r][0] * cdef[0];
#pragma unroll
        for (int m = 1; m < NX; ++m) acc += Psi[r][m] * cdef[m];
        ha[r] = acc;
      }
      {
        R T[NX][NX];
#pragma unroll
        for (int r = 0; r < NX; ++r)
#pragma unroll
          for (int c = 0; c < NX; ++c) {
            R acc = Psi[r][0] * Phi[0][c];
#pragma unroll
            for (int m = 1; m < NX; ++m) acc += Psi[r][m] * Phi[m][c];
            T[r][c] = acc;
          }
#pragma unroll
        for (int r = 0; r < NX; ++r)
#pragma unroll
          for (int c = 0; c < NX; ++c) Psi[r][c] = T[r][c];
      }
      // take the right neighbour's carry-out as my carry-in (the last interval keeps the constant start)
      if (pass + 1 < L) {
        const bool edge = (s == L - 1);
#pragma unroll
        for (int r = 0; r < NX; ++r) {
#pragma unroll
          for (int c = 0; c < NX; ++c) {
            const R v = from_lane(Psi[r][c], right);
            Psi_in[r][c] = edge ? Psi_in[r][c] : v;
          }
          const R v = from_lane(wprev[r], right);
          wprev_in[r] = edge ? wprev_in[r] : v;
        }
        const R v1 = from_lane(gwprev, right), v2 = from_lane(d_next, right);
        gw_in = edge ? gw_in : v1;
        d_in = edge ? d_in : v2;
      }
    }
    // upsilon_k = -wd2 / d_{k+1}: inside a block from my own 1/d, at its end from my right neighbour's first
    const R id_right = from_lane(lds_id[0 * 64 + lane], right);  // unused for s = L-1

    // ---- initial-state rows (node 0), terminal rows (node S-1), group sums ---------------------------------
    R ci[NX], e_term[NX];
    {
      R x0n[NX], xT[NX];
#pragma unroll
      for (int t = 0; t < NX; ++t) {
        x0n[t] = from_lane(xs[t], gbase);
        xT[t] = from_lane(xe[t], gbase + L - 1);
        ci[t] = x0n[t] - xm[t];
        e_term[t] = xT[t] - tgt[t];
      }
      wrap_angles<R, M>(ci);
      wrap_angles<R, M>(e_term);
    }
    if (s == 0) {  // Psi in lane 0 is now diag(w) Phi_{S-2}...Phi_0: contribution of dx_0 = -c_init
#pragma unroll
      for (int r = 0; r < NX; ++r) {
        R acc = Psi[r][0] * ci[0];
#pragma unroll
        for (int m = 1; m < NX; ++m) acc += Psi[r][m] * ci[m];
        ha[r] -= acc;
      }
    }
    R f = group_sum<R, L>(f_part), cn = group_sum<R, L>(cn_part);
    R hv[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      rho[i] = group_sum<R, L>(rho[i]);
      hv[i] = group_sum<R, L>(ha[i]);
#pragma unroll
      for (int j = 0; j <= i; ++j) Sm[i][j] = group_sum<R, L>(Sm[i][j]);
    }
    pd_ok = (group_sum<R, L>(pd_ok ? R(0) : R(1)) == R(0));  // AND over the group
#pragma unroll
    for (int t = 0; t < NX; ++t) {
      cn += Math<R>::fabs(ci[t]);
      if ((a.term_is_cost >> t) & 1) {
        const R r = a.term_w[t] * e_term[t];
        f += r * r;
      } else {
        cn += Math<R>::fabs(e_term[t]);
      }
      hv[t] += Rw[t] * e_term[t];
    }
    f *= R(0.5);
    if (live && (!Math<R>::finite(f) || !Math<R>::finite(cn))) status = kTermNonFinite;

    // ================= NX x NX LDL^T (every lane of the group, identical inputs) ==========================
    R q[NX];
    {
      R Lm[NX][NX], dv[NX];
#pragma unroll
      for (int i = 0; i < NX; ++i) Sm[i][i] += Dg[i];
#pragma unroll
      for (int j = 0; j < NX; ++j) {
        R dj = Sm[j][j];
#pragma unroll
        for (int m = 0; m < j; ++m) dj -= Lm[j][m] * Lm[j][m] * dv[m];
        if (!(dj > R(0))) pd_ok = false;
        dv[j] = dj;
        const R inv = R(1) / dj;
#pragma unroll
        for (int i = j + 1; i < NX; ++i) {
          R v = Sm[i][j];
#pragma unroll
          for (int m = 0; m < j; ++m) v -= Lm[i][m] * Lm[j][m] * dv[m];
          Lm[i][j] = v * inv;
        }
      }
      R y[NX];
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        R v = hv[i] - rho[i];
#pragma unroll
        for (int m = 0; m < i; ++m) v -= Lm[i][m] * y[m];
        y[i] = v;
      }
#pragma unroll
      for (int i = NX - 1; i >= 0; --i) {
        R v = y[i] / dv[i];
#pragma unroll
        for (int m = i + 1; m < NX; ++m) v -= Lm[m][i] * q[m];
        q[i] = v;
      }
    }
    if (live && status == kTermNone && !pd_ok) status = kTermQpIndefinite;

    // ================= sweep 1b, down the intervals: v_k = D^-1 U^-1 (-(g + R^T q)) without storing W ======
    //   W_k q = omega_k,  omega_k = psi_s . Gamma_k - ups_k omega_{k+1},  psi_s = Psi_s^T q,  psi_{s-1} = Phi_s^T psi_s
    //   v_k is parked in lds_du[k] (sweep 2 turns it into du_k in place)
    {
      R psi_in[NX], om_in = R(0);
#pragma unroll
      for (int r = 0; r < NX; ++r) psi_in[r] = Rw[r] * q[r];
#pragma unroll 1
      for (int pass = 0; pass < L; ++pass) {
        R om = om_in;
        R id_next = id_right;  // 1/d_{k+1}
#pragma unroll 1
        for (int i = SP - 1; i >= 0; --i) {
          const int kk = s * SP + i;
          R Gi[NX];
          unpack<R, NX>(lds_G[i * 64 + lane], Gi);
          R pg = psi_in[0] * Gi[0];
#pragma unroll
          for (int m = 1; m < NX; ++m) pg += psi_in[m] * Gi[m];
          const R idk = lds_id[i * 64 + lane];
          const R ups = (kk < N - 1) ? (-wd2 * id_next) : R(0);
          om = pg - ups * om;
          lds_du[i * 64 + lane] = -(lds_gw[i * 64 + lane] + om) * idk;
          id_next = idk;
        }
        R psi_out[NX];
#pragma unroll
        for (int c = 0; c < NX; ++c) {
          R acc = Phi[0][c] * psi_in[0];
#pragma unroll
          for (int r = 1; r < NX; ++r) acc += Phi[r][c] * psi_in[r];
          psi_out[c] = acc;
        }
        if (pass + 1 < L) {
          const bool edge = (s == L - 1);
#pragma unroll
          for (int r = 0; r < NX; ++r) {
            const R v = from_lane(psi_out[r], right);
            psi_in[r] = edge ? psi_in[r] : v;
          }
          const R v1 = from_lane(om, right);
          om_in = edge ? om_in : v1;
        }
      }
    }

    // ================= sweep 2, up the intervals ==========================================================
    // Each lane works once, in its turn (v_k -> du_k happens in place in LDS, so no recomputation here); it
    // hands {dx_{s+1}, du, upsilon of its last control} to its right neighbour.
    R dxs[NX], dxe[NX];
    R gd_part = R(0), curv_part = R(0);
    R du_prev = R(0), ups_prev = R(0);
#pragma unroll
    for (int t = 0; t < NX; ++t) {
      dxs[t] = -ci[t];  // interval 0's start; the others receive theirs from the left
      dxe[t] = R(0);
    }
#pragma unroll 1
    for (int turn = 0; turn < L; ++turn) {
      if (s == turn) {
        R acc[NX];
#pragma unroll
        for (int r = 0; r < NX; ++r) {
          R v = cdef[r];
#pragma unroll
          for (int m = 0; m < NX; ++m) v += Phi[r][m] * dxs[m];
          acc[r] = v;
        }
        R u_lo = u_left;
        R u_cur = lds_u[0 * 64 + lane];
#pragma unroll 1
        for (int i = 0; i < SP; ++i) {
          const int kk = s * SP + i;
          const R du = lds_du[i * 64 + lane] - ups_prev * du_prev;
          lds_du[i * 64 + lane] = du;
          R Gi[NX];
          unpack<R, NX>(lds_G[i * 64 + lane], Gi);
#pragma unroll
          for (int r = 0; r < NX; ++r) acc[r] += Gi[r] * du;
          // control-cost gradient g_k, recomputed from u exactly as in sweep 1
          const R u_hi = (i + 1 < SP) ? lds_u[(i + 1) * 64 + lane] : u_right;
          R g = wu2 * u_cur + wd2 * (u_cur - u_lo);
          if (kk < N - 1) g += wd2 * (u_cur - u_hi);
          gd_part += g * du;
          const R jd = a.wd * (du_prev - du);
          curv_part += wu2 * du * du + jd * jd + lam * du * du;
          du_prev = du;
          // upsilon_k = -wd2 / d_{k+1}
          const R id_next = (i + 1 < SP) ? lds_id[(i + 1) * 64 + lane] : id_right;
          ups_prev = (kk < N - 1) ? (-wd2 * id_next) : R(0);
          u_lo = u_cur;
          u_cur = u_hi;
        }
#pragma unroll
        for (int t = 0; t < NX; ++t) dxe[t] = acc[t];
      }
      if (turn + 1 < L) {
        const bool take = (s == turn + 1);
#pragma unroll
        for (int t = 0; t < NX; ++t) {
          const R v = from_lane(dxe[t], left);
          dxs[t] = take ? v : dxs[t];
        }
        const R v1 = from_lane(du_prev, left), v2 = from_lane(ups_prev, left);
        du_prev = take ? v1 : du_prev;
        ups_prev = take ? v2 : ups_prev;
      }
    }
    R gd = group_sum<R, L>(gd_part), curv = group_sum<R, L>(curv_part);
    {
      R dxT[NX];
#pragma unroll
      for (int t = 0; t < NX; ++t) dxT[t] = from_lane(dxe[t], gbase + L - 1);
#pragma unroll
      for (int t = 0; t < NX; ++t) {
        if ((a.term_is_cost >> t) & 1) {
          const R jd = a.term_w[t] * dxT[t];
          gd += (a.term_w[t] * e_term[t]) * jd;
          curv += jd * jd;
        }
      }
    }
    if (live && status == kTermNone && (!Math<R>::finite(gd) || !Math<R>::finite(curv))) status = kTermQpIndefinite;
    // du of my left neighbour's last control, for the (u_{k-1} - u_k) row of my first control
    R du_left = from_lane(lds_du[(SP - 1) * 64 + lane], left);
    if (s == 0) du_left = R(0);

    // ================= penalty, merit slope ==================================================================
    if (cn > R(0)) {
      const R mu_req = (gd + R(0.5) * curv) / ((R(1) - a.rho) * cn);
      if (mu < mu_req) mu = mu_req;
    }
    const R D = gd - mu * cn;
    const R phi0 = f + mu * cn;
    const bool first_order = Math<R>::fabs(D) < a.fo_tol;

    // ================= Armijo line search: every lane rolls out its own interval ============================
    bool active = live && (status == kTermNone);
    bool accepted = false;
    R alpha = a_start, phi_t = R(0), f_t = f, cn_t = cn;
    int evals = 0;
    for (int t = 0; t < a.max_ls; ++t) {
      if (!__any(active)) break;
      // (inactive groups ride along: their alpha is whatever it was, results are discarded)
      R x[NX], xn[NX];
#pragma unroll
      for (int r = 0; r < NX; ++r) {
        x[r] = xs[r] + alpha * dxs[r];
        xn[r] = xe[r] + alpha * dxe[r];
      }
      x[0] = clampr(x[0], -a.bx_lim, a.bx_lim);
      xn[0] = clampr(xn[0], -a.bx_lim, a.bx_lim);
      wrap_angles<R, M>(x);
      wrap_angles<R, M>(xn);
      R fp = R(0), cp = R(0);
      if (s == 0) {  // initial-state rows on the trial node 0
        R d[NX];
#pragma unroll
        for (int r = 0; r < NX; ++r) d[r] = x[r] - xm[r];
        wrap_angles<R, M>(d);
#pragma unroll
        for (int r = 0; r < NX; ++r) cp += Math<R>::fabs(d[r]);
      }
      R u_before = (s == 0) ? u_prev : clampr(u_left + alpha * du_left, -a.u_lim, a.u_lim);
#pragma unroll 1
      for (int i = 0; i < SP; ++i) {
        const R u = clampr(lds_u[i * 64 + lane] + alpha * lds_du[i * 64 + lane], -a.u_lim, a.u_lim);
        const R ru = a.wu * u, rd = a.wd * (u_before - u);
        fp += ru * ru + rd * rd;
        u_before = u;
        rk4_step_m<R, M, false>(k, a.dt, x, u, fe);
      }
      wrap_angles<R, M>(x);
      {
        R d[NX];
#pragma unroll
        for (int r = 0; r < NX; ++r) d[r] = x[r] - xn[r];
        wrap_angles<R, M>(d);
#pragma unroll
        for (int r = 0; r < NX; ++r) cp += Math<R>::fabs(d[r]);
      }
      if (s == L - 1) {  // terminal rows on the trial terminal node
        R d[NX];
#pragma unroll
        for (int r = 0; r < NX; ++r) d[r] = xn[r] - tgt[r];
        wrap_angles<R, M>(d);
#pragma unroll
        for (int r = 0; r < NX; ++r) {
          if ((a.term_is_cost >> r) & 1) {
            const R rr = a.term_w[r] * d[r];
            fp += rr * rr;
          } else {
            cp += Math<R>::fabs(d[r]);
          }
        }
      }
      const R ft = R(0.5) * group_sum<R, L>(fp);
      const R ct = group_sum<R, L>(cp);
      if (active) {
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

    // ================= accept / reject, step-length memory, damping, termination ===========================
    if (live && status == kTermNone) {
      a_start = R(1);
      if (accepted && a.alpha_growth > R(0)) {
        a_start = a.alpha_growth * alpha;
        if (!(a_start < R(1))) a_start = R(1);
      }
      if (accepted) {
#pragma unroll
        for (int r = 0; r < NX; ++r) {
          xs[r] += alpha * dxs[r];
          xe[r] += alpha * dxe[r];
        }
        xs[0] = clampr(xs[0], -a.bx_lim, a.bx_lim);
        xe[0] = clampr(xe[0], -a.bx_lim, a.bx_lim);
        wrap_angles<R, M>(xs);
        wrap_angles<R, M>(xe);
#pragma unroll
        for (int i = 0; i < SP; ++i)
          lds_u[i * 64 + lane] = clampr(lds_u[i * 64 + lane] + alpha * lds_du[i * 64 + lane], -a.u_lim, a.u_lim);
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
    if (live) {
      if (status != kTermNonFinite) iters += 1;
      evals_tot += evals;
      f_last = f_t;
      cn_last = cn_t;
    }
  }

  // ---- write the iterate (= warm start) and the per-problem solver state back -------------------------------
  if (valid) {
    a.zx[(int64_t)s * st + p] = pack<R, NX>(xs);
    if (s == L - 1) a.zx[(int64_t)L * st + p] = pack<R, NX>(xe);
#pragma unroll
    for (int i = 0; i < SP; ++i) a.zu[(int64_t)(s * SP + i) * st + p] = lds_u[i * 64 + lane];
    if (s == 0) {
      a.sc[SC_LAMBDA * st + p] = lam;
      a.sc[SC_MU * st + p] = mu;
      a.sc[SC_F_LAST * st + p] = f_last;
      a.sc[SC_CN_LAST * st + p] = cn_last;
      a.sc[SC_ALPHA * st + p] = a_start;
      a.ist[IS_STATUS * st + p] = status;
      a.ist[IS_ITERS * st + p] = iters;
      a.ist[IS_LS_EVALS * st + p] = evals_tot;
      a.ist[IS_FAILED * st + p] = failed;
    }
  }
}

}  // namespace cpmpc
