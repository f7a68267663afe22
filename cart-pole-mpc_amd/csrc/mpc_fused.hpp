// mpc_fused.hpp -- all SQP iterations of a batch in ONE kernel, a problem spread over L = S-1 lanes.
//
// The split pipeline (mpc_kernels.hpp) streams the sensitivities Gamma/Phi and the QP factors W/T through
// HBM between its kernels (measured 6.6 KB per problem and iteration, 40 % of the HBM peak, profiles/).
// Here lane s of a group of L consecutive lanes owns shooting interval s of one problem for the whole
// solve: its node x_s, Phi_s and defect live in registers; its SP controls, the step, Gamma_s and the
// per-control QP factors {1/d_k, (U^-1 g)_k} in lane-private LDS (80 scalars per lane = 20 KB per wave, so 8
// waves per CU = 2 per SIMD, which is also what the ~185 VGPRs allow); nothing of it reaches memory.
// Per iteration:
//   linearise      every lane integrates its own interval (RK4 + forward sensitivities), in parallel
//   QP sweeps      T = U D U^T, W = U^-1 R^T, gw = U^-1 g (down the horizon), then U^-T and the state recovery
//                  (up) are linear recurrences in k.  Every lane solves its own block in block-local
//                  coordinates (carry-in = 0) plus the scalar chain that says how a carry-in propagates; a
//                  chain of L-1 steps over the lanes (DPP row shifts) fixes the true block-boundary values
//                  {Psi_s, w_in, gw_in} / {dx_s, du_in}; each lane then corrects its block locally.  The
//                  pivots d_k depend on (lambda, weights) only: a Moebius recurrence whose block-boundary
//                  values come from powers of one 2x2 matrix, so no lane walks the whole horizon.
//   NX x NX LDL^T  every lane of the group solves the same small system (bitwise identical inputs: group
//                  sums are commutative butterflies over quad permutes)
//   line search    every lane rolls out its own interval of the trial point; the merit is a group sum
// 64/L problems share a wave, so the lock-step line search pays the maximum trial count over 16 (L = 4)
// problems instead of 64.  prepare_kernel / finalize_kernel of mpc_kernels.hpp run before / after,
// unchanged: the workspace fields zx, zu (iterate = warm start), sc and ist are this kernel's only global
// traffic.
//
// Same algorithm as the split pipeline (same QP, same line search, same decisions); the sweeps associate
// their sums differently, so results agree to rounding (fp64: 1e-9), not bitwise.
#pragma once
#include "mpc_kernels.hpp"

namespace cpmpc {

// ---- 2x2 matrices for the pivot recurrence (compile-time exponent, by squaring) -----------------------------
template <typename R>
struct Mat2 {
  R a, b, c, d;  // [[a, b], [c, d]]
};
template <typename R>
__device__ __forceinline__ Mat2<R> mat2_mul(const Mat2<R>& x, const Mat2<R>& y) {
  return Mat2<R>{x.a * y.a + x.b * y.c, x.a * y.b + x.b * y.d, x.c * y.a + x.d * y.c, x.c * y.b + x.d * y.d};
}
template <typename R, int E>
__device__ __forceinline__ Mat2<R> mat2_pow(const Mat2<R>& t) {
  if constexpr (E <= 0) {
    return Mat2<R>{R(1), R(0), R(0), R(1)};
  } else if constexpr (E == 1) {
    return t;
  } else if constexpr (E % 2 == 0) {
    const Mat2<R> h = mat2_pow<R, E / 2>(t);
    return mat2_mul<R>(h, h);
  } else {
    return mat2_mul<R>(mat2_pow<R, E - 1>(t), t);
  }
}

// ---- lane traffic inside a group ------------------------------------------------------------------------
// Groups are L consecutive lanes with L | 16, so they never straddle a 16-lane DPP row: neighbour moves are
// row shifts and, for L = 4 (one quad) and L = 2, sums and broadcasts are quad permutes.  DPP moves are VALU
// operand modifiers: no trip through the LDS crossbar as for ds_bpermute (__shfl), which matters in the
// boundary chains where every move is on the critical path.
template <int CTRL>
__device__ __forceinline__ int dpp32(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false);
}
template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
  return __builtin_bit_cast(float, dpp32<CTRL>(__builtin_bit_cast(int, v)));
}
template <int CTRL>
__device__ __forceinline__ double dpp(double v) {
  const long long b = __builtin_bit_cast(long long, v);
  const unsigned lo = (unsigned)dpp32<CTRL>((int)(unsigned)(b & 0xffffffffll));
  const unsigned hi = (unsigned)dpp32<CTRL>((int)(unsigned)((unsigned long long)b >> 32));
  return __builtin_bit_cast(double, (long long)(((unsigned long long)hi << 32) | lo));
}
constexpr int kDppQuadXor1 = 1 | (0 << 2) | (3 << 4) | (2 << 6);  // quad_perm:[1,0,3,2]
constexpr int kDppQuadXor2 = 2 | (3 << 2) | (0 << 4) | (1 << 6);  // quad_perm:[2,3,0,1]
constexpr int kDppRowShl1 = 0x101;                                // lane i reads lane i+1 (inside its row)
constexpr int kDppRowShr1 = 0x111;                                // lane i reads lane i-1 (inside its row)

// sum over the group, bitwise identical in all its lanes: a commutative butterfly when L is a power of two, else
// (groups of 5, 10, ... lanes) every lane adds the group's values in the same fixed order
template <typename R, int L>
__device__ __forceinline__ R group_sum(R v) {
  if constexpr (L == 4) {
    v += dpp<kDppQuadXor1>(v);
    v += dpp<kDppQuadXor2>(v);
  } else if constexpr (L == 2) {
    v += dpp<kDppQuadXor1>(v);
  } else if constexpr ((L & (L - 1)) == 0) {
#pragma unroll
    for (int off = 1; off < L; off <<= 1) v += __shfl_xor(v, off);
  } else {
    const int gb = (int)threadIdx.x - (int)threadIdx.x % L;
    R acc = __shfl(v, gb);
#pragma unroll
    for (int j = 1; j < L; ++j) acc += __shfl(v, gb + j);
    v = acc;
  }
  return v;
}
// value of my right / left neighbour lane; callers discard it at the group edge.  Groups whose size divides 16
// never straddle a DPP row; the others go through the wave shuffle.
template <typename R, int L>
__device__ __forceinline__ R lane_right(R v) {
  if constexpr (16 % L == 0) return dpp<kDppRowShl1>(v);
  else return __shfl(v, (int)threadIdx.x + 1);
}
template <typename R, int L>
__device__ __forceinline__ R lane_left(R v) {
  if constexpr (16 % L == 0) return dpp<kDppRowShr1>(v);
  else return __shfl(v, (int)threadIdx.x - 1);
}
// value held by the first / last lane of my group
template <typename R, int L>
__device__ __forceinline__ R group_first(R v, int gbase) {
  if constexpr (L == 4) return dpp<0x00>(v);                              // quad_perm:[0,0,0,0]
  else if constexpr (L == 2) return dpp<(0 | (0 << 2) | (2 << 4) | (2 << 6))>(v);  // quad_perm:[0,0,2,2]
  else return __shfl(v, gbase);
}
template <typename R, int L>
__device__ __forceinline__ R group_last(R v, int gbase) {
  if constexpr (L == 4) return dpp<0xff>(v);                              // quad_perm:[3,3,3,3]
  else if constexpr (L == 2) return dpp<(1 | (1 << 2) | (3 << 4) | (3 << 6))>(v);  // quad_perm:[1,1,3,3]
  else return __shfl(v, gbase + L - 1);
}

// Waves per SIMD the register allocator must leave room for.  The sweeps are latency chains (LDS round trips,
// lane shuffles), so a second resident wave is worth a few spilled dwords in fp32 (measured +15%); fp64 values
// take two registers each and stay at one wave.
#ifndef CPMPC_FUSED_WAVES_F32
#define CPMPC_FUSED_WAVES_F32 2
#endif
// unroll factor of the block-local sweep passes (LDS reads of several controls in flight)
#ifndef CPMPC_SWEEP_UNROLL
#define CPMPC_SWEEP_UNROLL 5
#endif
#define CPMPC_FUSED_BOUNDS __launch_bounds__(64, ((sizeof(R) == 4 && M::NX <= 4) ? CPMPC_FUSED_WAVES_F32 : 1))

// Debug build only (-DCPMPC_FUSED_TIMING): shader-clock cycles per phase, summed over waves, read back by
// cpmpc_debug_phase_cycles().  Not part of the product library.
#ifdef CPMPC_FUSED_TIMING
__device__ unsigned long long g_fused_phase_cycles[8];
#define CPMPC_TICK_INIT() unsigned long long tick_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tick_last = __builtin_readcyclecounter()
#define CPMPC_TICK(IDX)                                              \
  do {                                                               \
    __builtin_amdgcn_sched_barrier(0);                               \
    const unsigned long long tick_now = __builtin_readcyclecounter(); \
    tick_acc[IDX] += tick_now - tick_last;                           \
    tick_last = tick_now;                                            \
  } while (0)
#define CPMPC_TICK_FLUSH()                                                                       \
  do {                                                                                           \
    if (threadIdx.x == 0)                                                                        \
      for (int ti = 0; ti < 8; ++ti) atomicAdd(&g_fused_phase_cycles[ti], tick_acc[ti]);         \
  } while (0)
#else
#define CPMPC_TICK_INIT() do { } while (0)
#define CPMPC_TICK(IDX) do { } while (0)
#define CPMPC_TICK_FLUSH() do { } while (0)
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
  CPMPC_TICK_INIT();
  const int lane = threadIdx.x;
  const int s = lane % L;            // my shooting interval
  const int gbase = lane - s;        // first lane of my group
  int64_t pp = (int64_t)blockIdx.x * PPW + lane / L;
  const bool in_group = lane / L < PPW;  // 64 % L lanes at the end of the wave hold no whole group: they ride along
  if (!in_group) pp = (int64_t)blockIdx.x * PPW;
  bool valid;
  int budget = max_iters;            // SQP iterations of this launch
  if (a.active_list != nullptr) {    // a later stage: my problem comes from the compacted list of active ones
    const int64_t n_active = *a.active_count;
    if ((int64_t)blockIdx.x * PPW >= n_active) return;  // block-uniform: nothing left for this wave
    valid = in_group && pp < n_active;
    pp = a.active_list[valid ? pp : n_active - 1];
    // once the active set fits one round of resident waves, further compaction cannot shorten anything: finish
    // here (every problem stops at its own iteration cap), the stages still to come find an empty list
    if (n_active <= a.run_out_below) budget = a.iter_cap;
    // likewise when hardly anybody has stopped since the previous compaction and little is left to run (a cold
    // start with few iterations): another stage would cost a launch and buy nothing
    const int64_t n_prev = a.prev_count ? (int64_t)*a.prev_count : a.prev_total;
    if (n_active * 10 >= n_prev * 9 && a.remaining <= 2 * max_iters) budget = a.iter_cap;
  } else {
    valid = in_group && pp < a.B;
    if (pp >= a.B) pp = a.B - 1;     // compute redundantly, never store: keeps the shuffles well defined
  }
  const unsigned p = (unsigned)pp;
  const int64_t st = a.stride;
  const int N = a.N;
  typename M::Consts k_lane;
  if constexpr (!SHARED) k_lane = load_consts(a, p);
  const typename M::Consts& k = SHARED ? a.consts : k_lane;
  const ExtForce<R> fe{R(0), R(0), R(0)};
  const R wu2 = a.wu * a.wu, wd2 = a.wd * a.wd;

  // ---- per-problem state, replicated in the L lanes of the group -------------------------------------
  int status = a.ist[IS_STATUS * st + p];
  if (!in_group) status = kTermMaxIterations;  // riding lanes never keep the wave's loops alive (and never store)
  int iters = a.ist[IS_ITERS * st + p], evals_tot = a.ist[IS_LS_EVALS * st + p], failed = a.ist[IS_FAILED * st + p];
  R lam = a.sc[SC_LAMBDA * st + p], mu = a.sc[SC_MU * st + p], a_start = a.sc[SC_ALPHA * st + p];
  R f_last = a.sc[SC_F_LAST * st + p], cn_last = a.sc[SC_CN_LAST * st + p];
  const R u_prev = a.sc[SC_UPREV * st + p];
  R tgt[NX], xm[NX], Rw[NX], Dg[NX];
#pragma unroll
  for (int t = 0; t < NX; ++t) {
    tgt[t] = a.term_tgt[t];
    xm[t] = a.x0[t * a.B + p];
  }
  load_terminal<R, M>(a, p, Rw, Dg);
  if (a.set_point) tgt[0] = a.set_point[p];

  // ---- my piece of the iterate --------------------------------------------------------------------------
  R xs[NX], xe[NX];  // node s, node s+1 (a copy, or the terminal node itself for s = L-1)
  unpack<R, NX>(a.zx[(int64_t)s * st + p], xs);
  unpack<R, NX>(a.zx[(int64_t)(s + 1) * st + p], xe);
#pragma unroll
  for (int i = 0; i < SP; ++i) lds_u[i * 64 + lane] = a.zu[(int64_t)(s * SP + i) * st + p];

  for (int it = 0; it < budget; ++it) {
    const bool live = (status == kTermNone) && (iters < a.iter_cap);  // frozen once terminated or at max_iterations
    if (!__any(live)) break;

    CPMPC_TICK(7);
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

    CPMPC_TICK(0);
    // neighbours' boundary controls (u_{k-1} of my first control, u_{k+1} of my last one)
    const R u_first = lds_u[0 * 64 + lane], u_last = lds_u[(SP - 1) * 64 + lane];
    R u_left = lane_left<R, L>(u_last);
    if (s == 0) u_left = u_prev;
    const R u_right = lane_right<R, L>(u_first);  // unused for s = L-1

    // ================= block-parallel sweeps ===============================================================
    // All three sweeps of the structured QP are linear recurrences in k, so every lane first solves its own
    // block in block-local coordinates (carry-in = 0) together with the scalar chain e_k that says how a
    // carry-in would propagate; a short chain over the L lanes then fixes the true block-boundary values and
    // each lane corrects its block locally:
    //     w_k  = Psi_s wt_k + e_k w_in,      gw_k = gwt_k + e_k gw_in,      e_k = -ups_k e_{k+1}
    //     wt_k = Gamma_k - ups_k wt_{k+1},   gwt_k = g_k - ups_k gwt_{k+1}  (wt, gwt = 0 and e = 1 past the block)
    // with Psi_s = diag(w) Phi_{S-2} ... Phi_{s+1} constant inside a block.

    // ---- pivots d_k = diag_k - wd2^2 / d_{k+1} of T = U D U^T ---------------------------------------------------
    // They depend on (lambda, weights) only.  Scaled by b = wu2 + lambda + 2 wd2 the recurrence is the Moebius map
    //     delta_k = 1 - gamma / delta_{k+1},   gamma = (wd2 / b)^2 <= 1/4,   delta_{N-1} = (b - wd2) / b
    // i.e. (p, q)_k = [[1, -gamma], [1, 0]] (p, q)_{k+1} with delta = p / q: a constant 2x2 matrix whose powers
    // (by squaring; all entries stay <= 1) jump straight to the block boundaries, so no lane walks the whole
    // horizon: each one starts its own SP pivots from its boundary value inside the sweep-1 pass below.
    R id_right = R(0);  // 1/d of the first control after my block (unused for s = L-1)
    {
      const R bdiag = wu2 + lam + R(2) * wd2;
      const R rb = R(1) / bdiag;
      const R gam = (wd2 * rb) * (wd2 * rb);
      const Mat2<R> Tm{R(1), -gam, R(1), R(0)};
      const Mat2<R> Tsp = mat2_pow<R, SP>(Tm);
      const Mat2<R> Tsp1 = mat2_pow<R, SP - 1>(Tm);
      // (p, q) at the first control of the last block, then one block further down per step
      R pq0 = Tsp1.a * ((bdiag - wd2) * rb) + Tsp1.b;
      R pq1 = Tsp1.c * ((bdiag - wd2) * rb) + Tsp1.d;
      R myp = R(1), myq = R(0);
#pragma unroll
      for (int sb = L - 2; sb >= 0; --sb) {
        if (sb == s) {
          myp = pq0;
          myq = pq1;
        }
        const R n0 = Tsp.a * pq0 + Tsp.b * pq1;
        const R n1 = Tsp.c * pq0 + Tsp.d * pq1;
        // delta = p / q is scale invariant: renormalise, or (p, q) ~ 2^-k would leave the fp32 range on long horizons
        pq0 = n0 * Math<R>::rcp(n1);
        pq1 = R(1);
      }
      id_right = (myq * rb) / myp;  // 1 / (b delta)
    }
    bool pd_ok = true;

    CPMPC_TICK(1);
    // ---- sweep 1, local pass down my block ---------------------------------------------------------------
    R Sm[NX][NX], rho[NX], ha[NX];
    R f_part = R(0), cn_part = R(0);
    R wt[NX], gwt = R(0), e_blk = R(1);  // after the pass: wt_0, gwt_0, e_0 of my block
    R Psi[NX][NX], w_in[NX], gw_in = R(0);  // my block's carry-in (exact after the boundary chain)
    R ci[NX], e_term[NX];
    {
      R St[NX][NX], tt[NX], rr[NX], eps = R(0), sig = R(0);
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        wt[i] = R(0);
        tt[i] = R(0);
        rr[i] = R(0);
#pragma unroll
        for (int j = 0; j < NX; ++j) St[i][j] = R(0);
      }
      R u_hi = u_right;
      R u_cur = lds_u[(SP - 1) * 64 + lane];
      R idn = id_right;
#pragma unroll CPMPC_SWEEP_UNROLL
      for (int i = SP - 1; i >= 0; --i) {
        const int kk = s * SP + i;
        const R u_lo = (i > 0) ? lds_u[(i - 1) * 64 + lane] : u_left;
        const R ru = a.wu * u_cur, rd = a.wd * (u_lo - u_cur);
        f_part += ru * ru + rd * rd;
        R g = wu2 * u_cur + wd2 * (u_cur - u_lo);
        if (kk < N - 1) g += wd2 * (u_cur - u_hi);
        const R ups = (kk < N - 1) ? (-wd2 * idn) : R(0);
        const R dk = (wu2 + lam + wd2 * ((kk < N - 1 ? R(1) : R(0)) + R(1))) + wd2 * ups;
        if (!(dk > R(0))) pd_ok = false;
        const R idk = Math<R>::rcp(dk);
        lds_id[i * 64 + lane] = idk;
        R Gi[NX];
        unpack<R, NX>(lds_G[i * 64 + lane], Gi);
#pragma unroll
        for (int r = 0; r < NX; ++r) wt[r] = Gi[r] - ups * wt[r];
        gwt = g - ups * gwt;
        e_blk = -ups * e_blk;
        lds_gw[i * 64 + lane] = gwt;
        const R ei = e_blk * idk;
        eps += ei * e_blk;
        sig += ei * gwt;
#pragma unroll
        for (int i2 = 0; i2 < NX; ++i2) {
          const R wi = wt[i2] * idk;
          tt[i2] += wi * e_blk;
          rr[i2] += wi * gwt;
#pragma unroll
          for (int j2 = 0; j2 <= i2; ++j2) St[i2][j2] += wi * wt[j2];
        }
        idn = idk;
        u_hi = u_cur;
        u_cur = u_lo;
      }
#pragma unroll
      for (int t = 0; t < NX; ++t) cn_part += Math<R>::fabs(cdef[t]);

      CPMPC_TICK(2);
      // ---- initial-state residual (node 0) and terminal residual (node S-1), replicated ---------------------
      {
        R x0n[NX], xT[NX];
#pragma unroll
        for (int t = 0; t < NX; ++t) {
          x0n[t] = group_first<R, L>(xs[t], gbase);
          xT[t] = group_last<R, L>(xe[t], gbase);
          ci[t] = x0n[t] - xm[t];
          e_term[t] = xT[t] - tgt[t];
        }
        wrap_angles<R, M>(ci);
        wrap_angles<R, M>(e_term);
      }
      // ---- boundary chain, down the lanes: Psi_s, w_in, gw_in ---------------------------------------------
#pragma unroll
      for (int r = 0; r < NX; ++r) {
        w_in[r] = R(0);
#pragma unroll
        for (int c = 0; c < NX; ++c) Psi[r][c] = (r == c) ? Rw[r] : R(0);
      }
#pragma unroll 1
      for (int round = 0; round + 1 < L; ++round) {
        // my block's values at its first control, from my current carry-in (exact once the carry-in is)
        const bool edge = (s == L - 1);
#pragma unroll
        for (int r = 0; r < NX; ++r) {
          R wo = e_blk * w_in[r];
#pragma unroll
          for (int m = 0; m < NX; ++m) wo += Psi[r][m] * wt[m];
          const R v = lane_right<R, L>(wo);
          w_in[r] = edge ? w_in[r] : v;
        }
        {
          const R v = lane_right<R, L>(gwt + e_blk * gw_in);
          gw_in = edge ? gw_in : v;
        }
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
          for (int c = 0; c < NX; ++c) {
            const R v = lane_right<R, L>(T[r][c]);
            Psi[r][c] = edge ? Psi[r][c] : v;
          }
      }
      // ---- combine: S_s = sum_k w_k w_k^T / d_k,  rho_s = sum_k w_k gw_k / d_k,  ha_s = Psi_s c_s ----------------
      {
        R hvv[NX];
#pragma unroll
        for (int r = 0; r < NX; ++r) {
          R acc = Phi[r][0] * ci[0];
#pragma unroll
          for (int m = 1; m < NX; ++m) acc += Phi[r][m] * ci[m];
          hvv[r] = cdef[r] - ((s == 0) ? acc : R(0));  // dx_0 = -c_init enters through interval 0
        }
#pragma unroll
        for (int r = 0; r < NX; ++r) {
          R acc = Psi[r][0] * hvv[0];
#pragma unroll
          for (int m = 1; m < NX; ++m) acc += Psi[r][m] * hvv[m];
          ha[r] = acc;
        }
      }
      R PT[NX], PS[NX][NX];
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        R acc = Psi[i][0] * tt[0];
        R acr = Psi[i][0] * rr[0];
#pragma unroll
        for (int m = 1; m < NX; ++m) {
          acc += Psi[i][m] * tt[m];
          acr += Psi[i][m] * rr[m];
        }
        PT[i] = acc;
        rho[i] = acr + acc * gw_in + w_in[i] * (sig + eps * gw_in);
#pragma unroll
        for (int j = 0; j < NX; ++j) {
          R v = Psi[i][0] * (j <= 0 ? St[0][j] : St[j][0]);
#pragma unroll
          for (int m = 1; m < NX; ++m) v += Psi[i][m] * (j <= m ? St[m][j] : St[j][m]);
          PS[i][j] = v;
        }
      }
#pragma unroll
      for (int i = 0; i < NX; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {
          R v = PS[i][0] * Psi[j][0];
#pragma unroll
          for (int m = 1; m < NX; ++m) v += PS[i][m] * Psi[j][m];
          Sm[i][j] = v + PT[i] * w_in[j] + w_in[i] * (PT[j] + eps * w_in[j]);
        }
    }

    CPMPC_TICK(3);
    // ---- group sums, terminal rows ------------------------------------------------------------------------
    R f = group_sum<R, L>(f_part), cn = group_sum<R, L>(cn_part);
    pd_ok = (group_sum<R, L>(pd_ok ? R(0) : R(1)) == R(0));  // every lane checked the pivots of its own block
    R hv[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      rho[i] = group_sum<R, L>(rho[i]);
      hv[i] = group_sum<R, L>(ha[i]);
#pragma unroll
      for (int j = 0; j <= i; ++j) Sm[i][j] = group_sum<R, L>(Sm[i][j]);
    }
#pragma unroll
    for (int t = 0; t < NX; ++t) {
      cn += Math<R>::fabs(ci[t]);
      if (Dg[t] != R(0)) {
        const R r = Rw[t] * e_term[t];
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

    // ================= sweep 1b, local pass down my block: v_k = -(gw_k + w_k . q) / d_k ====================
    //   w_k . q = psi . wt_k + e_k (w_in . q),  psi = Psi_s^T q,  and  psi . wt_k  obeys wt's recurrence, so W is
    //   never stored.  v_k is parked in lds_du[k]; sweep 2 turns it into du_k in place.
    {
      R psi[NX];
      R om_in = gw_in;  // gw_in + w_in . q: what a unit of e_k carries into (gw_k + w_k . q)
#pragma unroll
      for (int c = 0; c < NX; ++c) {
        R acc = Psi[0][c] * q[0];
#pragma unroll
        for (int r = 1; r < NX; ++r) acc += Psi[r][c] * q[r];
        psi[c] = acc;
        om_in += w_in[c] * q[c];
      }
      R om = R(0), e = R(1);
      R idn = id_right;
#pragma unroll CPMPC_SWEEP_UNROLL
      for (int i = SP - 1; i >= 0; --i) {
        const int kk = s * SP + i;
        R Gi[NX];
        unpack<R, NX>(lds_G[i * 64 + lane], Gi);
        R pg = psi[0] * Gi[0];
#pragma unroll
        for (int m = 1; m < NX; ++m) pg += psi[m] * Gi[m];
        const R idk = lds_id[i * 64 + lane];
        const R ups = (kk < N - 1) ? (-wd2 * idn) : R(0);
        om = pg - ups * om;
        e = -ups * e;
        lds_du[i * 64 + lane] = -(lds_gw[i * 64 + lane] + om + e * om_in) * idk;
        idn = idk;
      }
    }

    CPMPC_TICK(4);
    // ================= sweep 2, up the blocks: du_k = v_k - ups_{k-1} du_{k-1},  dx_{s+1} = Phi dx_s + Gamma du + c ====
    //   local:  dut_k = v_k - ups_{k-1} dut_{k-1},  et_k = -ups_{k-1} et_{k-1}  (dut = 0, et = 1 before the block),
    //           du_k = dut_k + et_k du_in;  ups_{k-1} = -wd2 / d_k uses my own pivots only
    R dxs[NX], dxe[NX];
    R gd_part = R(0), curv_part = R(0);
    R du_in = R(0);  // du of the control before my block (0 for interval 0: u_prev is fixed)
    {
      R Gt[NX], Ht[NX], dut = R(0), et = R(1);
#pragma unroll
      for (int t = 0; t < NX; ++t) {
        Gt[t] = cdef[t];
        Ht[t] = R(0);
      }
#pragma unroll CPMPC_SWEEP_UNROLL
      for (int i = 0; i < SP; ++i) {
        const int kk = s * SP + i;
        const R upm = (kk > 0) ? (-wd2 * lds_id[i * 64 + lane]) : R(0);
        dut = lds_du[i * 64 + lane] - upm * dut;
        et = -upm * et;
        lds_du[i * 64 + lane] = dut;
        R Gi[NX];
        unpack<R, NX>(lds_G[i * 64 + lane], Gi);
#pragma unroll
        for (int r = 0; r < NX; ++r) {
          Gt[r] += Gi[r] * dut;
          Ht[r] += Gi[r] * et;
        }
      }
      // boundary chain, up the lanes: dx_s and du_in
#pragma unroll
      for (int t = 0; t < NX; ++t) dxs[t] = -ci[t];  // interval 0's start; the others receive theirs below
#pragma unroll 1
      for (int round = 0; round < L; ++round) {
#pragma unroll
        for (int r = 0; r < NX; ++r) {
          R v = Gt[r] + Ht[r] * du_in;
#pragma unroll
          for (int m = 0; m < NX; ++m) v += Phi[r][m] * dxs[m];
          dxe[r] = v;
        }
        if (round + 1 < L) {
          const bool edge = (s == 0);
#pragma unroll
          for (int t = 0; t < NX; ++t) {
            const R v = lane_left<R, L>(dxe[t]);
            dxs[t] = edge ? dxs[t] : v;
          }
          const R v1 = lane_left<R, L>(dut + et * du_in);
          du_in = edge ? du_in : v1;
        }
      }
      // local correction du_k = dut_k + et_k du_in, slope and curvature of the control rows
      R du_prev = du_in;
      R u_lo = u_left;
      R u_cur = lds_u[0 * 64 + lane];
      et = R(1);
#pragma unroll CPMPC_SWEEP_UNROLL
      for (int i = 0; i < SP; ++i) {
        const int kk = s * SP + i;
        const R upm = (kk > 0) ? (-wd2 * lds_id[i * 64 + lane]) : R(0);
        et = -upm * et;
        const R du = lds_du[i * 64 + lane] + et * du_in;
        lds_du[i * 64 + lane] = du;
        // control-cost gradient g_k, recomputed from u exactly as in sweep 1
        const R u_hi = (i + 1 < SP) ? lds_u[(i + 1) * 64 + lane] : u_right;
        R g = wu2 * u_cur + wd2 * (u_cur - u_lo);
        if (kk < N - 1) g += wd2 * (u_cur - u_hi);
        gd_part += g * du;
        const R jd = a.wd * (du_prev - du);
        curv_part += wu2 * du * du + jd * jd + lam * du * du;
        du_prev = du;
        u_lo = u_cur;
        u_cur = u_hi;
      }
    }
    R gd = group_sum<R, L>(gd_part), curv = group_sum<R, L>(curv_part);
    {
      R dxT[NX];
#pragma unroll
      for (int t = 0; t < NX; ++t) dxT[t] = group_last<R, L>(dxe[t], gbase);
#pragma unroll
      for (int t = 0; t < NX; ++t) {
        if (Dg[t] != R(0)) {
          const R jd = Rw[t] * dxT[t];
          gd += (Rw[t] * e_term[t]) * jd;
          curv += jd * jd;
        }
      }
    }
    if (live && status == kTermNone && (!Math<R>::finite(gd) || !Math<R>::finite(curv))) status = kTermQpIndefinite;
    // du of my left neighbour's last control, for the (u_{k-1} - u_k) row of my first control
    const R du_left = du_in;

    CPMPC_TICK(5);
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
          if (Dg[r] != R(0)) {
            const R rr = Rw[r] * d[r];
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

    CPMPC_TICK(6);
    // ================= accept / reject, step-length memory, damping, termination ===========================
    if (live && status == kTermNone) {
      a_start = R(1);
      if (accepted && a.alpha_growth > R(0)) {
        a_start = (evals > 1 ? a.alpha_growth_bt : a.alpha_growth) * alpha;
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

  CPMPC_TICK(7);
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
  CPMPC_TICK(7);
  CPMPC_TICK_FLUSH();
}

}  // namespace cpmpc
