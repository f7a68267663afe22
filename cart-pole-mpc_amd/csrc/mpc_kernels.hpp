// mpc_kernels.hpp -- the batched MPC inner loop as gfx950 kernels, one problem per lane, generic over
// the model policy M (models.hpp: state dimension NX = 4 cart + single pole, 6 cart + double pole).
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
// Workspace layout in HBM: field-major with the problem index fastest.  Every NX-vector of the problem
// (a shooting node, a column of Gamma, a row of Phi, a defect, a row of W) is ONE element of 1 (NX = 4)
// or 2 (NX = 6, padded to 8) 16-byte (fp32) / 32-byte (fp64) vectors, so lane i of a wave reads
// `field_base + i` with global_load_dwordx4 and the wave moves whole KiB fully coalesced; scalars
// (controls, per-problem solver state) are 4/8-byte elements, 256/512 B per wave.  The field base is
// wave-uniform (scalar registers), the lane offset is the 32-bit problem index.  No LDS: per-lane state
// lives in VGPRs, the per-interval sensitivities stream through the workspace once per SQP iteration.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "models.hpp"
#include "wide.hpp"
#include "workspace_layout.hpp"

// threads per workgroup of the one-thread-per-problem kernels before and after the SQP (prepare, finalize)
#ifndef CPMPC_PF_BLOCK
#define CPMPC_PF_BLOCK 256
#endif
#ifndef CPMPC_EXIT_FLOOR_F64
#define CPMPC_EXIT_FLOOR_F64 0  // see mpc_fused.hpp
#endif
#ifndef CPMPC_SKIP_MERIT
#define CPMPC_SKIP_MERIT 1      // 0: A/B build that evaluates the merit of a converged step as rounds 1-3 did (NOT the specification)
#endif

namespace cpmpc {

constexpr int kTermNone = 0;
constexpr int kTermMaxIterations = 1;
constexpr int kTermRelTol = 3;
constexpr int kTermFirstOrder = 4;
constexpr int kTermQpIndefinite = 5;
constexpr int kTermMaxLambda = 7;
constexpr int kTermNonFinite = 8;

constexpr int kMaxNX = 6;

// occupancy hints (second argument of __launch_bounds__ = minimum waves per SIMD); 0 = compiler's choice
#ifndef CPMPC_QPLS_WAVES
#define CPMPC_QPLS_WAVES 0
#endif
#ifndef CPMPC_LIN_WAVES
#define CPMPC_LIN_WAVES 0
#endif
#if CPMPC_QPLS_WAVES > 0
#define CPMPC_QPLS_BOUNDS __launch_bounds__(64, CPMPC_QPLS_WAVES)
#else
#define CPMPC_QPLS_BOUNDS __launch_bounds__(64)
#endif
#if CPMPC_LIN_WAVES > 0
#define CPMPC_LIN_BOUNDS __launch_bounds__(64, CPMPC_LIN_WAVES)
#else
#define CPMPC_LIN_BOUNDS __launch_bounds__(64)
#endif

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

// storage element of an NX-vector: ceil(NX/4) 4-vectors
template <typename R, int NX>
struct XV {
  typename VecT<R>::V4 v[(NX + 3) / 4];
};

template <typename R, int NX>
__device__ __forceinline__ void unpack(const XV<R, NX>& s, R (&x)[NX]) {
  x[0] = s.v[0].x;
  x[1] = s.v[0].y;
  x[2] = s.v[0].z;
  x[3] = s.v[0].w;
  if constexpr (NX > 4) {
    x[4] = s.v[1].x;
    x[5] = s.v[1].y;
  }
}
template <typename R, int NX>
__device__ __forceinline__ XV<R, NX> pack(const R (&x)[NX]) {
  XV<R, NX> s;
  s.v[0] = mk4<R>(x[0], x[1], x[2], x[3]);
  if constexpr (NX > 4) s.v[1] = mk4<R>(x[4], x[5], R(0), R(0));
  return s;
}

template <typename R, typename M>
struct SolverArgs {
  using V4 = typename VecT<R>::V4;
  using XVn = XV<R, M::NX>;
  // sizes
  int64_t B;       // problems in this call
  int64_t stride;  // elements between consecutive fields (capacity of the workspace)
  int N, S, SP;
  R dt;
  // cost weights (optimization.hpp:40-48)
  R wu, wd;            // u_cost_weight, u_derivative_cost_weight (0 disables the rows)
  R term_w[kMaxNX];    // residual weight of terminal row t (1 for an equality row)
  R term_tgt[kMaxNX];  // targets; [0] is the shared set-point unless set_point != nullptr
  int term_is_cost;    // bit t set: terminal row t is a cost (weight >= 0), else an equality
  // solver options
  int max_ls;
  R c1, shrink_max, shrink_min, alpha_growth, alpha_growth_bt, rho, full_step_below;
  R cn_floor_scale;  // exit_defect_floor * state_spacing * eps(R): |c|_1 up to this times sum |x_s| is rounding (first-order exit test)
  R lam_init, lam_fail_init, lam_up, lam_down, lam_min, lam_max;
  R bx_lim, u_lim;
  R rel_tol, fo_tol, mu_init;
  int64_t prev_B;  // problems [0, prev_B) hold a previous solution (warm start); the others start cold
  int refine_qp;   // split pipeline, double: passes of refinement of the whole QP solution (the fused kernels: one, by template)
  // workspace (device)
  XVn* zx;   // [S]        shooting nodes of the iterate          } persist between calls:
  R* zu;     // [N]        controls of the iterate                } the warm start
  XVn* dzx;  // [S]        QP step, nodes
  R* dzu;    // [N]        QP step, controls
  XVn* Phi;  // [NX(S-1)]  row r of Phi_s at field NX*s+r
  XVn* Gam;  // [N]        column k of Gamma = d x_end / d u_k
  XVn* cs;   // [S-1]      shooting defects
  XVn* Wk;   // [N]        row k of U^-1 R^T
  V4* Tk;    // [N]        {(U^-1 g)_k, upsilon_k, 1/d_k, g_k}
  R* sc;     // [SC_COUNT]
  int32_t* ist;        // [IS_COUNT]
  const R* sin_table;  // [N] device: u_guess_sinusoid_amplitude * sin(2 pi k / N), from the host
  // inputs, packed [field][B]
  const R* x0;         // [NX]
  const R* dyn;        // [NP] per-problem, or nullptr
  const R* set_point;  // [1] per-problem, or nullptr
  const R* term_w_pp;  // [NX] per-problem terminal weights in state order (negative = equality row), or nullptr
  // fused pipeline, later stages: the problems still iterating, compacted (nullptr: all problems, identity map)
  const int32_t* active_list;
  const int32_t* active_count;
  int iter_cap;            // max_iterations: a problem never iterates past it, however the launches are staged
  int64_t run_out_below;   // a later stage with at most this many active problems runs them to the end
  const int32_t* prev_count;  // active problems at the previous compaction (nullptr: prev_total)
  int64_t prev_total;
  int remaining;           // iterations still to run, this stage included
  int32_t* stage_count0;   // prepare_kernel clears it: the first compaction's counter (nullptr: one launch, no compaction)
  // feedback for the host's staging plan (finalize_kernel): histogram of this step's iterations per problem
  int32_t* fb_host;        // host-mapped [reporter][kFbBins + 1]: a workgroup's histogram, then fb_seq (written last); nullptr: off
  int fb_seq;
  int fb_stride;           // workgroups blockIdx.x % fb_stride == 0 report
  typename M::Consts consts;  // shared model constants (used when dyn == nullptr)
  // outputs, packed [field][B] (nullable)
  R* u_out;
  R* pred_out;
  int32_t* status_out;
  int32_t* iters_out;
  int32_t* ls_out;
  R* cost_out;
  R* eq_out;
  R* guess_out;
  R* sol_out;  // [dim][B] packed solution (MapKey order)
};

// Terminal rows of problem p (optimization.cc:236-267): residual weight Rw[t] (1 for an equality row) and
// Dg[t] = 1 for a cost row (weight >= 0), 0 for an equality row.  Shared by the batch unless per-problem
// weights were given (the UI's per-controller cost/constraint toggles, viz/src/application.ts:279-342).
template <typename R, typename M>
__device__ __forceinline__ void load_terminal(const SolverArgs<R, M>& a, unsigned p, R (&Rw)[M::NX], R (&Dg)[M::NX]) {
#pragma unroll
  for (int t = 0; t < M::NX; ++t) {
    if (a.term_w_pp != nullptr) {
      const R w = a.term_w_pp[(int64_t)t * a.B + p];
      const bool is_cost = w >= R(0);
      Rw[t] = is_cost ? w : R(1);
      Dg[t] = is_cost ? R(1) : R(0);
    } else {
      Rw[t] = a.term_w[t];
      Dg[t] = ((a.term_is_cost >> t) & 1) ? R(1) : R(0);
    }
  }
}

template <typename R, typename M>
__device__ __forceinline__ typename M::Consts load_consts(const SolverArgs<R, M>& a, unsigned p) {
  if (a.dyn == nullptr) return a.consts;
  R prm[M::NP];
#pragma unroll
  for (int i = 0; i < M::NP; ++i) prm[i] = a.dyn[i * a.B + p];
  return M::template make<R>(prm);
}

template <typename R>
__device__ __forceinline__ R clampr(R v, R lo, R hi) {
  return v < lo ? lo : (v > hi ? hi : v);
}

// running maximum for |dz|_inf of the full-step rule: one v_max.  The hardware maximum DROPS a NaN operand, so a step with
// a NaN component may still be classed "tiny" here where the CPU restatement (which propagates the NaN) says "not tiny".
// The two then differ only in the first trial step length (1 instead of alpha_start) of a line search that cannot
// succeed either way: every trial point has a NaN component, its merit is non-finite, and both the Armijo test and the
// full-step rule (which demands a finite merit) reject it -- same trial count, same rejection, same damping update.  In
// the usual case the NaN has already made g.du or |J dz|^2 non-finite and the problem is QP_INDEFINITE before this.
template <typename R>
__device__ __forceinline__ R nan_max(R a, R b) {
  return (sizeof(R) == 8) ? (R)__builtin_fmax((double)a, (double)b) : (R)__builtin_fmaxf((float)a, (float)b);
}

// reciprocal of a pivot of the terminal system in its wide type W; W == R keeps the kernel's own division
template <typename R, typename W>
__device__ __forceinline__ W wide_inv(const W d) {
  if constexpr (std::is_same<W, R>::value) return Math<R>::div(R(1), d);
  else return Math<double>::div(1.0, d);
}

// wrap the pole angles of a state / state difference
template <typename R, typename M>
__device__ __forceinline__ void wrap_angles(R (&x)[M::NX]) {
#pragma unroll
  for (int t = 1; t < M::NQ; ++t) x[t] = mod_pi(x[t]);
}

// ------------------------------------------------------------------------------------------------
// prepare: initial guess.  One thread per problem.
// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// Stream compaction between the stages of the fused pipeline: the indices of the problems that are
// still iterating (status NONE), densely packed, so that the next stage runs full waves instead of
// waves kept alive by one straggler.  One atomic per wave (ballot + prefix popcount); the order of the
// list is arbitrary, which is harmless: a problem's arithmetic does not depend on where it sits.
// ------------------------------------------------------------------------------------------------
template <typename Tag>  // Tag = the model policy: one instance per translation unit of the library (engine_*.hip)
__global__ __launch_bounds__(1024) void compact_active_kernel(const int32_t* status, const int32_t* iters,
                                                              int iter_cap, int64_t B, int32_t* list, int32_t* count,
                                                              int32_t* clear_for_next) {
  // three counters rotate through the stages: this compaction counts into `count`, the stage it feeds also reads the one
  // before, and the third -- nobody's until the next compaction, whose `count` it will be -- is cleared here (no memset
  // between the launches; prepare_kernel cleared the first one)
  if (blockIdx.x == 0 && threadIdx.x == 0) *clear_for_next = 0;
  // one atomic per 1024-thread workgroup (16 waves): the wave totals go through LDS, wave 0 reserves the block's range.
  // (One atomic per WAVE on a single counter serialised: 4096 of them took 42 us at B = 262144, profiles/r02b_closed_loop.)
  __shared__ int wave_total[16];
  __shared__ int block_base;
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = p < B && status[p] == kTermNone && iters[p] < iter_cap;
  const unsigned long long mask = __ballot(active);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rank = __popcll(mask & ((1ull << lane) - 1ull));
  if (lane == 0) wave_total[wave] = __popcll(mask);
  __syncthreads();
  if (threadIdx.x == 0) {
    int tot = 0;
    for (int w = 0; w < 16; ++w) {
      const int t = wave_total[w];
      wave_total[w] = tot;   // exclusive prefix
      tot += t;
    }
    block_base = tot ? atomicAdd(count, tot) : 0;
  }
  __syncthreads();
  if (active) list[block_base + wave_total[wave] + rank] = (int32_t)p;
}

template <typename R, typename M>
__global__ __launch_bounds__(CPMPC_PF_BLOCK) void prepare_kernel(const SolverArgs<R, M> a) {  // workgroups as finalize
  constexpr int NX = M::NX;
  const unsigned p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= a.B) return;
  if (p == 0 && a.stage_count0 != nullptr) *a.stage_count0 = 0;  // the counter of the first compaction of this step
  const int64_t st = a.stride;
  const typename M::Consts k = load_consts(a, p);
  const ExtForce<R> fe{R(0), R(0), R(0)};

  // BuildProblem reads u_prev before the previous solution is overwritten (optimization.cc:288-291)
  const bool warm = (int64_t)p < a.prev_B;  // per problem: a controller with no previous solution starts cold (optimization.cc:46-68)
  const R u_prev = warm ? a.zu[p] : R(0);
  a.sc[SC_UPREV * st + p] = u_prev;
  a.sc[SC_LAMBDA * st + p] = a.lam_init;
  a.sc[SC_MU * st + p] = a.mu_init;
  a.sc[SC_F_LAST * st + p] = R(0);
  a.sc[SC_CN_LAST * st + p] = R(0);
  a.sc[SC_ALPHA * st + p] = R(1);
  a.sc[SC_TRIAL * st + p] = R(0);
  a.ist[IS_STATUS * st + p] = kTermNone;
  a.ist[IS_ITERS * st + p] = 0;
  a.ist[IS_LS_EVALS * st + p] = 0;
  a.ist[IS_FAILED * st + p] = 0;

  // The guess of the controls -- warm: the previous ones shifted left by one, the last duplicated (optimization.cc:54-57),
  // cold: the sinusoid (optimization.cc:58-68) -- is written WHILE the states are rolled (FillInitialGuess,
  // optimization.cc:333-351, wrapping after every step), each control fetched one step before it is integrated: the shift
  // in place reads index k+2 while it writes index k, and no step waits for the load of a value stored just before it
  // (that round trip per step was 45 % of this kernel's wave-cycles).
  auto guess_u = [&](int i) -> R {  // control i of the new guess, i < N
    if (warm) return a.zu[(int64_t)((i + 1 < a.N) ? i + 1 : a.N - 1) * st + p];
    return a.sin_table[i];
  };
  R x[NX];
#pragma unroll
  for (int t = 0; t < NX; ++t) x[t] = a.x0[t * a.B + p];
  a.zx[p] = pack<R, NX>(x);
  if (a.guess_out) {
#pragma unroll
    for (int t = 0; t < NX; ++t) a.guess_out[(int64_t)t * a.B + p] = x[t];
  }
  int kk = 0;
  R u_next = guess_u(0);
  for (int s = 1; s < a.S; ++s) {
    typename M::StepCache chain;  // consecutive steps share the pole angle's sine / cosine base; re-anchored per interval
    for (int i = 0; i < a.SP; ++i, ++kk) {
      const R u = u_next;
      if (kk + 1 < a.N) u_next = guess_u(kk + 1);
      a.zu[(int64_t)kk * st + p] = u;
      if (a.guess_out) a.guess_out[(int64_t)(NX * a.S + kk) * a.B + p] = u;
      rk4_step_m<R, M, false>(k, a.dt, x, u, fe, chain);
      wrap_angles<R, M>(x);
    }
    a.zx[(int64_t)s * st + p] = pack<R, NX>(x);
    if (a.guess_out) {
#pragma unroll
      for (int t = 0; t < NX; ++t) a.guess_out[(int64_t)(NX * s + t) * a.B + p] = x[t];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// linearize: one thread per (problem, shooting interval).
// Integrates x_s through SP controls with RK4, accumulating Phi = dx_end/dx_s and
// Gamma = dx_end/du FORWARD (Phi <- A Phi, Gamma_j <- A Gamma_j, Gamma_i = B) so that nothing per
// step has to be stored: the NX x SP block lives in registers with static indices.  Algebraically
// equal to the reference's backward accumulation (optimization.cc:145-154).
// ------------------------------------------------------------------------------------------------
template <typename R, typename M, int SP>
__global__ CPMPC_LIN_BOUNDS void linearize_kernel(const SolverArgs<R, M> a, const XV<R, M::NX>* zx_in,
                                                        const R* zu_in, const int32_t* status) {
  constexpr int NX = M::NX;
  const int64_t gid = (int64_t)blockIdx.x * 64 + threadIdx.x;
  const int s = (int)(gid / a.B);
  const unsigned p = (unsigned)(gid - (int64_t)s * a.B);
  if (s >= a.S - 1) return;
  const int64_t st = a.stride;
  if (status != nullptr && status[IS_STATUS * st + p] != kTermNone) return;
  const typename M::Consts k = load_consts(a, p);
  const ExtForce<R> fe{R(0), R(0), R(0)};

  R x[NX], xe[NX];
  unpack<R, NX>(zx_in[(int64_t)s * st + p], x);
  unpack<R, NX>(zx_in[(int64_t)(s + 1) * st + p], xe);
  R Phi[NX][NX];
  R Gam[SP][NX];
#pragma unroll
  for (int r = 0; r < NX; ++r)
#pragma unroll
    for (int c = 0; c < NX; ++c) Phi[r][c] = (r == c) ? R(1) : R(0);
#pragma unroll
  for (int j = 0; j < SP; ++j)
#pragma unroll
    for (int r = 0; r < NX; ++r) Gam[j][r] = R(0);

  const R* zu = zu_in + (int64_t)(s * SP) * st;
  R u_next = zu[p];
  typename M::StepCache chain;  // consecutive steps of the interval share the sine / cosine base (models.hpp)
#pragma unroll 1
  for (int i = 0; i < SP; ++i) {
    const R u = u_next;
    if (i + 1 < SP) u_next = zu[(int64_t)(i + 1) * st + p];  // prefetch the next control
    R A[NX][NX], Bv[NX];
    rk4_step_jac_m<R, M, false>(k, a.dt, x, u, fe, A, Bv, chain);
    // Phi <- A Phi
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
      // Gamma columns: j < i propagate, j == i is the new control's column.  `i` is uniform over
      // the wave, so these are scalar branches and the register indices stay static.
#pragma unroll
    for (int j = 0; j < SP; ++j) {
      if (j < i) {
        R g[NX];
#pragma unroll
        for (int m = 0; m < NX; ++m) g[m] = Gam[j][m];
#pragma unroll
        for (int r = 0; r < NX; ++r) {
          R acc = A[r][0] * g[0];
#pragma unroll
          for (int m = 1; m < NX; ++m) acc += A[r][m] * g[m];
          Gam[j][r] = acc;
        }
      } else if (j == i) {
#pragma unroll
        for (int r = 0; r < NX; ++r) Gam[j][r] = Bv[r];
      }
    }
  }
  // wrap the angles once at the end of the interval, then the defect (optimization.cc:139,156-157)
  wrap_angles<R, M>(x);
  R c[NX];
#pragma unroll
  for (int t = 0; t < NX; ++t) c[t] = x[t] - xe[t];
  wrap_angles<R, M>(c);
  a.cs[(int64_t)s * st + p] = pack<R, NX>(c);
#pragma unroll
  for (int r = 0; r < NX; ++r) a.Phi[(int64_t)(NX * s + r) * st + p] = pack<R, NX>(Phi[r]);
#pragma unroll
  for (int j = 0; j < SP; ++j) a.Gam[(int64_t)(s * SP + j) * st + p] = pack<R, NX>(Gam[j]);
}

// ------------------------------------------------------------------------------------------------
// linearize for a state spacing without a register-resident specialisation: the same forward
// accumulation with the spacing a run-time value and the Gamma columns of the interval kept in the
// workspace (each thread re-reads and re-writes only its own elements, coalesced over the wave).
// O(SP^2) vector read-modify-writes per interval: correct for any spacing, not fast for long ones.
// ------------------------------------------------------------------------------------------------
template <typename R, typename M>
__global__ __launch_bounds__(64) void linearize_dyn_kernel(const SolverArgs<R, M> a, const XV<R, M::NX>* zx_in,
                                                           const R* zu_in, const int32_t* status) {
  constexpr int NX = M::NX;
  const int SP = a.SP;
  const int64_t gid = (int64_t)blockIdx.x * 64 + threadIdx.x;
  const int s = (int)(gid / a.B);
  const unsigned p = (unsigned)(gid - (int64_t)s * a.B);
  if (s >= a.S - 1) return;
  const int64_t st = a.stride;
  if (status != nullptr && status[IS_STATUS * st + p] != kTermNone) return;
  const typename M::Consts k = load_consts(a, p);
  const ExtForce<R> fe{R(0), R(0), R(0)};

  R x[NX], xe[NX];
  unpack<R, NX>(zx_in[(int64_t)s * st + p], x);
  unpack<R, NX>(zx_in[(int64_t)(s + 1) * st + p], xe);
  R Phi[NX][NX];
#pragma unroll
  for (int r = 0; r < NX; ++r)
#pragma unroll
    for (int c = 0; c < NX; ++c) Phi[r][c] = (r == c) ? R(1) : R(0);

  const R* zu = zu_in + (int64_t)s * SP * st;
  XV<R, NX>* gam = a.Gam + (int64_t)s * SP * st;
  typename M::StepCache chain;
#pragma unroll 1
  for (int i = 0; i < SP; ++i) {
    const R u = zu[(int64_t)i * st + p];
    R A[NX][NX], Bv[NX];
    rk4_step_jac_m<R, M, false>(k, a.dt, x, u, fe, A, Bv, chain);
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
#pragma unroll 1
    for (int j = 0; j < i; ++j) {
      R g[NX], gn[NX];
      unpack<R, NX>(gam[(int64_t)j * st + p], g);
#pragma unroll
      for (int r = 0; r < NX; ++r) {
        R acc = A[r][0] * g[0];
#pragma unroll
        for (int m = 1; m < NX; ++m) acc += A[r][m] * g[m];
        gn[r] = acc;
      }
      gam[(int64_t)j * st + p] = pack<R, NX>(gn);
    }
    gam[(int64_t)i * st + p] = pack<R, NX>(Bv);
  }
  wrap_angles<R, M>(x);
  R c[NX];
#pragma unroll
  for (int t = 0; t < NX; ++t) c[t] = x[t] - xe[t];
  wrap_angles<R, M>(c);
  a.cs[(int64_t)s * st + p] = pack<R, NX>(c);
#pragma unroll
  for (int r = 0; r < NX; ++r) a.Phi[(int64_t)(NX * s + r) * st + p] = pack<R, NX>(Phi[r]);
}

// ------------------------------------------------------------------------------------------------
// merit evaluation at z (+) alpha dz: 1/2 |r|^2 and |c|_1 through the retraction
// (optimization.cc:309-329) and a no-Jacobian rollout of every interval (optimization.cc:130-139).
// ------------------------------------------------------------------------------------------------
template <typename R, typename M>
__device__ __forceinline__ void trial_node(const SolverArgs<R, M>& a, const int s, const unsigned p,
                                           const R alpha, R (&xs)[M::NX]) {
  constexpr int NX = M::NX;
  R zv[NX], dv[NX];
  unpack<R, NX>(a.zx[(int64_t)s * a.stride + p], zv);
  unpack<R, NX>(a.dzx[(int64_t)s * a.stride + p], dv);
#pragma unroll
  for (int t = 0; t < NX; ++t) xs[t] = zv[t] + alpha * dv[t];
  xs[0] = clampr(xs[0], -a.bx_lim, a.bx_lim);
  wrap_angles<R, M>(xs);
}

template <typename R, typename M>
__device__ __forceinline__ void merit_eval(const SolverArgs<R, M>& a, const typename M::Consts& k,
                                           const unsigned p, const R alpha, const R (&xm)[M::NX],
                                           const R (&tgt)[M::NX], const R u_prev, R& f_out, R& cn_out) {
  constexpr int NX = M::NX;
  const int64_t st = a.stride;
  const ExtForce<R> fe{R(0), R(0), R(0)};
  R f = R(0), cn = R(0);

  // node 0 and the initial-state equality rows (optimization.cc:228-232)
  R xs[NX];
  trial_node<R, M>(a, 0, p, alpha, xs);
  {
    R d[NX];
#pragma unroll
    for (int t = 0; t < NX; ++t) d[t] = xs[t] - xm[t];
    wrap_angles<R, M>(d);
#pragma unroll
    for (int t = 0; t < NX; ++t) cn += Math<R>::fabs(d[t]);
  }

  R u_before = u_prev;  // u_{k-1} of the trial point, for the du rows
  int kk = 0;
  R u_raw = a.zu[p] + alpha * a.dzu[p];
  for (int s = 0; s + 1 < a.S; ++s) {
    R x[NX];
#pragma unroll
    for (int t = 0; t < NX; ++t) x[t] = xs[t];
    typename M::StepCache chain;  // per interval, as in the linearisation
    for (int i = 0; i < a.SP; ++i, ++kk) {
      const R u = clampr(u_raw, -a.u_lim, a.u_lim);
      if (kk + 1 < a.N)  // prefetch the next control of the trial point
        u_raw = a.zu[(int64_t)(kk + 1) * st + p] + alpha * a.dzu[(int64_t)(kk + 1) * st + p];
      // control cost rows (optimization.cc:270-301)
      const R ru = a.wu * u;
      const R rd = a.wd * (u_before - u);  // (u_{k-1} - u_k) w; for k = 0 it is -(u_0 - u_prev) w
      f += ru * ru + rd * rd;
      u_before = u;
      rk4_step_m<R, M, false>(k, a.dt, x, u, fe, chain);
    }
    wrap_angles<R, M>(x);
    trial_node<R, M>(a, s + 1, p, alpha, xs);  // next node of the trial point
    R d[NX];
#pragma unroll
    for (int t = 0; t < NX; ++t) d[t] = x[t] - xs[t];
    wrap_angles<R, M>(d);
#pragma unroll
    for (int t = 0; t < NX; ++t) cn += Math<R>::fabs(d[t]);
  }
  // terminal rows on the last node (optimization.cc:236-267)
  {
    R d[NX];
#pragma unroll
    for (int t = 0; t < NX; ++t) d[t] = xs[t] - tgt[t];
    wrap_angles<R, M>(d);
    R Rw[NX], Dg[NX];
    load_terminal<R, M>(a, p, Rw, Dg);
#pragma unroll
    for (int t = 0; t < NX; ++t) {
      if (Dg[t] != R(0)) {
        const R r = Rw[t] * d[t];
        f += r * r;
      } else {
        cn += Math<R>::fabs(d[t]);
      }
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
// <= NX terminal rows (cost or equality).
//   sweep 1 (k descending): T = U D U^T by a scalar recurrence; W = U^-1 R^T row by row from
//           m_k = Psi Gamma_k, Psi = diag(w) Phi_{S-2} ... Phi_{s+1}; gw = U^-1 g; accumulate
//           S = W^T D^-1 W (NX x NX), rho = W^T D^-1 gw and the weighted free response sum_s Psi_s c_s;
//           rows of W and {gw, upsilon, 1/d, g} are stored.
//   NX x NX LDL^T of S + diag(1 for cost rows, 0 for equality rows) in registers -> multipliers q.
//   sweep 2 (k ascending): y = -(gw + W q), U^T du = D^-1 y, state recovery through Phi/Gamma, and the
//           directional quantities g.du and |J dz|^2.
// Then the l1-merit penalty update and the Armijo line search with quadratic-interpolation
// backtracking, started from the remembered step length.
// ------------------------------------------------------------------------------------------------
// The terminal system (S, rho, the weighted free response, the LDL^T and its solve) is carried in the wide type W of
// wide.hpp: double in a float kernel, R itself in a double kernel.
// WIDEQ (float handles with CPMPC_CREATE_WIDE_QP, the default for the 6-state model; round 6): as in the fused kernel (type Q
// of mpc_fused_body.inc), everything between the linearisation and the multipliers' effect on the step is carried in double
// too -- the Psi products across the intervals, the columns w_k of U^-1 R^T, psi = Psi^T q and y = -(gw + W q).  W is then
// never read back in float: after the multipliers are known a pass of its own ("sweep 1b", k descending) forms
//     w_k . q = psi_s . Gamma_k - ups_k (w_{k+1} . q),   psi_s = Phi_{s+1}^T psi_{s+1},  psi_{S-2} = diag(w) q
// in double and leaves y_k in the slot of (U^-1 g)_k.  One more pass over Gamma, Phi and T than the plain kernel: this
// pipeline is the fall-back for shapes the fused kernel is not built for, correctness is its bar, not throughput.
template <typename R, typename M, bool WIDEQ = false>
__global__ CPMPC_QPLS_BOUNDS void qp_ls_kernel(const SolverArgs<R, M> a) {
  using V4 = typename VecT<R>::V4;
  using XVn = XV<R, M::NX>;
  using W = typename WideOf<R>::type;
  using WO = Wide<W>;
  constexpr bool kWidened = !std::is_same<W, R>::value;
  constexpr bool kWideQP = WIDEQ && kWidened;
  using Q = std::conditional_t<kWideQP, W, R>;
  constexpr int NX = M::NX;
  const unsigned p = blockIdx.x * 64u + threadIdx.x;
  if (p >= a.B) return;
  const int64_t st = a.stride;
  if (a.ist[IS_STATUS * st + p] != kTermNone) return;
  const typename M::Consts k = load_consts(a, p);
  const int N = a.N, S = a.S, SP = a.SP;

  R lam = a.sc[SC_LAMBDA * st + p];
  R mu = a.sc[SC_MU * st + p];
  const R u_prev = a.sc[SC_UPREV * st + p];
  R a_start = a.sc[SC_ALPHA * st + p];
  R tgt[NX], xm[NX];
#pragma unroll
  for (int t = 0; t < NX; ++t) {
    tgt[t] = a.term_tgt[t];
    xm[t] = a.x0[t * a.B + p];
  }
  if (a.set_point) tgt[0] = a.set_point[p];

  const R wu2 = a.wu * a.wu, wd2 = a.wd * a.wd;

  // ---- residuals at z: initial-state rows, terminal rows -----------------------------------------
  R f = R(0), cn = R(0);
  R ci[NX];
  {
    R z0[NX];
    unpack<R, NX>(a.zx[p], z0);
#pragma unroll
    for (int t = 0; t < NX; ++t) ci[t] = z0[t] - xm[t];
    wrap_angles<R, M>(ci);
#pragma unroll
    for (int t = 0; t < NX; ++t) cn += Math<R>::fabs(ci[t]);
  }
  // The weighted free response  ha = diag(w) dx_{S-1}|_{du=0} = sum_s Psi_s c_s - Psi_{-1} c_init  is
  // accumulated inside sweep 1, where Psi_s = diag(w) Phi_{S-2}...Phi_{s+1} is available anyway.
  W hv[NX];
  R Rw[NX], Dg[NX], e_term[NX];
  {
    R zt[NX];
    unpack<R, NX>(a.zx[(int64_t)(S - 1) * st + p], zt);
#pragma unroll
    for (int t = 0; t < NX; ++t) e_term[t] = zt[t] - tgt[t];
    wrap_angles<R, M>(e_term);
    load_terminal<R, M>(a, p, Rw, Dg);
#pragma unroll
    for (int t = 0; t < NX; ++t) {
      const bool is_cost = Dg[t] != R(0);
      if (is_cost) {
        const R r = Rw[t] * e_term[t];
        f += r * r;
      } else {
        cn += Math<R>::fabs(e_term[t]);
      }
      hv[t] = WO::prod(Rw[t], e_term[t]);  // + ha[t], added after sweep 1
    }
  }

  // ---- sweep 1 (k descending) -------------------------------------------------------------------
  W Sm[NX][NX], rho[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    rho[i] = WO::of(R(0));
#pragma unroll
    for (int j = 0; j < NX; ++j) Sm[i][j] = WO::of(R(0));
  }
  bool pd_ok = true;
  {
    Q Psi[NX][NX];
#pragma unroll
    for (int r = 0; r < NX; ++r)
#pragma unroll
      for (int c = 0; c < NX; ++c) Psi[r][c] = (r == c) ? Q(Rw[r]) : Q(0);
    Q wprev[NX];
    W ha[NX];
#pragma unroll
    for (int r = 0; r < NX; ++r) {
      wprev[r] = Q(0);
      ha[r] = WO::of(R(0));
    }
    R gwprev = R(0);
    R d_next = R(1);
    const XVn* __restrict__ gam_p = a.Gam + p;
    const R* __restrict__ zu_p = a.zu + p;
    R u_hi = R(0);                              // u_{k+1}
    R u_cur = zu_p[(int64_t)(N - 1) * st];      // u_k
    // software pipeline: the loads of column k-1 are issued before column k is consumed
    XVn G_nx = gam_p[(int64_t)(N - 1) * st];
    R u_nx = (N > 1) ? zu_p[(int64_t)(N - 2) * st] : u_prev;
    int kk = N - 1;
    for (int s = S - 2; s >= 0; --s) {
      for (int i = SP - 1; i >= 0; --i, --kk) {
        R gk[NX];
        unpack<R, NX>(G_nx, gk);
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
        Q wk[NX];
#pragma unroll
        for (int r = 0; r < NX; ++r) {
          Q m = Psi[r][0] * Q(gk[0]);
#pragma unroll
          for (int c = 1; c < NX; ++c) m += Psi[r][c] * Q(gk[c]);
          wk[r] = m - Q(ups) * wprev[r];
        }
        const R gw = g - ups * gwprev;
        if constexpr (!kWideQP) {  // (the wide kernel never reads W back: sweep 1b below)
          R wk_r[NX];
#pragma unroll
          for (int r = 0; r < NX; ++r) wk_r[r] = (R)wk[r];
          a.Wk[(int64_t)kk * st + p] = pack<R, NX>(wk_r);
        }
        a.Tk[(int64_t)kk * st + p] = mk4<R>(gw, ups, inv_d, g);
#pragma unroll
        for (int i2 = 0; i2 < NX; ++i2) {
          const W wi = (W)wk[i2] * (W)inv_d;  // exact in W for float columns: S is the Gram matrix of the rounded rows (wide.hpp)
          rho[i2] += wi * gw;
#pragma unroll
          for (int j2 = 0; j2 <= i2; ++j2) Sm[i2][j2] += wi * (W)wk[j2];
        }
#pragma unroll
        for (int r = 0; r < NX; ++r) wprev[r] = wk[r];
        gwprev = gw;
        u_hi = u_cur;
        u_cur = u_lo;
      }
      // defect of this interval: |c|_1 and its weighted propagation to the last node, Psi_s c_s
      {
        R c[NX];
        unpack<R, NX>(a.cs[(int64_t)s * st + p], c);
#pragma unroll
        for (int t = 0; t < NX; ++t) cn += Math<R>::fabs(c[t]);
#pragma unroll
        for (int r = 0; r < NX; ++r) {
          W acc = (W)Psi[r][0] * (W)c[0];
#pragma unroll
          for (int m = 1; m < NX; ++m) acc += (W)Psi[r][m] * (W)c[m];
          ha[r] += acc;
        }
      }
      // Psi <- Psi Phi_s
      R Ph[NX][NX];
      Q T[NX][NX];
#pragma unroll
      for (int r = 0; r < NX; ++r) unpack<R, NX>(a.Phi[(int64_t)(NX * s + r) * st + p], Ph[r]);
#pragma unroll
      for (int r = 0; r < NX; ++r)
#pragma unroll
        for (int c = 0; c < NX; ++c) {
          Q acc = Psi[r][0] * Q(Ph[0][c]);
#pragma unroll
          for (int m = 1; m < NX; ++m) acc += Psi[r][m] * Q(Ph[m][c]);
          T[r][c] = acc;
        }
#pragma unroll
      for (int r = 0; r < NX; ++r)
#pragma unroll
        for (int c = 0; c < NX; ++c) Psi[r][c] = T[r][c];
    }
    // Psi is now diag(w) Phi_{S-2}...Phi_0: contribution of dx_0 = -c_init
#pragma unroll
    for (int r = 0; r < NX; ++r) {
      W acc = (W)Psi[r][0] * (W)ci[0];
#pragma unroll
      for (int m = 1; m < NX; ++m) acc += (W)Psi[r][m] * (W)ci[m];
      hv[r] += ha[r] - acc;
    }
  }
  f *= R(0.5);

  int status = kTermNone;
  if (!Math<R>::finite(f) || !Math<R>::finite(cn)) status = kTermNonFinite;

  // ---- (S + Dg) q = h - rho by LDL^T on the lower triangle, in registers ------------------------
  Q q[NX];
  W Lm[NX][NX], dv[NX], idv[NX];  // (function scope: the refinement after sweep 2 solves with them again)
  {
#pragma unroll
    for (int i = 0; i < NX; ++i) Sm[i][i] += WO::of(Dg[i]);
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      W dj = Sm[j][j];
#pragma unroll
      for (int m = 0; m < j; ++m) dj -= Lm[j][m] * Lm[j][m] * dv[m];
      if (!(dj > W(0))) pd_ok = false;
      dv[j] = dj;
      W inv;
      if constexpr (kWidened) inv = wide_inv<R, W>(dj);
      else inv = R(1) / dj;
      idv[j] = inv;
#pragma unroll
      for (int i = j + 1; i < NX; ++i) {
        W v = Sm[i][j];
#pragma unroll
        for (int m = 0; m < j; ++m) v -= Lm[i][m] * Lm[j][m] * dv[m];
        Lm[i][j] = v * inv;
      }
    }
    auto ldl_solve = [&](const W (&b)[NX], W (&x)[NX]) {
      W y[NX];
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        W v = b[i];
#pragma unroll
        for (int m = 0; m < i; ++m) v -= Lm[i][m] * y[m];
        y[i] = v;
      }
#pragma unroll
      for (int i = NX - 1; i >= 0; --i) {
        W v;
        if constexpr (kWidened) v = y[i] * idv[i];
        else v = y[i] / dv[i];
#pragma unroll
        for (int m = i + 1; m < NX; ++m) v -= Lm[m][i] * x[m];
        x[i] = v;
      }
    };
    W rhs[NX], qw[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) rhs[i] = hv[i] - rho[i];
    ldl_solve(rhs, qw);
#pragma unroll
    for (int i = 0; i < NX; ++i) q[i] = (Q)qw[i];
    // One step of iterative refinement of q with the residual taken through the factored operator
    // (S = W^T D^-1 W is a normal-equations matrix; see mpc_fused_body.inc).  fp64 only HERE: this kernel is bound by
    // its workspace traffic and the pass re-reads W and T from HBM (+25 % bytes); the fused kernel, where the pass is
    // on-chip, refines in fp32 too, so the two fp32 pipelines differ by that one correction (both are fp32-rounding
    // level; the parity dtype is fp64, where both refine).
    if constexpr (sizeof(R) == 8) {
      R acc[NX];
#pragma unroll
      for (int r = 0; r < NX; ++r) acc[r] = R(0);
      for (int kk = 0; kk < N; ++kk) {
        R wk[NX];
        unpack<R, NX>(a.Wk[(int64_t)kk * st + p], wk);
        const V4 tk = a.Tk[(int64_t)kk * st + p];
        R om = wk[0] * q[0];
#pragma unroll
        for (int r = 1; r < NX; ++r) om += wk[r] * q[r];
        const R t = om * tk.z;  // (w_k . q) / d_k
#pragma unroll
        for (int r = 0; r < NX; ++r) acc[r] += wk[r] * t;
      }
      R rq[NX], dq[NX];
#pragma unroll
      for (int r = 0; r < NX; ++r) rq[r] = rhs[r] - Dg[r] * q[r] - acc[r];
      ldl_solve(rq, dq);
#pragma unroll
      for (int r = 0; r < NX; ++r) q[r] += dq[r];
    }
  }
  if (status == kTermNone && !pd_ok) status = kTermQpIndefinite;

  // ---- sweep 1b (wide QP only; k descending): y_k = -(gw_k + w_k . q) in double, left in the slot of gw_k -------------
  if constexpr (kWideQP) {
    Q psi[NX];
#pragma unroll
    for (int c = 0; c < NX; ++c) psi[c] = Q(Rw[c]) * q[c];  // Psi_{S-2}^T q, Psi_{S-2} = diag(w)
    Q om = Q(0);
    int kk = N - 1;
    for (int s = S - 2; s >= 0; --s) {
      for (int i = SP - 1; i >= 0; --i, --kk) {
        R gk[NX];
        unpack<R, NX>(a.Gam[(int64_t)kk * st + p], gk);
        V4 T = a.Tk[(int64_t)kk * st + p];
        Q pg = psi[0] * Q(gk[0]);
#pragma unroll
        for (int m = 1; m < NX; ++m) pg += psi[m] * Q(gk[m]);
        om = pg - Q(T.y) * om;                 // w_k . q
        T.x = (R)(-(Q(T.x) + om));             // y_k
        a.Tk[(int64_t)kk * st + p] = T;
      }
      // psi <- Phi_s^T psi for the interval below
      Q pn[NX];
#pragma unroll
      for (int c = 0; c < NX; ++c) pn[c] = Q(0);
#pragma unroll
      for (int r = 0; r < NX; ++r) {
        R row[NX];
        unpack<R, NX>(a.Phi[(int64_t)(NX * s + r) * st + p], row);
#pragma unroll
        for (int c = 0; c < NX; ++c) pn[c] += Q(row[c]) * psi[r];
      }
#pragma unroll
      for (int c = 0; c < NX; ++c) psi[c] = pn[c];
    }
  }

  // ---- sweep 2 (k ascending): U^T du = D^-1 y, state recovery, directional quantities -----------
  R gd = R(0), curv = R(0);
  R dz_inf = R(0);  // |dz|_inf, for the full-step rule of the line search (nan_max drops a NaN component: see above)
  {
    R dx[NX];
#pragma unroll
    for (int t = 0; t < NX; ++t) {
      dx[t] = -ci[t];
      dz_inf = nan_max(dz_inf, Math<R>::fabs(dx[t]));
    }
    a.dzx[p] = pack<R, NX>(dx);
    R du_prev = R(0);   // du_{k-1}; the (u_0 - u_prev) row sees only du_0
    R ups_prev = R(0);  // upsilon_{k-1}
    const XVn* __restrict__ w_p = a.Wk + p;
    const V4* __restrict__ t_p = a.Tk + p;
    const XVn* __restrict__ g_p = a.Gam + p;
    XVn W_nx, G_nx = g_p[0];  // software pipeline, one column ahead
    if constexpr (!kWideQP) W_nx = w_p[0];
    V4 T_nx = t_p[0];
    int kk = 0;
    for (int s = 0; s + 1 < S; ++s) {
      R acc[NX];
      unpack<R, NX>(a.cs[(int64_t)s * st + p], acc);
#pragma unroll
      for (int r = 0; r < NX; ++r) {
        R row[NX];
        unpack<R, NX>(a.Phi[(int64_t)(NX * s + r) * st + p], row);
#pragma unroll
        for (int m = 0; m < NX; ++m) acc[r] += row[m] * dx[m];
      }
      for (int i = 0; i < SP; ++i, ++kk) {
        R G[NX];
        unpack<R, NX>(G_nx, G);
        const V4 T = T_nx;
        R y;
        if constexpr (kWideQP) {
          y = T.x;  // sweep 1b left y_k here
        } else {
          R Wr[NX];
          unpack<R, NX>(W_nx, Wr);
          R wq = Wr[0] * (R)q[0];
#pragma unroll
          for (int m = 1; m < NX; ++m) wq += Wr[m] * (R)q[m];
          y = -(T.x + wq);
        }
        if (kk + 1 < N) {
          if constexpr (!kWideQP) W_nx = w_p[(int64_t)(kk + 1) * st];
          T_nx = t_p[(int64_t)(kk + 1) * st];
          G_nx = g_p[(int64_t)(kk + 1) * st];
        }
        const R du = y * T.z - ups_prev * du_prev;
        a.dzu[(int64_t)kk * st + p] = du;
        dz_inf = nan_max(dz_inf, Math<R>::fabs(du));
#pragma unroll
        for (int r = 0; r < NX; ++r) acc[r] += G[r] * du;
        gd += T.w * du;
        const R jd = a.wd * (du_prev - du);  // rows (u_{k-1} - u_k) w and, for k = 0, (u_0 - u_prev) w
        curv += wu2 * du * du + jd * jd + lam * du * du;
        du_prev = du;
        ups_prev = T.y;
      }
#pragma unroll
      for (int t = 0; t < NX; ++t) {
        dx[t] = acc[t];
        dz_inf = nan_max(dz_inf, Math<R>::fabs(dx[t]));
      }
      a.dzx[(int64_t)(s + 1) * st + p] = pack<R, NX>(dx);
    }
#pragma unroll
    for (int t = 0; t < NX; ++t) {
      if (Dg[t] != R(0)) {
        const R jd = Rw[t] * dx[t];
        gd += (Rw[t] * e_term[t]) * jd;
        curv += jd * jd;
      }
    }
    // ---- one step of iterative refinement of the whole QP solution (CPMPC_CREATE_REFINE_QP; see the block after
    // sweep 2 in mpc_fused_body.inc for the why): residuals of the terminal rows and of stationarity in the ORIGINAL
    // data at the recovered (du, dx), the adjoint walked back through Phi^T; a second solve with the same factors;
    // du, dx and the directional quantities replaced by the corrected ones.  Two more passes over the workspace.
    if constexpr (sizeof(R) == 8 && !kWidened) {
      // a.refine_qp passes (1: CPMPC_CREATE_REFINE_QP; 3 beyond cpmpc_max_parity_horizon(), round 6): every pass re-evaluates
      // the residuals at the corrected (du, dx, q) and solves once more with the same factors -- iterative refinement of the
      // KKT system with the condensed solve as the approximate inverse: each pass multiplies a lane's error by that solve's
      // relative error on the lane (1e-3 at worst at 1.6 s), where the condensed solve alone is left with it.
#pragma unroll 1
      for (int pass = 0; pass < a.refine_qp; ++pass) {
        R viol[NX], lamv[NX];
#pragma unroll
        for (int t = 0; t < NX; ++t) {
          const R rT = Rw[t] * (dx[t] + e_term[t]);
          const bool cost = Dg[t] != R(0);
          lamv[t] = Rw[t] * (cost ? rT : q[t]);  // the adjoint at the terminal node, diag(Rw) mult
          viol[t] = cost ? R(0) : rT;
        }
        R rho2[NX];
#pragma unroll
        for (int t = 0; t < NX; ++t) rho2[t] = R(0);
        {  // descending: rstat_k = (T du)_k + g_k + Gamma_k . lambda_s, gw'_k = rstat_k - ups_k gw'_{k+1}, rho' += w_k gw'_k / d_k
          R gw_next = R(0);
          R du_hi = R(0), du_cur = a.dzu[(int64_t)(N - 1) * st + p];
          R u_hi = R(0), u_cur = a.zu[(int64_t)(N - 1) * st + p];
          int k2 = N - 1;
          for (int s = S - 2; s >= 0; --s) {
            for (int i = SP - 1; i >= 0; --i, --k2) {
              const bool inner = k2 < N - 1;
              const R du_lo = (k2 > 0) ? a.dzu[(int64_t)(k2 - 1) * st + p] : R(0);
              const R u_lo = (k2 > 0) ? a.zu[(int64_t)(k2 - 1) * st + p] : u_prev;
              R g = wu2 * u_cur + wd2 * (u_cur - u_lo);
              if (inner) g += wd2 * (u_cur - u_hi);
              R G[NX], Wr[NX];
              unpack<R, NX>(a.Gam[(int64_t)k2 * st + p], G);
              unpack<R, NX>(a.Wk[(int64_t)k2 * st + p], Wr);
              V4 T = a.Tk[(int64_t)k2 * st + p];
              R rstat = (wu2 + lam + wd2 * ((inner ? R(1) : R(0)) + R(1))) * du_cur - wd2 * du_lo + g;
              if (inner) rstat -= wd2 * du_hi;
#pragma unroll
              for (int m = 0; m < NX; ++m) rstat += G[m] * lamv[m];
              const R gw2 = rstat - T.y * gw_next;
              const R t2 = gw2 * T.z;
#pragma unroll
              for (int m = 0; m < NX; ++m) rho2[m] += Wr[m] * t2;
              T.x = gw2;
              a.Tk[(int64_t)k2 * st + p] = T;
              gw_next = gw2;
              du_hi = du_cur;
              du_cur = du_lo;
              u_hi = u_cur;
              u_cur = u_lo;
            }
            R ln[NX];
#pragma unroll
            for (int c = 0; c < NX; ++c) ln[c] = R(0);
#pragma unroll
            for (int r = 0; r < NX; ++r) {
              R row[NX];
              unpack<R, NX>(a.Phi[(int64_t)(NX * s + r) * st + p], row);
#pragma unroll
              for (int c = 0; c < NX; ++c) ln[c] += row[c] * lamv[r];
            }
#pragma unroll
            for (int c = 0; c < NX; ++c) lamv[c] = ln[c];
          }
        }
        R dq[NX];
        {  // (S + Dg) dq = viol - rho' with the factors of the first solve
          R y[NX];
#pragma unroll
          for (int i = 0; i < NX; ++i) {
            R v = viol[i] - rho2[i];
#pragma unroll
            for (int m = 0; m < i; ++m) v -= Lm[i][m] * y[m];
            y[i] = v;
          }
#pragma unroll
          for (int i = NX - 1; i >= 0; --i) {
            R v = y[i] / dv[i];
#pragma unroll
            for (int m = i + 1; m < NX; ++m) v -= Lm[m][i] * dq[m];
            dq[i] = v;
          }
        }
#pragma unroll
        for (int t = 0; t < NX; ++t) q[t] += dq[t];  // the multipliers of the next pass's adjoint
        // ascending: the correction of the step, applied; the directional quantities from the corrected step
        gd = R(0);
        curv = R(0);
        dz_inf = R(0);
        R ddx[NX];
#pragma unroll
        for (int t = 0; t < NX; ++t) {
          ddx[t] = R(0);  // dx_0 = -c_init is exact
          dz_inf = nan_max(dz_inf, Math<R>::fabs(ci[t]));
        }
        R ddu_prev = R(0), du_prev2 = R(0), ups_prev2 = R(0);
        int k3 = 0;
        for (int s = 0; s + 1 < S; ++s) {
          R acc[NX];
#pragma unroll
          for (int r = 0; r < NX; ++r) {
            R row[NX];
            unpack<R, NX>(a.Phi[(int64_t)(NX * s + r) * st + p], row);
            R v = R(0);
#pragma unroll
            for (int m = 0; m < NX; ++m) v += row[m] * ddx[m];
            acc[r] = v;
          }
          for (int i = 0; i < SP; ++i, ++k3) {
            R G[NX], Wr[NX];
            unpack<R, NX>(a.Gam[(int64_t)k3 * st + p], G);
            unpack<R, NX>(a.Wk[(int64_t)k3 * st + p], Wr);
            const V4 T = a.Tk[(int64_t)k3 * st + p];
            R wq = Wr[0] * dq[0];
#pragma unroll
            for (int m = 1; m < NX; ++m) wq += Wr[m] * dq[m];
            const R ddu = -(T.x + wq) * T.z - ups_prev2 * ddu_prev;
            const R du = a.dzu[(int64_t)k3 * st + p] + ddu;
            a.dzu[(int64_t)k3 * st + p] = du;
            dz_inf = nan_max(dz_inf, Math<R>::fabs(du));
#pragma unroll
            for (int r = 0; r < NX; ++r) acc[r] += G[r] * ddu;
            gd += T.w * du;
            const R jd = a.wd * (du_prev2 - du);
            curv += wu2 * du * du + jd * jd + lam * du * du;
            ddu_prev = ddu;
            du_prev2 = du;
            ups_prev2 = T.y;
          }
          R xn[NX];
          unpack<R, NX>(a.dzx[(int64_t)(s + 1) * st + p], xn);
#pragma unroll
          for (int t = 0; t < NX; ++t) {
            ddx[t] = acc[t];
            xn[t] += acc[t];
            dz_inf = nan_max(dz_inf, Math<R>::fabs(xn[t]));
          }
          a.dzx[(int64_t)(s + 1) * st + p] = pack<R, NX>(xn);
#pragma unroll
          for (int t = 0; t < NX; ++t) dx[t] = xn[t];
        }
#pragma unroll
        for (int t = 0; t < NX; ++t) {
          if (Dg[t] != R(0)) {
            const R jd = Rw[t] * dx[t];
            gd += (Rw[t] * e_term[t]) * jd;
            curv += jd * jd;
          }
        }
      }
    }
  }
  if (status == kTermNone && (!Math<R>::finite(gd) || !Math<R>::finite(curv))) status = kTermQpIndefinite;

  // ---- the merit at the iterate, evaluated the way the trials are --------------------------------------------
  // f and cn above come out of the linearisation (RK4 with sensitivities, sums in sweep order); the line search
  // evaluates its trial points with the Jacobian-free rollout in another order of operations.  Both are correct to
  // rounding, but comparing phi(trial) from one path against phi(iterate) from the other puts a path-to-path rounding
  // difference (~1e-13 relative, ~1e-10 absolute at f ~ 1e3) into every Armijo test, and the iteration stalls once
  // the achievable decrease falls below it (the CPU restatement this path is tested against evaluates both sides with
  // one function).  The iterate IS the last accepted trial point, bit for bit, so its merit
  // pieces as the line search computed them are at hand: use those.
  const bool have_trial = a.sc[SC_TRIAL * st + p] != R(0);
  if (have_trial) {
    f = a.sc[SC_F_LAST * st + p];
    cn = a.sc[SC_CN_LAST * st + p];
  }
  // ---- penalty update (Nocedal & Wright 18.36, sigma = 1), merit slope --------------------------
  if (cn > R(0)) {
    const R mu_req = (gd + R(0.5) * curv) / ((R(1) - a.rho) * cn);
    if (mu < mu_req) mu = mu_req;
  }
  const R D = gd - mu * cn;
  const R phi0 = f + mu * cn;
  // equality residuals at the rounding floor of the rollout count as zero in the exit test
  bool first_order;
  if constexpr (sizeof(R) == 4 || CPMPC_EXIT_FLOOR_F64) {   // (the double kernels: see CPMPC_EXIT_FLOOR_F64)
    R x_l1 = R(0);   // the size of the states: target and distance to it at the terminal node, times the number of intervals
#pragma unroll
    for (int t = 0; t < NX; ++t) x_l1 += Math<R>::fabs(tgt[t]) + Math<R>::fabs(e_term[t]);
    const R D_exit = gd - mu * (cn > a.cn_floor_scale * R(S - 1) * x_l1 ? cn : R(0));
    first_order = Math<R>::fabs(D_exit) < a.fo_tol;
  } else {
    first_order = Math<R>::fabs(D) < a.fo_tol;
  }

  // ---- Armijo line search, lock-step over the wave ----------------------------------------------
  // Local convergence safeguard: a QP step that is tiny in every component is taken in full
  // without the merit test (the l1 merit cannot resolve the decrease such a step brings, and rejects it)
  // (only the undamped step: one that is small because lambda is large is no sign of convergence)
  const bool tiny = dz_inf <= a.full_step_below && lam == R(0);
  // converged (first-order test and a tiny undamped step): the full step without a merit evaluation
  const bool skip_merit = CPMPC_SKIP_MERIT && tiny && first_order;
  bool active = (status == kTermNone) && !skip_merit;
  bool accepted = (status == kTermNone) && skip_merit;
  R alpha = tiny ? R(1) : a_start, phi_t = R(0), f_t = f, cn_t = cn;
  int evals = 0;
  for (int t = 0; t < a.max_ls; ++t) {
    if (!__any(active)) break;
    if (active) {
      R ft, ct;
      merit_eval<R, M>(a, k, p, alpha, xm, tgt, u_prev, ft, ct);
      ++evals;
      phi_t = ft + mu * ct;
      if (phi_t <= phi0 + a.c1 * alpha * D || (tiny && Math<R>::finite(phi_t))) {
        accepted = true;
        active = false;
        f_t = ft;
        cn_t = ct;
      } else {
        const R denom = R(2) * (phi_t - phi0 - D * alpha);
        R a_new = (denom > R(0)) ? (-D * alpha * alpha / denom) : (a.shrink_max * alpha);
        // a diverged trial (merit NaN / inf) is an overlong step like a finite astronomically large one: lower safeguard
        if (!Math<R>::finite(phi_t)) a_new = a.shrink_min * alpha;
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
      a_start = (evals > 1 ? a.alpha_growth_bt : a.alpha_growth) * alpha;
      if (!(a_start < R(1))) a_start = R(1);
    }
    if (accepted) {
      for (int s = 0; s < S; ++s) {
        R xs[NX];
        trial_node<R, M>(a, s, p, alpha, xs);
        a.zx[(int64_t)s * st + p] = pack<R, NX>(xs);
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
      if (a.rel_tol > R(0) && (phi0 - phi_t) < a.rel_tol * phi0) status = kTermRelTol;
    } else {
      ++failed;
      lam = (lam > R(0)) ? lam * a.lam_up : a.lam_fail_init;
      if (lam > a.lam_max) status = kTermMaxLambda;
    }
  }
  if (status != kTermNonFinite) a.ist[IS_ITERS * st + p] += 1;
  a.sc[SC_LAMBDA * st + p] = lam;
  a.sc[SC_MU * st + p] = mu;
  a.sc[SC_F_LAST * st + p] = f_t;   // accepted: the trial's pieces; else f, cn as used above (unchanged iterate)
  a.sc[SC_CN_LAST * st + p] = cn_t;
  if (accepted && status != kTermNonFinite) a.sc[SC_TRIAL * st + p] = skip_merit ? R(0) : R(1);
  a.sc[SC_ALPHA * st + p] = a_start;
  a.ist[IS_STATUS * st + p] = status;
  a.ist[IS_LS_EVALS * st + p] += evals;
  a.ist[IS_FAILED * st + p] = failed;
}

// ------------------------------------------------------------------------------------------------
// finalize: predicted states (optimization.cc:353-371) and outputs.  One thread per problem.
// ------------------------------------------------------------------------------------------------
template <typename R, typename M>
// launched with 256-thread workgroups: 4x fewer workgroups to dispatch for a kernel that is a single round of waves
// (measured 92 -> 80 us at B = 262 144)
__global__ __launch_bounds__(CPMPC_PF_BLOCK) void finalize_kernel(const SolverArgs<R, M> a) {
  constexpr int NX = M::NX;
  const unsigned p = blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t st = a.stride;
  if (a.fb_host != nullptr && blockIdx.x % (unsigned)a.fb_stride == 0u) {
    // How many problems ran how many iterations: the host reads it (no copy, no synchronisation: host-mapped memory)
    // before a later step and plans that step's stages from it.  No device-wide reduction: every fb_stride-th workgroup
    // writes its own 16 counts and the host adds them up (at most 256 x 16 ints).  (A first version summed all
    // workgroups with device-scope atomics and published the total behind a system-scope release: 29 -> 92 us.)
    __shared__ int bins[kFbBins];
    if (threadIdx.x < kFbBins) bins[threadIdx.x] = 0;
    __syncthreads();
    int bin = -1;
    if (p < a.B) {
      const int it = a.ist[IS_ITERS * st + p];
      bin = it < kFbBins - 1 ? (it < 0 ? 0 : it) : kFbBins - 1;
    }
    for (int b = 0; b < kFbBins; ++b) {
      const unsigned long long m = __ballot(bin == b);
      if (m != 0ull && (threadIdx.x & 63) == 0) atomicAdd(&bins[b], __popcll(m));
    }
    __syncthreads();
    int32_t* const dst = a.fb_host + (size_t)(blockIdx.x / (unsigned)a.fb_stride) * (kFbBins + 1);
    // relaxed stores, no system-scope fence: a release at system scope writes the whole L2 back (this kernel's own
    // outputs: 29 -> 94 us measured), and all the host needs is a hint -- a count read half-updated costs nothing but speed
    // (the host reads the stamp before and after the counts and drops a block whose stamp moved: cpmpc_plan_stages)
    if (threadIdx.x < kFbBins) __hip_atomic_store(&dst[threadIdx.x], bins[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&dst[kFbBins], a.fb_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (p >= a.B) return;
  const int64_t ob = a.B;  // outputs are packed [field][B]
  int status = a.ist[IS_STATUS * st + p];
  if (status == kTermNone) status = kTermMaxIterations;
  if (a.status_out) a.status_out[p] = status;
  if (a.iters_out) a.iters_out[p] = a.ist[IS_ITERS * st + p];
  if (a.ls_out) a.ls_out[p] = a.ist[IS_LS_EVALS * st + p];
  if (a.cost_out) a.cost_out[p] = a.sc[SC_F_LAST * st + p];
  if (a.eq_out) a.eq_out[p] = a.sc[SC_CN_LAST * st + p];
  if (a.sol_out) {  // previous_solution_ = solver_->variables() (optimization.cc:85), packed in MapKey order
    for (int s = 0; s < a.S; ++s) {
      R xv[NX];
      unpack<R, NX>(a.zx[(int64_t)s * st + p], xv);
#pragma unroll
      for (int t = 0; t < NX; ++t) a.sol_out[(int64_t)(NX * s + t) * ob + p] = xv[t];
    }
    for (int i = 0; i < a.N; ++i) a.sol_out[(int64_t)(NX * a.S + i) * ob + p] = a.zu[(int64_t)i * st + p];
  }
  if (a.pred_out) {
    const typename M::Consts k = load_consts(a, p);
    const ExtForce<R> fe{R(0), R(0), R(0)};
    R x[NX];
#pragma unroll
    for (int t = 0; t < NX; ++t) x[t] = a.x0[t * ob + p];
    R u_next = a.zu[p];
    typename M::StepCache chain;
    for (int kk = 0; kk < a.N; ++kk) {
      const R u = u_next;
      if (kk + 1 < a.N) u_next = a.zu[(int64_t)(kk + 1) * st + p];
      if (a.u_out) a.u_out[(int64_t)kk * ob + p] = u;
      if (kk % 8 == 0) chain.invalidate();  // re-anchor the sine / cosine chain with a full evaluation every 8 steps
      rk4_step_m<R, M, false>(k, a.dt, x, u, fe, chain);
      wrap_angles<R, M>(x);
#pragma unroll
      for (int t = 0; t < NX; ++t) a.pred_out[((int64_t)kk * NX + t) * ob + p] = x[t];
    }
  } else if (a.u_out) {
    for (int kk = 0; kk < a.N; ++kk) a.u_out[(int64_t)kk * ob + p] = a.zu[(int64_t)kk * st + p];
  }
}

// ------------------------------------------------------------------------------------------------
// layout conversion between the packed external z [NX*S+N][B] (MapKey order) and the workspace
// ------------------------------------------------------------------------------------------------
template <typename R, int NX>
__global__ __launch_bounds__(64) void pack_z_kernel(int64_t B, int64_t st, int S, int N, const R* z_ext,
                                                     XV<R, NX>* zx, R* zu) {
  const unsigned p = blockIdx.x * 64u + threadIdx.x;
  if (p >= B) return;
  for (int s = 0; s < S; ++s) {
    R x[NX];
#pragma unroll
    for (int t = 0; t < NX; ++t) x[t] = z_ext[(int64_t)(NX * s + t) * B + p];
    zx[(int64_t)s * st + p] = pack<R, NX>(x);
  }
  for (int i = 0; i < N; ++i) zu[(int64_t)i * st + p] = z_ext[(int64_t)(NX * S + i) * B + p];
}

template <typename R, int NX>
__global__ __launch_bounds__(64) void unpack_z_kernel(int64_t B, int64_t st, int S, int N,
                                                       const XV<R, NX>* zx, const R* zu, R* z_ext) {
  const unsigned p = blockIdx.x * 64u + threadIdx.x;
  if (p >= B) return;
  for (int s = 0; s < S; ++s) {
    R x[NX];
    unpack<R, NX>(zx[(int64_t)s * st + p], x);
#pragma unroll
    for (int t = 0; t < NX; ++t) z_ext[(int64_t)(NX * s + t) * B + p] = x[t];
  }
  for (int i = 0; i < N; ++i) z_ext[(int64_t)(NX * S + i) * B + p] = zu[(int64_t)i * st + p];
}

// workspace linearisation -> packed c [NX(S-1)][B], Phi [NX*NX(S-1)][B] (row-major), Gamma [NX*N][B] (NX*k+r)
template <typename R, typename M>
__global__ __launch_bounds__(64) void unpack_lin_kernel(const SolverArgs<R, M> a, R* c, R* Phi, R* Gam) {
  constexpr int NX = M::NX;
  const unsigned p = blockIdx.x * 64u + threadIdx.x;
  if (p >= a.B) return;
  const int64_t st = a.stride, B = a.B;
  for (int s = 0; s + 1 < a.S; ++s) {
    R v[NX];
    unpack<R, NX>(a.cs[(int64_t)s * st + p], v);
#pragma unroll
    for (int t = 0; t < NX; ++t) c[(int64_t)(NX * s + t) * B + p] = v[t];
    for (int r = 0; r < NX; ++r) {
      unpack<R, NX>(a.Phi[(int64_t)(NX * s + r) * st + p], v);
#pragma unroll
      for (int t = 0; t < NX; ++t) Phi[(int64_t)(NX * NX * s + NX * r + t) * B + p] = v[t];
    }
  }
  for (int i = 0; i < a.N; ++i) {
    R v[NX];
    unpack<R, NX>(a.Gam[(int64_t)i * st + p], v);
#pragma unroll
    for (int t = 0; t < NX; ++t) Gam[(int64_t)(NX * i + t) * B + p] = v[t];
  }
}

// ------------------------------------------------------------------------------------------------
// stand-alone pieces (parity tests, callers that want them); arrays packed [field][B]
// ------------------------------------------------------------------------------------------------
template <typename R, typename M>
__global__ __launch_bounds__(64) void dynamics_kernel(int64_t B, typename M::Consts k, ExtForce<R> fe,
                                                       const R* x, const R* u, R* f, R* Jx, R* Ju) {
  constexpr int NX = M::NX, NQ = M::NQ;
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= B) return;
  R xs[NX], acc[NQ], Ja[NQ][NX], Jua[NQ];
#pragma unroll
  for (int t = 0; t < NX; ++t) xs[t] = x[t * B + p];
  M::template accel<true, true>(k, xs, u[p], fe, acc, Ja, Jua);
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    f[i * B + p] = xs[NQ + i];
    f[(NQ + i) * B + p] = acc[i];
  }
  if (Jx) {
#pragma unroll
    for (int i = 0; i < NQ; ++i)
#pragma unroll
      for (int c = 0; c < NX; ++c) {
        Jx[(i * NX + c) * B + p] = (c == NQ + i) ? R(1) : R(0);
        Jx[((NQ + i) * NX + c) * B + p] = Ja[i][c];
      }
  }
  if (Ju) {
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      Ju[i * B + p] = R(0);
      Ju[(NQ + i) * B + p] = Jua[i];
    }
  }
}

template <typename R, typename M>
__global__ __launch_bounds__(64) void rk4_kernel(int64_t B, typename M::Consts k, ExtForce<R> fe, R h,
                                                  const R* x, const R* u, R* xn, R* A, R* Bm) {
  constexpr int NX = M::NX;
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= B) return;
  R xs[NX];
#pragma unroll
  for (int t = 0; t < NX; ++t) xs[t] = x[t * B + p];
  if (A != nullptr || Bm != nullptr) {
    R Am[NX][NX], Bv[NX];
    rk4_step_jac_m<R, M, true>(k, h, xs, u[p], fe, Am, Bv);
    if (A)
#pragma unroll
      for (int r = 0; r < NX; ++r)
#pragma unroll
        for (int c = 0; c < NX; ++c) A[(r * NX + c) * B + p] = Am[r][c];
    if (Bm)
#pragma unroll
      for (int r = 0; r < NX; ++r) Bm[r * B + p] = Bv[r];
  } else {
    rk4_step_m<R, M, true>(k, h, xs, u[p], fe);
  }
#pragma unroll
  for (int t = 0; t < NX; ++t) xn[t * B + p] = xs[t];
}

// Simulator::Step (simulator.cc:11-36): fixed 1 ms sub-steps, angles wrapped after each.
template <typename R, typename M>
__global__ __launch_bounds__(64) void sim_kernel(int64_t B, typename M::Consts k, ExtForce<R> fe_shared,
                                                  const R* fext, int n_sub, R h_last, const R* u, R* state) {
  constexpr int NX = M::NX;
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= B) return;
  ExtForce<R> fe = fe_shared;
  if (fext) {
    fe.fbx = fext[p];
    fe.fmx = fext[2 * B + p];
    fe.fmy = fext[3 * B + p];
  }
  R xs[NX];
#pragma unroll
  for (int t = 0; t < NX; ++t) xs[t] = state[t * B + p];
  const R uu = u[p];
  // the host evaluates the reference's `while (dt > 0) { SubStep(min(dt, 0.001)); dt -= 0.001; }`
  // in double and passes the count and the last step, so f32 and f64 take the same sub-steps
  const R internal_dt = R(0.001);
  typename M::StepCache chain;
  for (int i = 0; i < n_sub; ++i) {
    const R h = (i + 1 == n_sub) ? h_last : internal_dt;
    if (i % 8 == 0) chain.invalidate();  // re-anchor the sine / cosine chain every 8 sub-steps
    rk4_step_m<R, M, true>(k, h, xs, uu, fe, chain);
    wrap_angles<R, M>(xs);
  }
#pragma unroll
  for (int t = 0; t < NX; ++t) state[t * B + p] = xs[t];
}

}  // namespace cpmpc
