// engine_impl.hpp -- host side of the kernels of ONE (dtype, model) pair: argument set-up, launches, staging of the
// host-pointer calls.  Included by engine_<dtype>_<model>.hip only, which instantiates everything for its pair and
// exports it as an `Engine` table (engine.hpp); the C-ABI unit never sees device code.
#pragma once
#include <cmath>
#include <cstring>
#include <limits>

#include "engine.hpp"
#include "mpc_kernels.hpp"
#include "mpc_fused.hpp"

using namespace cpmpc;

// ------------------------------------------------------------------------------------------------
// launch helpers
// ------------------------------------------------------------------------------------------------
static inline dim3 grid_for(int64_t threads) { return dim3((unsigned)((threads + 63) / 64)); }

template <typename R, typename M>
static void fill_args(const cpmpc_solver* s, int64_t B, SolverArgs<R, M>& a, int64_t col0 = 0) {
  const cpmpc_params& p = s->params;
  const cpmpc_solver_opts& o = s->opts;
  memset((void*)&a, 0, sizeof a);
  a.B = B;
  a.stride = s->cap;
  a.N = s->N;
  a.S = s->S;
  a.SP = s->SP;
  a.iter_cap = (int)p.max_iterations;
  a.dt = (R)p.control_dt;
  // rows exist only for strictly positive weights (optimization.cc:270,296)
  a.wu = (R)(p.u_cost_weight > 0.0 ? p.u_cost_weight : 0.0);
  a.wd = (R)(p.u_derivative_cost_weight > 0.0 ? p.u_derivative_cost_weight : 0.0);
  // terminal rows in BuildProblem order (optimization.cc:236-267).  For the double pendulum (no optimizer
  // in the reference) th_final / th_dot_final apply to both poles, both targets are upright.
  a.term_is_cost = 0;
  for (int t = 0; t < M::NX; ++t) {
    double w, tgt;
    if (t == 0) {
      w = p.b_x_final_cost_weight;
      tgt = 0.0;
    } else if (t < M::NQ) {
      w = p.th_final_cost_weight;
      tgt = M_PI / 2;
    } else if (t == M::NQ) {
      w = p.b_x_dot_final_cost_weight;
      tgt = 0.0;
    } else {
      w = p.th_dot_final_cost_weight;
      tgt = 0.0;
    }
    const bool is_cost = w >= 0.0;
    a.term_w[t] = (R)(is_cost ? w : 1.0);
    a.term_tgt[t] = (R)tgt;
    if (is_cost) a.term_is_cost |= (1 << t);
  }
  a.max_ls = o.max_line_search_iterations;
  a.c1 = (R)o.armijo_c1;
  a.shrink_max = (R)o.ls_shrink_max;
  a.shrink_min = (R)o.ls_shrink_min;
  a.alpha_growth = (R)o.ls_alpha_growth;
  a.alpha_growth_bt = (R)o.ls_alpha_growth_backtracked;
  a.full_step_below = (R)o.full_step_below;
  a.cn_floor_scale = (R)(o.exit_defect_floor * (double)s->SP * (double)std::numeric_limits<R>::epsilon());
  a.rho = (R)o.penalty_rho;
  a.lam_init = (R)o.lambda_initial;
  a.lam_fail_init = (R)o.lambda_failure_init;
  a.lam_up = (R)o.lambda_scale_up;
  a.lam_down = (R)o.lambda_scale_down;
  a.lam_min = (R)o.lambda_min;
  a.lam_max = (R)o.lambda_max;
  a.bx_lim = (R)o.b_x_limit;
  a.u_lim = (R)o.u_limit;
  a.rel_tol = (R)p.relative_exit_tol;
  a.fo_tol = (R)p.absolute_first_derivative_tol;
  a.mu_init = (R)p.equality_penalty_initial;
  // problems [col0, col0 + B) of the workspace: every field is [field][cap] with the problem index fastest, so the call
  // works on a column range by offsetting the field bases (a chunk of a pipelined host-pointer step)
  a.prev_B = s->prev_B - col0 < 0 ? 0 : (s->prev_B - col0 > B ? B : s->prev_B - col0);
  a.refine_qp = s->refine_qp ? s->refine_passes : 0;
  using V4 = typename VecT<R>::V4;
  using XVn = XV<R, M::NX>;
  a.zx = (XVn*)s->zx + col0;
  a.zu = (R*)s->zu + col0;
  a.dzx = (XVn*)s->dzx + col0;
  a.dzu = (R*)s->dzu + col0;
  a.Phi = (XVn*)s->Phi + col0;
  a.Gam = (XVn*)s->Gam + col0;
  a.cs = (XVn*)s->cs + col0;
  a.Wk = (XVn*)s->Wk + col0;
  a.Tk = (V4*)s->Tk + col0;
  a.sc = (R*)s->sc + col0;
  a.ist = s->ist + col0;
  a.sin_table = (const R*)s->sin_table;
}

template <typename R, typename M>
static void launch_linearize(const SolverArgs<R, M>& a, int SP, const XV<R, M::NX>* zx_in, const R* zu_in,
                             const int32_t* status, hipStream_t stream) {
  const dim3 grid = grid_for(a.B * (a.S - 1));
#define CPMPC_LIN(SPV)                                                                                       \
  case SPV:                                                                                                  \
    hipLaunchKernelGGL((linearize_kernel<R, M, SPV>), grid, dim3(64), 0, stream, a, zx_in, zu_in, status);   \
    break;
  switch (SP) {
    CPMPC_LIN(1)
    CPMPC_LIN(2)
    CPMPC_LIN(4)
    CPMPC_LIN(5)
    CPMPC_LIN(8)
    CPMPC_LIN(10)
    CPMPC_LIN(20)
    default:  // no register-resident specialisation: run-time spacing, Gamma accumulated in the workspace
      hipLaunchKernelGGL((linearize_dyn_kernel<R, M>), grid, dim3(64), 0, stream, a, zx_in, zu_in, status);
      break;
  }
#undef CPMPC_LIN
}

// 1: the fp64 fused kernels also take batch-shared model constants from the kernel-argument segment (SGPRs).  Off by
// default: see the note above launch_fused (tools/_build variant `shared64` measures it).
#ifndef CPMPC_FUSED_SHARED_F64
#define CPMPC_FUSED_SHARED_F64 0
#endif

// The shared-parameters specialisation (model constants wave-uniform, in SGPRs) is used in fp32 only: the fp64
// kernel already sits at the SGPR limit with its VGPR/AGPR file exhausted, and with the constants added to the
// scalar pressure hipcc 7.2 produced wrong results for it (caught by the fp64 parity tests); there the constants
// go through load_consts() into vector registers like per-problem parameters do.
template <typename R, typename M>
static void launch_fused(const SolverArgs<R, M>& a, int L, int SP, int max_iters, bool refine, hipStream_t stream) {
  {
    const int ppw = 64 / L;
    const dim3 grid((unsigned)((a.B + ppw - 1) / ppw));
    // REFINE = true: the double kernels' refinement of the whole QP solution (CPMPC_CREATE_REFINE_QP), the float kernels'
    // QP in double (CPMPC_CREATE_WIDE_QP); constants through vector registers there (SHARED = false)
#define CPMPC_FUSED(LV, SPV)                                                                                \
  if (L == LV && SP == SPV) {                                                                               \
    { /* double: REFINE_QP; float: WIDE_QP */                                                               \
      if (refine) {                                                                                         \
        hipLaunchKernelGGL((fused_sqp_kernel<R, M, SPV, LV, false, true>), grid, dim3(64), 0, stream, a, max_iters); \
        return;                                                                                             \
      }                                                                                                     \
    }                                                                                                       \
    if constexpr (sizeof(R) == 4 || CPMPC_FUSED_SHARED_F64) {                                              \
      if (a.dyn == nullptr) {                                                                               \
        hipLaunchKernelGGL((fused_sqp_kernel<R, M, SPV, LV, true, false>), grid, dim3(64), 0, stream, a, max_iters); \
        return;                                                                                             \
      }                                                                                                     \
    }                                                                                                       \
    hipLaunchKernelGGL((fused_sqp_kernel<R, M, SPV, LV, false, false>), grid, dim3(64), 0, stream, a, max_iters);  \
    return;                                                                                                 \
  }
    CPMPC_FUSED(4, 10)
    CPMPC_FUSED(8, 5)
    CPMPC_FUSED(2, 10)
    CPMPC_FUSED(4, 5)
    CPMPC_FUSED(2, 20)
    CPMPC_FUSED(5, 8)
    CPMPC_FUSED(10, 4)
#undef CPMPC_FUSED
    // no specialisation for this spacing: run-time SP, dynamic LDS
    const size_t lds = fused_dyn_lds_bytes<R, M>(SP);
#define CPMPC_FUSED_DYN(LV)                                                                                         \
  if (L == LV) {                                                                                                    \
    { /* double: REFINE_QP; float: WIDE_QP */                                                                       \
      if (refine) {                                                                                                 \
        hipLaunchKernelGGL((fused_sqp_dyn_kernel<R, M, LV, false, true>), grid, dim3(64), lds, stream, a, max_iters); \
        return;                                                                                                     \
      }                                                                                                             \
    }                                                                                                               \
    if constexpr (sizeof(R) == 4 || CPMPC_FUSED_SHARED_F64) {                                                      \
      if (a.dyn == nullptr) {                                                                                       \
        hipLaunchKernelGGL((fused_sqp_dyn_kernel<R, M, LV, true, false>), grid, dim3(64), lds, stream, a, max_iters); \
        return;                                                                                                     \
      }                                                                                                             \
    }                                                                                                               \
    hipLaunchKernelGGL((fused_sqp_dyn_kernel<R, M, LV, false, false>), grid, dim3(64), lds, stream, a, max_iters);  \
    return;                                                                                                         \
  }
    CPMPC_FUSED_DYN(2)
    CPMPC_FUSED_DYN(4)
    CPMPC_FUSED_DYN(5)
    CPMPC_FUSED_DYN(8)
    CPMPC_FUSED_DYN(10)
    CPMPC_FUSED_DYN(16)
#undef CPMPC_FUSED_DYN
  }
}

template <typename R, typename M>
static int step_batch_impl(cpmpc_solver* s, int64_t B, const cpmpc_step_inputs* in, const cpmpc_step_outputs* out,
                           hipStream_t stream, int64_t col0, int slot) {
  SolverArgs<R, M> a;
  fill_args<R, M>(s, B, a, col0);
  a.x0 = (const R*)in->x0;
  a.dyn = (const R*)in->dyn;
  a.set_point = (const R*)in->set_point;
  a.term_w_pp = (const R*)in->terminal_weights;
  if (in->dyn == nullptr) a.consts = M::template make<double>(in->dyn_shared_host);
  a.term_tgt[0] = (R)in->set_point_shared;
  if (out) {
    a.u_out = (R*)out->u;
    a.pred_out = (R*)out->predicted;
    a.status_out = out->status;
    a.iters_out = out->iterations;
    a.ls_out = out->ls_evals;
    a.cost_out = (R*)out->final_cost;
    a.eq_out = (R*)out->final_eq_l1;
    a.guess_out = (R*)out->guess;
    a.sol_out = (R*)out->solution;
  }
  const dim3 gridB = grid_for(B);
  ProfSpan sp;
  // float handles: the QP with its terminal part in double (CPMPC_CREATE_WIDE_QP).  A property of the HANDLE, never of the step:
  // round 6 tried "wide on cold-start steps only" as the 4-state default and withdrew it -- whether a call is a cold start
  // depends on which problems share the call, so a sharded handle (whose last shard may be all cold) and a single one (partly
  // warm) then ran different kernels on the same problem and batch-position independence was gone
  // (test_sharded_device_step_takes_per_problem_inputs...: bitwise equality across shardings)
  const bool wide_step = s->wide_qp;

  // (the first compaction's counter: cleared by prepare_kernel whether or not this step turns out to be staged)
  a.stage_count0 = (use_fused(s) && s->active != nullptr) ? s->active + s->cap + 3 * slot : nullptr;
  span_begin(s, CPMPC_KERNEL_PREPARE, stream, &sp);
  hipLaunchKernelGGL((prepare_kernel<R, M>), dim3((unsigned)((B + CPMPC_PF_BLOCK - 1) / CPMPC_PF_BLOCK)), dim3(CPMPC_PF_BLOCK), 0, stream, a);
  span_end(s, stream, &sp);

  if (use_fused(s)) {
    // With exit tolerances enabled problems stop after different numbers of iterations, and a wave lives as long
    // as its slowest problem (closed loop, measured: 4.3 iterations per problem, 7.7 per wave of 16).  The kernel
    // is restartable -- all solver state is in the workspace between launches -- so it runs in stages and the
    // problems still iterating are compacted into dense waves in between.  Results are bitwise those of a single
    // launch: a problem's arithmetic does not depend on the lanes it occupies.  Where the stages end is planned per
    // step (cpmpc_plan_stages: an explicit cpmpc_set_compaction, or from how many iterations the problems of an earlier
    // step needed).
    const int total = (int)s->params.max_iterations;
    const bool exits = s->params.relative_exit_tol > 0.0 || s->params.absolute_first_derivative_tol > 0.0;
    int bounds[kMaxStages + 1];
    const int n_stages = cpmpc_plan_stages(s, slot, B, exits, bounds);
    a.active_list = nullptr;
    a.active_count = nullptr;
    a.iter_cap = total;
    a.run_out_below = (int64_t)2048 * (64 / (s->S - 1));  // problems in one round of resident waves (2 per SIMD)
    span_begin(s, CPMPC_KERNEL_FUSED, stream, &sp);
    launch_fused<R, M>(a, s->S - 1, s->SP, bounds[1], sizeof(R) == 8 ? s->refine_qp : wide_step, stream);
    span_end(s, stream, &sp);
    for (int stage = 0; stage + 1 < n_stages; ++stage) {
      const int done = bounds[stage + 1];
      // three counters per host slot (chunks of a pipelined host step run concurrently), rotating: this compaction's, the
      // one before (prev_count), and the next one's, which this compaction clears; prepare_kernel cleared counters[0]
      int32_t* const counters = s->active + s->cap + 3 * slot;
      int32_t* count = counters + stage % 3;
      a.prev_count = stage ? counters + (stage - 1) % 3 : nullptr;
      a.prev_total = B;
      a.remaining = total - done;
      span_begin(s, CPMPC_KERNEL_FUSED, stream, &sp);
      hipLaunchKernelGGL((compact_active_kernel<M>), dim3((unsigned)((B + 1023) / 1024)), dim3(1024), 0, stream,
                         (const int32_t*)(a.ist + (size_t)IS_STATUS * (size_t)s->cap),
                         (const int32_t*)(a.ist + (size_t)IS_ITERS * (size_t)s->cap), total, B, s->active + col0, count,
                         counters + (stage + 1) % 3);
      a.active_list = s->active + col0;
      a.active_count = count;
      const int k = bounds[stage + 2] - done;
      launch_fused<R, M>(a, s->S - 1, s->SP, k, sizeof(R) == 8 ? s->refine_qp : wide_step, stream);
      span_end(s, stream, &sp);
    }
  } else {
    for (int it = 0; it < (int)s->params.max_iterations; ++it) {
      span_begin(s, CPMPC_KERNEL_LINEARIZE, stream, &sp);
      launch_linearize<R, M>(a, s->SP, a.zx, a.zu, a.ist, stream);
      span_end(s, stream, &sp);
      span_begin(s, CPMPC_KERNEL_QP_LS, stream, &sp);
      if (sizeof(R) == 4 && wide_step)   // float handle with the QP's terminal part in double (CPMPC_CREATE_WIDE_QP)
        hipLaunchKernelGGL((qp_ls_kernel<R, M, true>), gridB, dim3(64), 0, stream, a);
      else
        hipLaunchKernelGGL((qp_ls_kernel<R, M, false>), gridB, dim3(64), 0, stream, a);
      span_end(s, stream, &sp);
    }
  }

  // the histogram of iterations per problem goes back to the host for the plan of a later step (default staging only)
  if (use_fused(s) && s->stage_auto && s->fb_host != nullptr && s->active != nullptr &&
      (s->params.relative_exit_tol > 0.0 || s->params.absolute_first_derivative_tol > 0.0)) {
    const int64_t groups = (B + CPMPC_PF_BLOCK - 1) / CPMPC_PF_BLOCK;
    a.fb_stride = (int)((groups + kFbReporters - 1) / kFbReporters);
    a.fb_host = s->fb_host_dev + (size_t)slot * kFbReporters * (kFbBins + 1);
    // sequence numbers 1 .. 2^30 and round again (0 = "no report yet"); compared modulo 2^30 in cpmpc_plan_stages
    s->fb_seq[slot] = s->fb_seq[slot] >= (1 << 30) ? 1 : s->fb_seq[slot] + 1;
    a.fb_seq = s->fb_seq[slot];
    s->fb_reporters[slot] = (int)((groups + a.fb_stride - 1) / a.fb_stride);
  }
  span_begin(s, CPMPC_KERNEL_FINALIZE, stream, &sp);
  hipLaunchKernelGGL((finalize_kernel<R, M>), dim3((unsigned)((B + CPMPC_PF_BLOCK - 1) / CPMPC_PF_BLOCK)), dim3(CPMPC_PF_BLOCK), 0, stream, a);
  span_end(s, stream, &sp);

  HIP_TRY(hipGetLastError());
  if (col0 + B > s->prev_B) s->prev_B = col0 + B;  // previous_solution_ = solver_->variables()  (optimization.cc:85), per problem
  return CPMPC_OK;
}

// ---- host-pointer steps: one chunk of problems through one staging slot ------------------------------------------------
// Staging layout of a chunk of Bc problems, identical on the device and in the pinned mirror:
//   [x0 | dyn? | set_point? | terminal_weights? | u | cost | eq | status | iters | solution? | predicted?]
// (inputs first: one copy in; outputs after them: one copy back, the optional tails last)
struct HostChunkLayout {
  size_t off_dyn = 0, off_sp = 0, off_tw = 0, off_u = 0, off_cost = 0, off_eq = 0, off_status = 0, off_iters = 0, off_sol = 0,
         off_pred = 0, end = 0;  // bytes
};

template <typename R, typename M>
static HostChunkLayout host_chunk_layout(const cpmpc_solver* s, int64_t Bc, bool per_dyn, bool per_sp, bool per_tw) {
  const size_t nB = (size_t)Bc, e = sizeof(R);
  HostChunkLayout L;
  L.off_dyn = (size_t)M::NX * nB * e;
  L.off_sp = L.off_dyn + (per_dyn ? (size_t)M::NP * nB * e : 0);
  L.off_tw = L.off_sp + (per_sp ? nB * e : 0);
  L.off_u = L.off_tw + (per_tw ? (size_t)M::NX * nB * e : 0);
  L.off_cost = L.off_u + (size_t)s->N * nB * e;
  L.off_eq = L.off_cost + nB * e;
  L.off_status = L.off_eq + nB * e;
  L.off_iters = L.off_status + nB * sizeof(int32_t);
  L.off_sol = (L.off_iters + nB * sizeof(int32_t) + 7) & ~(size_t)7;  // the real-typed tail starts 8-byte aligned
  L.off_pred = L.off_sol + (size_t)s->dim * nB * e;
  L.end = L.off_pred + (size_t)M::NX * (size_t)s->N * nB * e;
  return L;
}

// rows of the caller's [rows][ld] double array, columns [g0, g0 + n)  <->  [rows][n] of R in the pinned mirror, the rows
// spread over the library's worker threads (a 262 144-problem fp64 step returns 420 MB: one thread copies ~10 GB/s)
template <typename R>
struct RowCopy {
  const double* src_d;
  double* dst_d;
  R* mir;
  size_t ld, g0, n;
  bool to_mirror;
  static void run(int64_t r, void* ctx) {
    const RowCopy& c = *(const RowCopy*)ctx;
    R* m = c.mir + (size_t)r * c.n;
    if (c.to_mirror) {
      const double* src = c.src_d + (size_t)r * c.ld + c.g0;
      for (size_t i = 0; i < c.n; ++i) m[i] = (R)src[i];
    } else {
      double* dst = c.dst_d + (size_t)r * c.ld + c.g0;
      if constexpr (sizeof(R) == 8) memcpy(dst, m, c.n * 8);
      else for (size_t i = 0; i < c.n; ++i) dst[i] = (double)m[i];
    }
  }
};

template <typename R, typename M>
static int host_chunk_begin(cpmpc_solver* s, int slot_i, int64_t c0, int64_t Bc, int64_t g0, int64_t ld,
                            const cpmpc_step_host_inputs& in, const cpmpc_step_host_outputs& ho, bool direct) {
  const size_t nB = (size_t)Bc;
  const bool per_dyn = in.dyn != nullptr, per_sp = in.set_point != nullptr, per_tw = in.terminal_weights != nullptr;
  const bool want_pred = ho.predicted != nullptr, want_sol = ho.solution != nullptr;
  const HostChunkLayout L = host_chunk_layout<R, M>(s, Bc, per_dyn, per_sp, per_tw);
  int rc = ensure_slot(s, slot_i, L.end + 64);
  if (rc) return rc;
  HostSlot& sl = s->slot[slot_i];
  char* d_base = (char*)sl.dev;
  char* h_base = (char*)sl.pin;
  const hipStream_t st = sl.stream;
  // from here on work is in flight on `st` that reads the pinned mirror and writes the staging buffer: every early
  // return drains the stream first, so that the next call never reuses them under a running copy
  auto bail = [&](int code) {
    (void)hipStreamSynchronize(st);
    return code;
  };
  auto stage_in = [&](const double* src, size_t off, size_t rows) {
    RowCopy<R> c{src, nullptr, (R*)(h_base + off), (size_t)ld, (size_t)g0, nB, true};
    if (rows * nB < 65536) for (size_t r = 0; r < rows; ++r) RowCopy<R>::run((int64_t)r, &c);
    else host_parallel_for((int64_t)rows, &RowCopy<R>::run, &c);
  };
  stage_in(in.x0, 0, (size_t)M::NX);
  if (per_dyn) stage_in(in.dyn, L.off_dyn, (size_t)M::NP);
  if (per_sp) stage_in(in.set_point, L.off_sp, 1);
  if (per_tw) stage_in(in.terminal_weights, L.off_tw, (size_t)M::NX);
  hipError_t e = hipMemcpyAsync(d_base, h_base, L.off_u, hipMemcpyHostToDevice, st);
  if (e != hipSuccess) return bail(fail(CPMPC_ERR_HIP, "hipMemcpyAsync (inputs) failed: %s", hipGetErrorString(e)));

  cpmpc_step_inputs di;
  memset(&di, 0, sizeof di);
  di.x0 = d_base;
  di.dyn_shared_host = per_dyn ? nullptr : in.dyn_shared;
  di.dyn = per_dyn ? d_base + L.off_dyn : nullptr;
  di.set_point_shared = in.set_point_shared;
  di.set_point = per_sp ? d_base + L.off_sp : nullptr;
  di.terminal_weights = per_tw ? d_base + L.off_tw : nullptr;
  cpmpc_step_outputs out;
  memset(&out, 0, sizeof out);
  out.u = d_base + L.off_u;
  out.predicted = want_pred ? d_base + L.off_pred : nullptr;
  out.status = (int32_t*)(d_base + L.off_status);
  out.iterations = (int32_t*)(d_base + L.off_iters);
  out.final_cost = d_base + L.off_cost;
  out.final_eq_l1 = d_base + L.off_eq;
  out.solution = want_sol ? d_base + L.off_sol : nullptr;
  rc = step_batch_impl<R, M>(s, Bc, &di, &out, st, c0, slot_i);
  if (rc) return bail(rc);

  // Copy back.  A double handle whose real-typed output arrays the caller has pinned (hipHostMalloc / hipHostRegister /
  // cpmpc_host_register) gets them by DMA straight into those arrays, no pass of the CPU over the data; everything else
  // comes back into the mirror in one copy and is scattered by host_chunk_end.
  direct = direct && sizeof(R) == 8;
  if (direct) {
    auto d2h = [&](double* dst, size_t off, size_t rows) -> hipError_t {
      if (!dst) return hipSuccess;
      return hipMemcpy2DAsync(dst + (size_t)g0, (size_t)ld * 8, d_base + off, nB * 8, nB * 8, rows, hipMemcpyDeviceToHost, st);
    };
    e = d2h(ho.u, L.off_u, (size_t)s->N);
    if (e == hipSuccess) e = d2h(ho.solution, L.off_sol, (size_t)s->dim);
    if (e == hipSuccess) e = d2h(ho.predicted, L.off_pred, (size_t)M::NX * (size_t)s->N);
    if (e == hipSuccess)  // the small arrays still go through the mirror: cost, eq, status, iterations
      e = hipMemcpyAsync(h_base + L.off_cost, d_base + L.off_cost, L.off_iters + nB * sizeof(int32_t) - L.off_cost,
                         hipMemcpyDeviceToHost, st);
  } else {
    const size_t end = want_pred ? L.end : (want_sol ? L.off_pred : L.off_iters + nB * sizeof(int32_t));
    e = hipMemcpyAsync(h_base + L.off_u, d_base + L.off_u, end - L.off_u, hipMemcpyDeviceToHost, st);
  }
  if (e == hipSuccess) e = hipEventRecord(sl.done, st);
  if (e != hipSuccess) return bail(fail(CPMPC_ERR_HIP, "copy back of a host-pointer step failed: %s", hipGetErrorString(e)));
  sl.busy = true;
  sl.c0 = c0;
  sl.Bc = Bc;
  sl.g0 = g0;
  sl.want_pred = want_pred;
  sl.want_sol = want_sol;
  sl.direct = direct;
  // the layout depends on which per-problem inputs were given: remember it through the flags host_chunk_end recomputes from
  sl.per_dyn = per_dyn;
  sl.per_sp = per_sp;
  sl.per_tw = per_tw;
  return CPMPC_OK;
}

template <typename R, typename M>
static int host_chunk_end(cpmpc_solver* s, int slot_i, int64_t ld, const cpmpc_step_host_outputs& ho) {
  HostSlot& sl = s->slot[slot_i];
  if (!sl.busy) return CPMPC_OK;
  sl.busy = false;
  HIP_TRY(hipEventSynchronize(sl.done));
  const size_t nB = (size_t)sl.Bc;
  const HostChunkLayout L = host_chunk_layout<R, M>(s, sl.Bc, sl.per_dyn, sl.per_sp, sl.per_tw);
  char* h_base = (char*)sl.pin;
  auto fetch = [&](size_t off, double* hdst, size_t rows) {
    if (!hdst) return;
    RowCopy<R> c{nullptr, hdst, (R*)(h_base + off), (size_t)ld, (size_t)sl.g0, nB, false};
    if (rows * nB < 65536) for (size_t r = 0; r < rows; ++r) RowCopy<R>::run((int64_t)r, &c);
    else host_parallel_for((int64_t)rows, &RowCopy<R>::run, &c);
  };
  if (!sl.direct) {
    fetch(L.off_u, ho.u, (size_t)s->N);
    if (sl.want_sol) fetch(L.off_sol, ho.solution, (size_t)s->dim);
    if (sl.want_pred) fetch(L.off_pred, ho.predicted, (size_t)M::NX * (size_t)s->N);
  }
  fetch(L.off_cost, ho.final_cost, 1);
  fetch(L.off_eq, ho.final_eq_l1, 1);
  if (ho.status) memcpy(ho.status + sl.g0, h_base + L.off_status, nB * sizeof(int32_t));
  if (ho.iterations) memcpy(ho.iterations + sl.g0, h_base + L.off_iters, nB * sizeof(int32_t));
  return CPMPC_OK;
}

template <typename R>
static ExtForce<R> ext_from_host(const double* fext_host) {
  ExtForce<R> fe{R(0), R(0), R(0)};
  if (fext_host) {
    fe.fbx = (R)fext_host[0];
    fe.fmx = (R)fext_host[2];
    fe.fmy = (R)fext_host[3];
  }
  return fe;
}

template <typename R, typename M>
static void linearize_batch_impl(cpmpc_solver* s, int64_t B, const double* dyn_shared_host, const void* z, void* c,
                                 void* Phi, void* Gamma, hipStream_t st) {
  SolverArgs<R, M> a;
  fill_args<R, M>(s, B, a);
  a.consts = M::template make<double>(dyn_shared_host);
  hipLaunchKernelGGL((pack_z_kernel<R, M::NX>), grid_for(B), dim3(64), 0, st, B, s->cap, s->S, s->N, (const R*)z,
                     a.dzx, a.dzu);
  launch_linearize<R, M>(a, s->SP, a.dzx, a.dzu, nullptr, st);
  hipLaunchKernelGGL((unpack_lin_kernel<R, M>), grid_for(B), dim3(64), 0, st, a, (R*)c, (R*)Phi, (R*)Gamma);
}


// ---- the remaining entry points of the table ----------------------------------------------------------------------
template <typename R, typename M>
static void pack_z_impl(cpmpc_solver* s, int64_t B, const void* z, hipStream_t stream) {
  hipLaunchKernelGGL((pack_z_kernel<R, M::NX>), grid_for(B), dim3(64), 0, stream, B, s->cap, s->S, s->N, (const R*)z,
                     (XV<R, M::NX>*)s->zx, (R*)s->zu);
}
template <typename R, typename M>
static void unpack_z_impl(cpmpc_solver* s, int64_t B, void* z_out, hipStream_t stream) {
  hipLaunchKernelGGL((unpack_z_kernel<R, M::NX>), grid_for(B), dim3(64), 0, stream, B, s->cap, s->S, s->N,
                     (const XV<R, M::NX>*)s->zx, (const R*)s->zu, (R*)z_out);
}
template <typename R, typename M>
static void dynamics_impl(int64_t B, const double* dyn_shared_host, const double* fext_host, const void* x, const void* u,
                          void* f, void* Jx, void* Ju, hipStream_t stream) {
  hipLaunchKernelGGL((dynamics_kernel<R, M>), grid_for(B), dim3(64), 0, stream, B,
                     M::template make<double>(dyn_shared_host), ext_from_host<R>(fext_host), (const R*)x, (const R*)u,
                     (R*)f, (R*)Jx, (R*)Ju);
}
template <typename R, typename M>
static void rk4_impl(int64_t B, const double* dyn_shared_host, const double* fext_host, double h, const void* x,
                     const void* u, void* x_new, void* A, void* Bm, hipStream_t stream) {
  hipLaunchKernelGGL((rk4_kernel<R, M>), grid_for(B), dim3(64), 0, stream, B, M::template make<double>(dyn_shared_host),
                     ext_from_host<R>(fext_host), (R)h, (const R*)x, (const R*)u, (R*)x_new, (R*)A, (R*)Bm);
}
template <typename R, typename M>
static void sim_impl(int64_t B, const double* dyn_shared_host, const double* fext_host, const void* fext, int n_sub,
                     double h_last, const void* u, void* state, hipStream_t stream) {
  hipLaunchKernelGGL((sim_kernel<R, M>), grid_for(B), dim3(64), 0, stream, B, M::template make<double>(dyn_shared_host),
                     ext_from_host<R>(fext_host), (const R*)fext, n_sub, (R)h_last, (const R*)u, (R*)state);
}

// debug builds: this unit's copies of the counters (every translation unit has its own __device__ variables)
static int debug_read_impl(int which, unsigned long long* out) {
#ifdef CPMPC_FUSED_TIMING
  if (which == 0) {
    unsigned long long v[8], zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyFromSymbol(v, HIP_SYMBOL(cpmpc::g_fused_phase_cycles), sizeof v) != hipSuccess) return -1;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(cpmpc::g_fused_phase_cycles), zero, sizeof zero);
    for (int i = 0; i < 8; ++i) out[i] += v[i];
    return 0;
  }
#endif
#ifdef CPMPC_FUSED_CLOCK
  if (which == 1) {
    unsigned long long v[4], zero[4] = {0, 0, 0, 0};
    if (hipMemcpyFromSymbol(v, HIP_SYMBOL(cpmpc::g_fused_clock), sizeof v) != hipSuccess) return -1;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(cpmpc::g_fused_clock), zero, sizeof zero);
    out[0] += v[0];
    out[1] += v[1];
    out[2] += v[2];
    if (v[3] > out[3]) out[3] = v[3];
    return 0;
  }
#endif
  (void)which;
  (void)out;
  return -1;
}

#define CPMPC_DEFINE_ENGINE(NAME, R, M)                                                                              \
  const Engine* NAME() {                                                                                             \
    static const Engine e = {&step_batch_impl<R, M>, &host_chunk_begin<R, M>, &host_chunk_end<R, M>, &pack_z_impl<R, M>, \
                             &unpack_z_impl<R, M>,   &dynamics_impl<R, M>,   &rk4_impl<R, M>,      &sim_impl<R, M>,    \
                             &linearize_batch_impl<R, M>, &debug_read_impl};                                         \
    return &e;                                                                                                       \
  }
