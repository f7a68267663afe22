// cpmpc_api.hip -- C-ABI (include/cpmpc.h) over the gfx950 kernels.
//
// Host side of the batched cart-pole MPC hot path.  There is no CPU compute path in this library:
// every entry point that computes launches HIP kernels and fails with CPMPC_ERR_NO_DEVICE when no
// gfx950 device is usable.  This unit holds no device code: the kernels of each (dtype, model) pair live in their own
// translation unit (engine_<dtype>_<model>.hip) and are reached through its `Engine` table (engine.hpp).
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <deque>
#include <mutex>
#include <thread>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "engine.hpp"

// ------------------------------------------------------------------------------------------------
// error text
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

int cpmpc_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}

extern "C" const char* cpmpc_last_error(void) { return g_err; }

// ------------------------------------------------------------------------------------------------
// defaults
// ------------------------------------------------------------------------------------------------
extern "C" void cpmpc_default_params(cpmpc_params* p) {  // optimization/optimization.hpp:12-48
  p->control_dt = 0.01;
  p->window_length = 40;
  p->state_spacing = 10;
  p->max_iterations = 8;
  p->relative_exit_tol = 1.0e-5;
  p->absolute_first_derivative_tol = 1.0e-6;
  p->equality_penalty_initial = 1.0;
  p->u_guess_sinusoid_amplitude = 10.0;
  p->u_cost_weight = 0.1;
  p->u_derivative_cost_weight = 0.1;
  p->b_x_final_cost_weight = 150.0;
  p->th_final_cost_weight = -1.0;
  p->b_x_dot_final_cost_weight = -1.0;
  p->th_dot_final_cost_weight = -1.0;
}

extern "C" void cpmpc_default_solver_opts(cpmpc_solver_opts* o) {
  o->max_line_search_iterations = 5;  // optimization.cc:76
  o->armijo_c1 = 1.0e-4;
  o->ls_shrink_max = 0.5;
  o->ls_shrink_min = 0.1;
  o->ls_alpha_growth = 2.0;
  o->penalty_rho = 0.1;
  o->lambda_initial = 0.0;
  o->lambda_failure_init = 1.0e-2;
  o->lambda_scale_up = 10.0;
  o->lambda_scale_down = 0.1;
  o->lambda_min = 1.0e-8;
  o->lambda_max = 1.0e6;
  o->b_x_limit = 5.0;  // optimization.cc:320
  o->u_limit = 300.0;  // optimization.cc:327
  o->ls_alpha_growth_backtracked = 2.0;
  o->full_step_below = 1.0e-4;
  o->exit_defect_floor = 2.0;
}

// ------------------------------------------------------------------------------------------------
// device discovery
// ------------------------------------------------------------------------------------------------
static bool device_is_gfx950(int dev) {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return false;
  return strncmp(prop.gcnArchName, "gfx950", 6) == 0;
}

extern "C" int cpmpc_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  int ok = 0;
  for (int i = 0; i < n; ++i)
    if (device_is_gfx950(i)) ++ok;
  return ok;
}

static int check_device(int dev) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    return fail(CPMPC_ERR_NO_DEVICE,
                "no HIP device visible; this library has no CPU fallback (libcpmpc needs a gfx950 GPU)");
  if (dev < 0 || dev >= n) return fail(CPMPC_ERR_NO_DEVICE, "device %d out of range (%d visible)", dev, n);
  if (!device_is_gfx950(dev))
    return fail(CPMPC_ERR_NO_DEVICE, "device %d is not gfx950; kernels are built for gfx950 only", dev);
  return CPMPC_OK;
}

struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) == hipSuccess && prev != dev) {
      if (hipSetDevice(dev) == hipSuccess) switched = true;
    }
  }
  ~DeviceGuard() {
    if (switched) (void)hipSetDevice(prev);
  }
};

// ------------------------------------------------------------------------------------------------
// models
// ------------------------------------------------------------------------------------------------
static int model_nx(int model) { return model == CPMPC_MODEL_DOUBLE ? 6 : 4; }
static int model_np(int model) { return model == CPMPC_MODEL_DOUBLE ? 6 : 9; }
extern "C" int cpmpc_model_state_dim(int model) {
  return (model == CPMPC_MODEL_SINGLE || model == CPMPC_MODEL_DOUBLE) ? model_nx(model) : -1;
}
extern "C" int cpmpc_model_num_params(int model) {
  return (model == CPMPC_MODEL_SINGLE || model == CPMPC_MODEL_DOUBLE) ? model_np(model) : -1;
}

// ------------------------------------------------------------------------------------------------
// the handle
// ------------------------------------------------------------------------------------------------
// 2: a register-resident linearisation is compiled for this spacing; 1: served by the generic kernel
// (run-time spacing, O(spacing^2) workspace traffic per interval); 0: not a spacing
extern "C" int cpmpc_supported_state_spacing(int spacing) {
  if (spacing < 1) return 0;
  return (spacing == 1 || spacing == 2 || spacing == 4 || spacing == 5 || spacing == 8 || spacing == 10 ||
          spacing == 20) ? 2 : 1;
}

static int validate_params(const cpmpc_params* p) {
  // the constructor's preconditions, optimization.cc:13-22
  if (!(p->control_dt > 0)) return fail(CPMPC_ERR_INVALID_ARG, "control_dt must be > 0 (optimization.cc:14)");
  if (!(p->window_length >= 1)) return fail(CPMPC_ERR_INVALID_ARG, "window_length must be >= 1 (optimization.cc:15)");
  if (p->state_spacing == 0 || p->window_length % p->state_spacing != 0)
    return fail(CPMPC_ERR_INVALID_ARG,
                "state_spacing (%llu) must divide into window_length (%llu) cleanly (optimization.cc:16-18)",
                (unsigned long long)p->state_spacing, (unsigned long long)p->window_length);
  if (!(p->max_iterations >= 1)) return fail(CPMPC_ERR_INVALID_ARG, "max_iterations must be >= 1 (optimization.cc:19)");
  if (!(p->u_cost_weight >= 0.0)) return fail(CPMPC_ERR_INVALID_ARG, "u_cost_weight must be >= 0 (optimization.cc:20)");
  if (!(p->u_derivative_cost_weight >= 0.0))
    return fail(CPMPC_ERR_INVALID_ARG, "u_derivative_cost_weight must be >= 0 (optimization.cc:21)");
  if (p->window_length > 4096) return fail(CPMPC_ERR_UNSUPPORTED, "window_length > 4096 is not supported");
  return CPMPC_OK;
}

// Horizon (window_length * control_dt, seconds) up to which the condensed QP is held to 1e-5 of a full-space KKT solve on
// every problem (include/cpmpc.h: cpmpc_max_parity_horizon)
static const double kMaxParityHorizon = 1.0;
// u_cost_weight below which the fp64 fused kernels refine the whole QP solution by default (half the reference's 0.1)
static const double kRefineBelowUCostWeight = 0.05;
extern "C" double cpmpc_max_parity_horizon(void) { return kMaxParityHorizon; }

static int create_impl(const cpmpc_params* params, const cpmpc_solver_opts* opts, size_t opts_size, int dtype,
                       int64_t max_batch, int device, int model, uint32_t flags, cpmpc_solver** out) {
  if (!params || !out) return fail(CPMPC_ERR_INVALID_ARG, "null argument");
  *out = nullptr;
  if (dtype != CPMPC_F32 && dtype != CPMPC_F64) return fail(CPMPC_ERR_INVALID_ARG, "dtype must be CPMPC_F32 or CPMPC_F64");
  if (model != CPMPC_MODEL_SINGLE && model != CPMPC_MODEL_DOUBLE) return fail(CPMPC_ERR_INVALID_ARG, "unknown model");
  if (max_batch < 1 || max_batch > (1ll << 30)) return fail(CPMPC_ERR_INVALID_ARG, "max_batch must be in [1, 2^30]");
  int rc = validate_params(params);
  if (rc) return rc;
  if ((flags & ~(uint32_t)(CPMPC_CREATE_ALLOW_LONG_HORIZON | CPMPC_CREATE_REFINE_QP | CPMPC_CREATE_NO_REFINE_QP |
                           CPMPC_CREATE_STRICT_HORIZON | CPMPC_CREATE_WIDE_QP | CPMPC_CREATE_NO_WIDE_QP)) != 0)
    return fail(CPMPC_ERR_INVALID_ARG, "unknown creation flags 0x%x", flags);
  if ((flags & CPMPC_CREATE_REFINE_QP) && (flags & CPMPC_CREATE_NO_REFINE_QP))
    return fail(CPMPC_ERR_INVALID_ARG, "CPMPC_CREATE_REFINE_QP and CPMPC_CREATE_NO_REFINE_QP exclude each other");
  if ((flags & CPMPC_CREATE_WIDE_QP) && (flags & CPMPC_CREATE_NO_WIDE_QP))
    return fail(CPMPC_ERR_INVALID_ARG, "CPMPC_CREATE_WIDE_QP and CPMPC_CREATE_NO_WIDE_QP exclude each other");
  // the struct in some release: the leading int (padded to 8) and a whole number of doubles, from the shortest layout that is
  // a PREFIX of today's struct -- 13 doubles, through u_limit: 112 bytes -- to this one's.  (The 104-byte struct of the very
  // first commit had no ls_alpha_growth, which was inserted mid-struct before any release: read as today's prefix it would
  // shift every later field by one slot, so it is refused, like a size that splits a field: ADVICE r4, r5.)
  if (opts != nullptr && (opts_size < 112 || opts_size > sizeof(cpmpc_solver_opts) || opts_size % 8 != 0))
    return fail(CPMPC_ERR_INVALID_ARG, "opts_size %zu is not the size of any cpmpc_solver_opts this library knows (8 + 8 k "
                "bytes, 112 .. %zu)", opts_size, sizeof(cpmpc_solver_opts));
  {
    // Every horizon the reference's constructor accepts is accepted (optimization.cc:13-22: this is a drop-in); beyond
    // cpmpc_max_parity_horizon() the first handle of the process says so once, and CPMPC_CREATE_STRICT_HORIZON refuses.
    const double horizon = (double)params->window_length * params->control_dt;
    if (horizon > kMaxParityHorizon * (1.0 + 1e-9)) {
      if (flags & CPMPC_CREATE_STRICT_HORIZON)
        return fail(CPMPC_ERR_UNSUPPORTED,
                    "horizon window_length * control_dt = %.3f s exceeds %.1f s, the longest the condensed QP is held to 1e-5 of a "
                    "full-space solve on every problem (state elimination through an unstable plant loses ~3 digits per QP at "
                    "1.6 s; 2 of 16384 cold starts are beyond 1e-5 at 1.2 s), and CPMPC_CREATE_STRICT_HORIZON was given",
                    horizon, kMaxParityHorizon);
      static std::atomic<bool> warned{false};
      if (!(flags & CPMPC_CREATE_ALLOW_LONG_HORIZON) && !warned.exchange(true))
        fprintf(stderr,
                "cpmpc: horizon window_length * control_dt = %.3f s is beyond %.1f s (cpmpc_max_parity_horizon): solved as "
                "asked, but cold starts far from the optimum may differ from a full-space solve by more than 1e-5 on a "
                "few problems in 10^4 (include/cpmpc.h, CPMPC_CREATE_STRICT_HORIZON).  Said once per process.\n",
                horizon, kMaxParityHorizon);
    }
  }
  // defaults first, then as many leading bytes as the caller's struct has: a caller compiled against an earlier header
  // (a shorter struct: fields are only ever appended) keeps this library's defaults for the options it does not know
  cpmpc_solver_opts merged;
  cpmpc_default_solver_opts(&merged);
  if (opts) memcpy(&merged, opts, opts_size);
  if (!(merged.full_step_below >= 0.0) || !(merged.exit_defect_floor >= 0.0) || !std::isfinite(merged.full_step_below) ||
      !std::isfinite(merged.exit_defect_floor))   // the two thresholds that switch a rule off at 0 (NaN fails both tests)
    return fail(CPMPC_ERR_INVALID_ARG, "full_step_below and exit_defect_floor must be finite and >= 0 (0 disables the rule)");
  rc = check_device(device);
  if (rc) return rc;
  DeviceGuard guard(device);

  cpmpc_solver* s = new (std::nothrow) cpmpc_solver();
  if (!s) return fail(CPMPC_ERR_ALLOC, "out of host memory");
  s->params = *params;
  s->opts = merged;
  s->dtype = dtype;
  s->beyond_parity = (double)params->window_length * params->control_dt > kMaxParityHorizon * (1.0 + 1e-9);
  // default: refine where the control cost is weak (measured: include/cpmpc.h, CPMPC_CREATE_REFINE_QP) and, since round 6,
  // beyond the parity horizon: there one pass of refinement with residuals from the original data takes the lanes on
  // which the kernels (not the CPU check) moved from 34 of 8 192 to 3 at N = 160 (profiles/r06_long_horizon_probe.json);
  // further passes change nothing (the refined solve and a dense pivoted one are then both at the problem's conditioning)
  s->refine_qp = (flags & CPMPC_CREATE_REFINE_QP) != 0 ||
                 (!(flags & CPMPC_CREATE_NO_REFINE_QP) && (params->u_cost_weight < kRefineBelowUCostWeight || s->beyond_parity));
  // split pipeline: one pass; two beyond the parity horizon (tools/long_horizon_refine_study.py, first QP of 300 cold starts at
  // 1.6 s against a long-double solve: worst lane 3.5e-5 condensed, 1.6e-11 after one pass, 1.2e-13 after two -- the dense
  // pivoted LU of the CPU check: 1.1e-12; further passes change nothing)
  s->refine_passes = s->beyond_parity ? 2 : 1;
  if (const char* e = getenv("CPMPC_QP_REFINE_PASSES")) {   // diagnostic (tools/long_horizon_gpu_probe.py): passes of the split pipeline
    const int n = atoi(e);
    if (n >= 1 && n <= 16) s->refine_passes = n;
  }
  // default: on for the 6-state model (measured: include/cpmpc.h, CPMPC_CREATE_WIDE_QP), off for the 4-state one
  s->wide_qp = dtype == CPMPC_F32 && ((flags & CPMPC_CREATE_WIDE_QP) != 0 ||
                                      (!(flags & CPMPC_CREATE_NO_WIDE_QP) && model == CPMPC_MODEL_DOUBLE));
  s->model = model;
  s->device = device;
  s->esize = dtype == CPMPC_F32 ? 4 : 8;
  s->NX = model_nx(model);
  s->NP = model_np(model);
  s->N = (int)params->window_length;
  s->SP = (int)params->state_spacing;
  s->S = s->N / s->SP + 1;  // optimization.hpp:52
  s->dim = s->NX * s->S + s->N;  // optimization.cc:204-205
  s->cap = (max_batch + 63) / 64 * 64;

  // scalars of the dtype per problem; an NX-vector field occupies XW = 4 or 8 scalars
  const size_t XW = s->NX > 4 ? 8 : 4;
  const size_t n_xv = (size_t)2 * s->S + (size_t)s->NX * (s->S - 1) + s->N + (s->S - 1) + s->N;  // zx dzx Phi Gam cs Wk
  const size_t fields_real = XW * n_xv + 4 * (size_t)s->N + 2 * (size_t)s->N + SC_COUNT;
  const size_t bytes_real = fields_real * (size_t)s->cap * s->esize;
  const size_t bytes_int = (size_t)IS_COUNT * (size_t)s->cap * sizeof(int32_t);
  s->ws_bytes = bytes_real + bytes_int;
  hipError_t e = hipMalloc(&s->ws, s->ws_bytes);
  if (e != hipSuccess) {
    delete s;
    return fail(CPMPC_ERR_ALLOC, "hipMalloc of %zu bytes of workspace failed: %s", bytes_real + bytes_int,
                hipGetErrorString(e));
  }
  (void)hipMemset(s->ws, 0, s->ws_bytes);
  char* pch = (char*)s->ws;
  auto carve = [&](size_t nfields) {
    char* r = pch;
    pch += nfields * (size_t)s->cap * s->esize;
    return r;
  };
  // vector fields first so that every one of them is 16/32-byte aligned (cap is a multiple of 64)
  s->zx = carve(XW * s->S);
  s->dzx = carve(XW * s->S);
  s->Phi = carve(XW * s->NX * (s->S - 1));
  s->Gam = carve(XW * s->N);
  s->cs = carve(XW * (s->S - 1));
  s->Wk = carve(XW * s->N);
  s->Tk = carve(4 * s->N);
  s->zu = carve(s->N);
  s->dzu = carve(s->N);
  s->sc = carve(SC_COUNT);
  s->ist = (int32_t*)pch;

  // sinusoid cold-start table, evaluated on the host in double exactly as optimization.cc:63-67
  std::vector<double> tab(s->N);
  for (int k = 0; k < s->N; ++k)
    tab[k] = params->u_guess_sinusoid_amplitude *
             std::sin(static_cast<double>(k) / static_cast<double>(s->N) * 2 * M_PI);
  e = hipMalloc(&s->sin_table, (size_t)s->N * s->esize);
  if (e != hipSuccess) {
    (void)hipFree(s->ws);
    delete s;
    return fail(CPMPC_ERR_ALLOC, "hipMalloc failed: %s", hipGetErrorString(e));
  }
  if (dtype == CPMPC_F32) {
    std::vector<float> tf(tab.begin(), tab.end());
    e = hipMemcpy(s->sin_table, tf.data(), tf.size() * 4, hipMemcpyHostToDevice);
  } else {
    e = hipMemcpy(s->sin_table, tab.data(), tab.size() * 8, hipMemcpyHostToDevice);
  }
  if (e != hipSuccess) {
    (void)hipFree(s->ws);
    (void)hipFree(s->sin_table);
    delete s;
    return fail(CPMPC_ERR_HIP, "hipMemcpy failed: %s", hipGetErrorString(e));
  }
  // index list of the staged fused pipeline (4 bytes per problem + three counters per host slot): allocated here, never in a step
  if (hipMalloc((void**)&s->active, ((size_t)s->cap + 3 * kHostSlots) * sizeof(int32_t)) != hipSuccess) {
    (void)hipGetLastError();
    s->active = nullptr;  // staging stays off for this handle (same results, single launch)
  }
  // histogram of iterations per problem, device -> host through mapped memory (the plan of the stages; optional as well)
  const size_t fb_bytes = (size_t)kHostSlots * kFbReporters * (kFbBins + 1) * sizeof(int32_t);
  if (s->active != nullptr &&
      hipHostMalloc((void**)&s->fb_host, fb_bytes, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess &&
      hipHostGetDevicePointer((void**)&s->fb_host_dev, s->fb_host, 0) == hipSuccess) {
    memset(s->fb_host, 0, fb_bytes);
  } else {
    (void)hipGetLastError();
    if (s->fb_host) (void)hipHostFree(s->fb_host);
    s->fb_host = s->fb_host_dev = nullptr;  // the fixed plan (2 iterations, then 1 at a time) for every step
  }
  *out = s;
  return CPMPC_OK;
}

extern "C" int cpmpc_create_model(const cpmpc_params* params, const cpmpc_solver_opts* opts, int dtype,
                                  int64_t max_batch, int device, int model, cpmpc_solver** out) {
  return create_impl(params, opts, CPMPC_SOLVER_OPTS_SIZE_POSITIONAL, dtype, max_batch, device, model, 0, out);
}

extern "C" int cpmpc_create(const cpmpc_params* params, const cpmpc_solver_opts* opts, int dtype,
                            int64_t max_batch, int device, cpmpc_solver** out) {
  return create_impl(params, opts, CPMPC_SOLVER_OPTS_SIZE_POSITIONAL, dtype, max_batch, device, CPMPC_MODEL_SINGLE, 0, out);
}

extern "C" int cpmpc_create_ex(const cpmpc_create_info* info, cpmpc_solver** out) {
  if (!info || !out) return fail(CPMPC_ERR_INVALID_ARG, "null argument");
  *out = nullptr;
  if (info->struct_size != sizeof(cpmpc_create_info))
    return fail(CPMPC_ERR_INVALID_ARG, "cpmpc_create_info.struct_size is %u, this library's is %zu (set it to sizeof(cpmpc_create_info))",
                info->struct_size, sizeof(cpmpc_create_info));
  const size_t osz = info->opts_size ? (size_t)info->opts_size : sizeof(cpmpc_solver_opts);
  return create_impl(info->params, info->opts, osz, info->dtype, info->max_batch, info->device, info->model, info->flags, out);
}

extern "C" void cpmpc_destroy(cpmpc_solver* s) {
  if (!s) return;
  DeviceGuard guard(s->device);
  for (auto& sp : s->spans) {
    (void)hipEventDestroy(sp.start);
    (void)hipEventDestroy(sp.stop);
  }
  for (auto& sp : s->free_spans) {
    (void)hipEventDestroy(sp.start);
    (void)hipEventDestroy(sp.stop);
  }
  if (s->ws) (void)hipFree(s->ws);
  if (s->sin_table) (void)hipFree(s->sin_table);
  for (HostSlot& sl : s->slot) {
    if (sl.stream) (void)hipStreamSynchronize(sl.stream);
    if (sl.dev) (void)hipFree(sl.dev);
    if (sl.pin) (void)hipHostFree(sl.pin);
    if (sl.done) (void)hipEventDestroy(sl.done);
    if (sl.stream) (void)hipStreamDestroy(sl.stream);
  }
  if (s->ev_last) (void)hipEventDestroy(s->ev_last);
  if (s->active) (void)hipFree(s->active);
  if (s->fb_host) (void)hipHostFree(s->fb_host);
  delete s;
}

extern "C" int cpmpc_dim(const cpmpc_solver* s) { return s ? s->dim : -1; }
extern "C" int cpmpc_num_states(const cpmpc_solver* s) { return s ? s->S : -1; }
extern "C" int cpmpc_dtype(const cpmpc_solver* s) { return s ? s->dtype : -1; }
extern "C" int cpmpc_model(const cpmpc_solver* s) { return s ? s->model : -1; }
extern "C" int cpmpc_refines_qp(const cpmpc_solver* s) { return s ? (s->refine_qp && s->dtype == CPMPC_F64 ? 1 : 0) : -1; }
// what a handle actually uses: defaults merged with as much of the caller's struct as its constructor read (the
// positional constructors read 128 bytes, so a later field -- exit_defect_floor -- set through them is NOT taken)
extern "C" int cpmpc_get_solver_opts(const cpmpc_solver* s, cpmpc_solver_opts* out, size_t out_size) {
  if (!s || !out) return fail(CPMPC_ERR_INVALID_ARG, "null argument");
  if (out_size < 112 || out_size > sizeof(cpmpc_solver_opts) || out_size % 8 != 0)
    return fail(CPMPC_ERR_INVALID_ARG, "out_size %zu is not the size of any cpmpc_solver_opts this library knows", out_size);
  memcpy(out, &s->opts, out_size);
  return CPMPC_OK;
}
extern "C" int cpmpc_horizon_beyond_parity(const cpmpc_solver* s) { return s ? (s->beyond_parity ? 1 : 0) : -1; }
extern "C" int cpmpc_wide_qp(const cpmpc_solver* s) {
  return s ? (s->wide_qp ? 1 : 0) : -1;  // either pipeline since round 6 (qp_ls_kernel<R, M, true>)
}
extern "C" int cpmpc_has_previous_solution(const cpmpc_solver* s) { return (s && s->prev_B > 0) ? 1 : 0; }
extern "C" int64_t cpmpc_previous_solution_batch(const cpmpc_solver* s) { return s ? s->prev_B : 0; }

extern "C" int cpmpc_reset(cpmpc_solver* s) {  // Optimization::Reset, optimization.hpp:83
  if (!s) return fail(CPMPC_ERR_INVALID_ARG, "null solver");
  s->prev_B = 0;
  return CPMPC_OK;
}

// ------------------------------------------------------------------------------------------------
// profiling spans
// ------------------------------------------------------------------------------------------------
static const char* kKernelNames[CPMPC_KERNEL_COUNT] = {"prepare_kernel", "linearize_kernel", "qp_ls_kernel",
                                                       "finalize_kernel", "fused_sqp_kernel"};
extern "C" const char* cpmpc_kernel_name(int kernel) {
  return (kernel >= 0 && kernel < CPMPC_KERNEL_COUNT) ? kKernelNames[kernel] : "?";
}

void span_begin(cpmpc_solver* s, int kernel, hipStream_t stream, ProfSpan* cur) {
  if (!s->profiling) return;
  if (!s->free_spans.empty()) {
    *cur = s->free_spans.back();
    s->free_spans.pop_back();
  } else {
    (void)hipEventCreate(&cur->start);
    (void)hipEventCreate(&cur->stop);
  }
  cur->kernel = kernel;
  (void)hipEventRecord(cur->start, stream);
}
void span_end(cpmpc_solver* s, hipStream_t stream, ProfSpan* cur) {
  if (!s->profiling) return;
  (void)hipEventRecord(cur->stop, stream);
  s->spans.push_back(*cur);
}

static void collect_spans(cpmpc_solver* s) {
  for (auto& sp : s->spans) {
    float ms = 0.f;
    if (hipEventSynchronize(sp.stop) == hipSuccess && hipEventElapsedTime(&ms, sp.start, sp.stop) == hipSuccess) {
      s->prof_ms[sp.kernel] += ms;
      s->prof_n[sp.kernel] += 1;
    }
    s->free_spans.push_back(sp);
  }
  s->spans.clear();
}

extern "C" int cpmpc_profile_enable(cpmpc_solver* s, int on) {
  if (!s) return fail(CPMPC_ERR_INVALID_ARG, "null solver");
  s->profiling = on ? 1 : 0;
  return CPMPC_OK;
}
extern "C" int cpmpc_profile_reset(cpmpc_solver* s) {
  if (!s) return fail(CPMPC_ERR_INVALID_ARG, "null solver");
  DeviceGuard guard(s->device);
  collect_spans(s);
  for (int i = 0; i < CPMPC_KERNEL_COUNT; ++i) {
    s->prof_ms[i] = 0;
    s->prof_n[i] = 0;
  }
  return CPMPC_OK;
}
extern "C" int cpmpc_profile_read(cpmpc_solver* s, int kernel, double* total_ms, int64_t* launches) {
  if (!s || kernel < 0 || kernel >= CPMPC_KERNEL_COUNT) return fail(CPMPC_ERR_INVALID_ARG, "bad argument");
  DeviceGuard guard(s->device);
  collect_spans(s);
  if (total_ms) *total_ms = s->prof_ms[kernel];
  if (launches) *launches = s->prof_n[kernel];
  return CPMPC_OK;
}

// Host-pointer entry points run on the handle's own stream.  Once one has been used, every device-pointer call on a
// caller's stream leaves an event behind so that the next host-pointer call is ordered after it.
static void track_caller_stream(cpmpc_solver* s, hipStream_t stream) {
  bool own = false;
  for (const HostSlot& sl : s->slot) own = own || (sl.stream != nullptr && stream == sl.stream);
  if (s->ev_last != nullptr && !own) {
    (void)hipEventRecord(s->ev_last, stream);
    s->ev_pending = true;
  }
}

// ------------------------------------------------------------------------------------------------
// launch helpers
// ------------------------------------------------------------------------------------------------
extern "C" int cpmpc_set_pipeline(cpmpc_solver* s, int mode) {
  if (!s) return fail(CPMPC_ERR_INVALID_ARG, "null solver");
  if (mode != CPMPC_PIPELINE_AUTO && mode != CPMPC_PIPELINE_SPLIT && mode != CPMPC_PIPELINE_FUSED)
    return fail(CPMPC_ERR_INVALID_ARG, "unknown pipeline mode");
  if (mode == CPMPC_PIPELINE_FUSED && !fused_built(s))
    return fail(CPMPC_ERR_UNSUPPORTED, "the fused pipeline is not built for model %d with S-1 = %d, state_spacing = %d",
                s->model, s->S - 1, s->SP);
  s->pipeline = mode;
  return CPMPC_OK;
}
extern "C" int cpmpc_set_compaction(cpmpc_solver* s, int first_iterations, int next_iterations) {
  if (!s) return fail(CPMPC_ERR_INVALID_ARG, "null solver");
  if (first_iterations < 0 || next_iterations < 0) return fail(CPMPC_ERR_INVALID_ARG, "iteration counts must be >= 0");
  s->stage_first = first_iterations;
  s->stage_next = next_iterations;
  s->stage_auto = false;  // an explicit setting applies to every batch size
  return CPMPC_OK;
}
// ---- the stages of the fused pipeline for one step ---------------------------------------------------------------------
// The kernel runs bounds[i+1] - bounds[i] iterations per launch; between launches the problems still iterating are
// compacted into dense waves (engine_impl.hpp: step_batch_impl).  Where to cut is a question of speed only.
//  * exits disabled, or a batch of at most one round of resident waves: one launch;
//  * an explicit cpmpc_set_compaction(first, next): first, then next at a time;
//  * default: from the histogram of iterations per problem of the most recent step whose finalize_kernel has finished
//    (host-mapped memory, read without synchronisation; a step queued but not finished simply does not count yet), by
//    dynamic programming over the cuts with this cost model, in units of one wave's time for one iteration:
//      - a stage from iteration a to b runs surv(a) / ppw waves (surv(j) = problems that need more than j iterations);
//        a wave lasts until the slowest of its ppw problems stops: sum over j of 1 - (1 - surv(a+j) / surv(a))^ppw;
//      - the stage takes max(wave-iterations / resident waves, its longest wave) + one launch and compaction (7 us);
//      - a stage with at most one round of resident waves left runs to the end (the kernel's own rule);
//    a plan that comes out as ONE launch although the histogram says nearly everybody (99.9 %) stops early gets one cut
//    where they do: in a settled closed loop (everybody stops after the first iteration) that is a second, empty launch
//    per tick (7 us), and it is what keeps a tick in which some controllers are disturbed from being bound by waves
//    that one disturbed problem in sixteen keeps alive.  Before the first histogram, and when the only histogram there is
//    is old and the counts are moving (see below): 2 iterations, then 1 at a time.
// the plan from a histogram (hist[b] = problems that ran b iterations, the last bin collecting every larger count):
// bounds[0] = 0 < ... < bounds[n] = T, returns n.  resident = waves of the fused kernel the machine holds (1 / 2 per SIMD)
static int plan_from_histogram(const double* hist, double n_hist, int64_t B, int T, int ppw, double resident, int window,
                               int* bounds) {
  // surv[j]: expected problems of THIS batch needing more than j iterations, j = 0 .. T
  double surv[kMaxStages + 1];
  const double scale = (double)B / n_hist;
  for (int j = 0; j <= T; ++j) {
    double above = 0.0;
    for (int b = j + 1; b < kFbBins; ++b) above += hist[b];
    surv[j] = j < T ? above * scale : 0.0;
  }
  surv[0] = (double)B;
  const double launch = 0.14 * 40.0 / (double)window;        // 7 us in units of one wave-iteration (50 us at N = 40)
  const double run_out = 2048.0 * (double)ppw;                // the kernel's run_out_below
  double best[kMaxStages + 2];
  int next_cut[kMaxStages + 2];
  best[T] = 0.0;
  for (int a = T - 1; a >= 0; --a) {
    best[a] = 1e300;
    next_cut[a] = T;
    // a stage that starts at a, by the iteration b it ends at (ascending, the sums carried along)
    double wave_iters = 0.0, longest = 0.0;
    for (int b = a + 1; b <= T; ++b) {
      const int j = b - 1 - a;
      const double q = surv[a] > 0.0 ? surv[a + j] / surv[a] : 0.0;
      wave_iters += 1.0 - std::pow(1.0 - (q > 1.0 ? 1.0 : q), (double)ppw);
      if (surv[a + j] >= 1.0) longest = (double)(j + 1);
      if (a > 0 && surv[a] <= run_out && b != T) continue;   // such a stage runs to the end by itself
      const double thr = wave_iters * surv[a] / (double)ppw / resident;
      const double cost = (thr > longest ? thr : longest) + launch + (b < T ? best[b] : 0.0);
      if (cost < best[a] - 1e-12 || (b == T && cost <= best[a] + 1e-12)) {   // ties: the fewer launches
        best[a] = cost;
        next_cut[a] = b;
      }
    }
  }
  int n = 0;
  bounds[0] = 0;
  for (int a = 0; a < T && n < kMaxStages; a = next_cut[a]) bounds[++n] = next_cut[a];
  bounds[n] = T;
  if (n == 1) {  // the insurance cut (see above)
    for (int j = 1; j < T; ++j)
      if (surv[j] <= 0.001 * (double)B) {
        bounds[1] = j;
        bounds[2] = T;
        n = 2;
        break;
      }
  }
  return n;
}

// the planner alone, for tests and tools (no device, no handle): hist[16] as finalize_kernel reports it
extern "C" int cpmpc_plan_stages_from_histogram(const int64_t* hist, int64_t B, int max_iterations, int intervals, int dtype,
                                                int window_length, int32_t* bounds, int capacity) {
  // (max_iterations <= kFbBins - 1: with more, the last bin -- "this many or more" -- would stand for problems that
  // stopped early as well as for those that ran to the cap, and surv[] could not tell them apart)
  if (!hist || !bounds || B < 1 || max_iterations < 1 || max_iterations > kFbBins - 1 || intervals < 1 || intervals > 64 ||
      window_length < 1 || capacity < max_iterations + 1 || (dtype != CPMPC_F32 && dtype != CPMPC_F64))
    return -1;
  double h[kFbBins], n_hist = 0.0;
  for (int b = 0; b < kFbBins; ++b) {
    if (hist[b] < 0) return -1;
    h[b] = (double)hist[b];
    n_hist += h[b];
  }
  if (n_hist <= 0.0) return -1;
  int bnd[kMaxStages + 1];
  const int n = plan_from_histogram(h, n_hist, B, max_iterations, 64 / intervals, dtype == CPMPC_F64 ? 1024.0 : 2048.0,
                                    window_length, bnd);
  for (int i = 0; i <= n; ++i) bounds[i] = bnd[i];
  return n;
}

int cpmpc_plan_stages(cpmpc_solver* s, int slot, int64_t B, bool exits, int* bounds) {
  const int T = (int)s->params.max_iterations;
  const int L = s->S - 1, ppw = 64 / L;
  auto finish = [&](int n) {
    s->last_plan_n = n;
    for (int i = 0; i <= n; ++i) s->last_plan[i] = bounds[i];
    return n;
  };
  bounds[0] = 0;
  bounds[1] = T;
  if (!exits || s->active == nullptr || T < 2) return finish(1);
  auto fixed = [&](int first, int next) {  // `first` iterations, then `next` at a time (the last launch takes what is left)
    if (first <= 0 || next <= 0 || T <= first) return finish(1);
    int n = 1;
    bounds[1] = first;
    while (bounds[n] < T && n < kMaxStages) {
      bounds[n + 1] = bounds[n] + next < T ? bounds[n] + next : T;
      ++n;
    }
    bounds[n] = T;
    return finish(n);
  };
  if (!s->stage_auto) return fixed(s->stage_first, s->stage_next);
  // default: only batches beyond one round of resident waves (2 per SIMD) are staged at all -- a smaller one ends with
  // its slowest wave either way
  if ((B * (int64_t)L + 63) / 64 <= 2048) return finish(1);
  if (s->fb_host == nullptr || T > kFbBins - 1) return fixed(s->stage_first, s->stage_next);  // (the last bin must mean "the cap")
  // the counts of the reporting workgroups that carry the same step's stamp as the first one (a step still running has
  // stamped only some: they wait for a later plan)
  const int32_t* fb = s->fb_host + (size_t)slot * kFbReporters * (kFbBins + 1);
  // Relaxed loads on purpose: the kernel publishes with relaxed system-scope stores (a release there writes the L2 back,
  // 29 -> 94 us of finalize), so no ordering is promised and none is pretended here.  A reporter's stamp is read before
  // AND after its counts; a block whose stamp moved in between (the next step's finalize is writing it) is left out, like
  // one that has not been written yet.  What can still slip through is a block whose counts belong to a newer step than
  // its stamp: a hint that is one step off costs speed, never results -- the plan decides only where the launches are cut.
  const int seq = __atomic_load_n(&fb[kFbBins], __ATOMIC_RELAXED);
  double hist[kFbBins], n_hist = 0.0;
  for (int b = 0; b < kFbBins; ++b) hist[b] = 0.0;
  for (int r = 0; seq != 0 && r < s->fb_reporters[slot]; ++r) {
    const int32_t* rep = fb + (size_t)r * (kFbBins + 1);
    if (__atomic_load_n(&rep[kFbBins], __ATOMIC_RELAXED) != seq) continue;
    double c[kFbBins];
    for (int b = 0; b < kFbBins; ++b) c[b] = (double)__atomic_load_n(&rep[b], __ATOMIC_RELAXED);
    __atomic_thread_fence(__ATOMIC_ACQUIRE);  // (host-side only: keeps the compiler and the CPU from hoisting the re-read)
    if (__atomic_load_n(&rep[kFbBins], __ATOMIC_RELAXED) != seq) continue;
    for (int b = 0; b < kFbBins; ++b) {
      hist[b] += c[b];
      n_hist += c[b];
    }
  }
  if (seq == 0 || n_hist <= 0.0) return fixed(s->stage_first, s->stage_next);  // nothing to plan from yet
  // How old may the histogram be?  A caller that synchronises every tick (a controller acting on u) plans from the tick
  // before.  One that queues ticks ahead plans from whatever has finished, many ticks back: right while the iteration
  // counts hold still, wrong in a transient (a swing-up queued ahead ran 2.0 -> 2.7 ms per tick on plans made for ticks
  // long gone).  So: a histogram at most two steps old is used as it is; an older one only if it agrees with the one seen
  // before it (no bin's share moved by more than 2 % of the batch); otherwise the fixed pattern.
  PlanCache& pc = s->plan_cache[slot];
  if (pc.seen_seq != seq) {
    double moved = 0.0;
    for (int b = 0; b < kFbBins; ++b) {
      const double share = hist[b] / n_hist, d = share - pc.seen[b];
      moved = (d < 0.0 ? -d : d) > moved ? (d < 0.0 ? -d : d) : moved;
      pc.seen[b] = share;
    }
    pc.stationary = pc.have_prev && moved <= 0.02;
    pc.have_prev = true;
    pc.seen_seq = seq;
  }
  const int age = s->fb_seq[slot] >= seq ? s->fb_seq[slot] - seq : s->fb_seq[slot] + (1 << 30) - seq;  // steps launched since
  const bool fresh = age <= 2;
  if (!fresh && !pc.stationary) return fixed(s->stage_first, s->stage_next);
  // the same histogram and batch as the last plan made on this slot: the same plan
  if (pc.seq == seq && pc.B == B && pc.n_hist == n_hist && pc.n > 0) {
    for (int i = 0; i <= pc.n; ++i) bounds[i] = pc.bounds[i];
    return finish(pc.n);
  }
  const int n = plan_from_histogram(hist, n_hist, B, T, ppw, s->esize == 8 ? 1024.0 : 2048.0, s->N, bounds);
  pc.seq = seq;
  pc.B = B;
  pc.n_hist = n_hist;
  pc.n = n;
  for (int i = 0; i <= n; ++i) pc.bounds[i] = bounds[i];
  return finish(n);
}

extern "C" int cpmpc_get_stage_plan(const cpmpc_solver* s, int32_t* bounds, int capacity) {
  if (!s || !bounds || capacity < 2) return -1;
  const int n = s->last_plan_n < capacity - 1 ? s->last_plan_n : capacity - 1;
  for (int i = 0; i <= n; ++i) bounds[i] = s->last_plan[i];
  return n;
}

extern "C" int cpmpc_get_pipeline(const cpmpc_solver* s) {
  if (!s) return -1;
  return use_fused(s) ? CPMPC_PIPELINE_FUSED : CPMPC_PIPELINE_SPLIT;
}
extern "C" int cpmpc_step_batch(cpmpc_solver* s, int64_t B, const cpmpc_step_inputs* in,
                                const cpmpc_step_outputs* out, void* stream) {
  if (!s || !in) return fail(CPMPC_ERR_INVALID_ARG, "null argument");
  if (B < 1) return fail(CPMPC_ERR_INVALID_ARG, "B must be >= 1");
  if (B > s->cap) return fail(CPMPC_ERR_BATCH, "B=%lld exceeds the capacity %lld given to cpmpc_create", (long long)B, (long long)s->cap);
  if (!in->x0) return fail(CPMPC_ERR_INVALID_ARG, "x0 is required");
  if ((in->dyn_shared_host == nullptr) == (in->dyn == nullptr))
    return fail(CPMPC_ERR_INVALID_ARG, "exactly one of dyn_shared_host / dyn must be given");
  if (!in->set_point && !std::isfinite(in->set_point_shared))
    return fail(CPMPC_ERR_INVALID_ARG, "set_point_shared must be finite");
  DeviceGuard guard(s->device);
  int rc = CPMPC_OK;
  rc = engine_of(s)->step_batch(s, B, in, out, (hipStream_t)stream, 0, 0);
  track_caller_stream(s, (hipStream_t)stream);
  return rc;
}

// ------------------------------------------------------------------------------------------------
// warm-start state
// ------------------------------------------------------------------------------------------------
extern "C" int cpmpc_set_previous_solution(cpmpc_solver* s, int64_t B, const void* z, void* stream) {
  if (!s || !z) return fail(CPMPC_ERR_INVALID_ARG, "null argument");
  if (B < 1 || B > s->cap) return fail(CPMPC_ERR_BATCH, "B out of range");
  DeviceGuard guard(s->device);
  // packed [dim][B] (MapKey order) -> workspace layout
  engine_of(s)->pack_z(s, B, z, (hipStream_t)stream);  // packed [dim][B] (MapKey order) -> workspace layout
  HIP_TRY(hipGetLastError());
  if (B > s->prev_B) s->prev_B = B;
  track_caller_stream(s, (hipStream_t)stream);
  return CPMPC_OK;
}

extern "C" int cpmpc_get_solution(cpmpc_solver* s, int64_t B, void* z_out, void* stream) {
  if (!s || !z_out) return fail(CPMPC_ERR_INVALID_ARG, "null argument");
  if (B < 1 || B > s->cap) return fail(CPMPC_ERR_BATCH, "B out of range");
  DeviceGuard guard(s->device);
  engine_of(s)->unpack_z(s, B, z_out, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  // this read of zx/zu on the caller's stream must finish before a later host-pointer call (on the handle's own
  // stream) overwrites them
  track_caller_stream(s, (hipStream_t)stream);
  return CPMPC_OK;
}

// ------------------------------------------------------------------------------------------------
// host-pointer entry points: staging slots, worker threads, the chunk pipeline
// ------------------------------------------------------------------------------------------------
int ensure_slot(cpmpc_solver* s, int k, size_t bytes) {
  HostSlot& sl = s->slot[k];
  if (sl.stream == nullptr) {
    bool first = true;
    for (int i = 0; i < kHostSlots; ++i) first = first && s->slot[i].stream == nullptr;
    hipError_t e = hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking);
    if (e != hipSuccess) return fail(CPMPC_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e));
    e = hipEventCreateWithFlags(&sl.done, hipEventDisableTiming);
    if (e == hipSuccess && s->ev_last == nullptr) e = hipEventCreateWithFlags(&s->ev_last, hipEventDisableTiming);
    if (e != hipSuccess) return fail(CPMPC_ERR_HIP, "hipEventCreate failed: %s", hipGetErrorString(e));
    // device-pointer calls made before the first host-pointer call were not tracked by ev_last: order after them once;
    // a slot stream created later orders itself after whatever the other slots have queued the same way
    (void)first;
    HIP_TRY(hipDeviceSynchronize());
  }
  if (s->ev_pending) {  // a device-pointer call on a caller's stream came in between: every slot stream runs after it
    for (int i = 0; i < kHostSlots; ++i)
      if (s->slot[i].stream) HIP_TRY(hipStreamWaitEvent(s->slot[i].stream, s->ev_last, 0));
    s->ev_pending = false;
  }
  if (sl.bytes >= bytes) return CPMPC_OK;
  HIP_TRY(hipStreamSynchronize(sl.stream));
  if (sl.dev) (void)hipFree(sl.dev);
  if (sl.pin) (void)hipHostFree(sl.pin);
  sl.dev = nullptr;
  sl.pin = nullptr;
  sl.bytes = 0;
  const size_t want = bytes < 4096 ? 4096 : bytes;
  hipError_t e = hipMalloc(&sl.dev, want);
  if (e != hipSuccess) return fail(CPMPC_ERR_ALLOC, "hipMalloc of %zu staging bytes failed: %s", want, hipGetErrorString(e));
  e = hipHostMalloc(&sl.pin, want, hipHostMallocDefault);
  if (e != hipSuccess) {
    (void)hipFree(sl.dev);
    sl.dev = nullptr;
    return fail(CPMPC_ERR_ALLOC, "hipHostMalloc of %zu staging bytes failed: %s", want, hipGetErrorString(e));
  }
  sl.bytes = want;
  return CPMPC_OK;
}

// Worker threads for the CPU side of large host-pointer steps (conversion and scatter of the results: a 262 144-problem
// fp64 step returns 420 MB, which one thread moves at ~10 GB/s while PCIe delivers 55).  Created on first use, never
// destroyed (a process-lifetime pool; CPMPC_HOST_THREADS overrides the count, 1 = the calling thread alone).
namespace {
struct HostPool {
  std::mutex m;
  std::condition_variable cv_work, cv_done;
  void (*fn)(int64_t, void*) = nullptr;
  void* ctx = nullptr;
  int64_t n = 0;
  std::atomic<int64_t> next{0};
  uint64_t generation = 0;
  int running = 0;
  int workers = 0;
  std::mutex call;  // one parallel_for at a time
};
HostPool* g_pool = nullptr;
std::once_flag g_pool_once;

void pool_worker(HostPool* p) {
  uint64_t seen = 0;
  for (;;) {
    std::unique_lock<std::mutex> lk(p->m);
    p->cv_work.wait(lk, [&] { return p->generation != seen; });
    seen = p->generation;
    void (*fn)(int64_t, void*) = p->fn;
    void* ctx = p->ctx;
    const int64_t n = p->n;
    lk.unlock();
    for (int64_t i = p->next.fetch_add(1); i < n; i = p->next.fetch_add(1)) fn(i, ctx);
    lk.lock();
    if (--p->running == 0) p->cv_done.notify_all();
  }
}
}  // namespace

void host_parallel_for(int64_t n, void (*fn)(int64_t, void*), void* ctx) {
  std::call_once(g_pool_once, [] {
    g_pool = new HostPool();
    int want = 0;
    if (const char* e = getenv("CPMPC_HOST_THREADS")) want = atoi(e);
    if (want <= 0) {
      const unsigned hw = std::thread::hardware_concurrency();
      want = hw >= 16 ? 8 : (hw >= 4 ? (int)hw / 2 : 1);
    }
    g_pool->workers = want - 1;
    for (int i = 0; i < g_pool->workers; ++i) std::thread(pool_worker, g_pool).detach();
  });
  HostPool* p = g_pool;
  if (p->workers == 0 || n <= 1) {
    for (int64_t i = 0; i < n; ++i) fn(i, ctx);
    return;
  }
  std::lock_guard<std::mutex> one(p->call);
  {
    std::lock_guard<std::mutex> lk(p->m);
    p->fn = fn;
    p->ctx = ctx;
    p->n = n;
    p->next.store(0);
    p->running = p->workers;
    ++p->generation;
  }
  p->cv_work.notify_all();
  for (int64_t i = p->next.fetch_add(1); i < n; i = p->next.fetch_add(1)) fn(i, ctx);
  std::unique_lock<std::mutex> lk(p->m);
  p->cv_done.wait(lk, [&] { return p->running == 0; });
}

extern "C" int cpmpc_host_register(void* ptr, uint64_t bytes) {
  if (!ptr || bytes == 0) return fail(CPMPC_ERR_INVALID_ARG, "null argument");
  const hipError_t e = hipHostRegister(ptr, (size_t)bytes, hipHostRegisterDefault);
  if (e != hipSuccess) return fail(CPMPC_ERR_HIP, "hipHostRegister of %llu bytes failed: %s", (unsigned long long)bytes, hipGetErrorString(e));
  return CPMPC_OK;
}
extern "C" int cpmpc_host_unregister(void* ptr) {
  if (!ptr) return fail(CPMPC_ERR_INVALID_ARG, "null argument");
  const hipError_t e = hipHostUnregister(ptr);
  if (e != hipSuccess) return fail(CPMPC_ERR_HIP, "hipHostUnregister failed: %s", hipGetErrorString(e));
  return CPMPC_OK;
}

extern "C" int cpmpc_set_host_chunk(cpmpc_solver* s, int64_t problems) {
  if (!s) return fail(CPMPC_ERR_INVALID_ARG, "null solver");
  if (problems < -1) return fail(CPMPC_ERR_INVALID_ARG, "chunk size must be >= 0 (0 = never split), or -1 for the default");
  s->host_chunk = problems <= 0 ? problems : (problems + 63) / 64 * 64;
  return CPMPC_OK;
}

static bool host_ptr_is_pinned(const void* p) {
  if (!p) return false;
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, p) != hipSuccess) {
    (void)hipGetLastError();  // an ordinary (pageable) host pointer is reported as an error
    return false;
  }
  return at.type == hipMemoryTypeHost;
}

// DMA straight into the caller's arrays pays when the predicted states are asked for (160 of the 200 rows a problem
// returns) and every real-typed output array is pinned; for u alone the worker threads' scatter of the mirror is hidden
// behind the kernels and the strided device-to-host copies are not (measured, profiles/r04_host_path.json).
static bool host_direct_outputs(const cpmpc_solver* s, const cpmpc_step_host_outputs& out) {
  if (s->dtype != CPMPC_F64 || out.predicted == nullptr) return false;
  if (const char* e = getenv("CPMPC_HOST_DIRECT"))  // measurement switch: 0 = always through the mirror
    if (e[0] == '0') return false;
  if (!host_ptr_is_pinned(out.predicted) || !host_ptr_is_pinned(out.u)) return false;
  if (out.solution && !host_ptr_is_pinned(out.solution)) return false;
  return true;
}

// One chunk of a host-pointer step: problems [c0, c0 + Bc) of handle h = columns [g0, g0 + Bc) of the caller's arrays
struct HostWork {
  cpmpc_solver* h;
  int64_t c0, Bc, g0;
};

// the chunks of problems [0, Bh) of handle h (its columns start at g_base in the caller's arrays), appended to `work`
static void host_chunks_of(cpmpc_solver* h, int64_t Bh, int64_t g_base, bool direct,
                           std::vector<std::vector<HostWork>>& per_handle) {
  std::vector<HostWork> w;
  int64_t n = 1;
  // default: eight chunks of at least 16 384 problems; sixteen of at least 8 192 when the results travel by DMA into the
  // caller's arrays (no CPU scatter to amortise: smaller chunks shorten the drain of the pipeline)
  int64_t chunk = h->host_chunk;
  if (chunk < 0) chunk = direct ? (Bh / 16 > 8192 ? Bh / 16 : 8192) : (Bh / 8 > 16384 ? Bh / 8 : 16384);
  if (chunk > 0 && Bh > chunk + chunk / 2) n = (Bh + chunk - 1) / chunk;
  const int64_t step = ((Bh + n - 1) / n + 63) / 64 * 64;
  for (int64_t c0 = 0; c0 < Bh; c0 += step) w.push_back(HostWork{h, c0, (Bh - c0 < step ? Bh - c0 : step), g_base + c0});
  per_handle.push_back(std::move(w));
}

// Runs the chunks as a pipeline: a handle's chunks rotate through its kHostSlots staging slots, so that while the CPU
// scatters one chunk's results the next is copying back and the one after is in the kernels; the chunks of several
// handles (the shards of cpmpc_sharded_*) are issued round-robin.  Results do not depend on the chunking: a problem's
// arithmetic does not depend on the lanes it occupies or on its neighbours.  Returns after every chunk has landed.
static int run_host_pipeline(const std::vector<std::vector<HostWork>>& per_handle, int64_t ld,
                             const cpmpc_step_host_inputs& in, const cpmpc_step_host_outputs& out, bool direct) {
  struct Flight {
    cpmpc_solver* h;
    int slot;
  };
  std::deque<Flight> inflight;
  int first_rc = CPMPC_OK;
  auto end_front = [&]() {
    const Flight f = inflight.front();
    inflight.pop_front();
    DeviceGuard guard(f.h->device);
    if (first_rc == CPMPC_OK) {
      const int rc = engine_of(f.h)->host_chunk_end(f.h, f.slot, ld, out);
      if (rc != CPMPC_OK) first_rc = rc;
    } else {  // after a failure: drain what was started, deliver nothing more
      (void)hipStreamSynchronize(f.h->slot[f.slot].stream);
      f.h->slot[f.slot].busy = false;
    }
  };
  size_t rounds = 0;
  for (const auto& w : per_handle) rounds = w.size() > rounds ? w.size() : rounds;
  for (size_t k = 0; k < rounds && first_rc == CPMPC_OK; ++k) {
    for (const auto& w : per_handle) {
      if (k >= w.size() || first_rc != CPMPC_OK) continue;
      const HostWork& c = w[k];
      const int slot = (int)(k % kHostSlots);
      while (c.h->slot[slot].busy && !inflight.empty()) end_front();  // oldest first: it is the one most likely done
      if (first_rc != CPMPC_OK) break;
      DeviceGuard guard(c.h->device);
      const int rc = engine_of(c.h)->host_chunk_begin(c.h, slot, c.c0, c.Bc, c.g0, ld, in, out, direct);
      if (rc != CPMPC_OK) {
        first_rc = rc;
        break;
      }
      inflight.push_back(Flight{c.h, slot});
    }
  }
  while (!inflight.empty()) end_front();
  return first_rc;
}

static int check_host_inputs(const cpmpc_step_host_inputs* in, const cpmpc_step_host_outputs* out) {
  if (!in || !out) return fail(CPMPC_ERR_INVALID_ARG, "null argument");
  if (!in->x0) return fail(CPMPC_ERR_INVALID_ARG, "x0 is required");
  if ((in->dyn_shared == nullptr) == (in->dyn == nullptr))
    return fail(CPMPC_ERR_INVALID_ARG, "exactly one of dyn_shared / dyn must be given");
  if (!in->set_point && !std::isfinite(in->set_point_shared))
    return fail(CPMPC_ERR_INVALID_ARG, "set_point_shared must be finite");
  return CPMPC_OK;
}

extern "C" int cpmpc_step_batch_host_in(cpmpc_solver* s, int64_t B, const cpmpc_step_host_inputs* in,
                                        const cpmpc_step_host_outputs* out) {
  if (!s) return fail(CPMPC_ERR_INVALID_ARG, "null argument");
  int rc = check_host_inputs(in, out);
  if (rc) return rc;
  if (B < 1) return fail(CPMPC_ERR_INVALID_ARG, "B must be >= 1");
  if (B > s->cap) return fail(CPMPC_ERR_BATCH, "B exceeds capacity");
  std::vector<std::vector<HostWork>> work;
  DeviceGuard guard(s->device);
  const bool direct = host_direct_outputs(s, *out);
  host_chunks_of(s, B, 0, direct, work);
  return run_host_pipeline(work, B, *in, *out, direct);
}

extern "C" int cpmpc_step_batch_host_ex(cpmpc_solver* s, int64_t B, const double* x0_host,
                                        const double* dyn_shared_host, double set_point,
                                        const cpmpc_step_host_outputs* out) {
  if (!s || !x0_host || !dyn_shared_host || !out) return fail(CPMPC_ERR_INVALID_ARG, "null argument");
  if (!std::isfinite(set_point)) return fail(CPMPC_ERR_INVALID_ARG, "set_point must be finite");
  const cpmpc_step_host_inputs in = {x0_host, dyn_shared_host, nullptr, set_point, nullptr, nullptr};
  return cpmpc_step_batch_host_in(s, B, &in, out);
}

extern "C" int cpmpc_step_batch_host(cpmpc_solver* s, int64_t B, const double* x0_host,
                                     const double* dyn_shared_host, double set_point, double* u_host,
                                     double* predicted_host, int32_t* status_host, int32_t* iterations_host,
                                     double* final_cost_host, double* final_eq_l1_host) {
  const cpmpc_step_host_outputs out = {u_host, predicted_host, status_host, iterations_host, final_cost_host,
                                       final_eq_l1_host, nullptr};
  return cpmpc_step_batch_host_ex(s, B, x0_host, dyn_shared_host, set_point, &out);
}

// packed z [dim][n] in the handle's dtype <-> rows [dim] of the caller's double array [dim][ld] at column g0
static int set_prev_host_cols(cpmpc_solver* s, int64_t n, const double* z_host, int64_t ld, int64_t g0) {
  DeviceGuard guard(s->device);
  const size_t cnt = (size_t)s->dim * (size_t)n;
  int rc = ensure_slot(s, 0, cnt * s->esize);
  if (rc) return rc;
  HostSlot& sl = s->slot[0];
  for (int r = 0; r < s->dim; ++r) {
    const double* src = z_host + (size_t)r * (size_t)ld + (size_t)g0;
    if (s->dtype == CPMPC_F32) {
      float* h = (float*)sl.pin + (size_t)r * (size_t)n;
      for (int64_t i = 0; i < n; ++i) h[i] = (float)src[i];
    } else {
      memcpy((double*)sl.pin + (size_t)r * (size_t)n, src, (size_t)n * 8);
    }
  }
  HIP_TRY(hipMemcpyAsync(sl.dev, sl.pin, cnt * s->esize, hipMemcpyHostToDevice, sl.stream));
  rc = cpmpc_set_previous_solution(s, n, sl.dev, sl.stream);
  const hipError_t e = hipStreamSynchronize(sl.stream);  // also on failure: the copy above still reads the pinned mirror
  if (rc) return rc;
  if (e != hipSuccess) return fail(CPMPC_ERR_HIP, "hipStreamSynchronize failed: %s", hipGetErrorString(e));
  return CPMPC_OK;
}

static int get_sol_host_cols(cpmpc_solver* s, int64_t n, double* z_host, int64_t ld, int64_t g0) {
  DeviceGuard guard(s->device);
  const size_t cnt = (size_t)s->dim * (size_t)n;
  int rc = ensure_slot(s, 0, cnt * s->esize);
  if (rc) return rc;
  HostSlot& sl = s->slot[0];
  rc = cpmpc_get_solution(s, n, sl.dev, sl.stream);
  if (rc) return rc;
  {
    const hipError_t e = hipMemcpyAsync(sl.pin, sl.dev, cnt * s->esize, hipMemcpyDeviceToHost, sl.stream);
    const hipError_t e2 = hipStreamSynchronize(sl.stream);  // also on failure: the unpack kernel is in flight
    if (e != hipSuccess) return fail(CPMPC_ERR_HIP, "hipMemcpyAsync failed: %s", hipGetErrorString(e));
    if (e2 != hipSuccess) return fail(CPMPC_ERR_HIP, "hipStreamSynchronize failed: %s", hipGetErrorString(e2));
  }
  for (int r = 0; r < s->dim; ++r) {
    double* dst = z_host + (size_t)r * (size_t)ld + (size_t)g0;
    if (s->dtype == CPMPC_F32) {
      const float* h = (const float*)sl.pin + (size_t)r * (size_t)n;
      for (int64_t i = 0; i < n; ++i) dst[i] = (double)h[i];
    } else {
      memcpy(dst, (const double*)sl.pin + (size_t)r * (size_t)n, (size_t)n * 8);
    }
  }
  return CPMPC_OK;
}

extern "C" int cpmpc_set_previous_solution_host(cpmpc_solver* s, int64_t B, const double* z_host) {
  if (!s || !z_host) return fail(CPMPC_ERR_INVALID_ARG, "null argument");
  if (B < 1 || B > s->cap) return fail(CPMPC_ERR_BATCH, "B out of range");
  return set_prev_host_cols(s, B, z_host, B, 0);
}

extern "C" int cpmpc_get_solution_host(cpmpc_solver* s, int64_t B, double* z_host) {
  if (!s || !z_host) return fail(CPMPC_ERR_INVALID_ARG, "null argument");
  if (B < 1 || B > s->cap) return fail(CPMPC_ERR_BATCH, "B out of range");
  return get_sol_host_cols(s, B, z_host, B, 0);
}

// ------------------------------------------------------------------------------------------------
// several GPUs from one process: one handle + stream per shard, contiguous split, concurrent shards
// ------------------------------------------------------------------------------------------------
struct Shard {
  cpmpc_solver* h = nullptr;
  int device = 0;
  hipStream_t stream = nullptr;   // device-pointer steps of this shard run here
  hipEvent_t done = nullptr;      // shard stream -> root stream (results have landed on the root device)
  void* buf = nullptr;            // per-shard device staging of the device-pointer step and of the warm-start hand-over
  size_t buf_bytes = 0;
  int peer = 1;                   // what cpmpc_sharded_create saw: 1 root <-> this device mapped both ways (or the same device), 0 not
};

struct cpmpc_sharded {
  std::vector<Shard> shards;
  int dtype = CPMPC_F64;
  int N = 0, NX = 4, NP = 9, dim = 0;
  size_t esize = 8;
  int64_t cap = 0;
  hipEvent_t ready = nullptr;  // on the ROOT device: the caller's inputs are there (root stream -> every shard stream)
  // Warm-start bookkeeping.  The split of a batch depends on its size, so the shards' previous solutions are those of
  // columns [0, warm_total) split as a batch of dist_B problems is split; a step (or set / get) with another size first
  // hands the warm start over to the new split (sharded_align).
  int64_t dist_B = 0;
  int64_t warm_total = 0;
};

static void shard_range(int64_t total, int i, int n, int64_t* lo, int64_t* hi) {
  const int64_t base = total / n, rem = total % n;
  *lo = (int64_t)i * base + (i < rem ? i : rem);
  *hi = *lo + base + (i < rem ? 1 : 0);
}

extern "C" void cpmpc_sharded_destroy(cpmpc_sharded* s) {
  if (!s) return;
  for (auto& sh : s->shards) {
    DeviceGuard guard(sh.device);
    if (sh.stream) (void)hipStreamSynchronize(sh.stream);
    if (sh.h) cpmpc_destroy(sh.h);
    if (sh.buf) (void)hipFree(sh.buf);
    if (sh.done) (void)hipEventDestroy(sh.done);
    if (sh.stream) (void)hipStreamDestroy(sh.stream);
  }
  if (s->ready && !s->shards.empty()) {
    DeviceGuard guard(s->shards[0].device);
    (void)hipEventDestroy(s->ready);
  }
  delete s;
}

extern "C" int cpmpc_sharded_create_ex(const cpmpc_create_info* info, const int* devices, int n_devices,
                                       cpmpc_sharded** out) {
  if (!info || !out) return fail(CPMPC_ERR_INVALID_ARG, "null argument");
  *out = nullptr;
  if (info->struct_size != sizeof(cpmpc_create_info))
    return fail(CPMPC_ERR_INVALID_ARG, "cpmpc_create_info.struct_size is %u, this library's is %zu", info->struct_size,
                sizeof(cpmpc_create_info));
  std::vector<int> devs;
  if (devices == nullptr) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
      return fail(CPMPC_ERR_NO_DEVICE, "no HIP device visible; this library has no CPU fallback");
    for (int i = 0; i < n; ++i)
      if (device_is_gfx950(i)) devs.push_back(i);
    if (devs.empty()) return fail(CPMPC_ERR_NO_DEVICE, "no gfx950 device visible");
  } else {
    if (n_devices < 1 || n_devices > 64) return fail(CPMPC_ERR_INVALID_ARG, "n_devices must be in [1, 64]");
    devs.assign(devices, devices + n_devices);
  }
  const int n = (int)devs.size();
  if (info->max_batch < n) return fail(CPMPC_ERR_INVALID_ARG, "max_batch must be at least the number of shards");
  cpmpc_sharded* s = new (std::nothrow) cpmpc_sharded();
  if (!s) return fail(CPMPC_ERR_ALLOC, "out of host memory");
  s->dtype = info->dtype;
  s->esize = info->dtype == CPMPC_F32 ? 4 : 8;
  s->cap = info->max_batch;
  s->shards.resize(n);
  for (int i = 0; i < n; ++i) {
    Shard& sh = s->shards[i];
    sh.device = devs[i];
    int64_t lo, hi;
    shard_range(info->max_batch, i, n, &lo, &hi);
    cpmpc_create_info one = *info;
    one.device = sh.device;
    one.max_batch = hi - lo + 1;  // +1: a smaller B may shift a remainder here
    int rc = cpmpc_create_ex(&one, &sh.h);
    if (rc == CPMPC_OK) {
      DeviceGuard guard(sh.device);
      if (hipStreamCreateWithFlags(&sh.stream, hipStreamNonBlocking) != hipSuccess ||
          hipEventCreateWithFlags(&sh.done, hipEventDisableTiming) != hipSuccess)
        rc = fail(CPMPC_ERR_HIP, "stream / event creation failed on device %d", sh.device);
    }
    if (rc != CPMPC_OK) {
      cpmpc_sharded_destroy(s);
      return rc;
    }
  }
  s->N = s->shards[0].h->N;
  s->NX = s->shards[0].h->NX;
  s->NP = s->shards[0].h->NP;
  s->dim = s->shards[0].h->dim;
  const int root = s->shards[0].device;
  {  // an event may only be recorded on a stream of the device it was created on: `ready` belongs to the ROOT device
    DeviceGuard guard(root);
    if (hipEventCreateWithFlags(&s->ready, hipEventDisableTiming) != hipSuccess) {
      cpmpc_sharded_destroy(s);
      return fail(CPMPC_ERR_HIP, "event creation failed on device %d", root);
    }
  }
  // peer access between the root device and every other shard's device (both directions); a pair that cannot be
  // mapped still works, the copies then go through host memory
  for (int i = 1; i < n; ++i) {
    const int d = s->shards[i].device;
    if (d == root) continue;
    int can = 0, both = 0;
    if (hipDeviceCanAccessPeer(&can, root, d) == hipSuccess && can) {
      DeviceGuard guard(root);
      const hipError_t pe = hipDeviceEnablePeerAccess(d, 0);
      both += (pe == hipSuccess || pe == hipErrorPeerAccessAlreadyEnabled);
    }
    if (hipDeviceCanAccessPeer(&can, d, root) == hipSuccess && can) {
      DeviceGuard guard(d);
      const hipError_t pe = hipDeviceEnablePeerAccess(root, 0);
      both += (pe == hipSuccess || pe == hipErrorPeerAccessAlreadyEnabled);
    }
    s->shards[i].peer = both == 2;
    (void)hipGetLastError();  // "already enabled" is fine
  }
  *out = s;
  return CPMPC_OK;
}

extern "C" int cpmpc_sharded_create(const cpmpc_params* params, const cpmpc_solver_opts* opts, int dtype,
                                    int64_t max_batch, const int* devices, int n_devices, cpmpc_sharded** out) {
  if (!params || !out) return fail(CPMPC_ERR_INVALID_ARG, "null argument");
  cpmpc_create_info info;
  memset(&info, 0, sizeof info);
  info.struct_size = sizeof info;
  info.dtype = dtype;
  info.model = CPMPC_MODEL_SINGLE;
  info.max_batch = max_batch;
  info.params = params;
  info.opts = opts;
  info.opts_size = opts ? CPMPC_SOLVER_OPTS_SIZE_POSITIONAL : 0;  // a positional constructor: the struct as it was frozen
  return cpmpc_sharded_create_ex(&info, devices, n_devices, out);
}

extern "C" int cpmpc_sharded_num_shards(const cpmpc_sharded* s) { return s ? (int)s->shards.size() : -1; }
extern "C" int cpmpc_sharded_peer_access(const cpmpc_sharded* s, int shard) {
  return (s && shard >= 0 && shard < (int)s->shards.size()) ? s->shards[shard].peer : -1;
}
extern "C" int cpmpc_sharded_device(const cpmpc_sharded* s, int shard) {
  return (s && shard >= 0 && shard < (int)s->shards.size()) ? s->shards[shard].device : -1;
}
extern "C" cpmpc_solver* cpmpc_sharded_handle(cpmpc_sharded* s, int shard) {
  return (s && shard >= 0 && shard < (int)s->shards.size()) ? s->shards[shard].h : nullptr;
}
extern "C" int cpmpc_sharded_range(const cpmpc_sharded* s, int shard, int64_t B, int64_t* lo, int64_t* hi) {
  if (!s || !lo || !hi || shard < 0 || shard >= (int)s->shards.size() || B < 0)
    return fail(CPMPC_ERR_INVALID_ARG, "bad argument");
  shard_range(B, shard, (int)s->shards.size(), lo, hi);
  return CPMPC_OK;
}
extern "C" int cpmpc_sharded_reset(cpmpc_sharded* s) {
  if (!s) return fail(CPMPC_ERR_INVALID_ARG, "null solver");
  for (auto& sh : s->shards) cpmpc_reset(sh.h);
  s->dist_B = 0;
  s->warm_total = 0;
  return CPMPC_OK;
}
extern "C" int64_t cpmpc_sharded_previous_solution_batch(const cpmpc_sharded* s) { return s ? s->warm_total : 0; }
// every shard was created from the same parameters: the status of shard 0 is the handle's
extern "C" int cpmpc_sharded_horizon_beyond_parity(const cpmpc_sharded* s) {
  return (s && !s->shards.empty()) ? cpmpc_horizon_beyond_parity(s->shards[0].h) : -1;
}

static int sharded_check(const cpmpc_sharded* s, int64_t B) {
  if (B < 1) return fail(CPMPC_ERR_INVALID_ARG, "B must be >= 1");
  if (B > s->cap) return fail(CPMPC_ERR_BATCH, "B=%lld exceeds the capacity %lld given to cpmpc_sharded_create", (long long)B, (long long)s->cap);
  return CPMPC_OK;
}

static int ensure_shard_buf(Shard& sh, size_t bytes) {
  if (sh.buf_bytes >= bytes) return CPMPC_OK;
  HIP_TRY(hipStreamSynchronize(sh.stream));
  if (sh.buf) (void)hipFree(sh.buf);
  sh.buf = nullptr;
  sh.buf_bytes = 0;
  HIP_TRY(hipMalloc(&sh.buf, bytes));
  sh.buf_bytes = bytes;
  return CPMPC_OK;
}

// How many of shard i's problems, split as a batch of `dist` problems is split, lie in columns [0, warm)
static int64_t warm_in_shard(int64_t dist, int i, int n, int64_t warm, int64_t* lo_out) {
  int64_t lo, hi;
  shard_range(dist, i, n, &lo, &hi);
  if (lo_out) *lo_out = lo;
  const int64_t w = warm - lo;
  return w < 0 ? 0 : (w > hi - lo ? hi - lo : w);
}

// The shards hold the previous solutions of columns [0, warm_total) split as a batch of dist_B problems is split.  A call
// with another batch size B would pair every shard's warm start with other columns: hand the warm start over to B's
// split first -- gather z of the warm columns on the root device, reset, scatter by the new ranges.  Columns beyond B
// are dropped (a single handle would keep them; a sharded one has nowhere to put them).  Rare and synchronous.
static int sharded_align(cpmpc_sharded* s, int64_t B) {
  if (s->warm_total == 0 || s->dist_B == B) {
    if (s->warm_total == 0) s->dist_B = B;
    return CPMPC_OK;
  }
  const int n = (int)s->shards.size();
  const int64_t W = s->warm_total;
  const size_t es = s->esize, dim = (size_t)s->dim;
  const int root = s->shards[0].device;
  void* tmp = nullptr;
  {
    DeviceGuard guard(root);
    HIP_TRY(hipMalloc(&tmp, dim * (size_t)W * es));
  }
  int rc = CPMPC_OK;
  auto body = [&]() -> int {
    for (int i = 0; i < n; ++i) {  // gather [dim][n_i] of every shard into columns [lo_i, lo_i + n_i) of tmp [dim][W]
      int64_t lo;
      const int64_t ni = warm_in_shard(s->dist_B, i, n, W, &lo);
      if (ni == 0) continue;
      Shard& sh = s->shards[i];
      DeviceGuard guard(sh.device);
      int r = ensure_shard_buf(sh, dim * (size_t)ni * es);
      if (r) return r;
      r = cpmpc_get_solution(sh.h, ni, sh.buf, sh.stream);
      if (r) return r;
      HIP_TRY(hipMemcpy2DAsync((char*)tmp + (size_t)lo * es, (size_t)W * es, sh.buf, (size_t)ni * es, (size_t)ni * es, dim,
                               hipMemcpyDefault, sh.stream));
      HIP_TRY(hipStreamSynchronize(sh.stream));
    }
    for (auto& sh : s->shards) cpmpc_reset(sh.h);
    const int64_t keep = W < B ? W : B;
    for (int i = 0; i < n; ++i) {
      int64_t lo;
      const int64_t ni = warm_in_shard(B, i, n, keep, &lo);
      if (ni == 0) continue;
      Shard& sh = s->shards[i];
      DeviceGuard guard(sh.device);
      int r = ensure_shard_buf(sh, dim * (size_t)ni * es);
      if (r) return r;
      HIP_TRY(hipMemcpy2DAsync(sh.buf, (size_t)ni * es, (const char*)tmp + (size_t)lo * es, (size_t)W * es, (size_t)ni * es, dim,
                               hipMemcpyDefault, sh.stream));
      r = cpmpc_set_previous_solution(sh.h, ni, sh.buf, sh.stream);
      if (r) return r;
      HIP_TRY(hipStreamSynchronize(sh.stream));
    }
    s->dist_B = B;
    s->warm_total = keep;
    return CPMPC_OK;
  };
  rc = body();
  {
    DeviceGuard guard(root);
    for (auto& sh : s->shards) (void)hipStreamSynchronize(sh.stream);
    (void)hipFree(tmp);
  }
  if (rc != CPMPC_OK) {  // half-moved warm starts are worse than none
    for (auto& sh : s->shards) cpmpc_reset(sh.h);
    s->dist_B = B;
    s->warm_total = 0;
  }
  return rc;
}

extern "C" int cpmpc_sharded_step_batch_host_in(cpmpc_sharded* s, int64_t B, const cpmpc_step_host_inputs* in,
                                                const cpmpc_step_host_outputs* out) {
  if (!s) return fail(CPMPC_ERR_INVALID_ARG, "null argument");
  int rc = check_host_inputs(in, out);
  if (rc) return rc;
  rc = sharded_check(s, B);
  if (rc) return rc;
  rc = sharded_align(s, B);
  if (rc) return rc;
  const int n = (int)s->shards.size();
  // every shard's chunks -- upload, kernels, download on its own streams -- are in flight together
  std::vector<std::vector<HostWork>> work;
  bool direct;
  {
    DeviceGuard guard(s->shards[0].device);
    direct = host_direct_outputs(s->shards[0].h, *out);
  }
  for (int i = 0; i < n; ++i) {
    int64_t lo, hi;
    shard_range(B, i, n, &lo, &hi);
    if (hi > lo) host_chunks_of(s->shards[i].h, hi - lo, lo, direct, work);
  }
  rc = run_host_pipeline(work, B, *in, *out, direct);
  if (rc == CPMPC_OK) s->warm_total = B;
  else cpmpc_sharded_reset(s);  // some shards stepped, others did not: no consistent warm start is left
  return rc;
}

extern "C" int cpmpc_sharded_step_batch_host(cpmpc_sharded* s, int64_t B, const double* x0_host,
                                             const double* dyn_shared_host, double set_point,
                                             const cpmpc_step_host_outputs* out) {
  if (!s || !x0_host || !dyn_shared_host || !out) return fail(CPMPC_ERR_INVALID_ARG, "null argument");
  if (!std::isfinite(set_point)) return fail(CPMPC_ERR_INVALID_ARG, "set_point must be finite");
  const cpmpc_step_host_inputs in = {x0_host, dyn_shared_host, nullptr, set_point, nullptr, nullptr};
  return cpmpc_sharded_step_batch_host_in(s, B, &in, out);
}

extern "C" int cpmpc_sharded_step_batch_ex(cpmpc_sharded* s, int64_t B, const cpmpc_step_inputs* in,
                                           const cpmpc_step_outputs* out, void* stream) {
  if (!s || !in || !out) return fail(CPMPC_ERR_INVALID_ARG, "null argument");
  if (!in->x0) return fail(CPMPC_ERR_INVALID_ARG, "x0 is required");
  if ((in->dyn_shared_host == nullptr) == (in->dyn == nullptr))
    return fail(CPMPC_ERR_INVALID_ARG, "exactly one of dyn_shared_host / dyn must be given");
  if (!in->set_point && !std::isfinite(in->set_point_shared))
    return fail(CPMPC_ERR_INVALID_ARG, "set_point_shared must be finite");
  int rc = sharded_check(s, B);
  if (rc) return rc;
  rc = sharded_align(s, B);
  if (rc) return rc;
  const int n = (int)s->shards.size();
  const size_t es = s->esize;
  const hipStream_t root_stream = (hipStream_t)stream;
  const int root = s->shards[0].device;
  const size_t NX = (size_t)s->NX, N = (size_t)s->N, NP = (size_t)s->NP, dim = (size_t)s->dim;
  // the caller's inputs are ready where its stream is now: ONE event of the root device, every shard stream waits on it
  {
    DeviceGuard guard(root);
    HIP_TRY(hipEventRecord(s->ready, root_stream));
  }
  int started = 0;
  auto one_shard = [&](int i) -> int {
    int64_t lo, hi;
    shard_range(B, i, n, &lo, &hi);
    if (hi == lo) return CPMPC_OK;
    const size_t Bs = (size_t)(hi - lo);
    Shard& sh = s->shards[i];
    DeviceGuard guard(sh.device);
    // per-shard staging on the shard's own device, 256-byte aligned pieces (absent arrays take no room)
    size_t off = 0;
    auto piece = [&](bool present, size_t bytes) {
      const size_t o = off;
      if (present) off += (bytes + 255) & ~(size_t)255;
      return o;
    };
    const size_t o_x0 = piece(true, NX * Bs * es), o_dyn = piece(in->dyn != nullptr, NP * Bs * es),
                 o_sp = piece(in->set_point != nullptr, Bs * es), o_tw = piece(in->terminal_weights != nullptr, NX * Bs * es),
                 o_u = piece(out->u != nullptr, N * Bs * es), o_pred = piece(out->predicted != nullptr, N * NX * Bs * es),
                 o_cost = piece(out->final_cost != nullptr, Bs * es), o_eq = piece(out->final_eq_l1 != nullptr, Bs * es),
                 o_st = piece(out->status != nullptr, Bs * 4), o_it = piece(out->iterations != nullptr, Bs * 4),
                 o_ls = piece(out->ls_evals != nullptr, Bs * 4), o_guess = piece(out->guess != nullptr, dim * Bs * es),
                 o_sol = piece(out->solution != nullptr, dim * Bs * es);
    int r = ensure_shard_buf(sh, off ? off : 256);
    if (r) return r;
    char* b = (char*)sh.buf;
    HIP_TRY(hipStreamWaitEvent(sh.stream, s->ready, 0));
    started = i + 1;  // from here on work of this call is (or may be) in flight on sh.stream
    // scatter: my columns of an [rows][B] array on the root device -> [rows][Bs] here
    auto scatter = [&](const void* src, size_t o, size_t rows) -> hipError_t {
      return hipMemcpy2DAsync(b + o, Bs * es, (const char*)src + (size_t)lo * es, (size_t)B * es, Bs * es, rows,
                              hipMemcpyDefault, sh.stream);
    };
    HIP_TRY(scatter(in->x0, o_x0, NX));
    if (in->dyn) HIP_TRY(scatter(in->dyn, o_dyn, NP));
    if (in->set_point) HIP_TRY(scatter(in->set_point, o_sp, 1));
    if (in->terminal_weights) HIP_TRY(scatter(in->terminal_weights, o_tw, NX));
    cpmpc_step_inputs si = *in;
    si.x0 = b + o_x0;
    si.dyn = in->dyn ? b + o_dyn : nullptr;
    si.set_point = in->set_point ? b + o_sp : nullptr;
    si.terminal_weights = in->terminal_weights ? b + o_tw : nullptr;
    cpmpc_step_outputs o;
    memset(&o, 0, sizeof o);
    o.u = out->u ? b + o_u : nullptr;
    o.predicted = out->predicted ? b + o_pred : nullptr;
    o.final_cost = out->final_cost ? b + o_cost : nullptr;
    o.final_eq_l1 = out->final_eq_l1 ? b + o_eq : nullptr;
    o.status = out->status ? (int32_t*)(b + o_st) : nullptr;
    o.iterations = out->iterations ? (int32_t*)(b + o_it) : nullptr;
    o.ls_evals = out->ls_evals ? (int32_t*)(b + o_ls) : nullptr;
    o.guess = out->guess ? b + o_guess : nullptr;
    o.solution = out->solution ? b + o_sol : nullptr;
    r = cpmpc_step_batch(sh.h, (int64_t)Bs, &si, &o, sh.stream);
    if (r) return r;
    // gather: rows of Bs scalars here -> rows of B scalars on the root device, at column lo
    auto gather = [&](void* dst, const void* src, size_t rows, size_t e) -> hipError_t {
      if (!dst) return hipSuccess;
      return hipMemcpy2DAsync((char*)dst + (size_t)lo * e, (size_t)B * e, src, Bs * e, Bs * e, rows, hipMemcpyDefault,
                              sh.stream);
    };
    HIP_TRY(gather(out->u, b + o_u, N, es));
    HIP_TRY(gather(out->predicted, b + o_pred, N * NX, es));
    HIP_TRY(gather(out->final_cost, b + o_cost, 1, es));
    HIP_TRY(gather(out->final_eq_l1, b + o_eq, 1, es));
    HIP_TRY(gather(out->status, b + o_st, 1, 4));
    HIP_TRY(gather(out->iterations, b + o_it, 1, 4));
    HIP_TRY(gather(out->ls_evals, b + o_ls, 1, 4));
    HIP_TRY(gather(out->guess, b + o_guess, dim, es));
    HIP_TRY(gather(out->solution, b + o_sol, dim, es));
    HIP_TRY(hipEventRecord(sh.done, sh.stream));
    return CPMPC_OK;
  };
  for (int i = 0; i < n && rc == CPMPC_OK; ++i) rc = one_shard(i);
  if (rc != CPMPC_OK) {
    // copies of the shards already started are still writing the caller's arrays: wait for them before reporting
    for (int i = 0; i < started; ++i) {
      DeviceGuard guard(s->shards[i].device);
      (void)hipStreamSynchronize(s->shards[i].stream);
    }
    cpmpc_sharded_reset(s);  // some shards stepped, others did not
    return rc;
  }
  {
    DeviceGuard guard(root);
    for (int i = 0; i < n; ++i) {
      int64_t lo, hi;
      shard_range(B, i, n, &lo, &hi);
      if (hi > lo) HIP_TRY(hipStreamWaitEvent(root_stream, s->shards[i].done, 0));
    }
  }
  s->warm_total = B;
  return CPMPC_OK;
}

extern "C" int cpmpc_sharded_step_batch(cpmpc_sharded* s, int64_t B, const void* x0, const double* dyn_shared_host,
                                        double set_point, const cpmpc_step_outputs* out, void* stream) {
  if (!s || !x0 || !dyn_shared_host || !out) return fail(CPMPC_ERR_INVALID_ARG, "null argument");
  cpmpc_step_inputs in;
  memset(&in, 0, sizeof in);
  in.x0 = x0;
  in.dyn_shared_host = dyn_shared_host;
  in.set_point_shared = set_point;
  return cpmpc_sharded_step_batch_ex(s, B, &in, out, stream);
}

// Optimization::SetPreviousSolution over all shards (optimization.hpp:86-89): z is [dim][B] on the root device, in the
// handle's dtype.  Replaces whatever warm start the shards held.
extern "C" int cpmpc_sharded_set_previous_solution(cpmpc_sharded* s, int64_t B, const void* z, void* stream) {
  if (!s || !z) return fail(CPMPC_ERR_INVALID_ARG, "null argument");
  int rc = sharded_check(s, B);
  if (rc) return rc;
  cpmpc_sharded_reset(s);
  const int n = (int)s->shards.size();
  const size_t es = s->esize, dim = (size_t)s->dim;
  const int root = s->shards[0].device;
  {
    DeviceGuard guard(root);
    HIP_TRY(hipEventRecord(s->ready, (hipStream_t)stream));
  }
  int started = 0;
  for (int i = 0; i < n && rc == CPMPC_OK; ++i) {
    int64_t lo, hi;
    shard_range(B, i, n, &lo, &hi);
    if (hi == lo) continue;
    const size_t Bs = (size_t)(hi - lo);
    Shard& sh = s->shards[i];
    DeviceGuard guard(sh.device);
    rc = ensure_shard_buf(sh, dim * Bs * es);
    if (rc) break;
    hipError_t e = hipStreamWaitEvent(sh.stream, s->ready, 0);
    started = i + 1;
    if (e == hipSuccess)
      e = hipMemcpy2DAsync(sh.buf, Bs * es, (const char*)z + (size_t)lo * es, (size_t)B * es, Bs * es, dim, hipMemcpyDefault, sh.stream);
    if (e != hipSuccess) {
      rc = fail(CPMPC_ERR_HIP, "scatter of the previous solution failed: %s", hipGetErrorString(e));
      break;
    }
    rc = cpmpc_set_previous_solution(sh.h, (int64_t)Bs, sh.buf, sh.stream);
    if (rc == CPMPC_OK && hipEventRecord(sh.done, sh.stream) != hipSuccess) rc = fail(CPMPC_ERR_HIP, "hipEventRecord failed");
  }
  if (rc != CPMPC_OK) {
    for (int i = 0; i < started; ++i) {
      DeviceGuard guard(s->shards[i].device);
      (void)hipStreamSynchronize(s->shards[i].stream);
    }
    cpmpc_sharded_reset(s);
    return rc;
  }
  {  // the caller may reuse z once its stream passes this point
    DeviceGuard guard(root);
    for (int i = 0; i < n; ++i) {
      int64_t lo, hi;
      shard_range(B, i, n, &lo, &hi);
      if (hi > lo) HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, s->shards[i].done, 0));
    }
  }
  s->dist_B = B;
  s->warm_total = B;
  return CPMPC_OK;
}

// The warm start of columns [0, B), B <= cpmpc_sharded_previous_solution_batch(): z_out is [dim][B] on the root device.
extern "C" int cpmpc_sharded_get_solution(cpmpc_sharded* s, int64_t B, void* z_out, void* stream) {
  if (!s || !z_out) return fail(CPMPC_ERR_INVALID_ARG, "null argument");
  int rc = sharded_check(s, B);
  if (rc) return rc;
  if (B > s->warm_total)
    return fail(CPMPC_ERR_BATCH, "only %lld problems hold a previous solution, %lld asked for", (long long)s->warm_total, (long long)B);
  const int n = (int)s->shards.size();
  const size_t es = s->esize, dim = (size_t)s->dim;
  const int root = s->shards[0].device;
  {
    DeviceGuard guard(root);
    HIP_TRY(hipEventRecord(s->ready, (hipStream_t)stream));  // z_out is free for writing from here on
  }
  int started = 0;
  std::vector<int> used;
  for (int i = 0; i < n && rc == CPMPC_OK; ++i) {
    int64_t lo;
    const int64_t ni = warm_in_shard(s->dist_B, i, n, B, &lo);  // the shards' columns follow dist_B's split
    if (ni == 0) continue;
    Shard& sh = s->shards[i];
    DeviceGuard guard(sh.device);
    rc = ensure_shard_buf(sh, dim * (size_t)ni * es);
    if (rc) break;
    hipError_t e = hipStreamWaitEvent(sh.stream, s->ready, 0);
    started = i + 1;
    if (e != hipSuccess) {
      rc = fail(CPMPC_ERR_HIP, "hipStreamWaitEvent failed: %s", hipGetErrorString(e));
      break;
    }
    rc = cpmpc_get_solution(sh.h, ni, sh.buf, sh.stream);
    if (rc) break;
    e = hipMemcpy2DAsync((char*)z_out + (size_t)lo * es, (size_t)B * es, sh.buf, (size_t)ni * es, (size_t)ni * es, dim,
                         hipMemcpyDefault, sh.stream);
    if (e == hipSuccess) e = hipEventRecord(sh.done, sh.stream);
    if (e != hipSuccess) {
      rc = fail(CPMPC_ERR_HIP, "gather of the solution failed: %s", hipGetErrorString(e));
      break;
    }
    used.push_back(i);
  }
  if (rc != CPMPC_OK) {
    for (int i = 0; i < started; ++i) {
      DeviceGuard guard(s->shards[i].device);
      (void)hipStreamSynchronize(s->shards[i].stream);
    }
    return rc;
  }
  DeviceGuard guard(root);
  for (int i : used) HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, s->shards[i].done, 0));
  return CPMPC_OK;
}

extern "C" int cpmpc_sharded_set_previous_solution_host(cpmpc_sharded* s, int64_t B, const double* z_host) {
  if (!s || !z_host) return fail(CPMPC_ERR_INVALID_ARG, "null argument");
  int rc = sharded_check(s, B);
  if (rc) return rc;
  cpmpc_sharded_reset(s);
  const int n = (int)s->shards.size();
  for (int i = 0; i < n; ++i) {
    int64_t lo, hi;
    shard_range(B, i, n, &lo, &hi);
    if (hi == lo) continue;
    rc = set_prev_host_cols(s->shards[i].h, hi - lo, z_host, B, lo);
    if (rc) {
      cpmpc_sharded_reset(s);
      return rc;
    }
  }
  s->dist_B = B;
  s->warm_total = B;
  return CPMPC_OK;
}

extern "C" int cpmpc_sharded_get_solution_host(cpmpc_sharded* s, int64_t B, double* z_host) {
  if (!s || !z_host) return fail(CPMPC_ERR_INVALID_ARG, "null argument");
  int rc = sharded_check(s, B);
  if (rc) return rc;
  if (B > s->warm_total)
    return fail(CPMPC_ERR_BATCH, "only %lld problems hold a previous solution, %lld asked for", (long long)s->warm_total, (long long)B);
  const int n = (int)s->shards.size();
  for (int i = 0; i < n; ++i) {
    int64_t lo;
    const int64_t ni = warm_in_shard(s->dist_B, i, n, B, &lo);
    if (ni == 0) continue;
    rc = get_sol_host_cols(s->shards[i].h, ni, z_host, B, lo);
    if (rc) return rc;
  }
  return CPMPC_OK;
}

// ------------------------------------------------------------------------------------------------
// stand-alone pieces
// ------------------------------------------------------------------------------------------------
static int current_device_ok() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess)
    return fail(CPMPC_ERR_NO_DEVICE, "no HIP device visible; this library has no CPU fallback");
  return check_device(dev);
}

static int check_piece_args(int model, int dtype, int64_t B) {
  if (model != CPMPC_MODEL_SINGLE && model != CPMPC_MODEL_DOUBLE) return fail(CPMPC_ERR_INVALID_ARG, "unknown model");
  if (dtype != CPMPC_F32 && dtype != CPMPC_F64) return fail(CPMPC_ERR_INVALID_ARG, "bad dtype");
  if (B < 1) return fail(CPMPC_ERR_INVALID_ARG, "B must be >= 1");
  return current_device_ok();
}

extern "C" int cpmpc_dynamics_batch_model(int model, int dtype, int64_t B, const double* dyn_shared_host,
                                          const void* x, const void* u, const double* fext_host, void* f, void* Jx,
                                          void* Ju, void* stream) {
  if (!dyn_shared_host || !x || !u || !f) return fail(CPMPC_ERR_INVALID_ARG, "null argument");
  int rc = check_piece_args(model, dtype, B);
  if (rc) return rc;
  engine_for(dtype, model)->dynamics(B, dyn_shared_host, fext_host, x, u, f, Jx, Ju, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return CPMPC_OK;
}
extern "C" int cpmpc_dynamics_batch(int dtype, int64_t B, const double* dyn_shared_host, const void* x,
                                    const void* u, const double* fext_host, void* f, void* Jx, void* Ju,
                                    void* stream) {
  return cpmpc_dynamics_batch_model(CPMPC_MODEL_SINGLE, dtype, B, dyn_shared_host, x, u, fext_host, f, Jx, Ju, stream);
}

extern "C" int cpmpc_rk4_batch_model(int model, int dtype, int64_t B, const double* dyn_shared_host, const void* x,
                                     const void* u, double h, const double* fext_host, void* x_new, void* A,
                                     void* Bm, void* stream) {
  if (!dyn_shared_host || !x || !u || !x_new) return fail(CPMPC_ERR_INVALID_ARG, "null argument");
  int rc = check_piece_args(model, dtype, B);
  if (rc) return rc;
  engine_for(dtype, model)->rk4(B, dyn_shared_host, fext_host, h, x, u, x_new, A, Bm, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return CPMPC_OK;
}
extern "C" int cpmpc_rk4_batch(int dtype, int64_t B, const double* dyn_shared_host, const void* x, const void* u,
                               double h, const double* fext_host, void* x_new, void* A, void* Bm, void* stream) {
  return cpmpc_rk4_batch_model(CPMPC_MODEL_SINGLE, dtype, B, dyn_shared_host, x, u, h, fext_host, x_new, A, Bm,
                               stream);
}

extern "C" int cpmpc_linearize_batch(cpmpc_solver* s, int64_t B, const double* dyn_shared_host, const void* z,
                                     void* c, void* Phi, void* Gamma, void* stream) {
  if (!s || !dyn_shared_host || !z || !c || !Phi || !Gamma) return fail(CPMPC_ERR_INVALID_ARG, "null argument");
  if (B < 1) return fail(CPMPC_ERR_INVALID_ARG, "B must be >= 1");
  if (B > s->cap) return fail(CPMPC_ERR_BATCH, "B exceeds capacity");
  DeviceGuard guard(s->device);
  // the caller's z is packed into the step buffers (dzx/dzu), which hold no state between calls, so the
  // warm start (zx/zu) is untouched; the linearisation lands in the workspace and is unpacked
  engine_of(s)->linearize_batch(s, B, dyn_shared_host, z, c, Phi, Gamma, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  track_caller_stream(s, (hipStream_t)stream);  // the step buffers of the workspace were used on the caller's stream
  return CPMPC_OK;
}

extern "C" int cpmpc_sim_step_batch_model(int model, int dtype, int64_t B, const double* dyn_shared_host, double dt,
                                          const void* u, const double* fext_host, const void* fext, void* state,
                                          void* stream) {
  if (!dyn_shared_host || !u || !state) return fail(CPMPC_ERR_INVALID_ARG, "null argument");
  if (!(dt >= 0.0) || !std::isfinite(dt)) return fail(CPMPC_ERR_INVALID_ARG, "dt must be finite and >= 0 (simulator.cc:13)");
  int rc = check_piece_args(model, dtype, B);
  if (rc) return rc;
  // simulator.cc:18-22 evaluated in double: number of sub-steps and the size of the last one
  const double internal_dt = 0.001;
  int n_sub = 0;
  double h_last = internal_dt;
  {
    double rem = dt;
    while (rem > 0.0) {
      h_last = rem < internal_dt ? rem : internal_dt;
      ++n_sub;
      rem -= internal_dt;
      if (n_sub > 100000000) return fail(CPMPC_ERR_INVALID_ARG, "dt too large");
    }
  }
  if (n_sub == 0) return CPMPC_OK;
  engine_for(dtype, model)->sim(B, dyn_shared_host, fext_host, fext, n_sub, h_last, u, state, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return CPMPC_OK;
}
extern "C" int cpmpc_sim_step_batch(int dtype, int64_t B, const double* dyn_shared_host, double dt, const void* u,
                                    const double* fext_host, const void* fext, void* state, void* stream) {
  return cpmpc_sim_step_batch_model(CPMPC_MODEL_SINGLE, dtype, B, dyn_shared_host, dt, u, fext_host, fext, state,
                                    stream);
}

// Staging of the handle-less host-pointer plant step: per host thread and device, grown on demand and kept (a
// Simulator::Step per 10 ms tick must not allocate; simulator.cc:11-36 has no allocation either).
struct SimStage {
  int device = -1;
  void* dev = nullptr;
  void* pin = nullptr;
  size_t bytes = 0;
  hipStream_t stream = nullptr;
  // never freed: at thread/process exit the HIP runtime may already be gone (a few KB per calling thread)
};
static thread_local SimStage g_sim_stage;

static int ensure_sim_stage(size_t bytes) {
  SimStage& g = g_sim_stage;
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  if (g.device != dev) {
    if (g.dev) (void)hipFree(g.dev);
    if (g.pin) (void)hipHostFree(g.pin);
    if (g.stream) (void)hipStreamDestroy(g.stream);
    g.dev = g.pin = nullptr;
    g.stream = nullptr;
    g.bytes = 0;
    g.device = dev;
  }
  if (g.stream == nullptr) HIP_TRY(hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking));
  if (g.bytes >= bytes) return CPMPC_OK;
  if (g.dev) (void)hipFree(g.dev);
  if (g.pin) (void)hipHostFree(g.pin);
  g.dev = g.pin = nullptr;
  g.bytes = 0;
  const size_t want = bytes < 4096 ? 4096 : bytes;
  HIP_TRY(hipMalloc(&g.dev, want));
  hipError_t e = hipHostMalloc(&g.pin, want, hipHostMallocDefault);
  if (e != hipSuccess) {
    (void)hipFree(g.dev);
    g.dev = nullptr;
    return fail(CPMPC_ERR_ALLOC, "hipHostMalloc of %zu staging bytes failed: %s", want, hipGetErrorString(e));
  }
  g.bytes = want;
  return CPMPC_OK;
}

extern "C" int cpmpc_sim_step_batch_host(int64_t B, const double* dyn_shared_host, double dt, const double* u_host,
                                         const double* fext_host, double* state_host) {
  if (!dyn_shared_host || !u_host || !state_host) return fail(CPMPC_ERR_INVALID_ARG, "null argument");
  if (B < 1) return fail(CPMPC_ERR_INVALID_ARG, "B must be >= 1");
  for (int64_t i = 0; i < B; ++i)
    if (!std::isfinite(u_host[i])) return fail(CPMPC_ERR_INVALID_ARG, "u = %g is not finite (simulator.cc:14)", u_host[i]);
  int rc = current_device_ok();
  if (rc) return rc;
  const size_t nB = (size_t)B;
  rc = ensure_sim_stage(5 * nB * sizeof(double));
  if (rc) return rc;
  SimStage& g = g_sim_stage;
  // [state 4B | u B]: one copy in, the kernel, one copy out, one synchronisation
  double* h = (double*)g.pin;
  double* d = (double*)g.dev;
  memcpy(h, state_host, 4 * nB * sizeof(double));
  memcpy(h + 4 * nB, u_host, nB * sizeof(double));
  HIP_TRY(hipMemcpyAsync(d, h, 5 * nB * sizeof(double), hipMemcpyHostToDevice, g.stream));
  rc = cpmpc_sim_step_batch(CPMPC_F64, B, dyn_shared_host, dt, d + 4 * nB, fext_host, nullptr, d, g.stream);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(h, d, 4 * nB * sizeof(double), hipMemcpyDeviceToHost, g.stream));
  HIP_TRY(hipStreamSynchronize(g.stream));
  memcpy(state_host, h, 4 * nB * sizeof(double));
  return CPMPC_OK;
}

// debug builds only (-DCPMPC_FUSED_TIMING / -DCPMPC_FUSED_CLOCK): the counters of fused_sqp_kernel, summed over the
// kernel translation units (each has its own copies), read and cleared
static int debug_read_all(int which, unsigned long long* out, int n) {
  (void)hipDeviceSynchronize();
  for (int i = 0; i < n; ++i) out[i] = 0;
  int ok = -1;
  for (int dtype = 0; dtype < 2; ++dtype)
    for (int model = 0; model < 2; ++model)
      if (engine_for(dtype, model)->debug_read(which, out) == 0) ok = 0;
  return ok;
}
#ifdef CPMPC_FUSED_TIMING
extern "C" int cpmpc_debug_phase_cycles(unsigned long long* out8) { return debug_read_all(0, out8, 8); }
#endif
#ifdef CPMPC_FUSED_CLOCK
extern "C" int cpmpc_debug_kernel_clock(unsigned long long* out4) { return debug_read_all(1, out4, 4); }
#endif
