// cpmpc_api.hip -- C-ABI (include/cpmpc.h) over the gfx950 kernels in mpc_kernels.hpp.
//
// Host side of the batched cart-pole MPC hot path.  There is no CPU compute path in this library:
// every entry point that computes launches HIP kernels and fails with CPMPC_ERR_NO_DEVICE when no
// gfx950 device is usable.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/cpmpc.h"
#include "mpc_kernels.hpp"
#include "mpc_fused.hpp"

using namespace cpmpc;

// ------------------------------------------------------------------------------------------------
// error text
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(expr)                                                                       \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess)                                                                   \
      return fail(CPMPC_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),     \
                  __FILE__, __LINE__);                                                      \
  } while (0)

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

// run `body` with R and M bound to the (dtype, model) pair
#define CPMPC_DISPATCH(dtype, model, body)                \
  do {                                                    \
    if ((dtype) == CPMPC_F32 && (model) == CPMPC_MODEL_SINGLE) { \
      using R = float;                                    \
      using M = SingleModel<float>;                       \
      body;                                               \
    } else if ((dtype) == CPMPC_F64 && (model) == CPMPC_MODEL_SINGLE) { \
      using R = double;                                   \
      using M = SingleModel<double>;                      \
      body;                                               \
    } else if ((dtype) == CPMPC_F32) {                    \
      using R = float;                                    \
      using M = DoubleModel<float>;                       \
      body;                                               \
    } else {                                              \
      using R = double;                                   \
      using M = DoubleModel<double>;                      \
      body;                                               \
    }                                                     \
  } while (0)

// ------------------------------------------------------------------------------------------------
// the handle
// ------------------------------------------------------------------------------------------------
struct ProfSpan {
  int kernel;
  hipEvent_t start, stop;
};

struct cpmpc_solver {
  cpmpc_params params;
  cpmpc_solver_opts opts;
  int dtype;
  int model;
  int device;
  int64_t cap;  // workspace stride (capacity rounded up to a multiple of 64)
  int N, S, SP, NX, NP, dim;
  size_t esize;
  // one allocation, carved into fields
  void* ws = nullptr;
  size_t ws_bytes = 0;
  char *zx, *zu, *dzx, *dzu, *Phi, *Gam, *cs, *Wk, *Tk, *sc;
  int32_t* ist;
  void* sin_table = nullptr;
  int64_t prev_B = 0;  // problems [0, prev_B) hold a previous solution; Reset() -> 0
  // staging for the *_host entry points (lazily allocated, grown on demand, owned by the handle): a device buffer,
  // its pinned host mirror and a stream, so that a host-pointer call is one async copy in, the kernels, one async
  // copy out and a single synchronisation
  void* stage = nullptr;
  void* pin = nullptr;
  size_t stage_bytes = 0;
  hipStream_t hstream = nullptr;
  hipEvent_t ev_last = nullptr;  // end of the last device-pointer call on a caller's stream; hstream waits on it
  bool ev_pending = false;
  // profiling
  int profiling = 0;
  std::vector<ProfSpan> spans;
  std::vector<ProfSpan> free_spans;
  double prof_ms[CPMPC_KERNEL_COUNT] = {0, 0, 0, 0, 0};
  int64_t prof_n[CPMPC_KERNEL_COUNT] = {0, 0, 0, 0, 0};
  int pipeline = CPMPC_PIPELINE_AUTO;
  // staged fused pipeline (compaction of the still-active problems between stages); 0/0 = single launch
  int stage_first = 3, stage_next = 1;
  bool stage_auto = true;  // default: stage only batches larger than one round of resident waves
  int32_t* active = nullptr;  // [cap] compacted problem indices, then two counters
};

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

extern "C" int cpmpc_create_model(const cpmpc_params* params, const cpmpc_solver_opts* opts, int dtype,
                                  int64_t max_batch, int device, int model, cpmpc_solver** out) {
  if (!params || !out) return fail(CPMPC_ERR_INVALID_ARG, "null argument");
  *out = nullptr;
  if (dtype != CPMPC_F32 && dtype != CPMPC_F64) return fail(CPMPC_ERR_INVALID_ARG, "dtype must be CPMPC_F32 or CPMPC_F64");
  if (model != CPMPC_MODEL_SINGLE && model != CPMPC_MODEL_DOUBLE) return fail(CPMPC_ERR_INVALID_ARG, "unknown model");
  if (max_batch < 1 || max_batch > (1ll << 30)) return fail(CPMPC_ERR_INVALID_ARG, "max_batch must be in [1, 2^30]");
  int rc = validate_params(params);
  if (rc) return rc;
  rc = check_device(device);
  if (rc) return rc;
  DeviceGuard guard(device);

  cpmpc_solver* s = new (std::nothrow) cpmpc_solver();
  if (!s) return fail(CPMPC_ERR_ALLOC, "out of host memory");
  s->params = *params;
  if (opts)
    s->opts = *opts;
  else
    cpmpc_default_solver_opts(&s->opts);
  s->dtype = dtype;
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
  // index list of the staged fused pipeline (4 bytes per problem + two counters): allocated here, never in a step
  if (hipMalloc((void**)&s->active, ((size_t)s->cap + 2) * sizeof(int32_t)) != hipSuccess) {
    (void)hipGetLastError();
    s->active = nullptr;  // staging stays off for this handle (same results, single launch)
  }
  *out = s;
  return CPMPC_OK;
}

extern "C" int cpmpc_create(const cpmpc_params* params, const cpmpc_solver_opts* opts, int dtype,
                            int64_t max_batch, int device, cpmpc_solver** out) {
  return cpmpc_create_model(params, opts, dtype, max_batch, device, CPMPC_MODEL_SINGLE, out);
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
  if (s->stage) (void)hipFree(s->stage);
  if (s->pin) (void)hipHostFree(s->pin);
  if (s->hstream) (void)hipStreamDestroy(s->hstream);
  if (s->ev_last) (void)hipEventDestroy(s->ev_last);
  if (s->active) (void)hipFree(s->active);
  delete s;
}

extern "C" int cpmpc_dim(const cpmpc_solver* s) { return s ? s->dim : -1; }
extern "C" int cpmpc_num_states(const cpmpc_solver* s) { return s ? s->S : -1; }
extern "C" int cpmpc_dtype(const cpmpc_solver* s) { return s ? s->dtype : -1; }
extern "C" int cpmpc_model(const cpmpc_solver* s) { return s ? s->model : -1; }
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

static void span_begin(cpmpc_solver* s, int kernel, hipStream_t stream, ProfSpan* cur) {
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
static void span_end(cpmpc_solver* s, hipStream_t stream, ProfSpan* cur) {
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
  if (s->ev_last != nullptr && stream != s->hstream) {
    (void)hipEventRecord(s->ev_last, stream);
    s->ev_pending = true;
  }
}

// ------------------------------------------------------------------------------------------------
// launch helpers
// ------------------------------------------------------------------------------------------------
static inline dim3 grid_for(int64_t threads) { return dim3((unsigned)((threads + 63) / 64)); }

template <typename R, typename M>
static void fill_args(const cpmpc_solver* s, int64_t B, SolverArgs<R, M>& a) {
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
  a.prev_B = s->prev_B;
  using V4 = typename VecT<R>::V4;
  using XVn = XV<R, M::NX>;
  a.zx = (XVn*)s->zx;
  a.zu = (R*)s->zu;
  a.dzx = (XVn*)s->dzx;
  a.dzu = (R*)s->dzu;
  a.Phi = (XVn*)s->Phi;
  a.Gam = (XVn*)s->Gam;
  a.cs = (XVn*)s->cs;
  a.Wk = (XVn*)s->Wk;
  a.Tk = (V4*)s->Tk;
  a.sc = (R*)s->sc;
  a.ist = s->ist;
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

// fused pipeline: compiled specialisations for these (L = S-1, SP) pairs ...
static bool fused_static(int L, int SP) {  // the default horizon's spacings (N = 40) and N = 20
  return (L == 4 && SP == 10) || (L == 8 && SP == 5) || (L == 2 && SP == 10) || (L == 4 && SP == 5) ||
         (L == 2 && SP == 20) || (L == 5 && SP == 8) || (L == 10 && SP == 4);
}
// ... and a run-time-spacing variant (dynamic LDS) for any other spacing with one of these interval counts whose
// per-wave LDS (80 scalars per lane and control for NX = 4) fits the 64 KB a dynamic allocation may take
static size_t fused_dyn_bytes(const cpmpc_solver* s) {
  const size_t xw = s->NX > 4 ? 8 : 4;
  return (size_t)s->SP * 64 * (4 + xw) * s->esize;
}
static bool fused_dynamic(const cpmpc_solver* s) {
  const int L = s->S - 1;
  const bool l_ok = L == 2 || L == 4 || L == 5 || L == 8 || L == 10 || L == 16;
  return l_ok && fused_dyn_bytes(s) <= 65536;
}
static bool fused_built(const cpmpc_solver* s) { return fused_static(s->S - 1, s->SP) || fused_dynamic(s); }
static bool use_fused(const cpmpc_solver* s) {
  if (s->pipeline == CPMPC_PIPELINE_SPLIT) return false;
  if (!fused_built(s)) return false;
  // AUTO: the 6-state model in fp64 needs 61 KB of LDS per wave in the fused kernel (2 waves per CU); the split
  // pipeline is as fast there (measured 10.1 vs 9.7 M re-plans/s), so it stays the default for that case
  if (s->pipeline == CPMPC_PIPELINE_AUTO && s->model == CPMPC_MODEL_DOUBLE && s->dtype == CPMPC_F64) return false;
  return true;
}

// The shared-parameters specialisation (model constants wave-uniform, in SGPRs) is used in fp32 only: the fp64
// kernel already sits at the SGPR limit with its VGPR/AGPR file exhausted, and with the constants added to the
// scalar pressure hipcc 7.2 produced wrong results for it (caught by the fp64 parity tests); there the constants
// go through load_consts() into vector registers like per-problem parameters do.
template <typename R, typename M>
static void launch_fused(const SolverArgs<R, M>& a, int L, int SP, int max_iters, hipStream_t stream) {
  {
    const int ppw = 64 / L;
    const dim3 grid((unsigned)((a.B + ppw - 1) / ppw));
#define CPMPC_FUSED(LV, SPV)                                                                                \
  if (L == LV && SP == SPV) {                                                                               \
    if constexpr (sizeof(R) == 4 || CPMPC_FUSED_SHARED_F64) {                                              \
      if (a.dyn == nullptr) {                                                                               \
        hipLaunchKernelGGL((fused_sqp_kernel<R, M, SPV, LV, true>), grid, dim3(64), 0, stream, a, max_iters); \
        return;                                                                                             \
      }                                                                                                     \
    }                                                                                                       \
    hipLaunchKernelGGL((fused_sqp_kernel<R, M, SPV, LV, false>), grid, dim3(64), 0, stream, a, max_iters);  \
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
    if constexpr (sizeof(R) == 4 || CPMPC_FUSED_SHARED_F64) {                                                      \
      if (a.dyn == nullptr) {                                                                                       \
        hipLaunchKernelGGL((fused_sqp_dyn_kernel<R, M, LV, true>), grid, dim3(64), lds, stream, a, max_iters);      \
        return;                                                                                                     \
      }                                                                                                             \
    }                                                                                                               \
    hipLaunchKernelGGL((fused_sqp_dyn_kernel<R, M, LV, false>), grid, dim3(64), lds, stream, a, max_iters);         \
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
extern "C" int cpmpc_get_pipeline(const cpmpc_solver* s) {
  if (!s) return -1;
  return use_fused(s) ? CPMPC_PIPELINE_FUSED : CPMPC_PIPELINE_SPLIT;
}

template <typename R, typename M>
static int step_batch_impl(cpmpc_solver* s, int64_t B, const cpmpc_step_inputs* in, const cpmpc_step_outputs* out,
                           hipStream_t stream) {
  SolverArgs<R, M> a;
  fill_args<R, M>(s, B, a);
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

  span_begin(s, CPMPC_KERNEL_PREPARE, stream, &sp);
  hipLaunchKernelGGL((prepare_kernel<R, M>), dim3((unsigned)((B + CPMPC_PF_BLOCK - 1) / CPMPC_PF_BLOCK)), dim3(CPMPC_PF_BLOCK), 0, stream, a);
  span_end(s, stream, &sp);

  if (use_fused(s)) {
    // With exit tolerances enabled problems stop after different numbers of iterations, and a wave lives as long
    // as its slowest problem (closed loop, measured: 4.3 iterations per problem, 7.7 per wave of 16).  The kernel
    // is restartable -- all solver state is in the workspace between launches -- so it runs in stages and the
    // problems still iterating are compacted into dense waves in between.  Results are bitwise those of a single
    // launch: a problem's arithmetic does not depend on the lanes it occupies.
    const int total = (int)s->params.max_iterations;
    const bool exits = s->params.relative_exit_tol > 0.0 || s->params.absolute_first_derivative_tol > 0.0;
    bool staged = exits && s->stage_first > 0 && s->stage_next > 0 && total > s->stage_first;
    // a batch that fits the machine in one round of resident waves (2 per SIMD) ends with its slowest wave either
    // way: staging would only add launches
    if (s->stage_auto && (B * (int64_t)(s->S - 1) + 63) / 64 <= 2048) staged = false;
    if (s->active == nullptr) staged = false;  // no index list (allocation failed at creation): single launch, same results
    a.active_list = nullptr;
    a.active_count = nullptr;
    a.iter_cap = total;
    a.run_out_below = (int64_t)2048 * (64 / (s->S - 1));  // problems in one round of resident waves (2 per SIMD)
    span_begin(s, CPMPC_KERNEL_FUSED, stream, &sp);
    launch_fused<R, M>(a, s->S - 1, s->SP, staged ? s->stage_first : total, stream);
    span_end(s, stream, &sp);
    int stage = 0;
    for (int done = s->stage_first; staged && done < total; done += s->stage_next, ++stage) {
      int32_t* count = s->active + s->cap + (stage & 1);       // two counters: this compaction and the one before
      a.prev_count = stage ? s->active + s->cap + ((stage - 1) & 1) : nullptr;
      a.prev_total = B;
      a.remaining = total - done;
      span_begin(s, CPMPC_KERNEL_FUSED, stream, &sp);
      const hipError_t memset_rc = hipMemsetAsync(count, 0, sizeof(int32_t), stream);
      hipLaunchKernelGGL(compact_active_kernel, dim3((unsigned)((B + 1023) / 1024)), dim3(1024), 0, stream,
                         (const int32_t*)(s->ist + (size_t)IS_STATUS * (size_t)s->cap),
                         (const int32_t*)(s->ist + (size_t)IS_ITERS * (size_t)s->cap), total, B, s->active, count);
      a.active_list = s->active;
      a.active_count = count;
      const int k = (total - done < s->stage_next) ? (total - done) : s->stage_next;
      launch_fused<R, M>(a, s->S - 1, s->SP, k, stream);
      span_end(s, stream, &sp);  // the span is closed (its events recycled) before any early return
      if (memset_rc != hipSuccess) {
        if (B > s->prev_B) s->prev_B = B;  // prepare has already shifted the warm start: keep the handle consistent
        return fail(CPMPC_ERR_HIP, "hipMemsetAsync failed: %s", hipGetErrorString(memset_rc));
      }
    }
  } else {
    for (int it = 0; it < (int)s->params.max_iterations; ++it) {
      span_begin(s, CPMPC_KERNEL_LINEARIZE, stream, &sp);
      launch_linearize<R, M>(a, s->SP, a.zx, a.zu, a.ist, stream);
      span_end(s, stream, &sp);
      span_begin(s, CPMPC_KERNEL_QP_LS, stream, &sp);
      hipLaunchKernelGGL((qp_ls_kernel<R, M>), gridB, dim3(64), 0, stream, a);
      span_end(s, stream, &sp);
    }
  }

  span_begin(s, CPMPC_KERNEL_FINALIZE, stream, &sp);
  hipLaunchKernelGGL((finalize_kernel<R, M>), dim3((unsigned)((B + CPMPC_PF_BLOCK - 1) / CPMPC_PF_BLOCK)), dim3(CPMPC_PF_BLOCK), 0, stream, a);
  span_end(s, stream, &sp);

  HIP_TRY(hipGetLastError());
  if (B > s->prev_B) s->prev_B = B;  // previous_solution_ = solver_->variables()  (optimization.cc:85), per problem
  return CPMPC_OK;
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
  CPMPC_DISPATCH(s->dtype, s->model, (rc = step_batch_impl<R, M>(s, B, in, out, (hipStream_t)stream)));
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
  CPMPC_DISPATCH(s->dtype, s->model,
                 hipLaunchKernelGGL((pack_z_kernel<R, M::NX>), grid_for(B), dim3(64), 0, (hipStream_t)stream, B,
                                    s->cap, s->S, s->N, (const R*)z, (XV<R, M::NX>*)s->zx, (R*)s->zu));
  HIP_TRY(hipGetLastError());
  if (B > s->prev_B) s->prev_B = B;
  track_caller_stream(s, (hipStream_t)stream);
  return CPMPC_OK;
}

extern "C" int cpmpc_get_solution(cpmpc_solver* s, int64_t B, void* z_out, void* stream) {
  if (!s || !z_out) return fail(CPMPC_ERR_INVALID_ARG, "null argument");
  if (B < 1 || B > s->cap) return fail(CPMPC_ERR_BATCH, "B out of range");
  DeviceGuard guard(s->device);
  CPMPC_DISPATCH(s->dtype, s->model,
                 hipLaunchKernelGGL((unpack_z_kernel<R, M::NX>), grid_for(B), dim3(64), 0, (hipStream_t)stream, B,
                                    s->cap, s->S, s->N, (const XV<R, M::NX>*)s->zx, (const R*)s->zu, (R*)z_out));
  HIP_TRY(hipGetLastError());
  // this read of zx/zu on the caller's stream must finish before a later host-pointer call (on the handle's own
  // stream) overwrites them
  track_caller_stream(s, (hipStream_t)stream);
  return CPMPC_OK;
}

// ------------------------------------------------------------------------------------------------
// host-pointer convenience (staging copies around the same GPU path)
// ------------------------------------------------------------------------------------------------
static int ensure_stage(cpmpc_solver* s, size_t bytes) {
  if (s->hstream == nullptr) {
    hipError_t e = hipStreamCreateWithFlags(&s->hstream, hipStreamNonBlocking);
    if (e != hipSuccess) return fail(CPMPC_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e));
    e = hipEventCreateWithFlags(&s->ev_last, hipEventDisableTiming);
    if (e != hipSuccess) return fail(CPMPC_ERR_HIP, "hipEventCreate failed: %s", hipGetErrorString(e));
    // device-pointer calls made before this first host-pointer call were not tracked by ev_last: order after them once
    HIP_TRY(hipDeviceSynchronize());
  }
  if (s->ev_pending) {  // a device-pointer call on a caller's stream came in between: run after it
    HIP_TRY(hipStreamWaitEvent(s->hstream, s->ev_last, 0));
    s->ev_pending = false;
  }
  if (s->stage_bytes >= bytes) return CPMPC_OK;
  if (s->stage) (void)hipFree(s->stage);
  if (s->pin) (void)hipHostFree(s->pin);
  s->stage = nullptr;
  s->pin = nullptr;
  s->stage_bytes = 0;
  const size_t want = bytes < 4096 ? 4096 : bytes;
  hipError_t e = hipMalloc(&s->stage, want);
  if (e != hipSuccess) return fail(CPMPC_ERR_ALLOC, "hipMalloc of %zu staging bytes failed: %s", want, hipGetErrorString(e));
  e = hipHostMalloc(&s->pin, want, hipHostMallocDefault);
  if (e != hipSuccess) {
    (void)hipFree(s->stage);
    s->stage = nullptr;
    return fail(CPMPC_ERR_ALLOC, "hipHostMalloc of %zu staging bytes failed: %s", want, hipGetErrorString(e));
  }
  s->stage_bytes = want;
  return CPMPC_OK;
}

// A host-pointer step in two halves, so that several handles (the shards of cpmpc_sharded_*) can have their copies and
// kernels in flight together: `begin` converts and uploads the inputs and queues the kernels and the copy back on the
// handle's own stream; `end` waits for that stream and scatters the results into the caller's arrays.  The caller's
// arrays are [field][ld] with this handle's B problems at columns [col0, col0 + B): ld = B, col0 = 0 for a plain call.
struct HostStepLayout {
  size_t off_u = 0, off_cost = 0, off_eq = 0, off_status = 0, off_iters = 0, off_sol = 0, off_pred = 0;  // bytes
};

template <typename R, typename M>
static HostStepLayout host_step_layout(const cpmpc_solver* s, int64_t B) {
  // staging layout, identical on the device and in the pinned mirror:
  //   [x0 | u | cost | eq | status | iters | solution | predicted]      (the optional tails last: one copy back)
  const size_t nB = (size_t)B;
  HostStepLayout L;
  L.off_u = (size_t)M::NX * nB * sizeof(R);
  L.off_cost = L.off_u + (size_t)s->N * nB * sizeof(R);
  L.off_eq = L.off_cost + nB * sizeof(R);
  L.off_status = L.off_eq + nB * sizeof(R);
  L.off_iters = L.off_status + nB * sizeof(int32_t);
  L.off_sol = (L.off_iters + nB * sizeof(int32_t) + 7) & ~(size_t)7;  // the real-typed tail starts 8-byte aligned
  L.off_pred = L.off_sol + (size_t)s->dim * nB * sizeof(R);
  return L;
}

template <typename R, typename M>
static int step_host_begin(cpmpc_solver* s, int64_t B, const double* x0_host, int64_t ld, int64_t col0,
                           const double* dyn_shared_host, double set_point, bool want_pred, bool want_sol) {
  const size_t nB = (size_t)B;
  const size_t n_pred = (size_t)M::NX * (size_t)s->N * nB;
  const HostStepLayout L = host_step_layout<R, M>(s, B);
  const size_t bytes = L.off_pred + n_pred * sizeof(R) + 64;
  int rc = ensure_stage(s, bytes);
  if (rc) return rc;
  char* d_base = (char*)s->stage;
  R* d_x0 = (R*)d_base;
  R* h_x0 = (R*)s->pin;
  const hipStream_t st = s->hstream;
  // from here on work is in flight on `st` that reads the pinned mirror and writes the staging buffer: every early
  // return drains the stream first, so that the next call never reuses them under a running copy
  auto bail = [&](int code) {
    (void)hipStreamSynchronize(st);
    return code;
  };
  for (int t = 0; t < M::NX; ++t) {
    const double* src = x0_host + (size_t)t * (size_t)ld + (size_t)col0;
    R* dst = h_x0 + (size_t)t * nB;
    for (size_t i = 0; i < nB; ++i) dst[i] = (R)src[i];
  }
  hipError_t e = hipMemcpyAsync(d_x0, h_x0, L.off_u, hipMemcpyHostToDevice, st);
  if (e != hipSuccess) return bail(fail(CPMPC_ERR_HIP, "hipMemcpyAsync (inputs) failed: %s", hipGetErrorString(e)));

  cpmpc_step_inputs in;
  memset(&in, 0, sizeof in);
  in.x0 = d_x0;
  in.dyn_shared_host = dyn_shared_host;
  in.set_point_shared = set_point;
  cpmpc_step_outputs out;
  memset(&out, 0, sizeof out);
  out.u = d_base + L.off_u;
  out.predicted = want_pred ? d_base + L.off_pred : nullptr;
  out.status = (int32_t*)(d_base + L.off_status);
  out.iterations = (int32_t*)(d_base + L.off_iters);
  out.final_cost = d_base + L.off_cost;
  out.final_eq_l1 = d_base + L.off_eq;
  out.solution = want_sol ? d_base + L.off_sol : nullptr;
  rc = step_batch_impl<R, M>(s, B, &in, &out, st);
  if (rc) return bail(rc);
  // one copy back, from u to the end of what was asked for
  const size_t end = want_pred ? L.off_pred + n_pred * sizeof(R)
                               : (want_sol ? L.off_pred : L.off_iters + nB * sizeof(int32_t));
  e = hipMemcpyAsync((char*)s->pin + L.off_u, d_base + L.off_u, end - L.off_u, hipMemcpyDeviceToHost, st);
  if (e != hipSuccess) return bail(fail(CPMPC_ERR_HIP, "hipMemcpyAsync (outputs) failed: %s", hipGetErrorString(e)));
  return CPMPC_OK;
}

template <typename R, typename M>
static int step_host_end(cpmpc_solver* s, int64_t B, const cpmpc_step_host_outputs& ho, int64_t ld, int64_t col0) {
  HIP_TRY(hipStreamSynchronize(s->hstream));
  const size_t nB = (size_t)B;
  const HostStepLayout L = host_step_layout<R, M>(s, B);
  const char* h_base = (const char*)s->pin;
  // rows of B scalars in the mirror -> rows of ld scalars in the caller's array, at column col0
  auto fetch = [&](size_t off, double* hdst, size_t rows) {
    if (!hdst) return;
    const R* h = (const R*)(h_base + off);
    for (size_t r = 0; r < rows; ++r) {
      double* dst = hdst + r * (size_t)ld + (size_t)col0;
      const R* src = h + r * nB;
      for (size_t i = 0; i < nB; ++i) dst[i] = (double)src[i];
    }
  };
  fetch(L.off_u, ho.u, (size_t)s->N);
  fetch(L.off_cost, ho.final_cost, 1);
  fetch(L.off_eq, ho.final_eq_l1, 1);
  fetch(L.off_sol, ho.solution, (size_t)s->dim);
  fetch(L.off_pred, ho.predicted, (size_t)M::NX * (size_t)s->N);
  if (ho.status) memcpy(ho.status + col0, h_base + L.off_status, nB * sizeof(int32_t));
  if (ho.iterations) memcpy(ho.iterations + col0, h_base + L.off_iters, nB * sizeof(int32_t));
  return CPMPC_OK;
}

template <typename R, typename M>
static int step_host_impl(cpmpc_solver* s, int64_t B, const double* x0_host, const double* dyn_shared_host,
                          double set_point, const cpmpc_step_host_outputs& ho) {
  const int rc = step_host_begin<R, M>(s, B, x0_host, B, 0, dyn_shared_host, set_point, ho.predicted != nullptr,
                                       ho.solution != nullptr);
  if (rc) return rc;
  return step_host_end<R, M>(s, B, ho, B, 0);
}

extern "C" int cpmpc_step_batch_host_ex(cpmpc_solver* s, int64_t B, const double* x0_host,
                                        const double* dyn_shared_host, double set_point,
                                        const cpmpc_step_host_outputs* out) {
  if (!s || !x0_host || !dyn_shared_host || !out) return fail(CPMPC_ERR_INVALID_ARG, "null argument");
  if (B < 1) return fail(CPMPC_ERR_INVALID_ARG, "B must be >= 1");
  if (B > s->cap) return fail(CPMPC_ERR_BATCH, "B exceeds capacity");
  if (!std::isfinite(set_point)) return fail(CPMPC_ERR_INVALID_ARG, "set_point must be finite");
  DeviceGuard guard(s->device);
  int rc = CPMPC_OK;
  CPMPC_DISPATCH(s->dtype, s->model, (rc = step_host_impl<R, M>(s, B, x0_host, dyn_shared_host, set_point, *out)));
  return rc;
}

extern "C" int cpmpc_step_batch_host(cpmpc_solver* s, int64_t B, const double* x0_host,
                                     const double* dyn_shared_host, double set_point, double* u_host,
                                     double* predicted_host, int32_t* status_host, int32_t* iterations_host,
                                     double* final_cost_host, double* final_eq_l1_host) {
  const cpmpc_step_host_outputs out = {u_host, predicted_host, status_host, iterations_host, final_cost_host,
                                       final_eq_l1_host, nullptr};
  return cpmpc_step_batch_host_ex(s, B, x0_host, dyn_shared_host, set_point, &out);
}

extern "C" int cpmpc_set_previous_solution_host(cpmpc_solver* s, int64_t B, const double* z_host) {
  if (!s || !z_host) return fail(CPMPC_ERR_INVALID_ARG, "null argument");
  if (B < 1 || B > s->cap) return fail(CPMPC_ERR_BATCH, "B out of range");
  DeviceGuard guard(s->device);
  const size_t n = (size_t)s->dim * (size_t)B;
  int rc = ensure_stage(s, n * s->esize);
  if (rc) return rc;
  if (s->dtype == CPMPC_F32) {
    float* h = (float*)s->pin;
    for (size_t i = 0; i < n; ++i) h[i] = (float)z_host[i];
  } else {
    memcpy(s->pin, z_host, n * 8);
  }
  HIP_TRY(hipMemcpyAsync(s->stage, s->pin, n * s->esize, hipMemcpyHostToDevice, s->hstream));
  rc = cpmpc_set_previous_solution(s, B, s->stage, s->hstream);
  if (rc) {
    (void)hipStreamSynchronize(s->hstream);  // the copy above still reads the pinned mirror
    return rc;
  }
  HIP_TRY(hipStreamSynchronize(s->hstream));
  return CPMPC_OK;
}

extern "C" int cpmpc_get_solution_host(cpmpc_solver* s, int64_t B, double* z_host) {
  if (!s || !z_host) return fail(CPMPC_ERR_INVALID_ARG, "null argument");
  if (B < 1 || B > s->cap) return fail(CPMPC_ERR_BATCH, "B out of range");
  DeviceGuard guard(s->device);
  const size_t n = (size_t)s->dim * (size_t)B;
  int rc = ensure_stage(s, n * s->esize);
  if (rc) return rc;
  rc = cpmpc_get_solution(s, B, s->stage, s->hstream);
  if (rc) return rc;
  {
    const hipError_t e = hipMemcpyAsync(s->pin, s->stage, n * s->esize, hipMemcpyDeviceToHost, s->hstream);
    const hipError_t e2 = hipStreamSynchronize(s->hstream);  // also on failure: the unpack kernel is in flight
    if (e != hipSuccess) return fail(CPMPC_ERR_HIP, "hipMemcpyAsync failed: %s", hipGetErrorString(e));
    if (e2 != hipSuccess) return fail(CPMPC_ERR_HIP, "hipStreamSynchronize failed: %s", hipGetErrorString(e2));
  }
  if (s->dtype == CPMPC_F32) {
    const float* h = (const float*)s->pin;
    for (size_t i = 0; i < n; ++i) z_host[i] = (double)h[i];
  } else {
    memcpy(z_host, s->pin, n * 8);
  }
  return CPMPC_OK;
}

// ------------------------------------------------------------------------------------------------
// several GPUs from one process: one handle + stream per shard, contiguous split, concurrent shards
// ------------------------------------------------------------------------------------------------
struct Shard {
  cpmpc_solver* h = nullptr;
  int device = 0;
  hipStream_t stream = nullptr;   // device-pointer steps of this shard run here
  hipEvent_t ready = nullptr;     // root stream -> shard stream (inputs are there)
  hipEvent_t done = nullptr;      // shard stream -> root stream (results have landed on the root device)
  void* buf = nullptr;            // per-shard device staging: [x0 | u | predicted | cost | eq | status | iters]
  size_t buf_bytes = 0;
};

struct cpmpc_sharded {
  std::vector<Shard> shards;
  int dtype = CPMPC_F64;
  int N = 0, NX = 4;
  size_t esize = 8;
  int64_t cap = 0;
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
    if (sh.ready) (void)hipEventDestroy(sh.ready);
    if (sh.done) (void)hipEventDestroy(sh.done);
    if (sh.stream) (void)hipStreamDestroy(sh.stream);
  }
  delete s;
}

extern "C" int cpmpc_sharded_create(const cpmpc_params* params, const cpmpc_solver_opts* opts, int dtype,
                                    int64_t max_batch, const int* devices, int n_devices, cpmpc_sharded** out) {
  if (!params || !out) return fail(CPMPC_ERR_INVALID_ARG, "null argument");
  *out = nullptr;
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
  if (max_batch < n) return fail(CPMPC_ERR_INVALID_ARG, "max_batch must be at least the number of shards");
  cpmpc_sharded* s = new (std::nothrow) cpmpc_sharded();
  if (!s) return fail(CPMPC_ERR_ALLOC, "out of host memory");
  s->dtype = dtype;
  s->esize = dtype == CPMPC_F32 ? 4 : 8;
  s->cap = max_batch;
  s->shards.resize(n);
  for (int i = 0; i < n; ++i) {
    Shard& sh = s->shards[i];
    sh.device = devs[i];
    int64_t lo, hi;
    shard_range(max_batch, i, n, &lo, &hi);
    int rc = cpmpc_create(params, opts, dtype, hi - lo + 1, sh.device, &sh.h);  // +1: a smaller B may shift a remainder here
    if (rc == CPMPC_OK) {
      DeviceGuard guard(sh.device);
      if (hipStreamCreateWithFlags(&sh.stream, hipStreamNonBlocking) != hipSuccess ||
          hipEventCreateWithFlags(&sh.ready, hipEventDisableTiming) != hipSuccess ||
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
  // peer access between the root device and every other shard's device (both directions); a pair that cannot be
  // mapped still works, the copies then go through host memory
  const int root = s->shards[0].device;
  for (int i = 1; i < n; ++i) {
    const int d = s->shards[i].device;
    if (d == root) continue;
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, root, d) == hipSuccess && can) {
      DeviceGuard guard(root);
      (void)hipDeviceEnablePeerAccess(d, 0);
    }
    if (hipDeviceCanAccessPeer(&can, d, root) == hipSuccess && can) {
      DeviceGuard guard(d);
      (void)hipDeviceEnablePeerAccess(root, 0);
    }
    (void)hipGetLastError();  // "already enabled" is fine
  }
  *out = s;
  return CPMPC_OK;
}

extern "C" int cpmpc_sharded_num_shards(const cpmpc_sharded* s) { return s ? (int)s->shards.size() : -1; }
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
  return CPMPC_OK;
}

static int sharded_check(const cpmpc_sharded* s, int64_t B) {
  if (B < 1) return fail(CPMPC_ERR_INVALID_ARG, "B must be >= 1");
  if (B > s->cap) return fail(CPMPC_ERR_BATCH, "B=%lld exceeds the capacity %lld given to cpmpc_sharded_create", (long long)B, (long long)s->cap);
  return CPMPC_OK;
}

extern "C" int cpmpc_sharded_step_batch_host(cpmpc_sharded* s, int64_t B, const double* x0_host,
                                             const double* dyn_shared_host, double set_point,
                                             const cpmpc_step_host_outputs* out) {
  if (!s || !x0_host || !dyn_shared_host || !out) return fail(CPMPC_ERR_INVALID_ARG, "null argument");
  if (!std::isfinite(set_point)) return fail(CPMPC_ERR_INVALID_ARG, "set_point must be finite");
  int rc = sharded_check(s, B);
  if (rc) return rc;
  const int n = (int)s->shards.size();
  // every shard's upload, kernels and download are queued on its own stream before anybody is waited for
  int begun = 0;
  for (int i = 0; i < n && rc == CPMPC_OK; ++i) {
    int64_t lo, hi;
    shard_range(B, i, n, &lo, &hi);
    if (hi == lo) continue;
    Shard& sh = s->shards[i];
    DeviceGuard guard(sh.device);
    CPMPC_DISPATCH(sh.h->dtype, sh.h->model,
                   (rc = step_host_begin<R, M>(sh.h, hi - lo, x0_host, B, lo, dyn_shared_host, set_point,
                                               out->predicted != nullptr, out->solution != nullptr)));
    if (rc == CPMPC_OK) begun = i + 1;
  }
  int first_rc = rc;
  for (int i = 0; i < begun; ++i) {   // also after a failure: drain what was started
    int64_t lo, hi;
    shard_range(B, i, n, &lo, &hi);
    if (hi == lo) continue;
    Shard& sh = s->shards[i];
    DeviceGuard guard(sh.device);
    int rc_i = CPMPC_OK;
    if (first_rc == CPMPC_OK) {
      CPMPC_DISPATCH(sh.h->dtype, sh.h->model, (rc_i = step_host_end<R, M>(sh.h, hi - lo, *out, B, lo)));
      if (rc_i != CPMPC_OK) first_rc = rc_i;
    } else {
      (void)hipStreamSynchronize(sh.h->hstream);
    }
  }
  return first_rc;
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

extern "C" int cpmpc_sharded_step_batch(cpmpc_sharded* s, int64_t B, const void* x0, const double* dyn_shared_host,
                                        double set_point, const cpmpc_step_outputs* out, void* stream) {
  if (!s || !x0 || !dyn_shared_host || !out) return fail(CPMPC_ERR_INVALID_ARG, "null argument");
  if (!std::isfinite(set_point)) return fail(CPMPC_ERR_INVALID_ARG, "set_point must be finite");
  if (out->guess || out->solution || out->ls_evals)
    return fail(CPMPC_ERR_UNSUPPORTED, "guess / solution / ls_evals are not gathered by the sharded step");
  int rc = sharded_check(s, B);
  if (rc) return rc;
  const int n = (int)s->shards.size();
  const size_t es = s->esize;
  const hipStream_t root_stream = (hipStream_t)stream;
  const int root = s->shards[0].device;
  const size_t NX = (size_t)s->NX, N = (size_t)s->N;
  // the caller's inputs are ready where its stream is now
  {
    DeviceGuard guard(root);
    for (int i = 0; i < n; ++i) HIP_TRY(hipEventRecord(s->shards[i].ready, root_stream));
  }
  for (int i = 0; i < n; ++i) {
    int64_t lo, hi;
    shard_range(B, i, n, &lo, &hi);
    if (hi == lo) continue;
    const size_t Bs = (size_t)(hi - lo);
    Shard& sh = s->shards[i];
    DeviceGuard guard(sh.device);
    // per-shard staging on the shard's own device, 256-byte aligned pieces
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_x0 = 0, o_u = al(NX * Bs * es), o_pred = o_u + al(N * Bs * es), o_cost = o_pred + al(N * NX * Bs * es),
                 o_eq = o_cost + al(Bs * es), o_st = o_eq + al(Bs * es), o_it = o_st + al(Bs * 4), o_end = o_it + al(Bs * 4);
    rc = ensure_shard_buf(sh, o_end);
    if (rc) return rc;
    char* b = (char*)sh.buf;
    HIP_TRY(hipStreamWaitEvent(sh.stream, sh.ready, 0));
    // scatter: my columns of x0 [NX][B] (root device) -> [NX][Bs] here
    HIP_TRY(hipMemcpy2DAsync(b + o_x0, Bs * es, (const char*)x0 + (size_t)lo * es, (size_t)B * es, Bs * es, NX,
                             hipMemcpyDefault, sh.stream));
    cpmpc_step_inputs in;
    memset(&in, 0, sizeof in);
    in.x0 = b + o_x0;
    in.dyn_shared_host = dyn_shared_host;
    in.set_point_shared = set_point;
    cpmpc_step_outputs o;
    memset(&o, 0, sizeof o);
    o.u = out->u ? b + o_u : nullptr;
    o.predicted = out->predicted ? b + o_pred : nullptr;
    o.final_cost = out->final_cost ? b + o_cost : nullptr;
    o.final_eq_l1 = out->final_eq_l1 ? b + o_eq : nullptr;
    o.status = out->status ? (int32_t*)(b + o_st) : nullptr;
    o.iterations = out->iterations ? (int32_t*)(b + o_it) : nullptr;
    rc = cpmpc_step_batch(sh.h, (int64_t)Bs, &in, &o, sh.stream);
    if (rc) return rc;
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
    HIP_TRY(hipEventRecord(sh.done, sh.stream));
  }
  {
    DeviceGuard guard(root);
    for (int i = 0; i < n; ++i) {
      int64_t lo, hi;
      shard_range(B, i, n, &lo, &hi);
      if (hi > lo) HIP_TRY(hipStreamWaitEvent(root_stream, s->shards[i].done, 0));
    }
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
  CPMPC_DISPATCH(dtype, model,
                 hipLaunchKernelGGL((dynamics_kernel<R, M>), grid_for(B), dim3(64), 0, (hipStream_t)stream, B,
                                    M::template make<double>(dyn_shared_host), ext_from_host<R>(fext_host),
                                    (const R*)x, (const R*)u, (R*)f, (R*)Jx, (R*)Ju));
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
  CPMPC_DISPATCH(dtype, model,
                 hipLaunchKernelGGL((rk4_kernel<R, M>), grid_for(B), dim3(64), 0, (hipStream_t)stream, B,
                                    M::template make<double>(dyn_shared_host), ext_from_host<R>(fext_host), (R)h,
                                    (const R*)x, (const R*)u, (R*)x_new, (R*)A, (R*)Bm));
  HIP_TRY(hipGetLastError());
  return CPMPC_OK;
}
extern "C" int cpmpc_rk4_batch(int dtype, int64_t B, const double* dyn_shared_host, const void* x, const void* u,
                               double h, const double* fext_host, void* x_new, void* A, void* Bm, void* stream) {
  return cpmpc_rk4_batch_model(CPMPC_MODEL_SINGLE, dtype, B, dyn_shared_host, x, u, h, fext_host, x_new, A, Bm,
                               stream);
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

extern "C" int cpmpc_linearize_batch(cpmpc_solver* s, int64_t B, const double* dyn_shared_host, const void* z,
                                     void* c, void* Phi, void* Gamma, void* stream) {
  if (!s || !dyn_shared_host || !z || !c || !Phi || !Gamma) return fail(CPMPC_ERR_INVALID_ARG, "null argument");
  if (B < 1) return fail(CPMPC_ERR_INVALID_ARG, "B must be >= 1");
  if (B > s->cap) return fail(CPMPC_ERR_BATCH, "B exceeds capacity");
  DeviceGuard guard(s->device);
  // the caller's z is packed into the step buffers (dzx/dzu), which hold no state between calls, so the
  // warm start (zx/zu) is untouched; the linearisation lands in the workspace and is unpacked
  CPMPC_DISPATCH(s->dtype, s->model,
                 (linearize_batch_impl<R, M>(s, B, dyn_shared_host, z, c, Phi, Gamma, (hipStream_t)stream)));
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
  CPMPC_DISPATCH(dtype, model,
                 hipLaunchKernelGGL((sim_kernel<R, M>), grid_for(B), dim3(64), 0, (hipStream_t)stream, B,
                                    M::template make<double>(dyn_shared_host), ext_from_host<R>(fext_host),
                                    (const R*)fext, n_sub, (R)h_last, (const R*)u, (R*)state));
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

#ifdef CPMPC_FUSED_TIMING
// debug build only: read and clear the per-phase cycle counters of fused_sqp_kernel
extern "C" int cpmpc_debug_phase_cycles(unsigned long long* out8) {
  hipDeviceSynchronize();
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(cpmpc::g_fused_phase_cycles), 8 * sizeof(unsigned long long)) != hipSuccess)
    return -1;
  unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  hipMemcpyToSymbol(HIP_SYMBOL(cpmpc::g_fused_phase_cycles), zero, sizeof(zero));
  return 0;
}
#endif

#ifdef CPMPC_FUSED_CLOCK
// debug build only: read and clear {sum of shader cycles, sum of 100 MHz ticks, waves, max cycles of a wave} of the
// fused_sqp_kernel launches since the last call
extern "C" int cpmpc_debug_kernel_clock(unsigned long long* out4) {
  hipDeviceSynchronize();
  if (hipMemcpyFromSymbol(out4, HIP_SYMBOL(cpmpc::g_fused_clock), 4 * sizeof(unsigned long long)) != hipSuccess)
    return -1;
  unsigned long long zero[4] = {0, 0, 0, 0};
  hipMemcpyToSymbol(HIP_SYMBOL(cpmpc::g_fused_clock), zero, sizeof(zero));
  return 0;
}
#endif
