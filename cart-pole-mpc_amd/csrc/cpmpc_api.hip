// cpmpc_api.hip -- C-ABI (include/cpmpc.h) over the gfx950 kernels.
//
// Host side of the batched cart-pole MPC hot path.  There is no CPU compute path in this library:
// every entry point that computes launches HIP kernels and fails with CPMPC_ERR_NO_DEVICE when no
// gfx950 device is usable.  This unit holds no device code: the kernels of each (dtype, model) pair live in their own
// translation unit (engine_<dtype>_<model>.hip) and are reached through its `Engine` table (engine.hpp).
#include <hip/hip_runtime.h>

#include <cmath>
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
  if (s->ev_last != nullptr && stream != s->hstream) {
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
  rc = engine_of(s)->step_batch(s, B, in, out, (hipStream_t)stream);
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
// host-pointer convenience (staging copies around the same GPU path)
// ------------------------------------------------------------------------------------------------
int ensure_stage(cpmpc_solver* s, size_t bytes) {
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

extern "C" int cpmpc_step_batch_host_ex(cpmpc_solver* s, int64_t B, const double* x0_host,
                                        const double* dyn_shared_host, double set_point,
                                        const cpmpc_step_host_outputs* out) {
  if (!s || !x0_host || !dyn_shared_host || !out) return fail(CPMPC_ERR_INVALID_ARG, "null argument");
  if (B < 1) return fail(CPMPC_ERR_INVALID_ARG, "B must be >= 1");
  if (B > s->cap) return fail(CPMPC_ERR_BATCH, "B exceeds capacity");
  if (!std::isfinite(set_point)) return fail(CPMPC_ERR_INVALID_ARG, "set_point must be finite");
  DeviceGuard guard(s->device);
  int rc = CPMPC_OK;
  const Engine* e = engine_of(s);
  rc = e->step_host_begin(s, B, x0_host, B, 0, dyn_shared_host, set_point, out->predicted != nullptr, out->solution != nullptr);
  if (rc == CPMPC_OK) rc = e->step_host_end(s, B, *out, B, 0);
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
    rc = engine_of(sh.h)->step_host_begin(sh.h, hi - lo, x0_host, B, lo, dyn_shared_host, set_point,
                                          out->predicted != nullptr, out->solution != nullptr);
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
      rc_i = engine_of(sh.h)->step_host_end(sh.h, hi - lo, *out, B, lo);
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
