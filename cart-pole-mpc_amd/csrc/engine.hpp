// engine.hpp -- what the C-ABI translation unit (cpmpc_api.hip) and the four kernel translation units
// (engine_<dtype>_<model>.hip) share: the solver handle, the error / profiling / staging helpers the API owns, and the
// table of entry points through which the API reaches the kernels of one (dtype, model) pair.  The library is built
// from five objects compiled in parallel; only the engine units contain device code.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "../../include/cpmpc.h"
#include "workspace_layout.hpp"

using namespace cpmpc;

#define CPMPC_HIDDEN __attribute__((visibility("hidden")))

// thread-local error text (cpmpc_last_error); returns `code`
CPMPC_HIDDEN int cpmpc_fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
#define fail cpmpc_fail

#define HIP_TRY(expr)                                                                       \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess)                                                                   \
      return fail(CPMPC_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),     \
                  __FILE__, __LINE__);                                                      \
  } while (0)

struct ProfSpan {
  int kernel;
  hipEvent_t start, stop;
};

// Staging of the host-pointer entry points: a device buffer, its pinned host mirror, a stream and an event.  A handle
// owns kHostSlots of them so that a large host-pointer step runs as a pipeline of chunks: while the CPU scatters chunk k's
// results into the caller's arrays, chunk k+1 is copying back and chunk k+2 is in the kernels.
constexpr int kHostSlots = 3;
constexpr int kMaxStages = 16;  // launches of the fused kernel per step at most (the plan of a step's stages)
struct HostSlot {
  void* dev = nullptr;
  void* pin = nullptr;
  size_t bytes = 0;
  hipStream_t stream = nullptr;
  hipEvent_t done = nullptr;  // the copy back of the chunk in flight has landed in `pin`
  // the chunk in flight (host_chunk_begin -> host_chunk_end)
  bool busy = false;
  int64_t c0 = 0, Bc = 0, g0 = 0;   // first problem in the handle's workspace, problems, first column in the caller's arrays
  bool want_pred = false, want_sol = false;
  bool direct = false;              // real-typed outputs were copied straight into the caller's (pinned) arrays
  bool per_dyn = false, per_sp = false, per_tw = false;  // which per-problem inputs the chunk carried (staging layout)
};

// the last plan made from a histogram, per host slot (cpmpc_plan_stages)
struct PlanCache {
  int seq = 0;
  int64_t B = 0;
  double n_hist = 0.0;
  int n = 0;
  int bounds[kMaxStages + 1] = {0};
  // the last two different histograms the host has seen on the slot, as fractions of problems per bin
  int seen_seq = 0;
  double seen[kFbBins] = {0};
  bool have_prev = false, stationary = false;
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
  // staging for the *_host entry points (lazily allocated, grown on demand, owned by the handle)
  HostSlot slot[kHostSlots];
  int64_t host_chunk = -1;       // problems per chunk of a pipelined host-pointer step (cpmpc_set_host_chunk); -1 = by batch size
  hipEvent_t ev_last = nullptr;  // end of the last device-pointer call on a caller's stream; the slot streams wait on it
  bool ev_pending = false;
  // profiling
  int profiling = 0;
  std::vector<ProfSpan> spans;
  std::vector<ProfSpan> free_spans;
  double prof_ms[CPMPC_KERNEL_COUNT] = {0, 0, 0, 0, 0};
  int64_t prof_n[CPMPC_KERNEL_COUNT] = {0, 0, 0, 0, 0};
  int pipeline = CPMPC_PIPELINE_AUTO;
  bool refine_qp = false;  // CPMPC_CREATE_REFINE_QP: the double fused kernels refine the whole QP solution once
  bool beyond_parity = false;  // window_length * control_dt > cpmpc_max_parity_horizon(): cpmpc_horizon_beyond_parity()
  int refine_passes = 0;   // split pipeline, double: passes of that refinement (1; 3 beyond the parity horizon: cpmpc_api.hip)
  bool wide_qp = false;    // CPMPC_CREATE_WIDE_QP (default for the 6-state model): the float fused kernels (compiled spacings) carry the QP's terminal part in double
  // staged fused pipeline (compaction of the still-active problems between stages); 0/0 = single launch
  // default 2 / 1 (round 4, tools/steady_state.py): in the warm-started closed loop most problems stop after one or two
  // iterations, so compacting after two pays (fp32 at the reference's tolerances 1.47 -> 1.34 ms per tick, the swing-up
  // transient -4 %); a stage that finds nobody left costs one empty launch (7 us)
  int stage_first = 2, stage_next = 1;
  bool stage_auto = true;  // default: stage only batches larger than one round of resident waves
  int32_t* active = nullptr;  // [cap] compacted problem indices, then two counters
  // the stages of the fused pipeline, planned per step (stage_auto) from the histogram of iterations per problem that
  // finalize_kernel left in host-mapped memory after an earlier step: per host slot, kFbReporters blocks of kFbBins + 1
  // ints (a reporting workgroup's counts + the sequence number of its step)
  int32_t* fb_host = nullptr;
  int32_t* fb_host_dev = nullptr;  // the device's address of fb_host
  int fb_seq[kHostSlots] = {0, 0, 0};
  int fb_reporters[kHostSlots] = {0, 0, 0};  // reporting workgroups of the last step launched on the slot
  int last_plan[kMaxStages + 1] = {0};  // boundaries of the stages of the last step (cpmpc_get_stage_plan)
  int last_plan_n = 0;
  PlanCache plan_cache[kHostSlots];
};


// owned by the API unit
CPMPC_HIDDEN void span_begin(cpmpc_solver* s, int kernel, hipStream_t stream, ProfSpan* cur);
CPMPC_HIDDEN void span_end(cpmpc_solver* s, hipStream_t stream, ProfSpan* cur);
CPMPC_HIDDEN int ensure_slot(cpmpc_solver* s, int slot, size_t bytes);
// the stages of the fused pipeline for a step of B problems: bounds[0] = 0 < ... < bounds[n] = max_iterations, returns n
CPMPC_HIDDEN int cpmpc_plan_stages(cpmpc_solver* s, int slot, int64_t B, bool exits, int* bounds);
// run fn(i) for i in [0, n) on the library's worker threads (the calling thread takes part); returns when all are done
CPMPC_HIDDEN void host_parallel_for(int64_t n, void (*fn)(int64_t i, void* ctx), void* ctx);

// fused pipeline: compiled specialisations for these (L = S-1, SP) pairs ...
static inline bool fused_static(int L, int SP) {  // the default horizon's spacings (N = 40) and N = 20
  return (L == 4 && SP == 10) || (L == 8 && SP == 5) || (L == 2 && SP == 10) || (L == 4 && SP == 5) ||
         (L == 2 && SP == 20) || (L == 5 && SP == 8) || (L == 10 && SP == 4);
}
// ... and a run-time-spacing variant (dynamic LDS) for any other spacing with one of these interval counts whose
// per-wave LDS (80 scalars per lane and control for NX = 4) fits the 64 KB a dynamic allocation may take
static inline size_t fused_dyn_bytes(const cpmpc_solver* s) {
  const size_t col = (size_t)s->NX * s->esize;
  const size_t g_bytes = col % 16 == 0 ? col : (col + 7) / 8 * 8;  // a column of Gamma in 16- or 8-byte pieces (mpc_fused.hpp)
  return (size_t)s->SP * 64 * (4 * (size_t)s->esize + g_bytes);
}
static inline bool fused_dynamic(const cpmpc_solver* s) {
  const int L = s->S - 1;
  const bool l_ok = L == 2 || L == 4 || L == 5 || L == 8 || L == 10 || L == 16;
  return l_ok && fused_dyn_bytes(s) <= 65536;
}
static inline bool fused_built(const cpmpc_solver* s) { return fused_static(s->S - 1, s->SP) || fused_dynamic(s); }

// LDS a wave of the fused kernel takes for this handle (mpc_fused.hpp: u, du, (U^-1 g), 1/d per control and lane plus a
// column of Gamma in 16-byte pieces; the slim layout of the double 6-state kernel keeps u, du and Gamma only)
static inline size_t fused_wave_lds_bytes(const cpmpc_solver* s) {
  const size_t col = (size_t)s->NX * s->esize;
  const size_t g_bytes = col % 16 == 0 ? col : (col + 7) / 8 * 8;
  // (the slim layout is the double 6-state kernel's: mpc_fused.hpp, CPMPC_FUSED_SLIM_F64_NX6 = 1)
  const bool slim = s->esize == 8 && s->NX > 4 && !s->refine_qp && s->SP <= 10 && fused_static(s->S - 1, s->SP);
  return (size_t)s->SP * 64 * ((slim ? 2 : 4) * (size_t)s->esize + g_bytes);
}

static inline bool use_fused(const cpmpc_solver* s) {
  if (s->pipeline == CPMPC_PIPELINE_SPLIT) return false;
  if (!fused_built(s)) return false;
  // AUTO: the double kernel of the 6-state model, which holds all 512 registers (one wave per SIMD at best), runs fused
  // where at least three of its waves fit a CU's 160 KB of LDS: 40 KB per wave at
  // state_spacing 10 since round 5 (four per CU: 23.4 M re-plans/s at B = 65 536 against 13.7 M split; with the padded
  // 60 KB layout of rounds 1-4, two per CU, it was 14.4 M and AUTO kept the split pipeline), 100 KB at spacing 20 (one per
  // CU: split).
  // (the other kernels keep what rounds 1-4 measured for them: fused wherever built)
  if (s->pipeline == CPMPC_PIPELINE_AUTO && s->model == CPMPC_MODEL_DOUBLE && s->dtype == CPMPC_F64 &&
      3 * fused_wave_lds_bytes(s) > 160u * 1024u)
    return false;
  // AUTO beyond the parity horizon (fp64; round 6): the split pipeline, whose QP kernel runs TWO refinement passes there -- from
  // the second pass on the refined condensed solve is more accurate than a dense pivoted LU on the worst cold starts
  // (kernels at fault on 1 lane of 8 192 at 1.6 s, like the CPU check; the fused kernel's one pass: 3).
  // Throughput is not the bar there; cpmpc_set_pipeline(FUSED) is the caller's to choose.
  if (s->pipeline == CPMPC_PIPELINE_AUTO && s->beyond_parity && s->dtype == CPMPC_F64 && s->refine_qp) return false;
  return true;
}

// Entry points of one (dtype, model) pair; every function launches the kernels of its own translation unit.
struct Engine {
  // Optimization::Step for B problems on `stream` (optimization.cc:39-97)
  // (col0: first problem of the handle's workspace the call works on; slot: which pair of compaction counters)
  int (*step_batch)(cpmpc_solver* s, int64_t B, const cpmpc_step_inputs* in, const cpmpc_step_outputs* out,
                    hipStream_t stream, int64_t col0, int slot);
  // one chunk of a host-pointer step in two halves: `begin` converts and uploads problems [c0, c0 + Bc) of the handle
  // (columns [g0, g0 + Bc) of the caller's [field][ld] arrays), queues the kernels and the copy back on the slot's
  // stream; `end` waits for it and scatters the results into the caller's arrays
  // (direct: copy the real-typed outputs by DMA straight into the caller's pinned arrays -- decided once per call)
  int (*host_chunk_begin)(cpmpc_solver* s, int slot, int64_t c0, int64_t Bc, int64_t g0, int64_t ld,
                          const cpmpc_step_host_inputs& in, const cpmpc_step_host_outputs& out, bool direct);
  int (*host_chunk_end)(cpmpc_solver* s, int slot, int64_t ld, const cpmpc_step_host_outputs& out);
  // packed z [dim][B] (MapKey order) <-> workspace
  void (*pack_z)(cpmpc_solver* s, int64_t B, const void* z, hipStream_t stream);
  void (*unpack_z)(cpmpc_solver* s, int64_t B, void* z_out, hipStream_t stream);
  // stand-alone pieces
  void (*dynamics)(int64_t B, const double* dyn_shared_host, const double* fext_host, const void* x, const void* u,
                   void* f, void* Jx, void* Ju, hipStream_t stream);
  void (*rk4)(int64_t B, const double* dyn_shared_host, const double* fext_host, double h, const void* x, const void* u,
              void* x_new, void* A, void* Bm, hipStream_t stream);
  void (*sim)(int64_t B, const double* dyn_shared_host, const double* fext_host, const void* fext, int n_sub,
              double h_last, const void* u, void* state, hipStream_t stream);
  void (*linearize_batch)(cpmpc_solver* s, int64_t B, const double* dyn_shared_host, const void* z, void* c, void* Phi,
                          void* Gamma, hipStream_t stream);
  // debug builds (-DCPMPC_FUSED_TIMING / -DCPMPC_FUSED_CLOCK): read and clear this unit's counters, ADDING them to out
  // (which = 0: eight phase counters, 1: {cycles, 100 MHz ticks, waves, max cycles}); -1 when not built in
  int (*debug_read)(int which, unsigned long long* out);
};
// (functions, not namespace-scope tables: hipcc would emit a constant table for the device side as well)
CPMPC_HIDDEN const Engine* cpmpc_engine_f32_single();
CPMPC_HIDDEN const Engine* cpmpc_engine_f64_single();
CPMPC_HIDDEN const Engine* cpmpc_engine_f32_double();
CPMPC_HIDDEN const Engine* cpmpc_engine_f64_double();
static inline const Engine* engine_for(int dtype, int model) {
  if (model == CPMPC_MODEL_SINGLE) return dtype == CPMPC_F32 ? cpmpc_engine_f32_single() : cpmpc_engine_f64_single();
  return dtype == CPMPC_F32 ? cpmpc_engine_f32_double() : cpmpc_engine_f64_double();
}
static inline const Engine* engine_of(const cpmpc_solver* s) { return engine_for(s->dtype, s->model); }
