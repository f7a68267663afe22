/*
 * cpmpc.h -- C-ABI of the MI355X-native batched cart-pole MPC hot path (libcpmpc.so).
 *
 * The reference (gareth-cross/cart-pole-mpc) has NO plugin/FFI boundary for this path: the hot
 * path is the C++ class API
 *     pendulum::Optimization   optimization/optimization.hpp:73-108
 *     pendulum::Simulator      optimization/simulator.hpp:10-29
 * linked statically into the nanobind module `pypendulum` (wrapper/wrapper.cc:40-102).  This header
 * is the boundary a maintainer would bind instead: plain pointers and sizes, int return codes, no
 * exceptions, caller owns every buffer, the handle owns the device-resident warm-start state
 * (the batched counterpart of Optimization::previous_solution_, optimization.hpp:107).
 * Each entry point names the reference interface it replaces.
 *
 * Data layout: every batched array is structure-of-arrays, [field][B] with the batch index
 * fastest, in the handle's dtype (float or double), in DEVICE memory unless the name ends in
 * `_host`.  Variable layout of the solution vector z follows MapKey (optimization.cc:27-37):
 *     z[4*s + t]  state t of shooting node s   (t: 0=b_x 1=th_1 2=b_x_dot 3=th_1_dot; key.hpp:8-19)
 *     z[4*S + k]  control u_k                  (S = window_length/state_spacing + 1, k < N)
 */
#ifndef CPMPC_H
#define CPMPC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CPMPC_STATE_DIM 4
#define CPMPC_NUM_DYN_PARAMS 9

/* ---- return codes ---------------------------------------------------------------------------- */
enum {
  CPMPC_OK = 0,
  CPMPC_ERR_INVALID_ARG = 1,  /* violates a precondition the reference asserts (F_ASSERT*) */
  CPMPC_ERR_UNSUPPORTED = 2,  /* valid in the reference but not built into this library */
  CPMPC_ERR_NO_DEVICE = 3,    /* no HIP device / wrong architecture: there is NO CPU fallback */
  CPMPC_ERR_HIP = 4,          /* a HIP runtime call failed; see cpmpc_last_error() */
  CPMPC_ERR_ALLOC = 5,
  CPMPC_ERR_BATCH = 6         /* B exceeds the capacity given to cpmpc_create */
};

enum { CPMPC_F32 = 0, CPMPC_F64 = 1 };

/* Per-problem termination state, written to `status`.  Names and meaning follow
 * mini_opt::NLSTerminationState as the reference uses it (optimization_test.cc:44-46). */
enum {
  CPMPC_TERM_NONE = 0,
  CPMPC_TERM_MAX_ITERATIONS = 1,
  CPMPC_TERM_SATISFIED_ABSOLUTE_TOL = 2,
  CPMPC_TERM_SATISFIED_RELATIVE_TOL = 3,
  CPMPC_TERM_SATISFIED_FIRST_ORDER_TOL = 4,
  CPMPC_TERM_QP_INDEFINITE = 5,
  CPMPC_TERM_USER_CALLBACK = 6,
  CPMPC_TERM_MAX_LAMBDA = 7,
  CPMPC_TERM_NON_FINITE = 8
};

/* ---- parameters ------------------------------------------------------------------------------ */

/* pendulum::OptimizationParams, field for field (optimization/optimization.hpp:12-53). */
typedef struct cpmpc_params {
  double control_dt;
  uint64_t window_length;
  uint64_t state_spacing;
  uint64_t max_iterations;
  double relative_exit_tol;
  double absolute_first_derivative_tol;
  double equality_penalty_initial;
  double u_guess_sinusoid_amplitude;
  double u_cost_weight;
  double u_derivative_cost_weight;
  double b_x_final_cost_weight;
  double th_final_cost_weight;
  double b_x_dot_final_cost_weight;
  double th_dot_final_cost_weight;
} cpmpc_params;

/* Knobs of the SQP that stands where mini_opt::ConstrainedNonlinearLeastSquares stands
 * (optimization.cc:73-81).  Only max_line_search_iterations (=5, optimization.cc:76) and the two
 * clamps (optimization.cc:320,327) come from the reference; see DESIGN.md section 4. */
typedef struct cpmpc_solver_opts {
  int32_t max_line_search_iterations;
  double armijo_c1;
  double ls_shrink_max;
  double ls_shrink_min;
  double ls_alpha_growth;
  double penalty_rho;
  double lambda_initial;
  double lambda_failure_init;
  double lambda_scale_up;
  double lambda_scale_down;
  double lambda_min;
  double lambda_max;
  double b_x_limit;
  double u_limit;
  double ls_alpha_growth_backtracked; /* growth used instead of ls_alpha_growth when the accepted search backtracked */
  double full_step_below; /* an UNDAMPED QP step (lambda = 0) with |dz|_inf <= this is taken in full without the merit
                           * test (local convergence safeguard; default 1e-4, 0 disables) */
  double exit_defect_floor; /* the first-order exit test |g.dz - mu |c|_1| < absolute_first_derivative_tol counts |c|_1 as
                             * zero when it is at most exit_defect_floor * state_spacing * eps * (S - 1) * sum_t (|target_t| +
                             * |x_{S-1,t} - target_t|) -- the size of the states, taken at the terminal node, times the number of
                             * intervals (eps of the kernels' arithmetic): equality residuals of that size are the rounding of
                             * the rollout itself -- state_spacing RK4 steps -- and no iteration can remove them.  In fp64
                             * that floor would be 3e-14: the rule is a SINGLE-PRECISION rule -- CPMPC_F64 handles ignore this
                             * option (their kernels do not carry the test, and neither does the double CPU check: one exit
                             * rule in the parity dtype); in fp32 it is 1.5e-5, and it is what lets settled
                             * controllers leave after one iteration at the reference's tolerance of 1e-6 instead of
                             * iterating on noise (DESIGN.md sections 4 and 6.4; default 2, 0 disables; appended in round 4).
                             * Measured, fp32, 65 536 settled controllers: 0 -> 2.8 iterations per tick, 2 -> 1.8, 8 -> 1.0,
                             * the same closed-loop accuracy throughout */
} cpmpc_solver_opts;

void cpmpc_default_params(cpmpc_params* p);           /* optimization.hpp:12-48 defaults */
void cpmpc_default_solver_opts(cpmpc_solver_opts* o); /* DESIGN.md section 4 defaults */

/* Thread-local text of the last failure on this thread. */
const char* cpmpc_last_error(void);

/* Number of usable gfx950 devices (0 if none); does not create a context. */
int cpmpc_device_count(void);

/* ---- the solver handle: replaces `pendulum::Optimization` ---------------------------------- */

typedef struct cpmpc_solver cpmpc_solver;

/* Replaces Optimization::Optimization(const OptimizationParams&) (optimization.cc:13-22), for a
 * batch of up to `max_batch` independent controllers on HIP device `device`.  The reference's
 * constructor preconditions give CPMPC_ERR_INVALID_ARG.  Any state_spacing that divides window_length is
 * accepted (optimization.cc:16-18); which ones have specialised kernels: cpmpc_supported_state_spacing(). */
int cpmpc_create(const cpmpc_params* params, const cpmpc_solver_opts* opts /*nullable*/, int dtype,
                 int64_t max_batch, int device, cpmpc_solver** out);
void cpmpc_destroy(cpmpc_solver* s);

/* The same constructor with its arguments in a size-versioned struct, plus what the positional forms cannot express:
 *   flags      CPMPC_CREATE_STRICT_HORIZON: refuse (CPMPC_ERR_UNSUPPORTED) a horizon window_length * control_dt beyond
 *              cpmpc_max_parity_horizon().  By DEFAULT every horizon the reference's constructor accepts
 *              (optimization.cc:13-22) is accepted, by every entry point -- the library is a drop-in -- and the first
 *              such handle of a process writes one warning to stderr (CPMPC_CREATE_ALLOW_LONG_HORIZON, the opt-in of
 *              round 4 when the refusal was the default, is still accepted and now only silences that warning).  What the
 *              bound means: the QP is solved by eliminating the states through the shooting recursion, and through more than
 *              ~1 s of the default pole (unstable at e^{6.3 t}) that loses digits against a full-space KKT solve with
 *              pivoting -- measured at control_dt 0.01 (profiles/r04_parity_sweep.json): window_length 80 and 100 keep
 *              every one of 32 768 cold-start problems within 3e-6 / 4e-7 of the CPU check (three to eight iterations);
 *              at 120, 2 of 16 384 are beyond 1e-5 (worst 1.6e-5); at 160, 0.4 % are (worst 1e-2 .. 0.4 depending on
 *              the sample).  Warm-started closed loops are not affected in practice, cold starts far from the optimum
 *              are; a caller that needs the 1e-5 bar on every problem asks for the refusal.
 *   opts_size  sizeof(cpmpc_solver_opts) as the CALLER was compiled (0 = this header's).  Option fields are only ever
 *              appended; a caller built against an earlier header passes its shorter size and keeps the library's
 *              defaults for the fields it does not know (full_step_below was appended in round 3, exit_defect_floor in
 *              round 4).  Must be the size of the struct in some release: 8 + 8 k bytes, 112 (the fields through u_limit) <= size <= this header's.
 *              Always start from cpmpc_default_solver_opts: a zero-initialised struct is NOT the defaults.
 *              The positional constructors (cpmpc_create, cpmpc_create_model, cpmpc_sharded_create) cannot be told the
 *              caller's size: they read CPMPC_SOLVER_OPTS_SIZE_POSITIONAL bytes -- the struct as it was when they were
 *              frozen, i.e. up to and including full_step_below -- and take the library's defaults for everything
 *              appended since (exit_defect_floor); newer options are set through cpmpc_create_ex.
 *   flags      CPMPC_CREATE_REFINE_QP / CPMPC_CREATE_NO_REFINE_QP: force on / off one step of iterative refinement of the
 *              whole QP solution in the fused CPMPC_F64 kernels -- residuals evaluated in the original data (terminal rows
 *              through the recovered states, stationarity through the adjoint), solved again with the factors at hand.
 *              It brings the condensed solve below the error of a dense KKT solve with pivoting (CPU model, worst of
 *              150 problems: 3e-14 against 3e-13 at w_u = 0, w_du = 0.1) and costs 7 % of the step (50.2 -> 46.7 M
 *              re-plans/s at B = 262 144).  DEFAULT (neither flag): on when u_cost_weight < 0.05, half the reference's
 *              0.1.  Measured (profiles/r04_fuzz_sweep_2000*.json: 2 000 random problem definitions x 2 048 lanes, the
 *              extended-precision arbiter on every lane that is off; "at fault" = beyond 1e-5 and more than twice as far
 *              from the extended-precision answer as the double CPU check):
 *                 u_cost_weight >= 0.05:  0 of 1 798 144 lanes at fault, refined or not;
 *                 u_cost_weight <  0.05:  144 of 1 374 208 without the refinement, 9 with it (two definitions whose
 *                                         controls run into the +-300 N clamp, where the retraction is not smooth);
 *                 and, whatever the weights, definitions whose friction the explicit RK4 cannot follow (v_mu_b = 1e-7
 *                 with mu_b > 0: a slope of 1e5..1e6 1/s against a stability limit of 280 1/s at 10 ms; |Phi| reaches
 *                 1e4 per interval, the terminal system's condition 1e18): 281 of 923 648 lanes, 72 with it.
 *              Both pipelines; ignored by CPMPC_F32 handles.  cpmpc_refines_qp() tells what a handle does. */
#define CPMPC_CREATE_ALLOW_LONG_HORIZON 1u /* accepted; silences the long-horizon warning */
#define CPMPC_CREATE_REFINE_QP 2u
#define CPMPC_CREATE_NO_REFINE_QP 4u
#define CPMPC_CREATE_STRICT_HORIZON 8u
#define CPMPC_CREATE_WIDE_QP 16u
#define CPMPC_CREATE_NO_WIDE_QP 32u
typedef struct cpmpc_create_info {
  uint32_t struct_size; /* = sizeof(cpmpc_create_info) */
  uint32_t flags;
  int32_t dtype;  /* CPMPC_F32 / CPMPC_F64 */
  int32_t model;  /* CPMPC_MODEL_* */
  int32_t device; /* HIP device */
  int32_t reserved; /* 0 */
  int64_t max_batch;
  const cpmpc_params* params;
  const cpmpc_solver_opts* opts; /* nullable */
  uint64_t opts_size;
} cpmpc_create_info;
int cpmpc_create_ex(const cpmpc_create_info* info, cpmpc_solver** out);
/* bytes of cpmpc_solver_opts the positional constructors read (the struct through full_step_below) */
#define CPMPC_SOLVER_OPTS_SIZE_POSITIONAL 128u
int cpmpc_refines_qp(const cpmpc_solver* s); /* 1: this handle's kernels refine the QP solution (CPMPC_CREATE_REFINE_QP) */
/* The solver options a handle actually uses -- this library's defaults overlaid with as many leading bytes of the caller's
 * struct as its constructor read: the positional constructors (cpmpc_create, cpmpc_create_model, cpmpc_sharded_create) read
 * CPMPC_SOLVER_OPTS_SIZE_POSITIONAL bytes, so a field appended later (exit_defect_floor) set through them is NOT taken and
 * keeps its default; cpmpc_create_ex reads opts_size.  out_size: sizeof of the caller's struct (112 .. sizeof here, 8 + 8 k).
 * A struct shorter than 112 bytes (the 13 doubles through u_limit: the shortest layout that is a prefix of today's) is
 * refused by every constructor. */
int cpmpc_get_solver_opts(const cpmpc_solver* s, cpmpc_solver_opts* out, size_t out_size);
/* CPMPC_CREATE_WIDE_QP / CPMPC_CREATE_NO_WIDE_QP (CPMPC_F32 handles; ignored by CPMPC_F64 ones): force on / off that the fused
 * kernels carry the whole terminal part of the QP in double -- the products of transition matrices across the shooting
 * intervals, the columns of U^-1 R^T, the multipliers and their effect on the step -- not only the NX x NX system.  It is
 * the precision of the QP solve, not the hardware's sin / cos / exp, that separates a float handle from the double CPU
 * check: with it the kernels end where the float build of that check (its KKT solve in double) ends.  Measured (round 5,
 * cold starts, 5 iterations, max |du| per problem against the double check: median / 99th percentile / share within 1e-2):
 *   4-state model, B = 262 144:  2.4e-4 / 0.11 / 93.0 %  ->  8.4e-5 / 4.9e-3 / 99.5 %   (float CPU check 8.0e-5 / 4.9e-3 / 99.3 %)
 *                                for 3.8 % of the cold-start throughput (122.5 -> 117.9 M re-plans/s) and 7 - 9 % of a
 *                                closed-loop tick, where it changes nothing that matters (warm-started iterations near the
 *                                optimum: same iteration counts, same final pole error);
 *   6-state model, B = 65 536:   4.2e-2 / 8.0 / 19.8 %   ->  5.7e-4 / 9.3e-3 / 99.1 %   (float CPU check 5.2e-4 / 8.7e-3 / 99.2 %)
 *                                for 0 - 5 % (51.0 -> 50.8 M near upright, 50.9 -> 48.3 M from within 0.5 rad).
 * DEFAULT (neither flag): ON for the 6-state model -- without it four of five float solves of that model are off by more
 * than 0.01 N after five iterations -- and OFF for the 4-state one (the reference's model: speed first, the bar of this
 * path is met by CPMPC_F64 handles; a caller who re-plans from cold starts in single precision should pass the flag: 3.8 %
 * of the throughput for 93.7 -> 99.4 % of the problems within 1e-2 of the double answer).  The option belongs to the handle,
 * not to the step: "on for cold-start steps only" was tried in round 6 and withdrawn (a problem's arithmetic would then depend
 * on which other problems share its call).  Both pipelines since round 6: a float handle that AUTO or cpmpc_set_pipeline() sends to
 * the split pipeline (state spacings the fused kernel is not built for) runs qp_ls_kernel's wide form -- the same quantities
 * in double, one more pass over the workspace -- and lands where the fused one does (tests/test_gpu_round6.py: the 6-state
 * model at state_spacing 20 against the float CPU check).  cpmpc_wide_qp() tells what a handle does, in either pipeline. */
int cpmpc_wide_qp(const cpmpc_solver* s);
/* seconds: the longest horizon held to 1e-5 of the CPU check on every problem (1.0) */
double cpmpc_max_parity_horizon(void);
/* 1: this handle's horizon window_length * control_dt is beyond cpmpc_max_parity_horizon() -- a PER-HANDLE status (round 6;
 * the once-per-process line on stderr is easy to miss in a notebook): such a handle is solved as asked, with the QP
 * refinement of CPMPC_CREATE_REFINE_QP on by default (CPMPC_F64) and, under CPMPC_PIPELINE_AUTO, by the split pipeline with
 * two refinement passes; on a few cold starts in 10^4 far from the optimum its controls may still differ from a full-space
 * solve by more than 1e-5 (measured at 1.6 s, three iterations: 1 of 8 192 lanes with the kernels at fault by the
 * extended-precision arbiter and 1 with the CPU check at fault, 34 / 0 without the refinement; a dense pivoted solve in
 * double is itself up to 6.6e-5 from the extended-precision answer there).  pendulum::Optimization::HorizonBeyondParity(), the
 * `horizon_beyond_parity` attribute of pypendulum.Optimization and a line in solver_summary() carry it to the caller.
 * 0: within the bound; -1: null handle.  Replaces nothing in the reference (optimization.cc:13-22 accepts any horizon). */
int cpmpc_horizon_beyond_parity(const cpmpc_solver* s);

/* 2: register-resident linearisation compiled for this spacing (1,2,4,5,8,10,20); 1: served by the generic
 * run-time-spacing kernel; 0: not a valid spacing */
int cpmpc_supported_state_spacing(int spacing);

/* Inputs of one batched re-plan.  Exactly one of dyn_shared_host / dyn must be non-NULL. */
typedef struct cpmpc_step_inputs {
  const void* x0;                /* [4][B]  measured states (Step's current_state) */
  const double* dyn_shared_host; /* HOST [9] SingleCartPoleParams shared by the batch, or NULL */
  const void* dyn;               /* [9][B]  per-problem SingleCartPoleParams, or NULL */
  double set_point_shared;       /* b_x_set_point shared by the batch (used if set_point NULL) */
  const void* set_point;         /* [B]     per-problem b_x_set_point, or NULL */
  /* [NX][B] per-problem terminal weights in state order {b_x, th.., b_x', th'..}, replacing the four *_final_cost_weight
   * parameters for this call: >= 0 a cost row with that weight, < 0 an equality row (optimization.cc:236-267; the
   * per-controller toggles of viz/src/application.ts:279-342).  NULL: the handle's parameters apply to every problem. */
  const void* terminal_weights;
} cpmpc_step_inputs;

/* Outputs; every pointer is nullable. */
typedef struct cpmpc_step_outputs {
  void* u;             /* [N][B]     OptimizationOutputs::u                (optimization.hpp:66) */
  void* predicted;     /* [N][4][B]  OptimizationOutputs::predicted_states (optimization.hpp:69) */
  int32_t* status;     /* [B]        solver_outputs.termination_state */
  int32_t* iterations; /* [B]        QP solves performed */
  int32_t* ls_evals;   /* [B]        merit evaluations performed */
  void* final_cost;    /* [B]        1/2 |r|^2 at the returned solution's last evaluation.  A problem that leaves with
                        *            SATISFIED_FIRST_ORDER_TOL after a tiny undamped step took that step without a merit
                        *            evaluation: its final_cost / final_eq_l1 are those of the iterate the
                        *            step was computed from, one (tiny: |dz|_inf <= full_step_below) step behind u / solution */
  void* final_eq_l1;   /* [B]        |c|_1 there */
  void* guess;         /* [dim][B]   the initial guess handed to the solver */
  void* solution;      /* [dim][B]   the solution z = solver_->variables() (optimization.cc:85), MapKey order: what the
                        *            next Step reports as OptimizationOutputs::previous_solution (optimization.cc:84) */
} cpmpc_step_outputs;

/* Replaces Optimization::Step (optimization.cc:39-97) for B problems in lock-step.  Warm start
 * (shift of the previous solution, optimization.cc:50-57) is used when the handle holds one, else
 * the sinusoid cold start (optimization.cc:58-68).  Asynchronous on `stream` (a hipStream_t, NULL
 * = the default stream); all pointers must stay valid until the stream reaches this point. */
int cpmpc_step_batch(cpmpc_solver* s, int64_t B, const cpmpc_step_inputs* in,
                     const cpmpc_step_outputs* out, void* stream);

/* Replaces Optimization::Reset (optimization.hpp:83). */
int cpmpc_reset(cpmpc_solver* s);
/* Replaces Optimization::SetPreviousSolution (optimization.hpp:86-89); z is [dim][B]. */
int cpmpc_set_previous_solution(cpmpc_solver* s, int64_t B, const void* z, void* stream);
/* Reads the warm-start state (the reference exposes it as OptimizationOutputs::previous_solution
 * of the NEXT step, optimization.cc:84); z_out is [dim][B]. */
int cpmpc_get_solution(cpmpc_solver* s, int64_t B, void* z_out, void* stream);
int cpmpc_has_previous_solution(const cpmpc_solver* s);
/* Warm-start state is per problem, as one Optimization object per controller is in the reference
 * (optimization.cc:46-68: a controller without a previous solution starts from the sinusoid guess): problems
 * [0, n) hold a previous solution, n = the largest B stepped or set since the last cpmpc_reset; a later step with a
 * larger B warm-starts those and cold-starts the rest. */
int64_t cpmpc_previous_solution_batch(const cpmpc_solver* s);

int cpmpc_dim(const cpmpc_solver* s);        /* 4*S + N (optimization.cc:204-205) */
int cpmpc_num_states(const cpmpc_solver* s); /* OptimizationParams::NumStates (optimization.hpp:52) */
int cpmpc_dtype(const cpmpc_solver* s);

/* Host-pointer convenience used by the C++ facade (cart-pole-mpc_amd/host): same semantics as
 * cpmpc_step_batch with HOST double arrays in the same SoA layouts; copies in, runs on the GPU,
 * copies out, synchronises.  There is no CPU compute path behind it. */
int cpmpc_step_batch_host(cpmpc_solver* s, int64_t B, const double* x0_host,
                          const double* dyn_shared_host, double set_point, double* u_host,
                          double* predicted_host, int32_t* status_host, int32_t* iterations_host,
                          double* final_cost_host, double* final_eq_l1_host);
/* The same with the outputs in a struct of HOST pointers (every one nullable), including the solution z, so that
 * Optimization::Step (which also keeps previous_solution_, optimization.cc:85) is ONE round trip: one copy in, the
 * kernels, one copy out, one synchronisation, on the handle's own stream and pinned staging. */
typedef struct cpmpc_step_host_outputs {
  double* u;             /* [N][B] */
  double* predicted;     /* [N][4][B] */
  int32_t* status;       /* [B] */
  int32_t* iterations;   /* [B] */
  double* final_cost;    /* [B] */
  double* final_eq_l1;   /* [B] */
  double* solution;      /* [dim][B] */
} cpmpc_step_host_outputs;
int cpmpc_step_batch_host_ex(cpmpc_solver* s, int64_t B, const double* x0_host, const double* dyn_shared_host,
                             double set_point, const cpmpc_step_host_outputs* out);
/* The general form: cpmpc_step_inputs with HOST double arrays, including the per-problem parameters, set-points and
 * terminal rows (optimization.cc:236-267).  Exactly one of dyn_shared / dyn must be non-NULL. */
typedef struct cpmpc_step_host_inputs {
  const double* x0;               /* [4][B] */
  const double* dyn_shared;       /* [9] shared by the batch, or NULL */
  const double* dyn;              /* [9][B] per problem, or NULL */
  double set_point_shared;        /* used if set_point is NULL */
  const double* set_point;        /* [B] or NULL */
  const double* terminal_weights; /* [4][B] (>= 0 cost row, < 0 equality row) or NULL */
} cpmpc_step_host_inputs;
int cpmpc_step_batch_host_in(cpmpc_solver* s, int64_t B, const cpmpc_step_host_inputs* in,
                             const cpmpc_step_host_outputs* out);
/* How the host-pointer calls run (round 4).  A step of more than 1.5 x `problems` problems is split into chunks that
 * rotate through three staging slots on three streams: while the CPU scatters chunk k's results into the caller's arrays
 * (on the library's worker threads; CPMPC_HOST_THREADS, default 8), chunk k+1 is copying back and chunk k+2 is in the
 * kernels.  Results are bitwise those of the unsplit call (a problem's arithmetic does not depend on its neighbours).
 * -1 (default): eight chunks of at least 16 384 problems (sixteen of at least 8 192 when the results go by DMA into
 * pinned caller arrays); 0 = never split.  Measured at B = 262 144, fp64: see INTEGRATION.md section 4. */
int cpmpc_set_host_chunk(cpmpc_solver* s, int64_t problems);
/* Pin a host array the caller keeps (page-locks it and maps it for DMA: hipHostRegister).  When a CPMPC_F64 handle's
 * host-pointer step asks for the predicted states and its real-typed output arrays (u, predicted, solution) are all
 * pinned -- by this call, hipHostMalloc or hipHostRegister -- the results are copied by DMA straight into them and no CPU
 * pass over the data remains.  The array must stay allocated until cpmpc_host_unregister. */
int cpmpc_host_register(void* ptr, uint64_t bytes);
int cpmpc_host_unregister(void* ptr);
int cpmpc_set_previous_solution_host(cpmpc_solver* s, int64_t B, const double* z_host);
int cpmpc_get_solution_host(cpmpc_solver* s, int64_t B, double* z_host);

/* ---- pieces of the path, exposed for callers and for parity tests ----------------------------- */

/* gen::single_pendulum_dynamics (single_pendulum_dynamics.hpp:13-186), batched.
 * fext_host = {f_base.x, f_base.y, f_mass.x, f_mass.y} shared, NULL = zero.
 * f [4][B]; Jx [16][B] row-major 4x4, nullable; Ju [4][B], nullable. */
int cpmpc_dynamics_batch(int dtype, int64_t B, const double* dyn_shared_host, const void* x,
                         const void* u, const double* fext_host, void* f, void* Jx, void* Ju,
                         void* stream);

/* runge_kutta_4th_order<4> (integration.hpp:13-49) when A/Bm are given, else
 * runge_kutta_4th_order_no_jacobians<4> (integration.hpp:52-62).  A [16][B], Bm [4][B]. */
int cpmpc_rk4_batch(int dtype, int64_t B, const double* dyn_shared_host, const void* x,
                    const void* u, double h, const double* fext_host, void* x_new, void* A,
                    void* Bm, void* stream);

/* The shooting constraints of BuildProblem (optimization.cc:99-160,208-225) linearised at z:
 * defect c [4*(S-1)][B], Phi [16*(S-1)][B] (row-major 4x4 per interval), Gamma [4*N][B]
 * (element (r, k) at field 4*k + r).  Runs the same kernel cpmpc_step_batch runs. */
int cpmpc_linearize_batch(cpmpc_solver* s, int64_t B, const double* dyn_shared_host, const void* z,
                          void* c, void* Phi, void* Gamma, void* stream);

/* Simulator::Step (simulator.cc:11-36), batched: state [4][B] in/out, u [B];
 * fext_host shared {f_base.x, f_base.y, f_mass.x, f_mass.y} or NULL; fext [4][B] per-problem or
 * NULL (takes precedence).  dt < 0 or non-finite -> CPMPC_ERR_INVALID_ARG (simulator.cc:13). */
int cpmpc_sim_step_batch(int dtype, int64_t B, const double* dyn_shared_host, double dt,
                         const void* u, const double* fext_host, const void* fext, void* state,
                         void* stream);

/* ---- models ----------------------------------------------------------------------------------- */
/* CPMPC_MODEL_SINGLE: the reference's cart + single pole (4 states, 9 parameters), everything above.
 * CPMPC_MODEL_DOUBLE: cart + double pole of symbolic/dynamics_double.py:25-148 (BASELINE config 5):
 *   state {b_x, th_1, th_2, b_x', th_1', th_2'} (6), parameters {m_b, m_1, m_2, l_1, l_2, g} (6), no
 *   dissipation / external forces.  The reference has no optimizer for it (optimization.cc:197-199
 *   hard-codes 4 states); here the same OptimizationParams apply with th_final / th_dot_final acting on
 *   both poles and both poles' targets upright, variables laid out by MapKey<6>.
 * With a model argument every array's leading extent 4 becomes the model's state dimension and the
 * parameter vector its parameter count. */
enum { CPMPC_MODEL_SINGLE = 0, CPMPC_MODEL_DOUBLE = 1 };
int cpmpc_model_state_dim(int model);  /* 4 or 6; -1 for an unknown model */
int cpmpc_model_num_params(int model); /* 9 or 6 */
int cpmpc_create_model(const cpmpc_params* params, const cpmpc_solver_opts* opts /*nullable*/, int dtype,
                       int64_t max_batch, int device, int model, cpmpc_solver** out);
int cpmpc_model(const cpmpc_solver* s);
int cpmpc_dynamics_batch_model(int model, int dtype, int64_t B, const double* dyn_shared_host, const void* x,
                               const void* u, const double* fext_host, void* f, void* Jx, void* Ju,
                               void* stream);
int cpmpc_rk4_batch_model(int model, int dtype, int64_t B, const double* dyn_shared_host, const void* x,
                          const void* u, double h, const double* fext_host, void* x_new, void* A, void* Bm,
                          void* stream);
int cpmpc_sim_step_batch_model(int model, int dtype, int64_t B, const double* dyn_shared_host, double dt,
                               const void* u, const double* fext_host, const void* fext, void* state,
                               void* stream);

/* Host-pointer convenience for Simulator::Step (fp64 on the GPU; used by the C++ facade):
 * state_host [4][B] in/out, u_host [B], fext_host shared or NULL.  No CPU compute path. */
int cpmpc_sim_step_batch_host(int64_t B, const double* dyn_shared_host, double dt,
                              const double* u_host, const double* fext_host, double* state_host);

/* ---- several GPUs from ONE process ------------------------------------------------------------------ */
/* The reference is single-threaded and single-device; a batch of independent controllers shards
 * embarrassingly, so this is new surface: one `cpmpc_sharded` owns one solver handle + one stream per shard, a shard
 * living on one HIP device.  A step splits the batch contiguously (shard i owns columns
 * [i*B/n + min(i, B%n), ...) -- the same rule as cart-pole-mpc_amd/sharding.py: shard_range), runs every shard
 * concurrently, and assembles the outputs in global problem order.  No data-path collective: the only traffic between
 * devices is the scatter of x0 and the gather of the results, point-to-point copies to/from the root device (shard 0's)
 * over xGMI, or through each shard's pinned staging for the host-pointer call.  `devices` may name a device more than
 * once (two shards on one GPU: how the tests run it on a one-GPU box; results are bitwise those of one handle).
 * pendulum::ShardedOptimization (cart-pole-mpc_amd/host/sharded_optimization.hpp) is the C++ class over it. */
typedef struct cpmpc_sharded cpmpc_sharded;
/* devices == NULL: every visible gfx950 device, one shard each (n_devices ignored).  max_batch is the TOTAL batch. */
int cpmpc_sharded_create(const cpmpc_params* params, const cpmpc_solver_opts* opts /*nullable*/, int dtype,
                         int64_t max_batch, const int* devices, int n_devices, cpmpc_sharded** out);
/* the same from a cpmpc_create_info (its `device` is ignored, its max_batch is the TOTAL batch): any model, creation flags */
int cpmpc_sharded_create_ex(const cpmpc_create_info* info, const int* devices, int n_devices, cpmpc_sharded** out);
void cpmpc_sharded_destroy(cpmpc_sharded* s);
int cpmpc_sharded_num_shards(const cpmpc_sharded* s);
int cpmpc_sharded_device(const cpmpc_sharded* s, int shard);          /* HIP device of a shard; -1 if out of range */
/* What cpmpc_sharded_create saw for this shard's device: 1 peer access between it and the root device (shard 0's) is
 * mapped in both directions (or it IS the root device) -- its slices travel device to device over xGMI; 0 not -- the
 * runtime stages those copies through host memory (slower, still correct); -1 out of range. */
int cpmpc_sharded_peer_access(const cpmpc_sharded* s, int shard);
cpmpc_solver* cpmpc_sharded_handle(cpmpc_sharded* s, int shard);      /* the shard's own solver (options, profiling) */
/* columns [*lo, *hi) of a B-problem batch that shard `shard` solves */
int cpmpc_sharded_range(const cpmpc_sharded* s, int shard, int64_t B, int64_t* lo, int64_t* hi);
int cpmpc_sharded_reset(cpmpc_sharded* s);                             /* Optimization::Reset on every shard */

/* Warm starts and the batch size.  Which shard owns a column depends on B, so the shards' previous solutions are tied to
 * the B of the call that produced them.  The rule, checked on every step / set / get:
 *   - a call with the same B as the last one uses the warm starts in place;
 *   - a call with another B first HANDS THE WARM START OVER to the new split (the solutions of the warm columns are
 *     gathered on the root device, every shard is reset, and each receives the columns it owns under the new B):
 *     columns [0, min(B, n)) stay warm, n = cpmpc_sharded_previous_solution_batch(); columns beyond B are dropped
 *     (a single handle would keep them: include/cpmpc.h, cpmpc_previous_solution_batch), new columns start cold;
 *   - a step that fails part-way resets every shard (no mixture of old and new solutions survives).
 * Never a silent misalignment.  The hand-over is synchronous and costs one gather + one scatter of [dim][n] scalars. */
int64_t cpmpc_sharded_previous_solution_batch(const cpmpc_sharded* s);
/* cpmpc_horizon_beyond_parity() of the sharded handle (its shards share one parameter set); -1: null handle */
int cpmpc_sharded_horizon_beyond_parity(const cpmpc_sharded* s);
/* Optimization::SetPreviousSolution over all shards (optimization.hpp:86-89): z [dim][B] on the ROOT device in the
 * handle's dtype (asynchronous on `stream`, a stream of the root device), or HOST doubles.  Replaces every warm start. */
int cpmpc_sharded_set_previous_solution(cpmpc_sharded* s, int64_t B, const void* z, void* stream);
int cpmpc_sharded_set_previous_solution_host(cpmpc_sharded* s, int64_t B, const double* z_host);
/* The warm start of columns [0, B), B <= cpmpc_sharded_previous_solution_batch() (what the next step reports as
 * OptimizationOutputs::previous_solution, optimization.cc:84): z_out [dim][B]. */
int cpmpc_sharded_get_solution(cpmpc_sharded* s, int64_t B, void* z_out, void* stream);
int cpmpc_sharded_get_solution_host(cpmpc_sharded* s, int64_t B, double* z_host);

/* cpmpc_step_batch_host_in over all shards: HOST arrays in the global layouts ([4][B] in, [N][B] etc. out), per-problem
 * parameters / set-points / terminal rows included; every shard's chunks are in flight together (cpmpc_set_host_chunk
 * on a shard's handle changes its chunk size). */
int cpmpc_sharded_step_batch_host_in(cpmpc_sharded* s, int64_t B, const cpmpc_step_host_inputs* in,
                                     const cpmpc_step_host_outputs* out);
/* shared parameters only (round 3's form) */
int cpmpc_sharded_step_batch_host(cpmpc_sharded* s, int64_t B, const double* x0_host, const double* dyn_shared_host,
                                  double set_point, const cpmpc_step_host_outputs* out);
/* cpmpc_step_batch over all shards with the data resident in HBM: every array of `in` (x0 [4][B]; dyn [9][B], set_point
 * [B], terminal_weights [4][B] when given) and every non-NULL output ([N][B] u, [N][4][B] predicted, [B] status /
 * iterations / ls_evals / final_cost / final_eq_l1, [dim][B] guess / solution) lives on the ROOT device (shard 0's), in
 * the handle's dtype.  Asynchronous on `stream` (a stream of the root device): slices travel to and from the other
 * shards' devices by peer copies ordered with events; on a failure part-way the call waits for the copies of the shards
 * already started before it returns the error. */
int cpmpc_sharded_step_batch_ex(cpmpc_sharded* s, int64_t B, const cpmpc_step_inputs* in,
                                const cpmpc_step_outputs* out, void* stream);
/* shared parameters only (round 3's form) */
int cpmpc_sharded_step_batch(cpmpc_sharded* s, int64_t B, const void* x0, const double* dyn_shared_host,
                             double set_point, const cpmpc_step_outputs* out, void* stream);

/* ---- measurement ------------------------------------------------------------------------------ */

enum {
  CPMPC_KERNEL_PREPARE = 0,   /* guess + FillInitialGuess */
  CPMPC_KERNEL_LINEARIZE = 1, /* RK4 + Jacobians over every shooting interval */
  CPMPC_KERNEL_QP_LS = 2,     /* structured QP + merit line search */
  CPMPC_KERNEL_FINALIZE = 3,  /* ComputePredictedStates + outputs */
  CPMPC_KERNEL_FUSED = 4,     /* all SQP iterations in one launch (fused pipeline) */
  CPMPC_KERNEL_COUNT = 5
};

/* Two implementations of the SQP iterations, same algorithm and decisions (results agree to rounding):
 *   CPMPC_PIPELINE_SPLIT  linearize_kernel + qp_ls_kernel per iteration, one problem per lane, the
 *                         sensitivities stream through HBM between the kernels (any configuration);
 *   CPMPC_PIPELINE_FUSED  one launch for all iterations, a problem spread over its S-1 shooting intervals' lanes,
 *                         sensitivities and QP factors in registers and LDS.  Both models; any state_spacing whose
 *                         interval count S-1 is one of 2, 4, 5, 8, 10, 16 and whose per-wave LDS fits 64 KB;
 *   CPMPC_PIPELINE_AUTO   fused where built, else split (default).  Two exceptions, both fp64: the 6-state model where fewer
 *                         than three of its fused waves fit a CU's LDS (state_spacing 20), and -- round 6 -- any handle whose
 *                         horizon is beyond cpmpc_max_parity_horizon(): the split QP kernel runs two refinement passes there
 *                         (cpmpc_horizon_beyond_parity).
 * The handle's options hold in either pipeline: cpmpc_refines_qp(), cpmpc_wide_qp() and cpmpc_get_solver_opts() tell what it
 * uses, cpmpc_get_pipeline() which pipeline its next step takes.
 * Returns CPMPC_ERR_UNSUPPORTED if FUSED is requested for a configuration it is not built for. */
enum { CPMPC_PIPELINE_AUTO = 0, CPMPC_PIPELINE_SPLIT = 1, CPMPC_PIPELINE_FUSED = 2 };
int cpmpc_set_pipeline(cpmpc_solver* s, int mode);
/* Fused pipeline with exit tolerances enabled: run `first_iterations` SQP iterations on every problem, then
 * compact the problems that are still iterating into dense waves before every further `next_iterations` (default
 * 2 and 1, applied to batches larger than one round of resident waves; an explicit call applies to every batch size;
 * 0 disables and gives the single launch).  A speed setting only: results do not change.  Which staging is fastest
 * depends on how the iteration counts are spread: tools/steady_state.py measures a workload under each. */
int cpmpc_set_compaction(cpmpc_solver* s, int first_iterations, int next_iterations);
/* Without an explicit cpmpc_set_compaction the stages are planned per step from how many iterations the problems of the
 * most recent FINISHED step needed (a histogram finalize leaves in host-mapped memory; read without synchronisation):
 * a settled closed loop, where every controller stops after one iteration, runs two launches instead of seven; a batch
 * whose iteration counts spread gets a cut wherever enough problems have stopped to pay for a compaction.  Speed only:
 * results never depend on the plan, but the SEQUENCE OF LAUNCHES of a default-staged step does depend on unsynchronised
 * timing (which earlier step's counts have reached host memory), so kernel counts in a trace differ from run to run, and
 * a HIP-graph capture bakes in the plan of the moment and the step's stamp -- a tick captured during a transient replays
 * its seven launches for ever, one captured when settled is bound by its stragglers after a disturbance.  Before capturing
 * a graph, or for reproducible traces, fix the pattern with cpmpc_set_compaction.  max_iterations above 15 always takes
 * the fixed pattern (the histogram has 16 bins, the last one standing for "ran to the cap").
 * This reads back the plan of the last step: bounds[0] = 0 < ... < bounds[n] = max_iterations, returns n (the number of
 * launches of the fused kernel), -1 on a bad argument. */
int cpmpc_get_stage_plan(const cpmpc_solver* s, int32_t* bounds, int capacity);
/* The planner alone (no device, no handle; for tests and tools): hist[16], hist[k] = problems that ran k iterations (the
 * last bin collects every larger count), for a batch of B problems with `intervals` = S - 1 shooting intervals.  Writes
 * bounds[0 .. n], returns n; -1 on a bad argument (max_iterations must be 1 .. 15, capacity >= max_iterations + 1). */
int cpmpc_plan_stages_from_histogram(const int64_t* hist, int64_t B, int max_iterations, int intervals, int dtype,
                                     int window_length, int32_t* bounds, int capacity);
int cpmpc_get_pipeline(const cpmpc_solver* s); /* the one a step will actually use: SPLIT or FUSED */

/* When enabled, every kernel launch of cpmpc_step_batch is bracketed by HIP events on the launch
 * stream; cpmpc_profile_read synchronises those events and returns the accumulated device time. */
int cpmpc_profile_enable(cpmpc_solver* s, int on);
int cpmpc_profile_reset(cpmpc_solver* s);
int cpmpc_profile_read(cpmpc_solver* s, int kernel, double* total_ms, int64_t* launches);
const char* cpmpc_kernel_name(int kernel);

#ifdef __cplusplus
}
#endif
#endif
