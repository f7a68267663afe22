/* cpmpc_oracle_f32.c -- the SAME restatement (cpmpc_oracle.c, included below) with every `double` of its arithmetic and
 * storage replaced by `float`: the single-precision twin of the CPU check (round 5; cpmpc_oracle_ld.c is the trick in the
 * other direction).  Compiled with -fsingle-precision-constant so that the literals do not promote the arithmetic.
 *
 * TEST INFRASTRUCTURE ONLY, like the rest of oracle/.  Purpose: the CPMPC_F32 kernels take decisions the double check
 * can never take -- the first-order exit test's rounding floor (exit_defect_floor x state_spacing x eps x ... is 1.5e-5 in
 * float and is a single-precision rule: the double builds do not apply it) -- and after five SQP iterations on a cold-start
 * swing-up a float solve is nowhere near a double one lane by lane (median 2e-4, tail of order one).  This build runs the
 * same algorithm at the kernels' precision, so that
 *   - the STATISTICS of a float closed loop (status and iteration histograms per tick) can be held against a CPU answer
 *     (tests/test_gpu_round5.py), and
 *   - the float headline's control sequences can be compared with a same-precision solve (bench.py parity_vs_f32_check).
 * It is not bitwise the kernels' arithmetic (dense KKT solve instead of the condensed one, libm's sinf / tanhf instead of
 * v_sin_f32 / v_exp_f32, other summation orders): agreement is statistical by construction.
 *
 * What stays in double, as in the float kernels (csrc/wide.hpp carries the terminal Schur complement in double): the dense
 * KKT solve (ORC_WIDE).  The public structs keep their double fields; functions are renamed orcf_*. */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "cpmpc_oracle.h" /* the double-typed declarations of orc_*: parsed BEFORE the substitutions below */

#define orc_default_opt_params orcf_default_opt_params
#define orc_default_solver_opts orcf_default_solver_opts
#define orc_dynamics orcf_dynamics
#define orc_dynamics_generated orcf_dynamics_generated
#define orc_dynamics_double orcf_dynamics_double
#define orc_energy_double orcf_energy_double
#define orc_mod_pi orcf_mod_pi
#define orc_model_np orcf_model_np
#define orc_model_nx orcf_model_nx
#define orc_opt_create orcf_opt_create
#define orc_opt_create_model orcf_opt_create_model
#define orc_opt_destroy orcf_opt_destroy
#define orc_opt_dim orcf_opt_dim
#define orc_opt_has_previous_solution orcf_opt_has_previous_solution
#define orc_opt_reset orcf_opt_reset
#define orc_opt_set_previous_solution orcf_opt_set_previous_solution
#define orc_opt_step orcf_opt_step
#define orc_problem_eval orcf_problem_eval
#define orc_problem_eval_model orcf_problem_eval_model
#define orc_retract_model orcf_retract_model
#define orc_problem_shape orcf_problem_shape
#define orc_problem_shape_model orcf_problem_shape_model
#define orc_qp_solve orcf_qp_solve
#define orc_retract orcf_retract
#define orc_rk4 orcf_rk4
#define orc_rk4_model orcf_rk4_model
#define orc_rk4_no_jacobians orcf_rk4_no_jacobians
#define orc_shooting_constraint orcf_shooting_constraint
#define orc_shooting_constraint_model orcf_shooting_constraint_model
#define orc_sim_step orcf_sim_step
#define orc_sim_step_model orcf_sim_step_model
#define orc_solve orcf_solve
#define orc_step_batch_cold orcf_step_batch_cold
#define orc_step_batch_cold_model orcf_step_batch_cold_model
#define orc_optimization orcf_optimization
typedef struct orcf_optimization orcf_optimization;

typedef double orcf_true_double;
#define ORC_WIDE orcf_true_double
#define ORC_EPS FLT_EPSILON
#define double float
#define sin sinf
#define cos cosf
#define tanh tanhf
#define sqrt sqrtf
#define fabs fabsf
#define fmod fmodf
#include "cpmpc_oracle.c"
#undef double
#undef sin
#undef cos
#undef tanh
#undef sqrt
#undef fabs
#undef fmod

static int orcf_threads(int num_threads) {
  int used = 1;
#ifdef _OPENMP
  if (num_threads > 0) omp_set_num_threads(num_threads);
  used = omp_get_max_threads();
#else
  (void)num_threads;
#endif
  return used;
}

/* double in, double out; everything in between in float.  Same argument meaning as orc_step_batch_cold_model;
 * eq_l1_out (nullable) receives the final |c|_1 of each problem. */
int orcf_step_batch_cold_d(int model, const orc_opt_params* p, const orc_solver_opts* o, const double* dyn,
                           double set_point, int64_t B, const double* x0_soa, double* u_out_soa, int32_t* status,
                           int32_t* iters, int32_t* ls_evals, double* eq_l1_out, int num_threads) {
  const int N = (int)p->window_length;
  const int n = orcf_model_nx(model), np = orcf_model_np(model);
  float dyn_f[16];
  for (int i = 0; i < np; ++i) dyn_f[i] = (float)dyn[i];
  const int used = orcf_threads(num_threads);
#pragma omp parallel
  {
    orcf_optimization* opt = orcf_opt_create_model(p, o, model);
    float* u = (float*)malloc(sizeof(float) * (size_t)N);
#pragma omp for schedule(dynamic, 16)
    for (int64_t b = 0; b < B; ++b) {
      if (!opt) continue;
      orcf_opt_reset(opt);
      float x0[ORC_MAXNX];
      for (int t = 0; t < n; ++t) x0[t] = (float)x0_soa[(int64_t)t * B + b];
      orc_solver_summary sum;
      orcf_opt_step(opt, x0, dyn_f, (float)set_point, u, NULL, NULL, NULL, &sum);
      for (int k = 0; k < N; ++k) u_out_soa[(int64_t)k * B + b] = (double)u[k];
      if (status) status[b] = sum.termination_state;
      if (iters) iters[b] = sum.iterations;
      if (ls_evals) ls_evals[b] = sum.line_search_evals;
      if (eq_l1_out) eq_l1_out[b] = sum.final_eq_l1;
    }
    free(u);
    orcf_opt_destroy(opt);
  }
  return used;
}

/* B independent controllers in closed loop, all in float: per tick Optimization::Step (warm-started after the first,
 * optimization.cc:39-97) -> apply u_0 -> Simulator::Step(dt = control_dt) (simulator.cc:11-36), as
 * optimization_test.cc:39-61 does for one.  status_out / iters_out: [ticks][B] (int8), state_out: [nx][B] after the last
 * tick.  Returns the threads used. */
int orcf_closed_loop_d(int model, const orc_opt_params* p, const orc_solver_opts* o, const double* dyn, double set_point,
                       int64_t B, const double* x0_soa, int ticks, int8_t* status_out, int8_t* iters_out,
                       double* state_out_soa, int num_threads) {
  const int N = (int)p->window_length;
  const int n = orcf_model_nx(model), np = orcf_model_np(model);
  float dyn_f[16];
  for (int i = 0; i < np; ++i) dyn_f[i] = (float)dyn[i];
  const int used = orcf_threads(num_threads);
#pragma omp parallel
  {
    orcf_optimization* opt = orcf_opt_create_model(p, o, model);
    float* u = (float*)malloc(sizeof(float) * (size_t)N);
#pragma omp for schedule(dynamic, 4)
    for (int64_t b = 0; b < B; ++b) {
      if (!opt) continue;
      orcf_opt_reset(opt);
      float x[ORC_MAXNX];
      for (int t = 0; t < n; ++t) x[t] = (float)x0_soa[(int64_t)t * B + b];
      for (int k = 0; k < ticks; ++k) {
        orc_solver_summary sum;
        orcf_opt_step(opt, x, dyn_f, (float)set_point, u, NULL, NULL, NULL, &sum);
        if (status_out) status_out[(int64_t)k * B + b] = (int8_t)sum.termination_state;
        if (iters_out) iters_out[(int64_t)k * B + b] = (int8_t)sum.iterations;
        orcf_sim_step_model(model, dyn_f, (float)p->control_dt, u[0], x);
      }
      if (state_out_soa)
        for (int t = 0; t < n; ++t) state_out_soa[(int64_t)t * B + b] = (double)x[t];
    }
    free(u);
    orcf_opt_destroy(opt);
  }
  return used;
}
