/* cpmpc_oracle_ld.c -- the SAME restatement (cpmpc_oracle.c, included below) with every `double` of its arithmetic
 * and storage replaced by the x87 `long double` (64-bit significand: ~2 000x finer than double).
 *
 * TEST INFRASTRUCTURE ONLY, like the rest of oracle/.  Purpose: an arbiter.  The GPU kernels and the double oracle
 * solve each QP by different (both backward-stable) algorithms, so their results differ by rounding, and the SQP
 * iteration on a far-from-converged swing-up problem amplifies that difference.  When a lane of a large batch ends
 * up more than 1e-5 from the double oracle, this build says which of the two moved: it runs the same algorithm with
 * the same constants (the literals and M_PI stay the double values) in extended precision, so that
 * |u_gpu - u_ld| and |u_oracle - u_ld| measure each implementation's own rounding sensitivity on that problem.
 *
 * The public structs (orc_opt_params, orc_solver_opts, orc_solver_summary) keep their double fields; the functions
 * are renamed orcld_* and take long double arrays; orcld_step_batch_cold_d below is the double-typed entry point the
 * tests call. */
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

#define orc_default_opt_params orcld_default_opt_params
#define orc_default_solver_opts orcld_default_solver_opts
#define orc_dynamics orcld_dynamics
#define orc_dynamics_generated orcld_dynamics_generated
#define orc_dynamics_double orcld_dynamics_double
#define orc_energy_double orcld_energy_double
#define orc_mod_pi orcld_mod_pi
#define orc_model_np orcld_model_np
#define orc_model_nx orcld_model_nx
#define orc_opt_create orcld_opt_create
#define orc_opt_create_model orcld_opt_create_model
#define orc_opt_destroy orcld_opt_destroy
#define orc_opt_dim orcld_opt_dim
#define orc_opt_has_previous_solution orcld_opt_has_previous_solution
#define orc_opt_reset orcld_opt_reset
#define orc_opt_set_previous_solution orcld_opt_set_previous_solution
#define orc_opt_step orcld_opt_step
#define orc_problem_eval orcld_problem_eval
#define orc_problem_eval_model orcld_problem_eval_model
#define orc_retract_model orcld_retract_model
#define orc_problem_shape orcld_problem_shape
#define orc_problem_shape_model orcld_problem_shape_model
#define orc_qp_solve orcld_qp_solve
#define orc_retract orcld_retract
#define orc_rk4 orcld_rk4
#define orc_rk4_model orcld_rk4_model
#define orc_rk4_no_jacobians orcld_rk4_no_jacobians
#define orc_shooting_constraint orcld_shooting_constraint
#define orc_shooting_constraint_model orcld_shooting_constraint_model
#define orc_sim_step orcld_sim_step
#define orc_sim_step_model orcld_sim_step_model
#define orc_solve orcld_solve
#define orc_step_batch_cold orcld_step_batch_cold
#define orc_step_batch_cold_model orcld_step_batch_cold_model
#define orc_optimization orcld_optimization
typedef struct orcld_optimization orcld_optimization;

#define ORC_EPS LDBL_EPSILON
#define double long double
#define sin sinl
#define cos cosl
#define tanh tanhl
#define sqrt sqrtl
#define fabs fabsl
#define fmod fmodl
#include "cpmpc_oracle.c"
#undef double
#undef sin
#undef cos
#undef tanh
#undef sqrt
#undef fabs
#undef fmod

/* double in, double out; everything in between in long double.  Same argument meaning as orc_step_batch_cold_model;
 * eq_l1_out (nullable) receives the final |c|_1 of each problem. */
int orcld_step_batch_cold_d(int model, const orc_opt_params* p, const orc_solver_opts* o, const double* dyn,
                            double set_point, int64_t B, const double* x0_soa, double* u_out_soa, int32_t* status,
                            int32_t* iters, int32_t* ls_evals, double* eq_l1_out, int num_threads) {
  const int N = (int)p->window_length;
  const int n = orcld_model_nx(model), np = orcld_model_np(model);
  long double dyn_ld[16];
  for (int i = 0; i < np; ++i) dyn_ld[i] = dyn[i];
  int used = 1;
#ifdef _OPENMP
  if (num_threads > 0) omp_set_num_threads(num_threads);
  used = omp_get_max_threads();
#else
  (void)num_threads;
#endif
#pragma omp parallel
  {
    orcld_optimization* opt = orcld_opt_create_model(p, o, model);
    long double* u = (long double*)malloc(sizeof(long double) * (size_t)N);
#pragma omp for schedule(dynamic, 16)
    for (int64_t b = 0; b < B; ++b) {
      if (!opt) continue;
      orcld_opt_reset(opt);
      long double x0[ORC_MAXNX];
      for (int t = 0; t < n; ++t) x0[t] = x0_soa[(int64_t)t * B + b];
      orc_solver_summary sum;
      orcld_opt_step(opt, x0, dyn_ld, (long double)set_point, u, NULL, NULL, NULL, &sum);
      for (int k = 0; k < N; ++k) u_out_soa[(int64_t)k * B + b] = (double)u[k];
      if (status) status[b] = sum.termination_state;
      if (iters) iters[b] = sum.iterations;
      if (ls_evals) ls_evals[b] = sum.line_search_evals;
      if (eq_l1_out) eq_l1_out[b] = sum.final_eq_l1;
    }
    free(u);
    orcld_opt_destroy(opt);
  }
  return used;
}
