/*
 * cpmpc_oracle.h -- CPU restatement ("oracle") of the cart-pole MPC hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as the
 * checker.  The product path (cart-pole-mpc_amd/) never links, imports or calls it.
 *
 * What it restates (all file:line citations are into the reference tree, /root/reference):
 *   - gen::single_pendulum_dynamics      optimization/single_pendulum_dynamics.hpp:13-186
 *     (re-derived from the Lagrangian spec symbolic/dynamics_single.py:58-143, not transcribed)
 *   - runge_kutta_4th_order<D>           optimization/integration.hpp:13-49
 *   - runge_kutta_4th_order_no_jacobians optimization/integration.hpp:52-62
 *   - mod_pi                             optimization/integration.hpp:65-73
 *   - CreateDynamicalConstraint          optimization/optimization.cc:99-160
 *   - BuildProblem / MapKey / costs      optimization/optimization.cc:27-37,162-331
 *   - Optimization::Step, FillInitialGuess, ComputePredictedStates
 *                                        optimization/optimization.cc:39-97,333-371
 *   - Simulator::Step/SubStep            optimization/simulator.cc:11-36
 *
 * PARITY STATUS
 *   - dynamics / RK4 / mod_pi: pinned by the known-answer values of SURVEY.md section 8(c) (taken
 *     from the reference's own headers), by an independent SymPy derivation
 *     (tests/golden/gen_dynamics_golden.py) and by the reference's tests re-stated in tests/.
 *   - SQP/QP solver: the reference delegates to mini_opt::ConstrainedNonlinearLeastSquares
 *     (optimization.cc:73-85,303-330), an un-vendored submodule (.gitmodules:13-15, pinned SHA
 *     unrecoverable, directory empty).  The solver restated here is THIS REPO'S specification
 *     (DESIGN.md section 4).  Iterate-level parity with mini_opt: **parity unpinned**.
 *     It is pinned only by the reference's convergence-property test
 *     (optimization/optimization_test.cc:12-77) re-stated in tests/test_oracle_closed_loop.py.
 *   - The reference itself cannot be built here (needs Eigen, mini_opt, fmt: all absent), so there
 *     is no oracle/_ref.
 */
#ifndef CPMPC_ORACLE_H
#define CPMPC_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Mirrors pendulum::OptimizationParams field-for-field (optimization/optimization.hpp:12-53). */
typedef struct orc_opt_params {
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
} orc_opt_params;

/* Knobs of this repo's SQP specification.  The reference sets only
 * max_line_search_iterations = 5 explicitly (optimization.cc:76); the rest are this repo's. */
typedef struct orc_solver_opts {
  int32_t max_line_search_iterations; /* 5 */
  double armijo_c1;                   /* 1e-4 */
  double ls_shrink_max;               /* 0.5: upper safeguard of the interpolated step */
  double ls_shrink_min;               /* 0.1: lower safeguard */
  double ls_alpha_growth;             /* 2: next search starts at min(1, this * last step); 0 = off */
  double penalty_rho;                 /* 0.1 */
  double lambda_initial;              /* 0 */
  double lambda_failure_init;         /* 1e-2 */
  double lambda_scale_up;             /* 10 */
  double lambda_scale_down;           /* 0.1 */
  double lambda_min;                  /* 1e-8: below this a decayed lambda snaps to 0 */
  double lambda_max;                  /* 1e6 */
  double b_x_limit;                   /* 5.0   (optimization.cc:320) */
  double u_limit;                     /* 300.0 (optimization.cc:327) */
  double ls_alpha_growth_backtracked; /* growth used instead of ls_alpha_growth when the accepted search had to backtrack */
  double full_step_below; /* a QP step with |dz|_inf <= this is taken in full without the merit test (0 disables) */
  double exit_defect_floor; /* the first-order exit test counts |c|_1 as zero up to this x state_spacing x eps x sum |x_s|
                             * over the shooting nodes: the rounding of the rollout (default 2, 0 disables) */
} orc_solver_opts;

/* Termination states; names follow mini_opt::NLSTerminationState as used by the reference
 * (optimization/optimization_test.cc:44-46). */
enum {
  ORC_TERM_NONE = 0,
  ORC_TERM_MAX_ITERATIONS = 1,
  ORC_TERM_SATISFIED_ABSOLUTE_TOL = 2,
  ORC_TERM_SATISFIED_RELATIVE_TOL = 3,
  ORC_TERM_SATISFIED_FIRST_ORDER_TOL = 4,
  ORC_TERM_QP_INDEFINITE = 5,
  ORC_TERM_USER_CALLBACK = 6,
  ORC_TERM_MAX_LAMBDA = 7,
  ORC_TERM_NON_FINITE = 8
};

typedef struct orc_solver_summary {
  int32_t termination_state;
  int32_t iterations;        /* QP solves performed */
  int32_t line_search_evals; /* total merit evaluations */
  int32_t failed_steps;      /* line searches that exhausted their trials */
  double initial_cost;       /* 1/2 |r|^2 at the guess */
  double initial_eq_l1;      /* |c|_1 at the guess */
  double final_cost;
  double final_eq_l1;
  double final_penalty;
  double final_lambda;
} orc_solver_summary;

void orc_default_opt_params(orc_opt_params* p);
void orc_default_solver_opts(orc_solver_opts* o);

/* ---- L0 / L1: dynamics, integrators ------------------------------------------------------- */

/* params = {m_b, m_1, l_1, g, mu_b, v_mu_b, c_d_1, x_s, k_s} (optimization/structs.hpp:8-41).
 * x = {b_x, th_1, b_x_dot, th_1_dot}.  Jx is 4x4 row-major, Ju is 4; either may be NULL. */
void orc_dynamics(const double params[9], const double x[4], double u, const double f_base[2],
                  const double f_mass[2], double f_out[4], double* Jx, double* Ju);
/* the same, evaluated by the code tools/gen_dynamics.py generates (oracle/single_pendulum_gen.inc) */
void orc_dynamics_generated(const double params[9], const double x[4], double u, const double f_base[2],
                            const double f_mass[2], double f_out[4], double* Jx, double* Ju);

void orc_rk4(const double params[9], const double x[4], double u, double h,
             const double f_base[2], const double f_mass[2], double x_new[4], double A[16],
             double B[4]);

void orc_rk4_no_jacobians(const double params[9], const double x[4], double u, double h,
                          const double f_base[2], const double f_mass[2], double x_new[4]);

double orc_mod_pi(double angle);

/* Shooting defect of one interval (optimization.cc:99-160).  vars = [x_k(4), x_k+1(4), u(sp)].
 * J (nullable) is 4 x (8+sp) row-major. */
void orc_shooting_constraint(const double params[9], int spacing, double dt, const double* vars,
                             double err[4], double* J);

/* ---- L3: problem assembly (BuildProblem) --------------------------------------------------- */

/* Row counts for a given parameter set: dimension, equality rows, cost rows. */
void orc_problem_shape(const orc_opt_params* p, int* dim, int* n_eq, int* n_cost);

/* Evaluate every residual at z.  r_cost[n_cost], c_eq[n_eq]; J_cost (n_cost x dim) and A_eq
 * (n_eq x dim), row-major, nullable.  Row order follows BuildProblem. */
void orc_problem_eval(const orc_opt_params* p, const double dyn[9], const double x_current[4],
                      double set_point, double u_prev, const double* z, double* r_cost,
                      double* c_eq, double* J_cost, double* A_eq);

/* The retraction of optimization.cc:309-329: z <- clamp/mod(z + alpha dz). */
void orc_retract(const orc_opt_params* p, const orc_solver_opts* o, const double* z,
                 const double* dz, double alpha, double* z_out);
/* the same two for either model (ORC_MODEL_*) */
void orc_problem_eval_model(int model, const orc_opt_params* p, const double* dyn, const double* x_current,
                            double set_point, double u_prev, const double* z, double* r_cost, double* c_eq,
                            double* J_cost, double* A_eq);
void orc_retract_model(int model, const orc_opt_params* p, const orc_solver_opts* o, const double* z,
                       const double* dz, double alpha, double* z_out);

/* One equality-constrained Gauss-Newton QP on the full space, dense KKT + LU.
 * Returns 0 on success, nonzero if the KKT matrix is singular. */
int orc_qp_solve(int dim, int n_eq, int n_cost, int n_u, const double* J_cost, const double* r_cost,
                 const double* A_eq, const double* c_eq, double lambda, double* dz);

/* ---- Optimization (stateful, mirrors pendulum::Optimization) ------------------------------- */

typedef struct orc_optimization orc_optimization;

orc_optimization* orc_opt_create(const orc_opt_params* p, const orc_solver_opts* o /*nullable*/);
void orc_opt_destroy(orc_optimization* opt);
void orc_opt_reset(orc_optimization* opt);
void orc_opt_set_previous_solution(orc_optimization* opt, const double* z, int n);
int orc_opt_has_previous_solution(const orc_optimization* opt);

/* One re-plan.  u_out[N]; predicted_out[N*4] (state-major per step), guess_out[dim] = the guess
 * actually handed to the solver, z_out[dim] = solver variables; any output may be NULL. */
int orc_opt_step(orc_optimization* opt, const double* state /*[nx]*/, const double* dyn /*[np]*/,
                 double set_point, double* u_out, double* predicted_out, double* guess_out,
                 double* z_out, orc_solver_summary* summary);

/* Solve from an explicit guess (no warm-start bookkeeping); used by unit tests. */
int orc_solve(const orc_opt_params* p, const orc_solver_opts* o, const double dyn[9],
              const double x_current[4], double set_point, double u_prev, const double* guess,
              double* z_out, orc_solver_summary* summary);

/* ---- Simulator (plant) --------------------------------------------------------------------- */
void orc_sim_step(const double params[9], double dt, double u, const double f_base[2],
                  const double f_mass[2], double state[4]);

/* ---- second model: cart + double pendulum (BASELINE config 5) ------------------------------ */
/* Specified by symbolic/dynamics_double.py:25-148; the reference ships neither a generated header
 * nor an optimizer for it (optimization.cc:197-199 hard-codes 4 states), so everything about this
 * model is pinned by golden vectors from an independent SymPy/mpmath evaluation, finite differences
 * and energy conservation only -- **parity unpinned** by the reference.
 * State {b_x, th_1, th_2, b_x', th_1', th_2'}, params {m_b, m_1, m_2, l_1, l_2, g}.  The optimizer
 * applies th_final / th_dot_final weights to both poles; targets are both poles upright. */
enum { ORC_MODEL_SINGLE = 0, ORC_MODEL_DOUBLE = 1 };
int orc_model_nx(int model);
int orc_model_np(int model);
void orc_dynamics_double(const double params[6], const double x[6], double u, const double f_base[2],
                         const double f_mass[2], double f_out[6], double* Jx /*6x6*/, double* Ju /*6*/);
double orc_energy_double(const double params[6], const double x[6]);
void orc_rk4_model(int model, const double* params, const double* x, double u, double h, double* x_new,
                   double* A /*nx*nx or NULL*/, double* B /*nx or NULL*/);
void orc_shooting_constraint_model(int model, const double* params, int spacing, double dt,
                                   const double* vars, double* err, double* J);
void orc_problem_shape_model(int model, const orc_opt_params* p, int* dim, int* n_eq, int* n_cost);
orc_optimization* orc_opt_create_model(const orc_opt_params* p, const orc_solver_opts* o, int model);
int orc_opt_dim(const orc_optimization* opt);
void orc_sim_step_model(int model, const double* params, double dt, double u, double* state);
int orc_step_batch_cold_model(int model, const orc_opt_params* p, const orc_solver_opts* o,
                              const double* dyn, double set_point, int64_t B, const double* x0_soa,
                              double* u_out_soa, double* pred_out_soa, int32_t* status, int32_t* iters,
                              int num_threads);

/* ---- batch driver (cpu_baseline leg of bench.py; OpenMP over problems) --------------------- */
/* SoA inputs like the product C-ABI: x0[4][B]; cold start for every problem.
 * u_out[N][B], pred_out[N][4][B] (nullable), status[B] (nullable).  Returns threads used. */
int orc_step_batch_cold(const orc_opt_params* p, const orc_solver_opts* o, const double dyn[9],
                        double set_point, int64_t B, const double* x0_soa, double* u_out_soa,
                        double* pred_out_soa, int32_t* status, int32_t* iters, int num_threads);

#ifdef __cplusplus
}
#endif
#endif
