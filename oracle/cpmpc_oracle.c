/*
 * cpmpc_oracle.c -- CPU restatement ("oracle") of the cart-pole MPC hot path.  fp64, scalar, C99.
 *
 * TEST INFRASTRUCTURE ONLY -- see cpmpc_oracle.h for the rules and the parity status
 * (dynamics/RK4 pinned; SQP solver: parity unpinned vs. mini_opt, which is absent).
 *
 * Every function cites the reference file:line it follows (paths relative to /root/reference).
 * Written from the reference's behaviour and from the Lagrangian specification; no reference
 * source is copied.
 */
#include "cpmpc_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* machine epsilon of the arithmetic this file is compiled in (cpmpc_oracle_ld.c sets the long double one) */
#ifndef ORC_EPS
#define ORC_EPS DBL_EPSILON
#endif

/* the type the dense KKT solve is carried in: the arithmetic's own, except in the single-precision twin
 * (cpmpc_oracle_f32.c), which keeps it in double like the float kernels keep their terminal system (csrc/wide.hpp) */
#ifndef ORC_WIDE
#define ORC_WIDE double
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------------------------------- */
/* defaults                                                                                     */
/* ------------------------------------------------------------------------------------------- */

/* optimization/optimization.hpp:12-48 */
void orc_default_opt_params(orc_opt_params* p) {
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

/* This repo's solver specification; max_line_search_iterations from
 * optimization/optimization.cc:76, clamps from optimization.cc:320,327. */
void orc_default_solver_opts(orc_solver_opts* o) {
  o->max_line_search_iterations = 5;
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
  o->b_x_limit = 5.0;
  o->u_limit = 300.0;
  o->ls_alpha_growth_backtracked = 2.0;
  o->full_step_below = 1.0e-4;
  o->exit_defect_floor = 2.0;
}

/* ------------------------------------------------------------------------------------------- */
/* L0: dynamics                                                                                 */
/* ------------------------------------------------------------------------------------------- */

/*
 * Forward dynamics of the cart + point-mass pole.
 * Follows optimization/single_pendulum_dynamics.hpp:13-186 (machine generated) by re-deriving the
 * Euler-Lagrange equations of symbolic/dynamics_single.py:58-143:
 *
 *   p1 = (b_x + l cos th, l sin th),  T = 1/2 m_b b_x'^2 + 1/2 m_1 |p1'|^2,  V = g m_1 l sin th
 *
 *   [ m_b+m_1      -m_1 l sin th ] [b_x''] = [ F_b  ]
 *   [ -m_1 l sin th   m_1 l^2    ] [th'' ]   [ F_th ]
 *
 *   F_b  = u + f_base.x + f_mass.x + F_fric + F_spring - dD/db_x' + m_1 l th'^2 cos th
 *   F_th = l(-f_mass.x sin th + f_mass.y cos th) - g m_1 l cos th - dD/dth'
 *   F_fric   = -mu_b (m_1+m_b) g tanh(b_x' / max(v_mu_b, 1e-6))      (dynamics_single.py:100-103)
 *   D        = (1/6) c_d |p1'|^3   if |p1'|^2 > 0 else 0              (dynamics_single.py:105-111)
 *   F_spring = -k_s max(0, b_x - x_s) + k_s max(0, -x_s - b_x)        (dynamics_single.py:113-115)
 *
 * f_base.y never enters the equations (the base moves on x only), as in the generated header.
 * Jx (4x4 row-major) and Ju (4) are optional, like the generated function's J_x / J_u.
 */
void orc_dynamics(const double params[9], const double x[4], double u, const double f_base[2],
                  const double f_mass[2], double f_out[4], double* Jx, double* Ju) {
  const double m_b = params[0], m_1 = params[1], L = params[2], g = params[3];
  const double mu = params[4], v_mu_in = params[5], cd = params[6], xs = params[7], ks = params[8];
  const double bx = x[0], th = x[1], v = x[2], w = x[3];
  const double s = sin(th), c = cos(th);
  const double mt = m_1 + m_b;

  /* bumper springs; the comparisons are strict `0 < arg` as in the generated branches */
  const double e_r = bx - xs;
  const double e_l = -(bx + xs);
  const int on_r = 0.0 < e_r;
  const int on_l = 0.0 < e_l;
  const double F_s = ks * ((on_l ? e_l : 0.0) - (on_r ? e_r : 0.0));
  const double dFs_dbx = ks * ((on_l ? -1.0 : 0.0) - (on_r ? 1.0 : 0.0));

  /* smoothed Coulomb friction on the base */
  const double v_mu = (1.0e-6 < v_mu_in) ? v_mu_in : 1.0e-6;
  const double inv_v_mu = 1.0 / v_mu;
  const double tv = tanh(v * inv_v_mu);
  const double fr = (mt * mu) * -g;
  const double F_f = tv * fr;
  const double dFf_dv = inv_v_mu * (1.0 - tv * tv) * fr;

  /* air drag on the pole mass: velocity of the mass */
  const double vx = v - L * w * s;
  const double vy = L * w * c;
  const double n2 = vx * vx + vy * vy;
  const double n = sqrt(n2);
  const int on_d = 0.0 < n2;
  const double e = c * vy - s * vx; /* (1/l) * p1' . dp1'/dth' */
  const double Dx = on_d ? 0.5 * cd * n * vx : 0.0;
  const double Dth = on_d ? 0.5 * cd * L * n * e : 0.0;

  const double F_b = u + f_base[0] + f_mass[0] + F_f + F_s - Dx + m_1 * L * w * w * c;
  const double F_th = L * (-f_mass[0] * s + f_mass[1] * c) - g * m_1 * L * c - Dth;

  const double den = mt - m_1 * s * s;
  const double inv_den = 1.0 / den;
  const double sl = s / L;
  const double kap = mt / (m_1 * L * L);
  const double N_x = F_b + sl * F_th;
  const double N_th = sl * F_b + kap * F_th;
  const double a_x = N_x * inv_den;
  const double a_th = N_th * inv_den;

  f_out[0] = v;
  f_out[1] = w;
  f_out[2] = a_x;
  f_out[3] = a_th;

  if (Jx) {
    /* partials with respect to (th, v, w) */
    const double dvx[3] = {-L * w * c, 1.0, -L * s};
    const double dvy[3] = {-L * w * s, 0.0, L * c};
    const double de[3] = {-s * vy - c * vx, -s, L};
    double dDx[3] = {0, 0, 0}, dDth[3] = {0, 0, 0};
    if (on_d) {
      const double inv_n = 1.0 / n;
      for (int i = 0; i < 3; ++i) {
        const double dn = (vx * dvx[i] + vy * dvy[i]) * inv_n;
        dDx[i] = 0.5 * cd * (dn * vx + n * dvx[i]);
        dDth[i] = 0.5 * cd * L * (dn * e + n * de[i]);
      }
    }
    const double dFb[3] = {-dDx[0] - m_1 * L * w * w * s, dFf_dv - dDx[1],
                           -dDx[2] + 2.0 * m_1 * L * w * c};
    const double dFth[3] = {L * (-f_mass[0] * c - f_mass[1] * s) + g * m_1 * L * s - dDth[0],
                            -dDth[1], -dDth[2]};
    const double dden_dth = -2.0 * m_1 * s * c;
    const double cl = c / L;
    const double dNx[3] = {dFb[0] + cl * F_th + sl * dFth[0], dFb[1] + sl * dFth[1],
                           dFb[2] + sl * dFth[2]};
    const double dNth[3] = {cl * F_b + sl * dFb[0] + kap * dFth[0], sl * dFb[1] + kap * dFth[1],
                            sl * dFb[2] + kap * dFth[2]};
    /* rows 0,1: d(b_x', th')/dx  (single_pendulum_dynamics.hpp:159-166) */
    for (int i = 0; i < 16; ++i) Jx[i] = 0.0;
    Jx[0 * 4 + 2] = 1.0;
    Jx[1 * 4 + 3] = 1.0;
    Jx[2 * 4 + 0] = dFs_dbx * inv_den;
    Jx[2 * 4 + 1] = (dNx[0] - a_x * dden_dth) * inv_den;
    Jx[2 * 4 + 2] = dNx[1] * inv_den;
    Jx[2 * 4 + 3] = dNx[2] * inv_den;
    Jx[3 * 4 + 0] = sl * dFs_dbx * inv_den;
    Jx[3 * 4 + 1] = (dNth[0] - a_th * dden_dth) * inv_den;
    Jx[3 * 4 + 2] = dNth[1] * inv_den;
    Jx[3 * 4 + 3] = dNth[2] * inv_den;
  }
  if (Ju) {
    /* single_pendulum_dynamics.hpp:179-184 */
    Ju[0] = 0.0;
    Ju[1] = 0.0;
    Ju[2] = inv_den;
    Ju[3] = sl * inv_den;
  }
}

/* The same function as emitted by this repo's dynamics generator (tools/gen_dynamics.py: SymPy Lagrangian of
 * symbolic/dynamics_single.py:58-143 -> CSE'd straight-line code).  Exposed so that the tests hold the generated
 * model against the hand-written one above and against the golden vectors; nothing else in the oracle uses it. */
#include "single_pendulum_gen.inc" /* generated by tools/gen_dynamics.py */

void orc_dynamics_generated(const double params[9], const double x[4], double u, const double f_base[2],
                            const double f_mass[2], double f_out[4], double* Jx, double* Ju) {
  const SinglePendulumGenConsts K = single_pendulum_gen_consts(params);
  double a[2], Ja[2][4], Jua[2];
  const int want_j = (Jx != NULL) || (Ju != NULL);
  if (f_base != NULL || f_mass != NULL) {
    const double fbx = f_base ? f_base[0] : 0.0, fmx = f_mass ? f_mass[0] : 0.0, fmy = f_mass ? f_mass[1] : 0.0;
    single_pendulum_gen_accel_ext(&K, x[0], x[1], x[2], x[3], u, fbx, fmx, fmy, a, Ja, Jua, want_j);
  } else {
    single_pendulum_gen_accel_noext(&K, x[0], x[1], x[2], x[3], u, 0.0, 0.0, 0.0, a, Ja, Jua, want_j);
  }
  f_out[0] = x[2];
  f_out[1] = x[3];
  f_out[2] = a[0];
  f_out[3] = a[1];
  if (Jx) {
    for (int i = 0; i < 16; ++i) Jx[i] = 0.0;
    Jx[0 * 4 + 2] = 1.0; /* single_pendulum_dynamics.hpp:159-166 */
    Jx[1 * 4 + 3] = 1.0;
    for (int k = 0; k < 4; ++k) {
      Jx[2 * 4 + k] = Ja[0][k];
      Jx[3 * 4 + k] = Ja[1][k];
    }
  }
  if (Ju) {
    Ju[0] = 0.0;
    Ju[1] = 0.0;
    Ju[2] = Jua[0];
    Ju[3] = Jua[1];
  }
}

/* ------------------------------------------------------------------------------------------- */
/* L0 (second model): cart + double pendulum                                                     */
/* ------------------------------------------------------------------------------------------- */

#include "double_pendulum_gen.inc" /* generated by tools/gen_dynamics.py */

/* Solve the symmetric positive definite 3x3 system M y = b by LDL^T (M row-major). */
static void spd3_factor(const double* M, double L[3], double d[3]) {
  d[0] = M[0];
  L[0] = M[3] / d[0];                       /* l10 */
  L[1] = M[6] / d[0];                       /* l20 */
  d[1] = M[4] - L[0] * L[0] * d[0];
  L[2] = (M[7] - L[1] * L[0] * d[0]) / d[1]; /* l21 */
  d[2] = M[8] - L[1] * L[1] * d[0] - L[2] * L[2] * d[1];
}
static void spd3_solve(const double L[3], const double d[3], const double b[3], double y[3]) {
  const double z0 = b[0];
  const double z1 = b[1] - L[0] * z0;
  const double z2 = b[2] - L[1] * z0 - L[2] * z1;
  y[2] = z2 / d[2];
  y[1] = z1 / d[1] - L[2] * y[2];
  y[0] = z0 / d[0] - L[0] * y[1] - L[1] * y[2];
}

/*
 * Forward dynamics of the cart + two point-mass poles, specified by symbolic/dynamics_double.py:25-148
 * (the reference ships no generated header for it and no optimizer: optimization.cc:197-199).
 * x = {b_x, th_1, th_2, b_x', th_1', th_2'} (dynamics_double.py:117-118,136),
 * params = {m_b, m_1, m_2, l_1, l_2, g} (dynamics_double.py:13-22); no dissipation, no external forces
 * (f_base / f_mass are accepted for signature uniformity and ignored).
 * a = M^-1 F;  da/dx_c = M^-1 (dF/dx_c - dM/dx_c a);  da/du = M^-1 e_0.
 */
void orc_dynamics_double(const double params[6], const double x[6], double u, const double f_base[2],
                         const double f_mass[2], double f_out[6], double* Jx, double* Ju) {
  (void)f_base;
  (void)f_mass;
  double M[9], F[3], dFdx[18], dM1[9], dM2[9], L[3], d[3], a[3];
  double_pendulum_terms(params, x, u, M, F, dFdx, dM1, dM2);
  spd3_factor(M, L, d);
  spd3_solve(L, d, F, a);
  f_out[0] = x[3];
  f_out[1] = x[4];
  f_out[2] = x[5];
  f_out[3] = a[0];
  f_out[4] = a[1];
  f_out[5] = a[2];
  if (Jx) {
    for (int i = 0; i < 36; ++i) Jx[i] = 0.0;
    Jx[0 * 6 + 3] = 1.0;
    Jx[1 * 6 + 4] = 1.0;
    Jx[2 * 6 + 5] = 1.0;
    for (int c = 0; c < 6; ++c) {
      double rhs[3], y[3];
      for (int i = 0; i < 3; ++i) {
        rhs[i] = dFdx[i * 6 + c];
        const double* dM = (c == 1) ? dM1 : ((c == 2) ? dM2 : NULL);
        if (dM)
          for (int k = 0; k < 3; ++k) rhs[i] -= dM[i * 3 + k] * a[k];
      }
      spd3_solve(L, d, rhs, y);
      for (int i = 0; i < 3; ++i) Jx[(3 + i) * 6 + c] = y[i];
    }
  }
  if (Ju) {
    const double e0[3] = {1.0, 0.0, 0.0};
    double y[3];
    spd3_solve(L, d, e0, y);
    Ju[0] = Ju[1] = Ju[2] = 0.0;
    Ju[3] = y[0];
    Ju[4] = y[1];
    Ju[5] = y[2];
  }
}

/* total mechanical energy of the double pendulum (for the conservation test) */
double orc_energy_double(const double params[6], const double x[6]) {
  const double m_b = params[0], m_1 = params[1], m_2 = params[2], l_1 = params[3], l_2 = params[4], g = params[5];
  const double s1 = sin(x[1]), c1 = cos(x[1]), s2 = sin(x[2]), c2 = cos(x[2]);
  const double v = x[3], w1 = x[4], w2 = x[5];
  const double p1x = v - l_1 * s1 * w1, p1y = l_1 * c1 * w1;
  const double p2x = p1x - l_2 * s2 * w2, p2y = p1y + l_2 * c2 * w2;
  const double T = 0.5 * m_b * v * v + 0.5 * m_1 * (p1x * p1x + p1y * p1y) + 0.5 * m_2 * (p2x * p2x + p2y * p2y);
  const double V = g * m_1 * l_1 * s1 + g * m_2 * (l_1 * s1 + l_2 * s2);
  return T + V;
}

/* ------------------------------------------------------------------------------------------- */
/* models                                                                                       */
/* ------------------------------------------------------------------------------------------- */

#define ORC_MAXNX 6
typedef void (*orc_dyn_fn)(const double* p, const double* x, double u, const double* fb, const double* fm,
                           double* f, double* Jx, double* Ju);
typedef struct orc_model {
  int nx;          /* state dimension: positions then velocities */
  int nq;          /* nx / 2; component 0 is the base position, 1..nq-1 are pole angles */
  int np;          /* number of dynamics parameters */
  orc_dyn_fn dyn;
} orc_model;

static const orc_model kModels[2] = {
    {4, 2, 9, (orc_dyn_fn)orc_dynamics},        /* ORC_MODEL_SINGLE */
    {6, 3, 6, (orc_dyn_fn)orc_dynamics_double}, /* ORC_MODEL_DOUBLE */
};
static const orc_model* model_of(int id) { return (id == ORC_MODEL_DOUBLE) ? &kModels[1] : &kModels[0]; }
static int is_angle(const orc_model* m, int t) { return t >= 1 && t < m->nq; }

int orc_model_nx(int model) { return model_of(model)->nx; }
int orc_model_np(int model) { return model_of(model)->np; }

/* ------------------------------------------------------------------------------------------- */
/* L1: integrators                                                                              */
/* ------------------------------------------------------------------------------------------- */

static void mat_mul(int n, const double* a, const double* b, double* out) { /* out = a*b, n x n */
  double t[ORC_MAXNX * ORC_MAXNX];
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      double acc = 0.0;
      for (int k = 0; k < n; ++k) acc += a[i * n + k] * b[k * n + j];
      t[i * n + j] = acc;
    }
  memcpy(out, t, sizeof(double) * (size_t)(n * n));
}

static void mat_vec(int n, const double* a, const double* v, double* out) {
  double t[ORC_MAXNX];
  for (int i = 0; i < n; ++i) {
    double acc = 0.0;
    for (int k = 0; k < n; ++k) acc += a[i * n + k] * v[k];
    t[i] = acc;
  }
  memcpy(out, t, sizeof(double) * (size_t)n);
}

/* optimization/integration.hpp:52-62 */
static void rk4_nj(const orc_model* m, const double* params, const double* x, double u, double h,
                   const double* f_base, const double* f_mass, double* x_new) {
  const int n = m->nx;
  double k1[ORC_MAXNX], k2[ORC_MAXNX], k3[ORC_MAXNX], k4[ORC_MAXNX], xt[ORC_MAXNX];
  m->dyn(params, x, u, f_base, f_mass, k1, NULL, NULL);
  for (int i = 0; i < n; ++i) xt[i] = x[i] + k1[i] * h / 2.0;
  m->dyn(params, xt, u, f_base, f_mass, k2, NULL, NULL);
  for (int i = 0; i < n; ++i) xt[i] = x[i] + k2[i] * h / 2.0;
  m->dyn(params, xt, u, f_base, f_mass, k3, NULL, NULL);
  for (int i = 0; i < n; ++i) xt[i] = x[i] + k3[i] * h;
  m->dyn(params, xt, u, f_base, f_mass, k4, NULL, NULL);
  for (int i = 0; i < n; ++i) x_new[i] = x[i] + (h / 6.0) * (k1[i] + k2[i] * 2.0 + k3[i] * 2.0 + k4[i]);
}

/* optimization/integration.hpp:13-49: state + A = dx+/dx + B = dx+/du by the stage chain rule */
static void rk4_j(const orc_model* m, const double* params, const double* x, double u, double h,
                  const double* f_base, const double* f_mass, double* x_new, double* A, double* B) {
  const int n = m->nx, nn = n * n;
  double k1[ORC_MAXNX], k2[ORC_MAXNX], k3[ORC_MAXNX], k4[ORC_MAXNX], xt[ORC_MAXNX];
  double K1[36], K2[36], K3[36], K4[36]; /* stage Jacobians at the stage arguments */
  double U1[ORC_MAXNX], U2[ORC_MAXNX], U3[ORC_MAXNX], U4[ORC_MAXNX];
  m->dyn(params, x, u, f_base, f_mass, k1, K1, U1);
  for (int i = 0; i < n; ++i) xt[i] = x[i] + k1[i] * h / 2.0;
  m->dyn(params, xt, u, f_base, f_mass, k2, K2, U2);
  for (int i = 0; i < n; ++i) xt[i] = x[i] + k2[i] * h / 2.0;
  m->dyn(params, xt, u, f_base, f_mass, k3, K3, U3);
  for (int i = 0; i < n; ++i) xt[i] = x[i] + k3[i] * h;
  m->dyn(params, xt, u, f_base, f_mass, k4, K4, U4);
  for (int i = 0; i < n; ++i) x_new[i] = x[i] + (h / 6.0) * (k1[i] + k2[i] * 2.0 + k3[i] * 2.0 + k4[i]);

  /* integration.hpp:36-39: k2_D_x = K2 (I + K1 h/2), k3_D_x = K3 (I + k2_D_x h/2), ... */
  double T[36], D2[36], D3[36], D4[36];
  for (int i = 0; i < nn; ++i) T[i] = K1[i] * h / 2.0 + ((i % (n + 1) == 0) ? 1.0 : 0.0);
  mat_mul(n, K2, T, D2);
  for (int i = 0; i < nn; ++i) T[i] = D2[i] * h / 2.0 + ((i % (n + 1) == 0) ? 1.0 : 0.0);
  mat_mul(n, K3, T, D3);
  for (int i = 0; i < nn; ++i) T[i] = D3[i] * h + ((i % (n + 1) == 0) ? 1.0 : 0.0);
  mat_mul(n, K4, T, D4);

  /* integration.hpp:41-43 */
  double d2[ORC_MAXNX], d3[ORC_MAXNX], d4[ORC_MAXNX], tv[ORC_MAXNX];
  mat_vec(n, K2, U1, tv);
  for (int i = 0; i < n; ++i) d2[i] = tv[i] * (h / 2.0) + U2[i];
  mat_vec(n, K3, d2, tv);
  for (int i = 0; i < n; ++i) d3[i] = tv[i] * (h / 2.0) + U3[i];
  mat_vec(n, K4, d3, tv);
  for (int i = 0; i < n; ++i) d4[i] = tv[i] * h + U4[i];

  /* integration.hpp:45-46 */
  for (int i = 0; i < nn; ++i)
    A[i] = ((i % (n + 1) == 0) ? 1.0 : 0.0) + (h / 6.0) * (K1[i] + D2[i] * 2.0 + D3[i] * 2.0 + D4[i]);
  for (int i = 0; i < n; ++i) B[i] = (h / 6.0) * (U1[i] + d2[i] * 2.0 + d3[i] * 2.0 + d4[i]);
}

void orc_rk4_no_jacobians(const double params[9], const double x[4], double u, double h,
                          const double f_base[2], const double f_mass[2], double x_new[4]) {
  rk4_nj(&kModels[0], params, x, u, h, f_base, f_mass, x_new);
}
void orc_rk4(const double params[9], const double x[4], double u, double h, const double f_base[2],
             const double f_mass[2], double x_new[4], double A[16], double B[4]) {
  rk4_j(&kModels[0], params, x, u, h, f_base, f_mass, x_new, A, B);
}
/* any model: A is nx x nx (nullable together with B -> Jacobian-free path) */
void orc_rk4_model(int model, const double* params, const double* x, double u, double h, double* x_new,
                   double* A, double* B) {
  static const double z2[2] = {0.0, 0.0};
  if (A && B)
    rk4_j(model_of(model), params, x, u, h, z2, z2, x_new, A, B);
  else
    rk4_nj(model_of(model), params, x, u, h, z2, z2, x_new);
}

/* optimization/integration.hpp:65-73: map to (-pi, pi] */
double orc_mod_pi(double angle) {
  const double pi = M_PI;
  const double two_pi = 2 * pi;
  angle = fmod(angle, two_pi);
  angle += (angle < 0) * two_pi;
  angle -= (angle > pi) * two_pi;
  return angle;
}

/* ------------------------------------------------------------------------------------------- */
/* L3: shooting constraint, problem assembly                                                    */
/* ------------------------------------------------------------------------------------------- */

static const double kZero2[2] = {0.0, 0.0};

/* optimization/optimization.cc:99-160.  The angles are wrapped only at the END of the interval here
 * (line 139), unlike FillInitialGuess / ComputePredictedStates which wrap after every step.
 * vars = [x_k(nx), x_k+1(nx), u(sp)];  J is nx x (2nx+sp) row-major = [Phi | -I | Gamma]. */
static void shoot(const orc_model* m, const double* params, int spacing, double dt, const double* vars,
                  double* err, double* J) {
  const int n = m->nx;
  const double* x_k = vars;
  const double* x_kp1 = vars + n;
  const double* u_k = vars + 2 * n;
  double x[ORC_MAXNX], xn[ORC_MAXNX];
  memcpy(x, x_k, sizeof(double) * (size_t)n);
  double* As = NULL;
  double* Bs = NULL;
  if (J) {
    As = (double*)malloc(sizeof(double) * (size_t)(n * n) * (size_t)spacing);
    Bs = (double*)malloc(sizeof(double) * (size_t)n * (size_t)spacing);
    for (int i = 0; i < spacing; ++i) {
      rk4_j(m, params, x, u_k[i], dt, kZero2, kZero2, xn, As + n * n * i, Bs + n * i);
      memcpy(x, xn, sizeof(double) * (size_t)n);
    }
  } else {
    for (int i = 0; i < spacing; ++i) {
      rk4_nj(m, params, x, u_k[i], dt, kZero2, kZero2, xn);
      memcpy(x, xn, sizeof(double) * (size_t)n);
    }
  }
  for (int t = 0; t < n; ++t)
    if (is_angle(m, t)) x[t] = orc_mod_pi(x[t]);

  if (J) {
    const int cols = 2 * n + spacing;
    double Phi[36];
    for (int i = 0; i < n * n; ++i) Phi[i] = (i % (n + 1) == 0) ? 1.0 : 0.0;
    /* optimization.cc:145-151: backward chain rule */
    for (int i = spacing - 1; i >= 0; --i) {
      double gv[ORC_MAXNX];
      mat_vec(n, Phi, Bs + n * i, gv);
      for (int r = 0; r < n; ++r) J[r * cols + 2 * n + i] = gv[r];
      mat_mul(n, Phi, As + n * n * i, Phi);
    }
    for (int r = 0; r < n; ++r)
      for (int cI = 0; cI < n; ++cI) {
        J[r * cols + cI] = Phi[r * n + cI];
        J[r * cols + n + cI] = (r == cI) ? -1.0 : 0.0;
      }
    free(As);
    free(Bs);
  }
  for (int i = 0; i < n; ++i) err[i] = x[i] - x_kp1[i];
  for (int t = 0; t < n; ++t)
    if (is_angle(m, t)) err[t] = orc_mod_pi(err[t]);
}

void orc_shooting_constraint(const double params[9], int spacing, double dt, const double* vars,
                             double err[4], double* J) {
  shoot(&kModels[0], params, spacing, dt, vars, err, J);
}
void orc_shooting_constraint_model(int model, const double* params, int spacing, double dt, const double* vars,
                                   double* err, double* J) {
  shoot(model_of(model), params, spacing, dt, vars, err, J);
}

static int num_states(const orc_opt_params* p) { /* optimization.hpp:52 */
  return (int)(p->window_length / p->state_spacing) + 1;
}

/* Terminal weights and targets in BuildProblem order (optimization.cc:236-267).  For the double pendulum
 * (no optimizer in the reference) th_final / th_dot_final apply to both poles and both are to be upright. */
static void terminal_spec(const orc_model* m, const orc_opt_params* p, double set_point, double* w, double* tgt) {
  for (int t = 0; t < m->nx; ++t) {
    if (t == 0) {
      w[t] = p->b_x_final_cost_weight;
      tgt[t] = set_point;
    } else if (t < m->nq) {
      w[t] = p->th_final_cost_weight;
      tgt[t] = M_PI / 2;
    } else if (t == m->nq) {
      w[t] = p->b_x_dot_final_cost_weight;
      tgt[t] = 0.0;
    } else {
      w[t] = p->th_dot_final_cost_weight;
      tgt[t] = 0.0;
    }
  }
}

static void problem_shape(const orc_model* m, const orc_opt_params* p, int* dim, int* n_eq, int* n_cost) {
  const int S = num_states(p);
  const int N = (int)p->window_length;
  const int n = m->nx;
  double w[ORC_MAXNX], tgt[ORC_MAXNX];
  terminal_spec(m, p, 0.0, w, tgt);
  int ne = n * (S - 1) + n, nc = 0;
  for (int t = 0; t < n; ++t) {
    if (w[t] >= 0.0)
      ++nc;
    else
      ++ne;
  }
  if (p->u_derivative_cost_weight > 0.0) nc += (N - 1) + 1;
  if (p->u_cost_weight > 0.0) nc += N;
  if (dim) *dim = n * S + N; /* optimization.cc:204-205 */
  if (n_eq) *n_eq = ne;
  if (n_cost) *n_cost = nc;
}

void orc_problem_shape(const orc_opt_params* p, int* dim, int* n_eq, int* n_cost) {
  problem_shape(&kModels[0], p, dim, n_eq, n_cost);
}
void orc_problem_shape_model(int model, const orc_opt_params* p, int* dim, int* n_eq, int* n_cost) {
  problem_shape(model_of(model), p, dim, n_eq, n_cost);
}

/* optimization/optimization.cc:194-301 (BuildProblem) evaluated at z.
 * Variable layout (MapKey<StateDim>, optimization.cc:27-37): state t of node s at nx*s+t, u_k at nx*S+k. */
static void problem_eval(const orc_model* m, const orc_opt_params* p, const double* dyn, const double* x_current,
                         double set_point, double u_prev, const double* z, double* r_cost, double* c_eq,
                         double* J_cost, double* A_eq) {
  const int S = num_states(p);
  const int N = (int)p->window_length;
  const int sp = (int)p->state_spacing;
  const int n = m->nx;
  int dim, n_eq, n_cost;
  problem_shape(m, p, &dim, &n_eq, &n_cost);
  if (J_cost) memset(J_cost, 0, sizeof(double) * (size_t)n_cost * (size_t)dim);
  if (A_eq) memset(A_eq, 0, sizeof(double) * (size_t)n_eq * (size_t)dim);

  int re = 0, rc = 0;
  const int nv = 2 * n + sp;
  double* vars = (double*)malloc(sizeof(double) * (size_t)nv);
  double* Jloc = A_eq ? (double*)malloc(sizeof(double) * (size_t)n * (size_t)nv) : NULL;

  /* shooting equality constraints between adjacent states (optimization.cc:208-225) */
  for (int s = 0; s + 1 < S; ++s) {
    for (int t = 0; t < n; ++t) {
      vars[t] = z[n * s + t];
      vars[n + t] = z[n * (s + 1) + t];
    }
    for (int k = 0; k < sp; ++k) vars[2 * n + k] = z[n * S + sp * s + k];
    double err[ORC_MAXNX];
    shoot(m, dyn, sp, p->control_dt, vars, err, Jloc);
    for (int r = 0; r < n; ++r) {
      c_eq[re + r] = err[r];
      if (A_eq) {
        double* row = A_eq + (size_t)(re + r) * dim;
        for (int t = 0; t < n; ++t) {
          row[n * s + t] = Jloc[r * nv + t];
          row[n * (s + 1) + t] = Jloc[r * nv + n + t];
        }
        for (int k = 0; k < sp; ++k) row[n * S + sp * s + k] = Jloc[r * nv + 2 * n + k];
      }
    }
    re += n;
  }
  free(vars);
  free(Jloc);

  /* equality constraint on the initial state, weight 1, angles wrapped (optimization.cc:228-232) */
  for (int t = 0; t < n; ++t) {
    double d = z[t] - x_current[t];
    if (is_angle(m, t)) d = orc_mod_pi(d);
    c_eq[re] = d * 1.0;
    if (A_eq) A_eq[(size_t)re * dim + t] = 1.0;
    ++re;
  }

  /* terminal rows: cost if weight >= 0 else equality with weight 1 (optimization.cc:236-267) */
  double w[ORC_MAXNX], tgt[ORC_MAXNX];
  terminal_spec(m, p, set_point, w, tgt);
  for (int t = 0; t < n; ++t) {
    const int idx = n * (S - 1) + t;
    double d = z[idx] - tgt[t];
    if (is_angle(m, t)) d = orc_mod_pi(d);
    if (w[t] >= 0.0) {
      r_cost[rc] = d * w[t];
      if (J_cost) J_cost[(size_t)rc * dim + idx] = w[t];
      ++rc;
    } else {
      c_eq[re] = d * 1.0;
      if (A_eq) A_eq[(size_t)re * dim + idx] = 1.0;
      ++re;
    }
  }

  /* penalty on the derivative of the control inputs (optimization.cc:270-294) */
  if (p->u_derivative_cost_weight > 0.0) {
    const double wd = p->u_derivative_cost_weight;
    for (int k = 0; k + 1 < N; ++k) {
      r_cost[rc] = (z[n * S + k] - z[n * S + k + 1]) * wd;
      if (J_cost) {
        J_cost[(size_t)rc * dim + n * S + k] = wd;
        J_cost[(size_t)rc * dim + n * S + k + 1] = -wd;
      }
      ++rc;
    }
    r_cost[rc] = (z[n * S + 0] - u_prev) * wd;
    if (J_cost) J_cost[(size_t)rc * dim + n * S + 0] = wd;
    ++rc;
  }
  /* penalty on the control inputs (optimization.cc:296-301) */
  if (p->u_cost_weight > 0.0) {
    const double wu = p->u_cost_weight;
    for (int k = 0; k < N; ++k) {
      r_cost[rc] = (z[n * S + k] - 0.0) * wu;
      if (J_cost) J_cost[(size_t)rc * dim + n * S + k] = wu;
      ++rc;
    }
  }
}

void orc_problem_eval(const orc_opt_params* p, const double dyn[9], const double x_current[4],
                      double set_point, double u_prev, const double* z, double* r_cost, double* c_eq,
                      double* J_cost, double* A_eq) {
  problem_eval(&kModels[0], p, dyn, x_current, set_point, u_prev, z, r_cost, c_eq, J_cost, A_eq);
}

void orc_problem_eval_model(int model, const orc_opt_params* p, const double* dyn, const double* x_current,
                            double set_point, double u_prev, const double* z, double* r_cost, double* c_eq,
                            double* J_cost, double* A_eq) {
  problem_eval(model_of(model), p, dyn, x_current, set_point, u_prev, z, r_cost, c_eq, J_cost, A_eq);
}

static double clampd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* optimization/optimization.cc:309-329 */
static void retract(const orc_model* m, const orc_opt_params* p, const orc_solver_opts* o, const double* z,
                    const double* dz, double alpha, double* z_out) {
  const int S = num_states(p);
  const int N = (int)p->window_length;
  const int n = m->nx;
  const int dim = n * S + N;
  for (int i = 0; i < dim; ++i) z_out[i] = z[i] + dz[i] * alpha;
  for (int s = 0; s < S; ++s) {
    for (int t = 1; t < m->nq; ++t) z_out[n * s + t] = orc_mod_pi(z_out[n * s + t]);
    z_out[n * s + 0] = clampd(z_out[n * s + 0], -o->b_x_limit, o->b_x_limit);
  }
  for (int k = 0; k < N; ++k) z_out[n * S + k] = clampd(z_out[n * S + k], -o->u_limit, o->u_limit);
}

void orc_retract(const orc_opt_params* p, const orc_solver_opts* o, const double* z, const double* dz,
                 double alpha, double* z_out) {
  retract(&kModels[0], p, o, z, dz, alpha, z_out);
}

void orc_retract_model(int model, const orc_opt_params* p, const orc_solver_opts* o, const double* z,
                       const double* dz, double alpha, double* z_out) {
  retract(model_of(model), p, o, z, dz, alpha, z_out);
}

/* ------------------------------------------------------------------------------------------- */
/* L2: the SQP specified by this repo (stands where mini_opt stands in the reference)          */
/* ------------------------------------------------------------------------------------------- */

/* Dense LU with partial pivoting; solves K x = b in place (b -> x).  Returns nonzero if singular. */
static int lu_solve(int n, ORC_WIDE* K, ORC_WIDE* b) {
#define ORC_WABS(v) ((v) < 0 ? -(v) : (v))
  ORC_WIDE kmax = 0;
  for (int i = 0; i < n * n; ++i) {
    const ORC_WIDE a = ORC_WABS(K[i]);
    if (a > kmax) kmax = a;
  }
  if (!(kmax > 0) || !isfinite(kmax)) return 1;
  const ORC_WIDE tiny = 0; /* only an exactly singular or non-finite pivot is a failure */
  for (int col = 0; col < n; ++col) {
    int piv = col;
    ORC_WIDE best = ORC_WABS(K[col * n + col]);
    for (int r = col + 1; r < n; ++r) {
      const ORC_WIDE a = ORC_WABS(K[r * n + col]);
      if (a > best) {
        best = a;
        piv = r;
      }
    }
    if (!(best > tiny)) return 1;
    if (piv != col) {
      for (int j = 0; j < n; ++j) {
        const ORC_WIDE t = K[col * n + j];
        K[col * n + j] = K[piv * n + j];
        K[piv * n + j] = t;
      }
      const ORC_WIDE t = b[col];
      b[col] = b[piv];
      b[piv] = t;
    }
    const ORC_WIDE inv = 1 / K[col * n + col];
    for (int r = col + 1; r < n; ++r) {
      const ORC_WIDE f = K[r * n + col] * inv;
      if (f == 0) continue;
      for (int j = col + 1; j < n; ++j) K[r * n + j] -= f * K[col * n + j];
      b[r] -= f * b[col];
    }
  }
  for (int r = n - 1; r >= 0; --r) {
    ORC_WIDE acc = b[r];
    for (int j = r + 1; j < n; ++j) acc -= K[r * n + j] * b[j];
    b[r] = acc / K[r * n + r];
  }
#undef ORC_WABS
  return 0;
}

/*
 * min_dz 1/2 |J dz + r|^2 + 1/2 lambda |dz_u|^2   s.t.  A dz + c = 0
 * on the full variable space, via the dense KKT system
 *     [ J^T J + lambda E_u   A^T ] [dz]   [ -J^T r ]
 *     [ A                    0   ] [nu] = [ -c     ]
 * (E_u selects the n_u control variables, which are last in z).  This is the role mini_opt's QP
 * plays at optimization.cc:81; the method (dense, full-space) is deliberately different from the
 * product's structure-exploiting per-lane solve so that the two check each other.
 */
int orc_qp_solve(int dim, int n_eq, int n_cost, int n_u, const double* J_cost, const double* r_cost,
                 const double* A_eq, const double* c_eq, double lambda, double* dz) {
  const int n = dim + n_eq;
  ORC_WIDE* K = (ORC_WIDE*)calloc((size_t)n * (size_t)n, sizeof(ORC_WIDE));
  ORC_WIDE* b = (ORC_WIDE*)calloc((size_t)n, sizeof(ORC_WIDE));
  for (int r = 0; r < n_cost; ++r) {
    const double* row = J_cost + (size_t)r * dim;
    for (int i = 0; i < dim; ++i) {
      const ORC_WIDE a = row[i];
      if (a == 0) continue;
      b[i] -= a * (ORC_WIDE)r_cost[r];
      for (int j = 0; j < dim; ++j) K[(size_t)i * n + j] += a * (ORC_WIDE)row[j]; /* (products of the arithmetic's values, summed in ORC_WIDE) */
    }
  }
  for (int k = dim - n_u; k < dim; ++k) K[(size_t)k * n + k] += lambda;
  for (int r = 0; r < n_eq; ++r) {
    for (int j = 0; j < dim; ++j) {
      const ORC_WIDE a = A_eq[(size_t)r * dim + j];
      K[(size_t)(dim + r) * n + j] = a;
      K[(size_t)j * n + dim + r] = a;
    }
    b[dim + r] = -c_eq[r];
  }
  const int bad = lu_solve(n, K, b);
  if (!bad) {
    for (int i = 0; i < dim; ++i) {
      dz[i] = (double)b[i];
      if (!isfinite(b[i])) {
        free(K);
        free(b);
        return 2;
      }
    }
  }
  free(K);
  free(b);
  return bad;
}

static double half_sq_norm(const double* v, int n) {
  double acc = 0.0;
  for (int i = 0; i < n; ++i) acc += v[i] * v[i];
  return 0.5 * acc;
}
static double l1_norm(const double* v, int n) {
  double acc = 0.0;
  for (int i = 0; i < n; ++i) acc += fabs(v[i]);
  return acc;
}

/*
 * The SQP iteration.  Stands where
 * mini_opt::ConstrainedNonlinearLeastSquares::Solve stands (optimization.cc:73-81):
 *   per iteration: linearise -> QP -> l1-merit penalty update -> Armijo backtracking with the
 *   retraction -> accept (decay lambda) or reject (raise lambda); exits on first-order tolerance,
 *   relative decrease, QP failure, lambda overflow, or max_iterations.
 */
static int solve(const orc_model* m, const orc_opt_params* p, const orc_solver_opts* o_in, const double* dyn,
                 const double* x_current, double set_point, double u_prev, const double* guess, double* z_out,
                 orc_solver_summary* summary) {
  orc_solver_opts o_def;
  if (!o_in) {
    orc_default_solver_opts(&o_def);
    o_in = &o_def;
  }
  const orc_solver_opts* o = o_in;
  int dim, n_eq, n_cost;
  problem_shape(m, p, &dim, &n_eq, &n_cost);
  const int N = (int)p->window_length;

  double* z = (double*)malloc(sizeof(double) * (size_t)dim);
  double* zt = (double*)malloc(sizeof(double) * (size_t)dim);
  double* dz = (double*)malloc(sizeof(double) * (size_t)dim);
  double* r = (double*)malloc(sizeof(double) * (size_t)(n_cost > 0 ? n_cost : 1));
  double* c = (double*)malloc(sizeof(double) * (size_t)n_eq);
  double* rt = (double*)malloc(sizeof(double) * (size_t)(n_cost > 0 ? n_cost : 1));
  double* ct = (double*)malloc(sizeof(double) * (size_t)n_eq);
  double* J = (double*)malloc(sizeof(double) * (size_t)(n_cost > 0 ? n_cost : 1) * (size_t)dim);
  double* A = (double*)malloc(sizeof(double) * (size_t)n_eq * (size_t)dim);
  memcpy(z, guess, sizeof(double) * (size_t)dim);

  double lambda = o->lambda_initial;
  double mu = p->equality_penalty_initial;
  double alpha_start = 1.0;
  int term = ORC_TERM_MAX_ITERATIONS;
  int iters = 0, ls_evals = 0, failed = 0;
  double f0 = 0.0, cn0 = 0.0, f_last = 0.0, cn_last = 0.0;

  for (int iter = 0; iter < (int)p->max_iterations; ++iter) {
    problem_eval(m, p, dyn, x_current, set_point, u_prev, z, r, c, J, A);
    const double f = half_sq_norm(r, n_cost);
    const double cn = l1_norm(c, n_eq);
    if (iter == 0) {
      f0 = f;
      cn0 = cn;
    }
    f_last = f;
    cn_last = cn;
    if (!isfinite(f) || !isfinite(cn)) {
      term = ORC_TERM_NON_FINITE;
      break;
    }
    ++iters;
    if (orc_qp_solve(dim, n_eq, n_cost, N, J, r, A, c, lambda, dz) != 0) {
      term = ORC_TERM_QP_INDEFINITE;
      break;
    }
    /* directional derivative of the cost and Gauss-Newton curvature along dz */
    double gd = 0.0, curv = 0.0;
    for (int i = 0; i < n_cost; ++i) {
      double jd = 0.0;
      const double* row = J + (size_t)i * dim;
      for (int j = 0; j < dim; ++j) jd += row[j] * dz[j];
      gd += r[i] * jd;
      curv += jd * jd;
    }
    for (int k = dim - N; k < dim; ++k) curv += lambda * dz[k] * dz[k];

    /* l1-merit penalty, Nocedal & Wright (18.36) with sigma = 1 */
    if (cn > 0.0) {
      const double mu_req = (gd + 0.5 * curv) / ((1.0 - o->penalty_rho) * cn);
      if (mu < mu_req) mu = mu_req;
    }
    const double D = gd - mu * cn;
    const double phi0 = f + mu * cn;
    /* first-order test: the step is still tried (and kept if it passes Armijo) before exiting.  Equality residuals at
     * the rounding floor of their own evaluation -- the shooting defects are differences of states after state_spacing
     * RK4 steps -- count as zero here: no iteration can remove them, and in single precision mu |c|_1 at that floor
     * (1e-6 .. 1e-4) would keep a converged controller iterating on noise for ever (round 4).  A SINGLE-PRECISION rule
     * (round 5): in double the floor would be 3e-14 -- it could move a decision only where |D| is within mu x 3e-14 of the
     * tolerance -- and the double kernels never carried the test, so the double builds of this file do not apply it either:
     * one exit rule in the parity dtype.  It is live in cpmpc_oracle_f32.c (ORC_EPS = FLT_EPSILON).  The merit and the
     * Armijo test use the residuals as they are. */
    /* the size of the states, taken at the terminal node (target and distance to it) times the number of intervals: the
     * nodes of a plan that is near its targets are all about that large, and a plan that is not has residuals far above
     * any rounding anyway */
    double x_l1 = 0.0;
    {
      double w_t[ORC_MAXNX], tgt_t[ORC_MAXNX];
      terminal_spec(m, p, set_point, w_t, tgt_t);
      const int S_ = num_states(p);
      for (int t = 0; t < m->nx; ++t) {
        double d = z[m->nx * (S_ - 1) + t] - tgt_t[t];
        if (is_angle(m, t)) d = orc_mod_pi(d);
        x_l1 += fabs(tgt_t[t]) + fabs(d);
      }
      x_l1 *= (double)(S_ - 1);
    }
    const double cn_floor = ((double)ORC_EPS > (double)1e-10) ? o->exit_defect_floor * (double)p->state_spacing * (double)ORC_EPS * x_l1 : (double)0.0;
    const double D_exit = gd - mu * (cn > cn_floor ? cn : 0.0);
    const int first_order = fabs(D_exit) < p->absolute_first_derivative_tol;

    /* Armijo backtracking along the retraction; the next trial step is the minimiser of the
     * quadratic through phi(0), phi'(0), phi(alpha), safeguarded to [ls_shrink_min, ls_shrink_max]
     * times the current step */
    /* Local convergence safeguard (round 3): once the QP step is tiny in every component the quadratic model is trusted
     * and the step is taken in full without the merit test.  Near the solution the achievable decrease of the l1 merit
     * (~1e-13 of an objective of ~1e3) is below the rounding of the merit itself and, with the penalty grown to ~1e4, a
     * full step raises mu |c|_1 by O(|dz|^2) more than it lowers the objective (Maratos): the Armijo test then rejects
     * the very steps that converge (measured against an independent solver: the iteration stalled 1e-7 .. 1e-4 from the
     * optimum; with this rule it reaches 1e-12). */
    double dz_inf = 0.0;
    for (int j = 0; j < dim; ++j) {
      const double aj = fabs(dz[j]);
      if (aj > dz_inf || aj != aj) dz_inf = aj; /* a NaN component makes dz_inf NaN for good: not tiny */
    }
    /* Only for the UNDAMPED step (lambda = 0): a step that is small because the damping is large says nothing about the
     * distance to the optimum, and taking it without the merit test would also lower lambda again, so that MAX_LAMBDA
     * could no longer be reached through the controls. */
    const int tiny = dz_inf <= o->full_step_below && lambda == 0.0; /* false for NaN and for full_step_below = 0 with any nonzero step */
    double alpha = tiny ? 1.0 : alpha_start;
    int accepted = 0, backtracked = 0;
    double phi_t = 0.0, f_t = 0.0, cn_t = 0.0;
    if (first_order && tiny) {
      /* Converged (round 4): the first-order test holds and the undamped QP step is tiny in every component.  The full
       * step is taken and the iteration ends, WITHOUT evaluating the merit at the new point: nothing depends on that
       * value any more, and it is a whole rollout -- a quarter of the work of a settled controller's tick.  The reported
       * cost and residual are those of the iterate the step was computed from.  The committed iterate is finite without a
       * check of its own: z is (f and |c|_1 were tested above and every variable enters a residual), |dz|_inf <= full_step_below
       * is (`tiny` is false for a NaN or infinite component), and the retraction only wraps and clamps; a problem whose
       * data are not finite never gets here -- it left with NON_FINITE at the linearisation
       * (tests/test_gpu_round5.py::test_poisoned_settled_controllers...). */
      retract(m, p, o, z, dz, 1.0, zt);
      memcpy(z, zt, sizeof(double) * (size_t)dim);
      lambda *= o->lambda_scale_down;
      if (lambda < o->lambda_min) lambda = 0.0;
      term = ORC_TERM_SATISFIED_FIRST_ORDER_TOL;
      break;
    }
    for (int t = 0; t < o->max_line_search_iterations; ++t) {
      backtracked = (t > 0);
      retract(m, p, o, z, dz, alpha, zt);
      problem_eval(m, p, dyn, x_current, set_point, u_prev, zt, rt, ct, NULL, NULL);
      ++ls_evals;
      f_t = half_sq_norm(rt, n_cost);
      cn_t = l1_norm(ct, n_eq);
      phi_t = f_t + mu * cn_t;
      if (phi_t <= phi0 + o->armijo_c1 * alpha * D || (tiny && isfinite(phi_t))) { /* false for NaN */
        accepted = 1;
        break;
      }
      {
        /* a trial whose merit is not finite (the rollout diverged: inf - inf, or an overflow) is a step that is far
         * too long, exactly like one whose merit is finite and astronomically large: cut to the lower safeguard.
         * (Which of NaN / inf / 1e90 a diverged rollout produces is an accident of the arithmetic; the step
         * length must not depend on it.) */
        const double denom = 2.0 * (phi_t - phi0 - D * alpha);
        double a_new = !isfinite(phi_t) ? (o->ls_shrink_min * alpha)
                                        : ((denom > 0.0) ? (-D * alpha * alpha / denom) : (o->ls_shrink_max * alpha));
        if (!(a_new >= o->ls_shrink_min * alpha)) a_new = o->ls_shrink_min * alpha;
        if (a_new > o->ls_shrink_max * alpha) a_new = o->ls_shrink_max * alpha;
        alpha = a_new;
      }
    }
    /* step-length memory: after an accepted step the next line search starts from ls_alpha_growth
     * times that step, capped at the full step; after a failed search (the damped direction will be
     * a different one) it starts from the full step again; 0 disables (always start from 1) */
    alpha_start = 1.0;
    if (accepted && o->ls_alpha_growth > 0.0) {
      alpha_start = (backtracked ? o->ls_alpha_growth_backtracked : o->ls_alpha_growth) * alpha;
      if (!(alpha_start < 1.0)) alpha_start = 1.0;
    }
    if (accepted) {
      memcpy(z, zt, sizeof(double) * (size_t)dim);
      f_last = f_t;
      cn_last = cn_t;
      lambda *= o->lambda_scale_down;
      if (lambda < o->lambda_min) lambda = 0.0;
    }
    if (first_order) {
      term = ORC_TERM_SATISFIED_FIRST_ORDER_TOL;
      break;
    }
    if (accepted) {
      /* (a tiny step taken without the merit test may raise the merit by rounding: with relative_exit_tol = 0 that is
       * not an exit) */
      if (p->relative_exit_tol > 0.0 && (phi0 - phi_t) < p->relative_exit_tol * phi0) {
        term = ORC_TERM_SATISFIED_RELATIVE_TOL;
        break;
      }
    } else {
      ++failed;
      lambda = (lambda > 0.0) ? lambda * o->lambda_scale_up : o->lambda_failure_init;
      if (lambda > o->lambda_max) {
        term = ORC_TERM_MAX_LAMBDA;
        break;
      }
    }
  }

  if (z_out) memcpy(z_out, z, sizeof(double) * (size_t)dim);
  if (summary) {
    summary->termination_state = term;
    summary->iterations = iters;
    summary->line_search_evals = ls_evals;
    summary->failed_steps = failed;
    summary->initial_cost = f0;
    summary->initial_eq_l1 = cn0;
    summary->final_cost = f_last;
    summary->final_eq_l1 = cn_last;
    summary->final_penalty = mu;
    summary->final_lambda = lambda;
  }
  free(z);
  free(zt);
  free(dz);
  free(r);
  free(c);
  free(rt);
  free(ct);
  free(J);
  free(A);
  return 0;
}

int orc_solve(const orc_opt_params* p, const orc_solver_opts* o_in, const double dyn[9],
              const double x_current[4], double set_point, double u_prev, const double* guess,
              double* z_out, orc_solver_summary* summary) {
  return solve(&kModels[0], p, o_in, dyn, x_current, set_point, u_prev, guess, z_out, summary);
}

/* ------------------------------------------------------------------------------------------- */
/* L3: Optimization::Step                                                                       */
/* ------------------------------------------------------------------------------------------- */

struct orc_optimization {
  orc_opt_params params;
  orc_solver_opts opts;
  const orc_model* model;
  int has_prev;
  int dim;
  double* prev; /* previous_solution_ (optimization.hpp:107) */
};

/* optimization/optimization.cc:13-22: the constructor's preconditions */
orc_optimization* orc_opt_create_model(const orc_opt_params* p, const orc_solver_opts* o, int model) {
  if (!(p->control_dt > 0)) return NULL;
  if (!(p->window_length >= 1)) return NULL;
  if (p->state_spacing == 0 || p->window_length % p->state_spacing != 0) return NULL;
  if (!(p->max_iterations >= 1)) return NULL;
  if (!(p->u_cost_weight >= 0.0)) return NULL;
  if (!(p->u_derivative_cost_weight >= 0.0)) return NULL;
  orc_optimization* opt = (orc_optimization*)calloc(1, sizeof(orc_optimization));
  opt->params = *p;
  if (o)
    opt->opts = *o;
  else
    orc_default_solver_opts(&opt->opts);
  opt->model = model_of(model);
  problem_shape(opt->model, p, &opt->dim, NULL, NULL);
  opt->prev = (double*)calloc((size_t)opt->dim, sizeof(double));
  opt->has_prev = 0;
  return opt;
}
orc_optimization* orc_opt_create(const orc_opt_params* p, const orc_solver_opts* o) {
  return orc_opt_create_model(p, o, ORC_MODEL_SINGLE);
}

void orc_opt_destroy(orc_optimization* opt) {
  if (!opt) return;
  free(opt->prev);
  free(opt);
}

void orc_opt_reset(orc_optimization* opt) { opt->has_prev = 0; } /* optimization.hpp:83 */

/* optimization.hpp:86-89 */
void orc_opt_set_previous_solution(orc_optimization* opt, const double* z, int n) {
  if (n != opt->dim) {
    opt->has_prev = 0;
    return;
  }
  memcpy(opt->prev, z, sizeof(double) * (size_t)n);
  opt->has_prev = 1;
}

int orc_opt_has_previous_solution(const orc_optimization* opt) { return opt->has_prev; }
int orc_opt_dim(const orc_optimization* opt) { return opt->dim; }

/* optimization/optimization.cc:39-97.  state[nx], dyn[np]; predicted_out[N*nx]. */
int orc_opt_step(orc_optimization* opt, const double* state, const double* dyn, double set_point,
                 double* u_out, double* predicted_out, double* guess_out, double* z_out,
                 orc_solver_summary* summary) {
  const orc_opt_params* p = &opt->params;
  const orc_model* m = opt->model;
  const int n = m->nx;
  const int S = num_states(p);
  const int N = (int)p->window_length;
  const int sp = (int)p->state_spacing;
  const int dim = opt->dim;

  /* BuildProblem runs first and reads u_prev from the not-yet-overwritten previous solution
   * (optimization.cc:44,288-291) */
  const double u_prev = opt->has_prev ? opt->prev[n * S + 0] : 0.0;

  double* guess = (double*)calloc((size_t)dim, sizeof(double));
  if (opt->has_prev) {
    /* optimization.cc:50-57: copy, overwrite x0, shift controls left by one */
    memcpy(guess, opt->prev, sizeof(double) * (size_t)dim);
    for (int t = 0; t < n; ++t) guess[t] = state[t];
    for (int k = 0; k + 1 < N; ++k) guess[n * S + k] = guess[n * S + k + 1];
  } else {
    /* optimization.cc:58-68: sinusoid control guess */
    for (int t = 0; t < n; ++t) guess[t] = state[t];
    for (int k = 0; k < N; ++k)
      guess[n * S + k] = p->u_guess_sinusoid_amplitude * sin((double)k / (double)N * 2 * M_PI);
  }
  /* FillInitialGuess, optimization.cc:333-351: roll the states, wrapping after EVERY step */
  {
    double x[ORC_MAXNX], xn[ORC_MAXNX];
    memcpy(x, guess, sizeof(double) * (size_t)n);
    for (int s = 1; s < S; ++s) {
      for (int k = 0; k < sp; ++k) {
        rk4_nj(m, dyn, x, guess[n * S + (s - 1) * sp + k], p->control_dt, kZero2, kZero2, xn);
        memcpy(x, xn, sizeof(double) * (size_t)n);
        for (int t = 1; t < m->nq; ++t) x[t] = orc_mod_pi(x[t]);
      }
      for (int t = 0; t < n; ++t) guess[n * s + t] = x[t];
    }
  }
  if (guess_out) memcpy(guess_out, guess, sizeof(double) * (size_t)dim);

  double* z = (double*)malloc(sizeof(double) * (size_t)dim);
  solve(m, p, &opt->opts, dyn, state, set_point, u_prev, guess, z, summary);

  /* optimization.cc:85 */
  memcpy(opt->prev, z, sizeof(double) * (size_t)dim);
  opt->has_prev = 1;

  if (u_out)
    for (int k = 0; k < N; ++k) u_out[k] = z[n * S + k]; /* optimization.cc:88-89 */

  /* ComputePredictedStates, optimization.cc:353-371 */
  if (predicted_out) {
    double x[ORC_MAXNX], xn[ORC_MAXNX];
    memcpy(x, state, sizeof(double) * (size_t)n);
    for (int k = 0; k < N; ++k) {
      rk4_nj(m, dyn, x, z[n * S + k], p->control_dt, kZero2, kZero2, xn);
      memcpy(x, xn, sizeof(double) * (size_t)n);
      for (int t = 1; t < m->nq; ++t) x[t] = orc_mod_pi(x[t]);
      for (int t = 0; t < n; ++t) predicted_out[n * k + t] = x[t];
    }
  }
  if (z_out) memcpy(z_out, z, sizeof(double) * (size_t)dim);
  free(z);
  free(guess);
  return 0;
}

/* ------------------------------------------------------------------------------------------- */
/* Simulator                                                                                    */
/* ------------------------------------------------------------------------------------------- */

/* optimization/simulator.cc:11-36: fixed 1 ms sub-steps, angles wrapped after each */
static void sim_step(const orc_model* m, const double* params, double dt, double u, const double* f_base,
                     const double* f_mass, double* state) {
  const double internal_dt = 0.001;
  while (dt > 0.0) {
    const double h = dt < internal_dt ? dt : internal_dt;
    double xn[ORC_MAXNX];
    rk4_nj(m, params, state, u, h, f_base, f_mass, xn);
    memcpy(state, xn, sizeof(double) * (size_t)m->nx);
    for (int t = 1; t < m->nq; ++t) state[t] = orc_mod_pi(state[t]);
    dt -= internal_dt;
  }
}
void orc_sim_step(const double params[9], double dt, double u, const double f_base[2], const double f_mass[2],
                  double state[4]) {
  sim_step(&kModels[0], params, dt, u, f_base, f_mass, state);
}
void orc_sim_step_model(int model, const double* params, double dt, double u, double* state) {
  sim_step(model_of(model), params, dt, u, kZero2, kZero2, state);
}

/* ------------------------------------------------------------------------------------------- */
/* batch driver                                                                                 */
/* ------------------------------------------------------------------------------------------- */

int orc_step_batch_cold_model(int model, const orc_opt_params* p, const orc_solver_opts* o, const double* dyn,
                              double set_point, int64_t B, const double* x0_soa, double* u_out_soa,
                              double* pred_out_soa, int32_t* status, int32_t* iters, int num_threads) {
  const int N = (int)p->window_length;
  const int n = model_of(model)->nx;
  int used = 1;
#ifdef _OPENMP
  if (num_threads > 0) omp_set_num_threads(num_threads);
  used = omp_get_max_threads();
#else
  (void)num_threads;
#endif
#pragma omp parallel
  {
    orc_optimization* opt = orc_opt_create_model(p, o, model);
    double* u = (double*)malloc(sizeof(double) * (size_t)N);
    double* pred = (double*)malloc(sizeof(double) * (size_t)n * (size_t)N);
#pragma omp for schedule(dynamic, 16)
    for (int64_t b = 0; b < B; ++b) {
      if (!opt) continue;
      orc_opt_reset(opt);
      double x0[ORC_MAXNX];
      for (int t = 0; t < n; ++t) x0[t] = x0_soa[(int64_t)t * B + b];
      orc_solver_summary sum;
      orc_opt_step(opt, x0, dyn, set_point, u, pred_out_soa ? pred : NULL, NULL, NULL, &sum);
      for (int k = 0; k < N; ++k) u_out_soa[(int64_t)k * B + b] = u[k];
      if (pred_out_soa)
        for (int k = 0; k < N; ++k)
          for (int t = 0; t < n; ++t) pred_out_soa[((int64_t)k * n + t) * B + b] = pred[n * k + t];
      if (status) status[b] = sum.termination_state;
      if (iters) iters[b] = sum.iterations;
    }
    free(u);
    free(pred);
    orc_opt_destroy(opt);
  }
  return used;
}

int orc_step_batch_cold(const orc_opt_params* p, const orc_solver_opts* o, const double dyn[9],
                        double set_point, int64_t B, const double* x0_soa, double* u_out_soa,
                        double* pred_out_soa, int32_t* status, int32_t* iters, int num_threads) {
  return orc_step_batch_cold_model(ORC_MODEL_SINGLE, p, o, dyn, set_point, B, x0_soa, u_out_soa, pred_out_soa,
                                   status, iters, num_threads);
}
