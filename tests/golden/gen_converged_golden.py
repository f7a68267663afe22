"""Converged control sequences from an INDEPENDENT statement of the NLP and an INDEPENDENT solver (scipy).

Why: mini_opt (the reference's solver) is an absent submodule, so iterate-level parity with it is unreachable
 and `step_golden.json` is this repo's own oracle frozen.  But a *converged* solution does not depend on
the solver: any correct method that reaches the same basin reaches the same KKT point of

    min 1/2 |r(z)|^2   s.t.  c(z) = 0                (optimization/optimization.cc:194-301)

This script states that NLP a second time, sharing NO code with oracle/ or the HIP kernels:
  * dynamics f, df/dx, df/du: the SymPy Lagrangian derivation of gen_dynamics_golden.py (symbolic/dynamics_single.py:58-143
    re-derived), lambdified to numpy;
  * RK4 with stage sensitivities (optimization/integration.hpp:13-49), the shooting defects with their chain rule
    (optimization.cc:99-160), the initial / terminal rows and the control costs (optimization.cc:228-301) in numpy;
and solves it with scipy.optimize SLSQP from the reference's own initial guess (sinusoid controls, states rolled out,
optimization.cc:58-71,333-351), then polishes with Newton steps on the KKT system (Hessian of the Lagrangian by
central differences of the analytic first derivatives) until the KKT residual is at rounding level.

Each case stores x0, the configuration, u*, z*, the objective, |c|_1, the KKT residual and which clamps (if any) are
active.  For cases where this repo's oracle, run to a fixed point from the same guess, ends in a different basin, a
second solution is stored: the independent Newton-KKT iteration started at the oracle's fixed point perturbed by 1e-3
(`near_oracle`), which shows whether that fixed point is a KKT point of the independently stated NLP.

The oracle is imported only for that labelling step and never enters u*.

Run from the repository root (8 processes, a few minutes):  python tests/golden/gen_converged_golden.py
"""
import json
import multiprocessing as mpz
import os
import sys

import numpy as np
import scipy.optimize as so
import sympy as sp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

PI = float(np.pi)
NX = 4

DYN_UI = [1.0, 0.1, 0.25, 9.81, 0.05, 0.1, 0.02, 0.8, 100.0]    # viz/src/application.ts:61-71
DYN_TEST = [1.0, 0.1, 0.25, 9.81, 0.03, 0.1, 0.13, 0.8, 100.0]  # optimization/optimization_test.cc:20

# OptimizationParams defaults (optimization/optimization.hpp:12-48)
DEFAULTS = dict(control_dt=0.01, window_length=40, state_spacing=10, u_guess_sinusoid_amplitude=10.0,
                u_cost_weight=0.1, u_derivative_cost_weight=0.1, b_x_final_cost_weight=150.0,
                th_final_cost_weight=-1.0, b_x_dot_final_cost_weight=-1.0, th_dot_final_cost_weight=-1.0)

CONFIGS = {
    "reference defaults": (dict(), DYN_UI),
    "optimization_test.cc:13-19 (state_spacing 5)": (dict(state_spacing=5), DYN_TEST),
}


# ----------------------------------------------------------------------------------------------------------------
# dynamics (independent SymPy derivation -> numpy)
# ----------------------------------------------------------------------------------------------------------------
_DYN = None


def dyn_fn():
    global _DYN
    if _DYN is None:
        from gen_dynamics_golden import derive_symbolic
        args, f, Jx, Ju = derive_symbolic()
        exprs = list(f) + [Jx[i, j] for i in range(4) for j in range(4)] + list(Ju)
        _DYN = sp.lambdify(args, exprs, modules=["numpy", {"Heaviside": lambda x, *a: np.where(x > 0, 1.0, 0.0)}], cse=True)
    return _DYN


def dynamics(prm, x, u):
    """x [L,4], u [L] -> f [L,4], Jx [L,4,4], Ju [L,4]   (no external forces inside the optimizer, optimization.cc:124-126)"""
    L = x.shape[0]
    with np.errstate(all="ignore"):
        out = dyn_fn()(*prm, x[:, 0], x[:, 1], x[:, 2], x[:, 3], u, 0.0, 0.0, 0.0, 0.0)
    out = [np.broadcast_to(np.asarray(o, dtype=float), (L,)) for o in out]
    f = np.stack(out[0:4], axis=1)
    Jx = np.stack(out[4:20], axis=1).reshape(L, 4, 4)
    Ju = np.stack(out[20:24], axis=1)
    return f, Jx, Ju


def rk4_jac(prm, x, u, h):
    """integration.hpp:13-49 for L states at once."""
    L = x.shape[0]
    eye = np.broadcast_to(np.eye(4), (L, 4, 4))
    k1, J1, B1 = dynamics(prm, x, u)
    K1x, K1u = J1, B1
    k2, J2, B2 = dynamics(prm, x + 0.5 * h * k1, u)
    K2x = J2 @ (eye + 0.5 * h * K1x)
    K2u = np.einsum("lij,lj->li", J2, 0.5 * h * K1u) + B2
    k3, J3, B3 = dynamics(prm, x + 0.5 * h * k2, u)
    K3x = J3 @ (eye + 0.5 * h * K2x)
    K3u = np.einsum("lij,lj->li", J3, 0.5 * h * K2u) + B3
    k4, J4, B4 = dynamics(prm, x + h * k3, u)
    K4x = J4 @ (eye + h * K3x)
    K4u = np.einsum("lij,lj->li", J4, h * K3u) + B4
    xn = x + h / 6.0 * (k1 + 2 * k2 + 2 * k3 + k4)
    A = eye + h / 6.0 * (K1x + 2 * K2x + 2 * K3x + K4x)
    B = h / 6.0 * (K1u + 2 * K2u + 2 * K3u + K4u)
    return xn, A, B


def rk4(prm, x, u, h):
    """integration.hpp:52-62 for one state (used for the guess roll-out)."""
    x = np.asarray(x, dtype=float)[None, :]
    u = np.asarray([u], dtype=float)
    k1 = dynamics(prm, x, u)[0]
    k2 = dynamics(prm, x + 0.5 * h * k1, u)[0]
    k3 = dynamics(prm, x + 0.5 * h * k2, u)[0]
    k4 = dynamics(prm, x + h * k3, u)[0]
    return (x + h / 6.0 * (k1 + 2 * k2 + 2 * k3 + k4))[0]


def mod_pi(a):
    """integration.hpp:65-73: (-pi, pi], mod_pi(+-pi) = +pi."""
    a = np.fmod(a, 2 * PI)
    a = np.where(a < 0, a + 2 * PI, a)
    return np.where(a > PI, a - 2 * PI, a)


# ----------------------------------------------------------------------------------------------------------------
# the NLP
# ----------------------------------------------------------------------------------------------------------------
class Problem:
    def __init__(self, cfg, prm, x_hat, set_point=0.0, u_prev=0.0):
        self.cfg = dict(DEFAULTS)
        self.cfg.update(cfg)
        c = self.cfg
        self.prm = list(prm)
        self.N = int(c["window_length"])
        self.sp = int(c["state_spacing"])
        self.S = self.N // self.sp + 1
        self.dim = NX * self.S + self.N
        self.h = float(c["control_dt"])
        self.x_hat = np.asarray(x_hat, dtype=float)
        self.u_prev = float(u_prev)
        self.wu = float(c["u_cost_weight"])
        self.wd = float(c["u_derivative_cost_weight"])
        self.term_w = [float(c["b_x_final_cost_weight"]), float(c["th_final_cost_weight"]),
                       float(c["b_x_dot_final_cost_weight"]), float(c["th_dot_final_cost_weight"])]
        self.term_tgt = [float(set_point), PI / 2, 0.0, 0.0]   # optimization.cc:236-267

    def split(self, z):
        xs = z[:NX * self.S].reshape(self.S, NX)
        us = z[NX * self.S:]
        return xs, us

    def guess(self):
        """optimization.cc:58-71,333-351: sinusoid controls, states rolled out with theta wrapped every step."""
        N, amp = self.N, float(self.cfg["u_guess_sinusoid_amplitude"])
        us = np.array([amp * np.sin(k / N * 2 * PI) for k in range(N)])
        xs = np.zeros((self.S, NX))
        x = self.x_hat.copy()
        xs[0] = x
        for k in range(N):
            x = rk4(self.prm, x, us[k], self.h)
            x[1] = float(mod_pi(x[1]))
            if (k + 1) % self.sp == 0:
                xs[(k + 1) // self.sp] = x
        return np.concatenate([xs.ravel(), us])

    def constraints(self, z, jac=True):
        """c(z) and dc/dz: shooting defects (optimization.cc:99-160), initial-state rows (:228-232), terminal
        equality rows (:236-267, negative weight)."""
        S, N, sp, L = self.S, self.N, self.sp, self.S - 1
        xs, us = self.split(z)
        x = xs[:L].copy()
        Phi = np.broadcast_to(np.eye(4), (L, 4, 4)).copy()
        Gam = np.zeros((L, 4, sp))
        for i in range(sp):
            u_i = us[np.arange(L) * sp + i]
            x, A, B = rk4_jac(self.prm, x, u_i, self.h)
            Phi = A @ Phi
            Gam = A @ Gam
            Gam[:, :, i] = B
        x[:, 1] = mod_pi(x[:, 1])
        err = x - xs[1:]
        err[:, 1] = mod_pi(err[:, 1])
        rows, Jrows = [], []
        for s in range(L):
            rows.append(err[s])
            if jac:
                J = np.zeros((4, self.dim))
                J[:, 4 * s:4 * s + 4] = Phi[s]
                J[:, 4 * (s + 1):4 * (s + 1) + 4] = -np.eye(4)
                J[:, 4 * S + s * sp:4 * S + (s + 1) * sp] = Gam[s]
                Jrows.append(J)
        d0 = xs[0] - self.x_hat
        d0[1] = float(mod_pi(d0[1]))
        rows.append(d0)
        if jac:
            J = np.zeros((4, self.dim))
            J[:, 0:4] = np.eye(4)
            Jrows.append(J)
        for t in range(4):
            if self.term_w[t] < 0:
                d = xs[S - 1, t] - self.term_tgt[t]
                if t == 1:
                    d = float(mod_pi(d))
                rows.append(np.array([d]))
                if jac:
                    J = np.zeros((1, self.dim))
                    J[0, 4 * (S - 1) + t] = 1.0
                    Jrows.append(J)
        c = np.concatenate(rows)
        return (c, np.vstack(Jrows)) if jac else c

    def residuals(self, z):
        """r(z) and dr/dz: terminal cost rows (weight >= 0), control rate and control costs (optimization.cc:270-301)."""
        S, N = self.S, self.N
        xs, us = self.split(z)
        r, J = [], []

        def row(val, idx_coef):
            r.append(val)
            j = np.zeros(self.dim)
            for i, cf in idx_coef:
                j[i] += cf
            J.append(j)
        for t in range(4):
            w = self.term_w[t]
            if w >= 0:
                d = xs[S - 1, t] - self.term_tgt[t]
                if t == 1:
                    d = float(mod_pi(d))
                row(w * d, [(4 * (S - 1) + t, w)])
        if self.wd > 0:
            for k in range(N - 1):
                row(self.wd * (us[k] - us[k + 1]), [(4 * S + k, self.wd), (4 * S + k + 1, -self.wd)])
            row(self.wd * (us[0] - self.u_prev), [(4 * S, self.wd)])
        if self.wu > 0:
            for k in range(N):
                row(self.wu * us[k], [(4 * S + k, self.wu)])
        return np.array(r), np.array(J)

    def objective(self, z):
        r, J = self.residuals(z)
        return 0.5 * float(r @ r), J.T @ r

    def kkt(self, z, lam=None):
        """(gradient of the Lagrangian, c, multipliers); multipliers by least squares when not given."""
        _, g = self.objective(z)
        c, A = self.constraints(z)
        if lam is None:
            lam = np.linalg.lstsq(A.T, -g, rcond=None)[0]
        return g + A.T @ lam, c, lam

    def newton_polish(self, z, iters=12, tol=1e-11):
        """Newton on F(z, lam) = [grad f + A^T lam; c] with H = J^T J + d/dz (A(z)^T lam) by central differences."""
        z = z.copy()
        gl, c, lam = self.kkt(z)
        hist = []
        for _ in range(iters):
            res = max(np.abs(gl).max(), np.abs(c).max())
            hist.append(float(res))
            if res < tol:
                break
            r, J = self.residuals(z)
            _, A = self.constraints(z)
            H = J.T @ J
            eps = 1e-6
            Hc = np.zeros((self.dim, self.dim))
            for i in range(self.dim):
                if i >= 4 * self.S or True:
                    zp, zm = z.copy(), z.copy()
                    zp[i] += eps
                    zm[i] -= eps
                    Ap = self.constraints(zp)[1]
                    Am = self.constraints(zm)[1]
                    Hc[:, i] = ((Ap - Am).T @ lam) / (2 * eps)
            H = H + 0.5 * (Hc + Hc.T)
            n_eq = A.shape[0]
            K = np.block([[H, A.T], [A, np.zeros((n_eq, n_eq))]])
            step = np.linalg.solve(K, -np.concatenate([gl, c]))
            t = 1.0
            for _ls in range(8):
                zn = z + t * step[:self.dim]
                ln = lam + t * step[self.dim:]
                gln, cn, _ = self.kkt(zn, ln)
                if max(np.abs(gln).max(), np.abs(cn).max()) < res or t < 1e-2:
                    break
                t *= 0.5
            z, lam, gl, c = zn, ln, gln, cn
        hist.append(float(max(np.abs(gl).max(), np.abs(c).max())))
        return z, lam, hist

    def solve(self, z0):
        cons = {"type": "eq", "fun": lambda z: self.constraints(z, jac=False), "jac": lambda z: self.constraints(z)[1]}
        lo = np.full(self.dim, -np.inf)
        hi = np.full(self.dim, np.inf)
        lo[0:4 * self.S:4], hi[0:4 * self.S:4] = -5.0, 5.0      # optimization.cc:320 (retraction clamp on b_x)
        lo[4 * self.S:], hi[4 * self.S:] = -300.0, 300.0        # optimization.cc:327 (on u)
        res = so.minimize(lambda z: self.objective(z)[0], z0, jac=lambda z: self.objective(z)[1], method="SLSQP",
                          constraints=[cons], bounds=list(zip(lo, hi)), options={"ftol": 1e-14, "maxiter": 600})
        z = res.x
        xs, us = self.split(z)
        clamped = bool((np.abs(xs[:, 0]) > 5.0 - 1e-9).any() or (np.abs(us) > 300.0 - 1e-9).any())
        z, lam, hist = self.newton_polish(z)
        return z, lam, hist, clamped, int(res.nit), str(res.message)


# ----------------------------------------------------------------------------------------------------------------
# cases
# ----------------------------------------------------------------------------------------------------------------
def make_cases():
    rng = np.random.default_rng(20261004)
    cases = []
    for tag, n_up, n_sw in (("reference defaults", 48, 48), ("optimization_test.cc:13-19 (state_spacing 5)", 16, 16)):
        for i in range(n_up):
            cases.append((tag, "near-upright", [rng.uniform(-0.3, 0.3), PI / 2 + rng.uniform(-0.4, 0.4),
                                                rng.uniform(-0.5, 0.5), rng.uniform(-1, 1)]))
        for i in range(n_sw):
            cases.append((tag, "swing-up", [rng.uniform(-0.6, 0.6), rng.uniform(-PI, PI), rng.uniform(-1, 1),
                                            rng.uniform(-3, 3)]))
    return cases


def oracle_fixed_point(cfg, prm, x0, iters=200):
    from oracle import oracle as orc
    over = {k: v for k, v in cfg.items()}
    p = orc.default_opt_params(max_iterations=iters, relative_exit_tol=0.0, absolute_first_derivative_tol=0.0, **over)
    opt = orc.Optimization(p)
    out = opt.step(np.asarray(x0, dtype=float), prm, 0.0)
    return out.z, int(out.solver_outputs.termination_state) if hasattr(out.solver_outputs, "termination_state") else -1


def run_case(case):
    tag, kind, x0 = case
    cfg, prm = CONFIGS[tag]
    P = Problem(cfg, prm, x0)
    z0 = P.guess()
    z, lam, hist, clamped, nit, msg = P.solve(z0)
    f = P.objective(z)[0]
    c = P.constraints(z, jac=False)
    rec = {"config": tag, "kind": kind, "x0": [float(v) for v in x0], "dyn": list(prm), "params": cfg, "set_point": 0.0,
           "u_star": [float(v) for v in P.split(z)[1]], "z_star": [float(v) for v in z], "objective": f,
           "eq_l1": float(np.abs(c).sum()), "kkt_residual": hist[-1], "kkt_history": hist, "clamp_active": clamped,
           "slsqp_iterations": nit, "slsqp_message": msg}
    try:
        z_orc, _ = oracle_fixed_point(cfg, prm, x0)
        gl, c_o, _ = P.kkt(z_orc)
        rec["oracle_fixed_point"] = {"max_abs_du_vs_u_star": float(np.abs(P.split(z_orc)[1] - P.split(z)[1]).max()),
                                     "objective": P.objective(z_orc)[0],
                                     "independent_kkt_residual": float(max(np.abs(gl).max(), np.abs(c_o).max()))}
        if rec["oracle_fixed_point"]["max_abs_du_vs_u_star"] > 1e-6:
            rng = np.random.default_rng(abs(hash(tuple(x0))) % (2 ** 32))
            z1, lam1, hist1 = P.newton_polish(z_orc + 1e-3 * rng.standard_normal(z_orc.size), iters=20)
            rec["near_oracle"] = {"u_star": [float(v) for v in P.split(z1)[1]], "objective": P.objective(z1)[0],
                                  "eq_l1": float(np.abs(P.constraints(z1, jac=False)).sum()), "kkt_residual": hist1[-1],
                                  "max_abs_du_vs_oracle": float(np.abs(P.split(z1)[1] - P.split(z_orc)[1]).max())}
    except Exception as exc:  # noqa: BLE001  (labelling only)
        rec["oracle_fixed_point"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
    return rec


def main():
    cases = make_cases()
    if "--quick" in sys.argv:
        cases = cases[:2] + cases[48:50]
    dyn_fn()  # derive once before forking
    with mpz.Pool(min(8, os.cpu_count() or 1)) as pool:
        out = pool.map(run_case, cases, chunksize=1)
    path = os.path.join(HERE, "converged_golden.json" if "--quick" not in sys.argv else "/tmp/converged_quick.json")
    with open(path, "w") as fh:
        json.dump({"generator": "tests/golden/gen_converged_golden.py (independent numpy NLP + scipy SLSQP + Newton-KKT polish)",
                   "scipy": __import__("scipy").__version__, "cases": out}, fh)
    same = sum(1 for r in out if r.get("oracle_fixed_point", {}).get("max_abs_du_vs_u_star", 1.0) <= 1e-5)
    print("wrote", path, len(out), "cases;", same, "where the oracle's fixed point is u* within 1e-5;",
          "worst kkt residual %.2e" % max(r["kkt_residual"] for r in out))


if __name__ == "__main__":
    main()
