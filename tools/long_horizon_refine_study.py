#!/usr/bin/env python3
"""CPU study (numpy, no GPU; VERDICT r5 item 4's prototype): iterative refinement of the FULL KKT system of the first QP at a
long horizon, with the kernels' condensed solve (state elimination, double) as the approximate inverse and the residual of
every KKT equation -- stationarity in the nodes and the controls, shooting defects, initial and terminal rows; node states and
multipliers are unknowns of their own -- evaluated (a) in double, (b) in long double ("mixed precision"), beside (c) the
restricted form the kernels' pass has (controls' stationarity and terminal rows only, the rest satisfied by construction).  Error of du against a long-double dense KKT solve after 0, 1, 2, 3 passes, beside the dense pivoted LU
in double (what the CPU check does).  Answers: does refinement converge at all (is the condensed solve a contraction on the tail
lanes), where does (a) stop, and would (b) -- a third solver in all but name -- beat the dense solve?
Usage: python tools/long_horizon_refine_study.py [N] [B]        (N = 160, B = 400: ~4 minutes on 8 cores)"""
import sys

import numpy as np

sys.argv = [sys.argv[0]] + sys.argv[1:3] + ([] if len(sys.argv) > 3 else [])
sys.path.insert(0, '.')
sys.path.insert(0, 'tools')
import os
os.environ.setdefault("VARIANTS", "none")          # import the study's helpers without running its sweep
_argv = sys.argv
sys.argv = [sys.argv[0], sys.argv[1] if len(sys.argv) > 1 else "160", "1"]
import long_horizon_cpu_study as st                 # noqa: E402  (N, S, L, SP, kkt, lu_solve, condensed, p, orc, DYN)
sys.argv = _argv
LD = np.longdouble
N, S, L, SP = st.N, st.S, st.L, st.SP
B = int(sys.argv[2]) if len(sys.argv) > 2 else 400
orc = st.orc


def condensed_general(Phi, Gam, cs, ci, e_term, Rw, Dg, g, wu, wd, dt=np.float64):
    """The kernels' elimination for GIVEN gradient g of the control rows (instead of forming it from u), returning du, dx and
    the terminal multipliers q -- the approximate inverse of the KKT system for any right-hand side."""
    f = lambda a: np.asarray(a, dt)   # noqa: E731
    Phi = [f(a) for a in Phi]; Gam = [f(a) for a in Gam]; cs = [f(a) for a in cs]
    ci, e_term, Rw, Dg, g = (f(a) for a in (ci, e_term, Rw, Dg, g))
    wu2, wd2 = dt(wu) ** 2, dt(wd) ** 2
    diag = np.array([wu2 + wd2 * (2 if k < N - 1 else 1) for k in range(N)], dt)
    d = np.zeros(N, dt); ups = np.zeros(N, dt)
    d[N - 1] = diag[N - 1]
    for k in range(N - 2, -1, -1):
        ups[k] = -wd2 / d[k + 1]
        d[k] = diag[k] + wd2 * ups[k]
    Psi = [None] * L
    Psi[L - 1] = np.diag(Rw).astype(dt)
    for s in range(L - 2, -1, -1):
        Psi[s] = Psi[s + 1] @ Phi[s + 1]
    hvv = [cs[s] - (Phi[0] @ ci if s == 0 else 0) for s in range(L)]
    W = np.zeros((N, 4), dt); gw = np.zeros(N, dt)
    for k in range(N - 1, -1, -1):
        s = k // SP
        Rt = Psi[s] @ Gam[s][:, k % SP]
        W[k] = Rt - (ups[k] * W[k + 1] if k < N - 1 else 0)
        gw[k] = g[k] - (ups[k] * gw[k + 1] if k < N - 1 else 0)
    Sm = np.diag(Dg).astype(dt).copy(); rho = np.zeros(4, dt)
    for k in range(N):
        Sm += np.outer(W[k], W[k]) / d[k]
        rho += W[k] * gw[k] / d[k]
    hv = Rw * e_term
    for s in range(L):
        hv = hv + Psi[s] @ hvv[s]
    q = st.lu_solve(Sm, hv - rho, dt)
    y = -(gw + W @ q)
    du = np.zeros(N, dt)
    du[0] = y[0] / d[0]
    for k in range(1, N):
        du[k] = y[k] / d[k] - ups[k - 1] * du[k - 1]
    dx = [(-ci).astype(dt)]
    for s in range(L):
        dx.append(Phi[s] @ dx[s] + Gam[s] @ du[s * SP:(s + 1) * SP] + cs[s])
    return du, dx, q


def multipliers(Phi, dxL, e_term, Rw, Dg, q, extra=None):
    """nu of the defect rows from the terminal vector m (cost rows: Rw^2 (dx_L + e); equality rows: q), walked back through Phi^T
    (+ `extra[s]`, the residual of node s+1's stationarity, in the correction solve); then the initial rows' and the terminal ones."""
    cost = Dg != 0
    m = np.where(cost, Rw * Rw * (dxL + e_term), q)
    nu = [None] * L
    nu[L - 1] = m + (extra[L - 1] if extra is not None else 0)
    for s in range(L - 2, -1, -1):
        nu[s] = Phi[s + 1].T @ nu[s + 1] + (extra[s] if extra is not None else 0)
    return nu


def pack(dx, du, nu, nu_init, nu_term):
    return np.concatenate([np.concatenate(dx), du, np.concatenate(nu), nu_init, nu_term])


def run(prec, restricted=False):
    """error of du after 0..3 refinement passes with residuals in `prec`, per problem.  restricted: as the kernels' pass does it
    (round 4): the node states and the multipliers are NOT unknowns of their own -- dx is whatever the forward recursion gave
    (defect and initial rows count as satisfied), the multipliers are re-derived from the terminal ones by the adjoint
    recursion (node stationarity counts as satisfied) -- only the controls' stationarity and the terminal rows are measured."""
    rng = np.random.default_rng(500 + 14)
    x0s = np.stack([rng.uniform(-0.6, 0.6, B), rng.uniform(-np.pi, np.pi, B), rng.uniform(-1, 1, B), rng.uniform(-3, 3, B)])
    x0s[1, ::2] = np.pi / 2 + rng.uniform(-0.5, 0.5, x0s[1, ::2].shape)
    errs = [[] for _ in range(4)]
    dense = []
    for b in range(B):
        x0 = x0s[:, b]
        out = orc.Optimization(st.p).step(x0, st.DYN, 0.0)
        z0 = out.guess
        r, c, J, A = orc.problem_eval(st.p, st.DYN, x0, 0.0, 0.0, z0)
        dim, ne = J.shape[1], A.shape[0]
        y_true = st.kkt(J, r, A, c, LD)
        dense.append(float(np.abs(st.kkt(J, r, A, c, np.float64)[4 * S:dim].astype(LD) - y_true[4 * S:dim]).max()))
        K = np.zeros((dim + ne, dim + ne), LD)
        K[:dim, :dim] = J.astype(LD).T @ J.astype(LD); K[:dim, dim:] = A.T; K[dim:, :dim] = A
        rhs = np.concatenate([-(J.astype(LD).T @ r.astype(LD)), -c.astype(LD)])
        Phi = [A[4 * s:4 * s + 4, 4 * s:4 * s + 4] for s in range(L)]
        Gam = [A[4 * s:4 * s + 4, 4 * S + s * SP:4 * S + (s + 1) * SP] for s in range(L)]
        cs = [c[4 * s:4 * s + 4] for s in range(L)]
        ci = c[4 * L:4 * L + 4]
        xT = z0[4 * (S - 1):4 * S]
        e_term = xT - np.array([0.0, np.pi / 2, 0.0, 0.0]); e_term[1] = orc.mod_pi(e_term[1])
        Rw = np.array([150.0, 1.0, 1.0, 1.0]); Dg = np.array([1.0, 0.0, 0.0, 0.0])
        eq = Dg == 0
        u = z0[4 * S:]
        wu = wd = 0.1
        g = np.zeros(N)
        for k in range(N):
            ul = u[k - 1] if k > 0 else 0.0
            g[k] = wu * wu * u[k] + wd * wd * (u[k] - ul) + (wd * wd * (u[k] - u[k + 1]) if k < N - 1 else 0.0)
        du, dx, q = condensed_general(Phi, Gam, cs, ci, e_term, Rw, Dg, g, wu, wd)
        nu = multipliers(Phi, dx[L], e_term, Rw, Dg, q)
        y = pack(dx, du, nu, -(Phi[0].T @ nu[0]), q[eq]).astype(LD)
        errs[0].append(float(np.abs(y[4 * S:dim] - y_true[4 * S:dim]).max()))
        for it in range(1, 4):
            res = (K.astype(prec) @ y.astype(prec) - rhs.astype(prec)).astype(np.float64)   # residual of every equation
            r_x = [res[4 * s:4 * s + 4] for s in range(S)]
            r_u = res[4 * S:dim]
            r_def = [res[dim + 4 * s:dim + 4 * s + 4] for s in range(L)]
            r_init = res[dim + 4 * L:dim + 4 * L + 4]
            r_eq = np.zeros(4); r_eq[eq] = res[dim + 4 * L + 4:]
            if restricted:
                r_def = [np.zeros(4) for _ in range(L)]
                r_init = np.zeros(4)
                r_x = [np.zeros(4) for _ in range(S)]
            # adjoint of the node-stationarity residuals, folded into the controls' gradient
            bb = [None] * L
            bb[L - 1] = r_x[L]
            for s in range(L - 2, -1, -1):
                bb[s] = Phi[s + 1].T @ bb[s + 1] + r_x[s + 1]
            gt = r_u.copy()
            for s in range(L):
                gt[s * SP:(s + 1) * SP] += Gam[s].T @ bb[s]
            ddu, ddx, dq = condensed_general(Phi, Gam, r_def, r_init, np.where(eq, r_eq, 0.0), Rw, Dg, gt, wu, wd)
            # multipliers of the correction: terminal part from the correction, plus the residuals' adjoint
            dm = np.where(eq, dq, Rw * Rw * ddx[L])
            dnu = [None] * L
            dnu[L - 1] = dm + bb[L - 1]
            for s in range(L - 2, -1, -1):
                dnu[s] = Phi[s + 1].T @ (dnu[s + 1] - bb[s + 1]) + bb[s]
            dnu_init = -r_x[0] - Phi[0].T @ dnu[0]
            # dq is the multiplier of the correction's terminal rows with their shifted right-hand side: the equality rows'
            # own multipliers absorb the residual of the terminal node's stationarity
            y = y + pack(ddx, ddu, dnu, dnu_init, (dq - 0.0)[eq]).astype(LD)
            errs[it].append(float(np.abs(y[4 * S:dim] - y_true[4 * S:dim]).max()))
    return errs, dense


def main():
    print("N = %d (%d intervals), %d problems; |du - du_true|_inf against a long-double dense KKT solve: median / p99 / max" % (N, L, B))
    for name, prec, restricted in (("controls + terminal residuals only, double (the kernels' pass)", np.float64, True),
                                   ("all KKT residuals in double", np.float64, False),
                                   ("all KKT residuals in long double (mixed precision)", LD, False)):
        errs, dense = run(prec, restricted)
        if restricted:
            a = np.array(dense)
            print("%-72s %.2e / %.2e / %.2e" % ("dense pivoted LU in double (the CPU check)", np.median(a), np.quantile(a, .99), a.max()))
        for it, e in enumerate(errs):
            a = np.array(e)
            if it or restricted:
                print("%-72s %.2e / %.2e / %.2e" % ("%s, %d passes" % (name, it) if it else "condensed solve alone", np.median(a), np.quantile(a, .99), a.max()))


if __name__ == "__main__":
    main()
