"""CPU study (numpy, no GPU): where does the condensed QP lose accuracy at long horizons, and what would not cure it?
The first QP of B cold starts at window_length N, solved (a) densely with pivoting in double, (b) by the kernels' state
elimination in double, with selectable pieces carried in long double ('S' the terminal Schur complement and its solve,
'rho', 'y', 'psi' the products of transition matrices, 'rec' the state recovery), (c) with the terminal system in
square-root form (a QR of W D^-1/2 instead of its Gram matrix), (d) by a node-level Riccati sweep (node states stay
unknowns, elimination only inside an interval, terminal equalities by a range-space step in the last interval) -- each
against a long-double dense KKT solve.  Prints median / p99 / max of the error of du and dx per variant, and for the
Riccati variant the rank correlation of its error with the condition number of the last interval's Schur complement.
DESIGN.md 6.4 quotes the N = 160, B = 1500 run (5 minutes on 8 cores).
Usage: python tools/long_horizon_cpu_study.py [N] [B] [f32]      VARIANTS="dense KKT f64;riccati f64" selects variants"""
import sys, numpy as np
sys.path.insert(0, '.')
from oracle import oracle as orc
LD = np.longdouble
DYN = [1.0, 0.1, 0.25, 9.81, 0.05, 0.1, 0.02, 0.8, 100.0]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 160
B = int(sys.argv[2]) if len(sys.argv) > 2 else 200
SP = 10
BASE = np.float32 if (len(sys.argv) > 3 and sys.argv[3] == 'f32') else np.float64
S = N // SP + 1
L = S - 1
over = dict(max_iterations=1, relative_exit_tol=0.0, absolute_first_derivative_tol=0.0, window_length=N)
p = orc.default_opt_params(**over)


def lu_solve(K, b, dt):
    K = K.astype(dt).copy(); b = b.astype(dt).copy(); n = len(b)
    for i in range(n):
        piv = i + int(np.argmax(np.abs(K[i:, i])))
        if piv != i:
            K[[i, piv]] = K[[piv, i]]; b[[i, piv]] = b[[piv, i]]
        f = K[i + 1:, i] / K[i, i]
        K[i + 1:, i:] -= f[:, None] * K[i, i:][None, :]
        b[i + 1:] -= f * b[i]
    x = np.zeros(n, dt)
    for i in range(n - 1, -1, -1):
        x[i] = (b[i] - K[i, i + 1:] @ x[i + 1:]) / K[i, i]
    return x


def kkt(J, r, A, c, dt):
    dim, ne = J.shape[1], A.shape[0]
    J, r, A, c = (a.astype(dt) for a in (J, r, A, c))
    K = np.zeros((dim + ne, dim + ne), dt)
    K[:dim, :dim] = J.T @ J; K[:dim, dim:] = A.T; K[dim:, :dim] = A
    rhs = np.concatenate([-(J.T @ r), -c])
    return lu_solve(K, rhs, dt)


def condensed(Phi, Gam, cs, ci, e_term, Rw, Dg, u, u_prev, wu, wd, dt, hi=()):
    """state elimination as the kernels do it; `hi` names the pieces carried in long double:
    'S' (block sums, combine, group sums, LDL and its solve), 'psi' (the Psi products), 'rec' (state recovery)"""
    HI = LD if dt == np.float64 else np.float64
    Phi = [a.astype(dt) for a in Phi]; Gam = [a.astype(dt) for a in Gam]; cs = [a.astype(dt) for a in cs]
    ci, e_term, Rw, Dg, u = (a.astype(dt) for a in (ci, e_term, Rw, Dg, u))
    wu2, wd2 = dt(wu) ** 2, dt(wd) ** 2
    diag = np.array([wu2 + wd2 * (2 if k < N - 1 else 1) for k in range(N)], dt)
    d = np.zeros(N, dt); ups = np.zeros(N, dt)
    d[N - 1] = diag[N - 1]
    for k in range(N - 2, -1, -1):
        ups[k] = -wd2 / d[k + 1]
        d[k] = diag[k] + wd2 * ups[k]
    tp = HI if 'psi' in hi else dt
    Psi = [None] * L
    Psi[L - 1] = np.diag(Rw).astype(tp)
    for s in range(L - 2, -1, -1):
        Psi[s] = Psi[s + 1] @ Phi[s + 1].astype(tp)
    tS = HI if 'S' in hi else dt
    g = np.zeros(N, dt)
    for k in range(N):
        ul = u[k - 1] if k > 0 else dt(u_prev)
        g[k] = wu2 * u[k] + wd2 * (u[k] - ul)
        if k < N - 1:
            g[k] += wd2 * (u[k] - u[k + 1])
    hvv = [cs[s] - (Phi[0] @ ci if s == 0 else 0) for s in range(L)]
    Rt = np.zeros((N, 4), tp)
    for k in range(N):
        s = k // SP
        Rt[k] = Psi[s] @ Gam[s][:, k % SP].astype(tp)
    tw = tp
    W = np.zeros((N, 4), tw); gw = np.zeros(N, dt)
    W[N - 1] = Rt[N - 1]; gw[N - 1] = g[N - 1]
    for k in range(N - 2, -1, -1):
        W[k] = Rt[k] - ups[k].astype(tw) * W[k + 1]
        gw[k] = g[k] - ups[k] * gw[k + 1]
    tR = HI if 'rho' in hi else dt
    Sm = np.diag(Dg).astype(tS).copy(); rho = np.zeros(4, tR)
    for k in range(N):
        Sm += np.outer(W[k].astype(tS), W[k].astype(tS)) / d[k].astype(tS)
        rho += W[k].astype(tR) * gw[k].astype(tR) / d[k].astype(tR)
    hv = (Rw * e_term).astype(tR)
    for s in range(L):
        hv = hv + (Psi[s].astype(tR) @ hvv[s].astype(tR))
    if 'qr' in hi:
        # square-root form: Sm = Vt^T Vt with Vt = [V; sqrt(Dg)], V_k = W_k / sqrt(d_k); R from a QR of Vt (never the Gram matrix)
        sd = np.sqrt(d)
        V = (W.astype(dt) / sd[:, None]).astype(dt)
        Vt = np.vstack([V, np.diag(np.sqrt(Dg)).astype(dt)])
        if 'givens' in hi:
            R = np.zeros((4, 4), dt)
            for row in Vt:      # row append by Givens rotations, sequentially, as one thread would
                v = row.copy()
                for j in range(4):
                    if v[j] != 0:
                        a, b = R[j, j], v[j]
                        h = np.hypot(a, b); cth, sth = a / h, b / h
                        Rj = R[j].copy()
                        R[j] = cth * Rj + sth * v
                        v = -sth * Rj + cth * v
        else:
            R = np.linalg.qr(Vt, mode='r').astype(dt)
        ct = (gw / sd).astype(dt)
        # Qt = V R^-1 row by row (back substitution), z = R^-T hv - Qt^T ct
        Qt = np.zeros((N, 4), dt)
        for k in range(N):
            for j in range(4):
                Qt[k, j] = (V[k, j] - Qt[k, :j] @ R[:j, j]) / R[j, j]
        t = np.zeros(4, dt)     # R^T t = hv
        hvd = hv.astype(dt)
        for j in range(4):
            t[j] = (hvd[j] - R[:j, j] @ t[:j]) / R[j, j]
        z = t - Qt.T @ ct
        y = (-(sd * (ct + Qt @ z))).astype(dt)
        if 'sn' in hi:   # semi-normal equations: q = R^-1 R^-T (hv - rho), y = -(gw + W q), with only R from the QR
            rhs = (hv - rho).astype(dt)
            t2 = np.zeros(4, dt)
            for j in range(4):
                t2[j] = (rhs[j] - R[:j, j] @ t2[:j]) / R[j, j]
            q = np.zeros(4, dt)
            for j in range(3, -1, -1):
                q[j] = (t2[j] - R[j, j + 1:] @ q[j + 1:]) / R[j, j]
            y = (-(gw + W.astype(dt) @ q)).astype(dt)
    else:
        q = lu_solve(Sm, (hv - rho).astype(tS), tS)
        ty = HI if 'y' in hi else dt
        y = (-(gw.astype(ty) + W.astype(ty) @ q.astype(ty))).astype(dt)
    du = np.zeros(N, dt)
    du[0] = y[0] / d[0]
    for k in range(1, N):
        du[k] = y[k] / d[k] - ups[k - 1] * du[k - 1]
    tr = HI if 'rec' in hi else dt
    dx = [(-ci).astype(tr)]
    for s in range(L):
        dx.append(Phi[s].astype(tr) @ dx[s] + Gam[s].astype(tr) @ du[s * SP:(s + 1) * SP].astype(tr) + cs[s].astype(tr))
    return np.concatenate([np.concatenate([x.astype(dt) for x in dx]), du])



def riccati(Phi, Gam, cs, ci, e_term, Rw, Dg, u, u_prev, wu, wd, dt, lam=0.0):
    """node-level Riccati sweep: node states stay unknowns, elimination only inside an interval"""
    f = lambda a: np.asarray(a, dt)
    Phi = [f(a) for a in Phi]; Gam = [f(a) for a in Gam]; cs = [f(a) for a in cs]
    ci, e_term, Rw, Dg, u = (f(a) for a in (ci, e_term, Rw, Dg, u))
    wu2, wd2 = dt(wu) ** 2, dt(wd) ** 2
    g = np.zeros(N, dt)
    for k in range(N):
        ul = u[k - 1] if k > 0 else dt(u_prev)
        g[k] = wu2 * u[k] + wd2 * (u[k] - ul)
        if k < N - 1:
            g[k] += wd2 * (u[k] - u[k + 1])
    diag = np.array([wu2 + wd2 * (2 if k < N - 1 else 1) + dt(lam) for k in range(N)], dt)
    cost = Dg != 0
    eq = ~cost
    ne = int(eq.sum())
    # terminal value function on xi = [dx; delta]
    P = np.zeros((5, 5), dt); pv = np.zeros(5, dt)
    for t in range(4):
        if cost[t]:
            P[t, t] = Rw[t] ** 2; pv[t] = Rw[t] ** 2 * e_term[t]
    Ks = [None] * L; ks = [None] * L
    for s in range(L - 1, -1, -1):
        A = np.zeros((5, 5), dt); A[:4, :4] = Phi[s]
        Bm = np.zeros((5, SP), dt); Bm[:4] = Gam[s]; Bm[4, SP - 1] = 1
        b = np.zeros(5, dt); b[:4] = cs[s]
        T = np.diag(diag[s * SP:(s + 1) * SP]).astype(dt)
        for i in range(SP - 1):
            T[i, i + 1] = T[i + 1, i] = -wd2
        H = T + Bm.T @ P @ Bm
        G = Bm.T @ P @ A
        G[0, 4] += -wd2
        h = g[s * SP:(s + 1) * SP] + Bm.T @ (P @ b + pv)
        Lc = np.linalg.cholesky(H)
        def Hsolve(X):
            Y = np.linalg.solve(Lc, X)   # triangular in effect
            return np.linalg.solve(Lc.T, Y)
        K = -Hsolve(G); k0 = -Hsolve(h)
        if s == L - 1 and ne > 0:
            E = np.eye(4, dtype=dt)[eq]
            EG = E @ Gam[s]                          # ne x SP
            # constraint: EG D = r0 + R1 xi,  r0 = -e_E - E c,  R1 = -[E Phi, 0]
            r0 = -(e_term[eq]) - E @ cs[s]
            R1 = np.zeros((ne, 5), dt); R1[:, :4] = -(E @ Phi[s])
            HiEGt = Hsolve(EG.T)                     # SP x ne
            Sl = EG @ HiEGt                          # ne x ne
            riccati.cond = float(np.linalg.cond(Sl.astype(np.float64)))
            # D = (K xi + k0) + HiEGt Sl^-1 (r(xi) - EG (K xi + k0))
            Cx = np.linalg.solve(Sl, R1 - EG @ K)
            c0 = np.linalg.solve(Sl, r0 - EG @ k0)
            K = K + HiEGt @ Cx
            k0 = k0 + HiEGt @ c0
        Ks[s] = K; ks[s] = k0
        Pn = A.T @ P @ A + K.T @ H @ K + G.T @ K + K.T @ G
        pn = A.T @ (P @ b + pv) + K.T @ (H @ k0 + h) + G.T @ k0
        P = 0.5 * (Pn + Pn.T); pv = pn
    xi = np.zeros(5, dt); xi[:4] = -ci
    dx = [xi[:4].copy()]; du = np.zeros(N, dt)
    for s in range(L):
        D = Ks[s] @ xi + ks[s]
        du[s * SP:(s + 1) * SP] = D
        nx = Phi[s] @ xi[:4] + Gam[s] @ D + cs[s]
        xi = np.concatenate([nx, [D[-1]]])
        dx.append(nx.copy())
    return np.concatenate([np.concatenate(dx), du])

rng = np.random.default_rng(500 + 14)
x0s = np.stack([rng.uniform(-0.6, 0.6, B), rng.uniform(-np.pi, np.pi, B), rng.uniform(-1, 1, B), rng.uniform(-3, 3, B)])
x0s[1, ::2] = np.pi / 2 + rng.uniform(-0.5, 0.5, x0s[1, ::2].shape)
variants = [("dense KKT f64", None), ("riccati f64", 'ric'), ("condensed f64", ()), ("S", ('S',)), ("S,rho", ('S', 'rho')), ("S,rho,y", ('S','rho','y')),
            ("qr (householder)", ('qr',)), ("qr (givens rows)", ('qr', 'givens')), ("qr semi-normal", ('qr', 'givens', 'sn')), ("qr + psi hi", ('qr', 'psi')), ("rho,y", ('rho', 'y')), ("S,y", ('S', 'y')), ("S,rho,y,psi", ('S','rho','y','psi')), ("all", ('S','rho','y','psi','rec'))]
import os
if os.environ.get("VARIANTS"):
    keep = os.environ["VARIANTS"].split(";")
    variants = [v for v in variants if v[0] in keep]
conds, thL = [], []
res = {nm: [] for nm, _ in variants}
resx = {nm: [] for nm, _ in variants}
for b in range(B):
    x0 = x0s[:, b]
    out = orc.Optimization(p).step(x0, DYN, 0.0)
    z0 = out.guess
    r, c, J, A = orc.problem_eval(p, DYN, x0, 0.0, 0.0, z0)
    dz_true = kkt(J, r, A, c, LD)[:4 * S + N]
    Phi = [A[4 * s:4 * s + 4, 4 * s:4 * s + 4] for s in range(L)]
    Gam = [A[4 * s:4 * s + 4, 4 * S + s * SP:4 * S + (s + 1) * SP] for s in range(L)]
    cs = [c[4 * s:4 * s + 4] for s in range(L)]
    ci = c[4 * L:4 * L + 4]
    xT = z0[4 * (S - 1):4 * S]
    tgt = np.array([0.0, np.pi / 2, 0.0, 0.0])
    e_term = xT - tgt; e_term[1] = orc.mod_pi(e_term[1])
    Rw = np.array([150.0, 1.0, 1.0, 1.0]); Dg = np.array([1.0, 0.0, 0.0, 0.0])
    u = z0[4 * S:]
    args = (Phi, Gam, cs, ci, e_term, Rw, Dg, u, 0.0, 0.1, 0.1)
    for nm, hi in variants:
        dz = kkt(J, r, A, c, BASE)[:4 * S + N] if hi is None else (riccati(*args, BASE) if hi == 'ric' else condensed(*args, BASE, hi=hi))
        e = np.abs(dz.astype(LD) - dz_true)
        res[nm].append(float(e[4 * S:].max()))
        if hi == 'ric':
            conds.append(riccati.cond); thL.append(float(z0[4 * (S - 2) + 1]))
        resx[nm].append(float(e[:4 * S].max()))
print("N = %d, %d intervals, %d problems; |dz_true|_inf median %.1f" % (N, L, B, 0.0))
for nm, _ in variants:
    a = np.array(res[nm]); ax = np.array(resx[nm])
    print("%-26s |du err| median %.2e p99 %.2e max %.2e   |dx err| median %.2e p99 %.2e max %.2e" %
          (nm, np.median(a), np.quantile(a, .99), a.max(), np.median(ax), np.quantile(ax, .99), ax.max()))

if conds:
    a = np.array(res["riccati f64"]); c = np.array(conds); th = np.array(thL)
    o = np.argsort(-a)[:8]
    print("worst lanes: err", a[o], "cond(S_loc)", c[o], "|cos th_{L-1}|", np.abs(np.cos(th[o])))
    print("median cond", np.median(c), "spearman(log err, log cond)", np.corrcoef(np.argsort(np.argsort(a)), np.argsort(np.argsort(c)))[0, 1])
