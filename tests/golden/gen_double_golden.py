"""Golden vectors for the cart + double pendulum from an INDEPENDENT SymPy/mpmath evaluation.

The reference specifies this model in symbolic/dynamics_double.py:25-148 but ships no generated header,
so there is nothing of the reference's to run; wrenfold is not installable here.  This script derives the
Euler-Lagrange equations with SymPy, solves them for the accelerations SYMBOLICALLY (closed-form 3x3
inverse, as the reference's generator does, symbolic/sympy_utils.py:43-50) and differentiates that
closed form -- a different route from tools/gen_dynamics.py (which emits M, F and their partials and
leaves the solve to hand-written code) -- then evaluates with mpmath at 40 digits.
Output: double_golden.json.   Run from the repository root: python tests/golden/gen_double_golden.py
"""
import json
import os

import mpmath as mp
import numpy as np
import sympy as sp

mp.mp.dps = 40


def derive():
    bx, th1, th2, v, w1, w2, u = sp.symbols("b_x th_1 th_2 b_x_dot th_1_dot th_2_dot u", real=True)
    m_b, m_1, m_2, l_1, l_2, g = sp.symbols("m_b m_1 m_2 l_1 l_2 g", real=True)
    q, qd = [bx, th1, th2], [v, w1, w2]
    a = list(sp.symbols("a0 a1 a2", real=True))

    def ddt(e):
        return sum(sp.diff(e, q[i]) * qd[i] + sp.diff(e, qd[i]) * a[i] for i in range(3))

    b = sp.Matrix([bx, 0])
    p1 = b + sp.Matrix([sp.cos(th1), sp.sin(th1)]) * l_1
    p2 = p1 + sp.Matrix([sp.cos(th2), sp.sin(th2)]) * l_2
    bd, p1d, p2d = b.applyfunc(ddt), p1.applyfunc(ddt), p2.applyfunc(ddt)
    T = (m_b * bd.dot(bd) + m_1 * p1d.dot(p1d) + m_2 * p2d.dot(p2d)) / 2
    V = g * m_1 * p1[1] + g * m_2 * p2[1]
    L = T - V
    el = [ddt(sp.diff(L, qd[i])) - sp.diff(L, q[i]) for i in range(3)]
    el[0] -= u
    sol = sp.solve(el, a, dict=True)[0]
    f = sp.Matrix([v, w1, w2, sol[a[0]], sol[a[1]], sol[a[2]]])
    x = sp.Matrix([bx, th1, th2, v, w1, w2])
    args = [m_b, m_1, m_2, l_1, l_2, g, bx, th1, th2, v, w1, w2, u]
    return sp.lambdify(args, [f, f.jacobian(x), f.jacobian(sp.Matrix([u]))], modules="mpmath")


def main():
    fn = derive()
    rng = np.random.default_rng(20241102)
    cases = []
    for i in range(40):
        prm = [1.0, 0.1, 0.1, 0.25, 0.2, 9.81] if i < 8 else \
            [rng.uniform(0.5, 2.0), rng.uniform(0.05, 0.5), rng.uniform(0.05, 0.5), rng.uniform(0.15, 0.6),
             rng.uniform(0.15, 0.6), 9.81]
        x = [rng.uniform(-1, 1), rng.uniform(-4, 4), rng.uniform(-4, 4), rng.uniform(-2, 2), rng.uniform(-6, 6),
             rng.uniform(-6, 6)]
        if i == 0:
            x = [0.0, np.pi / 2, np.pi / 2, 0.0, 0.0, 0.0]  # upright equilibrium
        uu = rng.uniform(-20, 20) if i else 0.0
        f, Jx, Ju = fn(*[mp.mpf(float(t)) for t in prm + x + [uu]])
        cases.append({"params": [float(t) for t in prm], "x": [float(t) for t in x], "u": float(uu),
                      "f": [float(f[k]) for k in range(6)],
                      "J_x": [[float(Jx[r, c]) for c in range(6)] for r in range(6)],
                      "J_u": [float(Ju[k]) for k in range(6)]})
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "double_golden.json")
    with open(out, "w") as fh:
        json.dump({"generator": "tests/golden/gen_double_golden.py", "mp_dps": mp.mp.dps, "cases": cases}, fh, indent=1)
    print("wrote", out, len(cases))


if __name__ == "__main__":
    main()
