#!/usr/bin/env python3
"""Dynamics generator: SymPy Lagrangian -> C (oracle) and HIP (kernels) source.

Counterpart of the reference's offline code generator (symbolic/generate.py + wrenfold, which is not
installable here).  For a cart with a chain of point-mass poles it derives the
Euler-Lagrange equations in the form

    M(q) q'' = F(q, q', u)

and emits, after common-subexpression elimination, one function that evaluates
    M (symmetric, nq x nq), F (nq), dF/dx (nq x 2nq), dM/dq_j for every angle q_j.
The accelerations and their Jacobians are then obtained numerically by the (hand-written) generic
solver next to the generated code:   a = M^-1 F,   da/dx_c = M^-1 (dF/dx_c - (dM/dx_c) a),
da/du = M^-1 dF/du, which is cheaper and better conditioned than emitting the closed-form inverse the
reference's generator emits (symbolic/sympy_utils.py:43-50).

Model generated today: the double pendulum of symbolic/dynamics_double.py:25-107
(state [b_x, th_1, th_2, b_x', th_1', th_2'], params (m_b, m_1, m_2, l_1, l_2, g), no dissipation).

Usage (from the repository root):  python tools/gen_dynamics.py
Writes   oracle/double_pendulum_gen.inc            (C, double)
         cart-pole-mpc_amd/csrc/double_pendulum_gen.hpp   (HIP device code, templated on the scalar)
"""
import os

import sympy as sp
from sympy.printing.c import C99CodePrinter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def derive_double():
    bx, th1, th2, v, w1, w2, u = sp.symbols("b_x th_1 th_2 b_x_dot th_1_dot th_2_dot u", real=True)
    m_b, m_1, m_2, l_1, l_2, g = sp.symbols("m_b m_1 m_2 l_1 l_2 g", real=True)
    q, qd = [bx, th1, th2], [v, w1, w2]
    qdd = sp.symbols("a0 a1 a2", real=True)

    def ddt(e):
        return sum(sp.diff(e, q[i]) * qd[i] + sp.diff(e, qd[i]) * qdd[i] for i in range(3))

    # dynamics_double.py:52-59
    b = sp.Matrix([bx, 0])
    p1 = b + sp.Matrix([sp.cos(th1), sp.sin(th1)]) * l_1
    p2 = p1 + sp.Matrix([sp.cos(th2), sp.sin(th2)]) * l_2
    bd, p1d, p2d = b.applyfunc(ddt), p1.applyfunc(ddt), p2.applyfunc(ddt)
    # dynamics_double.py:61-81
    T = (m_b * bd.dot(bd) + m_1 * p1d.dot(p1d) + m_2 * p2d.dot(p2d)) / 2
    V = g * m_1 * p1[1] + g * m_2 * p2[1]
    L = T - V
    # dynamics_double.py:89-93 (control force on the base only)
    el = [ddt(sp.diff(L, qd[i])) - sp.diff(L, q[i]) for i in range(3)]
    el[0] = el[0] - u
    el = [sp.expand(e) for e in el]
    M = sp.Matrix(3, 3, lambda i, j: sp.simplify(sp.diff(el[i], qdd[j])))
    F = sp.Matrix([sp.simplify(-(el[i].subs({a: 0 for a in qdd}))) for i in range(3)])
    return dict(name="double_pendulum", nq=3, q=q, qd=qd, u=u, params=[m_b, m_1, m_2, l_1, l_2, g], M=M, F=F,
                angles=[1, 2])


def build_outputs(model):
    nq, q, qd, u, M, F = model["nq"], model["q"], model["qd"], model["u"], model["M"], model["F"]
    x = q + qd
    outs = []
    for i in range(nq):
        for j in range(nq):
            outs.append(("M[%d]" % (i * nq + j), M[i, j]))
    for i in range(nq):
        outs.append(("F[%d]" % i, F[i]))
    for i in range(nq):
        for c in range(2 * nq):
            outs.append(("dFdx[%d]" % (i * 2 * nq + c), sp.diff(F[i], x[c])))
    for n, j in enumerate(model["angles"]):
        for i in range(nq):
            for k in range(nq):
                outs.append(("dM%d[%d]" % (n + 1, i * nq + k), sp.diff(M[i, k], q[j])))
    # trig of the angles -> symbols, so that sin/cos are evaluated once
    trig = {}
    for j in model["angles"]:
        s, c = sp.symbols("s%d c%d" % (j, j), real=True)
        trig[sp.sin(q[j])] = s
        trig[sp.cos(q[j])] = c
    exprs = [sp.expand_trig(e).subs(trig) for _, e in outs]
    repl, red = sp.cse(exprs, symbols=sp.numbered_symbols("t"), optimizations="basic")
    return [n for n, _ in outs], trig, repl, red


class _Printer(C99CodePrinter):
    def __init__(self, scalar):
        super().__init__()
        self.scalar = scalar

    def _print_Pow(self, e):
        if e.exp.is_Integer and 2 <= int(e.exp) <= 4:
            b = self._print(e.base)
            if not e.base.is_Atom:
                b = "(" + b + ")"
            return "*".join([b] * int(e.exp))
        return super()._print_Pow(e)

    def _print_Float(self, e):
        return "%s(%s)" % (self.scalar, C99CodePrinter._print_Float(self, e))

    def _print_Integer(self, e):
        return "%s(%d)" % (self.scalar, int(e)) if self.scalar != "double" else "%d.0" % int(e)

    def _print_Rational(self, e):
        return "(%s(%d)/%s(%d))" % (self.scalar, e.p, self.scalar, e.q)


def emit(model, scalar, header):
    names, trig, repl, red = build_outputs(model)
    pr = _Printer(scalar)
    nq = model["nq"]
    lines = list(header)
    sig = ("static void {n}_terms(const double* p, const double* x, double u, double* M, double* F, double* dFdx, "
           "double* dM1, double* dM2)").format(n=model["name"])
    if scalar != "double":
        sig = ("template <typename R>\n__device__ __forceinline__ void {n}_terms(const R* p, const R* x, const R u, R* M, "
               "R* F, R* dFdx, R* dM1, R* dM2)").format(n=model["name"])
    lines.append(sig + " {")
    for i, s in enumerate(model["params"]):
        lines.append("  const %s %s = p[%d];" % (scalar, s, i))
    xs = model["q"] + model["qd"]
    for i, s in enumerate(xs):
        lines.append("  const %s %s = x[%d];" % (scalar, s, i))
    for fn, sym in trig.items():
        ang = fn.args[0]
        call = ("sin" if fn.func == sp.sin else "cos")
        if scalar == "double":
            lines.append("  const double %s = %s(%s);" % (sym, call, ang))
    if scalar != "double":
        for j in model["angles"]:
            a = model["q"][j]
            lines.append("  R s%d, c%d;" % (j, j))
            lines.append("  Math<R>::sincos(%s, s%d, c%d);" % (a, j, j))
    for s, e in repl:
        lines.append("  const %s %s = %s;" % (scalar, s, pr.doprint(e)))
    for n, e in zip(names, red):
        lines.append("  %s = %s;" % (n, pr.doprint(e)))
    lines.append("  (void)%s; (void)%s;" % (model["q"][0], "u"))
    lines.append("}")
    return "\n".join(lines) + "\n"


def main():
    model = derive_double()
    banner = ["// GENERATED by tools/gen_dynamics.py from the Lagrangian of symbolic/dynamics_double.py:25-107 -- do not edit.",
              "// M(q) q'' = F(q, q', u): M[9] row-major, F[3], dFdx[3x6] row-major (x = [q, q']), dM1 = dM/dth_1, dM2 = dM/dth_2.",
              "// p = {m_b, m_1, m_2, l_1, l_2, g}; x = {b_x, th_1, th_2, b_x', th_1', th_2'}."]
    c_code = emit(model, "double", banner)
    with open(os.path.join(ROOT, "oracle", "double_pendulum_gen.inc"), "w") as fh:
        fh.write(c_code)
    hip = emit(model, "R", banner + ["#pragma once", "namespace cpmpc {"]) + "}  // namespace cpmpc\n"
    with open(os.path.join(ROOT, "cart-pole-mpc_amd", "csrc", "double_pendulum_gen.hpp"), "w") as fh:
        fh.write(hip)
    print("M =", model["M"])
    print("F =", model["F"])
    print("wrote oracle/double_pendulum_gen.inc (%d lines), csrc/double_pendulum_gen.hpp" % c_code.count("\n"))


if __name__ == "__main__":
    main()
