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

Models generated:
  double  the double pendulum of symbolic/dynamics_double.py:25-107 (state [b_x, th_1, th_2, b_x', th_1', th_2'],
          params (m_b, m_1, m_2, l_1, l_2, g), no dissipation): M, F, dF/dx, dM/dth as above.
  single  the model that is on the hot path, symbolic/dynamics_single.py:58-143: smoothed Coulomb friction
          (tanh(v / max(v_mu_b, 1e-6))), cubic air-drag power with the |v|^2 > 0 guard, bumper springs max(0, .),
          external forces on base and mass.  Emitted in the reference generator's own form -- accelerations through
          the closed-form 2x2 inverse, Jacobians by symbolic differentiation (symbolic/sympy_utils.py:43-50) -- with
          the non-smooth pieces carried as opaque symbols plus their derivative rules, so the output is straight-line
          code with per-lane selects, and with every parameter-only sub-expression split off into a constants
          function evaluated once on the host.  -DCPMPC_GENERATED_SINGLE=1 builds the kernels on it instead of the
          hand-written cartpole_accel (cartpole_device.hpp); tests/test_generated_dynamics.py holds the two together.

Usage (from the repository root):  python tools/gen_dynamics.py
Writes   oracle/double_pendulum_gen.inc, oracle/single_pendulum_gen.inc          (C, double)
         cart-pole-mpc_amd/csrc/double_pendulum_gen.hpp, single_pendulum_gen.hpp  (HIP device code, scalar-templated)
"""
import os

import sympy as sp
from sympy.printing.c import C99CodePrinter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def write_if_changed(path, text):
    """Leave an up-to-date file alone (its mtime is a build dependency of the libraries)."""
    if os.path.exists(path):
        with open(path) as fh:
            if fh.read() == text:
                return
    with open(path, "w") as fh:
        fh.write(text)


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
            return "(" + "*".join([b] * int(e.exp)) + ")"   # parenthesised: it may sit in a denominator
        return super()._print_Pow(e)

    def _print_Float(self, e):
        lit = C99CodePrinter._print_Float(self, e)
        return lit if self.scalar == "double" else "%s(%s)" % (self.scalar, lit)

    def _print_Integer(self, e):
        return "%s(%d)" % (self.scalar, int(e)) if self.scalar != "double" else "%d.0" % int(e)

    def _print_Rational(self, e):
        if self.scalar == "double":
            return "(%d.0/%d.0)" % (e.p, e.q)
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


def emit_hip_sc(model, header):
    """The HIP form of a mass-matrix model (round 6): what the hand-written 4-state model has, from the generator --
      * every output is read as a polynomial in the per-lane atoms (sines / cosines of the angles and of their differences,
        velocities, the control); its coefficients depend on the parameters only and are folded into a constants struct
        evaluated once (on the host, in the host's precision) by <name>_gen_consts(): kernel-argument segment / SGPRs,
      * the sines and cosines of the angles are INPUTS (<name>_terms_sc), so that the stages of an RK4 step can take them
        by one rotation from stage 1's pair (models.hpp: StepCache); sin / cos of an angle difference is one product pair,
      * the structure of the outputs as compile-time masks (<Name>GenSparsity): entries that are identically zero are
        neither written nor read by the solver next to this code, columns of da/dx that vanish identically (a state that
        does not enter the dynamics) are known to the RK4 sensitivity chain (models.hpp: stage_chain_m)."""
    nq, q, qd, u, M, F = model["nq"], model["q"], model["qd"], model["u"], model["M"], model["F"]
    name = model["name"]
    cname = "".join(w.capitalize() for w in name.split("_"))
    x = q + qd
    nx = 2 * nq
    outs = []
    for i in range(nq):
        for j in range(nq):
            outs.append(("M[%d]" % (i * nq + j), M[i, j]))
    for i in range(nq):
        outs.append(("F[%d]" % i, F[i]))
    for i in range(nq):
        for c in range(nx):
            outs.append(("dFdx[%d]" % (i * nx + c), sp.diff(F[i], x[c])))
    for n, j in enumerate(model["angles"]):
        for i in range(nq):
            for k in range(nq):
                outs.append(("dM%d[%d]" % (n + 1, i * nq + k), sp.diff(M[i, k], q[j])))
    # per-lane atoms: trig of the angles (inputs) and of their differences (one product pair each, emitted first)
    atoms, pre = {}, []
    for j in model["angles"]:
        sj, cj = sp.symbols("s%d c%d" % (j, j), real=True)
        atoms[sp.sin(q[j])] = sj
        atoms[sp.cos(q[j])] = cj
    ang = model["angles"]
    for ia in range(len(ang)):
        for ib in range(ia + 1, len(ang)):
            i, j = ang[ia], ang[ib]
            sd, cd = sp.symbols("sd%d%d cd%d%d" % (i, j, i, j), real=True)
            si, ci, sj, cj = atoms[sp.sin(q[i])], atoms[sp.cos(q[i])], atoms[sp.sin(q[j])], atoms[sp.cos(q[j])]
            atoms[sp.sin(q[i] - q[j])] = sd
            atoms[sp.cos(q[i] - q[j])] = cd
            atoms[sp.sin(q[j] - q[i])] = -sd
            atoms[sp.cos(q[j] - q[i])] = cd
            pre.append((sd, si * cj - ci * sj))
            pre.append((cd, ci * cj + si * sj))
    lane_syms = [v for v in atoms.values() if v.is_Symbol] + list(qd) + [u]
    lane_syms = list(dict.fromkeys(lane_syms))
    kconst = {}     # canonical parameter-only expression -> constant symbol

    def const_of(e):
        e = sp.factor(e)
        num, rest = e.as_coeff_Mul()
        if rest == 1:
            return sp.Integer(1), num
        key = sp.expand(rest)
        lead = sp.Poly(key, *model["params"]).coeffs()[0]
        if lead < 0:                      # one constant for +k and -k
            key, num = -key, -num
        if key not in kconst:
            kconst[key] = sp.Symbol("k%d" % len(kconst), real=True)
        return kconst[key], num

    lane_exprs = []
    for _, e in outs:
        e = sp.expand(e.subs(atoms))
        if e.has(sp.sin) or e.has(sp.cos):
            raise ValueError("trigonometric term outside the atoms: %s" % e)
        if e == 0:
            lane_exprs.append(sp.Integer(0))
            continue
        poly = sp.Poly(e, *lane_syms)
        acc = 0
        for mono, coeff in poly.terms():
            ksym, num = const_of(coeff)
            term = num * ksym
            for sym, pw in zip(lane_syms, mono):
                term = term * sym**pw
            acc = acc + term
        lane_exprs.append(acc)
    repl, red = sp.cse(lane_exprs, symbols=sp.numbered_symbols("t"), optimizations="basic")
    names = [n for n, _ in outs]
    pr, prc = _Printer("R"), _Printer("P")
    const_syms = list(kconst.values())
    lines = list(header)
    lines += ["template <typename R>", "struct %sGenConsts {" % cname, "  R k[%d];" % max(len(const_syms), 1), "};",
              "// the parameter-only coefficients, once per parameter set (host: in double; per-problem parameters: per lane)",
              "template <typename R, typename P>",
              "__host__ __device__ inline %sGenConsts<R> %s_gen_consts(const P* p) {" % (cname, name), "  %sGenConsts<R> K;" % cname]
    for i, sym in enumerate(model["params"]):
        lines.append("  const P %s = p[%d];" % (sym, i))
    for i, (ce, sym) in enumerate(kconst.items()):
        lines.append("  K.k[%d] = R(%s);" % (i, prc.doprint(ce)))
    lines.append("  " + " ".join("(void)%s;" % s_ for s_ in model["params"]))
    lines += ["  return K;", "}"]
    nz = {n: (e != 0) for n, e in zip(names, red)}

    def mask(prefix, count):
        return "{" + ", ".join("true" if nz["%s[%d]" % (prefix, i)] else "false" for i in range(count)) + "}"

    ja_zero = []
    for c in range(nx):
        z = all(not nz["dFdx[%d]" % (i * nx + c)] for i in range(nq))
        if c in model["angles"]:
            n_ = model["angles"].index(c) + 1
            z = z and all(not nz["dM%d[%d]" % (n_, i)] for i in range(nq * nq))
        ja_zero.append(z)
    lines += ["// which outputs are not identically zero (the others are never written: do not read them), and the columns of",
              "// da/dx that vanish identically (bit c: state component c does not enter the dynamics)",
              "struct %sGenSparsity {" % cname,
              "  static constexpr bool dFdx[%d] = %s;" % (nq * nx, mask("dFdx", nq * nx))]
    for n_ in range(len(model["angles"])):
        lines.append("  static constexpr bool dM%d[%d] = %s;" % (n_ + 1, nq * nq, mask("dM%d" % (n_ + 1), nq * nq)))
    lines += ["  static constexpr unsigned ja_zero_cols = 0x%xu;" % sum(1 << c for c in range(nx) if ja_zero[c]), "};"]
    args = ", ".join("const R s%d, const R c%d" % (j, j) for j in model["angles"])
    lines += ["template <typename R>",
              "__device__ __forceinline__ void %s_terms_sc(const %sGenConsts<R>& K, %s, const R* x, const R u, R* M, R* F, "
              "R* dFdx, R* dM1, R* dM2) {" % (name, cname, args)]
    used = set()
    for _, e in repl:
        used |= e.free_symbols
    for e in red:
        used |= e.free_symbols
    for i, sym in enumerate(x):
        if sym in used:
            lines.append("  const R %s = x[%d];" % (sym, i))
    for i, sym in enumerate(const_syms):
        lines.append("  const R %s = K.k[%d];" % (sym, i))
    for sym, e in pre:
        if sym in used:
            lines.append("  const R %s = %s;" % (sym, pr.doprint(e)))
    for sym, e in repl:
        lines.append("  const R %s = %s;" % (sym, pr.doprint(e)))
    for n, e in zip(names, red):
        if e != 0:
            lines.append("  %s = %s;" % (n, pr.doprint(e)))
    lines.append("  (void)x; (void)u; (void)dFdx; (void)dM1; (void)dM2;")
    lines.append("}")
    return "\n".join(lines) + "\n"


# ------------------------------------------------------------------------------------------------------------------
# cart + single pole with dissipation, springs and external forces (symbolic/dynamics_single.py:58-143)
# ------------------------------------------------------------------------------------------------------------------
def derive_single(factored_velocity=False):
    """Accelerations a = (b_x'', th_1'') and their partials wrt x = (b_x, th_1, b_x', th_1') and u, as expressions in
    the state, the control, the external forces, the parameters and these helper symbols (evaluated by the emitted
    prologue, differentiated by the rules below):
        s, c     sin th_1, cos th_1
        tv       tanh(b_x' * ivm),  ivm = 1 / max(v_mu_b, 1e-6)          d tv / d b_x' = (1 - tv^2) ivm
        n, inv_n |p_1'| and its reciprocal (0 at rest: the |v|^2 > 0 guard)  d n / d x_c = (p_1' . d p_1'/d x_c) inv_n
        sr, sl   max(0, b_x - x_s), max(0, -x_s - b_x);  on_r, on_l their 0/1 slopes"""
    bx, th, v, w, u = sp.symbols("b_x th_1 b_x_dot th_1_dot u", real=True)
    m_b, m_1, l_1, g, mu_b, v_mu_b, c_d_1, x_s, k_s = prm = sp.symbols("m_b m_1 l_1 g mu_b v_mu_b c_d_1 x_s k_s", real=True)
    fbx, fmx, fmy = sp.symbols("f_b_x f_m1_x f_m1_y", real=True)   # f_b_y does no work (b moves along x only)
    s, c, tv, n, inv_n, sr, sl, on_r, on_l, ivm = sp.symbols("s c tv n inv_n sr sl on_r on_l ivm", real=True)
    a0, a1 = sp.symbols("a0 a1", real=True)

    # kinematics with sin / cos as symbols: d s = c d th, d c = -s d th
    def d_th(e):
        return sp.diff(e, th) + sp.diff(e, s) * c - sp.diff(e, c) * s

    def ddt(e):  # total time derivative along (q, q', q'')
        return sp.diff(e, bx) * v + d_th(e) * w + sp.diff(e, v) * a0 + sp.diff(e, w) * a1

    b = sp.Matrix([bx, 0])
    p1 = b + sp.Matrix([c, s]) * l_1                                   # dynamics_single.py:58-64
    bd, p1d = b.applyfunc(ddt), p1.applyfunc(ddt)
    T = (m_b * bd.dot(bd) + m_1 * p1d.dot(p1d)) / 2                     # :66-83
    V = g * m_1 * p1[1]
    L = T - V
    Q_b = fbx * 1 + fmx * sp.diff(p1[0], bx) + fmy * sp.diff(p1[1], bx)   # :92-98
    Q_th = fmx * d_th(p1[0]) + fmy * d_th(p1[1])
    F_fric = -mu_b * (m_1 + m_b) * g * tv                                # :100-103
    # drag power (1/6) c_d |p_1'|^3 enters through d/dq' : (1/2) c_d |p_1'|^2 d|p_1'|/dq' = (1/2) c_d n (p_1' . dp_1'/dq')
    # factored_velocity (the M / F form of round 6): the components of p_1' stay the symbols vxs, vys in the drag terms (with
    # their derivative rules below) instead of being expanded into the state -- what a person would write
    vxs, vys, sech2 = sp.symbols("vx vy sech2", real=True)
    pv = (vxs, vys) if factored_velocity else (p1d[0], p1d[1])
    dtv = sech2 if factored_velocity else (1 - tv**2)   # d tanh / d arg: a helper of its own in the M / F form (no 1 - tanh^2 cancellation)
    drag_v = sp.Rational(1, 2) * c_d_1 * n * (pv[0] * sp.diff(p1d[0], v) + pv[1] * sp.diff(p1d[1], v))     # :105-111
    drag_w = sp.Rational(1, 2) * c_d_1 * n * (pv[0] * sp.diff(p1d[0], w) + pv[1] * sp.diff(p1d[1], w))
    F_s = -k_s * sr + k_s * sl                                           # :113-115
    el_b = ddt(sp.diff(L, v)) - sp.diff(L, bx) - u - Q_b - F_fric - F_s + drag_v      # :117-131
    el_th = ddt(sp.diff(L, w)) - d_th(L) - Q_th + drag_w
    el = [sp.expand(el_b), sp.expand(el_th)]
    # A(x, x') x'' = f(x, x', u)  (sympy_utils.get_euler_lagrange_coefficients), closed-form 2x2 inverse (get_mat_inverse)
    A = sp.Matrix(2, 2, lambda i, j: sp.diff(el[i], (a0, a1)[j]))
    f = sp.Matrix([-(e.subs({a0: 0, a1: 0})) for e in el])
    A = A.applyfunc(lambda e: sp.simplify(e.subs(c**2, 1 - s**2)))
    det = sp.simplify((A[0, 0] * A[1, 1] - A[0, 1] * A[1, 0]).subs(c**2, 1 - s**2))
    acc = sp.Matrix([(A[1, 1] * f[0] - A[0, 1] * f[1]) / det, (-A[1, 0] * f[0] + A[0, 0] * f[1]) / det])

    # total derivatives through the helper symbols
    vx, vy = p1d[0].subs({a0: 0, a1: 0}), p1d[1].subs({a0: 0, a1: 0})

    def dvel(var):   # d p_1' / d var
        return (d_th(vx) if var is th else sp.diff(vx, var)), (d_th(vy) if var is th else sp.diff(vy, var))

    def dn(var):   # d|p_1'| / d var,  var in (th, v, w)
        dvx, dvy = dvel(var)
        return ((pv[0] if factored_velocity else vx) * dvx + (pv[1] if factored_velocity else vy) * dvy) * inv_n

    def through_velocity(e, var):   # the chain rule through the symbols vxs, vys (zero when they are not in use)
        dvx, dvy = dvel(var)
        return sp.diff(e, vxs) * dvx + sp.diff(e, vys) * dvy

    def total(e, var):
        if var is bx:
            return sp.diff(e, bx) + sp.diff(e, sr) * on_r - sp.diff(e, sl) * on_l
        if var is th:
            return d_th(e) + sp.diff(e, n) * dn(th) + through_velocity(e, th)
        if var is v:
            return sp.diff(e, v) + sp.diff(e, tv) * dtv * ivm + sp.diff(e, n) * dn(v) + through_velocity(e, v)
        return sp.diff(e, w) + sp.diff(e, n) * dn(w) + through_velocity(e, w)

    Ja = sp.Matrix(2, 4, lambda r, k: total(acc[r], (bx, th, v, w)[k]))
    Jua = sp.Matrix([sp.diff(acc[0], u), sp.diff(acc[1], u)])
    return dict(prm=list(prm), state=[bx, th, v, w], u=u, ext=[fbx, fmx, fmy], acc=acc, Ja=Ja, Jua=Jua,
                helpers=dict(s=s, c=c, tv=tv, n=n, inv_n=inv_n, sr=sr, sl=sl, on_r=on_r, on_l=on_l, ivm=ivm),
                vx=vx, vy=vy, el=el, acc_syms=(a0, a1), total=total, d_th=d_th, vel_syms=(vxs, vys), sech2=sech2)


def cse_single(model):
    """One common-subexpression pass over everything the function returns (with the external forces in)."""
    outs = [("a[0]", model["acc"][0]), ("a[1]", model["acc"][1])]
    for r in range(2):
        for k in range(4):
            outs.append(("Ja[%d][%d]" % (r, k), sp.together(model["Ja"][r, k])))
    outs += [("Jua[0]", model["Jua"][0]), ("Jua[1]", model["Jua"][1])]
    repl, red = sp.cse([e for _, e in outs], symbols=sp.numbered_symbols("t"), optimizations="basic")
    return [n for n, _ in outs], repl, red


def specialise(model, repl, red, with_ext):
    """The temporaries and outputs with the external forces present or identically zero (forward-substituting what
    collapses to a number), split into: parameter-only constants / needed by the accelerations / needed by the
    Jacobians only."""
    zero = {} if with_ext else {e: 0 for e in model["ext"]}
    known, lane = {}, []
    for sym, e in repl:
        e2 = e.subs(zero).subs(known)
        if e2.is_Number or e2.is_Symbol:
            known[sym] = e2
        else:
            lane.append((sym, e2))
    red2 = [e.subs(zero).subs(known) for e in red]
    pset = set(model["prm"]) | {model["helpers"]["ivm"]}
    consts = []
    for sym, e in lane:
        if e.free_symbols <= (pset | set(consts)):
            consts.append(sym)
    # hoist the parameter-only factors that CSE left inside per-lane products (g*m_b*(mu_b*tv) -> k*tv, 1/l_1, ...)
    hoisted = model.setdefault("_hoisted", {})   # expression -> constant symbol, shared by both specialisations

    def is_const(e):
        return e.free_symbols <= (pset | set(consts))

    def hoist(e):
        if e.is_Atom:
            return e
        args = [hoist(a) for a in e.args]
        if e.is_Mul:
            cpart = [a for a in args if is_const(a)]
            rest = [a for a in args if not is_const(a)]
            nontrivial = [a for a in cpart if not a.is_Number]
            if rest and (len(nontrivial) >= 2 or any(not a.is_Atom for a in nontrivial)):
                ce = sp.Mul(*cpart)
                if ce not in hoisted:
                    hoisted[ce] = sp.Symbol("kc%d" % len(hoisted), real=True)
                return sp.Mul(hoisted[ce], *rest)
        if e.is_Pow and is_const(e.base) and not e.base.is_Number and e.exp.is_Number and e.exp < 0:
            if e not in hoisted:
                hoisted[e] = sp.Symbol("kc%d" % len(hoisted), real=True)
            return hoisted[e]
        return e.func(*args)

    lane = [(sym, e if sym in consts else hoist(e)) for sym, e in lane]
    red2 = [hoist(e) for e in red2]
    dep = {sym: e.free_symbols for sym, e in lane}

    def closure(exprs):
        need, stack = set(), [t for e in exprs for t in e.free_symbols]
        while stack:
            t = stack.pop()
            if t in dep and t not in need:
                need.add(t)
                stack.extend(dep[t])
        return need

    need_a = closure(red2[:2])
    need_j = closure(red2[2:]) - need_a
    return lane, red2, consts, need_a, need_j


def write_single(model):
    names, repl, red = cse_single(model)
    # the constants of the two specialisations are the same parameter-only temporaries: take the union, in CSE order
    spec = {w: specialise(model, repl, red, w) for w in (False, True)}
    const_syms = [sym for sym, _ in repl if any(sym in spec[w][2] for w in (False, True))]
    const_expr = dict(repl)
    for ce, sym in model.get("_hoisted", {}).items():   # the hoisted parameter-only factors come after the CSE constants
        const_syms.append(sym)
        const_expr[sym] = ce
    vx, vy = model["vx"], model["vy"]
    files = {}
    for lang in ("c",):
        scalar = "double" if lang == "c" else "R"
        pr = _Printer(scalar)
        z, one = ("%s(0)" % scalar, "%s(1)" % scalar) if lang == "hip" else ("0.0", "1.0")
        lines = ["// GENERATED by tools/gen_dynamics.py from the Lagrangian of symbolic/dynamics_single.py:58-143 -- do not edit.",
                 "// Accelerations a = (b_x'', th_1''), Ja = da/d(b_x, th_1, b_x', th_1') (2x4), Jua = da/du (2); the full",
                 "// state derivative is (b_x', th_1', a) and its Jacobian [[0 0 1 0],[0 0 0 1],[Ja]]",
                 "// (single_pendulum_dynamics.hpp:159-166).  p = {m_b, m_1, l_1, g, mu_b, v_mu_b, c_d_1, x_s, k_s};",
                 "// k[] = the parameter-only sub-expressions, evaluated once (on the host, in double) by ..._consts()."]
        nk = max(len(const_syms), 1)
        if lang == "hip":
            ctype, prc = "P", _Printer("P")
            lines += ["#pragma once", "namespace cpmpc {", "template <typename R>", "struct SinglePendulumGenConsts {",
                      "  R p[9], ivm, tanh_k2;", "  R k[%d];" % nk, "};", "template <typename R, typename P>",
                      "__host__ __device__ inline SinglePendulumGenConsts<R> single_pendulum_gen_consts(const P* q) {",
                      "  SinglePendulumGenConsts<R> K;"]
        else:
            ctype, prc = "double", _Printer("double")
            lines += ["typedef struct { double p[9], ivm; double k[%d]; } SinglePendulumGenConsts;" % nk,
                      "static SinglePendulumGenConsts single_pendulum_gen_consts(const double* q) {", "  SinglePendulumGenConsts K;"]
        for i, sym in enumerate(model["prm"]):
            lines.append("  const %s %s = q[%d];" % (ctype, sym, i))
        lines.append("  const {0} ivm = {0}(1) / (({0}(1.0e-6) < v_mu_b) ? v_mu_b : {0}(1.0e-6));".format(ctype) if lang == "hip"
                     else "  const double ivm = 1.0 / ((1.0e-6 < v_mu_b) ? v_mu_b : 1.0e-6);  /* max(v_mu_b, 1e-6) */")
        for sym in const_syms:
            lines.append("  const %s %s = %s;" % (ctype, sym, prc.doprint(const_expr[sym])))
        cast = "R" if lang == "hip" else ""
        lines.append("  for (int i = 0; i < 9; ++i) K.p[i] = %s(q[i]);" % cast)
        lines.append("  K.ivm = %s(ivm);" % cast)
        if lang == "hip":
            lines.append("  K.tanh_k2 = R(P(-2.8853900817779268) * ivm);  // fp32 tanh: exp2 argument scale")
        for i, sym in enumerate(const_syms):
            lines.append("  K.k[%d] = %s(%s);" % (i, cast, sym))
        lines += ["  (void)m_b; (void)m_1; (void)l_1; (void)g; (void)mu_b; (void)c_d_1; (void)x_s; (void)k_s;", "  return K;", "}"]
        for with_ext in (False, True):
            lane, red2, consts, need_a, need_j = spec[with_ext]
            tag = "ext" if with_ext else "noext"
            if lang == "hip":
                lines += ["template <typename R, bool WITH_J>",
                          "__device__ __forceinline__ void single_pendulum_gen_accel_%s(const SinglePendulumGenConsts<R>& K, const R b_x, "
                          "const R th_1, const R b_x_dot, const R th_1_dot, const R u, const R f_b_x, const R f_m1_x, "
                          "const R f_m1_y, R (&a)[2], R (&Ja)[2][4], R (&Jua)[2]) {" % tag]
            else:
                lines += ["static void single_pendulum_gen_accel_%s(const SinglePendulumGenConsts* Kp, double b_x, double th_1, "
                          "double b_x_dot, double th_1_dot, double u, double f_b_x, double f_m1_x, double f_m1_y, double a[2], "
                          "double Ja[2][4], double Jua[2], int WITH_J) {" % tag, "  const SinglePendulumGenConsts K = *Kp;"]
            for i, sym in enumerate(model["prm"]):
                lines.append("  const %s %s = K.p[%d];" % (scalar, sym, i))
            for i, sym in enumerate(const_syms):
                lines.append("  const %s %s = K.k[%d];" % (scalar, sym, i))
            lines.append("  const %s ivm = K.ivm;" % scalar)
            if lang == "hip":
                lines += ["  R s, c;", "  Math<R>::sincos(th_1, s, c);",
                          "  const R tv = Math<R>::tanh_scaled(b_x_dot, ivm, K.tanh_k2);"]
                boolt = "bool"
            else:
                lines += ["  const double s = sin(th_1), c = cos(th_1);", "  const double tv = tanh(b_x_dot * ivm);"]
                boolt = "int"
            lines += ["  const %s e_r = b_x - x_s, e_l = -x_s - b_x;  /* springs: strict comparisons, as the generated branches */" % scalar,
                      "  const %s is_r = %s < e_r, is_l = %s < e_l;" % (boolt, z, z),
                      "  const %s sr = is_r ? e_r : %s, sl = is_l ? e_l : %s;" % (scalar, z, z),
                      "  const %s on_r = is_r ? %s : %s, on_l = is_l ? %s : %s;" % (scalar, one, z, one, z),
                      "  const %s n2 = %s;  /* |p_1'|^2 */" % (scalar, pr.doprint(vx**2 + vy**2))]
            if lang == "hip":
                lines += ["  R n, inv_n;", "  Math<R>::sqrt_inv(n2, n, inv_n);  // inv_n = 0 at rest: the |v|^2 > 0 guard"]
            else:
                lines += ["  const double n = sqrt(n2);", "  const double inv_n = (0.0 < n2) ? 1.0 / n : 0.0;  /* the |v|^2 > 0 guard */"]
            for sym, e in lane:
                if sym in need_a and sym not in const_syms:
                    lines.append("  const %s %s = %s;" % (scalar, sym, pr.doprint(e)))
            lines.append("  a[0] = %s;" % pr.doprint(red2[0]))
            lines.append("  a[1] = %s;" % pr.doprint(red2[1]))
            lines.append("  if (WITH_J) {")
            for sym, e in lane:
                if sym in need_j and sym not in const_syms:
                    lines.append("    const %s %s = %s;" % (scalar, sym, pr.doprint(e)))
            for name, e in zip(names[2:], red2[2:]):
                lines.append("    %s = %s;" % (name, pr.doprint(e)))
            lines += ["  }", "  (void)th_1; (void)n; (void)inv_n; (void)on_r; (void)on_l; (void)f_b_x; (void)f_m1_x; (void)f_m1_y;",
                      "  (void)Ja; (void)Jua; (void)m_b; (void)m_1; (void)l_1; (void)g; (void)mu_b; (void)v_mu_b; (void)c_d_1; (void)k_s;", "}"]
        if lang == "hip":
            lines.append("}  // namespace cpmpc")
        files[lang] = "\n".join(lines) + "\n"
    files["hip"] = "\n".join([
        "// GENERATED by tools/gen_dynamics.py from the Lagrangian of symbolic/dynamics_single.py:58-143 -- do not edit.",
        "// M(q) q'' = F(q, q', u, external forces) of the cart + single pole with friction, drag, bumpers: M[4], F[2], dFdx[2x4]",
        "// (x = b_x, th_1, b_x', th_1'), dM1 = dM/dth_1, row-major; the accelerations and their partials come from the 2 x 2 solve",
        "// next to this code (models.hpp: SingleModelGenerated), as for the 6-state model.  (Rounds 2-5 emitted the reference",
        "// generator's own form here -- closed-form inverse, symbolic Jacobians, symbolic/sympy_utils.py:43-50; the C file the",
        "// generator writes for the CPU check still is that form.)",
        "#pragma once", "namespace cpmpc {", emit_single_mf(model).rstrip("\n"), "}  // namespace cpmpc"]) + "\n"
    write_if_changed(os.path.join(ROOT, "oracle", "single_pendulum_gen.inc"), files["c"])
    write_if_changed(os.path.join(ROOT, "cart-pole-mpc_amd", "csrc", "single_pendulum_gen.hpp"), files["hip"])
    return files


def emit_single_mf(model):
    """Round 6: the 4-state model in the SAME form as the 6-state one -- M(q) q'' = F with the solve left to hand-written code
    next to it (models.hpp: SingleModelGenerated) -- instead of the closed-form inverse and its symbolic derivatives:
    M (2 x 2), F (2), dF/dx (2 x 4) and dM/dth_1 (2 x 2) as polynomials in the per-lane atoms (s, c, the helper symbols of
    derive_single, velocities, control, external forces) with parameter-only coefficients folded into constants (host), sine
    and cosine as INPUTS (the RK4 stages rotate them, models.hpp: StepCache), identically-zero entries as masks.  The helper
    prologue (tanh, |p_1'| and its reciprocal, the bumper selects) is emitted as its own function."""
    model = derive_single(factored_velocity=True)
    bx, th, v, w = model["state"]
    u = model["u"]
    fbx, fmx, fmy = model["ext"]
    h = model["helpers"]
    vxs, vys = model["vel_syms"]
    sech2 = model["sech2"]
    s_, c_, tv, n, inv_n, sr, sl, on_r, on_l, ivm = (h[k] for k in ("s", "c", "tv", "n", "inv_n", "sr", "sl", "on_r", "on_l", "ivm"))
    prm = model["prm"]
    el, a_syms, total, d_th = model["el"], model["acc_syms"], model["total"], model["d_th"]
    M = sp.Matrix(2, 2, lambda i, j: sp.diff(el[i], a_syms[j]))
    F = sp.Matrix([-(e.subs({a_syms[0]: 0, a_syms[1]: 0})) for e in el])
    x = [bx, th, v, w]
    outs = [("M[%d]" % (i * 2 + j), M[i, j]) for i in range(2) for j in range(2)]
    outs += [("F[%d]" % i, F[i]) for i in range(2)]
    outs += [("dFdx[%d]" % (i * 4 + c), total(F[i], x[c])) for i in range(2) for c in range(4)]
    outs += [("dM1[%d]" % (i * 2 + j), d_th(M[i, j])) for i in range(2) for j in range(2)]
    outs += [("dFdu[%d]" % i, sp.diff(F[i], u)) for i in range(2)]
    lane_syms = [s_, c_, tv, sech2, n, inv_n, sr, sl, on_r, on_l, vxs, vys, v, w, u, fbx, fmx, fmy]
    pset = list(prm) + [ivm]

    def trig_reduce(e):   # c^2 -> 1 - s^2 (the polynomial expansion does not know the identity)
        e = sp.expand(e)
        e = e.replace(lambda t: t.is_Pow and t.base == c_ and t.exp.is_Integer and t.exp >= 2,
                      lambda t: (1 - s_**2) ** (int(t.exp) // 2) * c_ ** (int(t.exp) % 2))
        return sp.expand(e)
    files = {}
    text = ["// ---- round 6: M(q) q'' = F form with constants folded, sine / cosine and the helper terms as inputs, structure as masks ----"]
    kconst = {}

    def const_of(e):
        e = sp.factor(e)
        num, rest = e.as_coeff_Mul()
        if rest == 1:
            return sp.Integer(1), num
        key = sp.expand(rest)
        try:
            lead = sp.Poly(key, *pset).coeffs()[0]
        except sp.PolynomialError:
            lead = 1
        if lead.is_number and lead < 0:
            key, num = -key, -num
        if key not in kconst:
            kconst[key] = sp.Symbol("k%d" % len(kconst), real=True)
        return kconst[key], num

    specs = {}
    for with_ext in (False, True):
        zero = {} if with_ext else {fbx: 0, fmx: 0, fmy: 0}
        lane_exprs = []
        for _, e in outs:
            e = trig_reduce(e.subs(zero))
            if e == 0:
                lane_exprs.append(sp.Integer(0))
                continue
            poly = sp.Poly(e, *lane_syms)
            acc = 0
            for mono, coeff in poly.terms():
                ksym, num = const_of(coeff)
                term = num * ksym
                for sym, pw in zip(lane_syms, mono):
                    term = term * sym**pw
                acc = acc + term
            lane_exprs.append(acc)
        repl, red = sp.cse(lane_exprs, symbols=sp.numbered_symbols("t"), optimizations="basic")
        specs[with_ext] = (repl, red)
    names = [nm for nm, _ in outs]
    pr, prc = _Printer("R"), _Printer("P")
    const_syms = list(kconst.values())
    text += ["template <typename R>", "struct SinglePendulumMFConsts {", "  R k[%d];" % max(len(const_syms), 1),
             "  R ivm, tanh_k2, x_s, l_1;  // helper prologue: friction scale, bumper position, pole length (|p_1'|)", "};",
             "template <typename R, typename P>",
             "__host__ __device__ inline SinglePendulumMFConsts<R> single_pendulum_mf_consts(const P* q) {",
             "  SinglePendulumMFConsts<R> K;"]
    for i, sym in enumerate(prm):
        text.append("  const P %s = q[%d];" % (sym, i))
    text.append("  const P ivm = P(1) / ((P(1.0e-6) < v_mu_b) ? v_mu_b : P(1.0e-6));  // 1 / max(v_mu_b, 1e-6)")
    for i, (ce, sym) in enumerate(kconst.items()):
        text.append("  K.k[%d] = R(%s);" % (i, prc.doprint(ce)))
    text += ["  K.ivm = R(ivm);", "  K.tanh_k2 = R(P(-2.8853900817779268) * ivm);  // fp32 tanh: exp2 argument scale",
             "  K.x_s = R(x_s);", "  K.l_1 = R(l_1);",
             "  " + " ".join("(void)%s;" % p_ for p_ in prm), "  return K;", "}"]
    # structure (of the no-ext form; the external forces add no new non-zero entries to these masks' meaning for the solver:
    # the masks are taken over BOTH specialisations)
    nz = {nm: any(specs[w][1][i] != 0 for w in (False, True)) for i, nm in enumerate(names)}

    def mask(prefix, count):
        return "{" + ", ".join("true" if nz["%s[%d]" % (prefix, i)] else "false" for i in range(count)) + "}"

    ja_zero = []
    for c in range(4):
        z = all(not nz["dFdx[%d]" % (i * 4 + c)] for i in range(2))
        if c == 1:
            z = z and all(not nz["dM1[%d]" % i] for i in range(4))
        ja_zero.append(z)
    text += ["struct SinglePendulumMFSparsity {", "  static constexpr bool dFdx[8] = %s;" % mask("dFdx", 8),
             "  static constexpr bool dM1[4] = %s;" % mask("dM1", 4),
             "  static constexpr unsigned ja_zero_cols = 0x%xu;" % sum(1 << c for c in range(4) if ja_zero[c]), "};",
             "// the helper terms of the dynamics at a state: tv = tanh(b_x' / max(v_mu_b, 1e-6)) and its slope sech2 (where the scalar type",
             "// has the split tanh, Math<R>::tanh_parts, as 4 e^-2|x| / (1 + e^-2|x|)^2: full relative accuracy where 1 - tanh^2 cancels -- a",
             "// saturated friction term with a tiny v_mu_b multiplies that slope by up to 1e6), n = |p_1'| (inv_n its reciprocal, 0 at",
             "// rest: the |v|^2 > 0 guard; WITH_J = false skips it), the bumper springs' compressions and their 0 / 1 slopes",
             "template <typename R, bool WITH_J>",
             "__device__ __forceinline__ void single_pendulum_mf_helpers(const SinglePendulumMFConsts<R>& K, const R s, const R c, const R b_x, "
             "const R b_x_dot, const R th_1_dot, R& tv, R& sech2, R& n, R& inv_n, R& sr, R& sl, R& on_r, R& on_l, R& vx, R& vy) {",
             "  if constexpr (Math<R>::kMergedReciprocals) {",
             "    R num, den, e2;",
             "    Math<R>::tanh_parts(b_x_dot * K.ivm, num, den, e2);",
             "    const R iden = Math<R>::rcp(den);",
             "    tv = __builtin_copysign(num * iden, b_x_dot);",
             "    sech2 = R(4) * e2 * (iden * iden);",
             "  } else {",
             "    tv = Math<R>::tanh_scaled(b_x_dot, K.ivm, K.tanh_k2);",
             "    sech2 = R(1) - tv * tv;",
             "  }",
             "  const R e_r = b_x - K.x_s, e_l = -K.x_s - b_x;  // strict comparisons, as the reference's generated branches",
             "  const bool is_r = R(0) < e_r, is_l = R(0) < e_l;",
             "  sr = is_r ? e_r : R(0);", "  sl = is_l ? e_l : R(0);", "  on_r = is_r ? R(1) : R(0);", "  on_l = is_l ? R(1) : R(0);",
             "  vx = b_x_dot - K.l_1 * s * th_1_dot;", "  vy = K.l_1 * c * th_1_dot;",
             "  const R n2 = vx * vx + vy * vy;",
             "  if (WITH_J) Math<R>::sqrt_inv(n2, n, inv_n);", "  else {", "    n = Math<R>::sqrt_only(n2);", "    inv_n = R(0);", "  }", "}"]
    for with_ext in (False, True):
        repl, red = specs[with_ext]
        used = set()
        for _, e in repl:
            used |= e.free_symbols
        for e in red:
            used |= e.free_symbols
        tag = "ext" if with_ext else "noext"
        text += ["template <typename R, bool WITH_J>",
                 "__device__ __forceinline__ void single_pendulum_mf_terms_%s(const SinglePendulumMFConsts<R>& K, const R s, const R c, "
                 "const R tv, const R sech2, const R n, const R inv_n, const R sr, const R sl, const R on_r, const R on_l, const R vx, const R vy, const R b_x_dot, "
                 "const R th_1_dot, const R u, const R f_b_x, const R f_m1_x, const R f_m1_y, R* M, R* F, R* dFdx, R* dM1) {" % tag]
        for i, sym in enumerate(const_syms):
            if sym in used:
                text.append("  const R %s = K.k[%d];" % (sym, i))
        # temporaries needed by M and F first, the rest only with Jacobians
        dep = {sym: e.free_symbols for sym, e in repl}

        def closure(exprs):
            need, stack = set(), [t for e in exprs for t in e.free_symbols]
            while stack:
                t = stack.pop()
                if t in dep and t not in need:
                    need.add(t)
                    stack.extend(dep[t])
            return need

        n_val = 6   # M[4], F[2]
        need_a = closure(red[:n_val])
        for sym, e in repl:
            if sym in need_a:
                text.append("  const R %s = %s;" % (sym, pr.doprint(e)))
        for nm, e in zip(names[:n_val], red[:n_val]):
            text.append("  %s = %s;" % (nm, pr.doprint(e)))
        text.append("  if (WITH_J) {")
        for sym, e in repl:
            if sym not in need_a:
                text.append("    const R %s = %s;" % (sym, pr.doprint(e)))
        for nm, e in zip(names[n_val:], red[n_val:]):
            if nm.startswith("dFdu"):
                continue
            if e != 0:
                text.append("    %s = %s;" % (nm, pr.doprint(e)))
        text += ["  }", "  (void)s; (void)c; (void)tv; (void)sech2; (void)n; (void)inv_n; (void)sr; (void)sl; (void)on_r; (void)on_l; (void)vx; (void)vy; (void)b_x_dot; "
                 "(void)th_1_dot; (void)u; (void)f_b_x; (void)f_m1_x; (void)f_m1_y; (void)dFdx; (void)dM1;", "}"]
        dfdu = [red[names.index("dFdu[%d]" % i)] for i in range(2)]
        assert dfdu[0] == 1 and dfdu[1] == 0, dfdu   # the control acts on the base only: da/du = M^-1 e_0 (the solver assumes it)
    return "\n".join(text) + "\n"


def main():
    model = derive_double()
    banner = ["// GENERATED by tools/gen_dynamics.py from the Lagrangian of symbolic/dynamics_double.py:25-107 -- do not edit.",
              "// M(q) q'' = F(q, q', u): M[9] row-major, F[3], dFdx[3x6] row-major (x = [q, q']), dM1 = dM/dth_1, dM2 = dM/dth_2.",
              "// p = {m_b, m_1, m_2, l_1, l_2, g}; x = {b_x, th_1, th_2, b_x', th_1', th_2'}."]
    c_code = emit(model, "double", banner)
    write_if_changed(os.path.join(ROOT, "oracle", "double_pendulum_gen.inc"), c_code)
    hip = (emit(model, "R", banner + ["#pragma once", "namespace cpmpc {"])
           + emit_hip_sc(model, ["// ---- round 6: the same terms with constants folded, trigonometry as inputs and the structure as masks ----"])
           + "}  // namespace cpmpc\n")
    write_if_changed(os.path.join(ROOT, "cart-pole-mpc_amd", "csrc", "double_pendulum_gen.hpp"), hip)
    single = derive_single()
    files = write_single(single)
    print("single pendulum: a0 =", sp.simplify(single["acc"][0]))
    print("wrote oracle/single_pendulum_gen.inc (%d lines), csrc/single_pendulum_gen.hpp (%d lines)"
          % (files["c"].count("\n"), files["hip"].count("\n")))
    print("M =", model["M"])
    print("F =", model["F"])
    print("wrote oracle/double_pendulum_gen.inc (%d lines), csrc/double_pendulum_gen.hpp" % c_code.count("\n"))


if __name__ == "__main__":
    main()
