"""ctypes loader for the CPU oracle (oracle/libcpmpc_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never from the product package.  See oracle/cpmpc_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libcpmpc_oracle.so")
_LIB_LD_PATH = os.path.join(_HERE, "libcpmpc_oracle_ld.so")
_LIB_F32_PATH = os.path.join(_HERE, "libcpmpc_oracle_f32.so")


def build(force=False):
    """Compile the oracle with gcc (seconds)."""
    srcs = [os.path.join(_HERE, f) for f in ("cpmpc_oracle.c", "cpmpc_oracle.h", "single_pendulum_gen.inc",
                                             "double_pendulum_gen.inc")]
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(f) for f in srcs)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "libcpmpc_oracle.so"],
                          stdout=subprocess.DEVNULL)
    return _LIB_PATH


def build_ld(force=False):
    """Compile the extended-precision build of the same restatement (cpmpc_oracle_ld.c: long double arithmetic)."""
    srcs = [os.path.join(_HERE, f) for f in ("cpmpc_oracle_ld.c", "cpmpc_oracle.c", "cpmpc_oracle.h")]
    if (not force and os.path.exists(_LIB_LD_PATH)
            and os.path.getmtime(_LIB_LD_PATH) >= max(os.path.getmtime(f) for f in srcs)):
        return _LIB_LD_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "libcpmpc_oracle_ld.so"], stdout=subprocess.DEVNULL,
                          stderr=subprocess.DEVNULL)
    return _LIB_LD_PATH


def build_f32(force=False):
    """Compile the single-precision twin of the same restatement (cpmpc_oracle_f32.c: float arithmetic, KKT solve in double)."""
    srcs = [os.path.join(_HERE, f) for f in ("cpmpc_oracle_f32.c", "cpmpc_oracle.c", "cpmpc_oracle.h")]
    if (not force and os.path.exists(_LIB_F32_PATH)
            and os.path.getmtime(_LIB_F32_PATH) >= max(os.path.getmtime(f) for f in srcs)):
        return _LIB_F32_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "libcpmpc_oracle_f32.so"], stdout=subprocess.DEVNULL,
                          stderr=subprocess.DEVNULL)
    return _LIB_F32_PATH


class OptParams(C.Structure):
    """Mirror of orc_opt_params == pendulum::OptimizationParams (optimization.hpp:12-53)."""
    _fields_ = [
        ("control_dt", C.c_double),
        ("window_length", C.c_uint64),
        ("state_spacing", C.c_uint64),
        ("max_iterations", C.c_uint64),
        ("relative_exit_tol", C.c_double),
        ("absolute_first_derivative_tol", C.c_double),
        ("equality_penalty_initial", C.c_double),
        ("u_guess_sinusoid_amplitude", C.c_double),
        ("u_cost_weight", C.c_double),
        ("u_derivative_cost_weight", C.c_double),
        ("b_x_final_cost_weight", C.c_double),
        ("th_final_cost_weight", C.c_double),
        ("b_x_dot_final_cost_weight", C.c_double),
        ("th_dot_final_cost_weight", C.c_double),
    ]

    def num_states(self):
        return self.window_length // self.state_spacing + 1

    def dim(self):
        return 4 * self.num_states() + self.window_length


class SolverOpts(C.Structure):
    _fields_ = [
        ("max_line_search_iterations", C.c_int32),
        ("armijo_c1", C.c_double),
        ("ls_shrink_max", C.c_double),
        ("ls_shrink_min", C.c_double),
        ("ls_alpha_growth", C.c_double),
        ("penalty_rho", C.c_double),
        ("lambda_initial", C.c_double),
        ("lambda_failure_init", C.c_double),
        ("lambda_scale_up", C.c_double),
        ("lambda_scale_down", C.c_double),
        ("lambda_min", C.c_double),
        ("lambda_max", C.c_double),
        ("b_x_limit", C.c_double),
        ("u_limit", C.c_double),
        ("ls_alpha_growth_backtracked", C.c_double),
        ("full_step_below", C.c_double),
        ("exit_defect_floor", C.c_double),
    ]


class SolverSummary(C.Structure):
    _fields_ = [
        ("termination_state", C.c_int32),
        ("iterations", C.c_int32),
        ("line_search_evals", C.c_int32),
        ("failed_steps", C.c_int32),
        ("initial_cost", C.c_double),
        ("initial_eq_l1", C.c_double),
        ("final_cost", C.c_double),
        ("final_eq_l1", C.c_double),
        ("final_penalty", C.c_double),
        ("final_lambda", C.c_double),
    ]


TERM_NAMES = {
    0: "NONE", 1: "MAX_ITERATIONS", 2: "SATISFIED_ABSOLUTE_TOL", 3: "SATISFIED_RELATIVE_TOL",
    4: "SATISFIED_FIRST_ORDER_TOL", 5: "QP_INDEFINITE", 6: "USER_CALLBACK", 7: "MAX_LAMBDA",
    8: "NON_FINITE",
}
TERM = {v: k for k, v in TERM_NAMES.items()}
for _k, _v in TERM.items():
    globals()["TERM_" + _k] = _v

_dp = C.POINTER(C.c_double)
_lib = None


def _ptr(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _vec(a, n=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if n is not None:
        assert a.size == n, (a.size, n)
    return a


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    L.orc_default_opt_params.argtypes = [C.POINTER(OptParams)]
    L.orc_default_solver_opts.argtypes = [C.POINTER(SolverOpts)]
    L.orc_dynamics.argtypes = [_dp, _dp, C.c_double, _dp, _dp, _dp, _dp, _dp]
    L.orc_dynamics_generated.argtypes = [_dp, _dp, C.c_double, _dp, _dp, _dp, _dp, _dp]
    L.orc_rk4.argtypes = [_dp, _dp, C.c_double, C.c_double, _dp, _dp, _dp, _dp, _dp]
    L.orc_rk4_no_jacobians.argtypes = [_dp, _dp, C.c_double, C.c_double, _dp, _dp, _dp]
    L.orc_mod_pi.argtypes = [C.c_double]
    L.orc_mod_pi.restype = C.c_double
    L.orc_shooting_constraint.argtypes = [_dp, C.c_int, C.c_double, _dp, _dp, _dp]
    L.orc_problem_shape.argtypes = [C.POINTER(OptParams), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                    C.POINTER(C.c_int)]
    L.orc_problem_eval.argtypes = [C.POINTER(OptParams), _dp, _dp, C.c_double, C.c_double, _dp, _dp,
                                   _dp, _dp, _dp]
    L.orc_retract.argtypes = [C.POINTER(OptParams), C.POINTER(SolverOpts), _dp, _dp, C.c_double, _dp]
    L.orc_qp_solve.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp, C.c_double,
                               _dp]
    L.orc_qp_solve.restype = C.c_int
    L.orc_opt_create.argtypes = [C.POINTER(OptParams), C.POINTER(SolverOpts)]
    L.orc_opt_create.restype = C.c_void_p
    L.orc_opt_destroy.argtypes = [C.c_void_p]
    L.orc_opt_reset.argtypes = [C.c_void_p]
    L.orc_opt_set_previous_solution.argtypes = [C.c_void_p, _dp, C.c_int]
    L.orc_opt_has_previous_solution.argtypes = [C.c_void_p]
    L.orc_opt_has_previous_solution.restype = C.c_int
    L.orc_opt_step.argtypes = [C.c_void_p, _dp, _dp, C.c_double, _dp, _dp, _dp, _dp,
                               C.POINTER(SolverSummary)]
    L.orc_opt_step.restype = C.c_int
    L.orc_solve.argtypes = [C.POINTER(OptParams), C.POINTER(SolverOpts), _dp, _dp, C.c_double,
                            C.c_double, _dp, _dp, C.POINTER(SolverSummary)]
    L.orc_solve.restype = C.c_int
    L.orc_sim_step.argtypes = [_dp, C.c_double, C.c_double, _dp, _dp, _dp]
    L.orc_step_batch_cold.argtypes = [C.POINTER(OptParams), C.POINTER(SolverOpts), _dp, C.c_double,
                                      C.c_int64, _dp, _dp, _dp, C.POINTER(C.c_int32),
                                      C.POINTER(C.c_int32), C.c_int]
    L.orc_step_batch_cold.restype = C.c_int
    L.orc_model_nx.argtypes = [C.c_int]
    L.orc_model_np.argtypes = [C.c_int]
    L.orc_dynamics_double.argtypes = [_dp, _dp, C.c_double, _dp, _dp, _dp, _dp, _dp]
    L.orc_energy_double.argtypes = [_dp, _dp]
    L.orc_energy_double.restype = C.c_double
    L.orc_rk4_model.argtypes = [C.c_int, _dp, _dp, C.c_double, C.c_double, _dp, _dp, _dp]
    L.orc_shooting_constraint_model.argtypes = [C.c_int, _dp, C.c_int, C.c_double, _dp, _dp, _dp]
    L.orc_problem_shape_model.argtypes = [C.c_int, C.POINTER(OptParams), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                          C.POINTER(C.c_int)]
    L.orc_opt_create_model.argtypes = [C.POINTER(OptParams), C.POINTER(SolverOpts), C.c_int]
    L.orc_opt_create_model.restype = C.c_void_p
    L.orc_opt_dim.argtypes = [C.c_void_p]
    L.orc_sim_step_model.argtypes = [C.c_int, _dp, C.c_double, C.c_double, _dp]
    L.orc_step_batch_cold_model.argtypes = [C.c_int, C.POINTER(OptParams), C.POINTER(SolverOpts), _dp, C.c_double,
                                            C.c_int64, _dp, _dp, _dp, C.POINTER(C.c_int32),
                                            C.POINTER(C.c_int32), C.c_int]
    L.orc_step_batch_cold_model.restype = C.c_int
    _lib = L
    return L


_lib_ld = None


def lib_ld():
    global _lib_ld
    if _lib_ld is None:
        build_ld()
        L = C.CDLL(_LIB_LD_PATH)
        ip = C.POINTER(C.c_int32)
        L.orcld_step_batch_cold_d.argtypes = [C.c_int, C.POINTER(OptParams), C.POINTER(SolverOpts), _dp, C.c_double,
                                              C.c_int64, _dp, _dp, ip, ip, ip, _dp, C.c_int]
        L.orcld_step_batch_cold_d.restype = C.c_int
        _lib_ld = L
    return _lib_ld


_lib_f32 = None


def lib_f32():
    global _lib_f32
    if _lib_f32 is None:
        build_f32()
        L = C.CDLL(_LIB_F32_PATH)
        ip = C.POINTER(C.c_int32)
        bp = C.POINTER(C.c_int8)
        L.orcf_step_batch_cold_d.argtypes = [C.c_int, C.POINTER(OptParams), C.POINTER(SolverOpts), _dp, C.c_double,
                                             C.c_int64, _dp, _dp, ip, ip, ip, _dp, C.c_int]
        L.orcf_step_batch_cold_d.restype = C.c_int
        L.orcf_closed_loop_d.argtypes = [C.c_int, C.POINTER(OptParams), C.POINTER(SolverOpts), _dp, C.c_double, C.c_int64,
                                         _dp, C.c_int, bp, bp, _dp, C.c_int]
        L.orcf_closed_loop_d.restype = C.c_int
        _lib_f32 = L
    return _lib_f32


def default_opt_params(**overrides):
    p = OptParams()
    lib().orc_default_opt_params(C.byref(p))
    for k, v in overrides.items():
        assert hasattr(p, k), k
        setattr(p, k, v)
    return p


def default_solver_opts(**overrides):
    o = SolverOpts()
    lib().orc_default_solver_opts(C.byref(o))
    for k, v in overrides.items():
        assert hasattr(o, k), k
        setattr(o, k, v)
    return o


_Z2 = np.zeros(2)


def dynamics(params, x, u, f_base=None, f_mass=None, jacobians=True):
    params, x = _vec(params, 9), _vec(x, 4)
    fb = _vec(_Z2 if f_base is None else f_base, 2)
    fm = _vec(_Z2 if f_mass is None else f_mass, 2)
    f = np.zeros(4)
    if jacobians:
        Jx, Ju = np.zeros((4, 4)), np.zeros(4)
        lib().orc_dynamics(_ptr(params), _ptr(x), float(u), _ptr(fb), _ptr(fm), _ptr(f), _ptr(Jx),
                           _ptr(Ju))
        return f, Jx, Ju
    lib().orc_dynamics(_ptr(params), _ptr(x), float(u), _ptr(fb), _ptr(fm), _ptr(f), None, None)
    return f


def dynamics_generated(params, x, u, f_base=None, f_mass=None, jacobians=True):
    """orc_dynamics_generated: the code emitted by tools/gen_dynamics.py (forces None -> the specialisation generated
    without external forces)."""
    params, x = _vec(params, 9), _vec(x, 4)
    fb = None if f_base is None and f_mass is None else _vec(_Z2 if f_base is None else f_base, 2)
    fm = None if f_base is None and f_mass is None else _vec(_Z2 if f_mass is None else f_mass, 2)
    f = np.zeros(4)
    Jx, Ju = (np.zeros((4, 4)), np.zeros(4)) if jacobians else (None, None)
    lib().orc_dynamics_generated(_ptr(params), _ptr(x), float(u), _ptr(fb), _ptr(fm), _ptr(f), _ptr(Jx), _ptr(Ju))
    return (f, Jx, Ju) if jacobians else f


def rk4(params, x, u, h, f_base=None, f_mass=None):
    params, x = _vec(params, 9), _vec(x, 4)
    fb = _vec(_Z2 if f_base is None else f_base, 2)
    fm = _vec(_Z2 if f_mass is None else f_mass, 2)
    xn, A, B = np.zeros(4), np.zeros((4, 4)), np.zeros(4)
    lib().orc_rk4(_ptr(params), _ptr(x), float(u), float(h), _ptr(fb), _ptr(fm), _ptr(xn), _ptr(A),
                  _ptr(B))
    return xn, A, B


def rk4_no_jacobians(params, x, u, h, f_base=None, f_mass=None):
    params, x = _vec(params, 9), _vec(x, 4)
    fb = _vec(_Z2 if f_base is None else f_base, 2)
    fm = _vec(_Z2 if f_mass is None else f_mass, 2)
    xn = np.zeros(4)
    lib().orc_rk4_no_jacobians(_ptr(params), _ptr(x), float(u), float(h), _ptr(fb), _ptr(fm),
                               _ptr(xn))
    return xn


def mod_pi(a):
    return lib().orc_mod_pi(float(a))


def shooting_constraint(params, spacing, dt, vars_, jacobian=True):
    params = _vec(params, 9)
    vars_ = _vec(vars_, 8 + spacing)
    err = np.zeros(4)
    J = np.zeros((4, 8 + spacing)) if jacobian else None
    lib().orc_shooting_constraint(_ptr(params), int(spacing), float(dt), _ptr(vars_), _ptr(err),
                                  _ptr(J))
    return (err, J) if jacobian else err


def problem_shape(p):
    d, e, c = C.c_int(), C.c_int(), C.c_int()
    lib().orc_problem_shape(C.byref(p), C.byref(d), C.byref(e), C.byref(c))
    return d.value, e.value, c.value


def problem_eval(p, dyn, x_current, set_point, u_prev, z, jacobians=True):
    dim, n_eq, n_cost = problem_shape(p)
    dyn, x_current, z = _vec(dyn, 9), _vec(x_current, 4), _vec(z, dim)
    r, c = np.zeros(max(n_cost, 1)), np.zeros(n_eq)
    J = np.zeros((max(n_cost, 1), dim)) if jacobians else None
    A = np.zeros((n_eq, dim)) if jacobians else None
    lib().orc_problem_eval(C.byref(p), _ptr(dyn), _ptr(x_current), float(set_point), float(u_prev),
                           _ptr(z), _ptr(r), _ptr(c), _ptr(J), _ptr(A))
    r = r[:n_cost]
    if jacobians:
        return r, c, J[:n_cost], A
    return r, c


def retract(p, o, z, dz, alpha):
    dim = p.dim()
    z, dz = _vec(z, dim), _vec(dz, dim)
    out = np.zeros(dim)
    lib().orc_retract(C.byref(p), C.byref(o), _ptr(z), _ptr(dz), float(alpha), _ptr(out))
    return out


def qp_solve(J, r, A, c, n_u, lam=0.0):
    J, r, A, c = (np.ascontiguousarray(a, dtype=np.float64) for a in (J, r, A, c))
    dim = A.shape[1]
    dz = np.zeros(dim)
    Jp = J if J.size else np.zeros((1, dim))
    rp = r if r.size else np.zeros(1)
    rc = lib().orc_qp_solve(dim, A.shape[0], J.shape[0], int(n_u), _ptr(Jp), _ptr(rp), _ptr(A),
                            _ptr(c), float(lam), _ptr(dz))
    return rc, dz


def solve(p, dyn, x_current, set_point, u_prev, guess, opts=None):
    dim = p.dim()
    dyn, x_current, guess = _vec(dyn, 9), _vec(x_current, 4), _vec(guess, dim)
    z = np.zeros(dim)
    s = SolverSummary()
    lib().orc_solve(C.byref(p), C.byref(opts) if opts is not None else None, _ptr(dyn),
                    _ptr(x_current), float(set_point), float(u_prev), _ptr(guess), _ptr(z),
                    C.byref(s))
    return z, s


MODEL_SINGLE, MODEL_DOUBLE = 0, 1
MODELS = {"single": MODEL_SINGLE, "double": MODEL_DOUBLE, 0: 0, 1: 1}


def model_nx(model):
    return lib().orc_model_nx(MODELS[model])


def model_np(model):
    return lib().orc_model_np(MODELS[model])


def dynamics_double(params, x, u, jacobians=True):
    params, x = _vec(params, 6), _vec(x, 6)
    z2 = np.zeros(2)
    f = np.zeros(6)
    Jx = np.zeros((6, 6)) if jacobians else None
    Ju = np.zeros(6) if jacobians else None
    lib().orc_dynamics_double(_ptr(params), _ptr(x), float(u), _ptr(z2), _ptr(z2), _ptr(f), _ptr(Jx), _ptr(Ju))
    return (f, Jx, Ju) if jacobians else f


def energy_double(params, x):
    params, x = _vec(params, 6), _vec(x, 6)
    return lib().orc_energy_double(_ptr(params), _ptr(x))


def rk4_model(model, params, x, u, h, jacobians=True):
    m = MODELS[model]
    nx = model_nx(m)
    params, x = _vec(params, model_np(m)), _vec(x, nx)
    xn = np.zeros(nx)
    A = np.zeros((nx, nx)) if jacobians else None
    B = np.zeros(nx) if jacobians else None
    lib().orc_rk4_model(m, _ptr(params), _ptr(x), float(u), float(h), _ptr(xn), _ptr(A), _ptr(B))
    return (xn, A, B) if jacobians else xn


def shooting_constraint_model(model, params, spacing, dt, vars_, jacobian=True):
    m = MODELS[model]
    nx = model_nx(m)
    params = _vec(params, model_np(m))
    vars_ = _vec(vars_, 2 * nx + spacing)
    err = np.zeros(nx)
    J = np.zeros((nx, 2 * nx + spacing)) if jacobian else None
    lib().orc_shooting_constraint_model(m, _ptr(params), int(spacing), float(dt), _ptr(vars_), _ptr(err), _ptr(J))
    return (err, J) if jacobian else err


def problem_shape_model(model, p):
    d, e, c = C.c_int(), C.c_int(), C.c_int()
    lib().orc_problem_shape_model(MODELS[model], C.byref(p), C.byref(d), C.byref(e), C.byref(c))
    return d.value, e.value, c.value


def sim_step_model(model, params, dt, u, state):
    m = MODELS[model]
    params, state = _vec(params, model_np(m)), _vec(state, model_nx(m)).copy()
    lib().orc_sim_step_model(m, _ptr(params), float(dt), float(u), _ptr(state))
    return state


class StepOutputs:
    """Shape of pendulum::OptimizationOutputs (optimization.hpp:55-70)."""

    def __init__(self, initial_state, previous_solution, summary, u, predicted_states, guess, z):
        self.initial_state = initial_state
        self.previous_solution = previous_solution
        self.solver_outputs = summary
        self.u = u
        self.predicted_states = predicted_states  # [N, 4]
        self.guess = guess
        self.z = z


class Optimization:
    """Oracle counterpart of pendulum::Optimization (optimization.hpp:73-108)."""

    def __init__(self, params, opts=None, model="single"):
        self.params = params
        self._opts = opts
        self.model = MODELS[model]
        self.nx = model_nx(self.model)
        self._h = lib().orc_opt_create_model(C.byref(params), C.byref(opts) if opts is not None else None,
                                             self.model)
        if not self._h:
            raise ValueError("invalid OptimizationParams (optimization.cc:13-22 preconditions)")
        self._prev = None

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            try:
                _lib.orc_opt_destroy(self._h)
            except Exception:
                pass
            self._h = None

    def reset(self):
        lib().orc_opt_reset(self._h)
        self._prev = None

    def set_previous_solution(self, z):
        z = _vec(z)
        lib().orc_opt_set_previous_solution(self._h, _ptr(z), int(z.size))
        self._prev = z.copy()

    def step(self, state, dyn, set_point):
        p = self.params
        N, dim = int(p.window_length), lib().orc_opt_dim(self._h)
        state, dyn = _vec(state, self.nx), _vec(dyn, model_np(self.model))
        u, pred = np.zeros(N), np.zeros((N, self.nx))
        guess, z = np.zeros(dim), np.zeros(dim)
        s = SolverSummary()
        prev = self._prev.copy() if self._prev is not None else np.zeros(0)
        lib().orc_opt_step(self._h, _ptr(state), _ptr(dyn), float(set_point), _ptr(u), _ptr(pred),
                           _ptr(guess), _ptr(z), C.byref(s))
        self._prev = z.copy()
        return StepOutputs(state.copy(), prev, s, u, pred, guess, z)


class Simulator:
    """Oracle counterpart of pendulum::Simulator (simulator.hpp:10-29)."""

    def __init__(self):
        self.state = np.array([0.0, -np.pi / 2, 0.0, 0.0])  # simulator.hpp:28

    def set_state(self, s):
        self.state = _vec(s, 4).copy()

    def get_state(self):
        return self.state.copy()

    def step(self, params, dt, u, f_base=(0.0, 0.0), f_mass=(0.0, 0.0)):
        if not dt >= 0.0:
            raise ValueError("dt must be >= 0 (simulator.cc:13)")
        if not np.isfinite(u):
            raise ValueError("u must be finite (simulator.cc:14)")
        params, fb, fm = _vec(params, 9), _vec(f_base, 2), _vec(f_mass, 2)
        lib().orc_sim_step(_ptr(params), float(dt), float(u), _ptr(fb), _ptr(fm),
                           _ptr(self.state))


def step_batch_cold(p, dyn, set_point, x0_soa, opts=None, want_pred=False, num_threads=0, model="single"):
    """x0_soa: [nx, B].  Returns (u [N,B], pred [N,nx,B] or None, status [B], iters [B], threads)."""
    m = MODELS[model]
    nx = model_nx(m)
    x0 = np.ascontiguousarray(x0_soa, dtype=np.float64)
    assert x0.ndim == 2 and x0.shape[0] == nx
    B, N = x0.shape[1], int(p.window_length)
    dyn = _vec(dyn, model_np(m))
    u = np.zeros((N, B))
    pred = np.zeros((N, nx, B)) if want_pred else None
    status = np.zeros(B, dtype=np.int32)
    iters = np.zeros(B, dtype=np.int32)
    used = lib().orc_step_batch_cold_model(
        m, C.byref(p), C.byref(opts) if opts is not None else None, _ptr(dyn), float(set_point), B,
        _ptr(x0), _ptr(u), _ptr(pred), status.ctypes.data_as(C.POINTER(C.c_int32)),
        iters.ctypes.data_as(C.POINTER(C.c_int32)), int(num_threads))
    return u, pred, status, iters, used


def step_batch_cold_ld(p, dyn, set_point, x0_soa, opts=None, num_threads=0, model="single"):
    """The same cold-start re-plan in EXTENDED precision (x87 long double, oracle/cpmpc_oracle_ld.c): the arbiter for
    lanes on which the GPU and the double oracle end up apart.  Returns (u [N,B] rounded to double, status, iters,
    line-search evaluations, final |c|_1)."""
    m = MODELS[model]
    nx = model_nx(m)
    x0 = np.ascontiguousarray(x0_soa, dtype=np.float64)
    assert x0.ndim == 2 and x0.shape[0] == nx
    B, N = x0.shape[1], int(p.window_length)
    dyn = _vec(dyn, model_np(m))
    u = np.zeros((N, B))
    status, iters, evals = (np.zeros(B, dtype=np.int32) for _ in range(3))
    eq = np.zeros(B)
    ip = C.POINTER(C.c_int32)
    lib_ld().orcld_step_batch_cold_d(m, C.byref(p), C.byref(opts) if opts is not None else None, _ptr(dyn),
                                     float(set_point), B, _ptr(x0), _ptr(u), status.ctypes.data_as(ip),
                                     iters.ctypes.data_as(ip), evals.ctypes.data_as(ip), _ptr(eq), int(num_threads))
    return u, status, iters, evals, eq


def step_batch_cold_f32(p, dyn, set_point, x0_soa, opts=None, num_threads=0, model="single"):
    """The same cold-start re-plan in SINGLE precision (oracle/cpmpc_oracle_f32.c: float arithmetic, the KKT solve in
    double as the float kernels keep their terminal system): the checker at the CPMPC_F32 kernels' precision.  Returns
    (u [N,B] as double, status, iters, line-search evaluations, final |c|_1)."""
    m = MODELS[model]
    nx = model_nx(m)
    x0 = np.ascontiguousarray(x0_soa, dtype=np.float64)
    assert x0.ndim == 2 and x0.shape[0] == nx
    B, N = x0.shape[1], int(p.window_length)
    dyn = _vec(dyn, model_np(m))
    u = np.zeros((N, B))
    status, iters, evals = (np.zeros(B, dtype=np.int32) for _ in range(3))
    eq = np.zeros(B)
    ip = C.POINTER(C.c_int32)
    lib_f32().orcf_step_batch_cold_d(m, C.byref(p), C.byref(opts) if opts is not None else None, _ptr(dyn),
                                     float(set_point), B, _ptr(x0), _ptr(u), status.ctypes.data_as(ip),
                                     iters.ctypes.data_as(ip), evals.ctypes.data_as(ip), _ptr(eq), int(num_threads))
    return u, status, iters, evals, eq


def closed_loop_f32(p, dyn, set_point, x0_soa, ticks, opts=None, num_threads=0, model="single"):
    """B controllers in closed loop in SINGLE precision (warm-started Optimization::Step -> u_0 -> Simulator::Step per tick,
    optimization_test.cc:39-61 per controller).  Returns (status [ticks,B] int8, iterations [ticks,B] int8, final states
    [nx,B], threads used)."""
    m = MODELS[model]
    nx = model_nx(m)
    x0 = np.ascontiguousarray(x0_soa, dtype=np.float64)
    assert x0.ndim == 2 and x0.shape[0] == nx
    B = x0.shape[1]
    dyn = _vec(dyn, model_np(m))
    status = np.zeros((ticks, B), dtype=np.int8)
    iters = np.zeros((ticks, B), dtype=np.int8)
    xf = np.zeros((nx, B))
    bp = C.POINTER(C.c_int8)
    used = lib_f32().orcf_closed_loop_d(m, C.byref(p), C.byref(opts) if opts is not None else None, _ptr(dyn),
                                        float(set_point), B, _ptr(x0), int(ticks), status.ctypes.data_as(bp),
                                        iters.ctypes.data_as(bp), _ptr(xf), int(num_threads))
    return status, iters, xf, used
