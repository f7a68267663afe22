"""ctypes binding of include/cpmpc.h (libcpmpc.so): the C-ABI of the batched cart-pole MPC path.

This is plumbing only: structures, prototypes, error translation.  It never computes anything and
has no fallback: if libcpmpc.so is missing or no gfx950 device is usable, calls raise.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CPMPC_LIB selects another build of the SAME library (kernel experiments); never a fallback
LIB_PATH = os.environ.get("CPMPC_LIB") or os.path.join(_HERE, "lib", "libcpmpc.so")

OK, ERR_INVALID_ARG, ERR_UNSUPPORTED, ERR_NO_DEVICE, ERR_HIP, ERR_ALLOC, ERR_BATCH = range(7)
F32, F64 = 0, 1
MODEL_SINGLE, MODEL_DOUBLE = 0, 1
MODELS = {"single": MODEL_SINGLE, "double": MODEL_DOUBLE, 0: 0, 1: 1}

TERM_NAMES = {
    0: "NONE", 1: "MAX_ITERATIONS", 2: "SATISFIED_ABSOLUTE_TOL", 3: "SATISFIED_RELATIVE_TOL",
    4: "SATISFIED_FIRST_ORDER_TOL", 5: "QP_INDEFINITE", 6: "USER_CALLBACK", 7: "MAX_LAMBDA",
    8: "NON_FINITE",
}
TERM = {v: k for k, v in TERM_NAMES.items()}

KERNEL_PREPARE, KERNEL_LINEARIZE, KERNEL_QP_LS, KERNEL_FINALIZE, KERNEL_FUSED, KERNEL_COUNT = range(6)
PIPELINE_AUTO, PIPELINE_SPLIT, PIPELINE_FUSED = 0, 1, 2
PIPELINES = {"auto": 0, "split": 1, "fused": 2, 0: 0, 1: 1, 2: 2}

# every symbol include/cpmpc.h declares (tests check the library exports all of them)
SYMBOLS = [
    "cpmpc_default_params", "cpmpc_default_solver_opts", "cpmpc_last_error", "cpmpc_device_count",
    "cpmpc_create", "cpmpc_create_ex", "cpmpc_max_parity_horizon", "cpmpc_horizon_beyond_parity", "cpmpc_get_solver_opts", "cpmpc_refines_qp", "cpmpc_wide_qp", "cpmpc_destroy", "cpmpc_supported_state_spacing", "cpmpc_step_batch",
    "cpmpc_reset", "cpmpc_set_previous_solution", "cpmpc_get_solution",
    "cpmpc_has_previous_solution", "cpmpc_previous_solution_batch", "cpmpc_dim", "cpmpc_num_states", "cpmpc_dtype",
    "cpmpc_step_batch_host", "cpmpc_step_batch_host_ex", "cpmpc_set_previous_solution_host", "cpmpc_get_solution_host",
    "cpmpc_dynamics_batch", "cpmpc_rk4_batch", "cpmpc_linearize_batch", "cpmpc_sim_step_batch",
    "cpmpc_sim_step_batch_host", "cpmpc_model_state_dim", "cpmpc_model_num_params", "cpmpc_create_model",
    "cpmpc_model", "cpmpc_dynamics_batch_model", "cpmpc_rk4_batch_model", "cpmpc_sim_step_batch_model",
    "cpmpc_set_pipeline", "cpmpc_get_pipeline", "cpmpc_set_compaction", "cpmpc_get_stage_plan", "cpmpc_plan_stages_from_histogram",
    "cpmpc_profile_enable", "cpmpc_profile_reset", "cpmpc_profile_read", "cpmpc_kernel_name",
    "cpmpc_sharded_create", "cpmpc_sharded_destroy", "cpmpc_sharded_num_shards", "cpmpc_sharded_device", "cpmpc_sharded_peer_access",
    "cpmpc_sharded_handle", "cpmpc_sharded_range", "cpmpc_sharded_reset", "cpmpc_sharded_step_batch_host",
    "cpmpc_sharded_step_batch", "cpmpc_sharded_create_ex", "cpmpc_sharded_previous_solution_batch", "cpmpc_sharded_horizon_beyond_parity",
    "cpmpc_sharded_set_previous_solution", "cpmpc_sharded_set_previous_solution_host", "cpmpc_sharded_get_solution",
    "cpmpc_sharded_get_solution_host", "cpmpc_sharded_step_batch_host_in", "cpmpc_sharded_step_batch_ex",
    "cpmpc_step_batch_host_in", "cpmpc_set_host_chunk", "cpmpc_host_register", "cpmpc_host_unregister",
]


class Params(C.Structure):
    """cpmpc_params == pendulum::OptimizationParams (optimization/optimization.hpp:12-53)."""
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


class StepInputs(C.Structure):
    _fields_ = [
        ("x0", C.c_void_p),
        ("dyn_shared_host", C.POINTER(C.c_double)),
        ("dyn", C.c_void_p),
        ("set_point_shared", C.c_double),
        ("set_point", C.c_void_p),
        ("terminal_weights", C.c_void_p),
    ]


class StepOutputs(C.Structure):
    _fields_ = [
        ("u", C.c_void_p),
        ("predicted", C.c_void_p),
        ("status", C.c_void_p),
        ("iterations", C.c_void_p),
        ("ls_evals", C.c_void_p),
        ("final_cost", C.c_void_p),
        ("final_eq_l1", C.c_void_p),
        ("guess", C.c_void_p),
        ("solution", C.c_void_p),
    ]


class StepHostOutputs(C.Structure):
    """cpmpc_step_host_outputs: HOST pointers, every one nullable."""
    _fields_ = [
        ("u", C.POINTER(C.c_double)),
        ("predicted", C.POINTER(C.c_double)),
        ("status", C.POINTER(C.c_int32)),
        ("iterations", C.POINTER(C.c_int32)),
        ("final_cost", C.POINTER(C.c_double)),
        ("final_eq_l1", C.POINTER(C.c_double)),
        ("solution", C.POINTER(C.c_double)),
    ]


CREATE_ALLOW_LONG_HORIZON = 1
CREATE_REFINE_QP = 2
CREATE_NO_REFINE_QP = 4
CREATE_STRICT_HORIZON = 8
CREATE_WIDE_QP = 16
CREATE_NO_WIDE_QP = 32
SOLVER_OPTS_SIZE_POSITIONAL = 128


class CreateInfo(C.Structure):
    """cpmpc_create_info: size-versioned arguments of cpmpc_create_ex."""
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("flags", C.c_uint32),
        ("dtype", C.c_int32),
        ("model", C.c_int32),
        ("device", C.c_int32),
        ("reserved", C.c_int32),
        ("max_batch", C.c_int64),
        ("params", C.POINTER(Params)),
        ("opts", C.POINTER(SolverOpts)),
        ("opts_size", C.c_uint64),
    ]


class StepHostInputs(C.Structure):
    """cpmpc_step_host_inputs: HOST double arrays; exactly one of dyn_shared / dyn."""
    _fields_ = [
        ("x0", C.POINTER(C.c_double)),
        ("dyn_shared", C.POINTER(C.c_double)),
        ("dyn", C.POINTER(C.c_double)),
        ("set_point_shared", C.c_double),
        ("set_point", C.POINTER(C.c_double)),
        ("terminal_weights", C.POINTER(C.c_double)),
    ]


class CpmpcError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("cpmpc error %d: %s" % (code, text))
        self.code = code


_lib = None
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


def load():
    """Load libcpmpc.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH) and "CPMPC_LIB" not in os.environ:
        # not built yet (fresh checkout): build the product library itself -- this is not a fallback, there is none
        try:
            from . import build as _build
            _build.build_lib()
        except Exception as exc:  # no hipcc, compile error: report below with the reason
            raise ImportError("%s is missing and building it failed: %s.  There is no CPU fallback." % (LIB_PATH, exc))
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, i64, dbl, i32 = C.c_void_p, C.c_int64, C.c_double, C.c_int
    L.cpmpc_default_params.argtypes = [C.POINTER(Params)]
    L.cpmpc_default_params.restype = None
    L.cpmpc_default_solver_opts.argtypes = [C.POINTER(SolverOpts)]
    L.cpmpc_default_solver_opts.restype = None
    L.cpmpc_last_error.restype = C.c_char_p
    L.cpmpc_device_count.restype = i32
    L.cpmpc_create.argtypes = [C.POINTER(Params), C.POINTER(SolverOpts), i32, i64, i32,
                               C.POINTER(vp)]
    L.cpmpc_create_ex.argtypes = [C.POINTER(CreateInfo), C.POINTER(vp)]
    L.cpmpc_wide_qp.argtypes = [vp]
    L.cpmpc_wide_qp.restype = i32
    L.cpmpc_get_solver_opts.argtypes = [vp, C.POINTER(SolverOpts), C.c_size_t]
    L.cpmpc_get_solver_opts.restype = i32
    L.cpmpc_horizon_beyond_parity.argtypes = [vp]
    L.cpmpc_horizon_beyond_parity.restype = i32
    L.cpmpc_refines_qp.argtypes = [vp]
    L.cpmpc_max_parity_horizon.argtypes = []
    L.cpmpc_max_parity_horizon.restype = dbl
    L.cpmpc_destroy.argtypes = [vp]
    L.cpmpc_destroy.restype = None
    L.cpmpc_supported_state_spacing.argtypes = [i32]
    L.cpmpc_step_batch.argtypes = [vp, i64, C.POINTER(StepInputs), C.POINTER(StepOutputs), vp]
    L.cpmpc_reset.argtypes = [vp]
    L.cpmpc_set_previous_solution.argtypes = [vp, i64, vp, vp]
    L.cpmpc_get_solution.argtypes = [vp, i64, vp, vp]
    L.cpmpc_has_previous_solution.argtypes = [vp]
    L.cpmpc_previous_solution_batch.argtypes = [vp]
    L.cpmpc_previous_solution_batch.restype = i64
    L.cpmpc_dim.argtypes = [vp]
    L.cpmpc_num_states.argtypes = [vp]
    L.cpmpc_dtype.argtypes = [vp]
    L.cpmpc_step_batch_host.argtypes = [vp, i64, _dp, _dp, dbl, _dp, _dp, _ip, _ip, _dp, _dp]
    L.cpmpc_step_batch_host_ex.argtypes = [vp, i64, _dp, _dp, dbl, C.POINTER(StepHostOutputs)]
    L.cpmpc_set_previous_solution_host.argtypes = [vp, i64, _dp]
    L.cpmpc_get_solution_host.argtypes = [vp, i64, _dp]
    L.cpmpc_dynamics_batch.argtypes = [i32, i64, _dp, vp, vp, _dp, vp, vp, vp, vp]
    L.cpmpc_rk4_batch.argtypes = [i32, i64, _dp, vp, vp, dbl, _dp, vp, vp, vp, vp]
    L.cpmpc_linearize_batch.argtypes = [vp, i64, _dp, vp, vp, vp, vp, vp]
    L.cpmpc_sim_step_batch.argtypes = [i32, i64, _dp, dbl, vp, _dp, vp, vp, vp]
    L.cpmpc_sim_step_batch_host.argtypes = [i64, _dp, dbl, _dp, _dp, _dp]
    L.cpmpc_model_state_dim.argtypes = [i32]
    L.cpmpc_model_num_params.argtypes = [i32]
    L.cpmpc_create_model.argtypes = [C.POINTER(Params), C.POINTER(SolverOpts), i32, i64, i32, i32, C.POINTER(vp)]
    L.cpmpc_model.argtypes = [vp]
    L.cpmpc_dynamics_batch_model.argtypes = [i32, i32, i64, _dp, vp, vp, _dp, vp, vp, vp, vp]
    L.cpmpc_rk4_batch_model.argtypes = [i32, i32, i64, _dp, vp, vp, dbl, _dp, vp, vp, vp, vp]
    L.cpmpc_sim_step_batch_model.argtypes = [i32, i32, i64, _dp, dbl, vp, _dp, vp, vp, vp]
    L.cpmpc_set_pipeline.argtypes = [vp, i32]
    L.cpmpc_get_pipeline.argtypes = [vp]
    L.cpmpc_set_compaction.argtypes = [vp, i32, i32]
    L.cpmpc_get_stage_plan.argtypes = [vp, C.POINTER(C.c_int32), i32]
    L.cpmpc_plan_stages_from_histogram.argtypes = [C.POINTER(C.c_int64), i64, i32, i32, i32, i32, C.POINTER(C.c_int32), i32]
    L.cpmpc_profile_enable.argtypes = [vp, i32]
    L.cpmpc_profile_reset.argtypes = [vp]
    L.cpmpc_profile_read.argtypes = [vp, i32, _dp, C.POINTER(C.c_int64)]
    L.cpmpc_kernel_name.argtypes = [i32]
    L.cpmpc_kernel_name.restype = C.c_char_p
    L.cpmpc_sharded_create.argtypes = [C.POINTER(Params), C.POINTER(SolverOpts), i32, i64, C.POINTER(C.c_int), i32,
                                       C.POINTER(vp)]
    L.cpmpc_sharded_destroy.argtypes = [vp]
    L.cpmpc_sharded_destroy.restype = None
    L.cpmpc_sharded_num_shards.argtypes = [vp]
    L.cpmpc_sharded_peer_access.argtypes = [vp, i32]
    L.cpmpc_sharded_peer_access.restype = i32
    L.cpmpc_sharded_device.argtypes = [vp, i32]
    L.cpmpc_sharded_handle.argtypes = [vp, i32]
    L.cpmpc_sharded_handle.restype = vp
    L.cpmpc_sharded_range.argtypes = [vp, i32, i64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.cpmpc_sharded_reset.argtypes = [vp]
    L.cpmpc_sharded_step_batch_host.argtypes = [vp, i64, _dp, _dp, dbl, C.POINTER(StepHostOutputs)]
    L.cpmpc_sharded_step_batch.argtypes = [vp, i64, vp, _dp, dbl, C.POINTER(StepOutputs), vp]
    L.cpmpc_sharded_create_ex.argtypes = [C.POINTER(CreateInfo), C.POINTER(C.c_int), i32, C.POINTER(vp)]
    L.cpmpc_sharded_horizon_beyond_parity.argtypes = [vp]
    L.cpmpc_sharded_horizon_beyond_parity.restype = i32
    L.cpmpc_sharded_previous_solution_batch.argtypes = [vp]
    L.cpmpc_sharded_previous_solution_batch.restype = i64
    L.cpmpc_sharded_set_previous_solution.argtypes = [vp, i64, vp, vp]
    L.cpmpc_sharded_set_previous_solution_host.argtypes = [vp, i64, _dp]
    L.cpmpc_sharded_get_solution.argtypes = [vp, i64, vp, vp]
    L.cpmpc_sharded_get_solution_host.argtypes = [vp, i64, _dp]
    L.cpmpc_sharded_step_batch_host_in.argtypes = [vp, i64, C.POINTER(StepHostInputs), C.POINTER(StepHostOutputs)]
    L.cpmpc_sharded_step_batch_ex.argtypes = [vp, i64, C.POINTER(StepInputs), C.POINTER(StepOutputs), vp]
    L.cpmpc_step_batch_host_in.argtypes = [vp, i64, C.POINTER(StepHostInputs), C.POINTER(StepHostOutputs)]
    L.cpmpc_set_host_chunk.argtypes = [vp, i64]
    L.cpmpc_host_register.argtypes = [vp, C.c_uint64]
    L.cpmpc_host_unregister.argtypes = [vp]
    _lib = L
    return L


def check(rc):
    if rc != OK:
        raise CpmpcError(rc, load().cpmpc_last_error().decode("utf-8", "replace"))


def default_params(**overrides):
    p = Params()
    load().cpmpc_default_params(C.byref(p))
    for k, v in overrides.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def default_solver_opts(**overrides):
    o = SolverOpts()
    load().cpmpc_default_solver_opts(C.byref(o))
    for k, v in overrides.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


def dbl_array(values, n):
    vals = [float(v) for v in values]
    if len(vals) != n:
        raise ValueError("expected %d values, got %d" % (n, len(vals)))
    return (C.c_double * n)(*vals)
