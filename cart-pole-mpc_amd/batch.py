"""Batched host API over the C-ABI: torch tensors (device memory + streams only) in, tensors out.

Names follow the reference's interface for this path (optimization/optimization.hpp:73-108,
optimization/simulator.hpp:10-29): Optimization -> BatchOptimization with step / reset /
set_previous_solution; Simulator -> BatchSimulator with step / get_state / set_state.
Every array is structure-of-arrays, [field, B].  All compute happens in libcpmpc.so's HIP kernels;
this module raises if the library or a gfx950 device is missing.
"""
import ctypes as C

import torch

from . import capi

_TORCH_DTYPE = {capi.F32: torch.float32, capi.F64: torch.float64}
_CAPI_DTYPE = {torch.float32: capi.F32, torch.float64: capi.F64}


def _stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _require_cuda_tensor(t, name, dtype, shape=None):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise TypeError("%s must be a GPU tensor" % name)
    if t.dtype != dtype:
        raise TypeError("%s must have dtype %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous ([field, B], batch fastest)" % name)
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError("%s must have shape %s, got %s" % (name, tuple(shape), tuple(t.shape)))
    return t


def _ptr(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


class BatchOutputs:
    """Batched OptimizationOutputs (optimization/optimization.hpp:55-70)."""

    def __init__(self):
        self.u = None                 # [N, B]
        self.predicted_states = None  # [N, 4, B]
        self.status = None            # [B] int32, termination state
        self.iterations = None        # [B] int32
        self.ls_evals = None          # [B] int32
        self.final_cost = None        # [B]
        self.final_eq_l1 = None       # [B]
        self.guess = None             # [dim, B]
        self.horizon_beyond_parity = False   # cpmpc_horizon_beyond_parity() of the handle that filled these outputs

    def solver_summary(self):
        """Batch summary in the spirit of NLSSolverOutputs::ToString (wrapper/wrapper.cc:82-83)."""
        st = self.status.cpu()
        lines = ["batch of %d problems" % st.numel()]
        if getattr(self, "horizon_beyond_parity", False):
            lines.append("  horizon beyond %.1f s (cpmpc_max_parity_horizon): a few cold starts in 10^4 may differ from a "
                         "full-space solve by more than 1e-5" % capi.load().cpmpc_max_parity_horizon())
        for code in sorted(set(st.tolist())):
            lines.append("  %-28s %d" % (capi.TERM_NAMES.get(code, str(code)), int((st == code).sum())))
        if self.iterations is not None:
            it = self.iterations.cpu().float()
            lines.append("  iterations: mean %.2f max %d" % (it.mean().item(), int(it.max().item())))
        if self.final_eq_l1 is not None:
            lines.append("  max |c|_1: %.3e" % self.final_eq_l1.abs().max().item())
        return "\n".join(lines)

    def lane_json(self, lane, initial_state, previous_solution=None):
        """One problem of the batch in the JSON log format of OptimizationOutputs (optimization/wasm.cc:46-65,
        toJson :104-105): keys sorted, compact, as nlohmann::json::dump() prints them.  `initial_state` is the
        [4, B] tensor passed to step(); `previous_solution` the [dim, B] tensor of get_solution() (optional).
        solver_outputs carries this repo's fields (mini_opt's serialisation is not available)."""
        import json

        def col(t):
            return [float(v) for v in t[..., lane].double().cpu().reshape(-1)]

        names = ("b_x", "th_1", "b_x_dot", "th_1_dot")
        pred = self.predicted_states[:, :, lane].double().cpu().tolist()
        obj = {
            "initial_state": dict(zip(names, col(initial_state))),
            "previous_solution": col(previous_solution) if previous_solution is not None else [],
            "solver_outputs": {
                "termination_state": capi.TERM_NAMES[int(self.status[lane])],
                "iterations": int(self.iterations[lane]) if self.iterations is not None else 0,
                "final_cost": float(self.final_cost[lane]) if self.final_cost is not None else 0.0,
                "final_equality_l1": float(self.final_eq_l1[lane]) if self.final_eq_l1 is not None else 0.0,
            },
            "u": col(self.u),
            "predicted_states": [dict(zip(names, row)) for row in pred],
        }
        return json.dumps(obj, sort_keys=True, separators=(",", ":"))


class BatchOptimization:
    """B independent pendulum::Optimization controllers solved in lock-step on one GPU."""

    def __init__(self, params, max_batch, dtype=torch.float32, device=None, opts=None, model="single",
                 allow_long_horizon=False, refine_qp=None, strict_horizon=False, wide_qp=None):
        """strict_horizon: refuse (CpmpcError(ERR_UNSUPPORTED)) window_length * control_dt beyond
        cpmpc_max_parity_horizon() (1.0 s), where the condensed QP is no longer held to 1e-5 of a full-space solve on
        every problem (include/cpmpc.h, CPMPC_CREATE_STRICT_HORIZON).  By default every horizon the reference accepts is
        accepted, with one warning per process; allow_long_horizon=True silences the warning.
        refine_qp: True / False force on / off the refinement of the whole QP solution in the fp64 fused kernels
        (CPMPC_CREATE_[NO_]REFINE_QP: 7 % slower); None = the library's default: on when u_cost_weight < 0.05.
        wide_qp: True / False force on / off that float32 handles carry the QP's whole terminal part in double
        (CPMPC_CREATE_[NO_]WIDE_QP: cold starts end where a float solve can -- 3x to 20x closer to the double check for the
        4-state model at 3.8 % of the throughput, 70x to 900x for the 6-state one at 0 - 5 %); None = the library's
        default: on for the 6-state model, off for the 4-state one."""
        lib = capi.load()
        self.model = capi.MODELS[model]
        self.nx = lib.cpmpc_model_state_dim(self.model)
        self.np = lib.cpmpc_model_num_params(self.model)
        if dtype not in _CAPI_DTYPE:
            raise TypeError("dtype must be torch.float32 or torch.float64")
        if device is None:
            device = torch.cuda.current_device()
        self.params = params
        self.opts = opts
        self.dtype = dtype
        self.device = int(device)
        self.max_batch = int(max_batch)
        self._h = C.c_void_p()
        info = capi.CreateInfo(struct_size=C.sizeof(capi.CreateInfo),
                               flags=(capi.CREATE_ALLOW_LONG_HORIZON if allow_long_horizon else 0)
                               | (capi.CREATE_STRICT_HORIZON if strict_horizon else 0)
                               | (0 if wide_qp is None else (capi.CREATE_WIDE_QP if wide_qp else capi.CREATE_NO_WIDE_QP))
                               | (0 if refine_qp is None else (capi.CREATE_REFINE_QP if refine_qp else capi.CREATE_NO_REFINE_QP)),
                               dtype=_CAPI_DTYPE[dtype], model=self.model, device=self.device, reserved=0,
                               max_batch=self.max_batch, params=C.pointer(params),
                               opts=C.pointer(opts) if opts is not None else None,
                               opts_size=C.sizeof(capi.SolverOpts) if opts is not None else 0)
        capi.check(lib.cpmpc_create_ex(C.byref(info), C.byref(self._h)))
        self.N = int(params.window_length)
        self.S = lib.cpmpc_num_states(self._h)
        self.dim = lib.cpmpc_dim(self._h)
        self._keep = None

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            capi.load().cpmpc_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- Optimization::Step -------------------------------------------------------------------
    def step(self, x0, dyn, set_point=0.0, want_predicted=True, want_stats=True, want_guess=False,
             out=None, terminal_weights=None):
        """x0: [nx, B] tensor.  dyn: np floats (shared) or an [np, B] tensor.  set_point: float or [B].
        terminal_weights: optional [nx, B] tensor of per-problem terminal weights in state order (>= 0 cost row,
        < 0 equality row; optimization.cc:236-267), overriding the four *_final_cost_weight parameters."""
        lib = capi.load()
        dev = torch.device("cuda", self.device)
        _require_cuda_tensor(x0, "x0", self.dtype)
        if x0.dim() != 2 or x0.shape[0] != self.nx:
            raise ValueError("x0 must be [%d, B]" % self.nx)
        B = int(x0.shape[1])
        inp = capi.StepInputs()
        inp.x0 = x0.data_ptr()
        keep = [x0]
        if isinstance(dyn, torch.Tensor):
            _require_cuda_tensor(dyn, "dyn", self.dtype, (self.np, B))
            inp.dyn = dyn.data_ptr()
            inp.dyn_shared_host = None
            keep.append(dyn)
        else:
            arr = capi.dbl_array(dyn, self.np)
            inp.dyn_shared_host = C.cast(arr, C.POINTER(C.c_double))
            inp.dyn = None
            keep.append(arr)
        if isinstance(set_point, torch.Tensor):
            _require_cuda_tensor(set_point, "set_point", self.dtype, (B,))
            inp.set_point = set_point.data_ptr()
            inp.set_point_shared = 0.0
            keep.append(set_point)
        else:
            inp.set_point = None
            inp.set_point_shared = float(set_point)
        if terminal_weights is not None:
            _require_cuda_tensor(terminal_weights, "terminal_weights", self.dtype, (self.nx, B))
            inp.terminal_weights = terminal_weights.data_ptr()
            keep.append(terminal_weights)
        else:
            inp.terminal_weights = None

        o = out if out is not None else BatchOutputs()

        def reusable(t, shape, dtype):
            # a buffer handed back in `out` is written by the kernels as `dtype` on `dev`: anything else (another
            # optimizer's outputs of the other precision, another device) must be replaced, never written through
            return (t is not None and tuple(t.shape) == tuple(shape) and t.dtype == dtype and t.device == dev
                    and t.is_contiguous())
        if not reusable(o.u, (self.N, B), self.dtype):
            o.u = torch.empty((self.N, B), dtype=self.dtype, device=dev)
        if want_predicted and not reusable(o.predicted_states, (self.N, self.nx, B), self.dtype):
            o.predicted_states = torch.empty((self.N, self.nx, B), dtype=self.dtype, device=dev)
        if not reusable(o.status, (B,), torch.int32):
            o.status = torch.empty((B,), dtype=torch.int32, device=dev)
        if want_stats and not (reusable(o.iterations, (B,), torch.int32) and reusable(o.ls_evals, (B,), torch.int32)
                               and reusable(o.final_cost, (B,), self.dtype) and reusable(o.final_eq_l1, (B,), self.dtype)):
            o.iterations = torch.empty((B,), dtype=torch.int32, device=dev)
            o.ls_evals = torch.empty((B,), dtype=torch.int32, device=dev)
            o.final_cost = torch.empty((B,), dtype=self.dtype, device=dev)
            o.final_eq_l1 = torch.empty((B,), dtype=self.dtype, device=dev)
        if want_guess:
            o.guess = torch.empty((self.dim, B), dtype=self.dtype, device=dev)
        outp = capi.StepOutputs()
        outp.u = o.u.data_ptr()
        outp.predicted = o.predicted_states.data_ptr() if want_predicted else None
        outp.status = o.status.data_ptr()
        if want_stats:
            outp.iterations = o.iterations.data_ptr()
            outp.ls_evals = o.ls_evals.data_ptr()
            outp.final_cost = o.final_cost.data_ptr()
            outp.final_eq_l1 = o.final_eq_l1.data_ptr()
        outp.guess = o.guess.data_ptr() if want_guess else None
        with torch.cuda.device(self.device):
            capi.check(lib.cpmpc_step_batch(self._h, B, C.byref(inp), C.byref(outp), _stream_ptr()))
        self._keep = keep  # inputs must outlive the asynchronous launch
        o.horizon_beyond_parity = self.horizon_beyond_parity
        return o

    # -- Optimization::Reset / SetPreviousSolution ---------------------------------------------
    def reset(self):
        capi.check(capi.load().cpmpc_reset(self._h))

    def has_previous_solution(self):
        return bool(capi.load().cpmpc_has_previous_solution(self._h))

    def previous_solution_batch(self):
        """Problems [0, n) hold a previous solution (warm start); the others cold-start at the next step."""
        return int(capi.load().cpmpc_previous_solution_batch(self._h))

    def set_previous_solution(self, z):
        _require_cuda_tensor(z, "z", self.dtype)
        if z.dim() != 2 or z.shape[0] != self.dim:
            raise ValueError("z must be [dim=%d, B]" % self.dim)
        with torch.cuda.device(self.device):
            capi.check(capi.load().cpmpc_set_previous_solution(self._h, int(z.shape[1]), _ptr(z),
                                                               _stream_ptr()))

    def get_solution(self, B):
        z = torch.empty((self.dim, int(B)), dtype=self.dtype, device=torch.device("cuda", self.device))
        with torch.cuda.device(self.device):
            capi.check(capi.load().cpmpc_get_solution(self._h, int(B), _ptr(z), _stream_ptr()))
        return z

    # -- pieces -------------------------------------------------------------------------------
    def linearize(self, z, dyn):
        """Shooting constraints linearised at z [dim, B] -> (c [4(S-1), B], Phi [S-1,4,4,B],
        Gamma [N,4,B] with Gamma[k, r] = d x_end[r] / d u_k)."""
        _require_cuda_tensor(z, "z", self.dtype)
        B = int(z.shape[1])
        dev = z.device
        nx = self.nx
        c = torch.empty((nx * (self.S - 1), B), dtype=self.dtype, device=dev)
        Phi = torch.empty((self.S - 1, nx, nx, B), dtype=self.dtype, device=dev)
        Gam = torch.empty((self.N, nx, B), dtype=self.dtype, device=dev)
        arr = capi.dbl_array(dyn, self.np)
        with torch.cuda.device(self.device):
            capi.check(capi.load().cpmpc_linearize_batch(self._h, B, arr, _ptr(z), _ptr(c), _ptr(Phi),
                                                         _ptr(Gam), _stream_ptr()))
        return c, Phi, Gam

    # -- pipeline selection --------------------------------------------------------------------
    def set_pipeline(self, mode):
        """'auto' | 'split' | 'fused' (include/cpmpc.h: CPMPC_PIPELINE_*)."""
        capi.check(capi.load().cpmpc_set_pipeline(self._h, capi.PIPELINES[mode]))

    @property
    def horizon_beyond_parity(self):
        """True: window_length * control_dt exceeds cpmpc_max_parity_horizon() (include/cpmpc.h: cpmpc_horizon_beyond_parity)."""
        return bool(capi.load().cpmpc_horizon_beyond_parity(self._h))

    @property
    def refines_qp(self):
        return bool(capi.load().cpmpc_refines_qp(self._h))

    @property
    def wide_qp(self):
        """True: this float handle's steps carry the QP's terminal part in double (include/cpmpc.h: CPMPC_CREATE_WIDE_QP)."""
        return bool(capi.load().cpmpc_wide_qp(self._h))

    def set_compaction(self, first_iterations=2, next_iterations=1):
        """Staging of the fused pipeline when exit tolerances are enabled (0, 0 = one launch).  Speed only."""
        capi.check(capi.load().cpmpc_set_compaction(self._h, int(first_iterations), int(next_iterations)))

    def stage_plan(self):
        """Boundaries [0, ..., max_iterations] of the launches of the fused kernel in the last step (cpmpc_get_stage_plan)."""
        buf = (C.c_int32 * 32)()
        n = capi.load().cpmpc_get_stage_plan(self._h, buf, 32)
        return [int(buf[i]) for i in range(n + 1)] if n >= 0 else []

    def pipeline(self):
        return {capi.PIPELINE_SPLIT: "split", capi.PIPELINE_FUSED: "fused"}[capi.load().cpmpc_get_pipeline(self._h)]

    # -- measurement --------------------------------------------------------------------------
    def profile_enable(self, on=True):
        capi.check(capi.load().cpmpc_profile_enable(self._h, 1 if on else 0))

    def profile_reset(self):
        capi.check(capi.load().cpmpc_profile_reset(self._h))

    def profile_read(self):
        """{kernel name: (total_ms, launches)} accumulated since the last reset."""
        lib = capi.load()
        res = {}
        for kid in range(capi.KERNEL_COUNT):
            ms, n = C.c_double(), C.c_int64()
            capi.check(lib.cpmpc_profile_read(self._h, kid, C.byref(ms), C.byref(n)))
            res[lib.cpmpc_kernel_name(kid).decode()] = (ms.value, n.value)
        return res


def _fext(fext):
    return None if fext is None else capi.dbl_array(fext, 4)


def _model_dims(model):
    lib = capi.load()
    m = capi.MODELS[model]
    return m, lib.cpmpc_model_state_dim(m), lib.cpmpc_model_num_params(m)


def dynamics_batch(dyn, x, u, fext=None, jacobians=True, model="single"):
    """Forward dynamics batched (gen::single_pendulum_dynamics or the generated double pendulum):
    x [nx,B], u [B] -> f [nx,B] (+ Jx [nx,nx,B], Ju [nx,B])."""
    m, nx, npar = _model_dims(model)
    dt = x.dtype
    _require_cuda_tensor(x, "x", dt)
    B = int(x.shape[1])
    if x.shape[0] != nx:
        raise ValueError("x must be [%d, B]" % nx)
    _require_cuda_tensor(u, "u", dt, (B,))
    f = torch.empty_like(x)
    Jx = torch.empty((nx, nx, B), dtype=dt, device=x.device) if jacobians else None
    Ju = torch.empty((nx, B), dtype=dt, device=x.device) if jacobians else None
    with torch.cuda.device(x.device):
        capi.check(capi.load().cpmpc_dynamics_batch_model(m, _CAPI_DTYPE[dt], B, capi.dbl_array(dyn, npar), _ptr(x),
                                                          _ptr(u), _fext(fext), _ptr(f), _ptr(Jx), _ptr(Ju),
                                                          _stream_ptr()))
    return (f, Jx, Ju) if jacobians else f


def rk4_batch(dyn, x, u, h, fext=None, jacobians=True, model="single"):
    """runge_kutta_4th_order<D> (or _no_jacobians) batched: -> x_new [nx,B] (+ A [nx,nx,B], B [nx,B])."""
    m, nx, npar = _model_dims(model)
    dt = x.dtype
    _require_cuda_tensor(x, "x", dt)
    B = int(x.shape[1])
    if x.shape[0] != nx:
        raise ValueError("x must be [%d, B]" % nx)
    _require_cuda_tensor(u, "u", dt, (B,))
    xn = torch.empty_like(x)
    A = torch.empty((nx, nx, B), dtype=dt, device=x.device) if jacobians else None
    Bm = torch.empty((nx, B), dtype=dt, device=x.device) if jacobians else None
    with torch.cuda.device(x.device):
        capi.check(capi.load().cpmpc_rk4_batch_model(m, _CAPI_DTYPE[dt], B, capi.dbl_array(dyn, npar), _ptr(x),
                                                     _ptr(u), float(h), _fext(fext), _ptr(xn), _ptr(A), _ptr(Bm),
                                                     _stream_ptr()))
    return (xn, A, Bm) if jacobians else xn


class BatchSimulator:
    """B independent pendulum::Simulator plants (optimization/simulator.hpp:10-29)."""

    def __init__(self, batch, dtype=torch.float32, device=None, model="single"):
        self.model, self.nx, self.np = _model_dims(model)
        if device is None:
            device = torch.cuda.current_device()
        self.device = torch.device("cuda", int(device))
        self.dtype = dtype
        hang = -3.14159265358979323846 / 2  # poles hanging, at rest (simulator.hpp:28)
        init = torch.tensor([0.0] + [hang] * (self.nx // 2 - 1) + [0.0] * (self.nx // 2), dtype=dtype)
        self.state = init.to(self.device).reshape(self.nx, 1).repeat(1, int(batch)).contiguous()

    def get_state(self):
        return self.state

    def set_state(self, state):
        _require_cuda_tensor(state, "state", self.dtype, tuple(self.state.shape))
        self.state = state.clone()

    def step(self, params, dt, u, f_base=(0.0, 0.0), f_mass=(0.0, 0.0), fext=None):
        """u: [B] tensor.  f_base/f_mass shared (x, y) pairs, or fext a [4, B] tensor
        {f_base.x, f_base.y, f_mass.x, f_mass.y}."""
        B = int(self.state.shape[1])
        _require_cuda_tensor(u, "u", self.dtype, (B,))
        shared = capi.dbl_array([f_base[0], f_base[1], f_mass[0], f_mass[1]], 4)
        if fext is not None:
            _require_cuda_tensor(fext, "fext", self.dtype, (4, B))
        with torch.cuda.device(self.device):
            capi.check(capi.load().cpmpc_sim_step_batch_model(self.model, _CAPI_DTYPE[self.dtype], B,
                                                              capi.dbl_array(params, self.np), float(dt), _ptr(u),
                                                              shared, _ptr(fext), _ptr(self.state), _stream_ptr()))


class ClosedLoop:
    """B controllers and their plants in closed loop on one GPU, all state resident on the device: per tick, for every
    controller, Optimization::Step on the plant's state (optimization.cc:39-97, warm-started after the first tick), the first
    control applied, Simulator::Step(control_dt) (simulator.cc:11-36) -- optimization_test.cc:39-61 for a batch.

    `ranges` > 1 holds the batch as that many independent column ranges, each with its own solver handle, plant and
    stream, ticked one after the other without any synchronisation between them.  Nothing couples two controllers, so the
    results are bitwise those of one range (a problem's arithmetic never depends on the lanes it occupies), and the ranges
    drift out of phase: while one range is in its fused SQP kernel -- one wave per SIMD in fp64, which leaves a fifth of the
    SIMD's issue slots idle -- another range's `prepare` / plant / `finalize` (no LDS, about 110 registers) runs beside it.
    Measured, 262 144 settled fp64 controllers (round 5): 0.843 ms per tick as one range, 0.80 as two; fp32 0.677 / 0.632
    (three).  A caller that needs every control of a tick before the next one starts synchronises the streams itself
    (`synchronize()`); the loop as such never does."""

    def __init__(self, params, batch, dtype=torch.float64, device=None, ranges=1, opts=None, model="single", pipeline="auto"):
        if device is None:
            device = torch.cuda.current_device()
        self.device = torch.device("cuda", int(device))
        self.batch, self.dtype = int(batch), dtype
        ranges = max(1, min(int(ranges), self.batch))
        self.bounds = [self.batch * i // ranges for i in range(ranges + 1)]
        self.sims, self.opts, self.outs, self.streams = [], [], [], []
        for i in range(ranges):
            n = self.bounds[i + 1] - self.bounds[i]
            self.sims.append(BatchSimulator(n, dtype=dtype, device=device, model=model))
            o = BatchOptimization(params, max_batch=n, dtype=dtype, device=device, opts=opts, model=model)
            o.set_pipeline(pipeline)
            self.opts.append(o)
            self.outs.append(BatchOutputs())
            # one range: the caller's stream, like BatchOptimization alone; several: a stream each
            self.streams.append(torch.cuda.Stream(device=self.device) if ranges > 1 else None)
        self.ticks = 0
        self._inputs_dirty = len(self.sims) > 1   # the plants' initial states were written on the caller's stream

    def set_state(self, state):
        """state: [nx, B] tensor, columns in the caller's order.  May be called at any tick (a disturbance, a reset): with
        several ranges the copy is made ON each range's stream after that stream has waited for the caller's, so neither the
        range's solver / plant can read the new state before it has landed nor the caller's allocator hand the old one out
        while the range's plant is still reading it."""
        cur = torch.cuda.current_stream(self.device)
        for i, s in enumerate(self.sims):
            part = state[:, self.bounds[i]:self.bounds[i + 1]]
            st = self.streams[i]
            if st is None:
                s.set_state(part.contiguous())
                continue
            st.wait_stream(cur)                      # whatever produced `state` on the caller's stream comes first
            old = s.state
            _require_cuda_tensor(state, "state", self.dtype)
            if tuple(part.shape) != tuple(old.shape):
                raise ValueError("state must be [%d, %d]" % (old.shape[0], self.batch))
            with torch.cuda.stream(st):
                # the copy (contiguous: a column range of `state` is a strided view) is queued on, and belongs to the pool
                # of, the range's stream
                s.state = part.clone(memory_format=torch.contiguous_format)
            part.record_stream(st)                   # `state` is read there: its memory is not reused before that copy
            old.record_stream(st)                    # the plant of the last tick may still be reading the replaced tensor
        self._inputs_dirty = len(self.sims) > 1

    def _on(self, i):
        st = self.streams[i]
        return torch.cuda.stream(st) if st is not None else torch.cuda.stream(torch.cuda.current_stream(self.device))

    def tick(self, dyn, set_point=0.0, dt=0.01, want_stats=True):
        """One MPC tick of every controller (queued, not waited for)."""
        if isinstance(dyn, torch.Tensor):
            raise TypeError("ClosedLoop.tick: dyn is the plant's parameter set too (cpmpc_sim_step_batch takes shared host "
                            "parameters): pass the %d numbers, not a tensor" % self.sims[0].np)
        tensors = [t for t in (set_point,) if isinstance(t, torch.Tensor)]
        if len(self.sims) > 1 and (self._inputs_dirty or tensors):
            # whatever prepared the states -- or this tick's per-problem parameters / set-points -- on the caller's stream
            # comes first.  Shared (host) parameters and no set_state since the last tick: no wait, the ranges run free.
            cur = torch.cuda.current_stream(self.device)
            for st in self.streams:
                st.wait_stream(cur)
                for t in tensors:
                    t.record_stream(st)
            self._inputs_dirty = False
        one = len(self.sims) == 1
        for i, (s, o, out) in enumerate(zip(self.sims, self.opts, self.outs)):
            lo, hi = self.bounds[i], self.bounds[i + 1]
            with self._on(i):
                # per-problem set-points [B]: this range's columns (copied on the range's stream)
                sp = set_point if one or not isinstance(set_point, torch.Tensor) else set_point[lo:hi].contiguous()
                r = o.step(s.get_state(), dyn, sp, want_predicted=False, want_stats=want_stats, out=out)
                s.step(dyn, dt, r.u[0].contiguous())
        self.ticks += 1

    def synchronize(self):
        """Make the caller's stream wait for everything queued on the ranges' streams."""
        cur = torch.cuda.current_stream(self.device)
        for st in self.streams:
            if st is not None:
                cur.wait_stream(st)

    def state(self):
        self.synchronize()
        return torch.cat([s.get_state() for s in self.sims], dim=1)

    def controls(self):
        """u [N, B] of the last tick."""
        self.synchronize()
        return torch.cat([o.u for o in self.outs], dim=1)

    def iterations(self):
        self.synchronize()
        return torch.cat([o.iterations for o in self.outs])

    def status(self):
        self.synchronize()
        return torch.cat([o.status for o in self.outs])

    def stage_plan(self):
        return self.opts[0].stage_plan()

    def close(self):
        for o in self.opts:
            o.close()
