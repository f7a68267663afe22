"""The C-ABI library loads, exports every symbol include/cpmpc.h declares, mirrors the reference's
parameter struct and fails loudly without a GPU.  No compute calls (CPU only)."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def lib(pkg):
    import __graft_entry__
    __graft_entry__.build()
    return pkg.capi.load()


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "cpmpc.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cpmpc_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported(lib, pkg):
    declared = _declared_functions()
    assert len(declared) >= 25
    assert sorted(pkg.capi.SYMBOLS) == declared
    raw = C.CDLL(pkg.capi.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), name


def test_header_is_plain_c_and_struct_layouts_match(lib, pkg, tmp_path):
    """include/cpmpc.h compiles as C99 (the boundary is a C-ABI: cgo / JNI / ctypes bind it) and the ctypes
    mirrors in capi.py have the sizes and field offsets the C compiler gives the structs."""
    import subprocess
    src = tmp_path / "abi.c"
    fields = {
        "cpmpc_params": [f for f, _ in pkg.capi.Params._fields_],
        "cpmpc_solver_opts": [f for f, _ in pkg.capi.SolverOpts._fields_],
        "cpmpc_step_inputs": [f for f, _ in pkg.capi.StepInputs._fields_],
        "cpmpc_step_outputs": [f for f, _ in pkg.capi.StepOutputs._fields_],
        "cpmpc_step_host_outputs": [f for f, _ in pkg.capi.StepHostOutputs._fields_],
        "cpmpc_step_host_inputs": [f for f, _ in pkg.capi.StepHostInputs._fields_],
        "cpmpc_create_info": [f for f, _ in pkg.capi.CreateInfo._fields_],
    }
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "cpmpc.h"', "int main(void) {"]
    for st, fs in fields.items():
        lines.append('  printf("%s %%zu\\n", sizeof(%s));' % (st, st))
        for f in fs:
            lines.append('  printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (st, f, st, f))
    lines += ["  return 0;", "}"]
    src.write_text("\n".join(lines))
    exe = tmp_path / "abi"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                           "-o", str(exe), str(src)])
    got = dict(l.split() for l in subprocess.check_output([str(exe)], text=True).splitlines())
    mirrors = {"cpmpc_params": pkg.capi.Params, "cpmpc_solver_opts": pkg.capi.SolverOpts,
               "cpmpc_step_inputs": pkg.capi.StepInputs, "cpmpc_step_outputs": pkg.capi.StepOutputs,
               "cpmpc_step_host_outputs": pkg.capi.StepHostOutputs, "cpmpc_step_host_inputs": pkg.capi.StepHostInputs,
               "cpmpc_create_info": pkg.capi.CreateInfo}
    for st, cls in mirrors.items():
        assert int(got[st]) == C.sizeof(cls), st
        for f, _ in cls._fields_:
            assert int(got["%s.%s" % (st, f)]) == getattr(cls, f).offset, (st, f)


def test_params_mirror_reference_defaults(lib, pkg, orc):
    """optimization/optimization.hpp:12-48, field for field, and equal to the oracle's mirror."""
    p = pkg.default_params()
    want = dict(control_dt=0.01, window_length=40, state_spacing=10, max_iterations=8,
                relative_exit_tol=1.0e-5, absolute_first_derivative_tol=1.0e-6,
                equality_penalty_initial=1.0, u_guess_sinusoid_amplitude=10.0, u_cost_weight=0.1,
                u_derivative_cost_weight=0.1, b_x_final_cost_weight=150.0, th_final_cost_weight=-1.0,
                b_x_dot_final_cost_weight=-1.0, th_dot_final_cost_weight=-1.0)
    assert [f for f, _ in p._fields_] == list(want)
    op = orc.default_opt_params()
    for k, v in want.items():
        assert getattr(p, k) == v and getattr(op, k) == v
    assert C.sizeof(p) == 14 * 8
    o, oo = pkg.default_solver_opts(), orc.default_solver_opts()
    assert [f for f, _ in o._fields_] == [f for f, _ in oo._fields_]
    for f, _ in o._fields_:
        assert getattr(o, f) == getattr(oo, f), f
    assert o.max_line_search_iterations == 5  # optimization.cc:76


def test_supported_spacings(lib):
    for sp in (1, 2, 4, 5, 8, 10, 20):
        assert lib.cpmpc_supported_state_spacing(sp) == 2     # specialised kernels
    for sp in (3, 6, 7, 15, 40):
        assert lib.cpmpc_supported_state_spacing(sp) == 1     # generic run-time-spacing kernel
    assert lib.cpmpc_supported_state_spacing(0) == 0 and lib.cpmpc_supported_state_spacing(-1) == 0


def test_kernel_names(lib, pkg):
    names = [lib.cpmpc_kernel_name(i).decode() for i in range(pkg.capi.KERNEL_COUNT)]
    assert names == ["prepare_kernel", "linearize_kernel", "qp_ls_kernel", "finalize_kernel", "fused_sqp_kernel"]


def test_create_validates_like_the_reference_constructor(lib, pkg):
    """optimization.cc:13-22 preconditions -> CPMPC_ERR_INVALID_ARG, before any device is touched."""
    h = C.c_void_p()
    for bad in (dict(control_dt=0.0), dict(window_length=0), dict(state_spacing=7),
                dict(max_iterations=0), dict(u_cost_weight=-1.0), dict(u_derivative_cost_weight=-1.0)):
        p = pkg.default_params(**bad)
        rc = lib.cpmpc_create(C.byref(p), None, pkg.capi.F32, 64, 0, C.byref(h))
        assert rc == pkg.capi.ERR_INVALID_ARG, bad
        assert lib.cpmpc_last_error()
    p = pkg.default_params(state_spacing=40)  # valid in the reference: accepted (generic kernel); here it gets as far
    assert lib.cpmpc_create(C.byref(p), None, pkg.capi.F32, 64, 0, C.byref(h)) == pkg.capi.ERR_NO_DEVICE  # as the device
    p = pkg.default_params(window_length=8192, state_spacing=8192)
    assert lib.cpmpc_create(C.byref(p), None, pkg.capi.F32, 64, 0, C.byref(h)) == pkg.capi.ERR_UNSUPPORTED
    p = pkg.default_params()
    assert lib.cpmpc_create(C.byref(p), None, 7, 64, 0, C.byref(h)) == pkg.capi.ERR_INVALID_ARG
    assert lib.cpmpc_create(C.byref(p), None, pkg.capi.F32, 0, 0, C.byref(h)) == pkg.capi.ERR_INVALID_ARG


def test_fails_loudly_without_a_gpu(lib, pkg):
    """No CPU fallback: with no gfx950 device every compute entry point reports CPMPC_ERR_NO_DEVICE."""
    if lib.cpmpc_device_count() > 0:
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    p = pkg.default_params()
    rc = lib.cpmpc_create(C.byref(p), None, pkg.capi.F64, 64, 0, C.byref(h))
    assert rc == pkg.capi.ERR_NO_DEVICE
    assert b"no CPU fallback" in lib.cpmpc_last_error() or b"not gfx950" in lib.cpmpc_last_error()
    dyn = pkg.capi.dbl_array([1.0, 0.1, 0.25, 9.81, 0.05, 0.1, 0.02, 0.8, 100.0], 9)
    one = C.c_void_p(8)  # never dereferenced: the device check comes first
    assert lib.cpmpc_rk4_batch(pkg.capi.F64, 1, dyn, one, one, 0.01, None, one, None, None,
                               None) == pkg.capi.ERR_NO_DEVICE
    assert lib.cpmpc_sim_step_batch(pkg.capi.F64, 1, dyn, 0.01, one, None, None, one,
                                    None) == pkg.capi.ERR_NO_DEVICE
    with pytest.raises(pkg.CpmpcError):
        pkg.capi.check(rc)


def test_long_horizons_are_accepted_and_refused_only_when_asked(lib, pkg, capfd):
    """The reference's constructor accepts any horizon (optimization.cc:13-22), so every constructor of the drop-in does
    (ADVICE r4): window_length * control_dt beyond cpmpc_max_parity_horizon() (1.0 s) gets as far as the device check from
    the positional entry points, with one warning per process on stderr; CPMPC_CREATE_STRICT_HORIZON through cpmpc_create_ex
    is CPMPC_ERR_UNSUPPORTED with the reason in the message.  The versioned structs reject a wrong struct_size / opts_size.
    (No device is needed for any of it.)"""
    assert lib.cpmpc_max_parity_horizon() == pytest.approx(1.0)
    assert lib.cpmpc_horizon_beyond_parity(None) == -1   # the per-handle status (round 6) of no handle
    assert lib.cpmpc_get_solver_opts(None, None, 0) == pkg.capi.ERR_INVALID_ARG
    assert lib.cpmpc_sharded_horizon_beyond_parity(None) == -1
    h = C.c_void_p()
    past = (pkg.capi.OK, pkg.capi.ERR_NO_DEVICE)
    for over in (dict(window_length=160), dict(window_length=120, state_spacing=12), dict(control_dt=0.05)):
        p = pkg.capi.default_params(**over)
        assert lib.cpmpc_create(C.byref(p), None, pkg.capi.F64, 8, 0, C.byref(h)) in past, over
        if h.value:
            lib.cpmpc_destroy(h)
        hs = C.c_void_p()
        assert lib.cpmpc_sharded_create(C.byref(p), None, pkg.capi.F64, 8, None, 0, C.byref(hs)) in past
        if hs.value:
            lib.cpmpc_sharded_destroy(hs)
    err = capfd.readouterr().err
    assert err.count("cpmpc_max_parity_horizon") <= 1   # said once per process (another test may have been first)
    p = pkg.capi.default_params(window_length=160)

    def info(**kw):
        i = pkg.capi.CreateInfo(struct_size=C.sizeof(pkg.capi.CreateInfo), flags=0, dtype=pkg.capi.F64, model=0, device=0,
                                reserved=0, max_batch=8, params=C.pointer(p), opts=None, opts_size=0)
        for k, v in kw.items():
            setattr(i, k, v)
        return i
    h = C.c_void_p()
    assert lib.cpmpc_create_ex(C.byref(info(flags=pkg.capi.CREATE_STRICT_HORIZON)), C.byref(h)) == pkg.capi.ERR_UNSUPPORTED
    assert b"CPMPC_CREATE_STRICT_HORIZON" in lib.cpmpc_last_error()
    for fl in (0, pkg.capi.CREATE_ALLOW_LONG_HORIZON):
        rc = lib.cpmpc_create_ex(C.byref(info(flags=fl)), C.byref(h))
        assert rc in past   # past the horizon check
        if h.value:
            lib.cpmpc_destroy(h)
    p = pkg.capi.default_params(window_length=100)   # at the bound: strict accepts
    assert lib.cpmpc_create_ex(C.byref(info(flags=pkg.capi.CREATE_STRICT_HORIZON)), C.byref(h)) in past
    if h.value:
        lib.cpmpc_destroy(h)
    assert lib.cpmpc_create_ex(C.byref(info(struct_size=12)), C.byref(h)) == pkg.capi.ERR_INVALID_ARG
    assert lib.cpmpc_create_ex(C.byref(info(flags=0x80)), C.byref(h)) == pkg.capi.ERR_INVALID_ARG
    # the pairs of flags that force one thing on and off exclude each other; each alone is accepted
    for pair in (pkg.capi.CREATE_REFINE_QP | pkg.capi.CREATE_NO_REFINE_QP, pkg.capi.CREATE_WIDE_QP | pkg.capi.CREATE_NO_WIDE_QP):
        assert lib.cpmpc_create_ex(C.byref(info(flags=pair)), C.byref(h)) == pkg.capi.ERR_INVALID_ARG
    for one in (pkg.capi.CREATE_REFINE_QP, pkg.capi.CREATE_NO_REFINE_QP, pkg.capi.CREATE_WIDE_QP, pkg.capi.CREATE_NO_WIDE_QP):
        rc = lib.cpmpc_create_ex(C.byref(info(flags=one)), C.byref(h))
        assert rc in past, one
        if h.value:
            lib.cpmpc_destroy(h)
    o = pkg.capi.default_solver_opts()
    assert lib.cpmpc_create_ex(C.byref(info(flags=1, opts=C.pointer(o), opts_size=C.sizeof(o) + 8)),
                               C.byref(h)) == pkg.capi.ERR_INVALID_ARG
    # a size that is nobody's struct: it splits a double, or is shorter than the shortest layout that is a prefix of today's
    # (8 + 13 doubles, through u_limit = 112 bytes; the 104 bytes of the very first commit had no ls_alpha_growth: ADVICE r4, r5)
    assert pkg.capi.SolverOpts.u_limit.offset + 8 == 112
    for bad in (4, 100, 104, C.sizeof(o) - 4):
        assert lib.cpmpc_create_ex(C.byref(info(opts=C.pointer(o), opts_size=bad)), C.byref(h)) == pkg.capi.ERR_INVALID_ARG, bad
    for good in (112, pkg.capi.SOLVER_OPTS_SIZE_POSITIONAL, C.sizeof(o)):
        rc = lib.cpmpc_create_ex(C.byref(info(opts=C.pointer(o), opts_size=good)), C.byref(h))
        assert rc in past, good
        if h.value:
            lib.cpmpc_destroy(h)


def test_positional_constructors_read_the_frozen_options_struct(lib, pkg):
    """cpmpc_create / cpmpc_create_model / cpmpc_sharded_create cannot be told the caller's sizeof(cpmpc_solver_opts): they
    read the struct as it was when they were frozen (through full_step_below, CPMPC_SOLVER_OPTS_SIZE_POSITIONAL bytes) and
    never the fields appended since -- a caller compiled against that header is not read past its struct.
    Here: a buffer of exactly that many bytes followed by a poisoned double (NaN would be rejected as exit_defect_floor if
    it were read) is accepted."""
    o = pkg.capi.default_solver_opts()
    n = pkg.capi.SOLVER_OPTS_SIZE_POSITIONAL
    assert n == pkg.capi.SolverOpts.exit_defect_floor.offset   # the first appended field starts where the frozen struct ends
    buf = (C.c_ubyte * (n + 8))()
    C.memmove(buf, C.byref(o), n)
    C.memmove(C.byref(buf, n), C.byref(C.c_double(float("nan"))), 8)
    p = pkg.capi.default_params()
    h = C.c_void_p()
    rc = lib.cpmpc_create(C.byref(p), C.cast(buf, C.POINTER(pkg.capi.SolverOpts)), pkg.capi.F64, 8, 0, C.byref(h))
    assert rc in (pkg.capi.OK, pkg.capi.ERR_NO_DEVICE), lib.cpmpc_last_error()
    if h.value:
        lib.cpmpc_destroy(h)
    # through cpmpc_create_ex with the full size the same poison IS read, and refused
    info = pkg.capi.CreateInfo(struct_size=C.sizeof(pkg.capi.CreateInfo), flags=0, dtype=pkg.capi.F64, model=0, device=0,
                               reserved=0, max_batch=8, params=C.pointer(p),
                               opts=C.cast(buf, C.POINTER(pkg.capi.SolverOpts)), opts_size=n + 8)
    assert lib.cpmpc_create_ex(C.byref(info), C.byref(h)) == pkg.capi.ERR_INVALID_ARG


def test_product_does_not_import_the_oracle():
    """The product package must never route through oracle/ (or any CPU fallback)."""
    pkg_dir = os.path.join(ROOT, "cart-pole-mpc_amd")
    for base, _, files in os.walk(pkg_dir):
        for name in files:
            if name.endswith((".py", ".hip", ".hpp", ".h", ".cc", ".cpp")):
                text = open(os.path.join(base, name), errors="replace").read()
                assert "oracle" not in text.lower().replace("no cpu fallback", ""), os.path.join(base, name)


def test_rule_thresholds_are_validated(lib, pkg):
    """full_step_below / exit_defect_floor: finite and >= 0 (checked before any device is touched)."""
    p = pkg.default_params()
    for field in ("full_step_below", "exit_defect_floor"):
        for bad in (-1.0, float("nan"), float("inf")):
            o = pkg.capi.default_solver_opts(**{field: bad})
            h = C.c_void_p()
            info = pkg.capi.CreateInfo(struct_size=C.sizeof(pkg.capi.CreateInfo), flags=0, dtype=pkg.capi.F64, model=0,
                                       device=0, reserved=0, max_batch=8, params=C.pointer(p), opts=C.pointer(o),
                                       opts_size=C.sizeof(o))
            assert lib.cpmpc_create_ex(C.byref(info), C.byref(h)) == pkg.capi.ERR_INVALID_ARG, (field, bad)
            if field == "full_step_below":   # inside the struct the positional constructors read
                assert lib.cpmpc_create(C.byref(p), C.byref(o), pkg.capi.F64, 8, 0, C.byref(h)) == pkg.capi.ERR_INVALID_ARG


def _plan(lib, hist, B=262144, T=8, intervals=4, dtype=1, window=40):
    h = (C.c_int64 * 16)(*([int(v) for v in hist] + [0] * (16 - len(hist))))
    out = (C.c_int32 * 32)()
    n = lib.cpmpc_plan_stages_from_histogram(h, B, T, intervals, dtype, window, out, 32)
    assert n >= 1
    return [int(out[i]) for i in range(n + 1)]


def _plan_cost(bounds, hist, B, T, ppw, resident, window):
    """The cost model of csrc/cpmpc_api.hip: plan_from_histogram, restated: time of a plan in wave-iterations."""
    tot = float(sum(hist))
    surv = [B * sum(hist[j + 1:]) / tot if j < T else 0.0 for j in range(T + 1)]
    surv[0] = float(B)
    launch, run_out = 0.14 * 40.0 / window, 2048.0 * ppw
    cost = 0.0
    for a, b in zip(bounds, bounds[1:]):
        if a > 0 and surv[a] <= run_out and b != T:
            return None            # not a plan the kernel would follow: such a stage runs to the end
        wi, longest = 0.0, 0.0
        for j in range(b - a):
            q = min(1.0, surv[a + j] / surv[a]) if surv[a] > 0 else 0.0
            wi += 1.0 - (1.0 - q) ** ppw
            if surv[a + j] >= 1.0:
                longest = j + 1.0
        cost += max(wi * surv[a] / ppw / resident, longest) + launch
    return cost


def test_stage_planner_known_cases_and_optimality(lib, pkg):
    """cpmpc_plan_stages_from_histogram (host only): the plans of the workloads measured in DESIGN 6.4, and on random
    histograms the plan is the cheapest partition of [0, T) under the documented cost model (checked by enumeration)."""
    import itertools
    import numpy as np
    F32, F64 = pkg.capi.F32, pkg.capi.F64
    B = 262144
    # settled closed loop: everybody stops after one iteration -> that launch and the insurance cut
    assert _plan(lib, [0, B], dtype=F64) == [0, 1, 8]
    assert _plan(lib, [0, 0, B], dtype=F32) == [0, 2, 8]
    # cold start, nearly everybody runs to the cap: nothing to gain from compaction until late, if at all
    cold = [0, 3, 53, 60, 476, 1494, 2221, 3053, 254784]
    p = _plan(lib, cold, dtype=F32)
    assert p[0] == 0 and p[-1] == 8 and p[1] >= 5
    # fp32 at the reference's tolerances (profiles/r04_steady_state_f32.json): cuts where the bulk leaves
    spread = [0, 282, 117494, 93908, 36758, 10807, 2322, 469, 104]
    assert _plan(lib, spread, dtype=F32) == [0, 2, 3, 8]
    # everybody at the cap: one launch, no insurance cut (nobody stops early)
    assert _plan(lib, [0] * 8 + [B], dtype=F64) == [0, 8]
    # a sample of the batch plans like the whole batch
    assert _plan(lib, [v // 4 for v in spread], dtype=F32) == [0, 2, 3, 8]
    # bad arguments
    out = (C.c_int32 * 32)()
    h = (C.c_int64 * 16)()
    assert lib.cpmpc_plan_stages_from_histogram(h, B, 8, 4, F64, 40, out, 32) == -1      # empty histogram
    h[1] = 5
    assert lib.cpmpc_plan_stages_from_histogram(h, B, 17, 4, F64, 40, out, 32) == -1     # more stages than kMaxStages
    assert lib.cpmpc_plan_stages_from_histogram(h, B, 8, 4, F64, 40, out, 4) == -1       # bounds too short
    rng = np.random.default_rng(3)
    for case in range(300):
        T = int(rng.integers(2, 9))
        intervals = int(rng.choice([2, 4, 5, 8, 10, 16]))
        dtype = F64 if case % 2 else F32
        hist = np.zeros(16, dtype=np.int64)
        kind = case % 3
        if kind == 0:
            hist[1:T + 1] = rng.integers(0, 100000, T)
        elif kind == 1:   # most stop early, a thin tail
            hist[1] = 250000
            hist[2:T + 1] = rng.integers(0, 200, T - 1)
        else:             # most run to the cap
            hist[T] = 250000
            hist[1:T] = rng.integers(0, 3000, T - 1)
        if hist.sum() == 0:
            hist[1] = 1
        Bc = int(rng.choice([40000, 262144, 1000000]))
        got = _plan(lib, hist, B=Bc, T=T, intervals=intervals, dtype=dtype)
        assert got[0] == 0 and got[-1] == T and all(b > a for a, b in zip(got, got[1:]))
        ppw, resident = 64 // intervals, (1024.0 if dtype == F64 else 2048.0)
        best = None
        for k in range(T):
            for cuts in itertools.combinations(range(1, T), k):
                c = _plan_cost([0, *cuts, T], list(hist), Bc, T, ppw, resident, 40)
                if c is not None and (best is None or c < best - 1e-9):
                    best = c
        c_got = _plan_cost(got, list(hist), Bc, T, ppw, resident, 40)
        if len(got) == 3 and c_got is not None and c_got > best + 1e-9:
            # the insurance cut: one launch was cheapest, the cut sits where 99.9 % have stopped
            assert abs(_plan_cost([0, T], list(hist), Bc, T, ppw, resident, 40) - best) < 1e-9, (case, got)
            tot = hist.sum()
            assert hist[got[1] + 1:].sum() <= 0.001 * tot + 1e-9 and (got[1] == 1 or hist[got[1]:].sum() > 0.001 * tot)
        else:
            assert c_got is not None and c_got <= best + 1e-9, (case, got, c_got, best)
