"""The `pypendulum` module keeps the reference wrapper's names (wrapper/wrapper.cc:40-98) and the call
sequence of model/scratch.py:22-40 works unchanged -- on the GPU (there is no CPU solver behind it)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import DYN_TEST, ROOT

LIB_DIR = os.path.join(ROOT, "cart-pole-mpc_amd", "lib")


@pytest.fixture(scope="module")
def pp():
    import importlib
    importlib.import_module("cart-pole-mpc_amd.build").build_host()   # libcpmpc.so + facade + module (stale-checked)
    if LIB_DIR not in sys.path:
        sys.path.insert(0, LIB_DIR)
    import pypendulum
    return pypendulum


def test_module_surface(pp):
    """Classes, constructors and read/write attributes of wrapper.cc:40-98."""
    prm = pp.SingleCartPoleParams(1.0, 0.1, 0.25, 9.81, 0.05, 0.1, 0.02, 0.8, 100.0)
    kw = pp.SingleCartPoleParams(m_b=1.0, m_1=0.1, l_1=0.25, g=9.81, mu_b=0.05, v_mu_b=0.1, c_d_1=0.02, x_s=0.8,
                                 k_s=100.0)
    for name in ("m_b", "m_1", "l_1", "g", "mu_b", "v_mu_b", "c_d_1", "x_s", "k_s"):
        assert getattr(prm, name) == getattr(kw, name)
        setattr(prm, name, 2.0)
        assert getattr(prm, name) == 2.0
    pp.SingleCartPoleParams()
    op = pp.OptimizationParams()
    defaults = dict(control_dt=0.01, window_length=40, state_spacing=10, max_iterations=8, relative_exit_tol=1e-5,
                    absolute_first_derivative_tol=1e-6, equality_penalty_initial=1.0,
                    u_guess_sinusoid_amplitude=10.0, u_cost_weight=0.1, u_derivative_cost_weight=0.1,
                    b_x_final_cost_weight=150.0, th_final_cost_weight=-1.0, b_x_dot_final_cost_weight=-1.0,
                    th_dot_final_cost_weight=-1.0)
    for k, v in defaults.items():  # optimization.hpp:12-48
        assert getattr(op, k) == v
        setattr(op, k, type(v)(v))
    st = pp.SingleCartPoleState(0.1, 0.2, 0.3, 0.4)
    assert (st.b_x, st.th_1, st.b_x_dot, st.th_1_dot) == (0.1, 0.2, 0.3, 0.4)
    st.th_1 = 1.0
    pp.Vector2(1.0, 2.0)
    for cls, methods in ((pp.Optimization, ("step", "set_previous_solution")),
                         (pp.Simulator, ("step", "get_state")),
                         (pp.OptimizationOutputs, ("solver_summary", "u", "predicted_states"))):
        for mth in methods:
            assert hasattr(cls, mth), (cls, mth)


def test_constructor_preconditions_raise(pp):
    """optimization.cc:13-22 -> exceptions, as F_ASSERT in the reference."""
    for field, bad in (("control_dt", 0.0), ("state_spacing", 7), ("max_iterations", 0), ("u_cost_weight", -1.0)):
        op = pp.OptimizationParams()
        setattr(op, field, bad)
        with pytest.raises(ValueError):
            pp.Optimization(op)


def test_no_gpu_is_a_loud_error(pp, pkg):
    if pkg.capi.load().cpmpc_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError, match="no CPU fallback|gfx950"):
        pp.Optimization(pp.OptimizationParams())
    with pytest.raises(RuntimeError):
        pp.Simulator().step(pp.SingleCartPoleParams(*DYN_TEST), 0.01, 0.0, pp.Vector2(0, 0), pp.Vector2(0, 0))


def test_json_wire_format_of_the_structs(pp):
    """wasm.cc:19-28: every struct as a JSON object keyed by its field names; text as nlohmann::json::dump()
    prints it (keys sorted, no whitespace, integers as integers, shortest round-trip doubles with '.0' on
    integral values) -- for these magnitudes the same text as Python's compact sorted dump."""
    import json
    canon = lambda t: json.dumps(json.loads(t), sort_keys=True, separators=(",", ":"))
    assert pp.is_tracing_enabled() is False and pp.get_traces() == ""   # wasm.cc:121-140 without MINI_OPT_TRACING
    p = pp.get_default_optimization_params()           # wasm.cc:118-119 getDefaultOptimizationParams
    t = p.to_json()
    assert t == canon(t)
    assert json.loads(t) == dict(
        control_dt=0.01, window_length=40, state_spacing=10, max_iterations=8, relative_exit_tol=1e-5,
        absolute_first_derivative_tol=1e-6, equality_penalty_initial=1.0, u_guess_sinusoid_amplitude=10.0,
        u_cost_weight=0.1, u_derivative_cost_weight=0.1, b_x_final_cost_weight=150.0, th_final_cost_weight=-1.0,
        b_x_dot_final_cost_weight=-1.0, th_dot_final_cost_weight=-1.0)       # optimization.hpp:12-48 defaults
    assert '"window_length":40,' not in t and t.endswith('"window_length":40}')   # integers stay integers
    q = pp.OptimizationParams.from_json(json.dumps(json.loads(t), indent=2))         # any standard spelling parses
    assert q.to_json() == t
    d = pp.SingleCartPoleParams(*DYN_TEST)
    assert json.loads(d.to_json()) == dict(zip("m_b m_1 l_1 g mu_b v_mu_b c_d_1 x_s k_s".split(), DYN_TEST))
    assert pp.SingleCartPoleParams.from_json(d.to_json()).to_json() == d.to_json() == canon(d.to_json())
    rng = np.random.default_rng(5)
    for _ in range(300):
        v = (rng.standard_normal(4) * 10.0 ** rng.integers(-12, 13, 4)).tolist()
        v[rng.integers(0, 4)] = float(rng.integers(-1000, 1000))              # integral values print as 'd.0'
        st = pp.SingleCartPoleState(*v)
        t = st.to_json()
        assert t == canon(t), t
        back = pp.SingleCartPoleState.from_json(t)
        assert [back.b_x, back.th_1, back.b_x_dot, back.th_1_dot] == v         # shortest digits round-trip exactly
    assert set(json.loads(pp.SingleCartPoleState(0, 0, 0, 0).to_json())) == {"b_x", "th_1", "th_1_dot", "b_x_dot"}
    # layout edges of the number format: fixed for 1e-4 <= |v| < 1e15, exponent form outside, non-finite -> null
    edge = pp.SingleCartPoleState(1e-4, 99999e-9, 123456789012345.0, 1e15).to_json()
    assert edge == '{"b_x":0.0001,"b_x_dot":123456789012345.0,"th_1":9.9999e-05,"th_1_dot":1e+15}'
    assert pp.SingleCartPoleState(float("nan"), -0.0, float("inf"), 5e-324).to_json() == \
        '{"b_x":null,"b_x_dot":null,"th_1":-0.0,"th_1_dot":5e-324}'
    f = pp.Vector2.list_from_json('[{"x": 1, "y": 2.5}, {"y": -3e2, "x": 0}]')   # the f_external argument, wasm.cc:78-80
    assert [(a.x, a.y) for a in f] == [(1.0, 2.5), (0.0, -300.0)]
    for bad in ('{"b_x":1}', '{"b_x":1,"th_1":2,"b_x_dot":"3","th_1_dot":4}', '{"b_x":1,', "[1,2]", ""):
        with pytest.raises(ValueError):
            pp.SingleCartPoleState.from_json(bad)


@pytest.mark.gpu
def test_scratch_py_call_sequence(pp, orc):
    """model/scratch.py:22-40 verbatim, then compared with the oracle (BASELINE config 1 plumbing,
    here at N=40 as scratch.py sets it, and at N=20)."""
    for N in (40, 20):
        params = pp.SingleCartPoleParams(1.0, 0.1, 0.25, 9.81, 0.05, 0.1, 0.02, 0.8, 100.0)
        x0_initial = pp.SingleCartPoleState(0.0, 0.0, 0.0, 0.0)
        opt_params = pp.OptimizationParams()
        opt_params.max_iterations = 30
        opt_params.state_spacing = 10
        opt_params.window_length = N
        opt_params.absolute_first_derivative_tol = 1.0e-3
        opt_params.u_guess_sinusoid_amplitude = 10.0
        opt_params.u_cost_weight = 0.0
        opt_params.b_x_final_cost_weight = 5.0
        opt_params.th_final_cost_weight = -1.0
        opt_params.b_x_dot_final_cost_weight = 100.0
        opt_params.th_dot_final_cost_weight = 100.0
        opt = pp.Optimization(opt_params)
        outputs = opt.step(x0_initial, params, 0.0)
        assert isinstance(outputs.solver_summary(), str) and "termination" in outputs.solver_summary()
        assert len(outputs.u) == N and len(outputs.predicted_states) == N
        assert isinstance(outputs.u[0], float)
        th = [s.th_1 for s in outputs.predicted_states]
        assert all(-np.pi < a <= np.pi for a in th)
        o = orc.Optimization(orc.default_opt_params(
            max_iterations=30, state_spacing=10, window_length=N, absolute_first_derivative_tol=1e-3,
            u_cost_weight=0.0, b_x_final_cost_weight=5.0, b_x_dot_final_cost_weight=100.0,
            th_dot_final_cost_weight=100.0)).step([0, 0, 0, 0], [1.0, 0.1, 0.25, 9.81, 0.05, 0.1, 0.02, 0.8, 100.0], 0.0)
        assert outputs.termination_state == o.solver_outputs.termination_state
        np.testing.assert_allclose(outputs.u, o.u, rtol=0, atol=1e-5)
        np.testing.assert_allclose([[s.b_x, s.th_1, s.b_x_dot, s.th_1_dot] for s in outputs.predicted_states],
                                   o.predicted_states, rtol=0, atol=1e-5)
        # wasm.cc:46-65,86-105: the log of a step as JSON and the accessor names of the browser build
        import json
        text = outputs.to_json()
        j = json.loads(text)
        assert text == json.dumps(j, sort_keys=True, separators=(",", ":"))
        assert list(j) == ["initial_state", "predicted_states", "previous_solution", "solver_outputs", "u"]
        assert j["u"] == list(outputs.u) and j["previous_solution"] == list(outputs.previous_solution)
        assert len(j["predicted_states"]) == N and j["predicted_states"][3]["th_1"] == outputs.predicted_states[3].th_1
        assert j["initial_state"] == dict(b_x=0.0, th_1=0.0, b_x_dot=0.0, th_1_dot=0.0)
        assert j["solver_outputs"]["termination_state"] in outputs.solver_summary()
        back = pp.OptimizationOutputs.from_json(text)
        assert back.to_json() == text and back.termination_state == outputs.termination_state
        assert outputs.window_length() == N and outputs.get_control(1) == outputs.u[1]
        assert outputs.get_predicted_state(N - 1).b_x == outputs.predicted_states[N - 1].b_x
        assert outputs.get_log() == outputs.solver_summary()
        with pytest.raises(IndexError):
            outputs.get_control(N)
        # the batched API writes the same log for a lane
        import importlib
        import torch
        pkg = importlib.import_module("cart-pole-mpc_amd")
        bo = pkg.BatchOptimization(pkg.default_params(
            max_iterations=30, state_spacing=10, window_length=N, absolute_first_derivative_tol=1e-3,
            u_cost_weight=0.0, b_x_final_cost_weight=5.0, b_x_dot_final_cost_weight=100.0,
            th_dot_final_cost_weight=100.0), max_batch=3, dtype=torch.float64, device=0)
        x0b = torch.zeros(4, 3, dtype=torch.float64, device="cuda:0")
        bout = bo.step(x0b, [1.0, 0.1, 0.25, 9.81, 0.05, 0.1, 0.02, 0.8, 100.0], 0.0, want_stats=True)
        jb = json.loads(bout.lane_json(2, x0b, bo.get_solution(3)))
        assert list(jb) == list(j) and jb["solver_outputs"]["termination_state"] == j["solver_outputs"]["termination_state"]
        np.testing.assert_allclose(jb["u"], j["u"], rtol=0, atol=1e-9)
        # previous_solution is the solution of the step BEFORE (optimization.cc:84-85): empty on the first one
        assert j["previous_solution"] == []
        second = json.loads(opt.step(x0_initial, params, 0.0).to_json())
        np.testing.assert_allclose(second["previous_solution"], jb["previous_solution"], rtol=0, atol=1e-9)
        assert jb["predicted_states"][N - 1].keys() == j["predicted_states"][N - 1].keys()


@pytest.mark.gpu
def test_closed_loop_through_pypendulum(pp, orc):
    """20 ticks of Optimization.step + Simulator.step, against the oracle; set_previous_solution."""
    dyn = pp.SingleCartPoleParams(*DYN_TEST)
    op = pp.OptimizationParams()
    op.state_spacing = 5
    opt, sim = pp.Optimization(op), pp.Simulator()
    s0 = sim.get_state()
    assert (s0.b_x, s0.th_1, s0.b_x_dot, s0.th_1_dot) == (0.0, -np.pi / 2, 0.0, 0.0)  # simulator.hpp:28
    o_opt, o_sim = orc.Optimization(orc.default_opt_params(state_spacing=5)), orc.Simulator()
    for t in range(20):
        out = opt.step(sim.get_state(), dyn, 0.0)
        o = o_opt.step(o_sim.get_state(), DYN_TEST, 0.0)
        np.testing.assert_allclose(out.u, o.u, rtol=0, atol=1e-5)
        if t > 0:
            np.testing.assert_allclose(out.previous_solution, o.previous_solution, rtol=0, atol=1e-5)
        sim.step(dyn, 0.01, out.u[0], pp.Vector2(0.5, 0.0), pp.Vector2(0.0, 0.0))
        o_sim.step(DYN_TEST, 0.01, o.u[0], (0.5, 0.0), (0.0, 0.0))
    g = sim.get_state()
    np.testing.assert_allclose([g.b_x, g.th_1, g.b_x_dot, g.th_1_dot], o_sim.get_state(), rtol=0, atol=1e-6)
    opt2 = pp.Optimization(op)
    opt2.set_previous_solution(list(o.previous_solution))
    out2 = opt2.step(pp.SingleCartPoleState(*o.initial_state), dyn, 0.0)
    np.testing.assert_allclose(out2.u, o.u, rtol=0, atol=1e-5)
    with pytest.raises(ValueError):
        sim.step(dyn, -1.0, 0.0, pp.Vector2(0, 0), pp.Vector2(0, 0))  # simulator.cc:13
    with pytest.raises(ValueError):
        sim.step(dyn, 0.01, float("nan"), pp.Vector2(0, 0), pp.Vector2(0, 0))  # simulator.cc:14


@pytest.mark.gpu
def test_cpp_closed_loop_binary():
    """A C++ caller of this repo's pendulum::Optimization / Simulator (tests/host/facade_closed_loop.cc): swing-up and
    balance in closed loop with the configuration and acceptance numbers of optimization_test.cc:13-20,44-66."""
    r = subprocess.run([os.path.join(LIB_DIR, "host_smoke")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "OK closed loop" in r.stdout
