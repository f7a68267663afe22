"""Oracle, second model: cart + double pendulum (BASELINE config 5, symbolic/dynamics_double.py).
No executable reference exists for it (optimization.cc:197-199), so it is pinned by an independent
SymPy/mpmath closed form, finite differences, energy conservation and closed-loop behaviour only.  CPU."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

DYN_DOUBLE = [1.0, 0.1, 0.1, 0.25, 0.2, 9.81]   # m_b, m_1, m_2, l_1, l_2, g
DOUBLE_OVER = dict(u_guess_sinusoid_amplitude=0.0)


@pytest.fixture(scope="module")
def golden_double():
    with open(os.path.join(GOLDEN, "double_golden.json")) as fh:
        return json.load(fh)["cases"]


def test_model_dimensions(orc):
    assert orc.model_nx("single") == 4 and orc.model_np("single") == 9
    assert orc.model_nx("double") == 6 and orc.model_np("double") == 6
    p = orc.default_opt_params()
    # MapKey<6>: dim = 6S + N; equality rows 6(S-1) + 6 + 5 (all terminal rows but b_x), cost rows as before
    assert orc.problem_shape_model("double", p) == (6 * 5 + 40, 6 * 4 + 6 + 5, 81)
    assert orc.problem_shape_model("single", p) == orc.problem_shape(p)


def test_double_dynamics_golden(orc, golden_double):
    assert len(golden_double) == 40
    for c in golden_double:
        f, Jx, Ju = orc.dynamics_double(c["params"], c["x"], c["u"])
        for got, want in ((f, c["f"]), (Jx, c["J_x"]), (Ju, c["J_u"])):
            want = np.asarray(want)
            assert np.abs(got - want).max() / max(1.0, np.abs(want).max()) < 1e-12
    # the upright equilibrium is a fixed point
    f = orc.dynamics_double(DYN_DOUBLE, [0.3, np.pi / 2, np.pi / 2, 0, 0, 0], 0.0, jacobians=False)
    assert np.abs(f).max() < 1e-14


def test_double_rk4_derivatives(orc):
    """The reference's derivative test (integration_test.cc:45-80) applied to the 6-state model."""
    from test_oracle_dynamics import _numerical_jacobian
    x = np.array([0.2, 0.7, -0.4, 0.4, -0.15, 0.3])
    u, dt = 0.1, 0.01
    _, A, B = orc.rk4_model("double", DYN_DOUBLE, x, u, dt)
    A_num = _numerical_jacobian(x, lambda xp: orc.rk4_model("double", DYN_DOUBLE, xp, u, dt, jacobians=False))
    B_num = _numerical_jacobian(np.array([u]),
                                lambda up: orc.rk4_model("double", DYN_DOUBLE, x, up[0], dt, jacobians=False))
    assert np.linalg.norm(A - A_num) < 1e-11
    assert np.linalg.norm(B - B_num[:, 0]) < 1e-12


def test_double_energy_conservation(orc):
    """No dissipation in dynamics_double.py: with u = 0 the RK4 flow conserves T + V to O(h^4)."""
    x = np.array([0.0, 1.0, 2.0, 0.0, 0.0, 0.0])
    e0 = orc.energy_double(DYN_DOUBLE, x)
    for _ in range(3000):
        x = orc.rk4_model("double", DYN_DOUBLE, x, 0.0, 0.001, jacobians=False)
    assert abs(orc.energy_double(DYN_DOUBLE, x) - e0) < 1e-7 * max(1.0, abs(e0))


def test_double_shooting_jacobian(orc):
    from test_oracle_dynamics import _numerical_jacobian
    rng = np.random.default_rng(4)
    sp = 5
    vars_ = np.concatenate([[0.1, 1.2, 1.9, -0.3, 0.5, -0.4], [0.12, 1.25, 1.8, -0.2, 0.4, -0.1],
                            rng.uniform(-5, 5, sp)])
    err, J = orc.shooting_constraint_model("double", DYN_DOUBLE, sp, 0.01, vars_)
    Jn = _numerical_jacobian(vars_, lambda v: orc.shooting_constraint_model("double", DYN_DOUBLE, sp, 0.01, v,
                                                                            jacobian=False), h=0.002)
    assert np.abs(J - Jn).max() < 1e-8
    assert np.array_equal(J[:, 6:12], -np.eye(6))


def test_double_closed_loop_balances(orc):
    """Both poles stay upright under MPC for 3 s from a perturbed state (zero cold-start amplitude)."""
    p = orc.default_opt_params(max_iterations=10, **DOUBLE_OVER)
    opt = orc.Optimization(p, model="double")
    st = np.array([0.0, np.pi / 2 + 0.05, np.pi / 2 - 0.03, 0.0, 0.0, 0.0])
    for _ in range(300):
        out = opt.step(st, DYN_DOUBLE, 0.0)
        assert out.solver_outputs.termination_state not in (orc.TERM["QP_INDEFINITE"], orc.TERM["MAX_LAMBDA"])
        st = orc.sim_step_model("double", DYN_DOUBLE, 0.01, out.u[0], st)
    assert abs(st[1] - np.pi / 2) < 1e-4 and abs(st[2] - np.pi / 2) < 1e-4
    assert np.abs(st[3:]).max() < 1e-3
    assert out.predicted_states.shape == (40, 6) and out.z.shape == (70,)
