"""Oracle: problem assembly (BuildProblem), shooting constraint, QP, retraction, Step bookkeeping.
Follows optimization/optimization.cc.  CPU only."""
import numpy as np
import pytest

from conftest import DYN_UI, random_states


def _guess(orc, p, dyn, state):
    opt = orc.Optimization(p)
    out = opt.step(state, dyn, 0.0)
    return out.guess


def test_shapes_and_layout(orc):
    """dim = 4S+N (optimization.cc:204-205); 23 equality rows, 81 cost rows at the defaults
    (SURVEY.md section 8 a6); NumStates (optimization.hpp:52)."""
    p = orc.default_opt_params()
    assert p.num_states() == 5
    assert orc.problem_shape(p) == (60, 23, 81)
    p = orc.default_opt_params(state_spacing=5)
    assert p.num_states() == 9
    assert orc.problem_shape(p) == (76, 4 * 8 + 4 + 3, 81)
    # every terminal weight >= 0 -> all four are costs; no control costs
    p = orc.default_opt_params(th_final_cost_weight=1.0, b_x_dot_final_cost_weight=0.0,
                               th_dot_final_cost_weight=2.0, u_cost_weight=0.0,
                               u_derivative_cost_weight=0.0)
    assert orc.problem_shape(p) == (60, 20, 4)
    # scratch.py's configuration (model/scratch.py:26-36)
    p = orc.default_opt_params(u_cost_weight=0.0, b_x_final_cost_weight=5.0,
                               b_x_dot_final_cost_weight=100.0, th_dot_final_cost_weight=100.0)
    assert orc.problem_shape(p) == (60, 16 + 4 + 1, 3 + 40)


def test_constructor_preconditions(orc):
    """optimization.cc:13-22."""
    for bad in (dict(control_dt=0.0), dict(window_length=0), dict(state_spacing=7),
                dict(max_iterations=0), dict(u_cost_weight=-1.0), dict(u_derivative_cost_weight=-0.5)):
        with pytest.raises(ValueError):
            orc.Optimization(orc.default_opt_params(**bad))


def test_residual_values_and_rows(orc):
    p = orc.default_opt_params()
    S, N = 5, 40
    rng = np.random.default_rng(3)
    z = rng.normal(size=60)
    x_cur = rng.normal(size=4)
    r, c, J, A = orc.problem_eval(p, DYN_UI, x_cur, 0.3, 1.5, z)
    # initial-state rows (weight 1, angle wrapped): optimization.cc:228-232
    d = z[:4] - x_cur
    d[1] = orc.mod_pi(d[1])
    np.testing.assert_allclose(c[16:20], d, atol=1e-15)
    assert np.array_equal(A[16:20, :4], np.eye(4))
    # terminal: b_x is a cost with weight 150 against the set-point, the rest equalities
    assert r[0] == pytest.approx(150.0 * (z[16] - 0.3))
    assert J[0, 16] == 150.0 and np.count_nonzero(J[0]) == 1
    np.testing.assert_allclose(c[20:23], [orc.mod_pi(z[17] - np.pi / 2), z[18], z[19]], atol=1e-15)
    # du rows then (u0 - u_prev) then u rows: optimization.cc:270-301
    u = z[20:]
    np.testing.assert_allclose(r[1:40], 0.1 * (u[:-1] - u[1:]), atol=1e-15)
    assert r[40] == pytest.approx(0.1 * (u[0] - 1.5))
    np.testing.assert_allclose(r[41:81], 0.1 * u, atol=1e-15)
    assert J[1, 20] == 0.1 and J[1, 21] == -0.1
    # shooting rows reference [x_s, x_{s+1}, u block]
    for s in range(S - 1):
        blk = A[4 * s:4 * s + 4]
        assert np.array_equal(blk[:, 4 * (s + 1):4 * (s + 1) + 4], -np.eye(4))
        used = np.zeros(60, bool)
        used[4 * s:4 * s + 8] = True
        used[4 * S + 10 * s:4 * S + 10 * (s + 1)] = True
        assert np.all(blk[:, ~used] == 0)


def test_shooting_jacobian_finite_differences(orc):
    """Pins the chain rule of optimization.cc:145-154 by the 6th-order stencil of
    integration_test.cc:10-42 applied to the Jacobian-free path (optimization.cc:130-137)."""
    from test_oracle_dynamics import _numerical_jacobian
    rng = np.random.default_rng(5)
    for sp in (5, 10):
        vars_ = np.concatenate([[0.2, 1.0, -0.6, 1.5], [0.25, 1.1, -0.5, 1.0], rng.uniform(-15, 15, sp)])
        err, J = orc.shooting_constraint(DYN_UI, sp, 0.01, vars_)
        err2 = orc.shooting_constraint(DYN_UI, sp, 0.01, vars_, jacobian=False)
        np.testing.assert_allclose(err, err2, atol=1e-14)
        Jn = _numerical_jacobian(vars_, lambda v: orc.shooting_constraint(DYN_UI, sp, 0.01, v, jacobian=False),
                                 h=0.002)
        assert np.abs(J - Jn).max() < 1e-9


def test_shooting_wraps_only_at_interval_end(orc):
    """optimization.cc:139,156-157: angle wrapped once at the end, error angle wrapped."""
    x = np.array([0.0, 3.1, 0.0, 6.0])  # crosses +pi within the interval
    u = np.zeros(10)
    xe = x.copy()
    for k in range(10):
        xe = orc.rk4_no_jacobians(DYN_UI, xe, u[k], 0.01)
    assert xe[1] > np.pi
    vars_ = np.concatenate([x, [0.0, -3.0, 0.0, 0.0], u])
    err = orc.shooting_constraint(DYN_UI, 10, 0.01, vars_, jacobian=False)
    assert err[1] == pytest.approx(orc.mod_pi(orc.mod_pi(xe[1]) + 3.0), abs=1e-14)
    assert -np.pi < err[1] <= np.pi


def test_qp_solution_satisfies_kkt(orc):
    p = orc.default_opt_params()
    rng = np.random.default_rng(11)
    x0 = random_states(rng, 1)[:, 0]
    z = _guess(orc, p, DYN_UI, x0)
    r, c, J, A = orc.problem_eval(p, DYN_UI, x0, 0.0, 0.0, z)
    for lam in (0.0, 0.3):
        rc, dz = orc.qp_solve(J, r, A, c, 40, lam)
        assert rc == 0
        assert np.abs(A @ dz + c).max() < 1e-9          # linearised constraints hold
        G = J.T @ J
        G[20:, 20:] += lam * np.eye(40)
        grad = G @ dz + J.T @ r
        # stationarity on the null space of A
        _, _, Vt = np.linalg.svd(A)
        Z = Vt[A.shape[0]:].T
        assert np.abs(Z.T @ grad).max() < 1e-7 * max(1.0, np.abs(grad).max())


def test_retraction(orc):
    """optimization.cc:309-329."""
    p, o = orc.default_opt_params(), orc.default_solver_opts()
    z = np.zeros(60)
    dz = np.zeros(60)
    z[1], dz[1] = 3.0, 1.0        # angle wraps
    z[4], dz[4] = 4.0, 4.0        # b_x clamps at +5
    z[8], dz[8] = -4.0, -4.0      # b_x clamps at -5
    z[20], dz[20] = 250.0, 200.0  # u clamps at +300
    z[21], dz[21] = -250.0, -200.0
    z[22], dz[22] = 1.0, 2.0
    out = orc.retract(p, o, z, dz, 0.5)
    assert out[1] == pytest.approx(orc.mod_pi(3.5))
    assert out[4] == 5.0 and out[8] == -5.0
    assert out[20] == 300.0 and out[21] == -300.0 and out[22] == 2.0


def test_cold_start_guess(orc):
    """optimization.cc:58-71,333-351."""
    p = orc.default_opt_params()
    x0 = np.array([0.1, -1.0, 0.2, 0.5])
    g = _guess(orc, p, DYN_UI, x0)
    N, S = 40, 5
    np.testing.assert_array_equal(g[:4], x0)
    want_u = 10.0 * np.sin(np.arange(N) / N * 2 * np.pi)
    np.testing.assert_allclose(g[4 * S:], want_u, atol=1e-15)
    x = x0.copy()
    for s in range(1, S):
        for k in range(10):
            x = orc.rk4_no_jacobians(DYN_UI, x, want_u[(s - 1) * 10 + k], 0.01)
            x[1] = orc.mod_pi(x[1])
        np.testing.assert_allclose(g[4 * s:4 * s + 4], x, atol=1e-14)


def test_warm_start_shift_and_u_prev(orc):
    """optimization.cc:50-57 (shift left, last duplicated, x0 overwritten) and :288-291 (u_prev is the
    previous solution's u_0, read before it is overwritten)."""
    p = orc.default_opt_params(max_iterations=3)
    opt = orc.Optimization(p)
    x0 = np.array([0.05, 1.2, 0.0, 0.0])
    out1 = opt.step(x0, DYN_UI, 0.0)
    x1 = np.array([0.06, 1.25, 0.1, 0.2])
    out2 = opt.step(x1, DYN_UI, 0.0)
    S, N = 5, 40
    np.testing.assert_array_equal(out2.guess[:4], x1)
    np.testing.assert_array_equal(out2.guess[4 * S:4 * S + N - 1], out1.z[4 * S + 1:])
    assert out2.guess[-1] == out1.z[-1]
    np.testing.assert_array_equal(out2.previous_solution, out1.z)
    # the same second solve done by hand with u_prev = out1.u[0]
    z, s = orc.solve(p, DYN_UI, x1, 0.0, out1.u[0], out2.guess)
    np.testing.assert_array_equal(z, out2.z)
    z_wrong, _ = orc.solve(p, DYN_UI, x1, 0.0, 0.0, out2.guess)
    assert np.abs(z_wrong - out2.z).max() > 1e-9
    # reset -> cold start again (optimization.hpp:83)
    opt.reset()
    out3 = opt.step(x0, DYN_UI, 0.0)
    np.testing.assert_array_equal(out3.z, out1.z)
    # set_previous_solution (optimization.hpp:86-89)
    opt.set_previous_solution(out1.z)
    out4 = opt.step(x1, DYN_UI, 0.0)
    np.testing.assert_array_equal(out4.z, out2.z)


def test_predicted_states_and_outputs(orc):
    """optimization.cc:85-96,353-371."""
    p = orc.default_opt_params()
    x0 = np.array([0.0, 1.0, 0.0, 0.0])
    out = orc.Optimization(p).step(x0, DYN_UI, 0.0)
    np.testing.assert_array_equal(out.u, out.z[20:])
    x = x0.copy()
    for k in range(40):
        x = orc.rk4_no_jacobians(DYN_UI, x, out.u[k], 0.01)
        x[1] = orc.mod_pi(x[1])
        np.testing.assert_allclose(out.predicted_states[k], x, atol=1e-14)
    assert out.predicted_states.shape == (40, 4)


def test_solution_is_a_kkt_point_when_converged(orc):
    """From a state near upright the SQP converges; check feasibility and first-order optimality of
    the returned point directly on the problem functions."""
    p = orc.default_opt_params(max_iterations=30, relative_exit_tol=0.0,
                               absolute_first_derivative_tol=1e-10)
    x0 = np.array([0.05, np.pi / 2 - 0.2, 0.0, 0.1])
    out = orc.Optimization(p).step(x0, DYN_UI, 0.0)
    r, c, J, A = orc.problem_eval(p, DYN_UI, x0, 0.0, 0.0, out.z)
    assert np.abs(c).max() < 1e-8
    grad = J.T @ r
    _, _, Vt = np.linalg.svd(A)
    Z = Vt[A.shape[0]:].T
    assert np.abs(Z.T @ grad).max() < 1e-5
    assert out.solver_outputs.termination_state in (orc.TERM["SATISFIED_FIRST_ORDER_TOL"],
                                                    orc.TERM["SATISFIED_RELATIVE_TOL"])


def test_batch_driver_matches_single(orc):
    p = orc.default_opt_params(max_iterations=4)
    rng = np.random.default_rng(2)
    x0 = random_states(rng, 6)
    u, pred, st, it, _ = orc.step_batch_cold(p, DYN_UI, 0.0, x0, want_pred=True, num_threads=2)
    for b in range(6):
        out = orc.Optimization(p).step(x0[:, b], DYN_UI, 0.0)
        np.testing.assert_array_equal(u[:, b], out.u)
        np.testing.assert_array_equal(pred[:, :, b], out.predicted_states)
        assert st[b] == out.solver_outputs.termination_state and it[b] == out.solver_outputs.iterations
