"""Oracle pinned against golden vectors and the reference's own integration tests, re-stated
(optimization/integration_test.cc).  CPU only."""
import numpy as np

from conftest import DYN_DERIV


def test_survey_known_answers(orc, survey_answers):
    k = survey_answers
    f, Jx, Ju = orc.dynamics(k["params"], k["x"], k["u"])
    np.testing.assert_allclose(f, k["f"], rtol=0, atol=2e-14)
    np.testing.assert_allclose(Jx[0], [0, 0, 1, 0], atol=0)
    np.testing.assert_allclose(Jx[1], [0, 0, 0, 1], atol=0)
    np.testing.assert_allclose(Jx[2], k["J_x_row2"], rtol=0, atol=5e-14)
    np.testing.assert_allclose(Jx[3], k["J_x_row3"], rtol=0, atol=5e-14)
    np.testing.assert_allclose(Ju, k["J_u"], rtol=0, atol=5e-15)
    xn, A, B = orc.rk4(k["params"], k["x"], k["u"], k["dt"])
    np.testing.assert_allclose(xn, k["rk4_x_new"], rtol=0, atol=1e-15)
    np.testing.assert_allclose(A[0], k["rk4_A_row0"], rtol=0, atol=1e-16)
    np.testing.assert_allclose(B, k["rk4_B"], rtol=0, atol=1e-16)
    for a, want in k["mod_pi"]:
        assert orc.mod_pi(a) == want


def test_sympy_golden_vectors(orc, golden_dynamics):
    assert len(golden_dynamics) >= 50
    for c in golden_dynamics:
        f, Jx, Ju = orc.dynamics(c["params"], c["x"], c["u"], c["f_base"], c["f_mass"])
        for got, want in ((f, c["f"]), (Jx, c["J_x"]), (Ju, c["J_u"])):
            want = np.asarray(want)
            scale = max(1.0, np.abs(want).max())
            assert np.abs(got - want).max() / scale < 1e-13, c["tag"]
        f2 = orc.dynamics(c["params"], c["x"], c["u"], c["f_base"], c["f_mass"], jacobians=False)
        assert np.array_equal(f, f2)


def test_mod_pi_range_and_cut(orc):
    rng = np.random.default_rng(1)
    for a in rng.uniform(-50, 50, 2000):
        m = orc.mod_pi(a)
        assert -np.pi < m <= np.pi
        assert abs(np.sin(m) - np.sin(a)) < 1e-12 and abs(np.cos(m) - np.cos(a)) < 1e-12
    assert orc.mod_pi(np.pi) == np.pi and orc.mod_pi(-np.pi) == np.pi  # (-pi, pi] convention
    assert orc.mod_pi(0.0) == 0.0


def _numerical_derivative(dx, func):
    """6th-order central difference, as integration_test.cc:10-19."""
    c1 = func(dx) - func(-dx)
    c2 = func(2 * dx) - func(-2 * dx)
    c3 = func(3 * dx) - func(-3 * dx)
    return (c1 * 45 - c2 * 9 + c3) / (60 * dx)


def _numerical_jacobian(x, func, h=0.01):
    x = np.asarray(x, dtype=float)
    y0 = func(x)
    J = np.zeros((y0.size, x.size))
    for j in range(x.size):
        def along(dx, j=j):
            d = np.zeros_like(x)
            d[j] = dx
            return func(x + d) - y0
        J[:, j] = _numerical_derivative(h, along)
    return J


def test_derivatives_like_reference(orc):
    """IntegrationTest.TestDerivatives (integration_test.cc:45-80): |analytic - numerical|_F < 1e-12."""
    x = np.array([1.2, 0.7, 0.4, -0.15])
    u, dt = 0.1, 0.01
    _, A, B = orc.rk4(DYN_DERIV, x, u, dt)
    A_num = _numerical_jacobian(x, lambda xp: orc.rk4_no_jacobians(DYN_DERIV, xp, u, dt))
    B_num = _numerical_jacobian(np.array([u]), lambda up: orc.rk4_no_jacobians(DYN_DERIV, x, up[0], dt))
    assert np.linalg.norm(A - A_num) < 1.0e-12
    assert np.linalg.norm(B - B_num[:, 0]) < 1.0e-12


def test_derivatives_all_branches(orc, golden_dynamics):
    """Same check on every golden case that is not sitting on a branch point."""
    for c in golden_dynamics[:24]:
        x = np.array(c["x"])
        if min(abs(abs(x[0]) - c["params"][7]), 1.0) < 0.05:
            continue
        _, A, B = orc.rk4(c["params"], x, c["u"], 0.01)
        A_num = _numerical_jacobian(x, lambda xp: orc.rk4_no_jacobians(c["params"], xp, c["u"], 0.01), h=0.002)
        B_num = _numerical_jacobian(np.array([c["u"]]),
                                    lambda up: orc.rk4_no_jacobians(c["params"], x, up[0], 0.01))
        assert np.linalg.norm(A - A_num) < 5e-11, c["tag"]
        assert np.linalg.norm(B - B_num[:, 0]) < 1e-11, c["tag"]


def test_friction_dissipation(orc):
    """TestFrictionDissipation (integration_test.cc:82-103)."""
    p = [1.0, 0.5, 0.4, 9.81, 0.5, 0.1, 0.0, 0.0, 0.0]
    x = np.array([0.0, 0.0, 0.0, 0.0])
    for _ in range(20000):
        x = orc.rk4_no_jacobians(p, x, 0.0, 0.01)
    assert abs(x[2]) < 1.0e-6 and abs(x[3]) < 1.0e-4


def test_drag_dissipation(orc):
    """TestDragDissipation (integration_test.cc:105-125)."""
    p = [0.8, 0.1, 0.4, 9.81, 0.01, 0.1, 5.0, 0.0, 0.0]
    x = np.array([0.0, -np.pi, 0.0, 0.0])
    for _ in range(10000):
        x = orc.rk4_no_jacobians(p, x, 0.0, 0.01)
    assert abs(x[2]) < 1.0e-6 and abs(x[3]) < 3.0e-5


def test_external_force_symmetry(orc):
    """TestExternalForceSymmetry (integration_test.cc:127-175)."""
    p = [1.0, 0.1, 0.25, 9.81, 0.1, 0.1, 0.0, 0.0, 0.0]
    x0 = np.array([0.0, -np.pi / 2, 0.0, 0.0])
    finals = []
    for sign in (+1.0, -1.0):
        x = x0.copy()
        for i in range(3000):
            fb = [sign * 5.0, 0.0] if i < 500 else [0.0, 0.0]
            x = orc.rk4_no_jacobians(p, x, 0.0, 0.001, f_base=fb)
        finals.append(x)
    left, right = finals
    assert left[0] > 0 and right[0] < 0
    assert abs(left[0] + right[0]) < 1e-12
    assert abs(left[2] + right[2]) < 1e-12
    assert abs((-np.pi / 2 - left[1]) - (right[1] + np.pi / 2)) < 1e-12
    assert abs(left[3] + right[3]) < 1e-12


def test_simulator_substeps(orc):
    """Simulator::Step (simulator.cc:11-36): dt=0.01 is exactly 10 RK4 sub-steps of 1 ms with the
    angle wrapped after each; initial state (0, -pi/2, 0, 0) (simulator.hpp:28)."""
    sim = orc.Simulator()
    np.testing.assert_array_equal(sim.get_state(), [0.0, -np.pi / 2, 0.0, 0.0])
    p = DYN_DERIV
    sim.set_state([0.1, 3.1, 0.2, 4.0])
    sim.step(p, 0.01, 2.5, (1.0, 0.0), (0.5, -0.5))
    x = np.array([0.1, 3.1, 0.2, 4.0])
    for _ in range(10):
        x = orc.rk4_no_jacobians(p, x, 2.5, 0.001, f_base=(1.0, 0.0), f_mass=(0.5, -0.5))
        x[1] = orc.mod_pi(x[1])
    np.testing.assert_allclose(sim.get_state(), x, rtol=0, atol=1e-15)
    # a dt that is not a multiple of 1 ms ends with a short sub-step
    sim.set_state([0, 0, 0, 0])
    sim.step(p, 0.0025, 1.0)
    x = np.zeros(4)
    for h in (0.001, 0.001, 0.0005):
        x = orc.rk4_no_jacobians(p, x, 1.0, h)
        x[1] = orc.mod_pi(x[1])
    np.testing.assert_allclose(sim.get_state(), x, rtol=0, atol=1e-15)
