"""The single-precision twin of the CPU check (oracle/cpmpc_oracle_f32.c) -- CPU only.

The twin is the restatement of oracle/cpmpc_oracle.c compiled with float arithmetic (KKT solve in double, as the float
kernels keep their terminal system).  These tests pin it to the double build where the two must agree, and show that the
branch only single precision can take -- the first-order exit test's rounding floor -- is live in
it and dead in the double build."""
import numpy as np

from conftest import DYN_UI

NO_TOL = dict(max_iterations=5, relative_exit_tol=0.0, absolute_first_derivative_tol=0.0)


def _upright(rng, B, spread=0.3):
    return np.stack([rng.uniform(-0.2, 0.2, B), np.pi / 2 + rng.uniform(-spread, spread, B), rng.uniform(-0.4, 0.4, B),
                     rng.uniform(-0.8, 0.8, B)])


def test_twin_agrees_with_the_double_check_where_single_precision_can(orc):
    """Near-upright cold starts, five iterations, exits disabled: the float solve follows the double one to a few 1e-4 in
    the controls (newtons; the controls are of order 10) on every lane, median 1e-5 -- the size of the float rounding of a
    40-step rollout, not of an algorithmic difference; the first QP step alone agrees to 1e-3 relative."""
    rng = np.random.default_rng(5)
    x0 = _upright(rng, 128)
    p = orc.default_opt_params(**NO_TOL)
    u64, _, st64, it64, _ = orc.step_batch_cold(p, DYN_UI, 0.0, x0)
    u32, st32, it32, _, _ = orc.step_batch_cold_f32(p, DYN_UI, 0.0, x0)
    err = np.abs(u64 - u32).max(axis=0)
    scale = np.abs(u64).max(axis=0)
    assert (st32 == st64).all() and (it32 == it64).all()
    assert np.median(err) < 2e-4 and (err < 2e-2 * np.maximum(scale, 1.0)).all(), (np.median(err), err.max())
    assert (err > 0).all()   # it IS another arithmetic


def test_exit_floor_is_a_single_precision_rule(orc):
    """Settled controllers at the reference's tolerances (absolute_first_derivative_tol = 1e-6).
    Double build: exit_defect_floor = 0 and = 2 give bit-for-bit the same controls and counts (the rule is not applied in
    double: one exit rule in the parity dtype).  Float twin: with the floor most controllers leave after ONE iteration with
    SATISFIED_FIRST_ORDER_TOL; without it none can (mu |c|_1 of the float rollout's rounding exceeds 1e-6)."""
    rng = np.random.default_rng(6)
    B, ticks = 96, 260
    x0 = _upright(rng, B, spread=0.05)
    p = orc.default_opt_params()
    # double: the option changes nothing
    u_a, _, st_a, it_a, _ = orc.step_batch_cold(p, DYN_UI, 0.0, x0, opts=orc.default_solver_opts(exit_defect_floor=2.0))
    u_b, _, st_b, it_b, _ = orc.step_batch_cold(p, DYN_UI, 0.0, x0, opts=orc.default_solver_opts(exit_defect_floor=0.0))
    assert np.array_equal(u_a, u_b) and np.array_equal(st_a, st_b) and np.array_equal(it_a, it_b)
    # float: the branch is live
    st_f, it_f, xf, _ = orc.closed_loop_f32(p, DYN_UI, 0.0, x0, ticks, opts=orc.default_solver_opts(exit_defect_floor=2.0))
    st_0, it_0, x0f, _ = orc.closed_loop_f32(p, DYN_UI, 0.0, x0, ticks, opts=orc.default_solver_opts(exit_defect_floor=0.0))
    FIRST_ORDER = orc.TERM_SATISFIED_FIRST_ORDER_TOL
    last = slice(ticks - 20, ticks)
    with_floor, without = it_f[last].mean(), it_0[last].mean()
    assert with_floor < 2.0 < 2.4 < without, (with_floor, without)
    assert (st_f[last] == FIRST_ORDER).mean() > 0.5
    assert ((st_0[last] == FIRST_ORDER) & (it_0[last] == 1)).mean() < 0.05
    # no solver failure at any tick of either run (optimization_test.cc:44-46), poles upright either way
    for st in (st_f, st_0):
        assert not np.isin(st, [orc.TERM_QP_INDEFINITE, orc.TERM_MAX_LAMBDA, orc.TERM_NON_FINITE]).any()
    assert np.abs(xf[1] - np.pi / 2).max() < 2e-5 and np.abs(x0f[1] - np.pi / 2).max() < 2e-5


def test_twin_closed_loop_swings_up_without_solver_failures(orc):
    """The reference's closed-loop criterion (optimization_test.cc:39-66) on the float twin, from arbitrary pole angles:
    never QP_INDEFINITE / MAX_LAMBDA / NON_FINITE (the KKT solve in double is what makes that hold in float, as the
    terminal system in double does for the kernels), and after 3 s most poles stand."""
    rng = np.random.default_rng(8)
    B = 48
    x0 = np.stack([rng.uniform(-0.3, 0.3, B), rng.uniform(-np.pi, np.pi, B), rng.uniform(-0.5, 0.5, B), rng.uniform(-1, 1, B)])
    st, it, xf, _ = orc.closed_loop_f32(orc.default_opt_params(), DYN_UI, 0.0, x0, 300)
    assert not np.isin(st, [orc.TERM_QP_INDEFINITE, orc.TERM_MAX_LAMBDA, orc.TERM_NON_FINITE]).any()
    assert (np.abs(xf[1] - np.pi / 2) < 1e-3).mean() > 0.9
