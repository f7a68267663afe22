"""OptimizationTest.TestCartPoleMultipleShootingClosedLoop (optimization/optimization_test.cc:12-77)
re-stated on the oracle.  This is the only thing in the reference that pins the solver (mini_opt is
absent): convergence properties, not values."""
import numpy as np

from conftest import DYN_TEST, DYN_UI


def test_cart_pole_multiple_shooting_closed_loop(orc):
    num_steps = 200
    p = orc.default_opt_params(control_dt=0.01, window_length=40, state_spacing=5, max_iterations=10)
    sim = orc.Simulator()
    sim.set_state([0.0, -np.pi / 2, 0.0, 0.0])
    opt = orc.Optimization(p)
    states = [sim.get_state()]
    for t in range(num_steps):
        out = opt.step(sim.get_state(), DYN_TEST, 0.0)
        term = out.solver_outputs.termination_state
        assert term != orc.TERM["QP_INDEFINITE"]       # optimization_test.cc:44-45
        assert term != orc.TERM["MAX_LAMBDA"]          # optimization_test.cc:46
        terminal = out.predicted_states[-1]
        if t > 20:                                     # optimization_test.cc:51-55
            assert abs(terminal[2]) < 1.0e-4
            assert abs(terminal[3]) < 1.0e-4
            assert abs(terminal[1] - np.pi / 2) < 1.0e-4
        states.append(sim.get_state())
        sim.step(DYN_TEST, p.control_dt, out.u[0], (0, 0), (0, 0))
    final = states[-1]                                 # optimization_test.cc:63-66
    assert abs(final[2]) < 1.0e-4
    assert abs(final[3]) < 1.0e-3
    assert abs(final[1] - np.pi / 2) < 1.0e-4


def test_default_configuration_balances(orc):
    """The UI's configuration (defaults of optimization.hpp, params of application.ts:61-71)."""
    p = orc.default_opt_params()
    sim = orc.Simulator()
    opt = orc.Optimization(p)
    for _ in range(300):
        out = opt.step(sim.get_state(), DYN_UI, 0.0)
        sim.step(DYN_UI, 0.01, out.u[0])
    s = sim.get_state()
    assert abs(s[1] - np.pi / 2) < 1e-4 and abs(s[0]) < 1e-3 and abs(s[2]) < 1e-3 and abs(s[3]) < 1e-3


def test_a_converged_step_is_taken_without_a_merit_evaluation(orc):
    """DESIGN.md section 4 (round 4): once a controller has settled, the first-order test holds and the undamped QP step is
    tiny, so the step is taken in full and the iteration ends without evaluating the merit at the new point -- one
    iteration, no line-search evaluation, and the reported cost / residual are those of the iterate the step was computed
    from (the initial ones of that solve).  With full_step_below = 0 (no tiny rule) the same tick costs one evaluation."""
    p = orc.default_opt_params()
    for opts, evals in ((None, 0), (orc.default_solver_opts(full_step_below=0.0), 1)):
        sim = orc.Simulator()
        opt = orc.Optimization(p) if opts is None else orc.Optimization(p, opts)
        for _ in range(320):
            out = opt.step(sim.get_state(), DYN_UI, 0.0)
            sim.step(DYN_UI, 0.01, out.u[0])
        so = out.solver_outputs
        assert so.termination_state == orc.TERM["SATISFIED_FIRST_ORDER_TOL"] and so.iterations == 1
        assert so.line_search_evals == evals
        if evals == 0:
            assert so.final_cost == so.initial_cost and so.final_eq_l1 == so.initial_eq_l1
