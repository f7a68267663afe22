"""GPU parity tests: the HIP path, called through the C-ABI (libcpmpc.so), against the CPU oracle on
the same seeded inputs, against the committed golden fixtures, and -- at BASELINE.json's full sizes
-- through size-independent properties.

Tolerances: fp64 control sequences within 1e-5 of the oracle (BASELINE.json north_star); building
blocks to ~1e-12.  fp32 is reported and loosely bounded (the reference is fp64-only)."""
import numpy as np
import pytest

from conftest import DYN_DERIV, DYN_TEST, DYN_UI, random_states

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
DEV = "cuda:0"
NO_TOL = dict(max_iterations=5, relative_exit_tol=0.0, absolute_first_derivative_tol=0.0)


@pytest.fixture(scope="module", autouse=True)
def _gpu(pkg):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: the product has no CPU fallback")
    pkg.capi.load()
    assert pkg.capi.load().cpmpc_device_count() >= 1


def T(a, dtype=torch.float64):
    return torch.tensor(np.ascontiguousarray(a), dtype=dtype, device=DEV)


def N_(t):
    return t.detach().cpu().numpy()


# ------------------------------------------------------------------------------------------------
# a1-a4: dynamics, RK4, mod_pi
# ------------------------------------------------------------------------------------------------
def test_dynamics_golden_vectors(pkg, golden_dynamics):
    """Every golden case (independent SymPy derivation) through cpmpc_dynamics_batch, fp64."""
    for c in golden_dynamics:
        x = T(np.array(c["x"]).reshape(4, 1))
        u = T([c["u"]])
        f, Jx, Ju = pkg.dynamics_batch(c["params"], x, u, fext=c["f_base"] + c["f_mass"])
        for got, want in ((N_(f)[:, 0], c["f"]), (N_(Jx)[:, :, 0], c["J_x"]), (N_(Ju)[:, 0], c["J_u"])):
            want = np.asarray(want)
            assert np.abs(got - want).max() / max(1.0, np.abs(want).max()) < 1e-12, c["tag"]


def test_fp64_math_routines_over_wide_ranges(pkg, orc):
    """The fp64 kernels use their own bounded-range sincos / tanh (cartpole_device.hpp); the oracle uses the C
    library.  Angles from 1e-6 rad to hundreds of turns and exactly at multiples of pi/2 (worst case of the argument
    reduction), friction arguments from 1e-11 to 1e4: f and J agree to a few ulp, scaled by the size of the terms."""
    rng = np.random.default_rng(3)
    B = 4096
    x = np.zeros((4, B))
    x[0] = rng.uniform(-1.2, 1.2, B)
    x[1] = rng.uniform(-np.pi, np.pi, B) * 10.0 ** rng.integers(-6, 3, B)
    x[2] = rng.standard_normal(B) * 10.0 ** rng.integers(-12, 3, B)
    x[3] = rng.standard_normal(B) * 10.0 ** rng.integers(-8, 2, B)
    x[1, :512] = np.round(rng.uniform(-50, 50, 512)) * (np.pi / 2)
    x[2, 512:520] = [0.0, -0.0, 1e-300, -1e-300, 4.0, -4.0, 1e3, -1e3]
    u = rng.uniform(-50, 50, B)
    f, Jx, Ju = pkg.dynamics_batch(DYN_UI, T(x), T(u))
    f, Jx = N_(f), N_(Jx)
    for b in range(B):
        fo, Jo, _ = orc.dynamics(DYN_UI, x[:, b], u[b])
        scale = max(1.0, np.abs(fo).max())
        assert np.abs(f[:, b] - fo).max() <= 2e-13 * scale, (b, x[:, b])
        assert np.abs(Jx[:, :, b] - Jo).max() <= 2e-13 * max(1.0, np.abs(Jo).max()), (b, x[:, b])
    # odd symmetry of the friction term is exact: f(-v) mirrored
    xm = x.copy()
    xm[1] = 0.3
    xm[3] = 0.0
    xm[0] = 0.0
    fa, _, _ = pkg.dynamics_batch([1.0, 0.1, 0.25, 0.0, 0.05, 0.1, 0.0, 0.8, 100.0], T(xm), T(np.zeros(B)))
    xm[2] = -xm[2]
    fb, _, _ = pkg.dynamics_batch([1.0, 0.1, 0.25, 0.0, 0.05, 0.1, 0.0, 0.8, 100.0], T(xm), T(np.zeros(B)))
    assert torch.equal(fa[2], -fb[2])
    # angles no live problem holds: inf / NaN give NaN (as sin and cos do); a huge finite angle -- a diverging line
    # search trial -- gives a finite, bounded result like the C library's, so that such a trial is rejected the same way
    bad = np.array([[0.0, np.inf, 0.0, 0.0], [0.0, np.nan, 0.0, 0.0], [0.0, 1e300, 0.0, 0.0], [0.0, -3e9, 0.5, 1.0]]).T
    fn, _, _ = pkg.dynamics_batch(DYN_UI, T(bad), T(np.zeros(4)))
    assert not torch.isfinite(fn[2:, :2]).any()
    assert torch.isfinite(fn[:, 2:]).all() and fn[2:, 2:].abs().max().item() < 1e3


def test_survey_known_answers(pkg, survey_answers):
    k = survey_answers
    x, u = T(np.array(k["x"]).reshape(4, 1)), T([k["u"]])
    f, Jx, Ju = pkg.dynamics_batch(k["params"], x, u)
    np.testing.assert_allclose(N_(f)[:, 0], k["f"], rtol=0, atol=5e-14)
    np.testing.assert_allclose(N_(Jx)[2, :, 0], k["J_x_row2"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(N_(Jx)[3, :, 0], k["J_x_row3"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(N_(Ju)[:, 0], k["J_u"], rtol=0, atol=1e-14)
    xn, A, B = pkg.rk4_batch(k["params"], x, u, k["dt"])
    np.testing.assert_allclose(N_(xn)[:, 0], k["rk4_x_new"], rtol=0, atol=2e-15)
    np.testing.assert_allclose(N_(A)[0, :, 0], k["rk4_A_row0"], rtol=0, atol=1e-15)
    np.testing.assert_allclose(N_(B)[:, 0], k["rk4_B"], rtol=0, atol=1e-15)


@pytest.mark.parametrize("ext", [False, True])
def test_rk4_matches_oracle(pkg, orc, ext):
    rng = np.random.default_rng(7)
    B = 300  # ragged: not a multiple of the 64-lane wave
    x = np.stack([rng.uniform(-1.5, 1.5, B), rng.uniform(-4, 4, B), rng.uniform(-2, 2, B), rng.uniform(-6, 6, B)])
    u = rng.uniform(-50, 50, B)
    fext = [1.5, -0.7, 0.4, -0.9] if ext else None
    xn, A, Bm = pkg.rk4_batch(DYN_UI, T(x), T(u), 0.01, fext=fext)
    xn2 = pkg.rk4_batch(DYN_UI, T(x), T(u), 0.01, fext=fext, jacobians=False)
    xn, A, Bm, xn2 = N_(xn), N_(A), N_(Bm), N_(xn2)
    for b in range(B):
        fb, fm = (fext[:2], fext[2:]) if ext else (None, None)
        xo, Ao, Bo = orc.rk4(DYN_UI, x[:, b], u[b], 0.01, fb, fm)
        np.testing.assert_allclose(xn[:, b], xo, rtol=0, atol=1e-13)
        np.testing.assert_allclose(xn2[:, b], xo, rtol=0, atol=1e-13)
        np.testing.assert_allclose(A[:, :, b], Ao, rtol=0, atol=1e-13)
        np.testing.assert_allclose(Bm[:, b], Bo, rtol=0, atol=1e-14)


def test_rk4_derivatives_like_reference(pkg):
    """IntegrationTest.TestDerivatives (integration_test.cc:45-80) on the GPU path itself: analytic
    A, B against the 6th-order stencil applied to the GPU's Jacobian-free RK4."""
    x0 = np.array([1.2, 0.7, 0.4, -0.15])
    u0, dt, h = 0.1, 0.01, 0.01
    _, A, Bm = pkg.rk4_batch(DYN_DERIV, T(x0.reshape(4, 1)), T([u0]), dt)
    offs = np.array([-3, -2, -1, 1, 2, 3]) * h
    coef = np.array([-1, 9, -45, 45, -9, 1]) / (60 * h)
    cols = []
    for j in range(4):
        X = np.repeat(x0.reshape(4, 1), 6, axis=1)
        X[j] += offs
        Y = N_(pkg.rk4_batch(DYN_DERIV, T(X), T(np.full(6, u0)), dt, jacobians=False))
        cols.append(Y @ coef)
    A_num = np.stack(cols, axis=1)
    Y = N_(pkg.rk4_batch(DYN_DERIV, T(np.repeat(x0.reshape(4, 1), 6, axis=1)), T(u0 + offs), dt, jacobians=False))
    B_num = Y @ coef
    assert np.linalg.norm(N_(A)[:, :, 0] - A_num) < 1.0e-12
    assert np.linalg.norm(N_(Bm)[:, 0] - B_num) < 1.0e-12


def test_simulator_matches_oracle(pkg, orc):
    """Simulator::Step (simulator.cc:11-36) batched, with per-problem external forces."""
    rng = np.random.default_rng(9)
    B = 130
    sim = pkg.BatchSimulator(B, dtype=torch.float64, device=0)
    np.testing.assert_array_equal(N_(sim.get_state())[:, 0], [0.0, -np.pi / 2, 0.0, 0.0])  # simulator.hpp:28
    st = random_states(rng, B)
    st[1] = rng.uniform(2.9, 3.14, B)  # near the wrap
    st[3] = rng.uniform(2, 6, B)
    u = rng.uniform(-20, 20, B)
    fext = rng.uniform(-3, 3, (4, B))
    sim.set_state(T(st))
    for dt in (0.01, 0.0025):
        sim.step(DYN_TEST, dt, T(u), fext=T(fext))
    got = N_(sim.get_state())
    for b in range(B):
        o = orc.Simulator()
        o.set_state(st[:, b])
        for dt in (0.01, 0.0025):
            o.step(DYN_TEST, dt, u[b], fext[:2, b], fext[2:, b])
        np.testing.assert_allclose(got[:, b], o.get_state(), rtol=0, atol=1e-12)
    # shared forces path and dt = 0 (no-op)
    sim.set_state(T(st))
    sim.step(DYN_TEST, 0.0, T(u))
    np.testing.assert_array_equal(N_(sim.get_state()), st)
    sim.step(DYN_TEST, 0.003, T(u), f_base=(2.0, 0.0), f_mass=(0.0, -1.0))
    o = orc.Simulator()
    o.set_state(st[:, 5])
    o.step(DYN_TEST, 0.003, u[5], (2.0, 0.0), (0.0, -1.0))
    np.testing.assert_allclose(N_(sim.get_state())[:, 5], o.get_state(), rtol=0, atol=1e-12)
    with pytest.raises(pkg.CpmpcError):  # simulator.cc:13
        sim.step(DYN_TEST, -0.01, T(u))


# ------------------------------------------------------------------------------------------------
# a5: shooting constraints + chain rule
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,sp", [(40, 10), (40, 5), (20, 10), (40, 20), (8, 1), (8, 2), (16, 4), (16, 8),
                                  (30, 3), (30, 6), (30, 15), (21, 7), (40, 40)])   # second row: generic kernel
def test_linearize_matches_oracle(pkg, orc, N, sp):
    S = N // sp + 1
    rng = np.random.default_rng(N * 100 + sp)
    B = 70
    p = pkg.default_params(window_length=N, state_spacing=sp)
    opt = pkg.BatchOptimization(p, max_batch=B, dtype=torch.float64, device=0)
    z = np.concatenate([np.tile(random_states(rng, B), (S, 1)) + rng.normal(0, 0.05, (4 * S, B)),
                        rng.uniform(-30, 30, (N, B))])
    z[1::4][:S] += rng.uniform(-0.3, 0.3, (S, B))
    c, Phi, Gam = opt.linearize(T(z), DYN_UI)
    c, Phi, Gam = N_(c), N_(Phi), N_(Gam)
    for b in range(0, B, 3):
        for s in range(S - 1):
            vars_ = np.concatenate([z[4 * s:4 * s + 4, b], z[4 * (s + 1):4 * (s + 1) + 4, b],
                                    z[4 * S + s * sp:4 * S + (s + 1) * sp, b]])
            err, J = orc.shooting_constraint(DYN_UI, sp, 0.01, vars_)
            np.testing.assert_allclose(c[4 * s:4 * s + 4, b], err, rtol=0, atol=1e-12)
            np.testing.assert_allclose(Phi[s, :, :, b], J[:, :4], rtol=0, atol=1e-12)
            np.testing.assert_allclose(Gam[s * sp:(s + 1) * sp, :, b].T, J[:, 8:], rtol=0, atol=1e-12)


# ------------------------------------------------------------------------------------------------
# a6-a9: the full re-plan
# ------------------------------------------------------------------------------------------------
def _compare_step(pkg, orc, over, x0, dyn=DYN_UI, set_point=0.0, tol=1e-5):
    B = x0.shape[1]
    opt = pkg.BatchOptimization(pkg.default_params(**over), max_batch=B, dtype=torch.float64, device=0)
    out = opt.step(T(x0), dyn, set_point, want_guess=True)
    torch.cuda.synchronize()
    u_cpu, pred_cpu, st_cpu, it_cpu, _ = orc.step_batch_cold(orc.default_opt_params(**over), dyn, set_point, x0,
                                                             want_pred=True)
    err = np.abs(N_(out.u) - u_cpu).max(axis=0)
    perr = np.abs(N_(out.predicted_states) - pred_cpu).max(axis=(0, 1))
    return out, err, perr, N_(out.status) == st_cpu, N_(out.iterations) == it_cpu


def test_step_parity_config2(pkg, orc):
    """BASELINE.json configs[1]: batch=4096 random initial states, N=40, fp64, 5 SQP iterations, exits
    disabled.  Control sequence within 1e-5 of the oracle on every lane."""
    rng = np.random.default_rng(0)
    x0 = random_states(rng, 4096)
    out, err, perr, st_ok, it_ok = _compare_step(pkg, orc, NO_TOL, x0)
    print("config2: |du| median %.2e p99 %.2e max %.2e; pred max %.2e" % (
        np.median(err), np.quantile(err, 0.99), err.max(), perr.max()))
    assert st_ok.all() and it_ok.all()
    assert err.max() < 1e-5
    assert perr.max() < 1e-5
    assert (N_(out.iterations) == 5).all()


def test_step_parity_default_exits(pkg, orc):
    """Reference defaults (8 iterations, exits enabled, optimization.hpp:12-48): same termination state,
    same iteration count, same controls."""
    rng = np.random.default_rng(1)
    x0 = random_states(rng, 1000)
    x0[1, :500] = np.pi / 2 + rng.uniform(-0.4, 0.4, 500)  # half near upright: these converge and exit
    out, err, perr, st_ok, it_ok = _compare_step(pkg, orc, {}, x0)
    st = N_(out.status)
    assert len(set(st.tolist())) >= 2, "expected a mix of termination states"
    assert st_ok.all() and it_ok.all()
    assert err.max() < 1e-5


@pytest.mark.parametrize("over", [
    dict(window_length=40, state_spacing=5, max_iterations=6),                        # optimization_test.cc:13-19
    dict(window_length=20, state_spacing=10, max_iterations=6),                       # BASELINE config 1 shape
    dict(window_length=20, state_spacing=5, max_iterations=4),
    dict(window_length=40, state_spacing=20, max_iterations=4),
    dict(window_length=16, state_spacing=8, max_iterations=4),
    dict(window_length=8, state_spacing=1, max_iterations=4),                         # pure multiple shooting
    dict(window_length=30, state_spacing=3, max_iterations=4),                        # spacings served by the generic kernel
    dict(window_length=30, state_spacing=15, max_iterations=4),
    dict(window_length=21, state_spacing=7, max_iterations=4),
    dict(window_length=40, state_spacing=40, max_iterations=4),                       # single shooting: one interval
    dict(max_iterations=30, u_cost_weight=0.0, b_x_final_cost_weight=5.0, absolute_first_derivative_tol=1e-3,
         b_x_dot_final_cost_weight=100.0, th_dot_final_cost_weight=100.0),            # model/scratch.py:26-36
    dict(max_iterations=5, th_final_cost_weight=50.0, b_x_dot_final_cost_weight=0.0,
         th_dot_final_cost_weight=3.0),                                               # every terminal row a cost
    dict(max_iterations=5, b_x_final_cost_weight=-1.0),                               # every terminal row an equality
    dict(max_iterations=5, u_derivative_cost_weight=0.0),                             # no du rows
    dict(max_iterations=5, control_dt=0.02, u_guess_sinusoid_amplitude=3.0, equality_penalty_initial=10.0),
])
def test_step_parity_configurations(pkg, orc, over):
    rng = np.random.default_rng(42)
    x0 = random_states(rng, 192)
    x0[1, ::2] = np.pi / 2 + rng.uniform(-0.5, 0.5, 96)
    out, err, perr, st_ok, it_ok = _compare_step(pkg, orc, over, x0, dyn=DYN_TEST, set_point=0.1)
    # every lane: same termination state, same iteration count, controls and predicted states within 1e-5 (before
    # the multipliers of the terminal rows were refined -- DESIGN.md section 5 -- 1-2 % of the lanes of the longer
    # runs drifted past 1e-5; since then none does)
    assert st_ok.all() and it_ok.all()
    assert err.max() < 1e-5, np.sort(err)[-5:]
    assert perr.max() < 1e-5


@pytest.mark.parametrize("pipeline", ["fused", "split"])
@pytest.mark.parametrize("sopts", [
    dict(ls_alpha_growth=2.0, ls_alpha_growth_backtracked=1.0),   # grow the step only after a first-trial accept
    dict(ls_alpha_growth=0.0),                                    # no step-length memory
    dict(ls_shrink_max=0.7, ls_shrink_min=0.2, armijo_c1=1e-2, max_line_search_iterations=3,
         lambda_failure_init=1.0, penalty_rho=0.5),
])
def test_step_parity_solver_options(pkg, orc, sopts, pipeline):
    """The knobs of this repo's SQP specification (cpmpc_solver_opts) mean the same in the
    kernels and in the oracle."""
    rng = np.random.default_rng(7)
    x0 = random_states(rng, 512)
    opt = pkg.BatchOptimization(pkg.default_params(**NO_TOL), max_batch=512, dtype=torch.float64, device=0,
                                opts=pkg.capi.default_solver_opts(**sopts))
    opt.set_pipeline(pipeline)
    out = opt.step(T(x0), DYN_UI, 0.0, want_stats=True)
    u_cpu, _, st_cpu, it_cpu, _ = orc.step_batch_cold(orc.default_opt_params(**NO_TOL), DYN_UI, 0.0, x0,
                                                      opts=orc.default_solver_opts(**sopts))
    assert (N_(out.status) == st_cpu).all() and (N_(out.iterations) == it_cpu).all()
    err = np.abs(N_(out.u) - u_cpu).max(axis=0)
    assert err.max() < 1e-5, np.sort(err)[-5:]


def _random_case(rng):
    N, sp = [(40, 10), (40, 5), (20, 10), (20, 5), (40, 20), (80, 10), (40, 8), (30, 6), (24, 3), (16, 16)][rng.integers(0, 10)]
    sign = lambda w: float(w if rng.random() < 0.5 else -1.0)     # cost row or equality row
    over = dict(
        window_length=N, state_spacing=sp, max_iterations=int(rng.integers(2, 6)),
        control_dt=float(rng.choice([0.005, 0.01, 0.02])),
        relative_exit_tol=float(rng.choice([0.0, 1e-5, 1e-3])),
        absolute_first_derivative_tol=float(rng.choice([0.0, 1e-6, 1e-2])),
        equality_penalty_initial=float(10.0 ** rng.uniform(-1, 2)),
        u_guess_sinusoid_amplitude=float(rng.choice([0.0, 3.0, 10.0])),
        u_cost_weight=float(rng.choice([0.0, 0.01, 0.1, 1.0])),
        u_derivative_cost_weight=float(rng.choice([0.0, 0.05, 0.1, 1.0])),
        b_x_final_cost_weight=sign(10.0 ** rng.uniform(0, 2.5)),
        th_final_cost_weight=sign(10.0 ** rng.uniform(0, 2.5)),
        b_x_dot_final_cost_weight=sign(10.0 ** rng.uniform(0, 2)),
        th_dot_final_cost_weight=sign(10.0 ** rng.uniform(0, 2)))
    if over["u_cost_weight"] == 0.0 and over["u_derivative_cost_weight"] == 0.0:
        over["u_cost_weight"] = 0.1                      # some control cost, or the QP is singular by construction
    if over["window_length"] * over["control_dt"] > 1.0:
        over["control_dt"] = 0.01                        # the library refuses horizons beyond 1.0 s unless asked (cpmpc_create_ex)
    dyn = [float(rng.uniform(0.5, 2.0)), float(rng.uniform(0.05, 0.3)), float(rng.uniform(0.15, 0.5)), 9.81,
           float(rng.choice([0.0, 0.05, 0.2])), float(rng.choice([1e-7, 0.05, 0.1])), float(rng.choice([0.0, 0.02, 0.1])),
           float(rng.uniform(0.5, 1.0)), float(rng.choice([0.0, 50.0, 100.0]))]
    return over, dyn, float(rng.uniform(-0.3, 0.3))


@pytest.mark.parametrize("seed", range(32))
def test_step_parity_fuzz(pkg, orc, seed):
    """Seeded random problem definitions (horizons served by the fused, the split and the generic-spacing kernels;
    cost/equality terminal rows in every mix; zero weights; friction / drag / bumpers on and off; exits on and off)
    against the oracle, fp64: same termination state and iteration count, controls within 1e-5."""
    rng = np.random.default_rng(1000 + seed)
    over, dyn, sp = _random_case(rng)
    B = 96
    x0 = random_states(rng, B)
    x0[1, ::2] = np.pi / 2 + rng.uniform(-0.4, 0.4, B // 2)
    out, err, perr, st_ok, it_ok = _compare_step(pkg, orc, over, x0, dyn=dyn, set_point=sp)
    assert st_ok.all() and it_ok.all(), (over, dyn)           # every lane of every case
    assert err.max() < 1e-5, (over, dyn, np.sort(err)[-5:])
    assert np.median(err) < 1e-8


def test_edge_batches(pkg, orc):
    """B = 1, a ragged B, capacity errors, and the reference's scratch.py call sequence at B = 1."""
    rng = np.random.default_rng(3)
    for B in (1, 65, 127):
        x0 = random_states(rng, B)
        out, err, perr, st_ok, it_ok = _compare_step(pkg, orc, NO_TOL, x0)
        assert st_ok.all() and err.max() < 1e-5
    opt = pkg.BatchOptimization(pkg.default_params(), max_batch=64, dtype=torch.float64, device=0)
    with pytest.raises(pkg.CpmpcError) as ei:
        opt.step(T(random_states(rng, 65)), DYN_UI, 0.0)
    assert ei.value.code == pkg.capi.ERR_BATCH
    with pytest.raises(pkg.CpmpcError):
        opt.step(T(random_states(rng, 4)), DYN_UI, float("nan"))
    with pytest.raises(TypeError):
        opt.step(T(random_states(rng, 4), torch.float32), DYN_UI, 0.0)


def test_per_problem_parameters_and_set_points(pkg, orc):
    """Heterogeneous SingleCartPoleParams [9,B] and set-points [B]."""
    rng = np.random.default_rng(5)
    B = 96
    x0 = random_states(rng, B)
    x0[1] = np.pi / 2 + rng.uniform(-0.5, 0.5, B)
    dyn = np.tile(np.array(DYN_UI).reshape(9, 1), (1, B))
    dyn[0] *= rng.uniform(0.7, 1.3, B)
    dyn[1] *= rng.uniform(0.7, 1.3, B)
    dyn[2] *= rng.uniform(0.8, 1.2, B)
    dyn[4] = rng.uniform(0.0, 0.2, B)
    dyn[6] = rng.uniform(0.0, 0.3, B)
    sp = rng.uniform(-0.3, 0.3, B)
    over = dict(max_iterations=6)
    opt = pkg.BatchOptimization(pkg.default_params(**over), max_batch=B, dtype=torch.float64, device=0)
    out = opt.step(T(x0), T(dyn), T(sp))
    u = N_(out.u)
    p = orc.default_opt_params(**over)
    for b in range(B):
        o = orc.Optimization(p).step(x0[:, b], dyn[:, b], sp[b])
        assert N_(out.status)[b] == o.solver_outputs.termination_state
        np.testing.assert_allclose(u[:, b], o.u, rtol=0, atol=1e-5)


@pytest.mark.parametrize("pipeline", ["fused", "split"])
def test_per_problem_terminal_weights(pkg, orc, pipeline):
    """Every problem with its own terminal rows (cost <-> equality per sign, its own weights): the UI's per-controller
    toggles (viz/src/application.ts:279-342) at batch scale.  Checked problem by problem against the oracle built
    with those weights as its OptimizationParams; weights equal to the handle's parameters reproduce the shared path
    bitwise."""
    rng = np.random.default_rng(33)
    B = 160
    x0 = random_states(rng, B)
    x0[1, ::2] = np.pi / 2 + rng.uniform(-0.4, 0.4, B // 2)
    w = 10.0 ** rng.uniform(0, 2.3, (4, B))
    w[rng.random((4, B)) < 0.5] = -1.0
    names = ["b_x_final_cost_weight", "th_final_cost_weight", "b_x_dot_final_cost_weight", "th_dot_final_cost_weight"]
    opt = pkg.BatchOptimization(pkg.default_params(**NO_TOL), max_batch=B, dtype=torch.float64, device=0)
    opt.set_pipeline(pipeline)
    out = opt.step(T(x0), DYN_UI, 0.05, terminal_weights=T(w))
    u = N_(out.u)
    st = N_(out.status)
    bad = 0
    for b in range(B):
        over = dict(NO_TOL, **{n: float(w[t, b]) for t, n in enumerate(names)})
        uc, _, sc, _, _ = orc.step_batch_cold(orc.default_opt_params(**over), DYN_UI, 0.05, x0[:, b:b + 1])
        if sc[0] != st[b] or np.abs(u[:, b] - uc[:, 0]).max() >= 1e-5:
            bad += 1
    assert bad == 0, bad   # every problem, each with its own mix of cost / equality rows
    # the handle's own weights given per problem: bitwise the shared path
    p = pkg.default_params(**NO_TOL)
    shared = np.array([[getattr(p, n)] * B for n in names])
    opt.reset()
    a = opt.step(T(x0), DYN_UI, 0.05, terminal_weights=T(shared)).u.clone()
    opt.reset()
    assert torch.equal(a, opt.step(T(x0), DYN_UI, 0.05).u)


def test_warm_start_closed_loop(pkg, orc):
    """Optimization::Step over consecutive ticks with the plant in the loop (optimization_test.cc:39-61),
    64 controllers at once: warm-start shift, u_prev bookkeeping and Simulator all on the GPU, compared
    tick by tick with 64 oracle controllers."""
    rng = np.random.default_rng(8)
    B, ticks = 64, 25
    x0 = random_states(rng, B)
    x0[1] = np.pi / 2 + rng.uniform(-0.6, 0.6, B)
    over = dict(state_spacing=5, max_iterations=10)
    opt = pkg.BatchOptimization(pkg.default_params(**over), max_batch=B, dtype=torch.float64, device=0)
    sim = pkg.BatchSimulator(B, dtype=torch.float64, device=0)
    sim.set_state(T(x0))
    p = orc.default_opt_params(**over)
    o_opt = [orc.Optimization(p) for _ in range(B)]
    o_sim = [orc.Simulator() for _ in range(B)]
    for b in range(B):
        o_sim[b].set_state(x0[:, b])
    assert not opt.has_previous_solution()
    # Every lane at every tick (round 3: re-measured after round 2's fixes of the normal-equations conditioning and of
    # the diverged-trial rule -- 0 of the 64 x 25 re-plans differ by more than 1e-6 or in their termination state, so
    # nothing is dropped any more).
    worst = 0.0
    for t in range(ticks):
        out = opt.step(sim.get_state().clone(), DYN_TEST, 0.0)
        u0 = out.u[0].contiguous()
        sim.step(DYN_TEST, 0.01, u0)
        u_gpu, st_gpu = N_(out.u), N_(out.status)
        for b in range(B):
            o = o_opt[b].step(o_sim[b].get_state(), DYN_TEST, 0.0)
            o_sim[b].step(DYN_TEST, 0.01, o.u[0])
            assert st_gpu[b] == o.solver_outputs.termination_state, (t, b)
            worst = max(worst, np.abs(u_gpu[:, b] - o.u).max())
    assert opt.has_previous_solution()
    print("warm-start closed loop, 64 controllers x 25 ticks: worst |du| %.2e" % worst)
    assert worst < 1e-5
    state = N_(sim.get_state())
    for b in range(B):
        np.testing.assert_allclose(state[:, b], o_sim[b].get_state(), rtol=0, atol=1e-6)


def test_reset_and_set_previous_solution(pkg, orc):
    """Optimization::Reset / SetPreviousSolution (optimization.hpp:83-89)."""
    rng = np.random.default_rng(12)
    B = 80
    x0, x1 = random_states(rng, B), random_states(rng, B)
    over = dict(max_iterations=3)
    opt = pkg.BatchOptimization(pkg.default_params(**over), max_batch=128, dtype=torch.float64, device=0)
    out1 = opt.step(T(x0), DYN_UI, 0.0)
    u1 = out1.u.clone()
    z1 = opt.get_solution(B)
    out2 = opt.step(T(x1), DYN_UI, 0.0, want_guess=True)
    u2, g2 = out2.u.clone(), out2.guess.clone()
    S, N = 5, 40
    assert torch.equal(g2[4 * S:4 * S + N - 1], z1[4 * S + 1:])  # shift left (optimization.cc:54-57)
    assert torch.equal(g2[-1], z1[-1]) and torch.equal(g2[:4], T(x1))
    opt.reset()
    assert not opt.has_previous_solution()
    assert torch.equal(opt.step(T(x0), DYN_UI, 0.0).u, u1)
    opt.reset()
    opt.set_previous_solution(z1)
    assert torch.equal(opt.step(T(x1), DYN_UI, 0.0).u, u2)
    o = orc.Optimization(orc.default_opt_params(**over))
    o.step(x0[:, 7], DYN_UI, 0.0)
    np.testing.assert_allclose(N_(u2)[:, 7], o.step(x1[:, 7], DYN_UI, 0.0).u, rtol=0, atol=1e-5)


def test_non_finite_lane_does_not_poison_neighbours(pkg):
    rng = np.random.default_rng(13)
    B = 128
    x0 = random_states(rng, B)
    opt = pkg.BatchOptimization(pkg.default_params(**NO_TOL), max_batch=B, dtype=torch.float64, device=0)
    clean = opt.step(T(x0), DYN_UI, 0.0).u.clone()
    bad = x0.copy()
    bad[2, 17] = np.nan
    bad[0, 64] = np.inf
    opt.reset()
    out = opt.step(T(bad), DYN_UI, 0.0)
    st = N_(out.status)
    assert st[17] == pkg.capi.TERM["NON_FINITE"] and st[64] == pkg.capi.TERM["NON_FINITE"]
    keep = np.ones(B, bool)
    keep[[17, 64]] = False
    assert torch.equal(out.u[:, T(keep, torch.bool)], clean[:, T(keep, torch.bool)])


# ------------------------------------------------------------------------------------------------
# full-size properties (BASELINE.json configs[2]: batch = 262144, N = 40, fp32)
# ------------------------------------------------------------------------------------------------
def test_full_size_properties_fp32(pkg, orc):
    rng = np.random.default_rng(0)
    B = 262144
    x0 = random_states(rng, B)
    x0t = T(x0, torch.float32)
    opt = pkg.BatchOptimization(pkg.default_params(**NO_TOL), max_batch=B, dtype=torch.float32, device=0)
    out = opt.step(x0t, DYN_UI, 0.0)
    u, pred, st = out.u.clone(), out.predicted_states.clone(), out.status.clone()
    ok = st == pkg.capi.TERM["MAX_ITERATIONS"]
    assert ok.float().mean().item() > 0.999       # a handful of fp32 lanes may report QP_INDEFINITE
    assert torch.isfinite(u[:, ok]).all() and torch.isfinite(pred[:, :, ok]).all()
    assert u.abs().max().item() <= 300.0          # retraction clamp (optimization.cc:327)
    assert (pred[:, 1].abs() <= np.pi + 1e-6).all()  # mod_pi after every predicted step
    # determinism: same inputs, bitwise the same outputs
    opt.reset()
    out2 = opt.step(x0t, DYN_UI, 0.0)
    assert torch.equal(out2.u, u) and torch.equal(out2.status, st)
    # batch-position independence: a sub-batch solved alone gives bitwise the same answers
    idx = torch.arange(1000, 1000 + 4096, device=DEV)
    small = pkg.BatchOptimization(pkg.default_params(**NO_TOL), max_batch=4096, dtype=torch.float32, device=0)
    outs = small.step(x0t[:, idx].contiguous(), DYN_UI, 0.0)
    assert torch.equal(outs.u, u[:, idx])
    # predicted states are the single-shooting rollout of u (optimization.cc:353-371): every predicted
    # state is one RK4 step (+ mod_pi) from the one before it, re-derived with the RK4 entry point.  Checked
    # step by step, so fp32 rounding is not amplified through 40 steps of fast dynamics (|u| up to 300).
    x = x0t[:, :8192].contiguous()
    for k in range(40):
        x = pkg.rk4_batch(DYN_UI, x, u[k, :8192].contiguous(), 0.01, jacobians=False)
        x[1] = torch.remainder(x[1] + np.pi, 2 * np.pi) - np.pi
        d = (x - pred[k, :, :8192])
        d[1] = torch.remainder(d[1] + np.pi, 2 * np.pi) - np.pi
        assert (d.abs() <= 2e-5 * (1.0 + x.abs())).all(), k
        x = pred[k, :, :8192].contiguous()
    # fp32 against the fp64 oracle on a sample of 2048 lanes (the reference is fp64-only: this is the price of single
    # precision through five SQP iterations on unconverged problems, not a parity claim).  Measured: median 2.9e-4,
    # 90.6 % of the lanes within 1e-2 (92.3 % over the whole batch, bench.py parity_sample).
    samp = np.arange(0, B, B // 2048)[:2048]
    u_cpu, _, _, _, _ = orc.step_batch_cold(orc.default_opt_params(**NO_TOL), DYN_UI, 0.0, x0[:, samp])
    err = np.abs(N_(u[:, T(samp, torch.long)]) - u_cpu).max(axis=0)
    print("fp32 vs fp64 oracle on 2048 lanes: |du| median %.2e p90 %.2e within 1e-2: %.3f" % (
        np.median(err), np.quantile(err, 0.9), (err < 1e-2).mean()))
    assert np.median(err) < 2e-3
    assert (err < 1e-2).mean() >= 0.88


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-4), (torch.float64, 1e-6)])
def test_closed_loop_balances(pkg, dtype, tol):
    """Property at batch scale (the reference's closed-loop criterion, optimization_test.cc:63-66):
    256 controllers from random near-upright states all end upright and still after 3 s, in fp32 too."""
    rng = np.random.default_rng(3)
    B = 256
    x0 = np.stack([rng.uniform(-0.3, 0.3, B), np.pi / 2 + rng.uniform(-0.4, 0.4, B), rng.uniform(-0.5, 0.5, B),
                   rng.uniform(-1, 1, B)])
    opt = pkg.BatchOptimization(pkg.default_params(), max_batch=B, dtype=dtype, device=0)
    sim = pkg.BatchSimulator(B, dtype=dtype, device=0)
    sim.set_state(T(x0, dtype))
    for _ in range(300):
        out = opt.step(sim.get_state().clone(), DYN_UI, 0.0, want_predicted=False)
        sim.step(DYN_UI, 0.01, out.u[0].contiguous())
    s = N_(sim.get_state().double())
    assert np.abs(s[1] - np.pi / 2).max() < tol and np.abs(s[0]).max() < tol
    assert np.abs(s[2]).max() < 10 * tol and np.abs(s[3]).max() < 10 * tol
    st = N_(out.status)
    assert not np.isin(st, [pkg.capi.TERM["QP_INDEFINITE"], pkg.capi.TERM["MAX_LAMBDA"],
                            pkg.capi.TERM["NON_FINITE"]]).any()


def test_full_size_fp64_every_lane(pkg, orc):
    """BASELINE configs[2]'s batch (262 144 problems, N = 40, 5 iterations, cold start) in the parity dtype: EVERY lane
    within 1e-5 of the oracle, same termination state and iteration count (measured: worst lane 8e-7, 99th percentile
    3e-10).  The oracle runs the whole batch on the host's cores (a few seconds on the GPU box's 16)."""
    rng = np.random.default_rng(21)
    B = 262144
    x0 = random_states(rng, B)
    opt = pkg.BatchOptimization(pkg.default_params(**NO_TOL), max_batch=B, dtype=torch.float64, device=0)
    out = opt.step(T(x0), DYN_UI, 0.0)
    import os
    threads = min(len(os.sched_getaffinity(0)), 16)
    u_cpu, _, st_cpu, it_cpu, _ = orc.step_batch_cold(orc.default_opt_params(**NO_TOL), DYN_UI, 0.0, x0, num_threads=threads)
    err = np.abs(N_(out.u) - u_cpu).max(axis=0)
    print("full batch fp64: |du| max %.2e p99 %.2e median %.2e" % (err.max(), np.quantile(err, 0.99), np.median(err)))
    assert (N_(out.status) == st_cpu).all() and (N_(out.iterations) == it_cpu).all()
    assert err.max() < 1e-5, np.sort(err)[-5:]


def test_reference_defaults_full_batch_exits_enabled(pkg, orc):
    """The reference's own defaults (optimization.hpp:12-48: 8 iterations, relative_exit_tol 1e-5,
    absolute_first_derivative_tol 1e-6) on the benchmark's 262 144 problems: termination state and iteration count agree
    with the oracle on every lane, and every lane is within 1e-5 -- a lane that is not is handed to the extended-precision
    build of the oracle, and the GPU must then be the closer of the two to it (round 2's sweep found one such lane at
    1.3e-5, where the double oracle had moved 1.0e-5 and the GPU 3.2e-6)."""
    import os
    import bench
    B = 262144
    x0 = bench.synth_states(bench.SEED, B)
    opt = pkg.BatchOptimization(pkg.default_params(), max_batch=B, dtype=torch.float64, device=0)
    out = opt.step(T(x0), DYN_UI, 0.0)
    threads = min(len(os.sched_getaffinity(0)), 16)
    u_cpu, _, st_cpu, it_cpu, _ = orc.step_batch_cold(orc.default_opt_params(), DYN_UI, 0.0, x0, num_threads=threads)
    u_gpu = N_(out.u)
    err = np.abs(u_gpu - u_cpu).max(axis=0)
    assert (N_(out.status) == st_cpu).all() and (N_(out.iterations) == it_cpu).all()
    over = np.nonzero(err > 1e-5)[0]
    print("reference defaults, exits on, B = 262144: |du| max %.2e p99 %.2e; lanes over 1e-5: %d" % (
        err.max(), np.quantile(err, 0.99), over.size))
    assert over.size <= 4, over.size
    if over.size:
        u_ld, _, _, _, _ = orc.step_batch_cold_ld(orc.default_opt_params(), DYN_UI, 0.0, x0[:, over])
        e_gpu = np.abs(u_gpu[:, over] - u_ld).max(axis=0)
        e_cpu = np.abs(u_cpu[:, over] - u_ld).max(axis=0)
        print("  arbiter (extended precision): GPU %s, oracle %s" % (e_gpu, e_cpu))
        assert (e_gpu < 1e-5).all() and (e_gpu <= e_cpu).all(), (over, e_gpu, e_cpu)


def test_profiling_counts_launches(pkg):
    rng = np.random.default_rng(1)
    B = 256
    opt = pkg.BatchOptimization(pkg.default_params(max_iterations=5), max_batch=B, dtype=torch.float32, device=0)
    opt.profile_enable(True)
    opt.profile_reset()
    for _ in range(3):
        opt.step(T(random_states(rng, B), torch.float32), DYN_UI, 0.0)
    prof = opt.profile_read()
    assert prof["prepare_kernel"][1] == 3 and prof["finalize_kernel"][1] == 3
    assert prof["fused_sqp_kernel"][1] == 3 and prof["linearize_kernel"][1] == 0   # auto -> fused here
    opt.set_compaction(3, 2)                     # explicit staging: 3 + 2 iterations = two launches per step
    opt.profile_reset()
    for _ in range(3):
        opt.step(T(random_states(rng, B), torch.float32), DYN_UI, 0.0)
    assert opt.profile_read()["fused_sqp_kernel"][1] == 6
    opt.set_compaction(0, 0)
    opt.set_pipeline("split")
    opt.profile_reset()
    for _ in range(3):
        opt.step(T(random_states(rng, B), torch.float32), DYN_UI, 0.0)
    prof = opt.profile_read()
    assert prof["linearize_kernel"][1] == 15 and prof["qp_ls_kernel"][1] == 15 and prof["fused_sqp_kernel"][1] == 0
    assert all(ms > 0 for name, (ms, _) in prof.items() if name != "fused_sqp_kernel")


# ------------------------------------------------------------------------------------------------
# the two pipelines (include/cpmpc.h: CPMPC_PIPELINE_*) implement the same arithmetic
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("over", [dict(window_length=80, state_spacing=10, max_iterations=4),
                                  dict(window_length=160, state_spacing=10, max_iterations=3),
                                  dict(window_length=80, state_spacing=5, max_iterations=3)])
def test_long_horizons_fused(pkg, orc, over):
    """Horizons of 80 and 160 steps: 8 and 16 lanes per problem in the fused kernel (generic group traffic).
    N = 80 (0.8 s; within cpmpc_max_parity_horizon = 1.0 s): EVERY lane of both
    pipelines within 1e-5 of the oracle.  N = 160 (1.6 s) is solved like the reference would (refused only with
    CPMPC_CREATE_STRICT_HORIZON):
    eliminating the states through 16 intervals of an unstable plant loses about three digits per QP against the
    oracle's full-space KKT solve, and at batch scale 0.5 % of cold starts end beyond 1e-5 (profiles/r04_parity_sweep.json;
    neither wider accumulation of the terminal system nor more refinement passes change that); on
    these 96 lanes at most two may, and those are handed to the extended-precision build of the oracle."""
    rng = np.random.default_rng(11)
    x0 = random_states(rng, 96)
    x0[1, ::2] = np.pi / 2 + rng.uniform(-0.3, 0.3, 48)
    long_h = over["window_length"] > 100
    if long_h:  # refused when the caller asks for the 1e-5 bar on every problem, with the reason in the message
        with pytest.raises(pkg.CpmpcError) as exc:
            pkg.BatchOptimization(pkg.default_params(**over), max_batch=96, dtype=torch.float64, device=0, strict_horizon=True)
        assert exc.value.code == pkg.capi.ERR_UNSUPPORTED and "STRICT_HORIZON" in str(exc.value)
    opt = pkg.BatchOptimization(pkg.default_params(**over), max_batch=96, dtype=torch.float64, device=0)
    # round 6: beyond the parity horizon AUTO takes the split pipeline (two passes of QP refinement there); the fused kernel on request
    assert opt.pipeline() == ("split" if long_h else "fused")
    opt.set_pipeline("fused")
    assert opt.pipeline() == "fused"
    out = opt.step(T(x0), DYN_UI, 0.0)
    u_cpu, _, st_cpu, it_cpu, _ = orc.step_batch_cold(orc.default_opt_params(**over), DYN_UI, 0.0, x0)
    ok = (N_(out.status) == st_cpu) & (N_(out.iterations) == it_cpu)
    assert ok.all()
    err = np.abs(N_(out.u) - u_cpu).max(axis=0)
    # every lane within 1e-5 of the oracle; at N = 160 (sixteen intervals of an unstable plant, the worst conditioned case
    # in the suite) a lane may land a little above it -- one did at 1.8e-5 when round 3 changed the fp64 sine/cosine by
    # an ulp -- and is then handed to the extended-precision build of the oracle: the GPU must be as close to that
    # answer as the double oracle is (x2), i.e. the distance is the problem's rounding sensitivity, not a defect
    over_bar = np.nonzero(err >= 1e-5)[0]
    assert over_bar.size <= (2 if long_h else 0) and err.max() < 1e-4, np.sort(err)[-5:]
    if over_bar.size:
        u_ld, _, _, _, _ = orc.step_batch_cold_ld(orc.default_opt_params(**over), DYN_UI, 0.0, x0[:, over_bar])
        e_gpu = np.abs(N_(out.u)[:, over_bar] - u_ld).max(axis=0)
        e_cpu = np.abs(u_cpu[:, over_bar] - u_ld).max(axis=0)
        print("long horizon %s: lanes over 1e-5 %s, GPU vs extended %s, oracle vs extended %s" % (over, err[over_bar], e_gpu, e_cpu))
        assert (e_gpu <= np.maximum(1e-5, 2.0 * e_cpu)).all(), (err[over_bar], e_gpu, e_cpu)
    opt.set_pipeline("split")
    opt.reset()
    out2 = opt.step(T(x0), DYN_UI, 0.0)
    same = N_(out2.status) == N_(out.status)
    tol = 1e-6 if over["window_length"] <= 80 else 1e-4
    assert same.all() and np.abs(N_(out2.u) - N_(out.u)).max() < tol


def test_long_horizon_fp32_stays_finite(pkg):
    """fp32 at a 160-step horizon: state elimination through 1.6 s of an unstable plant is too ill-conditioned for
    single precision (both pipelines report QP_INDEFINITE on some lanes and differ on others; use fp64 there), but
    nothing may overflow or turn into NaN: with w_u = 0 the pivot recurrence's matrix powers decay like 2^-150,
    outside fp32's normal range unless the (p, q) pair is renormalised."""
    rng = np.random.default_rng(12)
    x0 = random_states(rng, 64)
    x0[1] = np.pi / 2 + rng.uniform(-0.05, 0.05, 64)
    over = dict(window_length=160, state_spacing=10, max_iterations=2, u_cost_weight=0.0)
    for pipe in ("fused", "split"):
        opt = pkg.BatchOptimization(pkg.default_params(**over), max_batch=64, dtype=torch.float32, device=0,
                                    allow_long_horizon=True)
        opt.set_pipeline(pipe)
        out = opt.step(T(x0, torch.float32), DYN_UI, 0.0)
        assert torch.isfinite(out.u).all() and torch.isfinite(out.predicted_states).all()
        st = set(N_(out.status).tolist())
        assert st <= {pkg.capi.TERM["MAX_ITERATIONS"], pkg.capi.TERM["QP_INDEFINITE"]}, st
        assert (out.status == pkg.capi.TERM["MAX_ITERATIONS"]).float().mean().item() > 0.5


@pytest.mark.parametrize("over", [dict(state_spacing=8, max_iterations=5), dict(state_spacing=4, max_iterations=4),
                                  dict(window_length=20, state_spacing=4, max_iterations=5)])
def test_fused_with_groups_that_straddle_dpp_rows(pkg, orc, over):
    """5 or 10 intervals per problem: groups of lanes that do not divide a 16-lane DPP row (wave shuffles, ordered
    group sums, 4 riding lanes per wave).  Against the oracle and against the split pipeline; ragged batch."""
    rng = np.random.default_rng(17)
    B = 203
    x0 = random_states(rng, B)
    x0[1, ::2] = np.pi / 2 + rng.uniform(-0.4, 0.4, x0[1, ::2].shape)
    opt = pkg.BatchOptimization(pkg.default_params(**over), max_batch=B, dtype=torch.float64, device=0)
    assert opt.pipeline() == "fused"
    out = opt.step(T(x0), DYN_UI, 0.0, want_stats=True)
    u_cpu, _, st_cpu, it_cpu, _ = orc.step_batch_cold(orc.default_opt_params(**over), DYN_UI, 0.0, x0)
    ok = (N_(out.status) == st_cpu) & (N_(out.iterations) == it_cpu)
    assert ok.all()
    err = np.abs(N_(out.u) - u_cpu).max(axis=0)
    assert err.max() < 1e-5 and np.median(err) < 1e-8
    opt.set_pipeline("split")
    opt.reset()
    out2 = opt.step(T(x0), DYN_UI, 0.0, want_stats=True)
    assert torch.equal(out2.status, out.status) and torch.equal(out2.ls_evals, out.ls_evals)
    assert np.abs(N_(out2.u) - N_(out.u)).max() < 1e-6


@pytest.mark.parametrize("over", [dict(NO_TOL), dict(), dict(state_spacing=5, max_iterations=6),
                                  dict(window_length=20, max_iterations=6),
                                  dict(window_length=20, state_spacing=5, max_iterations=6)])
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_fused_and_split_pipelines_agree(pkg, orc, over, dtype):
    rng = np.random.default_rng(77)
    B = 777  # not a multiple of the problems-per-wave of any group size
    x0 = random_states(rng, B)
    x0[1, ::3] = np.pi / 2 + rng.uniform(-0.4, 0.4, len(x0[1, ::3]))
    outs = {}
    for mode in ("split", "fused"):
        opt = pkg.BatchOptimization(pkg.default_params(**over), max_batch=B, dtype=dtype, device=0)
        opt.set_pipeline(mode)
        assert opt.pipeline() == mode
        o1 = opt.step(T(x0, dtype), DYN_UI, 0.1)
        u1, st1 = o1.u.clone(), o1.status.clone()
        o2 = opt.step(T(x0, dtype) * 1.0, DYN_UI, 0.1)  # a warm-started second step through the same pipeline
        outs[mode] = (u1, st1, o1.iterations.clone(), o1.ls_evals.clone(), o2.u.clone(), o2.status.clone(),
                      opt.get_solution(B))
    a, b = outs["split"], outs["fused"]
    same = (a[1] == b[1]) & (a[2] == b[2]) & (a[3] == b[3])
    if dtype == torch.float64:
        assert same.all()
        d1 = (a[0] - b[0]).abs().max(dim=0).values
        assert d1.max().item() < 1e-6
        same2 = a[5] == b[5]
        assert same2.all() and (a[4] - b[4]).abs().max().item() < 1e-5
        assert (a[6] - b[6]).abs().max(dim=0).values.median().item() < 1e-9
        # and both agree with the oracle
        u_cpu, _, st_cpu, _, _ = orc.step_batch_cold(orc.default_opt_params(**over), DYN_UI, 0.1, x0)
        assert (N_(b[1]) == st_cpu).all() and np.abs(N_(b[0]) - u_cpu).max() < 1e-5
    else:
        # fp32: the group-sum tree of the fused pipeline rounds differently from the split pipeline's serial
        # sums, which flips near-tie Armijo / exit decisions on a minority of lanes
        assert same.float().mean().item() > 0.75
        d1 = (a[0] - b[0]).abs().max(dim=0).values[same]
        assert d1.median().item() < 1e-2


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_staged_fused_pipeline_is_bitwise_the_single_launch(pkg, dtype):
    """With exit tolerances enabled the fused pipeline runs in stages and compacts the still-active problems in
    between (cpmpc_set_compaction).  The kernel restarts from workspace state only and a problem's arithmetic does
    not depend on its lanes, so every staging gives bitwise the single launch's results -- cold and warm."""
    rng = np.random.default_rng(44)
    B = 5000                                     # ragged: not a multiple of 16 or 64
    x0 = random_states(rng, B)
    x0[1, ::3] = np.pi / 2 + rng.uniform(-0.3, 0.3, x0[1, ::3].shape)   # a third converge early, the rest late
    over = dict(max_iterations=8)
    ref = None
    for first, nxt in ((0, 0), (3, 2), (1, 1), (2, 5), (7, 1)):
        opt = pkg.BatchOptimization(pkg.default_params(**over), max_batch=B, dtype=dtype, device=0)
        opt.set_compaction(first, nxt)
        sim = pkg.BatchSimulator(B, dtype=dtype, device=0)
        sim.set_state(T(x0, dtype))
        res = []
        for tick in range(3):                    # tick 0 cold, then warm starts
            o = opt.step(sim.get_state(), DYN_UI, 0.0, want_stats=True, out=pkg.BatchOutputs())
            res.append((o.u.clone(), o.status.clone(), o.iterations.clone(), o.ls_evals.clone(), o.final_cost.clone()))
            sim.step(DYN_UI, 0.01, o.u[0].contiguous())
        if ref is None:
            ref = res
            its = N_(res[0][2])
            assert its.min() < its.max(), "the test needs lanes that stop at different iterations"
            continue
        for got, want in zip(res, ref):
            for g, w in zip(got, want):
                assert torch.equal(g, w), (first, nxt)


def test_staging_engages_by_itself_on_large_batches(pkg):
    """Default settings: a batch larger than one round of resident waves is staged automatically when exit tolerances
    are enabled (more than one fused launch per step), a small one is not; the results are the single launch's."""
    rng = np.random.default_rng(45)
    B = 40000
    x0 = random_states(rng, B)
    x0[1, ::2] = np.pi / 2 + rng.uniform(-0.3, 0.3, B // 2)
    auto = pkg.BatchOptimization(pkg.default_params(), max_batch=B, dtype=torch.float32, device=0)
    auto.profile_enable(True)
    a = auto.step(T(x0, torch.float32), DYN_UI, 0.0, want_stats=True)
    assert auto.profile_read()["fused_sqp_kernel"][1] > 1
    single = pkg.BatchOptimization(pkg.default_params(), max_batch=B, dtype=torch.float32, device=0)
    single.set_compaction(0, 0)
    single.profile_enable(True)
    b = single.step(T(x0, torch.float32), DYN_UI, 0.0, want_stats=True)
    assert single.profile_read()["fused_sqp_kernel"][1] == 1
    assert torch.equal(a.u, b.u) and torch.equal(a.status, b.status) and torch.equal(a.iterations, b.iterations)
    assert len(set(N_(a.iterations).tolist())) > 3
    small = pkg.BatchOptimization(pkg.default_params(), max_batch=512, dtype=torch.float32, device=0)
    small.profile_enable(True)
    small.step(T(x0[:, :512], torch.float32), DYN_UI, 0.0)
    assert small.profile_read()["fused_sqp_kernel"][1] == 1


def test_two_handles_on_two_streams(pkg):
    """Handles are independent: two solvers stepping concurrently on two HIP streams give bitwise the results
    they give one after the other (every launch of a step goes to the caller's current stream)."""
    rng = np.random.default_rng(21)
    B = 8192
    xa, xb = T(random_states(rng, B), torch.float32), T(random_states(rng, B), torch.float32)
    mk = lambda: pkg.BatchOptimization(pkg.default_params(**NO_TOL), max_batch=B, dtype=torch.float32, device=0)
    ref_a, ref_b = mk().step(xa, DYN_UI, 0.0).u.clone(), mk().step(xb, DYN_UI, 0.1).u.clone()
    torch.cuda.synchronize()
    oa, ob = mk(), mk()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    outs_a, outs_b = [], []
    for _ in range(3):
        with torch.cuda.stream(sa):
            oa.reset()
            outs_a.append(oa.step(xa, DYN_UI, 0.0, out=pkg.BatchOutputs()).u)
        with torch.cuda.stream(sb):
            ob.reset()
            outs_b.append(ob.step(xb, DYN_UI, 0.1, out=pkg.BatchOutputs()).u)
    torch.cuda.synchronize()
    for ua, ub in zip(outs_a, outs_b):
        assert torch.equal(ua, ref_a) and torch.equal(ub, ref_b)


def test_pipeline_selection(pkg):
    opt = pkg.BatchOptimization(pkg.default_params(), max_batch=64, dtype=torch.float32, device=0)
    assert opt.pipeline() == "fused"                   # (S-1, state_spacing) = (4, 10) is built
    opt.set_pipeline("split")
    assert opt.pipeline() == "split"
    opt2 = pkg.BatchOptimization(pkg.default_params(state_spacing=2), max_batch=64, dtype=torch.float32, device=0)
    assert opt2.pipeline() == "split"                  # 20 intervals of 2 steps: not built
    with pytest.raises(pkg.CpmpcError) as ei:
        opt2.set_pipeline("fused")
    assert ei.value.code == pkg.capi.ERR_UNSUPPORTED
    # spacings without a compiled specialisation run the run-time-spacing fused kernel when the interval count is one
    # of {2, 4, 5, 8, 10, 16} and the per-wave LDS fits; otherwise the split pipeline
    for N, sp, want in ((30, 3, "fused"), (30, 6, "fused"), (30, 15, "fused"), (16, 4, "fused"), (8, 1, "fused"),
                        (21, 7, "split"), (40, 40, "split"), (400, 100, "split")):
        o = pkg.BatchOptimization(pkg.default_params(window_length=N, state_spacing=sp), max_batch=64,
                                  dtype=torch.float32, device=0, allow_long_horizon=N > 80)
        assert o.pipeline() == want, (N, sp)
    opt3 = pkg.BatchOptimization(pkg.default_params(), max_batch=64, dtype=torch.float32, device=0, model="double")
    assert opt3.pipeline() == "fused"                  # both models are built
    # fp64 double: fused where three waves of it fit a CU's LDS (round 5: 40 KB per wave at spacing 10, 50 KB with the QP
    # refinement; rounds 1-4 needed 60 KB and defaulted to split), split where they do not (100 KB at spacing 20)
    opt4 = pkg.BatchOptimization(pkg.default_params(), max_batch=64, dtype=torch.float64, device=0, model="double")
    assert opt4.pipeline() == "fused"
    opt4.set_pipeline("split")
    assert opt4.pipeline() == "split"
    opt5 = pkg.BatchOptimization(pkg.default_params(), max_batch=64, dtype=torch.float64, device=0, model="double", refine_qp=True)
    assert opt5.pipeline() == "fused"
    opt6 = pkg.BatchOptimization(pkg.default_params(state_spacing=20), max_batch=64, dtype=torch.float64, device=0, model="double")
    assert opt6.pipeline() == "split"
    opt6.set_pipeline("fused")
    assert opt6.pipeline() == "fused"
