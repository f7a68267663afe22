"""GPU parity tests added in round 2: the branch points of the dynamics and of mod_pi in fp32 AND fp64, an
iteration-resolved comparison of the SQP against the oracle (where does a lane first depart, and why), and the
per-problem warm-start rule when the batch grows between steps.  All through the C-ABI."""
import numpy as np
import pytest

from conftest import DYN_TEST, DYN_UI, random_states

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
DEV = "cuda:0"
NO_TOL = dict(max_iterations=5, relative_exit_tol=0.0, absolute_first_derivative_tol=0.0)
DTYPES = [torch.float64, torch.float32]


@pytest.fixture(scope="module", autouse=True)
def _gpu(pkg):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: the product has no CPU fallback")
    assert pkg.capi.load().cpmpc_device_count() >= 1


def T(a, dtype=torch.float64):
    return torch.tensor(np.ascontiguousarray(a), dtype=dtype, device=DEV)


def N_(t):
    return t.detach().cpu().numpy()


def _np_dtype(dtype):
    return np.float64 if dtype == torch.float64 else np.float32


def _ref_mod_pi(a, npdt):
    """integration.hpp:65-73 evaluated in the given precision: fmod, +2pi if negative, -2pi if above pi."""
    pi, two_pi = npdt(np.pi), npdt(2) * npdt(np.pi)
    r = np.fmod(a.astype(npdt), two_pi)          # exact remainder, as C fmod
    r = np.where(r < 0, r + two_pi, r).astype(npdt)
    r = np.where(r > pi, r - two_pi, r).astype(npdt)
    return r


# ------------------------------------------------------------------------------------------------
# a4: mod_pi at the cut, through the batched Simulator (simulator.cc:24-36 wraps after every sub-step)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
def test_mod_pi_cut_points(pkg, orc, dtype):
    """With g = 0, zero speeds, zero forces and the cart between the bumpers the accelerations vanish, so one
    sub-step returns the state unchanged except for the wrap of theta: the kernel's mod_pi, bit for bit, at
    +-pi, one ulp either side, multiples of 2 pi, zero and several turns (integration.hpp:65-73: (-pi, pi],
    mod_pi(+-pi) = +pi)."""
    npdt = _np_dtype(dtype)
    pi = npdt(np.pi)
    up, dn = np.nextafter(pi, npdt(10)), np.nextafter(pi, npdt(0))
    th = np.array([pi, -pi, up, -up, dn, -dn, 0.0, -0.0, 2 * pi, -2 * pi, np.nextafter(2 * pi, npdt(10)),
                   -np.nextafter(2 * pi, npdt(10)), 3 * pi, -3 * pi, 4.0, -4.0, 7 * pi + npdt(0.3), -9 * pi + npdt(0.1),
                   1e-30, -1e-30, 100.0, -100.0, 6.5 * pi, -6.5 * pi], dtype=npdt)
    rng = np.random.default_rng(5)
    th = np.concatenate([th, (rng.uniform(-40, 40, 232)).astype(npdt)])
    B = th.size
    st = np.zeros((4, B), dtype=npdt)
    st[0] = rng.uniform(-0.5, 0.5, B).astype(npdt)
    st[1] = th
    no_gravity = [1.0, 0.1, 0.25, 0.0, 0.05, 0.1, 0.02, 0.8, 100.0]
    sim = pkg.BatchSimulator(B, dtype=dtype, device=0)
    sim.set_state(T(st, dtype))
    sim.step(no_gravity, 0.001, T(np.zeros(B), dtype))
    got = N_(sim.get_state())
    want = _ref_mod_pi(th, npdt)
    np.testing.assert_array_equal(got[0], st[0])
    np.testing.assert_array_equal(got[2:], st[2:])
    assert got[1].dtype == npdt
    bad = np.nonzero(got[1] != want)[0]
    assert bad.size == 0, [(float(th[i]), float(got[1, i]), float(want[i])) for i in bad[:8]]
    assert got[1, 0] == pi and got[1, 1] == pi          # the cut: both ends map to +pi
    assert (got[1] > -pi).all() and (got[1] <= pi).all()
    if dtype == torch.float64:                          # and the oracle's C fmod agrees
        for i in range(24):
            assert orc.mod_pi(float(th[i])) == got[1, i]


# ------------------------------------------------------------------------------------------------
# a1: the branches of the dynamics, fp64 and fp32 (single_pendulum_dynamics.hpp:36-57,75-84)
# ------------------------------------------------------------------------------------------------
def _edge_cases(npdt):
    """(tag, params, x, u, fext) at the branch points.  x_s = 0.75 and the other values are exact in fp32, so both
    precisions see the same comparison operands."""
    xs = 0.75
    P = [1.0, 0.1, 0.25, 9.81, 0.05, 0.1, 0.02, xs, 100.0]
    up = lambda v: float(np.nextafter(npdt(v), npdt(10)))    # noqa: E731
    dn = lambda v: float(np.nextafter(npdt(v), npdt(-10)))   # noqa: E731
    cases = []
    for tag, bx in (("bx=+xs", xs), ("bx=+xs+ulp", up(xs)), ("bx=+xs-ulp", dn(xs)), ("bx=-xs", -xs),
                    ("bx=-xs-ulp", dn(-xs)), ("bx=-xs+ulp", up(-xs)), ("bx=0", 0.0), ("bx=2xs", 1.5), ("bx=-2xs", -1.5)):
        cases.append((tag, P, [bx, 0.625, 0.5, -0.75], 0.25, None))
    # friction: v_mu_b below the 1e-6 floor (max(v_mu, 1e-6), :51-57), at it, and normal; speeds around the floor
    for vmu in (1e-9, 0.0, 1e-6, 2e-6, 0.1):
        Pf = [1.0, 0.1, 0.25, 9.81, 0.2, vmu, 0.0, xs, 100.0]
        for v in (0.0, 1e-7, -1e-7, 1e-6, 3e-6, -0.5):
            cases.append(("vmu=%g v=%g" % (vmu, v), Pf, [0.125, -0.5, v, 0.25], 0.0, None))
    # drag: |v_mass|^2 == 0 exactly (guard 0 < |v|^2, :75-84) with c_d > 0: at rest, and b_x' = L w sin(th) with cos = 0
    Pd = [1.0, 0.1, 0.25, 9.81, 0.0, 0.1, 5.0, xs, 100.0]
    cases.append(("drag rest", Pd, [0.0, 0.75, 0.0, 0.0], 1.0, None))
    cases.append(("drag rest hanging", Pd, [0.25, float(npdt(-np.pi / 2)), 0.0, 0.0], 0.0, None))
    cases.append(("drag tiny speed", Pd, [0.0, 0.75, 1e-20, 0.0], 0.0, None))
    cases.append(("drag pole only", Pd, [0.0, 0.0, 0.0, 2.0], 0.0, None))
    cases.append(("drag cart only", Pd, [0.0, 1.0, -1.5, 0.0], 0.0, None))
    # theta at the cut and straight up / down
    for th in (np.pi, -np.pi, np.pi / 2, -np.pi / 2, 0.0):
        cases.append(("th=%g" % th, P, [0.25, float(npdt(th)), 0.5, 1.0], -2.0, [1.5, -0.75, 0.5, -1.0]))
    return cases


@pytest.mark.parametrize("dtype", DTYPES)
def test_dynamics_branch_points(pkg, orc, dtype):
    """f, J_x, J_u and one RK4 step at every branch point through cpmpc_dynamics_batch / cpmpc_rk4_batch against the
    oracle evaluated (in double) at the same rounded inputs.  fp64: 1e-12 of the term size.  fp32: 2e-5 (f) / 2e-4 (J)
    of the term size -- the hardware sin/cos/exp/rcp approximations -- and the *structure* exactly: the spring
    column of J_x is exactly zero up to and including |b_x| = x_s and switches on one ulp beyond (strict 0 < arg)."""
    npdt = _np_dtype(dtype)
    tol_f, tol_j = (1e-12, 1e-12) if dtype == torch.float64 else (2e-5, 2e-4)
    worst_f = worst_j = 0.0
    for tag, P, x, u, fext in _edge_cases(npdt):
        xr = np.array(x, dtype=npdt)
        ur = npdt(u)
        f, Jx, Ju = pkg.dynamics_batch(P, T(xr.reshape(4, 1), dtype), T([ur], dtype), fext=fext)
        f, Jx, Ju = N_(f)[:, 0].astype(float), N_(Jx)[:, :, 0].astype(float), N_(Ju)[:, 0].astype(float)
        fb, fm = (fext[:2], fext[2:]) if fext else (None, None)
        fo, Jo, Juo = orc.dynamics(P, xr.astype(float), float(ur), fb, fm)
        sf, sj = max(1.0, np.abs(fo).max()), max(1.0, np.abs(Jo).max())
        assert np.isfinite(f).all() and np.isfinite(Jx).all(), tag
        ef, ej = np.abs(f - fo).max() / sf, max(np.abs(Jx - Jo).max() / sj, np.abs(Ju - Juo).max())
        worst_f, worst_j = max(worst_f, ef), max(worst_j, ej)
        assert ef < tol_f, (tag, f, fo)
        assert ej < tol_j, (tag, Jx, Jo)
        # structure: rows 0-1 are [0 0 1 0; 0 0 0 1] exactly; the spring column is on/off exactly as in the oracle
        np.testing.assert_array_equal(Jx[:2], [[0, 0, 1, 0], [0, 0, 0, 1]])
        assert (Jx[2, 0] == 0.0) == (Jo[2, 0] == 0.0), (tag, Jx[2, 0], Jo[2, 0])
        assert (Jx[3, 0] == 0.0) == (Jo[3, 0] == 0.0), (tag, Jx[3, 0], Jo[3, 0])
        xn, A, Bm = pkg.rk4_batch(P, T(xr.reshape(4, 1), dtype), T([ur], dtype), 0.01, fext=fext)
        xo, Ao, Bo = orc.rk4(P, xr.astype(float), float(ur), 0.01, fb, fm)
        # (with v_mu at its 1e-6 floor the friction slope makes entries of A ~ 1e4: tolerances scale with the term size)
        sa = max(1.0, np.abs(Ao).max())
        assert np.abs(N_(xn)[:, 0] - xo).max() < (1e-13 if dtype == torch.float64 else 2e-6) * max(1.0, np.abs(xo).max()), tag
        assert np.abs(N_(A)[:, :, 0] - Ao).max() < (1e-13 if dtype == torch.float64 else 2e-5) * sa, tag
        assert np.abs(N_(Bm)[:, 0] - Bo).max() < (1e-13 if dtype == torch.float64 else 2e-6) * sa, tag
    print("branch points %s: worst f %.2e, worst J %.2e (relative to term size)" % (dtype, worst_f, worst_j))


@pytest.mark.parametrize("dtype", DTYPES)
def test_simulator_across_the_bumper_and_the_cut(pkg, orc, dtype):
    """Plant steps (10 sub-steps, simulator.cc:18-22) from states that cross a bumper edge and the theta cut inside
    the step, fp64 and fp32, against the oracle at the same rounded inputs."""
    npdt = _np_dtype(dtype)
    rng = np.random.default_rng(21)
    B = 256
    st = np.zeros((4, B), dtype=npdt)
    st[0] = np.where(rng.random(B) < 0.5, 0.8, -0.8) + rng.uniform(-2e-3, 2e-3, B)   # at the bumper edge, moving across
    st[2] = rng.uniform(-1, 1, B)
    st[1] = np.where(rng.random(B) < 0.5, np.pi, -np.pi) + rng.uniform(-1e-2, 1e-2, B)   # at the cut, rotating across
    st[3] = rng.uniform(-6, 6, B)
    st[0, :4] = [0.8, -0.8, np.nextafter(npdt(0.8), npdt(1)), -np.nextafter(npdt(0.8), npdt(1))]
    st[1, :4] = [np.pi, -np.pi, np.pi, -np.pi]
    st = st.astype(npdt)
    u = rng.uniform(-20, 20, B).astype(npdt)
    sim = pkg.BatchSimulator(B, dtype=dtype, device=0)
    sim.set_state(T(st, dtype))
    sim.step(DYN_TEST, 0.01, T(u, dtype))
    got = N_(sim.get_state()).astype(float)
    tol = 1e-12 if dtype == torch.float64 else 2e-5
    for b in range(B):
        o = orc.Simulator()
        o.set_state(st[:, b].astype(float))
        o.step(DYN_TEST, 0.01, float(u[b]))
        want = o.get_state()
        d = got[:, b] - want
        d[1] = (d[1] + np.pi) % (2 * np.pi) - np.pi       # a state within rounding of the cut may land on either side
        assert np.abs(d).max() < tol, (b, st[:, b], got[:, b], want)
    assert (got[1] > -np.pi - 1e-6).all() and (got[1] <= np.pi + 1e-6).all()


# ------------------------------------------------------------------------------------------------
# a9: iteration-resolved parity
# ------------------------------------------------------------------------------------------------
def test_iteration_resolved_parity_config2(pkg, orc):
    """BASELINE configs[1] (4096 random states, N = 40, fp64) with max_iterations = 1 ... 5: after ONE iteration every
    lane is within 2e-9 of the oracle and 99 % within 2e-10 (one linearisation, one structured QP -- state elimination +
    4x4 Schur complement with one refinement step -- against the oracle's dense KKT solve with partial pivoting, one
    line search: measured worst 2.2e-10, 99th percentile 1.9e-11, the same as between the double oracle and its
    extended-precision build; without the refinement step the worst lane was 1.9e-8); the worst-lane error then grows by a measured factor per iteration and stays below 1e-5 through
    iteration 5 (measured 1.6e-8).  Round 1 quoted "rounding differences grow ~30x per iteration on unconverged
    lanes"; measured here: 2-5x per iteration for the worst lane, 1.3-3.5x for the 99th percentile (bounded at 20x / 10x)."""
    rng = np.random.default_rng(0)
    x0 = random_states(rng, 4096)
    worst, p99 = [], []
    for k in range(1, 6):
        over = dict(NO_TOL, max_iterations=k)
        opt = pkg.BatchOptimization(pkg.default_params(**over), max_batch=4096, dtype=torch.float64, device=0)
        out = opt.step(T(x0), DYN_UI, 0.0)
        u_cpu, _, st_cpu, it_cpu, _ = orc.step_batch_cold(orc.default_opt_params(**over), DYN_UI, 0.0, x0)
        err = np.abs(N_(out.u) - u_cpu).max(axis=0)
        assert (N_(out.status) == st_cpu).all() and (N_(out.iterations) == it_cpu).all(), k
        worst.append(err.max())
        p99.append(np.quantile(err, 0.99))
    growth_w = [worst[i + 1] / max(worst[i], 1e-16) for i in range(4)]
    growth_p = [p99[i + 1] / max(p99[i], 1e-16) for i in range(4)]
    print("iteration-resolved |du| worst: %s  p99: %s" % (["%.1e" % w for w in worst], ["%.1e" % w for w in p99]))
    print("growth per iteration, worst lane: %s  p99: %s" % (["%.1f" % g for g in growth_w], ["%.1f" % g for g in growth_p]))
    assert worst[0] < 2e-9 and p99[0] < 2e-10, (worst, p99)
    assert max(worst) < 1e-6, worst
    assert np.exp(np.mean(np.log(growth_w))) < 20.0 and np.exp(np.mean(np.log(growth_p))) < 10.0


def _lane_history(pkg, orc, over, dyn, sp, x0_lane, kmax):
    """Per iteration count k = 1..kmax: (|du| between GPU and oracle, GPU decisions, oracle decisions, oracle |c|_1)."""
    hist = []
    for k in range(1, kmax + 1):
        ov = dict(over, max_iterations=k)
        opt = pkg.BatchOptimization(pkg.default_params(**ov), max_batch=1, dtype=torch.float64, device=0)
        out = opt.step(T(x0_lane.reshape(4, 1)), dyn, sp, want_stats=True)
        o = orc.Optimization(orc.default_opt_params(**ov)).step(x0_lane, dyn, sp)
        so = o.solver_outputs
        gpu_dec = (int(out.status[0]), int(out.iterations[0]), int(out.ls_evals[0]))
        cpu_dec = (int(so.termination_state), int(so.iterations), int(so.line_search_evals))
        hist.append((float(np.abs(N_(out.u)[:, 0] - o.u).max()), gpu_dec, cpu_dec, float(so.final_eq_l1)))
    return hist


def test_no_lane_of_the_fuzz_cases_is_dropped(pkg, orc):
    """Round 1 tolerated a few percent of the fuzz lanes beyond 1e-5 and round 2 localised where each departed; since
    round 2's two fixes none does.  This test pins that: over the 32 fuzz configurations every lane agrees with the
    oracle in control sequence (1e-5), termination state and iteration count -- and should one ever reappear, its
    history (agreement iteration by iteration up to the departure) is printed for the failure message."""
    from test_gpu_parity import _random_case
    dropped = []
    for seed in range(32):
        rng = np.random.default_rng(1000 + seed)
        over, dyn, sp = _random_case(rng)
        B = 96
        x0 = random_states(rng, B)
        x0[1, ::2] = np.pi / 2 + rng.uniform(-0.4, 0.4, B // 2)
        opt = pkg.BatchOptimization(pkg.default_params(**over), max_batch=B, dtype=torch.float64, device=0)
        out = opt.step(T(x0), dyn, sp)
        u_cpu, _, st_cpu, it_cpu, _ = orc.step_batch_cold(orc.default_opt_params(**over), dyn, sp, x0)
        err = np.abs(N_(out.u) - u_cpu).max(axis=0)
        bad = np.nonzero((err > 1e-5) | (N_(out.status) != st_cpu) | (N_(out.iterations) != it_cpu))[0]
        for b in bad[:2]:
            dropped.append((seed, int(b), float(err[b]),
                            _lane_history(pkg, orc, over, dyn, sp, x0[:, b], int(over["max_iterations"]))))
    assert not dropped, dropped


# ------------------------------------------------------------------------------------------------
# warm start is per problem (one flag per handle made a grown batch shift stale workspace)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("pipeline", ["fused", "split"])
def test_batch_grows_between_steps(pkg, orc, pipeline):
    """Step B = 40, then B = 100 on the same handle: problems 0..39 warm-start from their previous solution
    (optimization.cc:50-57), problems 40..99 have none and start from the sinusoid guess (:58-68) with u_prev = 0,
    exactly as 100 separate Optimization objects of which 40 have stepped before."""
    rng = np.random.default_rng(77)
    over = dict(max_iterations=4)
    x_a, x_b = random_states(rng, 40), random_states(rng, 100)
    x_a[1] = np.pi / 2 + rng.uniform(-0.3, 0.3, 40)
    x_b[1] = np.pi / 2 + rng.uniform(-0.3, 0.3, 100)
    opt = pkg.BatchOptimization(pkg.default_params(**over), max_batch=128, dtype=torch.float64, device=0)
    opt.set_pipeline(pipeline)
    assert opt.previous_solution_batch() == 0
    opt.step(T(x_a), DYN_UI, 0.0)
    assert opt.previous_solution_batch() == 40
    out = opt.step(T(x_b), DYN_UI, 0.0, want_guess=True)
    assert opt.previous_solution_batch() == 100
    u_gpu, guess = N_(out.u), N_(out.guess)
    p = orc.default_opt_params(**over)
    for b in range(100):
        o = orc.Optimization(p)
        if b < 40:
            o.step(x_a[:, b], DYN_UI, 0.0)
        r = o.step(x_b[:, b], DYN_UI, 0.0)
        # a cold guess is the sinusoid rolled out (1e-12); a warm one is the shifted previous SOLUTION (solver accuracy)
        np.testing.assert_allclose(guess[:, b], r.guess, rtol=0, atol=(1e-6 if b < 40 else 1e-12),
                                   err_msg="guess of problem %d" % b)
        assert np.abs(u_gpu[:, b] - r.u).max() < 1e-5, b
    # a smaller step afterwards keeps the others' previous solutions; reset drops all of them
    opt.step(T(x_a[:, :10]), DYN_UI, 0.0)
    assert opt.previous_solution_batch() == 100
    opt.reset()
    assert opt.previous_solution_batch() == 0 and not opt.has_previous_solution()


# ------------------------------------------------------------------------------------------------
# the reference's own integration tests (optimization/integration_test.cc:82-175) on the GPU path
# ------------------------------------------------------------------------------------------------
def _rk4_chain(pkg, params, x, steps, dt, dtype, fext_of_step=None):
    """`steps` Jacobian-free RK4 steps (cpmpc_rk4_batch) of a [4, B] state tensor, u = 0."""
    u = torch.zeros(x.shape[1], dtype=dtype, device=DEV)
    for i in range(steps):
        x = pkg.rk4_batch(params, x, u, dt, fext=fext_of_step(i) if fext_of_step else None, jacobians=False)
    return x


@pytest.mark.parametrize("dtype,tol_v,tol_w", [(torch.float64, 1e-6, 1e-4), (torch.float32, 1e-4, 1e-3)])
def test_friction_dissipation_like_reference(pkg, dtype, tol_v, tol_w):
    """TestFrictionDissipation (integration_test.cc:82-103): 20 000 steps from a level pole with mu_b = 0.5, no
    control: the velocities die out (reference tolerances 1e-6 / 1e-4 in fp64; fp32 reported at 1e-4 / 1e-3)."""
    p = [1.0, 0.5, 0.4, 9.81, 0.5, 0.1, 0.0, 0.0, 0.0]
    x = _rk4_chain(pkg, p, T(np.zeros((4, 1)), dtype), 20000, 0.01, dtype)
    x = N_(x)[:, 0]
    assert abs(x[2]) < tol_v and abs(x[3]) < tol_w, x


@pytest.mark.parametrize("dtype,tol_v,tol_w", [(torch.float64, 1e-6, 3e-5), (torch.float32, 1e-4, 1e-3)])
def test_drag_dissipation_like_reference(pkg, dtype, tol_v, tol_w):
    """TestDragDissipation (integration_test.cc:105-125): 10 000 steps from theta = -pi with c_d = 5."""
    p = [0.8, 0.1, 0.4, 9.81, 0.01, 0.1, 5.0, 0.0, 0.0]
    x0 = np.array([[0.0], [-np.pi], [0.0], [0.0]])
    x = N_(_rk4_chain(pkg, p, T(x0, dtype), 10000, 0.01, dtype))[:, 0]
    assert abs(x[2]) < tol_v and abs(x[3]) < tol_w, x


def test_external_force_symmetry_like_reference(pkg):
    """TestExternalForceSymmetry (integration_test.cc:127-175): +-5 N on the base for the first 500 of 3 000 steps of
    1 ms from the hanging pole: mirror-symmetric final states to 1e-12 (both signs in one batch of two)."""
    p = [1.0, 0.1, 0.25, 9.81, 0.1, 0.1, 0.0, 0.0, 0.0]
    finals = []
    for sign in (+1.0, -1.0):
        x0 = np.array([[0.0], [-np.pi / 2], [0.0], [0.0]])
        x = _rk4_chain(pkg, p, T(x0), 3000, 0.001, torch.float64,
                       fext_of_step=lambda i, s=sign: [s * 5.0, 0.0, 0.0, 0.0] if i < 500 else None)
        finals.append(N_(x)[:, 0])
    left, right = finals
    assert left[0] > 0 and right[0] < 0
    assert abs(left[0] + right[0]) < 1e-12
    assert abs(left[2] + right[2]) < 1e-12
    assert abs((-np.pi / 2 - left[1]) - (right[1] + np.pi / 2)) < 1e-12
    assert abs(left[3] + right[3]) < 1e-12
