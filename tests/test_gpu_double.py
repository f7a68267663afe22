"""BASELINE config 5 on the GPU: cart + double pendulum (6 states), through the C-ABI, against the oracle
and the independent golden vectors.  The reference has no executable counterpart (parity unpinned by the
reference; see oracle/cpmpc_oracle.h)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
DEV = "cuda:0"
DYN = [1.0, 0.1, 0.1, 0.25, 0.2, 9.81]
OVER = dict(u_guess_sinusoid_amplitude=0.0)


def T(a, dtype=torch.float64):
    return torch.tensor(np.ascontiguousarray(a), dtype=dtype, device=DEV)


def N_(t):
    return t.detach().cpu().numpy()


def near_upright(rng, B, spread=0.15):
    return np.stack([rng.uniform(-0.3, 0.3, B), np.pi / 2 + rng.uniform(-spread, spread, B),
                     np.pi / 2 + rng.uniform(-spread, spread, B), rng.uniform(-0.3, 0.3, B),
                     rng.uniform(-0.5, 0.5, B), rng.uniform(-0.5, 0.5, B)])


def test_double_dynamics_golden(pkg):
    with open(os.path.join(GOLDEN, "double_golden.json")) as fh:
        cases = json.load(fh)["cases"]
    for c in cases:
        f, Jx, Ju = pkg.dynamics_batch(c["params"], T(np.array(c["x"]).reshape(6, 1)), T([c["u"]]), model="double")
        for got, want in ((N_(f)[:, 0], c["f"]), (N_(Jx)[:, :, 0], c["J_x"]), (N_(Ju)[:, 0], c["J_u"])):
            want = np.asarray(want)
            assert np.abs(got - want).max() / max(1.0, np.abs(want).max()) < 1e-11


def test_double_rk4_and_sim_match_oracle(pkg, orc):
    rng = np.random.default_rng(17)
    B = 200
    x = np.stack([rng.uniform(-1, 1, B), rng.uniform(-4, 4, B), rng.uniform(-4, 4, B), rng.uniform(-2, 2, B),
                  rng.uniform(-6, 6, B), rng.uniform(-6, 6, B)])
    u = rng.uniform(-20, 20, B)
    xn, A, Bm = pkg.rk4_batch(DYN, T(x), T(u), 0.01, model="double")
    xn2 = pkg.rk4_batch(DYN, T(x), T(u), 0.01, jacobians=False, model="double")
    xn, A, Bm, xn2 = N_(xn), N_(A), N_(Bm), N_(xn2)
    for b in range(B):
        xo, Ao, Bo = orc.rk4_model("double", DYN, x[:, b], u[b], 0.01)
        np.testing.assert_allclose(xn[:, b], xo, rtol=0, atol=1e-11)
        np.testing.assert_allclose(xn2[:, b], xo, rtol=0, atol=1e-11)
        np.testing.assert_allclose(A[:, :, b], Ao, rtol=0, atol=1e-11)
        np.testing.assert_allclose(Bm[:, b], Bo, rtol=0, atol=1e-12)
    sim = pkg.BatchSimulator(B, dtype=torch.float64, device=0, model="double")
    assert tuple(sim.get_state().shape) == (6, B)
    sim.set_state(T(x))
    sim.step(DYN, 0.01, T(u))
    got = N_(sim.get_state())
    for b in range(0, B, 7):
        np.testing.assert_allclose(got[:, b], orc.sim_step_model("double", DYN, 0.01, u[b], x[:, b]), rtol=0, atol=1e-10)


@pytest.mark.parametrize("N,sp", [(40, 10), (40, 5), (20, 10)])
def test_double_linearize_matches_oracle(pkg, orc, N, sp):
    S = N // sp + 1
    rng = np.random.default_rng(N + sp)
    B = 66
    opt = pkg.BatchOptimization(pkg.default_params(window_length=N, state_spacing=sp), max_batch=B,
                                dtype=torch.float64, device=0, model="double")
    assert opt.dim == 6 * S + N and opt.nx == 6
    z = np.concatenate([np.tile(near_upright(rng, B, 1.0), (S, 1)) + rng.normal(0, 0.05, (6 * S, B)),
                        rng.uniform(-10, 10, (N, B))])
    c, Phi, Gam = opt.linearize(T(z), DYN)
    c, Phi, Gam = N_(c), N_(Phi), N_(Gam)
    for b in range(0, B, 5):
        for s in range(S - 1):
            vars_ = np.concatenate([z[6 * s:6 * s + 6, b], z[6 * (s + 1):6 * (s + 1) + 6, b],
                                    z[6 * S + s * sp:6 * S + (s + 1) * sp, b]])
            err, J = orc.shooting_constraint_model("double", DYN, sp, 0.01, vars_)
            np.testing.assert_allclose(c[6 * s:6 * s + 6, b], err, rtol=0, atol=1e-10)
            np.testing.assert_allclose(Phi[s, :, :, b], J[:, :6], rtol=0, atol=1e-9)
            np.testing.assert_allclose(Gam[s * sp:(s + 1) * sp, :, b].T, J[:, 12:], rtol=0, atol=1e-10)


def _step_vs_oracle(pkg, orc, over, x0, set_point, pipeline="auto"):
    B = x0.shape[1]
    opt = pkg.BatchOptimization(pkg.default_params(**over), max_batch=B, dtype=torch.float64, device=0, model="double")
    opt.set_pipeline(pipeline)
    assert pipeline == "auto" or opt.pipeline() == pipeline
    out = opt.step(T(x0), DYN, set_point)
    u_cpu, pred_cpu, st_cpu, it_cpu, _ = orc.step_batch_cold(orc.default_opt_params(**over), DYN, set_point, x0,
                                                             want_pred=True, model="double")
    assert tuple(out.predicted_states.shape) == (40, 6, B)
    ok = (N_(out.status) == st_cpu) & (N_(out.iterations) == it_cpu)
    err = np.abs(N_(out.u) - u_cpu).max(axis=0)
    perr = np.abs(N_(out.predicted_states) - pred_cpu).max(axis=(0, 1))
    return out, ok, err, perr


@pytest.mark.parametrize("pipeline", ["split", "fused"])
def test_double_step_parity_fixed_iterations(pkg, orc, pipeline):
    """5 SQP iterations, exits disabled, states up to 0.15 rad from upright (most of these do not converge
    within the 0.4 s horizon): every lane within 1e-5 of the oracle on u (fp64).
    Until round 2 one lane of the 320 ended 8.0 away, identically in both GPU pipelines, and was tolerated as
    "cancellation".  The extended-precision build of the oracle (oracle/cpmpc_oracle_ld.c) showed the oracle was
    right to 4e-10 and the GPU was not: on that problem the full step and the half step diverge, the oracle's
    rollout ended finite-and-astronomic (|c|_1 = 5e91: step cut to a tenth) while the kernels' ended NaN (step
    halved), so the five trials ran through different step lengths.  The rule now says what it always meant -- a
    trial whose merit is not finite is an overlong step, lower safeguard -- in the oracle and in both pipelines."""
    rng = np.random.default_rng(5)
    x0 = near_upright(rng, 320)
    over = dict(OVER, max_iterations=5, relative_exit_tol=0.0, absolute_first_derivative_tol=0.0)
    out, ok, err, perr = _step_vs_oracle(pkg, orc, over, x0, 0.05, pipeline)
    assert ok.all()
    assert err.max() < 1e-5 and np.median(err) < 1e-7, np.sort(err)[-5:]
    assert perr.max() < 1e-5


SOFT = dict(state_spacing=5, th_final_cost_weight=200.0, th_dot_final_cost_weight=20.0, b_x_dot_final_cost_weight=20.0)


@pytest.mark.parametrize("pipeline", ["split", "fused"])
@pytest.mark.parametrize("over,min_conv", [(dict(max_iterations=10), 0.15), (dict(max_iterations=10, **SOFT), 0.95)])
def test_double_step_parity_with_exits(pkg, orc, over, min_conv, pipeline):
    """Exits enabled: same termination state and iteration count, controls within 1e-5 on every lane.
    With the default hard terminal equalities only part of the batch converges inside 10 iterations; with
    soft terminal weights (the configuration that balances robustly) nearly all of it does."""
    rng = np.random.default_rng(6)
    B = 256
    x0 = near_upright(rng, B, 0.05)
    x0[0] *= 0.2
    x0[3:] *= 0.2
    out, ok, err, perr = _step_vs_oracle(pkg, orc, dict(OVER, **over), x0, 0.02, pipeline)
    assert ok.all()
    assert (N_(out.final_eq_l1) < 1e-4).mean() >= min_conv
    assert err.max() < 1e-5 and perr.max() < 1e-5
    assert len(set(N_(out.status).tolist())) >= 2


def test_double_closed_loop_and_warm_start(pkg, orc):
    """64 double-pendulum controllers balance for 2 s (warm start + plant on the GPU); for the first 20
    ticks every lane is followed by its own oracle controller."""
    rng = np.random.default_rng(9)
    B = 64
    x0 = near_upright(rng, B, 0.05)
    x0[0] *= 0.2
    x0[3:] *= 0.2
    over = dict(OVER, max_iterations=10, **SOFT)
    opt = pkg.BatchOptimization(pkg.default_params(**over), max_batch=B, dtype=torch.float64, device=0, model="double")
    sim = pkg.BatchSimulator(B, dtype=torch.float64, device=0, model="double")
    sim.set_state(T(x0))
    o_opt = [orc.Optimization(orc.default_opt_params(**over), model="double") for _ in range(B)]
    o_state = [x0[:, b].copy() for b in range(B)]
    worst = 0.0
    for t in range(200):
        out = opt.step(sim.get_state().clone(), DYN, 0.0, want_predicted=False)
        sim.step(DYN, 0.01, out.u[0].contiguous())
        if t < 20:
            u_gpu, st_gpu = N_(out.u), N_(out.status)
            for b in range(B):
                o = o_opt[b].step(o_state[b], DYN, 0.0)
                assert st_gpu[b] == o.solver_outputs.termination_state, (t, b)
                worst = max(worst, np.abs(u_gpu[:, b] - o.u).max())
                o_state[b] = orc.sim_step_model("double", DYN, 0.01, o.u[0], o_state[b])
    assert worst < 1e-5
    s = N_(sim.get_state())
    assert np.abs(s[1] - np.pi / 2).max() < 5e-3 and np.abs(s[2] - np.pi / 2).max() < 5e-3  # soft terminal costs
    assert np.abs(s[3:]).max() < 2e-2


def test_config5_full_size(pkg, orc):
    """BASELINE configs[4]: double pendulum, batch = 65536, N = 40, one GPU, the default pipeline (fused in both dtypes
    since round 5).  Half of the batch starts near upright (both poles within 0.15 rad), the other half anywhere within
    0.5 rad of upright with larger velocities (the round-4 form of this test sampled 256 near-upright lanes).
    fp32 run: finite, clamped, deterministic; fp64 run: 4 096 sampled lanes -- the first and last waves of each half and an
    even spread -- within 1e-5 of the oracle on u, same termination state and iteration count."""
    rng = np.random.default_rng(50)
    B = 65536
    x0 = near_upright(rng, B)
    h = B // 2
    x0[1:3, h:] = np.pi / 2 + rng.uniform(-0.5, 0.5, (2, B - h))
    x0[3:, h:] = rng.uniform(-1.0, 1.0, (3, B - h))
    over = dict(OVER, max_iterations=5, relative_exit_tol=0.0, absolute_first_derivative_tol=0.0)
    opt32 = pkg.BatchOptimization(pkg.default_params(**over), max_batch=B, dtype=torch.float32, device=0, model="double")
    assert opt32.pipeline() == "fused"
    o1 = opt32.step(T(x0, torch.float32), DYN, 0.0)
    u1, st1 = o1.u.clone(), o1.status.clone()
    assert torch.isfinite(u1).all() and u1.abs().max().item() <= 300.0
    assert (o1.predicted_states[:, 1:3].abs() <= np.pi + 1e-5).all()
    opt32.reset()
    o2 = opt32.step(T(x0, torch.float32), DYN, 0.0)
    assert torch.equal(o2.u, u1) and torch.equal(o2.status, st1)
    del opt32
    opt64 = pkg.BatchOptimization(pkg.default_params(**over), max_batch=B, dtype=torch.float64, device=0, model="double")
    assert opt64.pipeline() == "fused"
    out = opt64.step(T(x0), DYN, 0.0)
    samp = np.unique(np.concatenate([np.arange(64), np.arange(h - 64, h + 64), np.arange(B - 64, B),
                                     np.linspace(0, B - 1, 3900).astype(np.int64)]))
    assert samp.size >= 4096 and (samp >= h).sum() > 1900
    u_cpu, _, st_cpu, it_cpu, _ = orc.step_batch_cold(orc.default_opt_params(**over), DYN, 0.0, x0[:, samp], model="double")
    err = np.abs(N_(out.u)[:, samp] - u_cpu).max(axis=0)
    assert (N_(out.status)[samp] == st_cpu).all() and (N_(out.iterations)[samp] == it_cpu).all()
    if not (err < 1e-5).all():   # who moved?  (oracle/cpmpc_oracle_ld.c)
        off = np.where(err >= 1e-5)[0]
        u_ld, _, _, _, _ = orc.step_batch_cold_ld(orc.default_opt_params(**over), DYN, 0.0, x0[:, samp[off]], model="double")
        e_g = np.abs(N_(out.u)[:, samp[off]] - u_ld).max(axis=0)
        e_c = np.abs(u_cpu[:, off] - u_ld).max(axis=0)
        raise AssertionError("lanes beyond 1e-5: %s; GPU vs extended %s, oracle vs extended %s" % (samp[off], e_g, e_c))
    e32 = np.abs(N_(u1.double())[:, samp] - u_cpu).max(axis=0)
    far = samp >= h
    print("config 5: fp64 |du| max %.2e median %.2e (near upright max %.2e, within 0.5 rad max %.2e);  fp32 median %.2e"
          % (err.max(), np.median(err), err[~far].max(), err[far].max(), np.median(e32)))


@pytest.mark.parametrize("N,sp,refine", [(40, 10, None), (40, 10, True), (40, 5, None), (20, 10, None), (20, 5, None), (40, 8, None),
                                          (40, 4, None), (40, 20, None), (30, 6, None), (30, 3, None)])
def test_double_fused_layouts_across_shapes(pkg, orc, N, sp, refine):
    """The LDS layouts of the double 6-state fused kernel (round 5) over every compiled (intervals, spacing) pair and the
    run-time-spacing kernel: slim (u, du, 48-byte Gamma columns; 1/d_k in registers) for spacings up to 10, the four-array
    layout for the REFINE instantiation and for spacing 20 (100 KB per wave: AUTO would take the split pipeline, here the fused
    one is forced), dynamic LDS for spacings without a specialisation.  192 lanes within 0.15 rad of upright, 4 iterations,
    exits disabled: same termination state and iteration count as the oracle, controls within 1e-5 on every lane; the fused
    and the split pipeline agree to 1e-7."""
    rng = np.random.default_rng(100 * N + sp)
    B = 192
    x0 = near_upright(rng, B)
    over = dict(OVER, window_length=N, state_spacing=sp, max_iterations=4, relative_exit_tol=0.0, absolute_first_derivative_tol=0.0)
    u_cpu, _, st_cpu, it_cpu, _ = orc.step_batch_cold(orc.default_opt_params(**over), DYN, 0.03, x0, model="double")
    res = {}
    for pipeline in ("fused", "split"):
        opt = pkg.BatchOptimization(pkg.default_params(**over), max_batch=B, dtype=torch.float64, device=0, model="double",
                                    refine_qp=refine)
        opt.set_pipeline(pipeline)
        assert opt.pipeline() == pipeline
        out = opt.step(T(x0), DYN, 0.03)
        u = N_(out.u)
        err = np.abs(u - u_cpu).max(axis=0)
        assert (N_(out.status) == st_cpu).all() and (N_(out.iterations) == it_cpu).all(), (pipeline, N, sp)
        assert err.max() < 1e-5, (pipeline, N, sp, np.sort(err)[-3:])
        res[pipeline] = u
    assert np.abs(res["fused"] - res["split"]).max() < 1e-7


def test_double_float_handles_carry_the_qp_in_double_by_default(pkg, orc):
    """The float kernels of the 6-state model carry the whole terminal part of the QP in double by default
    (CPMPC_CREATE_WIDE_QP, round 5): without it four of five float solves are off by more than 0.01 N after five iterations
    (the QP's precision, not single precision as such: the float build of the CPU check is not).  4 096 near-upright cold
    starts against the double check (max |du| per problem): default handle >= 97 % within 1e-2 and median <= 1.5e-3 (measured
    99.1 %, 5.7e-4; the float CPU check 99.2 %, 5.2e-4); with the option forced off < 50 % (measured 19.8 %)."""
    rng = np.random.default_rng(1005)
    B = 4096
    x0 = near_upright(rng, B)
    over = dict(OVER, max_iterations=5, relative_exit_tol=0.0, absolute_first_derivative_tol=0.0)
    u64, _, st64, _, _ = orc.step_batch_cold(orc.default_opt_params(**over), DYN, 0.0, x0, model="double")
    u32, st32, _, _, _ = orc.step_batch_cold_f32(orc.default_opt_params(**over), DYN, 0.0, x0, model="double")
    e_cpu = np.abs(u32 - u64).max(axis=0)
    res = {}
    for wide in (None, False):
        opt = pkg.BatchOptimization(pkg.default_params(**over), max_batch=B, dtype=torch.float32, device=0, model="double", wide_qp=wide)
        assert opt.wide_qp == (wide is None)
        o = opt.step(T(x0, torch.float32), DYN, 0.0)
        err = np.abs(N_(o.u.double()) - u64).max(axis=0)
        assert (N_(o.status) == st64).all()
        res[wide] = (float((err < 1e-2).mean()), float(np.median(err)))
    print("within 1e-2 / median: default (wide) %.3f %.2e, forced off %.3f %.2e, float CPU check %.3f %.2e"
          % (res[None] + res[False] + ((e_cpu < 1e-2).mean(), np.median(e_cpu))))
    assert res[None][0] >= 0.97 and res[None][1] <= 1.5e-3
    assert (e_cpu < 1e-2).mean() >= 0.97
    assert res[False][0] < 0.5
    # a spacing without a compiled specialisation (N = 30, spacing 6: the run-time-spacing kernel) has it as well
    over6 = dict(over, window_length=30, state_spacing=6)
    u64, _, st64, _, _ = orc.step_batch_cold(orc.default_opt_params(**over6), DYN, 0.0, x0[:, :1024], model="double")
    opt = pkg.BatchOptimization(pkg.default_params(**over6), max_batch=1024, dtype=torch.float32, device=0, model="double")
    assert opt.wide_qp and opt.pipeline() == "fused"
    o = opt.step(T(x0[:, :1024], torch.float32), DYN, 0.0)
    err = np.abs(N_(o.u.double()) - u64).max(axis=0)
    assert (N_(o.status) == st64).all() and (err < 1e-2).mean() >= 0.97, ((err < 1e-2).mean(), np.median(err))


def test_double_float_closed_loop_balances_without_solver_failures(pkg, orc):
    """The 6-state model in single precision, closed loop (pkg.ClosedLoop, the float handle's default: QP in double): 2 048
    controllers balance both poles for 3 s from within 0.05 rad of upright with the soft terminal weights of
    test_double_closed_loop_and_warm_start -- never QP_INDEFINITE / MAX_LAMBDA / NON_FINITE (optimization_test.cc:44-46 for
    this model), poles upright within 1e-2 rad and the cart at rest at the end."""
    rng = np.random.default_rng(19)
    B = 2048
    x0 = near_upright(rng, B, 0.05)
    x0[0] *= 0.2
    x0[3:] *= 0.2
    over = dict(OVER, max_iterations=10, **SOFT)
    loop = pkg.ClosedLoop(pkg.default_params(**over), B, dtype=torch.float32, device=0, model="double")
    assert loop.opts[0].wide_qp
    loop.set_state(T(x0, torch.float32))
    bad = torch.zeros((), dtype=torch.int64, device=DEV)
    fails = torch.tensor([orc.TERM_QP_INDEFINITE, orc.TERM_MAX_LAMBDA, orc.TERM_NON_FINITE], dtype=torch.int32, device=DEV)
    for _ in range(300):
        loop.tick(DYN, 0.0)
        bad += torch.isin(loop.status(), fails).sum()
    s = N_(loop.state().double())
    assert int(bad.item()) == 0
    assert np.abs(s[1] - np.pi / 2).max() < 1e-2 and np.abs(s[2] - np.pi / 2).max() < 1e-2, (np.abs(s[1] - np.pi / 2).max(), np.abs(s[2] - np.pi / 2).max())
    assert np.abs(s[3:]).max() < 5e-2
