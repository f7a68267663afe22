"""Round 6 on the GPU: CPMPC_CREATE_WIDE_QP in the split pipeline, the per-handle long-horizon
status (item 4) and the 6-state model's structure exploitation held to the unstructured build."""
import numpy as np
import pytest

from conftest import DYN_UI, random_states

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
DEV = "cuda:0"
DYN_D = [1.0, 0.1, 0.1, 0.25, 0.2, 9.81]


def T(a, dtype=torch.float64):
    return torch.tensor(np.ascontiguousarray(a), dtype=dtype, device=DEV)


def N_(t):
    return t.detach().cpu().numpy()


def near_upright(rng, B, spread=0.15):
    return np.stack([rng.uniform(-0.3, 0.3, B), np.pi / 2 + rng.uniform(-spread, spread, B),
                     np.pi / 2 + rng.uniform(-spread, spread, B), rng.uniform(-0.3, 0.3, B),
                     rng.uniform(-0.5, 0.5, B), rng.uniform(-0.5, 0.5, B)])


def test_float_6state_handles_keep_the_wide_qp_in_the_split_pipeline(pkg, orc):
    """A float 6-state handle carries the QP's terminal part in double by default; until round 6 a handle that AUTO or
    set_pipeline() sent to the SPLIT pipeline lost that silently (four of five cold starts then end more than 0.01 N from the
    double check).  Now qp_ls_kernel has the wide form too: (a) the default shape forced to split, (b) a shape the fused kernel
    is not built for (N = 30, spacing 10: three intervals), which AUTO sends to split -- both >= 97 % within 1e-2 of the double
    CPU check with the same statuses, like the fused pipeline and like the float CPU check; forced off: < 50 %."""
    rng = np.random.default_rng(1006)
    B = 4096
    x0 = near_upright(rng, B)
    over = dict(u_guess_sinusoid_amplitude=0.0, max_iterations=5, relative_exit_tol=0.0, absolute_first_derivative_tol=0.0)
    u64, _, st64, _, _ = orc.step_batch_cold(orc.default_opt_params(**over), DYN_D, 0.0, x0, model="double")
    u32, _, _, _, _ = orc.step_batch_cold_f32(orc.default_opt_params(**over), DYN_D, 0.0, x0, model="double")
    e_cpu = np.abs(u32 - u64).max(axis=0)
    res = {}
    for pipe in ("fused", "split"):
        for wide in (None, False):
            opt = pkg.BatchOptimization(pkg.default_params(**over), max_batch=B, dtype=torch.float32, device=0, model="double", wide_qp=wide)
            opt.set_pipeline(pipe)
            assert opt.pipeline() == pipe and opt.wide_qp == (wide is None)   # the option is the handle's, whatever the pipeline
            o = opt.step(T(x0, torch.float32), DYN_D, 0.0)
            err = np.abs(N_(o.u.double()) - u64).max(axis=0)
            assert (N_(o.status) == st64).all()
            res[pipe, wide] = (float((err < 1e-2).mean()), float(np.median(err)))
            opt.close()
    print("within 1e-2 / median:", res, "float CPU check", ((e_cpu < 1e-2).mean(), np.median(e_cpu)))
    assert res["split", None][0] >= 0.97 and res["split", None][1] <= 1.5e-3, res
    assert res["fused", None][0] >= 0.97 and res["split", False][0] < 0.5 and res["fused", False][0] < 0.5, res
    # (b) three shooting intervals: no fused kernel, AUTO takes the split pipeline -- with the option
    over3 = dict(over, window_length=30, state_spacing=10)
    u64, _, st64, _, _ = orc.step_batch_cold(orc.default_opt_params(**over3), DYN_D, 0.0, x0[:, :2048], model="double")
    opt = pkg.BatchOptimization(pkg.default_params(**over3), max_batch=2048, dtype=torch.float32, device=0, model="double")
    assert opt.pipeline() == "split" and opt.wide_qp
    o = opt.step(T(x0[:, :2048], torch.float32), DYN_D, 0.0)
    err = np.abs(N_(o.u.double()) - u64).max(axis=0)
    assert (N_(o.status) == st64).all() and (err < 1e-2).mean() >= 0.97, ((err < 1e-2).mean(), np.median(err))


def test_float_4state_wide_qp_in_the_split_pipeline(pkg, orc):
    """The 4-state float handle with CPMPC_CREATE_WIDE_QP through the split pipeline: the benchmark's cold starts end closer to
    the double check than the plain float handle's (fused, measured: median 2.4e-4 -> 8.3e-5, within 1e-2 93.7 -> 99.4 %)."""
    rng = np.random.default_rng(77)
    B = 4096
    x0 = random_states(rng, B)
    over = dict(max_iterations=5, relative_exit_tol=0.0, absolute_first_derivative_tol=0.0)
    u64, _, st64, _, _ = orc.step_batch_cold(orc.default_opt_params(**over), DYN_UI, 0.0, x0)
    res = {}
    for wide in (False, True):
        opt = pkg.BatchOptimization(pkg.default_params(**over), max_batch=B, dtype=torch.float32, device=0, wide_qp=wide)
        opt.set_pipeline("split")
        assert opt.wide_qp == wide and opt.pipeline() == "split"
        o = opt.step(T(x0, torch.float32), DYN_UI, 0.0)
        err = np.abs(N_(o.u.double()) - u64).max(axis=0)
        res[wide] = (float((err < 1e-2).mean()), float(np.median(err)), float(np.quantile(err, 0.99)))
        opt.close()
    print("4-state split: within 1e-2 / median / p99: plain %s, wide %s" % (res[False], res[True]))
    assert res[True][0] >= 0.985 and res[True][0] >= res[False][0] and res[True][2] < res[False][2], res


def test_horizon_beyond_parity_is_a_per_handle_status(pkg, orc):
    """a horizon beyond cpmpc_max_parity_horizon() is a status of the HANDLE -- cpmpc_horizon_beyond_parity(),
    BatchOptimization.horizon_beyond_parity, a line in solver_summary(), pypendulum.Optimization.horizon_beyond_parity and the
    summary of its steps -- not only one line on stderr per process; such fp64 handles refine the QP by default (one pass with
    residuals from the original data: 34 -> 3 lanes of 8 192 with the kernels at fault at N = 160,
    profiles/r06_long_horizon_probe.json), CPMPC_CREATE_NO_REFINE_QP still switches that off."""
    rng = np.random.default_rng(31)
    x0 = random_states(rng, 64)
    x0[1] = np.pi / 2 + rng.uniform(-0.2, 0.2, 64)
    short = pkg.BatchOptimization(pkg.default_params(), max_batch=64, dtype=torch.float64, device=0)
    assert not short.horizon_beyond_parity and not short.refines_qp
    o = short.step(T(x0), DYN_UI, 0.0)
    assert not o.horizon_beyond_parity and "cpmpc_max_parity_horizon" not in o.solver_summary()
    over = dict(window_length=160, max_iterations=3)
    long_ = pkg.BatchOptimization(pkg.default_params(**over), max_batch=64, dtype=torch.float64, device=0)
    assert long_.horizon_beyond_parity and long_.refines_qp and long_.pipeline() == "split"   # AUTO: two-pass split kernel there
    o = long_.step(T(x0), DYN_UI, 0.0)
    assert o.horizon_beyond_parity and "cpmpc_max_parity_horizon" in o.solver_summary()
    u_cpu, _, st_cpu, _, _ = orc.step_batch_cold(orc.default_opt_params(**over), DYN_UI, 0.0, x0)
    assert (N_(o.status) == st_cpu).all() and np.abs(N_(o.u) - u_cpu).max() < 1e-5   # near-upright starts: every lane
    off = pkg.BatchOptimization(pkg.default_params(**over), max_batch=64, dtype=torch.float64, device=0, refine_qp=False)
    assert off.horizon_beyond_parity and not off.refines_qp
    # the facade and pypendulum
    pyp = pkg.pypendulum()
    op = pyp.OptimizationParams()
    assert not pyp.Optimization(op).horizon_beyond_parity
    op.window_length = 160
    op.max_iterations = 2
    opt = pyp.Optimization(op)
    assert opt.horizon_beyond_parity
    out = opt.step(pyp.SingleCartPoleState(0.0, np.pi / 2 - 0.1, 0.0, 0.0), pyp.SingleCartPoleParams(*DYN_UI), 0.0)
    assert "cpmpc_max_parity_horizon" in out.solver_summary() and len(out.u) == 160
    assert pyp.ShardedOptimization(op, 8, [0, 0]).horizon_beyond_parity
    op.window_length = 40
    assert not pyp.ShardedOptimization(op, 8, [0, 0]).horizon_beyond_parity


def test_wide_qp_is_a_property_of_the_handle_not_of_the_step(pkg):
    """Round 6 tried "wide on cold-start steps only" as the default of 4-state float handles and withdrew it: whether a call
    is a cold start depends on which problems share it, so the same problem would run different kernels in a single handle
    (partly warm call) and in a sharded one (an all-cold shard).  The option is fixed at creation: a default handle's steps --
    cold, warm, after Reset -- are bit for bit the CPMPC_CREATE_NO_WIDE_QP handle's, never the wide one's."""
    rng = np.random.default_rng(5)
    B = 2048
    x0 = T(random_states(rng, B), torch.float32)
    mk = lambda **kw: pkg.BatchOptimization(pkg.default_params(max_iterations=5), max_batch=B, dtype=torch.float32, device=0, **kw)  # noqa: E731
    auto, wide, narrow = mk(), mk(wide_qp=True), mk(wide_qp=False)
    assert (auto.wide_qp, wide.wide_qp, narrow.wide_qp) == (False, True, False)
    for k in range(3):
        x = x0 + 0.01 * k
        u_a, u_n, u_w = (h.step(x, DYN_UI, 0.0).u.clone() for h in (auto, narrow, wide))
        assert torch.equal(u_a, u_n) and not torch.equal(u_a, u_w), k
    auto.reset()
    narrow.reset()
    assert torch.equal(auto.step(x0, DYN_UI, 0.0).u, narrow.step(x0, DYN_UI, 0.0).u)


def test_get_solver_opts_tells_what_a_handle_uses(pkg):
    """the positional constructors read 128 bytes of cpmpc_solver_opts, so a field appended later
    (exit_defect_floor) given to them is not taken -- invisible until now.  cpmpc_get_solver_opts returns the merged options
    of a handle: through cpmpc_create the field keeps its default (2), through cpmpc_create_ex with the full size it is the
    caller's; fields inside the 128 bytes arrive either way."""
    import ctypes as C
    lib = pkg.capi.load()
    params = pkg.default_params()
    mine = pkg.capi.default_solver_opts(full_step_below=3e-5, exit_defect_floor=0.0)
    got = pkg.capi.SolverOpts()
    h = C.c_void_p()
    pkg.capi.check(lib.cpmpc_create(C.byref(params), C.byref(mine), pkg.capi.F32, 64, 0, C.byref(h)))
    pkg.capi.check(lib.cpmpc_get_solver_opts(h, C.byref(got), C.sizeof(got)))
    lib.cpmpc_destroy(h)
    assert got.full_step_below == 3e-5 and got.exit_defect_floor == 2.0
    info = pkg.capi.CreateInfo(struct_size=C.sizeof(pkg.capi.CreateInfo), flags=0, dtype=pkg.capi.F32, model=0, device=0, reserved=0,
                               max_batch=64, params=C.pointer(params), opts=C.pointer(mine), opts_size=C.sizeof(mine))
    pkg.capi.check(lib.cpmpc_create_ex(C.byref(info), C.byref(h)))
    pkg.capi.check(lib.cpmpc_get_solver_opts(h, C.byref(got), C.sizeof(got)))
    assert got.full_step_below == 3e-5 and got.exit_defect_floor == 0.0
    assert lib.cpmpc_get_solver_opts(h, C.byref(got), 100) == pkg.capi.ERR_INVALID_ARG
    lib.cpmpc_destroy(h)
