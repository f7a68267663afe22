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
