"""Converged control sequences pinned by an INDEPENDENT statement of the NLP and an INDEPENDENT solver
(tests/golden/gen_converged_golden.py: SymPy-derived dynamics, numpy shooting constraints, scipy SLSQP + Newton-KKT
polish; no code shared with oracle/ or the kernels).  mini_opt, the reference's solver, is absent, so iterate-level
parity is unreachable -- but a converged solution does not depend on the solver: any correct method that
lands in the same basin reaches the same KKT point of optimization.cc:194-301's problem.

  * the golden points themselves are KKT points of the ORACLE's statement of the problem (orc_problem_eval): the two
    statements of the NLP agree;
  * the oracle's SQP, run from the reference's initial guess until it stops moving, ends at u* --
    to 1e-9, not merely the 1e-5 of the parity bar -- on every near-upright case and on the swing-up cases that land
    in the same basin;
  * the GPU (fp64, both pipelines) does the same.
(Before round 3's full-step rule -- a QP step that is tiny in every component is taken without the merit test -- both
stalled 1e-7 .. 1e-4 short of u*: the l1 merit cannot resolve the decrease of the last steps and rejected them.)"""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

FIXED_POINT = dict(max_iterations=1000, relative_exit_tol=0.0, absolute_first_derivative_tol=0.0)


@pytest.fixture(scope="module")
def cases():
    with open(os.path.join(GOLDEN, "converged_golden.json")) as fh:
        d = json.load(fh)
    # a golden point is one the independent solve certified: KKT residual and defects at rounding level, no clamp active
    good = [c for c in d["cases"] if c["kkt_residual"] < 1e-9 and c["eq_l1"] < 1e-10 and not c["clamp_active"]]
    assert len(d["cases"]) == 128 and len(good) >= 120
    return good


def _kkt_through_oracle(orc, c, z):
    p = orc.default_opt_params(**c["params"])
    r, ceq, J, A = orc.problem_eval(p, c["dyn"], c["x0"], c["set_point"], 0.0, np.array(z))
    g = J.T @ r
    lam = np.linalg.lstsq(A.T, -g, rcond=None)[0]
    return float(np.abs(g + A.T @ lam).max()), float(np.abs(ceq).sum()), 0.5 * float(r @ r)


def test_golden_points_are_kkt_points_of_the_oracles_problem(orc, cases):
    """The independently stated NLP and the oracle's orc_problem_eval (rows, Jacobians) describe the same problem:
    z* satisfies the oracle's constraints to 1e-9, its objective is the recorded one, and the reduced gradient
    vanishes."""
    worst_g = worst_c = 0.0
    for c in cases:
        g, ceq, f = _kkt_through_oracle(orc, c, c["z_star"])
        worst_g, worst_c = max(worst_g, g), max(worst_c, ceq)
        assert abs(f - c["objective"]) <= 1e-9 * (1.0 + c["objective"]), (c["kind"], c["x0"])
    print("golden points through the oracle's problem: worst |grad L| %.2e, worst |c|_1 %.2e" % (worst_g, worst_c))
    assert worst_c < 1e-9 and worst_g < 1e-7


def test_oracle_fixed_point_is_the_independent_optimum(orc, cases):
    errs, objs = [], []
    for c in cases:
        p = orc.default_opt_params(**dict(c["params"], **FIXED_POINT))
        out = orc.Optimization(p).step(np.array(c["x0"]), c["dyn"], c["set_point"])
        errs.append(float(np.abs(out.u - np.array(c["u_star"])).max()))
        _, _, f = _kkt_through_oracle(orc, c, out.z)
        objs.append(f)
    e = np.array(errs)
    up = np.array([c["kind"] == "near-upright" for c in cases])
    hit = e <= 1e-5
    print("oracle fixed point vs u*: near-upright median %.1e worst %.1e; swing-up within 1e-5: %d of %d; all cases within "
          "1e-8: %d of %d" % (np.median(e[up]), e[up].max(), hit[~up].sum(), (~up).sum(), (e <= 1e-8).sum(), e.size))
    assert e[up].max() < 1e-9, sorted(e[up])[-5:]      # every near-upright problem converges TO the independent optimum
    assert hit[~up].sum() >= 0.75 * (~up).sum()
    assert (e[hit] <= 1e-8).mean() >= 0.95             # where it gets there at all it gets there properly
    # a case that misses u* is in another basin or has not settled: its objective differs, or it is still far
    for c, x, h, f in zip(cases, errs, hit, objs):
        if not h:
            assert abs(f - c["objective"]) > 1e-9 * (1.0 + c["objective"]) or x > 1e-5, (c["x0"], x, f, c["objective"])


@pytest.mark.gpu
@pytest.mark.parametrize("pipeline", ["fused", "split"])
def test_gpu_fixed_point_is_the_independent_optimum(pkg, orc, cases, pipeline):
    """GPU fp64 run to its fixed point from the reference's initial guess, against the independently computed optimum:
    how close the specification's SQP gets on the GPU (bounds and measured values below), never far where the oracle's
    gets there."""
    torch = pytest.importorskip("torch")
    by_cfg = {}
    for i, c in enumerate(cases):
        by_cfg.setdefault(json.dumps([c["params"], c["dyn"]], sort_keys=True), []).append(i)
    errs = [None] * len(cases)
    orc_hit = [None] * len(cases)
    for idx in by_cfg.values():
        c0 = cases[idx[0]]
        x0 = np.array([cases[i]["x0"] for i in idx]).T.copy()
        over = dict(c0["params"], **FIXED_POINT)
        opt = pkg.BatchOptimization(pkg.default_params(**over), max_batch=len(idx), dtype=torch.float64, device=0)
        opt.set_pipeline(pipeline)
        u = opt.step(torch.tensor(x0, dtype=torch.float64, device="cuda:0"), c0["dyn"], 0.0).u.cpu().numpy()
        u_orc, _, _, _, _ = orc.step_batch_cold(orc.default_opt_params(**over), c0["dyn"], 0.0, x0)
        for j, i in enumerate(idx):
            errs[i] = float(np.abs(u[:, j] - np.array(cases[i]["u_star"])).max())
            orc_hit[i] = float(np.abs(u_orc[:, j] - np.array(cases[i]["u_star"])).max()) <= 1e-5
    e = np.array(errs)
    up = np.array([c["kind"] == "near-upright" for c in cases])
    hit = e <= 1e-5
    orc_hit = np.array(orc_hit)
    print("%s: GPU fixed point vs u*: near-upright median %.1e worst %.1e; within 1e-5: %d of %d (oracle: %d); within 1e-8: %d"
          % (pipeline, np.median(e[up]), e[up].max(), hit.sum(), e.size, orc_hit.sum(), (e <= 1e-8).sum()))
    # measured (MI355X, round 3): near-upright median 1.5e-12, worst 1e-10; 110 / 111 of the 123 within 1e-5 (oracle 113)
    assert e[up].max() < 1e-8, sorted(e[up])[-5:]
    assert hit.sum() >= 0.8 * e.size
    assert (e[hit] <= 1e-8).mean() >= 0.95
    # where the oracle's SQP reaches u*, the GPU's does too, but for a few swing-up problems whose long iteration the two
    # implementations finish in different basins (they part at a line-search decision made at rounding level)
    lost = [(c["x0"], x) for c, x, oh in zip(cases, errs, orc_hit) if oh and x > 1e-5]
    print("   reached by the oracle's iteration but not by the GPU's:", lost)
    assert len(lost) <= 0.05 * e.size
