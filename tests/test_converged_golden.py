"""Converged control sequences pinned by an INDEPENDENT statement of the NLP and an INDEPENDENT solver
(tests/golden/gen_converged_golden.py: SymPy-derived dynamics, numpy shooting constraints, scipy SLSQP + Newton-KKT
polish; no code shared with oracle/ or the kernels).  mini_opt, the reference's solver, is absent, so iterate-level
parity is unreachable -- but a converged solution does not depend on the solver: any correct method that
lands in the same basin reaches the same KKT point of optimization.cc:194-301's problem.

  * the golden points themselves are KKT points of the ORACLE's statement of the problem (orc_problem_eval): the two
    statements of the NLP agree;
  * the oracle's SQP, run from the reference's initial guess until it stops moving, ends within
    1e-5 of u* on every near-upright case and on the swing-up cases that land in the same basin;
  * the GPU (fp64, both pipelines) stops a little further out (near-upright: median 1e-6, worst 6e-5 .. 9e-5): the
    bounds in the GPU test are the measured ones, with the reason."""
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


def _split(cases, errs, objs):
    """(cases that reached u*, cases in another basin or not converged) and the list of near-upright misses."""
    hit = [e <= 1e-5 for e in errs]
    miss_upright = [(c["x0"], e) for c, e, h in zip(cases, errs, hit) if not h and c["kind"] == "near-upright"]
    return hit, miss_upright


def test_oracle_fixed_point_is_the_independent_optimum(orc, cases):
    errs, objs = [], []
    for c in cases:
        p = orc.default_opt_params(**dict(c["params"], **FIXED_POINT))
        out = orc.Optimization(p).step(np.array(c["x0"]), c["dyn"], c["set_point"])
        errs.append(float(np.abs(out.u - np.array(c["u_star"])).max()))
        _, _, f = _kkt_through_oracle(orc, c, out.z)
        objs.append(f)
    hit, miss_upright = _split(cases, errs, objs)
    swing = [h for c, h in zip(cases, hit) if c["kind"] == "swing-up"]
    print("oracle fixed point = u* (<= 1e-5): %d of %d cases (swing-up %d of %d); median |du| on those %.1e" % (
        sum(hit), len(cases), sum(swing), len(swing), np.median([e for e, h in zip(errs, hit) if h])))
    assert not miss_upright, miss_upright            # every near-upright problem converges to the independent optimum
    assert sum(swing) >= 0.75 * len(swing)
    # A case that misses u* is in another basin or never settled (its objective differs by 20 % and more) -- or it is
    # a NEAR miss: the objective equal to ~1e-15 relative, u a few 1e-5 .. 1e-4 away.  Those optima are flat: an
    # objective of ~1.3e3 is resolved to ~3e-13 in double, and along their flattest feasible direction that hides
    # control changes of up to 1e-4, so a merit-function line search (any, not only this one) stops accepting steps
    # there, while the independent Newton iteration on the KKT *equations* does not have that limit.  Counted, bounded,
    # reported; they must stay rare and close.
    near = []
    for c, e, h, f in zip(cases, errs, hit, objs):
        if not h and abs(f - c["objective"]) <= 1e-8 * (1.0 + c["objective"]):
            near.append((c["x0"], e))
    print("same basin, stalled near u* at the merit function's resolution:", near)
    assert all(e <= 3e-4 for _, e in near), near
    assert len(near) <= 0.05 * len(cases)


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
    hit, miss_upright = _split(cases, errs, None)
    up = np.array([c["kind"] == "near-upright" for c in cases])
    e = np.array(errs)
    print("%s: GPU fixed point = u* (<= 1e-5) on %d of %d cases (oracle: %d); near-upright: median %.1e, worst %.1e, "
          "%d of %d within 1e-5" % (pipeline, sum(hit), len(cases), sum(orc_hit), np.median(e[up]), e[up].max(),
                                    (e[up] <= 1e-5).sum(), up.sum()))
    # Measured (MI355X, round 3): 98-101 of 123 within 1e-5 (oracle 115), near-upright median 1e-6, worst 6e-5 (fused) /
    # 9e-5 (split) against the oracle's 2.5e-7 / 7.6e-6.  Both SQPs stop at exact (bitwise) fixed points a little short
    # of the optimum: with the l1 penalty grown to ~1e4 a full step raises mu |c|_1 by O(|dz|^2) more than it lowers the
    # objective (the Maratos effect), the line search cuts the step until nothing moves, and how short of u* that happens
    # differs between two implementations at the 1e-5 level.  The bound below is what the specification delivers at
    # its fixed point; the 1e-5 bar of north_star is a bar on the two implementations after the SAME number of
    # iterations (test_gpu_parity.py), not on the distance of either from the exact optimum.
    assert e[up].max() < 2e-4 and np.median(e[up]) < 3e-6, sorted(e[up])[-5:]
    assert (e[up] <= 1e-5).mean() >= 0.6
    assert sum(hit) >= 0.75 * len(cases)
    # where the oracle's SQP reaches u* within 1e-5, the GPU's is never far
    far = [(c["x0"], x) for c, x, oh in zip(cases, errs, orc_hit) if oh and x > 3e-4]
    assert not far, far
