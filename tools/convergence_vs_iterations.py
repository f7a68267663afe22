"""Distance of the GPU's (both pipelines) and the oracle's iterate from the independent optimum of
tests/golden/converged_golden.json after 20 ... 1000 iterations (near-upright cases at reference defaults): the
measurement behind the full-step rule of DESIGN.md section 4.  Run on the GPU box from the repo root."""
import importlib, json, os, sys
import numpy as np, torch
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
pkg = importlib.import_module("cart-pole-mpc_amd")
from oracle import oracle as orc
d = json.load(open(os.path.join(R, "tests/golden/converged_golden.json")))
cases = [c for c in d["cases"] if c["kkt_residual"] < 1e-9 and c["eq_l1"] < 1e-10 and not c["clamp_active"] and c["config"] == "reference defaults" and c["kind"] == "near-upright"]
x0 = np.array([c["x0"] for c in cases]).T.copy(); ustar = np.array([c["u_star"] for c in cases]).T
B = x0.shape[1]
for its in (20, 40, 80, 160, 320, 1000):
    over = dict(max_iterations=its, relative_exit_tol=0.0, absolute_first_derivative_tol=0.0)
    row = []
    for pipe in ("fused", "split"):
        opt = pkg.BatchOptimization(pkg.default_params(**over), max_batch=B, dtype=torch.float64, device=0)
        opt.set_pipeline(pipe)
        out = opt.step(torch.tensor(x0, dtype=torch.float64, device="cuda:0"), cases[0]["dyn"], 0.0, want_stats=True)
        e = np.abs(out.u.cpu().numpy() - ustar).max(axis=0)
        row.append((pipe, np.median(e), e.max(), (e <= 1e-5).sum(), float(out.ls_evals.float().mean())))
    u_o, _, st, it, _ = orc.step_batch_cold(orc.default_opt_params(**over), cases[0]["dyn"], 0.0, x0)
    eo = np.abs(u_o - ustar).max(axis=0)
    print(its, " ".join("%s med %.1e max %.1e hit %d evals %.0f |" % r for r in row), "oracle med %.1e max %.1e hit %d" % (np.median(eo), eo.max(), (eo <= 1e-5).sum()))
over = dict(max_iterations=1000, relative_exit_tol=0.0, absolute_first_derivative_tol=0.0)
opt = pkg.BatchOptimization(pkg.default_params(**over), max_batch=B, dtype=torch.float64, device=0)
out = opt.step(torch.tensor(x0, dtype=torch.float64, device="cuda:0"), cases[0]["dyn"], 0.0, want_stats=True)
print("gpu status", np.unique(out.status.cpu().numpy(), return_counts=True), "iters", out.iterations.cpu().numpy()[:16], "eq", out.final_eq_l1.cpu().numpy()[:8])
p = orc.default_opt_params(**over)
for b in range(8):
    o = orc.Optimization(p).step(x0[:, b], cases[0]["dyn"], 0.0).solver_outputs
    print("oracle lane", b, "status", o.termination_state, "iters", o.iterations, "failed", o.failed_steps, "pen %.2e eq %.2e evals %d" % (o.final_penalty, o.final_eq_l1, o.line_search_evals))
