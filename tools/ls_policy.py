"""Step-length-memory policies of the line search compared on the oracle (cost in merit evaluations vs quality).
Run from the repo root:  python tools/ls_policy.py"""
import sys, time, numpy as np
sys.path.insert(0, '.')
from oracle import oracle as orc
from concurrent.futures import ProcessPoolExecutor
DYN_UI = [1.0, 0.1, 0.25, 9.81, 0.05, 0.1, 0.02, 0.8, 100.0]
rng = np.random.default_rng(1000)
B = 512
x0 = np.stack([rng.uniform(-0.6, 0.6, B), rng.uniform(-np.pi, np.pi, B), rng.uniform(-1, 1, B), rng.uniform(-3, 3, B)])

def run(args):
    g, gbt, iters, lo, hi = args
    p = orc.default_opt_params(max_iterations=iters, relative_exit_tol=0.0, absolute_first_derivative_tol=0.0)
    o = orc.default_solver_opts(ls_alpha_growth=g, ls_alpha_growth_backtracked=gbt)
    res = []
    for i in range(lo, hi):
        opt = orc.Optimization(p, o)
        out = opt.step(x0[:, i], DYN_UI, 0.0)
        s = out.solver_outputs
        res.append((s.line_search_evals, s.iterations, s.final_cost, s.final_eq_l1, s.failed_steps, s.termination_state))
    return res

if __name__ == "__main__":
    pols = [(2.0, 2.0), (2.0, 1.0), (2.0, 1.5), (1.5, 1.5), (1.0, 1.0), (0.0, 0.0)]
    for iters in (5, 20):
        for g, gbt in pols:
            with ProcessPoolExecutor(8) as ex:
                parts = list(ex.map(run, [(g, gbt, iters, lo, lo + 64) for lo in range(0, B, 64)]))
            r = np.array([x for p in parts for x in p], dtype=float)
            ev, it, fc, cn, fl = r[:, 0], r[:, 1], r[:, 2], r[:, 3], r[:, 4]
            w16 = (ev.reshape(-1, 16)).max(axis=1).mean() / iters
            print("iters %2d growth %.1f/%.1f: evals/iter %.2f (wave16 %.2f)  failed/iter %.3f  final cost median %.4g mean %.4g  |c|1 median %.3g mean %.3g p90 %.3g  f+10|c| mean %.4g"
                  % (iters, g, gbt, (ev / it).mean(), w16, (fl / it).mean(), np.median(fc), fc.mean(), np.median(cn), cn.mean(), np.quantile(cn, 0.9), (fc + 10 * cn).mean()))
