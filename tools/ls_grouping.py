"""How much of the line search's lock-step cost could regrouping recover?    Per-problem trial
counts of every iteration from the oracle (4096 problems of the benchmark distribution), then the mean over waves of 16
of the per-wave maximum, for the natural order and for problems regrouped inside blocks by various keys.
Run from the repo root:  python tools/ls_grouping.py"""
import sys, numpy as np
sys.path.insert(0, '.')
from oracle import oracle as orc
from concurrent.futures import ProcessPoolExecutor
DYN_UI = [1.0, 0.1, 0.25, 9.81, 0.05, 0.1, 0.02, 0.8, 100.0]
rng = np.random.default_rng(1000)
B = 4096
x0 = np.stack([rng.uniform(-0.6, 0.6, B), rng.uniform(-np.pi, np.pi, B), rng.uniform(-1, 1, B), rng.uniform(-3, 3, B)])
IT = 5
def run(args):
    lo, hi = args
    res = np.zeros((hi - lo, IT), int)
    for i in range(lo, hi):
        prev = 0
        for k in range(1, IT + 1):
            p = orc.default_opt_params(max_iterations=k, relative_exit_tol=0.0, absolute_first_derivative_tol=0.0)
            s = orc.Optimization(p).step(x0[:, i], DYN_UI, 0.0).solver_outputs
            res[i - lo, k - 1] = s.line_search_evals - prev
            prev = s.line_search_evals
    return res
if __name__ == "__main__":
    with ProcessPoolExecutor(8) as ex:
        ev = np.concatenate(list(ex.map(run, [(lo, lo + 128) for lo in range(0, B, 128)])))
    print("mean evals/iter per problem", ev.mean(axis=0), ev.mean())
    def wave_cost(order_fn, block):
        tot = 0.0
        for k in range(IT):
            d = ev[:, k]
            if k == 0 or order_fn is None:
                perm = np.arange(B)
            else:
                key = order_fn(k)
                perm = np.concatenate([b0 + np.argsort(key[b0:b0 + block], kind='stable') for b0 in range(0, B, block)])
            tot += d[perm].reshape(-1, 16).max(axis=1).mean()
        return tot / IT
    print("natural order, wave of 16: %.3f" % wave_cost(None, 64))
    for block in (64, 256, 1024, B):
        print("block %5d sort by previous iteration's count: %.3f   by cumulative count so far: %.3f   perfect (this iteration's count): %.3f"
              % (block, wave_cost(lambda k: ev[:, k - 1], block), wave_cost(lambda k: ev[:, :k].sum(axis=1), block),
                 wave_cost(lambda k: ev[:, k], block)))
