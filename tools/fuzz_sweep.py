"""Fuzz at scale (round 4): SEEDS random problem definitions -- horizon / spacing served by the tuned, the run-time-spacing
and the split kernels, every mix of cost and equality terminal rows, zero weights, friction / drag / bumpers on and off,
exit tolerances on and off, random dynamics parameters -- each solved for LANES random states on the GPU in fp64, with
shared parameters AND with per-problem parameters / set-points / terminal rows drawn around them, against the CPU check:
termination state and iteration count on every lane, controls within 1e-5.  The unit-test fuzz
(tests/test_gpu_parity.py::test_step_parity_fuzz) is this at 32 seeds x 96 lanes.
    python tools/fuzz_sweep.py [out.json] [--seeds 200] [--lanes 2048]"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("cart-pole-mpc_amd")
from oracle import oracle as orc  # noqa: E402  (the checker)

ap = argparse.ArgumentParser()
ap.add_argument("out", nargs="?", default=os.path.join(ROOT, "profiles", "r04_fuzz_sweep.json"))
ap.add_argument("--seeds", type=int, default=200)
ap.add_argument("--lanes", type=int, default=2048)
ap.add_argument("--per-problem-lanes", type=int, default=256, help="lanes of each case also solved with per-problem inputs")
args = ap.parse_args()
THREADS = int(os.environ.get("CPMPC_ORACLE_THREADS", "16"))


def random_case(rng):
    N, sp = [(40, 10), (40, 5), (20, 10), (20, 5), (40, 20), (80, 10), (40, 8), (30, 6), (24, 3), (16, 16), (100, 10), (60, 12)][rng.integers(0, 12)]
    sign = lambda w: float(w if rng.random() < 0.5 else -1.0)     # noqa: E731  cost row or equality row
    over = dict(
        window_length=N, state_spacing=sp, max_iterations=int(rng.integers(2, 7)),
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
        over["u_cost_weight"] = 0.1
    if over["window_length"] * over["control_dt"] > 1.0:
        over["control_dt"] = 0.01
    dyn = [float(rng.uniform(0.5, 2.0)), float(rng.uniform(0.05, 0.3)), float(rng.uniform(0.15, 0.5)), 9.81,
           float(rng.choice([0.0, 0.05, 0.2])), float(rng.choice([1e-7, 0.05, 0.1])), float(rng.choice([0.0, 0.02, 0.1])),
           float(rng.uniform(0.5, 1.0)), float(rng.choice([0.0, 50.0, 100.0]))]
    return over, dyn, float(rng.uniform(-0.3, 0.3))


def states(rng, B):
    x = np.stack([rng.uniform(-0.6, 0.6, B), rng.uniform(-np.pi, np.pi, B), rng.uniform(-1, 1, B), rng.uniform(-3, 3, B)])
    x[1, ::2] = np.pi / 2 + rng.uniform(-0.4, 0.4, x[1, ::2].shape)
    return x


cases = []
tot = {"lanes": 0, "over_1e-5": 0, "status_or_iterations_differ": 0, "pp_lanes": 0, "pp_over_1e-5": 0, "pp_status_differ": 0}
worst = 0.0
t_all = time.perf_counter()
for seed in range(args.seeds):
    rng = np.random.default_rng(4000 + seed)
    over, dyn, sp = random_case(rng)
    B = args.lanes
    x0 = states(rng, B)
    opt = pkg.BatchOptimization(pkg.default_params(**over), max_batch=B, dtype=torch.float64, device=0)
    out = opt.step(torch.tensor(x0, dtype=torch.float64, device="cuda:0"), dyn, sp, want_stats=True)
    u_g, st_g, it_g = out.u.cpu().numpy(), out.status.cpu().numpy(), out.iterations.cpu().numpy()
    u_c, _, st_c, it_c, _ = orc.step_batch_cold(orc.default_opt_params(**over), dyn, sp, x0, num_threads=THREADS)
    err = np.abs(u_g - u_c).max(axis=0)
    bad = int(((st_g != st_c) | (it_g != it_c)).sum())
    rec = {"seed": seed, "over": over, "dyn": dyn, "set_point": sp, "pipeline": opt.pipeline(), "lanes": B,
           "lanes_over_1e-5": int((err > 1e-5).sum()), "max": float(err.max()), "p99": float(np.quantile(err, 0.99)),
           "median": float(np.median(err)), "status_or_iterations_differ": bad}
    # the same definition with per-problem parameters, set-points and terminal rows around it, lane by lane on the CPU
    n = min(args.per_problem_lanes, B)
    if n:
        opt.reset()
        dyn_pp = np.array(dyn)[:, None] * (1.0 + 0.08 * rng.uniform(-1, 1, (9, n)))
        dyn_pp[3] = 9.81
        dyn_pp[5] = dyn[5]
        sp_pp = sp + rng.uniform(-0.1, 0.1, n)
        base_w = np.array([over["b_x_final_cost_weight"], over["th_final_cost_weight"], over["b_x_dot_final_cost_weight"],
                           over["th_dot_final_cost_weight"]])
        tw = np.where(rng.random((4, n)) < 0.5, np.abs(base_w)[:, None] * rng.uniform(0.5, 2.0, (4, n)), -1.0)
        T = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda:0")  # noqa: E731
        o2 = opt.step(T(x0[:, :n]), T(dyn_pp), T(sp_pp), want_stats=True, terminal_weights=T(tw))
        u2, st2 = o2.u.cpu().numpy(), o2.status.cpu().numpy()
        e2 = np.zeros(n)
        sd = 0
        for i in range(n):
            p = orc.default_opt_params(**dict(over, b_x_final_cost_weight=float(tw[0, i]), th_final_cost_weight=float(tw[1, i]),
                                             b_x_dot_final_cost_weight=float(tw[2, i]), th_dot_final_cost_weight=float(tw[3, i])))
            so = orc.Optimization(p).step(x0[:, i], dyn_pp[:, i], float(sp_pp[i]))
            e2[i] = np.abs(u2[:, i] - so.u).max()
            sd += int(st2[i] != so.solver_outputs.termination_state)
        rec.update({"pp_lanes": n, "pp_lanes_over_1e-5": int((e2 > 1e-5).sum()), "pp_max": float(e2.max()), "pp_status_differ": sd})
        tot["pp_lanes"] += n
        tot["pp_over_1e-5"] += rec["pp_lanes_over_1e-5"]
        tot["pp_status_differ"] += sd
        worst = max(worst, float(e2.max()))
    cases.append(rec)
    tot["lanes"] += B
    tot["over_1e-5"] += rec["lanes_over_1e-5"]
    tot["status_or_iterations_differ"] += bad
    worst = max(worst, rec["max"])
    if rec["lanes_over_1e-5"] or bad or rec.get("pp_lanes_over_1e-5") or rec.get("pp_status_differ") or seed % 20 == 0:
        print(json.dumps({k: rec[k] for k in rec if k not in ("dyn",)}), flush=True)
    del opt
report = {"what": __doc__.split("\n\n")[0] if False else "GPU fp64 vs the CPU check over random problem definitions", "seeds": args.seeds,
          "lanes_per_seed": args.lanes, "totals": tot, "worst_max_abs_du": worst, "wall_s": time.perf_counter() - t_all, "cases": cases}
with open(args.out, "w") as fh:
    json.dump(report, fh, indent=1)
print(json.dumps({"totals": tot, "worst": worst, "wall_s": report["wall_s"]}))
