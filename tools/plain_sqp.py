#!/usr/bin/env python3
"""What the repo's additions to the textbook SQP buy: the closed loop of 262 144 controllers -- a
1 000-tick swing-up soak from arbitrary pole angles, then 50 settled ticks -- at the reference's tolerances
(optimization.hpp:30-34) under the solver options given on the command line, both dtypes, on the library CPMPC_LIB names
(default: the product).  bench.py's `variants.plain_sqp` runs it twice in child processes:
    defaults                                             the specification of DESIGN.md section 4
    --full-step-below 0 --exit-defect-floor 0            on the -DCPMPC_SKIP_MERIT=0 build (tools/_build/lib_noskip): the
                                                         iteration without the full-step rule (round 3), the exit floor
                                                         (round 4) and the merit-free converged step (round 4)
and prints ms/tick, iterations/tick, the final pole error and the solver failures of each, the reference's closed-loop
criterion (optimization_test.cc:44-66) applied to every controller.  One JSON line on stdout."""
import argparse
import gc
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
pkg = importlib.import_module("cart-pole-mpc_amd")
DYN_UI = [1.0, 0.1, 0.25, 9.81, 0.05, 0.1, 0.02, 0.8, 100.0]
FAIL = ("QP_INDEFINITE", "MAX_LAMBDA", "NON_FINITE")


def run(dt, B, ticks, settled, opts):
    rng = np.random.default_rng(7)
    x0 = np.stack([rng.uniform(-0.6, 0.6, B), rng.uniform(-np.pi, np.pi, B), rng.uniform(-1, 1, B), rng.uniform(-3, 3, B)])
    sim = pkg.BatchSimulator(B, dtype=dt, device=0)
    sim.set_state(torch.tensor(x0, dtype=dt, device="cuda:0"))
    opt = pkg.BatchOptimization(pkg.default_params(), max_batch=B, dtype=dt, device=0, opts=opts)
    out = pkg.BatchOutputs()
    hist = torch.zeros(9, dtype=torch.int64, device="cuda:0")
    its = torch.zeros((), dtype=torch.float64, device="cuda:0")
    codes = torch.arange(9, dtype=torch.int32, device="cuda:0").unsqueeze(1)

    def tick():
        o = opt.step(sim.get_state(), DYN_UI, 0.0, want_predicted=False, want_stats=True, out=out)
        sim.step(DYN_UI, 0.01, o.u[0].contiguous())
        return o

    def count(o):   # per-tick statistics that stay on the device (torch.bincount would synchronise every tick)
        hist.add_((o.status.unsqueeze(0) == codes).sum(dim=1))
        its.add_(o.iterations.double().mean())

    for _ in range(2):   # kernels and torch ops loaded before anything is timed (two ticks of the soak, untimed)
        count(tick())
    gc.collect()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(ticks - 2):
        count(tick())
    torch.cuda.synchronize()
    soak_s = time.perf_counter() - t0
    s = sim.get_state().double()
    err = (s[1] - np.pi / 2).abs()
    h = hist.cpu().numpy()
    names = pkg.capi.TERM_NAMES
    rec = {"soak": {"ticks": ticks, "ms_per_tick": soak_s / (ticks - 2) * 1e3, "iterations_per_tick": float(its.item() / ticks),
                    "status_histogram": {names[i]: int(h[i]) for i in range(9) if h[i]},
                    "solver_failures": int(sum(h[i] for i in range(9) if names[i] in FAIL)),
                    "final_pole_error_max": float(err.max().item()), "final_pole_error_median": float(err.median().item()),
                    "upright_within_1e-3": float((err < 1e-3).double().mean().item())}}
    hist.zero_()
    its.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(settled):
        count(tick())
    torch.cuda.synchronize()
    st_s = time.perf_counter() - t0
    err = (sim.get_state().double()[1] - np.pi / 2).abs()
    h = hist.cpu().numpy()
    rec["settled"] = {"ticks": settled, "ms_per_tick": st_s / settled * 1e3, "iterations_per_tick": float(its.item() / settled),
                      "status_histogram": {names[i]: int(h[i]) for i in range(9) if h[i]},
                      "solver_failures": int(sum(h[i] for i in range(9) if names[i] in FAIL)),
                      "pole_error_max": float(err.max().item()), "pole_error_median": float(err.median().item()),
                      "stage_plan_last_tick": opt.stage_plan()}
    opt.close()
    return rec


def main():
    gc.disable()   # no collector pause inside a timed loop (bench.py: run_rank)
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=262144)
    ap.add_argument("--ticks", type=int, default=1000)
    ap.add_argument("--settled", type=int, default=50)
    ap.add_argument("--full-step-below", type=float, default=None)
    ap.add_argument("--exit-defect-floor", type=float, default=None)
    ap.add_argument("--dtypes", default="f32,f64")
    a = ap.parse_args()
    over = {}
    if a.full_step_below is not None:
        over["full_step_below"] = a.full_step_below
    if a.exit_defect_floor is not None:
        over["exit_defect_floor"] = a.exit_defect_floor
    opts = pkg.capi.default_solver_opts(**over)
    res = {"library": os.environ.get("CPMPC_LIB", "product"), "batch": a.batch,
           "full_step_below": opts.full_step_below, "exit_defect_floor": opts.exit_defect_floor}
    for name in a.dtypes.split(","):
        res[name] = run(torch.float32 if name == "f32" else torch.float64, a.batch, a.ticks, a.settled, opts)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
