#!/usr/bin/env python3
"""Secondary workloads for profiling (rocprofv3 wraps this; the headline workload is bench.py itself):
  double       BASELINE configs[4]: cart + double pendulum, B = 65536, N = 40, 5 iterations, cold start
  closed_loop  the warm-started closed loop of SURVEY 8(f1): re-plan (reference defaults, exits enabled, staged fused
               pipeline with compaction) + batched Simulator step, B = 262144
  per_problem  BASELINE configs[2] with per-problem model parameters (+-10 %), set-points and terminal weights:
               the SHARED = false instantiation of the fused kernel, B = 262144, 5 iterations, cold start
usage: run_workload.py <double|closed_loop|per_problem> [--dtype f32|f64] [--steps K] [--batch B]"""
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
DYN_UI = [1.0, 0.1, 0.25, 9.81, 0.05, 0.1, 0.02, 0.8, 100.0]
DYN_DOUBLE = [1.0, 0.1, 0.1, 0.25, 0.2, 9.81]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload", choices=["double", "closed_loop", "per_problem"])
    ap.add_argument("--dtype", choices=["f32", "f64"], default="f32")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--pipeline", choices=["auto", "split", "fused"], default="auto")
    ap.add_argument("--settle", type=int, default=0, help="closed_loop: untimed ticks first, from the benchmark's state "
                    "distribution, so that the timed ones are settled ticks (tools/settled_trace.py reads the trace)")
    ap.add_argument("--preheat", type=float, default=0.0, help="seconds of untimed steps first (the kernel-trace pass of "
                    "prof_workload.sh: rocprofv3's average is then that of a warm process, like bench.py's line)")
    a = ap.parse_args()
    dt = torch.float32 if a.dtype == "f32" else torch.float64
    rng = np.random.default_rng(7)
    out = pkg.BatchOutputs()
    if a.workload == "double":
        B = a.batch or 65536
        x0 = np.stack([rng.uniform(-0.3, 0.3, B), np.pi / 2 + rng.uniform(-0.15, 0.15, B), np.pi / 2 + rng.uniform(-0.15, 0.15, B),
                       rng.uniform(-0.3, 0.3, B), rng.uniform(-0.5, 0.5, B), rng.uniform(-0.5, 0.5, B)])
        p = pkg.default_params(max_iterations=5, relative_exit_tol=0.0, absolute_first_derivative_tol=0.0,
                               u_guess_sinusoid_amplitude=0.0)
        opt = pkg.BatchOptimization(p, max_batch=B, dtype=dt, device=0, model="double")
        opt.set_pipeline(a.pipeline)
        xt = torch.tensor(x0, dtype=dt, device="cuda:0")

        def step():
            opt.reset()
            opt.step(xt, DYN_DOUBLE, 0.0, out=out)
    elif a.workload == "per_problem":
        B = a.batch or 262144
        x0 = np.stack([rng.uniform(-0.6, 0.6, B), rng.uniform(-np.pi, np.pi, B), rng.uniform(-1, 1, B), rng.uniform(-3, 3, B)])
        dyn = torch.tensor(np.array(DYN_UI)[:, None] * (1.0 + 0.1 * rng.uniform(-1, 1, (9, B))), dtype=dt, device="cuda:0")
        sp = torch.tensor(rng.uniform(-0.2, 0.2, B), dtype=dt, device="cuda:0")
        tw = torch.tensor(np.stack([150.0 * (1.0 + 0.2 * rng.uniform(-1, 1, B)), np.where(np.arange(B) % 3 == 0, 40.0, -1.0),
                                    -np.ones(B), -np.ones(B)]), dtype=dt, device="cuda:0")
        p = pkg.default_params(max_iterations=5, relative_exit_tol=0.0, absolute_first_derivative_tol=0.0)
        opt = pkg.BatchOptimization(p, max_batch=B, dtype=dt, device=0)
        opt.set_pipeline(a.pipeline)
        xt = torch.tensor(x0, dtype=dt, device="cuda:0")

        def step():
            opt.reset()
            opt.step(xt, dyn, sp, out=out, terminal_weights=tw)
    else:
        B = a.batch or 262144
        xs = np.stack([rng.uniform(-0.3, 0.3, B), np.pi / 2 + rng.uniform(-0.4, 0.4, B), rng.uniform(-0.5, 0.5, B),
                       rng.uniform(-1, 1, B)])
        if a.settle:
            xs = np.stack([rng.uniform(-0.6, 0.6, B), rng.uniform(-np.pi, np.pi, B), rng.uniform(-1, 1, B), rng.uniform(-3, 3, B)])
        sim = pkg.BatchSimulator(B, dtype=dt, device=0)
        sim.set_state(torch.tensor(xs, dtype=dt, device="cuda:0"))
        opt = pkg.BatchOptimization(pkg.default_params(), max_batch=B, dtype=dt, device=0)
        opt.set_pipeline(a.pipeline)

        def step():
            o = opt.step(sim.get_state(), DYN_UI, 0.0, want_predicted=False, want_stats=True, out=out)
            sim.step(DYN_UI, 0.01, o.u[0].contiguous())
    for _ in range(3 + (a.settle if a.workload == "closed_loop" else 0)):
        step()
        if a.settle:
            torch.cuda.synchronize()   # a controller acts on u every tick: the stage plan then follows the tick before
    torch.cuda.synchronize()
    t_pre = time.perf_counter()
    while a.preheat > 0.0 and time.perf_counter() - t_pre < a.preheat:
        for _ in range(16):
            step()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    dtm = (time.perf_counter() - t0) / a.steps
    print(json.dumps({"workload": a.workload, "dtype": a.dtype, "batch": B, "ms_per_step": dtm * 1e3,
                      "units_per_s": B / dtm, "pipeline": opt.pipeline()}))


if __name__ == "__main__":
    main()
