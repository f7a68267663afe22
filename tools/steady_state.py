"""The closed loop in steady state: B controllers are brought to their set-point (SETTLE ticks from the
benchmark's state distribution), then TICKS ticks are timed with no host synchronisation inside the loop, once per
staging of the fused pipeline (auto = the library's default, planned per step from the iteration histogram of an earlier
step; first:next = cpmpc_set_compaction; 0:0 = one launch): wall clock per tick, HIP-event
time per kernel, and the histogram of iterations per problem in the last tick.
    python tools/steady_state.py [--dtype f64] [--batch 262144] [--settle 300] [--ticks 200] [--fo-tol 1e-6] [--out x.json]"""
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

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", choices=["f32", "f64"], default="f64")
ap.add_argument("--batch", type=int, default=262144)
ap.add_argument("--settle", type=int, default=300)
ap.add_argument("--ticks", type=int, default=200)
ap.add_argument("--fo-tol", type=float, default=None)
ap.add_argument("--stagings", default="auto,3:1,1:1,2:1,1:2,0:0")
ap.add_argument("--cold", action="store_true", help="instead of the settled closed loop: cold-start re-plans of the "
                "benchmark's states with the exits enabled (bench.py's `exits_enabled` variant), per staging")
ap.add_argument("--iters", type=int, default=8, help="max_iterations")
ap.add_argument("--sync", action="store_true", help="synchronise with the device after every tick, as a controller that "
                "acts on u must (the default loop queues ticks ahead; the default staging then plans from an older tick)")
ap.add_argument("--out", default=None)
args = ap.parse_args()
dt = torch.float32 if args.dtype == "f32" else torch.float64
B = args.batch
rng = np.random.default_rng(7)
x0 = np.stack([rng.uniform(-0.6, 0.6, B), rng.uniform(-np.pi, np.pi, B), rng.uniform(-1, 1, B), rng.uniform(-3, 3, B)])
over = {} if args.fo_tol is None else {"absolute_first_derivative_tol": args.fo_tol}
over["max_iterations"] = args.iters

sim = pkg.BatchSimulator(B, dtype=dt, device=0)
sim.set_state(torch.tensor(x0, dtype=dt, device="cuda:0"))
opt = pkg.BatchOptimization(pkg.default_params(**over), max_batch=B, dtype=dt, device=0)
out = pkg.BatchOutputs()


x0_dev = torch.tensor(x0, dtype=dt, device="cuda:0")


def tick():
    if args.cold:
        opt.reset()
        return opt.step(x0_dev, DYN_UI, 0.0, want_predicted=True, want_stats=True, out=out)
    o = opt.step(sim.get_state(), DYN_UI, 0.0, want_predicted=False, want_stats=True, out=out)
    sim.step(DYN_UI, 0.01, o.u[0])
    if args.sync:
        torch.cuda.synchronize()
    return o


for _ in range(0 if args.cold else args.settle):
    tick()
torch.cuda.synchronize()
settled_state = sim.get_state().clone()
settled_z = None if args.cold else opt.get_solution(B).clone()
res = {"workload": "cold start, exits enabled" if args.cold else "settled closed loop", "max_iterations": args.iters, "dtype": args.dtype, "batch": B, "settle_ticks": args.settle, "ticks": args.ticks,
       "absolute_first_derivative_tol": args.fo_tol if args.fo_tol is not None else 1e-6, "stagings": {}}
for name in args.stagings.split(","):
    # every staging starts from the same settled state and warm start
    if not args.cold:
        sim.set_state(settled_state.clone())
        opt.set_previous_solution(settled_z)
    if name == "auto":
        pass   # the library's default: planned per step from the iteration histogram of an earlier step (cpmpc.h)
    else:
        first, nxt = (int(v) for v in name.split(":"))
        opt.set_compaction(first, nxt)
    for _ in range(5):
        tick()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.ticks):
        o = tick()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / args.ticks
    opt.profile_enable(True)
    opt.profile_reset()
    n_prof = 20
    for _ in range(n_prof):
        o = tick()
    torch.cuda.synchronize()
    prof = opt.profile_read()
    opt.profile_enable(False)
    its = torch.bincount(o.iterations.long(), minlength=9).cpu().numpy()
    st = torch.bincount(o.status.long(), minlength=9).cpu().numpy()
    rec = {"stage_plan_last_tick": opt.stage_plan(), "ms_per_tick": wall * 1e3, "controller_ticks_per_s": B / wall,
           "kernels_ms_per_tick": {k: round(v[0] / n_prof, 4) for k, v in prof.items()},
           "launches_per_tick": {k: v[1] / n_prof for k, v in prof.items()},
           "iterations_histogram_last_tick": {str(i): int(c) for i, c in enumerate(its) if c},
           "status_last_tick": {pkg.capi.TERM_NAMES[i]: int(c) for i, c in enumerate(st) if c}}
    res["stagings"][name] = rec
    print(name, json.dumps(rec), flush=True)
if args.out:
    with open(args.out, "w") as fh:
        json.dump(res, fh, indent=1)
