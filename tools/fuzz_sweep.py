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
sys.path.insert(0, os.path.join(ROOT, "tools"))
pkg = importlib.import_module("cart-pole-mpc_amd")
from oracle import oracle as orc  # noqa: E402  (the checker)

ap = argparse.ArgumentParser()
ap.add_argument("out", nargs="?", default=os.path.join(ROOT, "profiles", "r04_fuzz_sweep.json"))
ap.add_argument("--seeds", type=int, default=200)
ap.add_argument("--lanes", type=int, default=2048)
ap.add_argument("--refine-qp", choices=["auto", "on", "off"], default="auto",
                help="CPMPC_CREATE_REFINE_QP / NO_REFINE_QP / the library's default (on when u_cost_weight < 0.05)")
ap.add_argument("--per-problem-lanes", type=int, default=256, help="lanes of each case also solved with per-problem inputs")
args = ap.parse_args()
THREADS = int(os.environ.get("CPMPC_ORACLE_THREADS", "16"))


from fuzz_sweep_case import random_case  # noqa: E402


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
    opt = pkg.BatchOptimization(pkg.default_params(**over), max_batch=B, dtype=torch.float64, device=0,
                                refine_qp={"auto": None, "on": True, "off": False}[args.refine_qp])
    out = opt.step(torch.tensor(x0, dtype=torch.float64, device="cuda:0"), dyn, sp, want_stats=True)
    u_g, st_g, it_g = out.u.cpu().numpy(), out.status.cpu().numpy(), out.iterations.cpu().numpy()
    u_c, _, st_c, it_c, _ = orc.step_batch_cold(orc.default_opt_params(**over), dyn, sp, x0, num_threads=THREADS)
    err = np.abs(u_g - u_c).max(axis=0)
    bad = int(((st_g != st_c) | (it_g != it_c)).sum())
    stiff = dyn[4] > 0.0 and dyn[5] < 1e-3   # friction slope mu (m_b + m_1) g / max(v_mu, 1e-6) that RK4 at this dt cannot follow
    rec = {"seed": seed, "over": over, "dyn": dyn, "set_point": sp, "pipeline": opt.pipeline(), "lanes": B, "stiff_friction": stiff,
           "lanes_over_1e-5": int((err > 1e-5).sum()), "max": float(err.max()), "p99": float(np.quantile(err, 0.99)),
           "median": float(np.median(err)), "status_or_iterations_differ": bad}
    idx = np.nonzero((err > 1e-5) | (st_g != st_c) | (it_g != it_c))[0]
    if idx.size:
        # the arbiter: the same restatement in x87 extended precision on the lanes that are off.  A lane on which the
        # double CPU check is itself far from its extended-precision twin is decided by rounding (an Armijo test or an
        # exit test within an ulp of its threshold), not by either implementation
        u_ld, _, st_ld, _, _ = orc.step_batch_cold_ld(orc.default_opt_params(**over), dyn, sp, x0[:, idx])
        e_g = np.abs(u_g[:, idx] - u_ld).max(axis=0)
        e_c = np.abs(u_c[:, idx] - u_ld).max(axis=0)
        gpu_at_fault = (e_g > 1e-5) & (e_g > 2.0 * e_c)
        rec["arbiter"] = {"lanes": int(idx.size), "gpu_vs_extended": [float(v) for v in e_g], "cpu_check_vs_extended": [float(v) for v in e_c],
                          "lanes_gpu_at_fault": int(gpu_at_fault.sum()), "lanes_cpu_check_moved_more": int(((e_c > 1e-5) & (e_c >= e_g)).sum())}
        tot["gpu_at_fault"] = tot.get("gpu_at_fault", 0) + int(gpu_at_fault.sum())
        key = "gpu_at_fault_stiff_friction" if stiff else "gpu_at_fault_otherwise"
        tot[key] = tot.get(key, 0) + int(gpu_at_fault.sum())
        tot["arbitrated"] = tot.get("arbitrated", 0) + int(idx.size)
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
            if e2[i] > 1e-5:   # the arbiter on this one problem (its own parameters and terminal rows)
                u_ld = orc.step_batch_cold_ld(p, dyn_pp[:, i], float(sp_pp[i]), x0[:, i:i + 1])[0][:, 0]
                eg, ec = float(np.abs(u2[:, i] - u_ld).max()), float(np.abs(so.u - u_ld).max())
                fault = eg > 1e-5 and eg > 2.0 * ec
                rec.setdefault("pp_arbiter", []).append({"lane": i, "gpu_vs_extended": eg, "cpu_check_vs_extended": ec, "gpu_at_fault": fault})
                key = "pp_gpu_at_fault_stiff_friction" if stiff else "pp_gpu_at_fault_otherwise"
                tot[key] = tot.get(key, 0) + int(fault)
        rec.update({"pp_lanes": n, "pp_lanes_over_1e-5": int((e2 > 1e-5).sum()), "pp_max": float(e2.max()), "pp_status_differ": sd})
        tot["pp_lanes"] += n
        tot["pp_over_1e-5"] += rec["pp_lanes_over_1e-5"]
        tot["pp_status_differ"] += sd
        worst = max(worst, float(e2.max()))
    cases.append(rec)
    tot["lanes"] += B
    tot["lanes_stiff_friction" if stiff else "lanes_otherwise"] = tot.get("lanes_stiff_friction" if stiff else "lanes_otherwise", 0) + B
    tot["over_1e-5"] += rec["lanes_over_1e-5"]
    tot["status_or_iterations_differ"] += bad
    worst = max(worst, rec["max"])
    if rec["lanes_over_1e-5"] or bad or rec.get("pp_lanes_over_1e-5") or rec.get("pp_status_differ") or seed % 20 == 0:
        print(json.dumps({k: rec[k] for k in rec if k not in ("dyn",)}), flush=True)
    del opt
report = {"what": "GPU fp64 vs the CPU check over random problem definitions; lanes that are off are re-solved by the extended-"
                  "precision build of the CPU check (the arbiter): the GPU is at fault where it is further than 1e-5 AND more than "
                  "twice as far from that answer as the double CPU check", "refine_qp": args.refine_qp, "seeds": args.seeds,
          "lanes_per_seed": args.lanes, "totals": tot, "worst_max_abs_du": worst, "wall_s": time.perf_counter() - t_all, "cases": cases}
with open(args.out, "w") as fh:
    json.dump(report, fh, indent=1)
print(json.dumps({"totals": tot, "worst": worst, "wall_s": report["wall_s"]}))
