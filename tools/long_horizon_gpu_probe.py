#!/usr/bin/env python3
"""Long horizons on the GPU (fp64): lanes beyond 1e-5 of the CPU check at N = 120 / 160 per pipeline and QP option
(refine_qp on / off) -- the experiment behind DESIGN.md section 8 "long horizons".  usage: long_horizon_gpu_probe.py [B] [out.json]"""
import importlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("cart-pole-mpc_amd")
from oracle import oracle as orc  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tools"))
from parity_sweep import DYN_UI, NO_TOL, THREADS, states  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
SEED_SHIFT = int(os.environ.get("PROBE_SEED", "0"))
CASES = [("N=120 sp=12, 3 its", dict(NO_TOL, window_length=120, state_spacing=12, max_iterations=3), 17),
         ("N=160 sp=10, 3 its", dict(NO_TOL, window_length=160, max_iterations=3), 18),
         ("N=160 sp=10, 5 its", dict(NO_TOL, window_length=160, max_iterations=5), 18)]
res = []
for tag, over, seed in CASES:
    x0 = states(np.random.default_rng(500 + seed + SEED_SHIFT), B)
    u_cpu, _, st_cpu, it_cpu, _ = orc.step_batch_cold(orc.default_opt_params(**over), DYN_UI, 0.0, x0, num_threads=THREADS)
    for pipe in ("auto", "fused", "split"):
        for refine in ((None,) if pipe == "auto" else (False, True)):   # auto: the handle as a caller gets it (refinement on, split)
            opt = pkg.BatchOptimization(pkg.default_params(**over), max_batch=B, dtype=torch.float64, device=0,
                                        allow_long_horizon=True, refine_qp=refine)
            opt.set_pipeline(pipe)
            out = opt.step(torch.tensor(x0, dtype=torch.float64, device="cuda:0"), DYN_UI, 0.0, want_stats=True)
            err = np.abs(out.u.cpu().numpy() - u_cpu).max(axis=0)
            u_gpu = out.u.cpu().numpy()
            rec = {"case": tag, "pipeline": opt.pipeline(), "refine_qp": refine, "lanes": B, "lanes_over_1e-5": int((err > 1e-5).sum()),
                   "max": float(err.max()), "p99": float(np.quantile(err, 0.99)), "median": float(np.median(err)),
                   "status_agree": int((out.status.cpu().numpy() == st_cpu).sum())}
            far = np.nonzero(err > 1e-5)[0][:128]
            if far.size:   # the arbiter: the same restatement in x87 extended precision says which side moved
                u_ld = orc.step_batch_cold_ld(orc.default_opt_params(**over), DYN_UI, 0.0, x0[:, far])[0]
                e_g = np.abs(u_gpu[:, far] - u_ld).max(axis=0)
                e_c = np.abs(u_cpu[:, far] - u_ld).max(axis=0)
                rec["arbiter"] = {"lanes": int(far.size), "gpu_at_fault": int(((e_g > 1e-5) & (e_g > 2 * e_c)).sum()),
                                  "oracle_at_fault": int(((e_c > 1e-5) & (e_c > 2 * e_g)).sum()),
                                  "neither_reproducible": int(((e_g > 1e-5) & (e_c > 1e-5) & (e_g <= 2 * e_c) & (e_c <= 2 * e_g)).sum()),
                                  "gpu_vs_extended_max": float(e_g.max()), "oracle_vs_extended_max": float(e_c.max())}
            print(json.dumps(rec), flush=True)
            res.append(rec)
            opt.close()
if len(sys.argv) > 2:
    json.dump(res, open(sys.argv[2], "w"), indent=1)
