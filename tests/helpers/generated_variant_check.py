"""Runs in a child process of tests/test_gpu_generated.py with CPMPC_LIB pointing at the library built with
-DCPMPC_GENERATED_SINGLE=1: the kernels on the generated single-pendulum dynamics, against the golden vectors and the
oracle, plus the benchmark workload's rate for DESIGN.md.  Prints one JSON line."""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

pkg = importlib.import_module("cart-pole-mpc_amd")
from oracle import oracle as orc  # noqa: E402

assert "lib_generated" in pkg.capi.LIB_PATH, pkg.capi.LIB_PATH
DYN_UI = [1.0, 0.1, 0.25, 9.81, 0.05, 0.1, 0.02, 0.8, 100.0]
T = lambda a, dt=torch.float64: torch.tensor(np.ascontiguousarray(a), dtype=dt, device="cuda:0")  # noqa: E731
res = {}
worst = 0.0
for c in json.load(open(os.path.join(ROOT, "tests", "golden", "dynamics_golden.json")))["cases"]:
    f, Jx, Ju = pkg.dynamics_batch(c["params"], T(np.array(c["x"]).reshape(4, 1)), T([c["u"]]), fext=c["f_base"] + c["f_mass"])
    for got, want in ((f[:, 0], c["f"]), (Jx[:, :, 0], c["J_x"]), (Ju[:, 0], c["J_u"])):
        want = np.asarray(want)
        worst = max(worst, float(np.abs(got.cpu().numpy() - want).max() / max(1.0, np.abs(want).max())))
res["golden_worst_rel"] = worst
rng = np.random.default_rng(0)
B = 512
x0 = np.stack([rng.uniform(-0.6, 0.6, B), rng.uniform(-np.pi, np.pi, B), rng.uniform(-1, 1, B), rng.uniform(-3, 3, B)])
over = dict(max_iterations=5, relative_exit_tol=0.0, absolute_first_derivative_tol=0.0)
for pipe in ("fused", "split"):
    opt = pkg.BatchOptimization(pkg.default_params(**over), max_batch=B, dtype=torch.float64, device=0)
    opt.set_pipeline(pipe)
    out = opt.step(T(x0), DYN_UI, 0.0)
    u_cpu, _, st_cpu, it_cpu, _ = orc.step_batch_cold(orc.default_opt_params(**over), DYN_UI, 0.0, x0)
    res["step_%s_max_abs_du" % pipe] = float(np.abs(out.u.cpu().numpy() - u_cpu).max())
    res["step_%s_status_agree" % pipe] = bool((out.status.cpu().numpy() == st_cpu).all())
for name, dt in (("f32", torch.float32), ("f64", torch.float64)):
    Bb = 262144
    xb = np.stack([rng.uniform(-0.6, 0.6, Bb), rng.uniform(-np.pi, np.pi, Bb), rng.uniform(-1, 1, Bb), rng.uniform(-3, 3, Bb)])
    opt = pkg.BatchOptimization(pkg.default_params(**over), max_batch=Bb, dtype=dt, device=0)
    xt = T(xb, dt)
    o = pkg.BatchOutputs()
    for _ in range(3):
        opt.reset()
        opt.step(xt, DYN_UI, 0.0, out=o)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        opt.reset()
        opt.step(xt, DYN_UI, 0.0, out=o)
    torch.cuda.synchronize()
    res["replans_per_s_" + name] = Bb * 10 / (time.perf_counter() - t0)
print(json.dumps(res))
