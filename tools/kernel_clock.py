"""The shader clock the part holds while fused_sqp_kernel runs the benchmark workload.

Builds the library with -DCPMPC_FUSED_CLOCK (two stamps per wave: s_memtime and the constant 100 MHz s_memrealtime on
entry and exit, csrc/mpc_fused.hpp) into tools/_build/lib_clock/, runs the bench workload back to back for `--seconds`
(so the power management has settled), then reads the sums of one more batch of launches:
    clock = sum(shader cycles) / sum(100 MHz ticks) x 100 MHz
and, from the same stamps, the wave's residence time in cycles.  Together with the static instruction count of the
kernel this gives cycles per vector instruction per SIMD, i.e. the issue fraction.
Run from the repo root on a GPU box:  python tools/kernel_clock.py [--seconds 2] [--out profiles/r03_clock.json]
(`--build-only` compiles here, where there is no GPU, so that the library travels to the box)."""
import argparse
import ctypes
import importlib
import json
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
# the variant must be selected before the package (whose capi module reads CPMPC_LIB at import) is imported
os.environ["CPMPC_LIB"] = os.path.join(ROOT, "tools", "_build", "lib_clock", "libcpmpc.so")
build = importlib.import_module("cart-pole-mpc_amd.build")

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=2.0)
ap.add_argument("--batch", type=int, default=262144)
ap.add_argument("--out", default=None)
ap.add_argument("--build-only", action="store_true")
args = ap.parse_args()

LIB = build.build_variant("clock", ["-DCPMPC_FUSED_CLOCK"])
if args.build_only:
    sys.exit(0)

import numpy as np  # noqa: E402
import torch  # noqa: E402

assert os.path.samefile(LIB, os.environ["CPMPC_LIB"])
pkg = importlib.import_module("cart-pole-mpc_amd")
lib = ctypes.CDLL(LIB)
DYN_UI = [1.0, 0.1, 0.25, 9.81, 0.05, 0.1, 0.02, 0.8, 100.0]
B = args.batch
rng = np.random.default_rng(1000)
x0 = np.stack([rng.uniform(-0.6, 0.6, B), rng.uniform(-np.pi, np.pi, B), rng.uniform(-1, 1, B), rng.uniform(-3, 3, B)])
res = {"batch": B, "settle_seconds": args.seconds, "records": []}
for name, dt in (("f32", torch.float32), ("f64", torch.float64)):
    x0t = torch.tensor(x0, dtype=dt, device="cuda")
    opt = pkg.BatchOptimization(pkg.default_params(max_iterations=5, relative_exit_tol=0.0, absolute_first_derivative_tol=0.0),
                                max_batch=B, dtype=dt, device=0)
    opt.set_pipeline("fused")
    out = pkg.BatchOutputs()
    buf = (ctypes.c_ulonglong * 4)()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < args.seconds:
        for _ in range(20):
            opt.reset()
            opt.step(x0t, DYN_UI, 0.0, want_predicted=True, out=out)
        torch.cuda.synchronize()
        n += 20
    lib.cpmpc_debug_kernel_clock(buf)   # clear what the settling phase accumulated
    reps = 50
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        opt.reset()
        opt.step(x0t, DYN_UI, 0.0, want_predicted=True, out=out)
    e1.record()
    torch.cuda.synchronize()
    lib.cpmpc_debug_kernel_clock(buf)
    cyc, real, waves, cmax = [int(v) for v in buf]
    rec = {"dtype": name, "settle_steps": n, "measured_steps": reps, "ms_per_step": e0.elapsed_time(e1) / reps,
           "waves": waves, "clock_GHz": cyc / real * 0.1, "cycles_per_wave_mean": cyc / waves, "cycles_per_wave_max": cmax,
           "us_per_wave_mean": real / waves * 0.01}
    res["records"].append(rec)
    print(json.dumps(rec), flush=True)
    del opt, x0t
if args.out:
    with open(args.out, "w") as fh:
        json.dump(res, fh, indent=1)
