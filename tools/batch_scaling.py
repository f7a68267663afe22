"""Re-plans/s against batch size, fp32 and fp64, device-resident data (cpmpc_step_batch) and host buffers
(cpmpc_step_batch_host through the C-ABI: one copy in, the kernels, one copy out): from which batch on the GPU path beats
the host's cores.  Run on the GPU box from the repo root:
    python tools/batch_scaling.py [out.json]        (default profiles/r04_batch_scaling.json)"""
import ctypes as C
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
OVER = dict(max_iterations=5, relative_exit_tol=0.0, absolute_first_derivative_tol=0.0)
out_path = sys.argv[1] if len(sys.argv) > 1 else "profiles/r04_batch_scaling.json"
lib = pkg.capi.load()
rows = []
for name, dt in (("f32", torch.float32), ("f64", torch.float64)):
    for B in (1, 4, 16, 64, 256, 1024, 4096, 16384, 32768, 65536, 131072, 262144, 524288, 1048576):
        rng = np.random.default_rng(1000)
        x0 = np.stack([rng.uniform(-0.6, 0.6, B), rng.uniform(-np.pi, np.pi, B), rng.uniform(-1, 1, B), rng.uniform(-3, 3, B)])
        x0t = torch.tensor(x0, dtype=dt, device="cuda")
        opt = pkg.BatchOptimization(pkg.default_params(**OVER), max_batch=B, dtype=dt, device=0)
        out = pkg.BatchOutputs()
        reps = 200 if B <= 4096 else (40 if B <= 131072 else 10)
        for _ in range(3):
            opt.reset()
            opt.step(x0t, DYN_UI, 0.0, out=out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            opt.reset()
            opt.step(x0t, DYN_UI, 0.0, out=out)
        torch.cuda.synchronize()
        dev_s = (time.perf_counter() - t0) / reps
        # host buffers in and out (what pendulum::Optimization::StepBatch and pypendulum.step_batch do)
        host_s = None
        if B <= 262144:
            u = np.zeros((40, B))
            st = np.zeros(B, dtype=np.int32)
            dyn = (C.c_double * 9)(*DYN_UI)
            dp = C.POINTER(C.c_double)
            def host_step():
                lib.cpmpc_reset(opt._h)
                pkg.capi.check(lib.cpmpc_step_batch_host(opt._h, B, x0.ctypes.data_as(dp), dyn, 0.0, u.ctypes.data_as(dp), None,
                                                         st.ctypes.data_as(C.POINTER(C.c_int32)), None, None, None))
            for _ in range(2):
                host_step()
            hreps = max(3, reps // 4)
            t0 = time.perf_counter()
            for _ in range(hreps):
                host_step()
            host_s = (time.perf_counter() - t0) / hreps
        row = {"dtype": name, "batch": B, "ms_per_step_device_resident": dev_s * 1e3, "replans_per_s_device_resident": B / dev_s,
               "ms_per_step_host_buffers": host_s * 1e3 if host_s else None, "replans_per_s_host_buffers": B / host_s if host_s else None}
        rows.append(row)
        print(json.dumps(row), flush=True)
        del opt, x0t
with open(out_path, "w") as fh:
    json.dump({"workload": "cold start, N=40, state_spacing=10, 5 SQP iterations, exits disabled; u + predicted + status written "
                           "(device-resident) / u + status returned (host buffers)", "rows": rows}, fh, indent=1)
