"""The host-pointer step (cpmpc_step_batch_host_in: what pendulum::Optimization::StepBatchInto and pypendulum.step_batch
deliver) at B = 262 144 and 65 536, fp64 and fp32: unsplit (round 3's form: one copy in, the kernels, one copy out, one
CPU scatter) against the chunk pipeline of round 4, with and without predicted states, into pageable and into pinned
(cpmpc_host_register) caller arrays.  Beside it the kernels alone (device-resident data) and the PCIe floor of the copy
back at the measured device-to-host rate.  Run on the GPU box from the repo root:
    python tools/host_path.py [out.json]        (default profiles/r04_host_path.json)"""
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
out_path = sys.argv[1] if len(sys.argv) > 1 else "profiles/r04_host_path.json"
lib = pkg.capi.load()
DP, IP = C.POINTER(C.c_double), C.POINTER(C.c_int32)
N = 40


def d2h_rate():
    """GB/s of a large pinned device-to-host copy on this box (the floor's denominator)."""
    n = 256 << 20
    d = torch.empty(n, dtype=torch.uint8, device="cuda")
    h = torch.empty(n, dtype=torch.uint8, pin_memory=True)
    for _ in range(2):
        h.copy_(d, non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        h.copy_(d, non_blocking=True)
    torch.cuda.synchronize()
    return 5 * n / (time.perf_counter() - t0) / 1e9


rows = []
rate = d2h_rate()
print("pinned D2H %.1f GB/s" % rate, flush=True)
for name, dt, cdt in (("f64", torch.float64, pkg.capi.F64), ("f32", torch.float32, pkg.capi.F32)):
    for B in (262144, 65536):
        rng = np.random.default_rng(1000)
        x0 = np.stack([rng.uniform(-0.6, 0.6, B), rng.uniform(-np.pi, np.pi, B), rng.uniform(-1, 1, B), rng.uniform(-3, 3, B)])
        # the kernels alone
        opt = pkg.BatchOptimization(pkg.default_params(**OVER), max_batch=B, dtype=dt, device=0)
        x0t = torch.tensor(x0, dtype=dt, device="cuda")
        out = pkg.BatchOutputs()
        for _ in range(3):
            opt.reset()
            opt.step(x0t, DYN_UI, 0.0, out=out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            opt.reset()
            opt.step(x0t, DYN_UI, 0.0, out=out)
        torch.cuda.synchronize()
        dev_ms = (time.perf_counter() - t0) / 10 * 1e3
        h = opt._h
        u = np.zeros((N, B))
        pred = np.zeros((N, 4, B))
        st = np.zeros(B, dtype=np.int32)
        dyn = (C.c_double * 9)(*DYN_UI)
        i = pkg.capi.StepHostInputs(x0=x0.ctypes.data_as(DP), dyn_shared=dyn, dyn=None, set_point_shared=0.0, set_point=None,
                                    terminal_weights=None)
        for pinned in (False, True):
            if pinned:
                pkg.capi.check(lib.cpmpc_host_register(u.ctypes.data, u.nbytes))
                pkg.capi.check(lib.cpmpc_host_register(pred.ctypes.data, pred.nbytes))
            for want_pred in (False, True):
                o = pkg.capi.StepHostOutputs(u=u.ctypes.data_as(DP), predicted=pred.ctypes.data_as(DP) if want_pred else None,
                                             status=st.ctypes.data_as(IP), iterations=None, final_cost=None, final_eq_l1=None,
                                             solution=None)
                for chunk in (0, -1, 32768, 16384, 8192):
                    pkg.capi.check(lib.cpmpc_set_host_chunk(h, chunk))

                    def step():
                        lib.cpmpc_reset(h)
                        pkg.capi.check(lib.cpmpc_step_batch_host_in(h, B, C.byref(i), C.byref(o)))
                    for _ in range(6 if pinned else 3):   # freshly registered pages are mapped for DMA on first touch
                        step()
                    reps = 8
                    t0 = time.perf_counter()
                    for _ in range(reps):
                        step()
                    ms = (time.perf_counter() - t0) / reps * 1e3
                    out_bytes = (N + (4 * N if want_pred else 0)) * B * (8 if name == "f64" else 4) + 4 * B
                    row = {"dtype": name, "batch": B, "predicted": want_pred, "caller_arrays": "pinned" if pinned else "pageable",
                           "chunk": chunk, "ms_host_to_host": ms, "replans_per_s": B / ms * 1e3, "ms_kernels_device_resident": dev_ms,
                           "ms_pcie_floor_of_the_copy_back": out_bytes / rate / 1e6,
                           "direct_dma": bool(pinned and name == "f64" and want_pred)}
                    rows.append(row)
                    print(json.dumps(row), flush=True)
            if pinned:
                pkg.capi.check(lib.cpmpc_host_unregister(u.ctypes.data))
                pkg.capi.check(lib.cpmpc_host_unregister(pred.ctypes.data))
        del opt, x0t
with open(out_path, "w") as fh:
    json.dump({"workload": "cold start, N=40, state_spacing=10, 5 SQP iterations, exits disabled; host double arrays in "
                           "(x0 [4][B]) and out (u [N][B], status [B], optionally predicted [N][4][B]); chunk 0 = unsplit "
                           "(round 3's path); CPMPC_HOST_THREADS=%s" % os.environ.get("CPMPC_HOST_THREADS", "default (8)"),
               "pinned_d2h_GBps": rate, "rows": rows}, fh, indent=1)
