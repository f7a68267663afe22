import importlib, sys, time, numpy as np, torch
sys.path.insert(0, '.')
pkg = importlib.import_module("cart-pole-mpc_amd")
DYN_UI = [1.0, 0.1, 0.25, 9.81, 0.05, 0.1, 0.02, 0.8, 100.0]
for B in (32768, 65536, 131072, 196608, 229376, 262144, 294912, 327680, 524288, 1048576):
    rng = np.random.default_rng(1000)
    x0 = np.stack([rng.uniform(-0.6, 0.6, B), rng.uniform(-np.pi, np.pi, B), rng.uniform(-1, 1, B), rng.uniform(-3, 3, B)])
    x0t = torch.tensor(x0, dtype=torch.float32, device='cuda')
    opt = pkg.BatchOptimization(pkg.default_params(max_iterations=5, relative_exit_tol=0.0, absolute_first_derivative_tol=0.0), max_batch=B, dtype=torch.float32, device=0)
    opt.profile_enable(True)
    for _ in range(3):
        opt.reset(); opt.step(x0t, DYN_UI, 0.0)
    torch.cuda.synchronize(); opt.profile_reset(); t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        opt.reset(); opt.step(x0t, DYN_UI, 0.0)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    pr = opt.profile_read()
    fused = pr["fused_sqp_kernel"][0] / reps
    waves = B * 4 / 64
    print("B=%8d waves=%6d (%.2f rounds of 2048): step %.3f ms  %.1f M/s   fused %.3f ms = %.1f us per round-equivalent" % (B, waves, waves / 2048, dt * 1e3, B / dt / 1e6, fused, fused * 1e3 / (waves / 2048)))
