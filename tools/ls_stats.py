import importlib, sys, numpy as np, torch
sys.path.insert(0, '.')
pkg = importlib.import_module("cart-pole-mpc_amd")
DYN_UI = [1.0, 0.1, 0.25, 9.81, 0.05, 0.1, 0.02, 0.8, 100.0]
rng = np.random.default_rng(1000)
B = 65536
x0 = np.stack([rng.uniform(-0.6, 0.6, B), rng.uniform(-np.pi, np.pi, B), rng.uniform(-1, 1, B), rng.uniform(-3, 3, B)])
x0t = torch.tensor(x0, dtype=torch.float32, device='cuda')
prev = np.zeros(B, int)
for it in range(1, 8):
    opt = pkg.BatchOptimization(pkg.default_params(max_iterations=it, relative_exit_tol=0.0, absolute_first_derivative_tol=0.0), max_batch=B, dtype=torch.float32, device=0)
    out = opt.step(x0t, DYN_UI, 0.0, want_stats=True)
    ev = out.ls_evals.cpu().numpy().astype(int)
    d = ev - prev
    prev = ev
    hist = np.bincount(d, minlength=7)[:7]
    g16 = d.reshape(-1, 16).max(axis=1)   # what a wave of 16 problems pays
    print("iter %d: mean %.2f  hist(0..6) %s  wave-of-16 mean max %.2f  failed-steps mean %.3f" % (it, d.mean(), hist.tolist(), g16.mean(), out.failed_steps.float().mean().item() if hasattr(out,'failed_steps') else -1))
