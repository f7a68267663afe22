"""random_case of tools/fuzz_sweep.py, importable on its own (used by the fuzz sweep and by lane-level debugging)."""


def random_case(rng):
    N, sp = [(40, 10), (40, 5), (20, 10), (20, 5), (40, 20), (80, 10), (40, 8), (30, 6), (24, 3), (16, 16), (100, 10), (60, 12)][rng.integers(0, 12)]
    sign = lambda w: float(w if rng.random() < 0.5 else -1.0)     # noqa: E731  cost row or equality row
    over = dict(
        window_length=N, state_spacing=sp, max_iterations=int(rng.integers(2, 7)),
        control_dt=float(rng.choice([0.005, 0.01, 0.02])),
        relative_exit_tol=float(rng.choice([0.0, 1e-5, 1e-3])),
        absolute_first_derivative_tol=float(rng.choice([0.0, 1e-6, 1e-2])),
        equality_penalty_initial=float(10.0 ** rng.uniform(-1, 2)),
        u_guess_sinusoid_amplitude=float(rng.choice([0.0, 3.0, 10.0])),
        u_cost_weight=float(rng.choice([0.0, 0.01, 0.1, 1.0])),
        u_derivative_cost_weight=float(rng.choice([0.0, 0.05, 0.1, 1.0])),
        b_x_final_cost_weight=sign(10.0 ** rng.uniform(0, 2.5)),
        th_final_cost_weight=sign(10.0 ** rng.uniform(0, 2.5)),
        b_x_dot_final_cost_weight=sign(10.0 ** rng.uniform(0, 2)),
        th_dot_final_cost_weight=sign(10.0 ** rng.uniform(0, 2)))
    if over["u_cost_weight"] == 0.0 and over["u_derivative_cost_weight"] == 0.0:
        over["u_cost_weight"] = 0.1
    if over["window_length"] * over["control_dt"] > 1.0:
        over["control_dt"] = 0.01
    dyn = [float(rng.uniform(0.5, 2.0)), float(rng.uniform(0.05, 0.3)), float(rng.uniform(0.15, 0.5)), 9.81,
           float(rng.choice([0.0, 0.05, 0.2])), float(rng.choice([1e-7, 0.05, 0.1])), float(rng.choice([0.0, 0.02, 0.1])),
           float(rng.uniform(0.5, 1.0)), float(rng.choice([0.0, 50.0, 100.0]))]
    return over, dyn, float(rng.uniform(-0.3, 0.3))
