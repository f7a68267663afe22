"""Golden control sequences of the full re-plan (this repo's SQP specification), written by the
CPU oracle: a frozen record of the specification.  NOT reference output (mini_opt is absent) -- it pins
the oracle against accidental changes of the specification and gives the GPU tests an oracle-independent target.
Regenerate (only when the specification is changed on purpose):  python tests/golden/gen_step_golden.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402

DYN_UI = [1.0, 0.1, 0.25, 9.81, 0.05, 0.1, 0.02, 0.8, 100.0]
DYN_TEST = [1.0, 0.1, 0.25, 9.81, 0.03, 0.1, 0.13, 0.8, 100.0]
CASES = [
    ("config2: N=40 sp=10, 5 iterations, exits off", dict(max_iterations=5, relative_exit_tol=0.0, absolute_first_derivative_tol=0.0), DYN_UI, 0.0),
    ("reference defaults (8 iterations, exits on)", dict(), DYN_UI, 0.0),
    ("optimization_test.cc:13-19: sp=5, 10 iterations", dict(state_spacing=5, max_iterations=10), DYN_TEST, 0.0),
    ("scratch.py:26-36", dict(window_length=20, max_iterations=30, u_cost_weight=0.0, b_x_final_cost_weight=5.0,
                             absolute_first_derivative_tol=1e-3, b_x_dot_final_cost_weight=100.0,
                             th_dot_final_cost_weight=100.0), [1.0, 0.1, 0.25, 9.81, 0.05, 0.1, 0.02, 0.8, 100.0], 0.0),
    ("all terminal rows costs, set-point 0.2", dict(max_iterations=5, th_final_cost_weight=50.0, b_x_dot_final_cost_weight=0.0,
                                                   th_dot_final_cost_weight=3.0), DYN_TEST, 0.2),
]


def main():
    rng = np.random.default_rng(20261003)
    out = []
    for tag, over, dyn, sp in CASES:
        B = 12
        x0 = np.stack([rng.uniform(-0.6, 0.6, B), rng.uniform(-np.pi, np.pi, B), rng.uniform(-1, 1, B), rng.uniform(-3, 3, B)])
        x0[1, ::2] = np.pi / 2 + rng.uniform(-0.4, 0.4, B // 2)
        u, _, st, it, _ = orc.step_batch_cold(orc.default_opt_params(**over), dyn, sp, x0)
        # how far the double oracle is from ITSELF in extended precision (oracle/cpmpc_oracle_ld.c): a problem that runs
        # 30 iterations without converging amplifies rounding until even the oracle is only reproducible to ~1e-5
        u_ld, _, _, _, eq = orc.step_batch_cold_ld(orc.default_opt_params(**over), dyn, sp, x0)
        out.append({"tag": tag, "params": over, "dyn": dyn, "set_point": sp, "x0": x0.tolist(), "u": u.tolist(),
                    "status": st.tolist(), "iterations": it.tolist(),
                    "oracle_vs_extended": np.abs(u - u_ld).max(axis=0).tolist(), "final_eq_l1": eq.tolist()})
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "step_golden.json")
    with open(path, "w") as fh:
        json.dump({"generator": "tests/golden/gen_step_golden.py (CPU oracle, fp64)", "cases": out}, fh)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
