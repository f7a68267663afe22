"""Frozen control sequences of the full re-plan (tests/golden/step_golden.json, written by the CPU oracle): the oracle
must keep reproducing them (the specification of DESIGN.md section 4 has not drifted), and the GPU must match them
without the oracle in the loop."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN


@pytest.fixture(scope="module")
def cases():
    with open(os.path.join(GOLDEN, "step_golden.json")) as fh:
        return json.load(fh)["cases"]


def test_oracle_reproduces_the_frozen_specification(orc, cases):
    for c in cases:
        x0 = np.array(c["x0"])
        u, _, st, it, _ = orc.step_batch_cold(orc.default_opt_params(**c["params"]), c["dyn"], c["set_point"], x0)
        assert st.tolist() == c["status"] and it.tolist() == c["iterations"], c["tag"]
        assert np.abs(u - np.array(c["u"])).max() < 1e-9, c["tag"]


@pytest.mark.gpu
@pytest.mark.parametrize("pipeline", ["fused", "split"])
def test_gpu_matches_the_frozen_sequences(pkg, cases, pipeline):
    torch = pytest.importorskip("torch")
    for c in cases:
        x0 = np.array(c["x0"])
        B = x0.shape[1]
        opt = pkg.BatchOptimization(pkg.default_params(**c["params"]), max_batch=B, dtype=torch.float64, device=0)
        opt.set_pipeline(pipeline)
        out = opt.step(torch.tensor(x0, dtype=torch.float64, device="cuda:0"), c["dyn"], c["set_point"])
        assert out.status.cpu().tolist() == c["status"] and out.iterations.cpu().tolist() == c["iterations"], c["tag"]
        err = np.abs(out.u.cpu().numpy() - np.array(c["u"])).max(axis=0)
        # 1e-5 on every lane that was solved (terminated by a tolerance, or closed its defects) or stopped within ten
        # iterations.  A lane that ran 20-30 iterations WITHOUT converging (|c|_1 still > 1e-3: a failed solve) is an
        # expansive iteration through dozens of ill-conditioned step-length interpolations: measured on the two such
        # lanes of the fixture, the double oracle is itself only reproducible to 5e-8 / 3e-5 against its extended-
        # precision build (recorded as oracle_vs_extended) and the GPU, a few ulp per operation where the C library is
        # half an ulp, ends 1.1e-5 .. 1.4e-4 from it (round 3, both pipelines), the ratio steady from iteration to
        # iteration.  Those lanes (three of the sixty in the fixture): 3e-4.
        its, eq = np.array(c["iterations"]), np.array(c["final_eq_l1"])
        failed_long = (its > 10) & (eq > 1e-3)
        tol = np.where(failed_long, 3e-4, 1e-5)
        assert (err < tol).all(), (c["tag"], err, tol)
        assert failed_long.sum() <= len(its) // 4, c["tag"]   # a minority of hard problems per case
