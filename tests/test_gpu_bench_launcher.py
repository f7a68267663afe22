"""`python bench.py --gpus 2` the way the driver calls it (no WORLD_SIZE in the environment): the parent starts two
ranks, each solves its shard of the one global batch on the GPU, rank 0 gathers and prints the line.  On the one-GPU
test box both ranks share the device (CPMPC_BENCH_SHARE_DEVICE=1: the gather then goes through gloo, RCCL refuses two
ranks per device); three GPU processes in total, within the box's limit."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_bench_gpus_2_spawns_two_ranks_on_the_gpu():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["CPMPC_BENCH_SHARE_DEVICE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "8192", "--steps", "3",
                        "--warmup", "1"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["distributed"]["world_size_seen"] == 2 and line["distributed"]["spawned_by_bench"]
    assert line["config"]["global_batch"] == 16384 and line["distributed"]["shard_of_rank0"] == [0, 8192]
    assert line["gathered"]["shape"] == [40, 16384] and line["gathered"]["own_block_intact"]
    assert line["value"] > 0 and line["scaling"] == "weak" and line["steps"] == 3
