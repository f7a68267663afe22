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


def test_bench_gpus_2_spawns_two_ranks_on_the_gpu(tmp_path):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["CPMPC_BENCH_SHARE_DEVICE"] = "1"
    detail = str(tmp_path / "bench_detail.json")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "8192", "--steps", "3",
                        "--warmup", "1", "--detail", detail], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    # stdout: the launcher relays rank 0's compact line and nothing else that looks like JSON
    json_lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(json_lines) == 1 and len(json_lines[0]) < 1900
    compact = json.loads(r.stdout.strip().splitlines()[-1])
    assert compact["n_gpus"] == 2 and compact["backend"] == "gloo" and compact["gather_ms"] > 0
    assert compact["config"]["global_batch"] == 16384 and compact["roofline"]["kernel"] == "fused_sqp_kernel"
    line = json.load(open(detail))      # the full record, written by rank 0 only
    assert compact["value"] == pytest.approx(line["value"], rel=1e-5)
    assert line["n_gpus"] == 2 and line["distributed"]["world_size_seen"] == 2 and line["distributed"]["spawned_by_bench"]
    assert line["config"]["global_batch"] == 16384 and line["distributed"]["shard_of_rank0"] == [0, 8192]
    assert line["gathered"]["shape"] == [40, 16384] and line["gathered"]["own_block_intact"]
    assert line["value"] > 0 and line["scaling"] == "weak" and line["steps"] == 3
