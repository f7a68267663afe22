"""bench.py --gpus N starts N ranks itself when WORLD_SIZE is unset (the way the driver calls it), before anything
touches the GPU; each rank takes its shard of the one global batch.  Driven here on CPU with gloo ranks."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

import bench

STUB = os.path.join(ROOT, "tests", "helpers", "rank_stub.py")


def test_rank_environments():
    envs = bench.rank_environments(4, 8, False, 12345)
    assert [e["RANK"] for e in envs] == ["0", "1", "2", "3"]
    assert [e["LOCAL_RANK"] for e in envs] == ["0", "1", "2", "3"]
    assert all(e["WORLD_SIZE"] == "4" and e["MASTER_ADDR"] == "127.0.0.1" and e["MASTER_PORT"] == "12345"
               and e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" for e in envs)
    with pytest.raises(SystemExit):           # more ranks than GPUs is refused ...
        bench.rank_environments(2, 1, False, 1)
    envs = bench.rank_environments(3, 2, True, 1)   # ... unless device sharing was asked for
    assert [e["LOCAL_RANK"] for e in envs] == ["0", "1", "0"]
    with pytest.raises(SystemExit):
        bench.rank_environments(1, 0, True, 1)      # no GPU: no CPU fallback


def test_last_json_line():
    assert bench.last_json_line("x\n{\"a\": 1}\nRCCL banner\n") == "{\"a\": 1}"
    assert bench.last_json_line("{not json}\n") is None


def test_global_batch_is_one_seeded_batch():
    full = bench.synth_states(bench.SEED, 64)
    assert np.array_equal(bench.synth_states(bench.SEED, 64, 16, 32), full[:, 16:32])


def test_launcher_spawns_world_of_two(monkeypatch):
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    rc, out0 = bench.launch_ranks(2, ["--gpus", "2", "--batch", "96"], n_devices=2, script=STUB, timeout=300)
    assert rc == 0, out0
    line = json.loads(bench.last_json_line(out0))
    assert line["n_gpus"] == 2 and line["world_size_seen"] == 2
    assert line["gathered"] == [4, 192] and line["in_global_order"]
    assert line["local_ranks"] == [0, 1] and line["max_rank"] == 1.0


def test_world_size_mismatch_fails_loudly():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr
