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


def _full_record(world=1):
    """A stand-in for the record run_rank() assembles (every key the compact line reads, plus bulk that must NOT travel)."""
    line = {"metric": "MPC re-plans/sec (whole node), N=40 horizon, 5 SQP iters, batch 256k", "value": 121234567.891 * world,
            "unit": "re-plans/s", "n_gpus": world, "steps": 20, "warmup": 5, "ms_per_step": 2.162345678, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "preheated": True,
            "config": {"workload": "BASELINE configs[3]: batch=262144 per GPU (2097152 in total), N=40, state_spacing=10, f32, "
                                   "cold start, 5 SQP iterations (exits disabled), u+predicted+status written, u gathered to "
                                   "rank 0 (RCCL)", "batch_per_gpu": 262144, "global_batch": 262144 * world, "horizon": 40,
                       "sqp_iterations": 5, "pipeline": "fused", "parallelism": "dp%d" % world},
            "roofline": {"bound": "valu", "kernel": "fused_sqp_kernel", "achieved": 74.212345, "peak": 157.3, "unit": "TFLOP/s",
                         "frac": 0.47181234, "traffic": 221098765.4321, "avg_launch_ms": 1.97923456, "launches": 20,
                         "note": "n" * 600, "hbm": {"x": 1}, "kernels_ms_per_step": {"a": 1.0}, "issue": {"note": "i" * 500}},
            "cpu_baseline": {"value": 63123.456789, "unit": "re-plans/s", "cores": 16, "kind": "port",
                             "one_core_value": 4061.23456, "parallel_efficiency": 0.97, "host": {"os_cpu_count": 256},
                             "sample": "first 262144 problems of rank 0's batch, same N=40/5-iteration cold-start workload, "
                                       "fp64, oracle/cpmpc_oracle.c with OpenMP (16 threads), 4.1 s " + "s" * 300},
            "fp64": {"value": 50234567.89, "roofline": {"frac": 0.38123456, "note": "x" * 500}},
            "parity_f64": {"lanes_over_1e-5": 0, "lanes": 262144, "arbiter": {"note": "a" * 300}},
            "variants": {"wide_qp_f32": {"wide_qp": {"re-plans/s": 117912345.6, "parity_vs_cpu_check": {"fraction_within_1e-2": 0.99426}},
                                         "default": {"re-plans/s": 122.0e6, "parity_vs_cpu_check": {"fraction_within_1e-2": 0.9366}}},
                         "double_pendulum": {"within_0.5rad": {"f64": {"re-plans/s": 23.3e6, "roofline": {"frac": 0.33}},
                                                               "f32": {"re-plans/s": 50.1e6}}},
                         "closed_loop_settled": {"note": "c" * 4000}}}
    if world > 1:
        line["distributed"] = {"backend": "nccl", "gather_ms": 0.31234567,
                               "per_rank": {"ms_per_step_own": [2.1] * world, "ms_per_step_own_min": 2.1012345,
                                            "ms_per_step_own_max": 2.1698765, "sqp_kernel_ms_per_launch": [1.98] * world,
                                            "gather_ms": [0.3] * world, "note": "p" * 300}}
    return line


@pytest.mark.parametrize("world", [1, 8])
def test_final_line_is_short(world, tmp_path):
    """the line the driver parses is < 1 900 bytes (it keeps a 2 000-character tail of stdout) and still
    carries the contract's keys, `roofline`, `cpu_baseline` and the summary scalars; the bulk goes to the detail file."""
    full = _full_record(world)
    assert len(json.dumps(full)) > 5000
    detail = bench.write_detail(full, str(tmp_path / "bench_detail.json"))
    assert detail and json.load(open(detail)) == json.loads(json.dumps(full))
    text = json.dumps(bench.compact_line(full, detail), separators=(",", ":"))
    assert len(text) < 1900 and "\n" not in text
    c = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in c, k
    assert c["n_gpus"] == world and abs(c["value"] / full["value"] - 1) < 1e-5
    assert set(c["roofline"]) == {"bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "launches"}
    assert abs(c["roofline"]["frac"] - c["roofline"]["achieved"] / c["roofline"]["peak"]) < 1e-3
    assert {"value", "unit", "cores", "kind", "sample", "one_core_value"} <= set(c["cpu_baseline"])
    assert c["config"]["workload"].startswith("BASELINE configs[") and "model" not in c["config"]
    assert c["fp64_value"] == pytest.approx(50234567.89, rel=1e-5) and c["fp64_frac"] == pytest.approx(0.381235, rel=1e-5)
    assert c["parity_f64_lanes_over_1e-5"] == 0 and c["wide_qp_f32_value"] == pytest.approx(117912345.6, rel=1e-5)
    assert c["wide_qp_f32_within_1e-2"] == pytest.approx(0.99426) and c["detail"] == "bench_detail.json"
    assert c["f32_within_1e-2"] == pytest.approx(0.9366)
    if world > 1:
        assert c["backend"] == "nccl" and c["rank_ms_per_step_max"] == pytest.approx(2.16988, rel=1e-5)
    assert "variants" not in c and "note" not in c["roofline"] and "distributed" not in c


def test_final_line_stays_short_when_strings_grow():
    """Optional keys are shed from the end and over-long strings cut before the line may pass the limit."""
    full = _full_record(8)
    full["config"]["workload"] = "w" * 3000
    text = json.dumps(bench.compact_line(full, "bench_detail.json"), separators=(",", ":"))
    assert len(text) < 1900
    c = json.loads(text)
    assert "roofline" in c and "cpu_baseline" in c and c["value"] > 0


def test_the_driver_command_does_not_run_the_soaks():
    """The 1 000-tick plain-SQP soaks (four child processes) run only behind --plain-sqp."""
    assert bench.parse_args(["--gpus", "1", "--steps", "20", "--warmup", "5"]).plain_sqp is False
    assert bench.parse_args(["--plain-sqp"]).plain_sqp is True


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
    assert line["per_rank"] == [[0.0, 0.0], [1.0, 10.0]]      # every rank's own timings reach the line


def test_launcher_spawns_world_of_eight(monkeypatch):
    """The driver's scaling run is `bench.py --gpus 8`: eight ranks, one gather to rank 0, eight entries in every per-rank
    list.  (On the GPU box at most six processes may use the card, so the eight-rank form is exercised here, on gloo.)"""
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    rc, out0 = bench.launch_ranks(8, ["--gpus", "8", "--batch", "32"], n_devices=8, script=STUB, timeout=600)
    assert rc == 0, out0
    line = json.loads(bench.last_json_line(out0))
    assert line["n_gpus"] == 8 and line["world_size_seen"] == 8
    assert line["gathered"] == [4, 256] and line["in_global_order"]
    assert line["local_ranks"] == list(range(8)) and line["max_rank"] == 7.0
    assert len(line["per_rank"]) == 8 and line["per_rank"][7] == [7.0, 70.0]


def test_a_rank_that_dies_ends_the_run_at_once(monkeypatch):
    """a non-zero rank that dies early must not leave rank 0 waiting in a collective until its timeout
    (the stub's other ranks sleep 120 s): the launcher polls all ranks, kills the rest and returns that exit code."""
    import time
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    for failing in (1, 0):
        monkeypatch.setenv("CPMPC_STUB_FAIL_RANK", str(failing))
        t0 = time.monotonic()
        rc, _ = bench.launch_ranks(2, ["--gpus", "2"], n_devices=2, script=STUB, timeout=300)
        assert rc == 3 and time.monotonic() - t0 < 60


def test_a_failing_rank_leaves_its_last_stderr_lines(monkeypatch, capfd):
    """when a rank exits non-zero the launcher prints that rank's last 40 stderr lines under its verdict
    (the stub's failing rank writes 61 lines; the first 21 must not be in the excerpt, the last one must)."""
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setenv("CPMPC_STUB_FAIL_RANK", "1")
    rc, _ = bench.launch_ranks(2, ["--gpus", "2"], n_devices=2, script=STUB, timeout=300)
    assert rc == 3
    err = capfd.readouterr().err
    assert "rank 1 exited with code 3" in err
    head, _, excerpt = err.partition("last 40 stderr line(s) of rank 1:")
    assert excerpt, err[-2000:]
    lines = [ln for ln in excerpt.splitlines() if ln.startswith("    | ")]
    assert len(lines) == 40
    assert lines[-1].endswith("stub rank 1: ncclCommInitRank failed (pretend)")
    assert "diagnostic line 20" not in excerpt and "diagnostic line 21" in excerpt
    assert "[rank 1] stub rank 1 diagnostic line 0" in head      # the full stderr of every rank is relayed as well


def test_device_report_names_device_backend_and_peers():
    """The line every rank writes before the timed region (no GPU needed: a stand-in for torch.cuda)."""
    class Props:
        name, pci_domain_id, pci_bus_id, pci_device_id = "AMD Instinct MI355X", 0, 0x85, 0

    class Cuda:
        @staticmethod
        def get_device_properties(i):
            return Props

        @staticmethod
        def can_device_access_peer(a, b):
            return b != 3

    class Torch:
        cuda = Cuda

    class Dist:
        @staticmethod
        def is_initialized():
            return True

        @staticmethod
        def get_backend():
            return "nccl"

        @staticmethod
        def get_world_size():
            return 8

    ln = bench.device_report(Torch, Dist, 2, 8, 2, 4, "nccl")
    assert "rank 2/8" in ln and "device 2 of 4" in ln and "0000:85:00" in ln and "MI355X" in ln
    assert "process group nccl world 8" in ln and "peer access to devices [1 1 - 0]" in ln


def test_launcher_overall_timeout(monkeypatch):
    import time
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setenv("CPMPC_STUB_FAIL_RANK", "99")       # nobody fails, everybody sleeps
    t0 = time.monotonic()
    rc, _ = bench.launch_ranks(2, ["--gpus", "2"], n_devices=2, script=STUB, timeout=20)
    assert rc == 124 and time.monotonic() - t0 < 60


def test_gpu_count_comes_from_sysfs_not_from_the_runtime(tmp_path, monkeypatch):
    """The launcher parent counts GPUs in the KFD topology (sysfs): no HIP call, no torch.cuda in that process."""
    for i, simd in enumerate((0, 0, 1024, 1024, 1024)):      # two CPU nodes, three GPUs
        d = tmp_path / str(i)
        d.mkdir()
        (d / "properties").write_text("cpu_cores_count %d\nsimd_count %d\nmem_banks_count 1\n" % (64 if simd == 0 else 0, simd))
    assert bench.kfd_gpu_nodes(str(tmp_path)) == 3
    assert bench.kfd_gpu_nodes(str(tmp_path / "missing")) is None
    import inspect
    src = inspect.getsource(bench.visible_gpus) + inspect.getsource(bench.kfd_gpu_nodes) + inspect.getsource(bench.launch_ranks)
    assert "import torch" not in src.replace('"import torch; print(torch.cuda.device_count())"', "")
    monkeypatch.setattr(bench, "kfd_gpu_nodes", lambda root=None: 8)
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "2,3")
    assert bench.visible_gpus() == 2
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    monkeypatch.delenv("CUDA_VISIBLE_DEVICES", raising=False)
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES", raising=False)
    assert bench.visible_gpus() == 8


def test_world_size_mismatch_fails_loudly():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr
