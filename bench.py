#!/usr/bin/env python3
"""bench.py -- MPC re-plans/sec of the batched cart-pole MPC hot path on MI355X.

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): batch = 262144
independent cart-poles per GPU, horizon N = 40, state_spacing = 10, fp32, cold start, exactly 5 SQP
iterations (exit tolerances disabled so every lane does the full fixed work), seeded random initial
states, shared dynamics parameters, outputs u [N,B], predicted states
[N,4,B] and status written.  A "step" = one full batched re-plan (Optimization::Step for every
problem).

`--gpus N` (BASELINE.json configs[3] at N = 8): one process per GPU.  Rank r solves problems
shard_range(N * 262144, r, N) of ONE seeded global batch (weak scaling, no data-path collective) and
the control sequences are gathered to rank 0 over RCCL.  When WORLD_SIZE is not in the environment
(plain `python bench.py --gpus N`) this process starts the N ranks itself -- before anything touches
the GPU -- and relays rank 0's line; under `python -m torch.distributed.run` it is one of the ranks.

For N > 1 the line carries every rank's own time per step, its SQP-kernel time and the stand-alone gather
(`distributed.per_rank`, `distributed.gather_ms`).  `--as-rank R --of W` solves exactly rank R's shard of the W-GPU
global batch alone on this GPU (no process group; `--parity-lanes K` holds K sampled lanes of it to the oracle).
The launcher counts GPUs in sysfs (never through the runtime), polls all ranks and ends the run with the first non-zero
exit code.

The line's `roofline.issue` / `roofline.clock_GHz_measured` come from a child process on a diagnostic build of the same
kernels (two clock stamps per wave, tools/kernel_clock.py), outside the timed region.

The same line also carries the record of the parity dtype (`fp64`: the reference computes in double
only, optimization/single_pendulum_dynamics.hpp:185): the same workload at the same batch in fp64,
timed the same way, with its own roofline and its control sequences compared with the CPU oracle.

Rank 0 prints ONE compact JSON line (< 1 900 bytes: the driver contract's keys, `roofline`, `cpu_baseline` and the
summary scalars of the parity dtype -- `compact_line`) and writes everything else the run measured (variants, notes,
per-rank tables) to `bench_detail.json` next to this script (`--detail PATH`; copied to gpurun_out/ when that exists).
"""
import argparse
import gc
import importlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

DYN_UI = [1.0, 0.1, 0.25, 9.81, 0.05, 0.1, 0.02, 0.8, 100.0]  # viz/src/application.ts:61-71

# Algorithmic work per re-plan and SQP iteration, SURVEY.md section 8(d) (restated in DESIGN.md):
FLOPS_LINEARIZE = 40 * (1560 + 140)   # 68 k: 40 x (RK4 with Jacobians + chain rule)
FLOPS_MERIT = 40 * 350                # 14 k: one Jacobian-free rollout
FLOPS_QP = 30_000                     # structured QP
FLOPS_PER_LAUNCH_UNIT = {
    "linearize_kernel": FLOPS_LINEARIZE,
    "qp_ls_kernel": FLOPS_QP + FLOPS_MERIT,   # nominal: one merit evaluation per iteration
    "prepare_kernel": FLOPS_MERIT,
    "finalize_kernel": FLOPS_MERIT,
    # one launch = all SQP iterations; multiplied by --iters below
    "fused_sqp_kernel": FLOPS_LINEARIZE + FLOPS_QP + FLOPS_MERIT,
}
# The same tally for the cart + double pendulum (BASELINE configs[4], NX = 6), derived the way SURVEY 8(d) derives the
# 4-state one: one evaluation of the dynamics with its Jacobians 249 flops (generated terms 74, 3 x 3
# LDL^T 19, eight solves of 15, right-hand sides 36) + 2 sincos; RK4 with sensitivities as integration.hpp:13-49 does it
# for D = 6 (four evaluations, three dense 6 x 6 products and 6 x 6 . 6 x 1, the sums) 2 916; the chain rule of
# optimization.cc:145-154 per step 396 + 66; a Jacobian-free step 4 x 79 + 78 = 394; the structured QP scaled from the
# 4-state 30 k by the mean of (6/4) and (6/4)^2.
FLOPS_NX6 = {"linearize": 40 * (2916 + 462), "merit": 40 * 394, "qp": 56_000}
FLOPS_PER_LAUNCH_UNIT_NX6 = {
    "linearize_kernel": FLOPS_NX6["linearize"], "qp_ls_kernel": FLOPS_NX6["qp"] + FLOPS_NX6["merit"],
    "prepare_kernel": FLOPS_NX6["merit"], "finalize_kernel": FLOPS_NX6["merit"],
    "fused_sqp_kernel": FLOPS_NX6["linearize"] + FLOPS_NX6["qp"] + FLOPS_NX6["merit"],
}
DYN_DOUBLE = [1.0, 0.1, 0.1, 0.25, 0.2, 9.81]   # m_b, m_1, m_2, l_1, l_2, g (symbolic/dynamics_double.py:13-22; the tests' values)
PEAK_VALU_TFLOPS = {"f32": 157.3, "f64": 78.6}  # MI355X_MICROARCH.md (vector peak); f64 = public spec
PEAK_HBM_GBPS = 8000.0
SEED = 1000


# ------------------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` with no WORLD_SIZE starts the N ranks (nothing here touches the GPU)
# ------------------------------------------------------------------------------------------------------------------
def _env_device_list():
    """Devices a *_VISIBLE_DEVICES variable leaves visible, or None when none is set."""
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return [t for t in v.split(",") if t.strip() != ""]
    return None


def kfd_gpu_nodes(root="/sys/class/kfd/kfd/topology/nodes"):
    """GPU nodes of the KFD topology (a node with simd_count > 0 is a GPU, the others are CPUs): read from sysfs,
    so counting them loads no runtime and creates no context."""
    n = 0
    try:
        for node in sorted(os.listdir(root)):
            try:
                with open(os.path.join(root, node, "properties")) as fh:
                    props = dict(ln.split()[:2] for ln in fh if len(ln.split()) >= 2)
                if int(props.get("simd_count", "0")) > 0:
                    n += 1
            except (OSError, ValueError):
                continue
    except OSError:
        return None
    return n


def visible_gpus():
    """Number of GPUs the ranks will see, found WITHOUT touching the HIP runtime in this (launcher) process: the
    KFD topology in sysfs, narrowed by a *_VISIBLE_DEVICES list; if sysfs is not there, a throw-away child process asks
    the runtime (the launcher itself never does)."""
    n = kfd_gpu_nodes()
    env = _env_device_list()
    if n is not None and n > 0:
        return min(n, len(env)) if env is not None else n
    try:
        r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"],
                           capture_output=True, text=True, timeout=300)
        return int(r.stdout.strip().splitlines()[-1])
    except Exception:  # noqa: BLE001
        return 0


def rank_environments(n, n_devices, share, port):
    """Environment of each of the n ranks.  One GPU per rank; with `share` (CPMPC_BENCH_SHARE_DEVICE=1, a
    rehearsal on a box with fewer GPUs than ranks) ranks wrap around the visible devices."""
    if n_devices < 1:
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    if n > n_devices and not share:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible (set CPMPC_BENCH_SHARE_DEVICE=1 to rehearse "
                         "with several ranks per GPU; the gather then goes through gloo)" % (n, n_devices))
    envs = []
    for r in range(n):
        e = dict(os.environ)
        e.update({"RANK": str(r), "LOCAL_RANK": str(r % n_devices), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                  "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0",
                  "CPMPC_BENCH_SPAWNED": "1"})
        envs.append(e)
    return envs


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n, argv, n_devices=None, script=None, timeout=None):
    """Start n fresh rank processes (never a re-exec of a process that has touched the GPU), wait for them and return
    (exit code, rank 0's stdout).  Rank 0's stdout is captured (read on a thread, so a full pipe never blocks it).  Every
    rank's stderr (and the stdout of ranks > 0) is TEED: a reader thread per rank forwards each line to this process's stderr
    as it arrives, prefixed "[rank r] " -- the device report, warnings and an RCCL error reach the log even if the driver
    kills the launcher before the run ends -- and keeps the last 40 lines, which are printed once more under the
    launcher's verdict when a rank fails (on an 8-GPU node the first contact with RCCL fails in one rank, and its message
    must not be lost among seven others').  All ranks are polled together: the first one to exit non-zero ends
    the run at once -- the others (which would otherwise sit in a collective until its timeout) are killed and that exit
    code is returned.  `timeout` (default CPMPC_BENCH_TIMEOUT or 3600 s) bounds the whole run the same way.  Never restarts
    anything."""
    import collections
    import threading
    share = os.environ.get("CPMPC_BENCH_SHARE_DEVICE", "0") == "1"
    if n_devices is None:
        n_devices = visible_gpus()
    if timeout is None:
        timeout = float(os.environ.get("CPMPC_BENCH_TIMEOUT", "3600"))
    envs = rank_environments(n, n_devices, share, free_port())
    cmd = [sys.executable, script or os.path.abspath(__file__)] + list(argv)
    procs = []
    chunks = []
    readers = []
    tails = [collections.deque(maxlen=40) for _ in range(n)]
    out_lock = threading.Lock()
    rc = 0
    failed_rank = None

    def tee(r, stream):
        for ln in iter(stream.readline, ""):
            ln = ln.rstrip("\n")
            tails[r].append(ln)
            with out_lock:
                sys.stderr.write("[rank %d] %s\n" % (r, ln))
                sys.stderr.flush()
        stream.close()

    try:
        for r, e in enumerate(envs):
            # rank 0: stdout is the result (captured), stderr is teed; ranks > 0: stdout and stderr both go to the tee
            p = subprocess.Popen(cmd, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE if r == 0 else subprocess.STDOUT,
                                 text=True, bufsize=1)
            procs.append(p)
            if r == 0:
                t0 = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
                t0.start()
                readers.append(t0)
                t = threading.Thread(target=tee, args=(0, p.stderr), daemon=True)
            else:
                t = threading.Thread(target=tee, args=(r, p.stdout), daemon=True)
            t.start()
            readers.append(t)
        deadline = time.monotonic() + timeout
        while True:
            codes = [p.poll() for p in procs]
            bad = [c for c in codes if c not in (None, 0)]
            if bad:
                rc = bad[0]
                failed_rank = codes.index(bad[0])
                with out_lock:
                    sys.stderr.write("bench.py launcher: rank %d exited with code %d; stopping the other ranks\n" % (failed_rank, bad[0]))
                break
            if all(c == 0 for c in codes):
                break
            if time.monotonic() > deadline:
                rc = 124
                with out_lock:
                    sys.stderr.write("bench.py launcher: no result after %.0f s; stopping the ranks\n" % timeout)
                break
            time.sleep(0.05)
    finally:
        for p in procs:   # exactly the processes started here
            if p.poll() is None:
                p.kill()
        for p in procs:
            try:
                p.wait(timeout=30)
            except subprocess.TimeoutExpired:
                pass
        for t in readers:
            t.join(timeout=30)
        if failed_rank is not None or rc == 124:
            which = [failed_rank] if failed_rank is not None else list(range(len(procs)))
            for r in which:
                tail = list(tails[r])
                sys.stderr.write("bench.py launcher: last %d stderr line(s) of rank %d:\n" % (len(tail), r))
                for ln in tail:
                    sys.stderr.write("    | %s\n" % ln)
        sys.stderr.flush()
    return rc, "".join(c for c in chunks if c)


def last_json_line(text):
    for ln in reversed(text.strip().splitlines()):
        ln = ln.strip()
        if ln.startswith("{") and ln.endswith("}"):
            try:
                json.loads(ln)
                return ln
            except ValueError:
                continue
    return None


# ------------------------------------------------------------------------------------------------------------------
# workload
# ------------------------------------------------------------------------------------------------------------------
def synth_states(seed, B, lo=0, hi=None):
    """Columns [lo, hi) of the seeded global batch of B initial states."""
    rng = np.random.default_rng(seed)
    x = np.stack([rng.uniform(-0.6, 0.6, B), rng.uniform(-np.pi, np.pi, B), rng.uniform(-1, 1, B),
                  rng.uniform(-3, 3, B)])
    return np.ascontiguousarray(x[:, lo:(B if hi is None else hi)])


def host_cpu_info():
    """Threads this process may run on, and what the cgroup grants (the GPU box hands out a share of a large host)."""
    info = {"os_cpu_count": os.cpu_count()}
    try:
        info["affinity"] = len(os.sched_getaffinity(0))
    except Exception:  # noqa: BLE001
        info["affinity"] = None
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, per = fh.read().split()
            if q != "max":
                quota = float(q) / float(per)
    except Exception:  # noqa: BLE001
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fh:
                q = float(fh.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh:
                per = float(fh.read())
            if q > 0:
                quota = q / per
        except Exception:  # noqa: BLE001
            quota = None
    info["cgroup_cpu_quota"] = quota
    usable = info["affinity"] or info["os_cpu_count"] or 1
    if quota:
        usable = max(1, min(usable, int(quota + 0.5)))
    info["usable"] = int(usable)
    return info


def cpu_baseline(x0_np, over, seconds_target=20.0):
    """The oracle (CPU restatement, fp64, OpenMP over problems) timed on this host on a bounded sample
    of the same workload.  Reported baseline only."""
    from oracle import oracle as orc
    p = orc.default_opt_params(**over)
    info = host_cpu_info()
    threads = info["usable"]
    probe = min(max(2048, 32 * threads), x0_np.shape[1])   # large enough that thread start-up does not bias the rate
    t0 = time.perf_counter()
    orc.step_batch_cold(p, DYN_UI, 0.0, x0_np[:, :probe], num_threads=threads)
    rate = probe / max(time.perf_counter() - t0, 1e-6)
    n = int(min(x0_np.shape[1], max(probe, rate * seconds_target)))
    t0 = time.perf_counter()
    u, _, st, _, used = orc.step_batch_cold(p, DYN_UI, 0.0, x0_np[:, :n], num_threads=threads)
    dt = time.perf_counter() - t0
    n1 = min(256, x0_np.shape[1])
    t1 = time.perf_counter()
    orc.step_batch_cold(p, DYN_UI, 0.0, x0_np[:, :n1], num_threads=1)
    one = n1 / (time.perf_counter() - t1)
    return {"value": n / dt, "unit": "re-plans/s", "cores": int(used), "kind": "port", "one_core_value": one,
            "parallel_efficiency": (n / dt) / (one * max(int(used), 1)), "host": info,
            "sample": "first %d problems of rank 0's batch, same N=40/5-iteration cold-start workload, fp64, "
                      "oracle/cpmpc_oracle.c with OpenMP (%d threads), %.1f s" % (n, int(used), dt)}, u, st, n


def timed_region(torch, dist, sharding, opt, x0, outs, gather, steps, warmup, dev, local_rank, distributed, preheat_s=0.0):
    """An untimed pre-heat of `preheat_s` seconds of the same steps (the device of a fresh lease runs its first tens of
    milliseconds 3 % slower than a warm one: the driver's 20-step run saw 117.3 M where 200 steps gave 122 M),
    then W untimed steps, then exactly K steps bracketed by barrier + synchronize on both sides.  Returns
    (max-over-ranks seconds, this rank's own seconds up to the end of its last step and gather -- before the closing
    barrier --, per-kernel HIP-event profile of the timed steps, outputs of the last step, its slot, pre-heat steps run)."""
    state = {"n": 0, "slot": 0}

    def one_step():
        slot = state["n"] % 2        # output buffers and gather slots advance together, warm-up included
        state["n"] += 1
        if gather is not None:
            gather.wait_slot(slot)   # the buffer we are about to overwrite has been sent
        opt.reset()                  # cold start: every step is a full re-plan from the sinusoid guess
        o = opt.step(x0, DYN_UI, 0.0, want_predicted=True, want_stats=True, out=outs[slot])
        if gather is not None:
            assert gather.submit(o.u) == slot
        state["slot"] = slot
        return o

    def fence():
        if gather is not None:
            gather.finish()
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    out = None
    n_pre = 0
    # pending destructors run NOW, before anything that is meant to keep the device busy up to the timed steps: a collection
    # between the warm-up and the timed region (where it sat until round 5) idles the device for tens of milliseconds, and
    # the first ~20 steps after such a gap run 4 % slower (the collector itself is off for the whole run: run_rank)
    gc.collect()
    if preheat_s > 0.0:
        # every rank runs the same number of pre-heat steps (the gather's slots advance together): rank 0's clock decides
        # in rounds of 32 steps, the decision travels through the max-reduce the timing uses anyway
        t_pre = time.perf_counter()
        where = dev if dist.is_initialized() and dist.get_backend() == "nccl" else "cpu"
        while True:
            for _ in range(32):
                out = one_step()
            n_pre += 32
            torch.cuda.synchronize()
            more = 1.0 if time.perf_counter() - t_pre < preheat_s else 0.0
            if distributed:
                more = sharding.max_over_ranks(more, where)
            if more <= 0.0 or n_pre >= 100000:
                break
    for _ in range(warmup):
        out = one_step()
    fence()   # synchronize + barrier + synchronize: the only gap between the last warm-up step and the first timed one
    opt.profile_enable(True)
    opt.profile_reset()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = one_step()
    if gather is not None:
        gather.finish()
    torch.cuda.synchronize()
    own = time.perf_counter() - t0      # this rank's work alone (diagnostic; the reported time is the one below)
    fence()
    elapsed = time.perf_counter() - t0
    elapsed = sharding.max_over_ranks(elapsed, dev if dist.is_initialized() and dist.get_backend() == "nccl" else "cpu")
    prof = opt.profile_read()
    opt.profile_enable(False)
    return elapsed, own, prof, out, state["slot"], n_pre


def roofline_of(prof, dtype, B, iters, steps, rate_per_gpu, nx=4):
    """Roofline object of the dominant kernel (by HIP-event device time over the timed region)."""
    dom = max(prof, key=lambda k: prof[k][0])
    dom_ms, dom_n = prof[dom]
    avg_s = dom_ms / max(dom_n, 1) * 1e-3
    table = FLOPS_PER_LAUNCH_UNIT if nx == 4 else FLOPS_PER_LAUNCH_UNIT_NX6
    flops_launch = table[dom] * B * (iters if dom == "fused_sqp_kernel" else 1)
    achieved_tf = flops_launch / avg_s / 1e12
    peak_tf = PEAK_VALU_TFLOPS[dtype]
    esz = 4 if dtype == "f32" else 8
    S = 5
    bytes_replan = esz * ((nx + 1) + (nx * S + 40 + 40 * nx)) + 4   # read x0 + set-point, write z + predicted, status
    traffic = None
    tpath = os.path.join(ROOT, "profiles", ("traffic_latest.json" if dtype == "f32" else "traffic_latest_%s.json" % dtype)
                         if nx == 4 else "traffic_latest_double_%s.json" % dtype)
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if tj.get("dtype") == dtype and tj.get("batch") == B and dom in tj.get("per_launch_bytes", {}):
                traffic = tj.get("per_launch_bytes", {}).get(dom)
        except Exception:  # noqa: BLE001
            traffic = None
    return {
        "bound": "valu", "kernel": dom, "achieved": round(achieved_tf, 3), "peak": peak_tf, "unit": "TFLOP/s",
        "frac": round(achieved_tf / peak_tf, 4), "traffic": traffic,
        "traffic_GBps": (round(traffic / avg_s / 1e9, 1) if traffic else None),
        "traffic_frac_of_hbm_peak": (round(traffic / avg_s / 1e9 / PEAK_HBM_GBPS, 4) if traffic else None),
        "avg_launch_ms": round(avg_s * 1e3, 4), "launches": int(dom_n),
        "algorithmic_flops_per_launch": flops_launch,
        "hbm": {"algorithmic_bytes_per_replan": bytes_replan,
                "achieved_GBps": round(bytes_replan * rate_per_gpu / 1e9, 3), "peak_GBps": PEAK_HBM_GBPS,
                "frac": round(bytes_replan * rate_per_gpu / 1e9 / PEAK_HBM_GBPS, 6)},
        "kernels_ms_per_step": {k: round(v[0] / steps, 4) for k, v in prof.items()},
        "note": "the path as specified is vector-ALU/transcendental bound (SURVEY.md 8d), neither HBM nor MFMA: "
                "achieved = SURVEY 8(d) algorithmic flops of the dominant kernel / its HIP-event time vs the "
                "vector peak. `traffic` = HBM bytes per launch of that kernel from separate rocprofv3 --pmc "
                "FETCH_SIZE / WRITE_SIZE passes (profiles/traffic_latest*.json)",
    }


# Issue cost of one wave64 vector instruction on one SIMD when enough waves share it (cycles; tools/ubench/clock.hip on
# MI355X, profiles/r03_clock_ubench.jsonl: wall-clock rates at 8 waves per SIMD; the guide's figures are 2 / 8 for the
# 32-bit classes).  Used only to express measured cycles as a fraction of the part's issue ceiling.
ISSUE_CYCLES = {"f32_arith": 2.0, "other_32bit": 2.0, "f32_trans": 8.0, "f64_arith": 4.0, "f64_trans": 16.0}


def clock_leg(B, timeout=600):
    """The shader clock the part holds while fused_sqp_kernel runs this workload, and the cycles a wave spends in it:
    tools/kernel_clock.py in a child process on the diagnostic build (-DCPMPC_FUSED_CLOCK: two stamps per wave, s_memtime
    and the constant 100 MHz s_memrealtime; tools/_build/lib_clock).  Outside the timed region, never the product library."""
    tool = os.path.join(ROOT, "tools", "kernel_clock.py")
    lib = os.path.join(ROOT, "tools", "_build", "lib_clock", "libcpmpc.so")
    if not os.path.exists(lib):
        return {"error": "tools/_build/lib_clock/libcpmpc.so is not built (python tools/kernel_clock.py --build-only)"}
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "CPMPC_LIB")}
    r = subprocess.run([sys.executable, tool, "--seconds", "1.5", "--batch", str(B)], env=env, capture_output=True,
                       text=True, timeout=timeout)
    recs = {}
    for ln in r.stdout.splitlines():
        ln = ln.strip()
        if ln.startswith("{"):
            try:
                d = json.loads(ln)
                recs[d["dtype"]] = d
            except ValueError:
                pass
    if not recs:
        return {"error": "kernel_clock.py gave no record (rc %d): %s" % (r.returncode, r.stderr[-300:])}
    return recs


def issue_of(clock_rec, dtype, B):
    """Cycles per vector instruction per SIMD and the fraction of the issue ceiling, from the clock leg's cycles per
    wave and the dynamic instruction mix of the last committed profile (profiles/traffic_latest*.json)."""
    if not clock_rec or "cycles_per_wave_mean" not in clock_rec:
        return None
    out = {"clock_GHz_measured": round(clock_rec["clock_GHz"], 4),
           "cycles_per_wave_mean": round(clock_rec["cycles_per_wave_mean"], 1),
           "us_per_wave_mean": round(clock_rec["us_per_wave_mean"], 2),
           "waves_per_simd": 2 if dtype == "f32" else 1,
           "ms_per_step_of_the_diagnostic_build": round(clock_rec["ms_per_step"], 4)}
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json" if dtype == "f32" else "traffic_latest_%s.json" % dtype)
    try:
        tj = json.load(open(tpath))
        mix = tj.get("instruction_mix_per_wave", {}).get("fused_sqp_kernel")
        if mix and tj.get("batch") == B and mix.get("valu", 0) > 0:
            floor = sum(ISSUE_CYCLES[k] * mix.get(k, 0.0) for k in ISSUE_CYCLES)
            per_simd = clock_rec["cycles_per_wave_mean"] / out["waves_per_simd"]
            out.update({"valu_instructions_per_wave": round(mix["valu"], 1),
                        "cycles_per_valu_instruction_per_simd": round(per_simd / mix["valu"], 3),
                        "issue_floor_cycles_per_wave": round(floor, 1),
                        "issue_frac": round(floor / per_simd, 4),
                        "mix_source": tj.get("source"),
                        "note": "issue_frac = sum over instruction classes of (dynamic count per wave x full-rate issue "
                                "cycles: 2 for 32-bit, 4 for fp64 arithmetic, 8 / 16 for f32 / f64 transcendentals) / (mean "
                                "cycles a wave is resident / waves per SIMD)"})
    except Exception:  # noqa: BLE001
        pass
    return out


def quiesce(torch):
    """Before a timed loop: run the pending destructors NOW.  A solver handle that has gone out of scope is destroyed when
    Python's collector gets to it, and destroying one frees gigabytes of workspace (hipFree synchronises the device) and a
    few thousand events -- seen landing inside the first closed loop of the default run, 1.95 -> 2.7 .. 2.95 ms per tick,
    and not with --steps 100, because the collector's trigger is a count of allocations."""
    gc.collect()
    torch.cuda.synchronize()


def variants(torch, pkg, args, tdt, dev, local_rank, x0, B):
    """Secondary measurements of SURVEY.md 8(d), outside the timed region and never `value`:
    (1) the same cold-start re-plan with the reference's exit tolerances enabled (optimization.hpp:30-34:
        lanes stop at SATISFIED_RELATIVE_TOL / SATISFIED_FIRST_ORDER_TOL; a wave ends with its last lane);
    (2) closed loop: 50 MPC ticks = warm-started re-plan + batched Simulator step (10 RK4 sub-steps), all
        state resident on the GPU (simulator.cc:11-36, optimization.cc:50-57);
    (3) one controller at 100 Hz through the C++ facade (viz/src/application.ts:393-399)."""
    res = {}
    p1 = pkg.default_params(max_iterations=args.iters)
    quiesce(torch)
    opt = pkg.BatchOptimization(p1, max_batch=B, dtype=tdt, device=local_rank)
    opt.set_pipeline(args.pipeline)
    out = pkg.BatchOutputs()
    for _ in range(3):
        opt.reset()
        opt.step(x0, DYN_UI, 0.0, want_predicted=True, want_stats=True, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        opt.reset()
        o = opt.step(x0, DYN_UI, 0.0, want_predicted=True, want_stats=True, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    st = o.status.cpu().numpy()
    res["exits_enabled"] = {"re-plans/s": B / dt, "ms_per_step": dt * 1e3, "stage_plan": opt.stage_plan(),
                            "mean_iterations": float(o.iterations.float().mean().item()),
                            "status_histogram": {pkg.capi.TERM_NAMES[int(c)]: int((st == c).sum()) for c in np.unique(st)},
                            "note": "cold start, max_iterations=%d, relative_exit_tol=1e-5, "
                                    "absolute_first_derivative_tol=1e-6 (reference defaults)" % args.iters}
    del opt
    # closed loop from near-upright states: re-plan (warm after the first tick) -> apply u_0 -> plant step
    ticks = 50
    rng = np.random.default_rng(7)
    xs = np.stack([rng.uniform(-0.3, 0.3, B), np.pi / 2 + rng.uniform(-0.4, 0.4, B), rng.uniform(-0.5, 0.5, B),
                   rng.uniform(-1, 1, B)])

    # the torch kernels the loops below call for their statistics, loaded before anything is timed (on a fresh box with a
    # cold page cache the first use of one costs tens of milliseconds: seen as 2.6 instead of 1.9 ms per tick once)
    _w = torch.zeros(8, dtype=torch.int32, device=dev)
    _ = _w.float().mean().item(), (_w.double() - 1.0).abs().median().item(), (_w.float() < 0.1).float().mean().item()

    def closed_loop(dt_torch, settle=0, **over):
        sim = pkg.BatchSimulator(B, dtype=dt_torch, device=local_rank)
        sim.set_state(torch.tensor(xs, dtype=dt_torch, device=dev))
        opt = pkg.BatchOptimization(pkg.default_params(**over), max_batch=B, dtype=dt_torch, device=local_rank)
        opt.set_pipeline(args.pipeline)
        out = pkg.BatchOutputs()   # this loop's own buffers (its dtype)
        for _ in range(settle):    # untimed: bring the controllers to their set-point first
            o = opt.step(sim.get_state(), DYN_UI, 0.0, want_predicted=False, want_stats=True, out=out)
            sim.step(DYN_UI, 0.01, o.u[0].contiguous())
        its = 0.0
        quiesce(torch)
        for _ in range(10 if settle else 0):   # a settled loop: a few more untimed ticks between the collection and the clock
            o = opt.step(sim.get_state(), DYN_UI, 0.0, want_predicted=False, want_stats=True, out=out)
            sim.step(DYN_UI, 0.01, o.u[0].contiguous())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(ticks):
            o = opt.step(sim.get_state(), DYN_UI, 0.0, want_predicted=False, want_stats=True, out=out)
            sim.step(DYN_UI, 0.01, o.u[0].contiguous())
            if k >= ticks - 10:
                its += o.iterations.float().mean().item() / 10
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / ticks
        err = (sim.get_state()[1] - np.pi / 2).abs()
        plan = opt.stage_plan()
        opt.close()
        return {"ticks/s (controllers x ticks)": B / dt, "ms_per_tick": dt * 1e3, "ticks": ticks, "stage_plan_last_tick": plan,
                "mean_iterations_last_10_ticks": its,
                "median_abs_pole_angle_error_after_0.5s": float(err.median().item()),
                "fraction_within_0.1rad_after_0.5s": float((err < 0.1).float().mean().item())}

    res["closed_loop_warm_start"] = closed_loop(tdt)
    res["closed_loop_warm_start"]["note"] = ("reference defaults (8 iterations max, exits enabled), warm start from the "
                                             "shifted previous solution, plant = 10 RK4 sub-steps per tick, states "
                                             "start within 0.4 rad of upright")
    if tdt == torch.float32:
        # Single precision cannot resolve the reference's absolute_first_derivative_tol = 1e-6 on its own: the merit
        # slope carries mu x |c|_1 of the rounding of the fp32 rollout.  Since round 4 the exit test counts residuals at
        # that floor as zero (cpmpc_solver_opts.exit_defect_floor): 1.8 iterations per settled tick instead of 2.8 (fp64:
        # 1.0).  With the tolerance at 1e-4 it is 1.0; the closed-loop accuracy (fp32 state storage, ~6e-6 rad) is the same
        res["closed_loop_warm_start_fo_tol_1e-4"] = closed_loop(tdt, absolute_first_derivative_tol=1e-4)
        res["closed_loop_warm_start_fp64"] = closed_loop(torch.float64)
    # the same loops once the controllers have settled (300 untimed ticks first): what a tick costs in steady state
    try:
        res["closed_loop_settled"] = {
            "note": "300 untimed ticks, then 50 timed; reference defaults; the stages of the fused pipeline planned per step "
                    "from the iteration histogram of an earlier step (DESIGN.md 6.4)",
            "f32" if tdt == torch.float32 else "f64": closed_loop(tdt, settle=300)}
        if tdt == torch.float32:
            res["closed_loop_settled"]["f32_fo_tol_1e-4"] = closed_loop(tdt, settle=300, absolute_first_derivative_tol=1e-4)
            res["closed_loop_settled"]["f64"] = closed_loop(torch.float64, settle=300)
        res["closed_loop_settled"]["column_ranges"] = closed_loop_ranges(torch, pkg, args, dev, local_rank, B, xs)
    except Exception as exc:  # noqa: BLE001
        res["closed_loop_settled"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
    try:
        res["per_problem_params"] = per_problem_variant(torch, pkg, args, dev, local_rank, B)
    except Exception as exc:  # noqa: BLE001
        res["per_problem_params"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
    if tdt == torch.float32:
        try:
            res["wide_qp_f32"] = wide_qp_variant(torch, pkg, args, dev, local_rank, B)
        except Exception as exc:  # noqa: BLE001
            res["wide_qp_f32"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
    try:
        res["double_pendulum"] = double_pendulum_variant(torch, pkg, args, dev, local_rank)
    except Exception as exc:  # noqa: BLE001
        res["double_pendulum"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
    try:
        res["shards_on_streams"] = split_streams_variant(torch, pkg, args, dev, local_rank, B)
    except Exception as exc:  # noqa: BLE001
        res["shards_on_streams"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
    if args.plain_sqp:   # four 1 000-tick soaks in child processes (~15 s): only when asked for, never in the driver's command
        try:
            quiesce(torch)
            res["plain_sqp"] = plain_sqp_variant()
        except Exception as exc:  # noqa: BLE001
            res["plain_sqp"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
    try:
        res["single_controller_facade"] = single_controller_latency(pkg)
    except Exception as exc:  # noqa: BLE001
        res["single_controller_facade"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
    return res


def per_problem_variant(torch, pkg, args, dev, local_rank, B, lanes=4096, steps=20):
    """SURVEY 8(f3) at benchmark scale: the configs[2] workload with PER-PROBLEM model parameters (+-10 % around the UI
    defaults), set-points and terminal weights (b_x a cost row of varying weight; the pole angle a cost row for every
    third controller, an equality otherwise).  That is another instantiation of the fused kernel (constants in vector
    registers instead of the kernel-argument segment, [9][B] loads) than the one the headline measures.  Both dtypes:
    re-plans/s, the kernel's HIP-event time, and the controls of a sample of lanes against the CPU check solved problem
    by problem (its own Optimization object per lane: different parameters AND different terminal rows)."""
    from oracle import oracle as orc
    over = dict(max_iterations=args.iters, relative_exit_tol=0.0, absolute_first_derivative_tol=0.0)
    rng = np.random.default_rng(SEED + 17)
    x_np = synth_states(SEED, B)
    dyn_np = np.array(DYN_UI)[:, None] * (1.0 + 0.1 * rng.uniform(-1.0, 1.0, (9, B)))
    sp_np = rng.uniform(-0.2, 0.2, B)
    tw_np = np.stack([150.0 * (1.0 + 0.2 * rng.uniform(-1.0, 1.0, B)), np.where(np.arange(B) % 3 == 0, 40.0, -1.0),
                      -np.ones(B), -np.ones(B)])
    res = {"note": "cold start, %d iterations, exits disabled; dyn [9][B] = UI defaults x (1 +- 0.1), set-point [B] in "
                   "+-0.2 m, terminal weights [4][B] (theta a cost row of weight 40 for every third controller)" % args.iters}
    for name, dt in (("f32", torch.float32), ("f64", torch.float64)):
        x0 = torch.tensor(x_np, dtype=dt, device=dev)
        dyn = torch.tensor(dyn_np, dtype=dt, device=dev)
        sp = torch.tensor(sp_np, dtype=dt, device=dev)
        tw = torch.tensor(tw_np, dtype=dt, device=dev)
        quiesce(torch)
        opt = pkg.BatchOptimization(pkg.default_params(**over), max_batch=B, dtype=dt, device=local_rank)
        opt.set_pipeline(args.pipeline)
        out = pkg.BatchOutputs()
        for _ in range(3):
            opt.reset()
            opt.step(x0, dyn, sp, out=out, terminal_weights=tw)
        torch.cuda.synchronize()
        opt.profile_enable(True)
        el, prof = None, None
        for _ in range(2):   # the faster of two timed runs: one stall of the box (seen: 85 ms once) does not become the record
            opt.profile_reset()
            t0 = time.perf_counter()
            for _ in range(steps):
                opt.reset()
                o = opt.step(x0, dyn, sp, out=out, terminal_weights=tw)
            torch.cuda.synchronize()
            el_k = (time.perf_counter() - t0) / steps
            if el is None or el_k < el:
                el, prof = el_k, opt.profile_read()
        opt.profile_enable(False)
        rec = {"re-plans/s": B / el, "ms_per_step": el * 1e3, "pipeline": opt.pipeline(),
               "kernels_ms_per_step": {k: round(v[0] / steps, 4) for k, v in prof.items()},
               "roofline": roofline_of(prof, name, B, args.iters, steps, B / el)}
        # the sample against the CPU check, one Optimization per lane (parameters and terminal rows differ per lane)
        idx = np.linspace(0, B - 1, min(lanes, B)).astype(np.int64)
        u_g = o.u[:, torch.as_tensor(idx, device=dev)].double().cpu().numpy()
        st_g = o.status.cpu().numpy()[idx]
        u_c = np.zeros_like(u_g)
        st_c = np.zeros(idx.size, dtype=np.int64)
        for j, i in enumerate(idx):
            w = tw_np[:, i]
            p = orc.default_opt_params(b_x_final_cost_weight=float(w[0]), th_final_cost_weight=float(w[1]),
                                       b_x_dot_final_cost_weight=float(w[2]), th_dot_final_cost_weight=float(w[3]), **over)
            so = orc.Optimization(p).step(x_np[:, i], dyn_np[:, i], float(sp_np[i]))
            u_c[:, j] = so.u
            st_c[j] = so.solver_outputs.termination_state
        ps, err = parity_stats(u_g, u_c, st_g, st_c)
        ps["bar"] = 1e-5 if name == "f64" else None
        ps["fraction_within_1e-2"] = float((err < 1e-2).mean())
        rec["parity_vs_cpu_check"] = ps
        res[name] = rec
        del opt
    return res


def wide_qp_variant(torch, pkg, args, dev, local_rank, B, lanes=8192, steps=20):
    """The headline's workload on a CPMPC_CREATE_WIDE_QP handle (float kernels that carry the whole terminal part of the QP in
    double, not only the NX x NX system) next to the default float handle: re-plans/s (the faster of two runs of `steps`), the
    fused kernel's HIP-event time, and the controls of `lanes` problems against the double CPU check -- what the option costs
    and what it buys."""
    from oracle import oracle as orc
    over = dict(max_iterations=args.iters, relative_exit_tol=0.0, absolute_first_derivative_tol=0.0)
    x_np = synth_states(SEED, B)
    x0 = torch.tensor(x_np, dtype=torch.float32, device=dev)
    idx = np.unique(np.linspace(0, B - 1, min(lanes, B)).astype(np.int64))
    u_c, _, st_c, _, _ = orc.step_batch_cold(orc.default_opt_params(**over), DYN_UI, 0.0, x_np[:, idx])
    res = {"note": "fp32, B = %d, cold start, %d iterations, exits disabled; parity = max |du| per problem against the double CPU "
                   "check on %d evenly spaced lanes" % (B, args.iters, idx.size)}
    for name, wide in (("default", None), ("wide_qp", True)):
        quiesce(torch)
        opt = pkg.BatchOptimization(pkg.default_params(**over), max_batch=B, dtype=torch.float32, device=local_rank, wide_qp=wide)
        opt.set_pipeline(args.pipeline)
        out = pkg.BatchOutputs()
        for _ in range(8):
            opt.reset()
            opt.step(x0, DYN_UI, 0.0, out=out)
        torch.cuda.synchronize()
        opt.profile_enable(True)
        el, prof = None, None
        for _ in range(2):
            opt.profile_reset()
            t0 = time.perf_counter()
            for _ in range(steps):
                opt.reset()
                o = opt.step(x0, DYN_UI, 0.0, out=out)
            torch.cuda.synchronize()
            el_k = (time.perf_counter() - t0) / steps
            if el is None or el_k < el:
                el, prof = el_k, opt.profile_read()
        opt.profile_enable(False)
        u_g = o.u[:, torch.as_tensor(idx, device=dev)].double().cpu().numpy()
        ps, err = parity_stats(u_g, u_c, o.status.cpu().numpy()[idx], st_c)
        ps["fraction_within_1e-2"] = float((err < 1e-2).mean())
        res[name] = {"re-plans/s": B / el, "ms_per_step": el * 1e3, "wide_qp": opt.wide_qp,
                     "fused_sqp_kernel_ms": round(prof["fused_sqp_kernel"][0] / max(prof["fused_sqp_kernel"][1], 1), 4),
                     "parity_vs_cpu_check": ps}
        opt.close()
        del opt
    res["throughput_ratio"] = res["wide_qp"]["re-plans/s"] / res["default"]["re-plans/s"]
    return res


def closed_loop_ranges(torch, pkg, args, dev, local_rank, B, xs, settle=300, ticks=100):
    """The settled closed loop held as 1, 2 and 3 independent column ranges (pkg.ClosedLoop: own handle, plant and stream
    each, no synchronisation between them): ms per tick (the best of three runs of `ticks`) and that controls and plant
    states are bitwise those of the single range.  Ranges drift out of phase, so one range's prepare / plant / finalize
    runs beside another's fused SQP kernel (which leaves issue slots idle at one wave per SIMD)."""
    res = {"note": "300 untimed ticks, then the best of 3 x %d timed; reference defaults; 262 144 controllers from within 0.4 rad of "
                   "upright" % ticks}
    for name, dt, parts in (("f64", torch.float64, (1, 2)), ("f32", torch.float32, (1, 3))):
        rec, ref = {}, None
        for n in parts:
            quiesce(torch)
            loop = pkg.ClosedLoop(pkg.default_params(), B, dtype=dt, device=local_rank, ranges=n, pipeline=args.pipeline)
            loop.set_state(torch.tensor(xs, dtype=dt, device=dev))
            for _ in range(settle):
                loop.tick(DYN_UI, 0.0)
            quiesce(torch)
            best = None
            for _ in range(3):
                t0 = time.perf_counter()
                for _ in range(ticks):
                    loop.tick(DYN_UI, 0.0)
                torch.cuda.synchronize()
                el = (time.perf_counter() - t0) / ticks
                best = el if best is None or el < best else best
            u, x = loop.controls(), loop.state()
            r = {"ms_per_tick": best * 1e3, "mean_iterations": float(loop.iterations().float().mean().item())}
            if ref is None:
                ref = (u.clone(), x.clone())
            else:
                r["bitwise_equal_to_one_range"] = bool(torch.equal(u, ref[0]) and torch.equal(x, ref[1]))
                r["speedup"] = rec["ranges_1"]["ms_per_tick"] / r["ms_per_tick"]
            rec["ranges_%d" % n] = r
            loop.close()
            del loop
        res[name] = rec
    return res


def double_pendulum_variant(torch, pkg, args, dev, local_rank, B=65536, lanes=4096, steps=20):
    """BASELINE configs[4] as a measured configuration: cart + double pendulum (6 states,
    symbolic/dynamics_double.py:53-148), B = 65 536, N = 40, state_spacing 10, 5 SQP iterations, exits disabled, cold start
    (zero control guess: the 10 N sinusoid throws the light poles over) from two state distributions -- the tests'
    near-upright one (both poles within 0.15 rad) and poles anywhere within 0.5 rad of upright -- in both dtypes:
    re-plans/s (HIP-event kernel times, the faster of two runs of `steps`), a roofline from the NX = 6 algorithmic flops,
    and the controls of `lanes` sampled problems against the CPU check (fp64 bar 1e-5), the lanes that are off handed to
    the extended-precision build of the check."""
    from oracle import oracle as orc
    over = dict(max_iterations=args.iters, relative_exit_tol=0.0, absolute_first_derivative_tol=0.0,
                u_guess_sinusoid_amplitude=0.0)
    rng = np.random.default_rng(SEED + 5)

    def states(spread):
        return np.stack([rng.uniform(-0.3, 0.3, B), np.pi / 2 + rng.uniform(-spread, spread, B),
                         np.pi / 2 + rng.uniform(-spread, spread, B), rng.uniform(-0.3, 0.3, B),
                         rng.uniform(-0.5, 0.5, B), rng.uniform(-0.5, 0.5, B)])

    res = {"note": "cart + double pendulum, B = %d, N = 40, spacing 10, %d iterations, exits disabled, cold start with a zero "
                   "control guess; dynamics parameters %s" % (B, args.iters, DYN_DOUBLE),
           "algorithmic_flops_per_replan": (FLOPS_NX6["linearize"] + FLOPS_NX6["qp"] + FLOPS_NX6["merit"]) * args.iters
           + 2 * FLOPS_NX6["merit"]}
    for start, spread in (("near_upright_0.15rad", 0.15), ("within_0.5rad", 0.5)):
        x_np = states(spread)
        idx = np.unique(np.linspace(0, B - 1, min(lanes, B)).astype(np.int64))
        u_c, _, st_c, it_c, _ = orc.step_batch_cold(orc.default_opt_params(**over), DYN_DOUBLE, 0.0, x_np[:, idx], model="double")
        rec = {}
        for name, dt in (("f32", torch.float32), ("f64", torch.float64)):
            x0 = torch.tensor(x_np, dtype=dt, device=dev)
            quiesce(torch)
            opt = pkg.BatchOptimization(pkg.default_params(**over), max_batch=B, dtype=dt, device=local_rank, model="double")
            opt.set_pipeline(args.pipeline)
            out = pkg.BatchOutputs()
            for _ in range(5):
                opt.reset()
                opt.step(x0, DYN_DOUBLE, 0.0, out=out)
            torch.cuda.synchronize()
            opt.profile_enable(True)
            el, prof = None, None
            for _ in range(2):
                opt.profile_reset()
                t0 = time.perf_counter()
                for _ in range(steps):
                    opt.reset()
                    o = opt.step(x0, DYN_DOUBLE, 0.0, out=out)
                torch.cuda.synchronize()
                el_k = (time.perf_counter() - t0) / steps
                if el is None or el_k < el:
                    el, prof = el_k, opt.profile_read()
            opt.profile_enable(False)
            r = {"re-plans/s": B / el, "ms_per_step": el * 1e3, "pipeline": opt.pipeline(),
                 "roofline": roofline_of(prof, name, B, args.iters, steps, B / el, nx=6)}
            u_g = o.u[:, torch.as_tensor(idx, device=dev)].double().cpu().numpy()
            st_g = o.status.cpu().numpy()[idx]
            ps, err = parity_stats(u_g, u_c, st_g, st_c)
            ps["bar"] = 1e-5 if name == "f64" else None
            ps["iterations_agree"] = int((o.iterations.cpu().numpy()[idx] == it_c).sum())
            ps["fraction_within_1e-2"] = float((err < 1e-2).mean())
            if name == "f64" and (err > 1e-5).any():
                off = np.where(err > 1e-5)[0][:64]
                u_ld, _, _, _, _ = orc.step_batch_cold_ld(orc.default_opt_params(**over), DYN_DOUBLE, 0.0, x_np[:, idx[off]],
                                                          model="double")
                e_g = np.abs(u_g[:, off] - u_ld).max(axis=0)
                e_c = np.abs(u_c[:, off] - u_ld).max(axis=0)
                ps["arbiter"] = {"lanes": int(off.size), "gpu_vs_extended_max": float(e_g.max()),
                                 "oracle_vs_extended_max": float(e_c.max()),
                                 "lanes_gpu_at_fault": int(((e_g > 1e-5) & (e_g > 2 * e_c)).sum())}
            r["parity_vs_cpu_check"] = ps
            rec[name] = r
            opt.close()
            del opt
        res[start] = rec
    return res


def plain_sqp_variant(timeout=900):
    """What the repo's additions to the textbook SQP cost or buy: tools/plain_sqp.py in two child
    processes -- the product with its defaults, and the -DCPMPC_SKIP_MERIT=0 build with full_step_below = 0 and
    exit_defect_floor = 0 (the iteration of rounds 1-2) -- 262 144 controllers, 1 000-tick swing-up soak + 50 settled ticks,
    both dtypes: ms/tick, iterations/tick, final pole error, solver failures."""
    tool = os.path.join(ROOT, "tools", "plain_sqp.py")
    lib = os.path.join(ROOT, "tools", "_build", "lib_noskip", "libcpmpc.so")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "CPMPC_LIB")}
    res = {"note": "reference tolerances (optimization.hpp:30-34), 8 iterations max; `defaults` = DESIGN.md section 4 as "
                   "shipped; `plain` = no full-step rule, no exit floor, merit evaluated on every step (-DCPMPC_SKIP_MERIT=0 "
                   "build): what a maintainer pays for the textbook iteration"}
    for name, e, extra in (("defaults", env, []),
                           ("plain", dict(env, CPMPC_LIB=lib), ["--full-step-below", "0", "--exit-defect-floor", "0"])):
        if name == "plain" and not os.path.exists(lib):
            res[name] = {"error": "tools/_build/lib_noskip/libcpmpc.so is not built"}
            continue
        r = subprocess.run([sys.executable, tool] + extra, env=e, capture_output=True, text=True, timeout=timeout)
        ln = last_json_line(r.stdout)
        res[name] = json.loads(ln) if ln else {"error": "rc %d: %s" % (r.returncode, r.stderr[-300:])}
    return res


def split_streams_variant(torch, pkg, args, dev, local_rank, B, steps=20):
    """Shards of the batch on concurrent streams of ONE GPU (never `value`): the headline's cold-start re-plan with the
    batch held by 2 / 4 handles of B / parts problems, each stepped on its own stream with no synchronisation between the
    shards.  A step is prepare (HBM-bound) -> fused SQP kernel (issue-bound, ending in a tail of partly filled compute
    units) -> finalize (HBM-bound); free-running shards drift apart, so the memory-bound kernels and the tail of one shard
    can overlap the arithmetic of another.  Measured: within +-3 % of the single handle in either dtype (profiles/r04_bench.json,
    r05_bench.json) -- nothing reliable for this workload, where prepare + finalize are 6 % of a step; the settled closed loop,
    where they and the plant are a quarter of a tick, does gain (closed_loop_ranges).  (Splitting ONE step into column
    ranges that fork from and join the caller's stream was built and measured in round 4: +0.2 % fp64, -1.5..-5 % fp32 --
    ranges that start together reach every phase together -- and removed.)
    Wall-clock re-plans/s per setting and that the controls are bitwise those of the single handle."""
    over = dict(max_iterations=args.iters, relative_exit_tol=0.0, absolute_first_derivative_tol=0.0)
    x_np = synth_states(SEED, B)
    res = {"note": "cold start, %d iterations, exits disabled, u + predicted written; parts = 1 is the headline's "
                   "configuration measured here the same way (wall clock over %d steps)" % (args.iters, steps)}
    for name, dt in (("f32", torch.float32), ("f64", torch.float64)):
        rec, u_ref = {}, None
        for parts in (1, 2, 4):
            Bp = B // parts
            opts = [pkg.BatchOptimization(pkg.default_params(**over), max_batch=Bp, dtype=dt, device=local_rank) for _ in range(parts)]
            for o in opts:
                o.set_pipeline(args.pipeline)
            xs = [torch.tensor(x_np[:, i * Bp:(i + 1) * Bp], dtype=dt, device=dev) for i in range(parts)]
            outs = [pkg.BatchOutputs() for _ in range(parts)]
            streams = [torch.cuda.Stream(device=dev) for _ in range(parts)]

            def step():
                for o, x, out, st in zip(opts, xs, outs, streams):
                    with torch.cuda.stream(st):
                        o.reset()
                        o.step(x, DYN_UI, 0.0, want_predicted=True, out=out)

            quiesce(torch)   # (collect first: the steps after a long idle gap run slower, see timed_region)
            for _ in range(8):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            torch.cuda.synchronize()
            el = (time.perf_counter() - t0) / steps
            r = {"re-plans/s": Bp * parts / el, "ms_per_step": el * 1e3}
            u = torch.cat([o.u for o in outs], dim=1)
            if parts == 1:
                u_ref = u
            else:
                r["bitwise_equal_to_one_handle"] = bool(torch.equal(u, u_ref[:, :Bp * parts]))
                r["speedup"] = rec["parts_1"]["ms_per_step"] / r["ms_per_step"]
            rec["parts_%d" % parts] = r
            del opts, xs, outs
        res[name] = rec
    return res


def single_controller_latency(pkg):
    """B = 1 through pypendulum (the reference's real workload: one Optimization::Step + one Simulator::Step per
    10 ms tick, viz/src/application.ts:393-399), wall time per call in a warm-started closed loop."""
    pyp = pkg.pypendulum()
    dyn = pyp.SingleCartPoleParams(*DYN_UI)
    op = pyp.OptimizationParams()
    opt = pyp.Optimization(op)
    sim = pyp.Simulator()
    zero = pyp.Vector2(0.0, 0.0)
    t_opt, t_sim = [], []
    for k in range(60):
        t0 = time.perf_counter()
        o = opt.step(sim.get_state(), dyn, 0.0)
        t1 = time.perf_counter()
        sim.step(dyn, 0.01, o.u[0], zero, zero)
        t2 = time.perf_counter()
        if k >= 10:
            t_opt.append(t1 - t0)
            t_sim.append(t2 - t1)
    return {"Optimization.step_ms_median": float(np.median(t_opt) * 1e3), "Optimization.step_ms_p95": float(np.quantile(t_opt, 0.95) * 1e3),
            "Simulator.step_ms_median": float(np.median(t_sim) * 1e3), "ticks": len(t_opt),
            "note": "one controller, reference defaults (N=40, spacing 10, <= 8 iterations, exits enabled), fp64, "
                    "host buffers in and out, swing-up from hanging"}


def parity_stats(u_gpu, u_cpu, st_gpu, st_cpu):
    err = np.abs(u_gpu - u_cpu).max(axis=0)
    return {"lanes": int(err.size), "max_abs_du_max": float(err.max()), "max_abs_du_median": float(np.median(err)),
            "max_abs_du_p99": float(np.quantile(err, 0.99)), "lanes_over_1e-5": int((err > 1e-5).sum()),
            "status_agree": int((st_gpu == st_cpu).sum())}, err


def arbiter(u_gpu, u_cpu, x0_np, over, worst=256):
    """Where GPU fp64 and the double oracle are furthest apart, which one moved?  The `worst` lanes by |u_gpu - u_oracle|
    re-solved by the SAME restatement in x87 extended precision (oracle/cpmpc_oracle_ld.c, 64-bit significand): the
    distance of each implementation from that answer is its own rounding sensitivity on the problem."""
    from oracle import oracle as orc
    err = np.abs(u_gpu - u_cpu).max(axis=0)
    idx = np.argsort(err)[-worst:]
    u_ld, _, _, _, eq = orc.step_batch_cold_ld(orc.default_opt_params(**over), DYN_UI, 0.0, x0_np[:, idx])
    e_gpu = np.abs(u_gpu[:, idx] - u_ld).max(axis=0)
    e_cpu = np.abs(u_cpu[:, idx] - u_ld).max(axis=0)
    return {"lanes": int(idx.size), "gpu_vs_oracle_max": float(err[idx].max()),
            "gpu_vs_extended_max": float(e_gpu.max()), "oracle_vs_extended_max": float(e_cpu.max()),
            "gpu_vs_extended_median": float(np.median(e_gpu)), "oracle_vs_extended_median": float(np.median(e_cpu)),
            "lanes_gpu_over_1e-5_vs_extended": int((e_gpu > 1e-5).sum()),
            "lanes_oracle_over_1e-5_vs_extended": int((e_cpu > 1e-5).sum()),
            "median_final_eq_l1_of_these_lanes": float(np.median(eq)),
            "note": "the %d lanes with the largest |u_gpu - u_oracle| re-solved in extended precision (same algorithm, "
                    "same constants)" % idx.size}


def device_report(torch, dist, rank, world, local_rank, n_dev, backend):
    """One line per rank, written to stderr BEFORE the timed region: which device this rank drives (ordinal, PCI bus id,
    name), the process group's backend and size as seen from here, and which of the other visible devices this one can
    reach peer to peer (the row of the peer-access matrix cpmpc_sharded_create would see).  Cheap insurance for the first
    run on an 8-GPU node: a wrong device mapping or a missing xGMI link shows here, not as a hang in the gather."""
    try:
        bus = torch.cuda.get_device_properties(local_rank)
        pci = "%04x:%02x:%02x" % (getattr(bus, "pci_domain_id", 0), getattr(bus, "pci_bus_id", -1) & 0xff, getattr(bus, "pci_device_id", 0))
        name = bus.name
    except Exception as exc:  # noqa: BLE001
        pci, name = "?", "? (%s)" % exc
    peers = []
    for j in range(n_dev):
        if j == local_rank:
            peers.append("-")
            continue
        try:
            peers.append("1" if torch.cuda.can_device_access_peer(local_rank, j) else "0")
        except Exception:  # noqa: BLE001
            peers.append("?")
    pg = "none"
    if backend is not None and dist.is_initialized():
        pg = "%s world %d" % (dist.get_backend(), dist.get_world_size())
    vis = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("CUDA_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES") or "all"
    return ("bench.py rank %d/%d: device %d of %d visible (%s) pci %s '%s'; process group %s; peer access to devices [%s]; "
            "HSA_ENABLE_IPC_MODE_LEGACY=%s" % (rank, world, local_rank, n_dev, vis, pci, name, pg, " ".join(peers),
                                                os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "unset")))


def run_rank(args):
    import torch
    import torch.distributed as dist

    # Python's cyclic collector is off for the whole run: a generation-2 pass of a process that has torch loaded takes
    # ~40 ms, and when it lands in a timed loop that synchronises per tick the device idles for all of it (seen: one block of
    # 50 settled ticks at 1.56 ms per tick between blocks at 0.77).  quiesce() and timed_region() collect explicitly, outside
    # the timed steps.
    gc.disable()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # --as-rank R --of W: this ONE process solves exactly what rank R of a W-GPU run would solve (its contiguous block
    # of the one seeded W x batch global batch) with no process group: configs[3]'s per-GPU work on the GPU at hand
    as_rank = args.as_rank is not None
    if as_rank:
        if args.of is None or not 0 <= args.as_rank < args.of:
            raise SystemExit("--as-rank R needs --of W with 0 <= R < W")
        if world != 1 or args.gpus != 1:
            raise SystemExit("--as-rank runs in one process on one GPU (no --gpus, no torch.distributed.run)")
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d but WORLD_SIZE=%d: launch one rank per GPU (python bench.py --gpus N starts "
                         "them itself; torch.distributed.run must use --nproc-per-node N)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    share = os.environ.get("CPMPC_BENCH_SHARE_DEVICE", "0") == "1"
    n_dev = torch.cuda.device_count()
    if local_rank >= n_dev:
        if not share:
            raise SystemExit("rank %d: LOCAL_RANK=%d but %d GPU(s) visible" % (rank, local_rank, n_dev))
        local_rank %= n_dev
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # CPMPC_BENCH_FORCE_DIST=1: run the RCCL path (process group, gather, barrier, max-reduce) in a world of one
    force_dist = os.environ.get("CPMPC_BENCH_FORCE_DIST", "0") == "1"
    distributed = world > 1 or force_dist
    # RCCL refuses two ranks on one device: a shared-device rehearsal gathers through gloo (host) instead
    backend = "gloo" if (share and world > n_dev) else "nccl"
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    pkg = importlib.import_module("cart-pole-mpc_amd")
    sharding = importlib.import_module("cart-pole-mpc_amd.sharding")
    if distributed or world > 1:
        sys.stderr.write(device_report(torch, dist, rank, world, local_rank, n_dev, backend if distributed else None) + "\n")
        sys.stderr.flush()
    tdt = torch.float32 if args.dtype == "f32" else torch.float64
    B = args.batch
    over = dict(max_iterations=args.iters, relative_exit_tol=0.0, absolute_first_derivative_tol=0.0)
    params = pkg.default_params(**over)
    N = int(params.window_length)

    shard_world, shard_rank = (args.of, args.as_rank) if as_rank else (world, rank)
    total = shard_world * B
    lo, hi = sharding.shard_range(total, shard_rank, shard_world)    # my contiguous block of the one global batch
    assert hi - lo == B
    x0_np = synth_states(SEED, total, lo, hi)
    x0 = torch.tensor(x0_np, dtype=tdt, device=dev)
    opt = pkg.BatchOptimization(params, max_batch=B, dtype=tdt, device=local_rank)
    opt.set_pipeline(args.pipeline)
    outs = [pkg.BatchOutputs(), pkg.BatchOutputs()]
    gather = None
    if distributed and not args.no_gather:
        gather = sharding.ResultGather(N, B, tdt, dev, dst=0, depth=2, force=force_dist,
                                       via_host=(backend == "gloo"))

    elapsed, own_s, prof, out, slot, n_pre = timed_region(torch, dist, sharding, opt, x0, outs, gather, args.steps, args.warmup,
                                                          dev, local_rank, distributed, preheat_s=args.preheat_seconds)

    # the gather on its own: the same [N, B] block from every rank to rank 0, nothing else in flight
    gather_ms = None
    if gather is not None:
        reps = 5
        gather.finish()
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            s_ = gather.submit(out.u)
            gather.wait_slot(s_)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - t0) / reps * 1e3

    # per-rank diagnostics (one small all-gather, outside the timed region): each rank's own time per step, the
    # average launch time of its SQP kernel, and its view of the stand-alone gather
    per_rank = None
    if distributed:
        fused = prof.get("fused_sqp_kernel", (0.0, 0))
        per_rank = sharding.all_ranks([own_s / args.steps * 1e3, fused[0] / max(fused[1], 1),
                                       gather_ms if gather_ms is not None else -1.0],
                                      dev if dist.get_backend() == "nccl" else "cpu")

    if rank != 0:
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
        return

    value = world * B * args.steps / elapsed    # problems all ranks of THIS run solved per second
    line = {
        "metric": "MPC re-plans/sec (whole node), N=40 horizon, 5 SQP iters, batch 256k",
        "value": value, "unit": "re-plans/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "preheated": args.preheat_seconds > 0.0, "preheat_s": args.preheat_seconds, "preheat_steps": n_pre,
        "config": {"workload": "BASELINE configs[%d]%s: batch=%d per GPU (%d in total), N=40, state_spacing=10, %s, cold start, "
                               "%d SQP iterations (exits disabled), u+predicted+status written%s%s"
                               % (2 if shard_world == 1 else 3,
                                  " -- the shard of rank %d of %d, solved alone on this GPU" % (shard_rank, shard_world)
                                  if as_rank else "", B, total, args.dtype, args.iters,
                                  ", u gathered to rank 0 (%s)" % ("RCCL" if backend == "nccl" else "gloo") if gather else "",
                                  ", QP terminal part in double" if (args.dtype == "f32" and opt.wide_qp) else ""),
                   "batch_per_gpu": B, "global_batch": total, "horizon": N, "sqp_iterations": args.iters,
                   "pipeline": opt.pipeline(), "parallelism": "dp%d" % world},
        "roofline": roofline_of(prof, args.dtype, B, args.iters, args.steps, value / world),
    }
    if as_rank:
        line["as_rank"] = {"rank": shard_rank, "of": shard_world, "columns_of_global_batch": [lo, hi],
                           "note": "value is this one GPU's rate on that shard; no process group, no gather"}
    if distributed:
        own_ms = [r[0] for r in per_rank]
        line["distributed"] = {"world_size_seen": dist.get_world_size(), "backend": dist.get_backend(),
                               "spawned_by_bench": os.environ.get("CPMPC_BENCH_SPAWNED", "0") == "1",
                               "shard_of_rank0": [lo, hi], "gather_ms": gather_ms,
                               "devices_visible": n_dev,
                               "per_rank": {"ms_per_step_own": own_ms, "ms_per_step_own_min": min(own_ms),
                                            "ms_per_step_own_max": max(own_ms),
                                            "sqp_kernel_ms_per_launch": [r[1] for r in per_rank],
                                            "gather_ms": [r[2] if r[2] >= 0 else None for r in per_rank],
                                            "note": "own = W..K steps + gather on that rank up to its own synchronize, "
                                                    "before the closing barrier; ms_per_step is the max over ranks "
                                                    "including the barrier"}}
    # everything below is secondary: a failure in one leg is recorded, never raised (the primary line must print)
    try:
        st = out.status.cpu().numpy()
        line["status_histogram"] = {pkg.capi.TERM_NAMES[int(c)]: int((st == c).sum()) for c in np.unique(st)}
        line["mean_merit_evals_per_iter"] = float(out.ls_evals.float().mean().item() / args.iters)
    except Exception as exc:  # noqa: BLE001
        line["status_histogram"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
    if gather is not None:
        try:
            # what rank 0 holds after the last step: the control sequences of all ranks, in global problem order
            full = gather.assembled(slot)
            line["gathered"] = {"shape": list(full.shape), "own_block_intact": bool(torch.equal(full[:, :B].to(out.u.device), out.u))}
            del full
        except Exception as exc:  # noqa: BLE001
            line["gathered"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    u_primary = out.u
    eq_primary = out.final_eq_l1

    # ---- the parity dtype as a first-class record: same workload, same batch, fp64, timed the same way ----------
    u64 = st64 = None
    if as_rank and args.parity_lanes > 0:
        # sampled lanes of this shard against the CPU oracle on the same columns of the global batch (checker only)
        try:
            from oracle import oracle as orc
            idx = np.unique(np.linspace(0, B - 1, args.parity_lanes).astype(np.int64))
            u_cpu, _, st_cpu, _, _ = orc.step_batch_cold(orc.default_opt_params(**over), DYN_UI, 0.0, x0_np[:, idx])
            u_g = out.u.double().cpu().numpy()[:, idx]
            ps, _ = parity_stats(u_g, u_cpu, st[idx], st_cpu)
            ps.update({"bar": 1e-5, "global_columns_first_last": [int(lo + idx[0]), int(lo + idx[-1])]})
            line["as_rank"]["parity"] = ps
        except Exception as exc:  # noqa: BLE001
            line["as_rank"]["parity"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
    if as_rank:   # the secondary legs belong to the headline run
        args.no_fp64 = args.no_variants = args.no_cpu_baseline = args.no_clock = True
    if world == 1 and args.dtype == "f32" and not args.no_fp64:
        try:
            opt64 = pkg.BatchOptimization(params, max_batch=B, dtype=torch.float64, device=local_rank)
            opt64.set_pipeline(args.pipeline)
            x64 = torch.tensor(x0_np, dtype=torch.float64, device=dev)
            outs64 = [pkg.BatchOutputs(), pkg.BatchOutputs()]
            el64, _, prof64, o64, _, n_pre64 = timed_region(torch, dist, sharding, opt64, x64, outs64, None, args.steps,
                                                            args.warmup, dev, local_rank, False,
                                                            preheat_s=min(args.preheat_seconds, 1.0))
            v64 = B * args.steps / el64
            line["fp64"] = {"value": v64, "unit": "re-plans/s", "dtype": "f64", "steps": args.steps, "warmup": args.warmup,
                            "ms_per_step": el64 / args.steps * 1e3, "batch": B, "pipeline": opt64.pipeline(),
                            "preheat_steps": n_pre64,
                            "roofline": roofline_of(prof64, "f64", B, args.iters, args.steps, v64),
                            "note": "the parity dtype (the reference computes in double only): the workload of the "
                                    "timed region in fp64 at the same batch, timed the same way"}
            u64 = o64.u.cpu().numpy()
            st64 = o64.status.cpu().numpy()
            opt64.close()   # now, not whenever the collector gets to it (a 2 GB hipFree inside somebody's timed loop)
            del opt64, x64, outs64, o64
        except Exception as exc:  # noqa: BLE001
            line["fp64"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
    if world == 1 and not args.no_clock and not as_rank and args.pipeline != "split":
        # the clock the part holds under this kernel and what that makes of the issue ceiling
        try:
            recs = clock_leg(B)
            if "error" in recs:
                line["roofline"]["issue"] = recs
            else:
                line["roofline"]["issue"] = issue_of(recs.get(args.dtype), args.dtype, B)
                line["roofline"]["clock_GHz_measured"] = (line["roofline"]["issue"] or {}).get("clock_GHz_measured")
                if isinstance(line.get("fp64"), dict) and "roofline" in line["fp64"] and "f64" in recs:
                    line["fp64"]["roofline"]["issue"] = issue_of(recs["f64"], "f64", B)
                    line["fp64"]["roofline"]["clock_GHz_measured"] = recs["f64"]["clock_GHz"]
        except Exception as exc:  # noqa: BLE001
            line["roofline"]["issue"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
    if world == 1 and not args.no_variants:
        try:
            line["variants"] = variants(torch, pkg, args, tdt, dev, local_rank, x0, B)
        except Exception as exc:  # noqa: BLE001
            line["variants"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
    if not args.no_cpu_baseline and world == 1:
        try:
            base, u_cpu, st_cpu, n = cpu_baseline(x0_np, over)
            line["cpu_baseline"] = base
            line["gpu_over_cpu"] = value / base["value"]
        except Exception as exc:  # noqa: BLE001
            base = None
            line["cpu_baseline"] = {"value": None, "unit": "re-plans/s", "cores": 0, "kind": "port",
                                    "sample": "failed: %s: %s" % (type(exc).__name__, exc)}
        if base is not None:
            try:
                ps, err = parity_stats(u_primary[:, :n].double().cpu().numpy(), u_cpu, st[:n], st_cpu)
                cl1 = eq_primary[:n].double().cpu().numpy()
                conv = cl1 < 1e-3   # lanes whose shooting defects have closed after the fixed 5 iterations
                ps.update({"fraction_within_1e-2": float((err < 1e-2).mean()), "converged_lanes": int(conv.sum()),
                           "max_abs_du_median_on_converged_lanes": float(np.median(err[conv])) if conv.any() else None,
                           "note": "GPU %s vs fp64 oracle on the cpu_baseline sample; these cold-start swing-up problems "
                                   "are far from converged after 5 iterations (median |c|_1 %.1f) and the SQP iteration "
                                   "amplifies rounding differences there; the parity bar is the fp64 one (parity_f64)"
                                   % (args.dtype, float(np.median(cl1)))})
                line["parity_sample"] = ps
            except Exception as exc:  # noqa: BLE001
                line["parity_sample"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
            if args.dtype == "f32":
                # the same lanes against the SINGLE-PRECISION build of the CPU check (oracle/cpmpc_oracle_f32.c): how far
                # the float kernels are from a same-precision answer, next to how far that answer is from the double one
                try:
                    from oracle import oracle as orc
                    m = min(n, 8192)
                    u_f, st_f, _, _, _ = orc.step_batch_cold_f32(orc.default_opt_params(**over), DYN_UI, 0.0, x0_np[:, :m],
                                                                 num_threads=base["cores"])
                    u_g = u_primary[:, :m].double().cpu().numpy()
                    e_gf = np.abs(u_g - u_f).max(axis=0)
                    e_fd = np.abs(u_f - u_cpu[:, :m]).max(axis=0)
                    e_gd = np.abs(u_g - u_cpu[:, :m]).max(axis=0)
                    q = lambda e: {"median": float(np.median(e)), "p90": float(np.quantile(e, 0.9)), "p99": float(np.quantile(e, 0.99)),  # noqa: E731
                                   "within_1e-2": float((e < 1e-2).mean())}
                    line["parity_vs_f32_check"] = {
                        "lanes": int(m), "gpu_f32_vs_cpu_f32": q(e_gf), "cpu_f32_vs_cpu_f64": q(e_fd), "gpu_f32_vs_cpu_f64": q(e_gd),
                        "status_agree_gpu_f32_vs_cpu_f32": int((st[:m] == st_f).sum()),
                        "note": "max |du| per lane, three pairings of the same lanes.  The single-precision CPU check is the same "
                                "restatement compiled in float with its dense KKT solve in double (the kernels keep only "
                                "their terminal system in double and take sin / cos / exp from the hardware's approximations): "
                                "read cpu_f32_vs_cpu_f64 as what single precision costs this algorithm at best, and "
                                "gpu_f32_vs_cpu_f32 as what the float kernels add to it"}
                except Exception as exc:  # noqa: BLE001
                    line["parity_vs_f32_check"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
            if u64 is not None:
                try:
                    ps64, _ = parity_stats(u64[:, :n], u_cpu, st64[:n], st_cpu)
                    ps64.update({"bar": 1e-5, "batch": B,
                                 "note": "GPU fp64 (the fp64 record above, all of its lanes the CPU sample covers) vs the "
                                         "fp64 oracle, same inputs"})
                    line["parity_f64"] = ps64
                    line["parity_f64"]["arbiter"] = arbiter(u64[:, :n], u_cpu, x0_np[:, :n], over)
                except Exception as exc:  # noqa: BLE001
                    line["parity_f64"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
    # RCCL writes a version banner through C stdio, which is flushed at exit: flush it now so that the JSON line
    # is the last line on stdout
    import ctypes
    sys.stdout.flush()
    ctypes.CDLL(None).fflush(None)
    detail_path = write_detail(line, args.detail)
    print(json.dumps(compact_line(line, detail_path), separators=(",", ":")), flush=True)


def _sig(v, digits=6):
    """A float at `digits` significant digits (the compact line is read by people and by a 2 000-character tail)."""
    if isinstance(v, bool) or not isinstance(v, float):
        return v
    if v != v or v in (float("inf"), float("-inf")):
        return None
    return float("%.*g" % (digits, v))


def _dig(d, *path):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return d


COMPACT_LIMIT = 1900   # bytes; the driver keeps a 2 000-character tail of stdout and parses its last line


def compact_line(line, detail_path=None):
    """The ONE line stdout carries: the driver contract's keys, `roofline` and `cpu_baseline` without
    their notes, and the summary scalars of the parity dtype and of the float kernels' parity option.  Everything else
    the run measured (variants, notes, per-rank tables) is the detail record `write_detail` leaves next to this script.
    Never longer than COMPACT_LIMIT bytes: optional keys are dropped from the end until it fits."""
    c = {k: _sig(line.get(k)) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                                        "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    cfg = line.get("config") or {}
    c["config"] = {k: cfg.get(k) for k in ("workload", "batch_per_gpu", "global_batch", "horizon", "sqp_iterations",
                                           "pipeline", "parallelism")}
    r = line.get("roofline") or {}
    c["roofline"] = {k: _sig(r.get(k)) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic",
                                                 "avg_launch_ms", "launches")}
    b = line.get("cpu_baseline")
    if isinstance(b, dict):
        c["cpu_baseline"] = {k: _sig(b.get(k)) for k in ("value", "unit", "cores", "kind", "one_core_value")}
        c["cpu_baseline"]["sample"] = (b.get("sample") or "")[:150]
    # the parity dtype (fp64: the reference's own arithmetic) and what the float kernels' accuracy option costs / buys
    opt = [("fp64_value", _dig(line, "fp64", "value")),
           ("fp64_frac", _dig(line, "fp64", "roofline", "frac")),
           ("parity_f64_lanes_over_1e-5", _dig(line, "parity_f64", "lanes_over_1e-5")),
           ("parity_f64_lanes", _dig(line, "parity_f64", "lanes")),
           ("wide_qp_f32_value", _dig(line, "variants", "wide_qp_f32", "wide_qp", "re-plans/s")),
           ("wide_qp_f32_within_1e-2", _dig(line, "variants", "wide_qp_f32", "wide_qp", "parity_vs_cpu_check", "fraction_within_1e-2")),
           ("f32_within_1e-2", _dig(line, "variants", "wide_qp_f32", "default", "parity_vs_cpu_check", "fraction_within_1e-2")),
           ("config5_f64_value", _dig(line, "variants", "double_pendulum", "within_0.5rad", "f64", "re-plans/s")),
           ("config5_f64_frac", _dig(line, "variants", "double_pendulum", "within_0.5rad", "f64", "roofline", "frac")),
           ("config5_f32_value", _dig(line, "variants", "double_pendulum", "within_0.5rad", "f32", "re-plans/s")),
           ("gather_ms", _dig(line, "distributed", "gather_ms")),
           ("backend", _dig(line, "distributed", "backend")),
           ("rank_ms_per_step_min", _dig(line, "distributed", "per_rank", "ms_per_step_own_min")),
           ("rank_ms_per_step_max", _dig(line, "distributed", "per_rank", "ms_per_step_own_max")),
           ("as_rank", [line["as_rank"]["rank"], line["as_rank"]["of"]] if isinstance(line.get("as_rank"), dict) else None),
           ("detail", os.path.basename(detail_path) if detail_path else None)]
    for k, v in opt:
        if v is not None:
            c[k] = _sig(v)
    order = [k for k, _ in opt]
    while len(json.dumps(c, separators=(",", ":"))) >= COMPACT_LIMIT and order:
        c.pop(order.pop(), None)
    if len(json.dumps(c, separators=(",", ":"))) >= COMPACT_LIMIT:   # a workload string somebody made long: cut it, keep the numbers
        c["config"]["workload"] = (c["config"].get("workload") or "")[:200]
        if isinstance(c.get("cpu_baseline"), dict):
            c["cpu_baseline"]["sample"] = c["cpu_baseline"]["sample"][:60]
    return c


def write_detail(line, path):
    """The full record of the run (what rounds 1-5 printed as one 25 KB line), written by rank 0 only, atomically; also
    copied to gpurun_out/ when that directory exists (the only thing a gpurun call brings home).  Returns the path, or
    None when nothing could be written -- the compact line prints regardless."""
    if not path:
        return None
    text = json.dumps(line, indent=1)
    done = None
    for p in (path, os.path.join(ROOT, "gpurun_out", os.path.basename(path))):
        if p != path and not os.path.isdir(os.path.dirname(p)):
            continue
        try:
            tmp = "%s.tmp.%d" % (p, os.getpid())
            with open(tmp, "w") as fh:
                fh.write(text + "\n")
            os.replace(tmp, p)
            done = done or p
        except OSError as exc:
            sys.stderr.write("bench.py: could not write %s: %s\n" % (p, exc))
    return done


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=262144, help="problems per GPU")
    ap.add_argument("--dtype", choices=["f32", "f64"], default="f32")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--preheat-seconds", type=float, default=2.0, help="untimed pre-heat of the same steps before the "
                    "warm-up (0 disables); the timed region stays exactly --steps after --warmup")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--no-fp64", action="store_true", help="skip the fp64 record")
    ap.add_argument("--no-variants", action="store_true", help="skip the secondary measurements (SURVEY 8d)")
    ap.add_argument("--no-clock", action="store_true", help="skip the shader-clock / issue-ceiling leg")
    ap.add_argument("--plain-sqp", action="store_true", help="also run tools/plain_sqp.py's soaks (the price of the solver "
                    "specification's three additions, DESIGN.md section 4) into the detail record")
    ap.add_argument("--detail", default=os.path.join(ROOT, "bench_detail.json"), help="where rank 0 writes the full record "
                    "(variants, notes, per-rank tables); stdout carries only the compact line")
    ap.add_argument("--pipeline", choices=["auto", "split", "fused"], default="auto")
    ap.add_argument("--as-rank", type=int, default=None, help="with --of W: solve rank R's shard of the W-GPU global "
                    "batch alone on this GPU (no process group)")
    ap.add_argument("--of", type=int, default=None)
    ap.add_argument("--parity-lanes", type=int, default=0, help="with --as-rank: compare this many sampled lanes of the "
                    "shard with the CPU oracle (fp64 bar 1e-5)")
    return ap.parse_args(argv)


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this process only starts the ranks and relays rank 0's line
        rc, out0 = launch_ranks(args.gpus, sys.argv[1:])
        ln = last_json_line(out0)
        if ln is None or rc != 0:
            sys.stderr.write(out0)
            raise SystemExit(rc or 1)
        print(ln, flush=True)
        return
    run_rank(args)


if __name__ == "__main__":
    main()
