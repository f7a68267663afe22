#!/usr/bin/env python3
"""Summarise rocprofv3 output of `tools/prof.sh <tag>` (gpurun_out/prof_<tag>/) into profiles/.

Writes
  profiles/<tag>_kernel_stats.csv    the --kernel-trace --stats table (our kernels only)
  profiles/<tag>_pmc_summary.json    per-kernel, per-launch averages of every PMC counter collected
                                     (separate --pmc passes), plus HBM traffic per launch
  profiles/traffic_latest.json       {kernel: HBM bytes per launch} read by bench.py ("traffic"); fp64 runs write
                                     traffic_latest_f64.json

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE
come from separate passes, are in KiB, and FETCH_SIZE under-reports by a pattern-dependent factor on
gfx950 (x2 for 16 B/lane streams; other widths must be calibrated on a known byte count in the same
access pattern).  Calibration kernel: finalize_kernel, whose traffic is known exactly: it reads
u (N) + x0 (4) + 5 per-problem scalars and writes u (N) + predicted (4N) + 5 scalars, all as 4- or
8-byte-per-lane coalesced accesses, the same pattern every other kernel here uses.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    for k in ("fused_sqp_dyn_kernel", "fused_sqp_kernel", "compact_active_kernel", "qp_ls_kernel", "linearize_kernel", "finalize_kernel", "prepare_kernel", "sim_kernel",
              "rk4_kernel", "dynamics_kernel"):
        if k in name:
            return k
    return None


def read_counters(path):
    """{kernel: {counter: [values per dispatch]}}; values of one dispatch are summed over rows."""
    per = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))
    with open(path) as fh:
        for row in csv.DictReader(fh):
            k = short(row["Kernel_Name"])
            if k is None:
                continue
            per[k][row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
    return {k: {c: list(v.values()) for c, v in cs.items()} for k, cs in per.items()}


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    tag = args[0]
    dtype = args[1] if len(args) > 1 else "f32"
    batch = int(args[2]) if len(args) > 2 else 262144
    N = 40
    # --nx=6: the 6-state model's workload (tools/prof_workload.sh ... double): finalize reads x0 [6] and writes predicted [N][6];
    # its traffic file is profiles/traffic_latest_double_<dtype>.json (bench.py variants.double_pendulum reads it)
    nx = next((int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--nx=")), 4)
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    out_dir = os.path.join(ROOT, "profiles")
    os.makedirs(out_dir, exist_ok=True)

    # kernel stats
    stats = {}
    def newest(pattern):
        """The most recent file matching the pattern (a tag profiled twice leaves both runs' files behind)."""
        found = sorted(glob.glob(pattern), key=os.path.getmtime)
        return found[-1:]

    for path in newest(os.path.join(src, "kt", "*", "*_kernel_stats.csv")):
        with open(path) as fh, open(os.path.join(out_dir, tag + "_kernel_stats.csv"), "w") as out:
            rd = csv.reader(fh)
            wr = csv.writer(out)
            header = next(rd)
            wr.writerow(header)
            for row in rd:
                k = short(row[0])
                if k:
                    wr.writerow(row)
                    stats[k] = {"calls": int(row[1]), "avg_ns": float(row[3]), "pct": float(row[4])}
    # counters
    counters = defaultdict(dict)
    for sub in ("fetch", "write", "sq", "sq2", "mix1", "mix2", "lds"):
        for path in newest(os.path.join(src, sub, "*", "*_counter_collection.csv")):
            for k, cs in read_counters(path).items():
                for c, vals in cs.items():
                    counters[k][c] = sum(vals) / len(vals)
    esz = 4 if dtype == "f32" else 8
    known_read = (N + 4) * esz * batch + 5 * 4 * batch      # finalize: u, x0 (R) + status/iters/ls (int) + f, cn
    known_read = ((N + nx + 2) * esz + 3 * 4) * batch
    known_write = ((N + nx * N + 2) * esz + 3 * 4) * batch
    summary = {"tag": tag, "dtype": dtype, "batch": batch, "kernel_trace": stats, "pmc_per_launch": counters,
               "units": "FETCH_SIZE/WRITE_SIZE in KiB as reported by rocprofv3; *_bytes fields are corrected bytes"}
    traffic = {}
    fin = counters.get("finalize_kernel", {})
    cal_r = cal_w = None
    if "FETCH_SIZE" in fin and fin["FETCH_SIZE"] > 0:
        cal_r = known_read / (fin["FETCH_SIZE"] * 1024.0)
    if "WRITE_SIZE" in fin and fin["WRITE_SIZE"] > 0:
        cal_w = known_write / (fin["WRITE_SIZE"] * 1024.0)
    summary["calibration"] = {
        "kernel": "finalize_kernel", "known_read_bytes": known_read, "known_write_bytes": known_write,
        "fetch_factor": cal_r, "write_factor": cal_w,
        "note": "factor = known bytes / (counter KiB * 1024); applied to every kernel's counters"}
    for k, cs in counters.items():
        if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs and cal_r and cal_w:
            rb = cs["FETCH_SIZE"] * 1024.0 * cal_r
            wb = cs["WRITE_SIZE"] * 1024.0 * cal_w
            traffic[k] = rb + wb
            t = stats.get(k, {}).get("avg_ns")
            summary.setdefault("hbm", {})[k] = {
                "read_bytes": rb, "write_bytes": wb, "bytes_per_problem": (rb + wb) / batch,
                "GBps": (rb + wb) / t if t else None}
    for k, cs in counters.items():
        if "SQ_ACTIVE_INST_VALU" in cs and "SQ_WAVE_CYCLES" in cs and cs["SQ_WAVE_CYCLES"] > 0:
            summary.setdefault("derived", {})[k] = {
                "valu_active_over_wave_cycles": cs["SQ_ACTIVE_INST_VALU"] / cs["SQ_WAVE_CYCLES"],
                "wait_inst_any_over_wave_cycles": cs.get("SQ_WAIT_INST_ANY", 0) / cs["SQ_WAVE_CYCLES"],
                "wait_any_over_wave_cycles": cs.get("SQ_WAIT_ANY", 0) / cs["SQ_WAVE_CYCLES"],
                "valu_insts_per_wave": cs.get("SQ_INSTS_VALU", 0) / max(cs.get("SQ_WAVES", 1), 1),
            }
    for k, cs in counters.items():   # LDS pass of prof.sh: conflict cycles against all LDS-array cycles, per wave
        if "SQ_LDS_IDX_ACTIVE" in cs and cs.get("SQ_WAVES", 0) > 0:
            w = cs["SQ_WAVES"]
            summary.setdefault("lds", {})[k] = {
                "lds_instructions_per_wave": cs.get("SQ_INSTS_LDS", 0) / w,
                "lds_array_cycles_per_wave": cs["SQ_LDS_IDX_ACTIVE"] / w,
                "bank_conflict_cycles_per_wave": cs.get("SQ_LDS_BANK_CONFLICT", 0) / w,
                "addr_conflict_cycles_per_wave": cs.get("SQ_LDS_ADDR_CONFLICT", 0) / w,
                "bank_conflict_share_of_lds_cycles": cs.get("SQ_LDS_BANK_CONFLICT", 0) / cs["SQ_LDS_IDX_ACTIVE"] if cs["SQ_LDS_IDX_ACTIVE"] else None,
                "wait_any_over_wave_cycles": cs.get("SQ_WAIT_ANY", 0) / cs["SQ_WAVE_CYCLES"] if cs.get("SQ_WAVE_CYCLES") else None,
                "wait_inst_lds_over_wave_cycles": cs.get("SQ_WAIT_INST_LDS", 0) / cs["SQ_WAVE_CYCLES"] if cs.get("SQ_WAVE_CYCLES") else None}
    # dynamic vector-instruction mix per wave (the mix1 / mix2 passes of prof.sh), by issue-cost class
    mix = {}
    for k, cs in counters.items():
        waves = cs.get("SQ_WAVES", 0)
        if waves <= 0 or "SQ_INSTS_VALU" not in cs or "SQ_INSTS_VALU_FMA_F32" not in cs:
            continue   # the mix passes (prof.sh mix1 / mix2) were not run for this workload
        g = lambda name: cs.get(name, 0.0) / waves   # noqa: E731
        m = {"valu": g("SQ_INSTS_VALU"),
             "f32_arith": g("SQ_INSTS_VALU_ADD_F32") + g("SQ_INSTS_VALU_MUL_F32") + g("SQ_INSTS_VALU_FMA_F32"),
             "f32_trans": g("SQ_INSTS_VALU_TRANS_F32"),
             "f64_arith": g("SQ_INSTS_VALU_ADD_F64") + g("SQ_INSTS_VALU_MUL_F64") + g("SQ_INSTS_VALU_FMA_F64"),
             "f64_trans": g("SQ_INSTS_VALU_TRANS_F64"),
             "int32": g("SQ_INSTS_VALU_INT32"), "int64": g("SQ_INSTS_VALU_INT64"), "cvt": g("SQ_INSTS_VALU_CVT"),
             "salu": g("SQ_INSTS_SALU"), "lds": g("SQ_INSTS_LDS")}
        m["other_32bit"] = max(0.0, m["valu"] - m["f32_arith"] - m["f32_trans"] - m["f64_arith"] - m["f64_trans"])
        mix[k] = m
    if mix:
        summary["instruction_mix_per_wave"] = mix
    with open(os.path.join(out_dir, tag + "_pmc_summary.json"), "w") as fh:
        json.dump(summary, fh, indent=1, sort_keys=True)
    if traffic and "--no-traffic" not in sys.argv:   # the headline workload only: bench.py reads this file
        tname = "traffic_latest.json" if dtype == "f32" else "traffic_latest_%s.json" % dtype
        if nx != 4:
            tname = "traffic_latest_double_%s.json" % dtype
        with open(os.path.join(out_dir, tname), "w") as fh:
            json.dump({"tag": tag, "dtype": dtype, "batch": batch, "per_launch_bytes": traffic,
                       "instruction_mix_per_wave": mix,
                       "source": "profiles/%s_pmc_summary.json" % tag}, fh, indent=1, sort_keys=True)
    print(json.dumps(summary, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
