"""The settled closed-loop tick as rocprofv3 sees it: from the kernel trace of
    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/run_workload.py closed_loop --settle 300 --steps 20 [--dtype f64]
take the launches of the LAST `ticks` ticks (a tick ends with sim_kernel) and print launches per tick and mean duration
per kernel.   python tools/settled_trace.py DIR [ticks] [out.json]"""
import csv
import glob
import json
import os
import sys

d = sys.argv[1]
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 20
files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
assert files, "no kernel_trace.csv under " + d
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
short = lambda n: next((k for k in ("fused_sqp_dyn_kernel", "fused_sqp_kernel", "compact_active_kernel", "prepare_kernel", "finalize_kernel",
                                     "sim_kernel", "qp_ls_kernel", "linearize_kernel") if k in n), "other")
ends = [i for i, r in enumerate(rows) if short(r[2]) == "sim_kernel"]
assert len(ends) > ticks, "fewer ticks in the trace than asked for"
lo = ends[-ticks - 1] + 1
sel = rows[lo:ends[-1] + 1]
per = {}
for s, e, n in sel:
    k = short(n)
    per.setdefault(k, []).append((e - s) / 1e3)
span = (sel[-1][1] - sel[0][0]) / 1e6 / ticks
out = {"ticks": ticks, "ms_per_tick_first_start_to_last_end": span,
       "kernels": {k: {"launches_per_tick": len(v) / ticks, "mean_us": sum(v) / len(v), "us_per_tick": sum(v) / ticks} for k, v in per.items()}}
out["kernel_us_per_tick_total"] = sum(v["us_per_tick"] for v in out["kernels"].values())
print(json.dumps(out, indent=1))
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)
