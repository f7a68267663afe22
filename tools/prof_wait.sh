#!/bin/bash
# usage: prof_wait.sh <tag> [extra bench.py flags ...]   (GPU box) -- what a wave of the fused kernel waits for: latency and
# count of LDS, scalar-memory, vector-memory instructions and of instruction fetches (derived metrics LdsLatency,
# SmemLatency, VmemLatency, InstrFetchLatency of rocprofv3, one pass each), beside SQ_WAIT_ANY.
set -uo pipefail
TAG=${1:?tag}
shift || true
EXTRA=("$@")
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
cd /tmp
O="$R/gpurun_out/prof_$TAG"
mkdir -p "$O"
COMMON=(--no-cpu-baseline --no-variants --no-fp64 --no-clock --steps 3 --warmup 1 "${EXTRA[@]}")
i=0
for set in "LdsLatency" "SmemLatency" "VmemLatency" "InstrFetchLatency" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_SMEM SQ_INSTS_LDS SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_SALU" \
           "SQ_WAVES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d "$O/w$i" -- python3 "$R/bench.py" "${COMMON[@]}" > "$O/w$i.log" 2>&1 || echo "pass $i ($set) failed"
done
python3 - "$O" <<'PY'
import csv, glob, sys, json, collections
O = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/w*/*/*_counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "fused_sqp" not in k: continue
        acc[row["Counter_Name"]][row["Dispatch_Id"]].append(float(row["Counter_Value"]))
out = {}
for c, d in acc.items():
    vals = [sum(v) for v in d.values()]
    out[c] = sum(vals) / len(vals)
w = out.get("SQ_WAVES", 1.0)
res = {"per_launch": out, "per_wave": {c: v / w for c, v in out.items() if c.startswith("SQ_") and c != "SQ_WAVES"}}
for lat, cnt in (("LdsLatency", "SQ_INSTS_LDS"), ("SmemLatency", "SQ_INSTS_SMEM"), ("InstrFetchLatency", "SQ_IFETCH")):
    if lat in out and cnt in out:
        res.setdefault("latency_cycles_x_count_per_wave", {})[lat] = out[lat] * out[cnt] / w
print(json.dumps(res, indent=1))
json.dump(res, open(O + "/wait_summary.json", "w"), indent=1)
PY
