#!/bin/bash
# usage: prof_workload.sh <tag> <double|closed_loop|per_problem> [run_workload.py flags]   kernel trace + the SQ counters of a secondary workload
set -euo pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
cd /tmp
O="$R/gpurun_out/prof_$TAG"
mkdir -p "$O"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt" -- python3 "$R/tools/run_workload.py" "$@" --steps 10 > "$O/kt.log" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$O/sq" -- python3 "$R/tools/run_workload.py" "$@" --steps 3 > "$O/sq.log" 2>&1
tail -1 "$O/kt.log"
