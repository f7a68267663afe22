#!/bin/bash
# usage: prof.sh <tag> [extra bench.py flags ...]   (run on the GPU box; e.g. prof.sh r02a_f64 --dtype f64)
# python3 is the program directly after `--` in every rocprofv3 command (no env/bash/launcher hop).
set -euo pipefail
TAG=${1:-r02}
shift || true
EXTRA=("$@")
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
cd /tmp
O="$R/gpurun_out/prof_$TAG"
mkdir -p "$O"
COMMON=(--no-cpu-baseline --no-variants --no-fp64 "${EXTRA[@]}")
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt" -- python3 "$R/bench.py" --steps 10 --warmup 2 "${COMMON[@]}" > "$O/kt.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/fetch" -- python3 "$R/bench.py" --steps 3 --warmup 1 "${COMMON[@]}" > "$O/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/write" -- python3 "$R/bench.py" --steps 3 --warmup 1 "${COMMON[@]}" > "$O/write.log" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$O/sq" -- python3 "$R/bench.py" --steps 3 --warmup 1 "${COMMON[@]}" > "$O/sq.log" 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INST_CYCLES_VMEM --output-format csv -d "$O/sq2" -- python3 "$R/bench.py" --steps 3 --warmup 1 "${COMMON[@]}" > "$O/sq2.log" 2>&1
cd "$O"
find . -name "*.csv" | sort | sed -n '1,30p'
du -sh .
