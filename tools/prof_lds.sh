#!/bin/bash
# usage: prof_lds.sh <tag> [extra bench.py flags ...]   (GPU box) -- only the kernel trace and the LDS counter pass of
# prof.sh, for A/Bs of the LDS layout (CPMPC_LIB selects the build variant); summarise with tools/summarize_prof.py --no-traffic
set -euo pipefail
TAG=${1:?tag}
shift || true
EXTRA=("$@")
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
cd /tmp
O="$R/gpurun_out/prof_$TAG"
mkdir -p "$O"
COMMON=(--no-cpu-baseline --no-variants --no-fp64 --no-clock "${EXTRA[@]}")
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt" -- python3 "$R/bench.py" --steps 10 --warmup 2 "${COMMON[@]}" > "$O/kt.log" 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS --output-format csv -d "$O/lds" -- python3 "$R/bench.py" --steps 3 --warmup 1 "${COMMON[@]}" > "$O/lds.log" 2>&1
