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
COMMON=(--no-cpu-baseline --no-variants --no-fp64 --no-clock --preheat-seconds 0 "${EXTRA[@]}")
# the kernel trace runs the driver's command line (20 steps after 5 warm-up steps, behind bench.py's 2 s pre-heat: the trace then
# averages ~1 000 launches of each kernel on a device in the state the bench line is measured in); the counter passes below
# serialise kernels and take 3 steps without the pre-heat
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt" -- python3 "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-variants --no-fp64 --no-clock "${EXTRA[@]}" > "$O/kt.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/fetch" -- python3 "$R/bench.py" --steps 3 --warmup 1 "${COMMON[@]}" > "$O/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/write" -- python3 "$R/bench.py" --steps 3 --warmup 1 "${COMMON[@]}" > "$O/write.log" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$O/sq" -- python3 "$R/bench.py" --steps 3 --warmup 1 "${COMMON[@]}" > "$O/sq.log" 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INST_CYCLES_VMEM --output-format csv -d "$O/sq2" -- python3 "$R/bench.py" --steps 3 --warmup 1 "${COMMON[@]}" > "$O/sq2.log" 2>&1
# dynamic instruction mix of the vector stream (what the issue-ceiling accounting of DESIGN.md section 6 prices); these two
# passes are optional: a counter this rocprofv3 does not know must not cost the passes above
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 --output-format csv -d "$O/mix1" -- python3 "$R/bench.py" --steps 3 --warmup 1 "${COMMON[@]}" > "$O/mix1.log" 2>&1 || echo "mix1 pass failed (see mix1.log)"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d "$O/mix2" -- python3 "$R/bench.py" --steps 3 --warmup 1 "${COMMON[@]}" > "$O/mix2.log" 2>&1 || echo "mix2 pass failed (see mix2.log)"
# LDS: cycles the LDS array worked for this kernel and the extra cycles bank conflicts cost
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS --output-format csv -d "$O/lds" -- python3 "$R/bench.py" --steps 3 --warmup 1 "${COMMON[@]}" > "$O/lds.log" 2>&1 || echo "lds pass failed (see lds.log)"
cd "$O"
find . -name "*.csv" | sort | sed -n '1,30p'
du -sh .
