#!/bin/bash
# usage: prof_workload.sh <tag> <double|closed_loop|per_problem> [run_workload.py flags]   kernel trace + the SQ counters (issue,
# waits, instruction mix, LDS) of a secondary workload; summarise with tools/summarize_prof.py <tag> <dtype> <batch> --no-traffic
set -euo pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
cd /tmp
O="$R/gpurun_out/prof_$TAG"
mkdir -p "$O"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt" -- python3 "$R/tools/run_workload.py" "$@" --steps 20 --preheat 1.0 > "$O/kt.log" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$O/sq" -- python3 "$R/tools/run_workload.py" "$@" --steps 3 > "$O/sq.log" 2>&1
# HBM traffic: FETCH_SIZE and WRITE_SIZE in passes of their own (MI355X_MICROARCH.md, HBM / rocprofv3 section)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/fetch" -- python3 "$R/tools/run_workload.py" "$@" --steps 3 > "$O/fetch.log" 2>&1 || echo "fetch pass failed (see fetch.log)"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/write" -- python3 "$R/tools/run_workload.py" "$@" --steps 3 > "$O/write.log" 2>&1 || echo "write pass failed (see write.log)"
if [ "${PROF_MIX:-1}" = 1 ]; then
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 --output-format csv -d "$O/mix1" -- python3 "$R/tools/run_workload.py" "$@" --steps 3 > "$O/mix1.log" 2>&1 || echo "mix1 pass failed (see mix1.log)"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d "$O/mix2" -- python3 "$R/tools/run_workload.py" "$@" --steps 3 > "$O/mix2.log" 2>&1 || echo "mix2 pass failed (see mix2.log)"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INST_CYCLES_VMEM --output-format csv -d "$O/sq2" -- python3 "$R/tools/run_workload.py" "$@" --steps 3 > "$O/sq2.log" 2>&1 || echo "sq2 pass failed"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS --output-format csv -d "$O/lds" -- python3 "$R/tools/run_workload.py" "$@" --steps 3 > "$O/lds.log" 2>&1 || echo "lds pass failed (see lds.log)"
fi
tail -1 "$O/kt.log"
