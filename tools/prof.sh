#!/bin/bash
# usage: prof.sh <tag>   (run on the GPU box from the repo root)
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-variants > $O/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants > $O/write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/sq -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants > $O/sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $O/sq2 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants > $O/sq2.log 2>&1
cd $O
find . -name "*.csv" | head -30
du -sh .
