#!/bin/bash
# Round-6 measurement suite, in parts so that each fits one GPU call (run on the GPU box from the repo root):
#   r06_measure.sh S <n>   the driver's command (`bench.py --gpus 1 --steps 20 --warmup 5`) on this lease -> compact line +
#                          bench_detail.json, plus the 200-step run next to it
#   r06_measure.sh D       BASELINE configs[4] (double pendulum): rocprofv3 kernel trace + SQ counters + FETCH/WRITE, both dtypes
#   r06_measure.sh H       the headline workload: rocprofv3 passes in fp32 and fp64 (tools/prof.sh)
#   r06_measure.sh P       parity sweep (GPU fp64 vs the CPU check) incl. the long horizons
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
PART=${1:-S}
O=gpurun_out/r06m
mkdir -p $O
case $PART in
S)
  python bench.py --gpus 1 --steps 20 --warmup 5 --detail $O/bench_detail_20_${2:-x}.json > $O/bench_20_${2:-x}.json 2> $O/bench_20_${2:-x}.err; echo "bench 20 rc=$?"
  python bench.py --gpus 1 --steps 200 --warmup 5 --no-variants --no-cpu-baseline --no-clock --detail $O/bench_detail_200_${2:-x}.json > $O/bench_200_${2:-x}.json 2> $O/bench_200_${2:-x}.err; echo "bench 200 rc=$?"
  ;;
D)
  ./tools/prof_workload.sh r06_double_f32 double --dtype f32 > $O/prof_d32.log 2>&1; echo "prof double f32 rc=$?"
  ./tools/prof_workload.sh r06_double_f64 double --dtype f64 > $O/prof_d64.log 2>&1; echo "prof double f64 rc=$?"
  python tools/summarize_prof.py r06_double_f32 f32 65536 --nx=6 > /dev/null
  python tools/summarize_prof.py r06_double_f64 f64 65536 --nx=6 > /dev/null
  mkdir -p $O/profiles; cp profiles/r06_double* profiles/traffic_latest_double* $O/profiles/
  ;;
H)
  ./tools/prof.sh r06 > $O/prof_f32.log 2>&1; echo "prof f32 rc=$?"
  ./tools/prof.sh r06_f64 --dtype f64 > $O/prof_f64.log 2>&1; echo "prof f64 rc=$?"
  python tools/summarize_prof.py r06 f32 262144 > /dev/null; python tools/summarize_prof.py r06_f64 f64 262144 > /dev/null
  mkdir -p $O/profiles; cp profiles/r06_kernel* profiles/r06_pmc* profiles/r06_f64* profiles/traffic_latest.json profiles/traffic_latest_f64.json $O/profiles/
  ;;
P)
  python tools/parity_sweep.py $O/parity_sweep.json > $O/parity_sweep.log 2>&1; echo "parity sweep rc=$?"
  ;;
esac
