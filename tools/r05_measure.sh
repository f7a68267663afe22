#!/bin/bash
# Round-5 measurement suite, in parts so that each fits one GPU call (run on the GPU box from the repo root):
#   r05_measure.sh S <n>   the driver's command (`bench.py --gpus 1 --steps 20 --warmup 5`) on this lease -> one record of
#                          profiles/r05_box_spread.json, plus the 200-step run next to it
#   r05_measure.sh D       BASELINE configs[4] (double pendulum): rocprofv3 kernel trace + SQ counter passes, both dtypes
#   r05_measure.sh H       the headline workload: rocprofv3 passes in fp32 and fp64 (tools/prof.sh)
#   r05_measure.sh B       closed-loop soaks (1000 ticks x 262144, both dtypes)
#   r05_measure.sh P       parity sweep (GPU fp64 vs the CPU check, eighteen configurations, 1.6 M problems)
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
PART=${1:-S}
O=gpurun_out/r05m
mkdir -p $O
case $PART in
S)
  python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20_${2:-x}.json 2> $O/bench_20_${2:-x}.err; echo "bench 20 rc=$?"
  python bench.py --gpus 1 --steps 200 --warmup 5 --no-variants --no-cpu-baseline --no-clock > $O/bench_200_${2:-x}.json 2> $O/bench_200_${2:-x}.err; echo "bench 200 rc=$?"
  ;;
D)
  ./tools/prof_workload.sh r05_double_f32 double --dtype f32 > $O/prof_d32.log 2>&1; echo "prof double f32 rc=$?"
  ./tools/prof_workload.sh r05_double_f64 double --dtype f64 > $O/prof_d64.log 2>&1; echo "prof double f64 rc=$?"
  PROF_MIX=0 ./tools/prof_workload.sh r05_double_f64_split double --dtype f64 --pipeline split > $O/prof_d64s.log 2>&1; echo "prof double f64 split rc=$?"
  python tools/summarize_prof.py r05_double_f32 f32 65536 --no-traffic > /dev/null
  python tools/summarize_prof.py r05_double_f64 f64 65536 --no-traffic > /dev/null
  python tools/summarize_prof.py r05_double_f64_split f64 65536 --no-traffic > /dev/null
  mkdir -p $O/profiles; cp profiles/r05_double* $O/profiles/
  ;;
H)
  ./tools/prof.sh r05 > $O/prof_f32.log 2>&1; echo "prof f32 rc=$?"
  ./tools/prof.sh r05_f64 --dtype f64 > $O/prof_f64.log 2>&1; echo "prof f64 rc=$?"
  python tools/summarize_prof.py r05 f32 262144 > /dev/null; python tools/summarize_prof.py r05_f64 f64 262144 > /dev/null
  mkdir -p $O/profiles; cp profiles/r05_kernel* profiles/r05_pmc* profiles/r05_f64* profiles/traffic_latest* $O/profiles/
  ;;
B)
  python tools/soak.py --dtype f32 --ticks 1000 --out $O/soak_f32.json > $O/soak_f32.log 2>&1; echo "soak f32 rc=$?"
  python tools/soak.py --dtype f64 --ticks 1000 --out $O/soak_f64.json > $O/soak_f64.log 2>&1; echo "soak f64 rc=$?"
  ;;
P)
  python tools/parity_sweep.py $O/parity_sweep.json > $O/parity_sweep.log 2>&1; echo "parity sweep rc=$?"
  ;;
esac
