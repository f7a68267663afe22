#!/bin/bash
# Round-4 measurement suite, in parts so that each fits one GPU call (run on the GPU box from the repo root):
#   r04_measure.sh A   bench.py default run (200 steps) + rocprofv3 passes of the headline workload in fp32 and fp64
#                      (incl. the LDS counter pass) + the per-problem-parameters workload in both dtypes
#   r04_measure.sh B   closed-loop soaks (1000 ticks x 262144, both dtypes), host path, batch scaling, N = 160 resolved
#   r04_measure.sh C   all eight shards of BASELINE configs[3] (--as-rank r --of 8, fp64, 512 parity lanes each)
#   r04_measure.sh D   parity sweep (GPU fp64 vs the CPU check, eighteen configurations, 1.6 M problems)
#   r04_measure.sh E <n>  bench.py default run only (box-to-box spread: one call per box)
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
PART=${1:-A}
O=gpurun_out/r04m
mkdir -p $O
case $PART in
A)
  python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
  ./tools/prof.sh r04 > $O/prof_f32.log 2>&1; echo "prof f32 rc=$?"
  ./tools/prof.sh r04_f64 --dtype f64 > $O/prof_f64.log 2>&1; echo "prof f64 rc=$?"
  ./tools/prof_workload.sh r04_per_problem_f32 per_problem > $O/prof_pp32.log 2>&1; echo "prof pp f32 rc=$?"
  ./tools/prof_workload.sh r04_per_problem_f64 per_problem --dtype f64 > $O/prof_pp64.log 2>&1; echo "prof pp f64 rc=$?"
  ./tools/prof_workload.sh r04_closed_loop closed_loop > $O/prof_cl.log 2>&1; echo "prof closed loop rc=$?"
  python tools/summarize_prof.py r04 f32 262144 > /dev/null; python tools/summarize_prof.py r04_f64 f64 262144 > /dev/null
  python tools/summarize_prof.py r04_per_problem_f32 f32 262144 --no-traffic > /dev/null
  python tools/summarize_prof.py r04_per_problem_f64 f64 262144 --no-traffic > /dev/null
  python tools/summarize_prof.py r04_closed_loop f32 262144 --no-traffic > /dev/null
  mkdir -p $O/profiles; cp profiles/r04* profiles/traffic_latest* $O/profiles/
  ;;
B)
  python tools/soak.py --dtype f32 --ticks 1000 --out $O/soak_f32.json > $O/soak_f32.log 2>&1; echo "soak f32 rc=$?"
  python tools/soak.py --dtype f64 --ticks 1000 --out $O/soak_f64.json > $O/soak_f64.log 2>&1; echo "soak f64 rc=$?"
  python tools/host_path.py $O/host_path.json > $O/host_path.log 2>&1; echo "host path rc=$?"
  python tools/batch_scaling.py $O/batch_scaling.json > $O/batch_scaling.log 2>&1; echo "batch scaling rc=$?"
  python tools/long_horizon_resolved.py > $O/n160.log 2>&1; echo "n160 rc=$?"
  python tools/kernel_clock.py --seconds 3 --out $O/kernel_clock.json > $O/kernel_clock.log 2>&1; echo "clock rc=$?"
  ;;
C)
  for r in 0 1 2 3 4 5 6 7; do
    python bench.py --as-rank $r --of 8 --dtype f64 --steps 3 --warmup 1 --parity-lanes 512 > $O/shard_$r.json 2> $O/shard_$r.err; echo "shard $r rc=$?"
  done
  ;;
D)
  python tools/parity_sweep.py $O/parity_sweep.json > $O/parity_sweep.log 2>&1; echo "parity sweep rc=$?"
  ;;
E)
  python bench.py > $O/bench_box_${2:-x}.json 2> $O/bench_box_${2:-x}.err; echo "bench rc=$?"
  ;;
esac
