#!/bin/bash
# Round-3 measurement suite (run on the GPU box from the repo root; outputs under gpurun_out/r03f/ and gpurun_out/prof_*):
#   bench.py default run (200 steps), rocprofv3 passes of the headline workload in fp32 and fp64, of the double pendulum
#   and of the closed loop, the clock of the fused kernel, the instruction-cost micro-benchmark, closed-loop soaks.
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
O=gpurun_out/r03f
mkdir -p $O
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
./tools/prof.sh r03d > $O/prof_f32.log 2>&1; echo "prof f32 rc=$?"
./tools/prof.sh r03d_f64 --dtype f64 > $O/prof_f64.log 2>&1; echo "prof f64 rc=$?"
./tools/prof_workload.sh r03d_closed_loop closed_loop > $O/prof_cl.log 2>&1; echo "prof closed loop rc=$?"
./tools/prof_workload.sh r03d_double_f32 double > $O/prof_double.log 2>&1; echo "prof double rc=$?"
python tools/kernel_clock.py --seconds 3 --out $O/kernel_clock.json > $O/kernel_clock.log 2>&1; echo "clock rc=$?"
./tools/ubench/clock --seconds 1 > $O/clock_ubench.jsonl 2> /dev/null; echo "ubench rc=$?"
python tools/soak.py --dtype f32 --ticks 1000 --out $O/soak_f32.json > $O/soak_f32.log 2>&1; echo "soak f32 rc=$?"
python tools/soak.py --dtype f64 --ticks 1000 --out $O/soak_f64.json > $O/soak_f64.log 2>&1; echo "soak f64 rc=$?"
# fp32 with / without the refinement pass of the terminal multipliers, same session
for v in default norefine32; do
  if [ "$v" = default ]; then unset CPMPC_LIB; else export CPMPC_LIB=$PWD/tools/_build/lib_$v/libcpmpc.so; fi
  python bench.py --steps 100 --no-variants --no-clock --no-fp64 > $O/bench_ab_$v.json 2> /dev/null; echo "bench $v rc=$?"
done
unset CPMPC_LIB
python tools/parity_sweep.py $O/parity_sweep.json > $O/parity_sweep.log 2>&1; echo "parity sweep rc=$?"
