#!/bin/bash
# same-session A/B of library builds on BASELINE configs[4] (cart + double pendulum, B = 65536, N = 40, 5 iterations, cold
# start): tools/run_workload.py double for every build in VARIANTS (default = the product; others tools/_build/lib_<name>)
# x PIPELINES x DTYPES, REPS times round-robin.  One JSON line per run on stdout.
cd "$(dirname "$0")/.."
for rep in $(seq 1 ${REPS:-2}); do
for v in ${VARIANTS:-default}; do
  if [ "$v" = default ]; then unset CPMPC_LIB; else export CPMPC_LIB=$PWD/tools/_build/lib_$v/libcpmpc.so; fi
  for dt in ${DTYPES:-f64}; do
  for pipe in ${PIPELINES:-fused split}; do
    python3 tools/run_workload.py double --dtype $dt --pipeline $pipe --steps ${STEPS:-20} | sed "s/^{/{\"build\": \"$v\", \"asked\": \"$pipe\", /"
  done; done
done; done
