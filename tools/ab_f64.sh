#!/bin/bash
# same-session A/B of the fp64 arithmetic variants (built with cart-pole-mpc_amd/build.py: build_variant("notable",
# ["-DCPMPC_F64_COEF_TABLE=0"]), "norotate" -DCPMPC_F64_TRIG_ROTATE=0, "nomerge" -DCPMPC_F64_MERGED_RCP=0, "r2math" all three):
# bench.py --dtype f64, 60 steps each, twice round-robin
cd "$(dirname "$0")/.."
for rep in 1 2; do
for v in ${VARIANTS:-default notable norotate nomerge r2math}; do
  if [ "$v" = default ]; then unset CPMPC_LIB; else export CPMPC_LIB=$PWD/tools/_build/lib_$v/libcpmpc.so; fi
  python bench.py --dtype f64 --steps 60 --no-variants --no-cpu-baseline --no-clock --no-fp64 --detail /tmp/ab_detail.json > /dev/null 2>&1; python -c "
import json,sys
d=json.load(open('/tmp/ab_detail.json')); print('$v', round(d['value']/1e6,2), d['roofline']['kernels_ms_per_step'])"
done; done
