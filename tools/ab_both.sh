#!/bin/bash
# same-session A/B of build variants in BOTH dtypes: bench.py (fp32 headline + fp64 record), 60 steps each, twice
# round-robin.  VARIANTS="default name1 name2" with tools/_build/lib_<name>/libcpmpc.so built by build.build_variant().
cd "$(dirname "$0")/.."
for rep in 1 2; do
for v in ${VARIANTS:-default}; do
  if [ "$v" = default ]; then unset CPMPC_LIB; else export CPMPC_LIB=$PWD/tools/_build/lib_$v/libcpmpc.so; fi
  python bench.py --steps 60 --no-variants --no-cpu-baseline --no-clock --detail /tmp/ab_detail.json > /dev/null 2>&1; python -c "
import json,sys
d=json.load(open('/tmp/ab_detail.json')); k=d['roofline']['kernels_ms_per_step']; k6=d['fp64']['roofline']['kernels_ms_per_step']
print('$v', 'f32', round(d['value']/1e6,2), k['fused_sqp_kernel'], 'prep/fin', k['prepare_kernel'], k['finalize_kernel'], 'f64', round(d['fp64']['value']/1e6,2), k6['fused_sqp_kernel'], 'prep/fin', k6['prepare_kernel'], k6['finalize_kernel'])"
done; done
