#!/bin/bash
# same-session A/B of fp32 build variants: re-plans/s AND the gap to the fp64 CPU check on the benchmark sample
# (bench.py parity_sample), plus a 200-tick swing-up soak per variant.  VARIANTS="default name1 ..." as ab_both.sh.
cd "$(dirname "$0")/.."
for v in ${VARIANTS:-default}; do
  if [ "$v" = default ]; then unset CPMPC_LIB; else export CPMPC_LIB=$PWD/tools/_build/lib_$v/libcpmpc.so; fi
  python bench.py --steps 40 --no-fp64 --no-variants --no-clock --detail /tmp/ab_detail.json > /dev/null 2>&1; python -c "
import json,sys
d=json.load(open('/tmp/ab_detail.json')); p=d['parity_sample']
print('$v', 'f32 M/s', round(d['value']/1e6,2), 'fused ms', d['roofline']['kernels_ms_per_step']['fused_sqp_kernel'], 'median', p.get('max_abs_du_median'), 'p99', p.get('max_abs_du_p99'), 'max', p.get('max_abs_du_max'), 'within1e-2', p.get('fraction_within_1e-2'), 'status_agree', p.get('status_agree'), {k: p[k] for k in p if 'lanes' in k and 'note' not in k})"
  python tools/soak.py --dtype f32 --ticks ${SOAK_TICKS:-200} --out gpurun_out/soak_f32_$v.json > /dev/null 2>&1
  python -c "
import json; d=json.load(open('gpurun_out/soak_f32_$v.json'))
print('$v', 'soak ms/tick', round(d['ms_per_tick'],3), d['status_histogram_total'], 'bad ticks', d['ticks_with_QP_INDEFINITE_MAX_LAMBDA_or_NON_FINITE'], 'final pole err', d['final']['max_abs_pole_angle_error'])"
done
