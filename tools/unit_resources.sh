#!/bin/bash
# usage: unit_resources.sh <unit> <kernel-name-pattern> [extra hipcc flags]   compile ONE translation unit of the library to an
# object (the product's flags) and list registers / spills / LDS of its kernels (tools/kernel_resources.py)
set -euo pipefail
U=$1; PAT=$2; shift 2
R=$(cd "$(dirname "$0")/.." && pwd)
O=${TMPDIR:-/tmp}/cpmpc_unit_$U.$$.o
SCHED=""
case $U in engine_f32_*) SCHED="-mllvm -amdgpu-sched-strategy=iterative-ilp -mllvm -enable-post-misched=0";; engine_f64_*) SCHED="-mllvm -amdgpu-schedule-relaxed-occupancy=1";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fPIC $SCHED "$@" -c -o "$O" "$R/cart-pole-mpc_amd/csrc/$U.hip" 2>/dev/null
python3 "$R/tools/kernel_resources.py" "$O" "$PAT" | sed 's/_ZN5cpmpc//'
rm -f "$O"
