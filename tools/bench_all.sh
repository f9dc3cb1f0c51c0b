#!/bin/bash
# The benchmark's step through match_many for every workload (one GPU): step time beside kernel time.
#   bash tools/bench_all.sh > gpurun_out/bench_all.txt
set -e
for w in cfg1 cfg2 cfg3 cfg3_hw cfg3_hw2 cfg4 cfg4_ce cfg5; do
    timeout -k 10 200 python3 bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline --no-stress 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline())
k=r['step_kernels']
c=k['cost_kernel_ms_per_step'] or 0.0
print('%-9s step %7.1f us (gc paused %7.1f)  kernels %7.1f us (cost %6.1f + assignment %6.1f)  flat path %7.1f us  %9.0f images/s' % ('$w', 1e3*r['ms_per_step'], 1e3*r['python_gc_paused']['ms_per_step'], 1e3*(c+k['assignment_kernel_ms_per_step']), 1e3*c, 1e3*k['assignment_kernel_ms_per_step'], 1e3*r['flat_path']['ms_per_step'], r['value']) + '   per-layer calls %7.1f us, behind attach() %7.1f us' % (1e3*r['per_layer_calls']['ms_per_step'], 1e3*r['per_layer_calls_prefetched']['ms_per_step']))"
done
