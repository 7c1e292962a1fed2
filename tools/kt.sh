#!/bin/bash
# per-kernel table of the headline step under an environment setting:  tools/kt.sh "ENV=1 ..." [bench args]
# (bench.py's own rocprofv3 child trace; prints kernel, launches per step, avg us, frac)
envs="$1"; shift
env $envs python bench.py --steps 50 --warmup 10 --no-full-step --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms_per_step', d['ms_per_step'], d['timed_regions']['ms_per_step'], 'kernels', d.get('roofline_check',{}).get('kernels_per_step'), 'sum_us', d.get('roofline_check',{}).get('sum_kernel_us_per_step'))
for r in d.get('roofline_kernels',[]): print('  %-44s x%d %7.2f us  frac %s' % (r['kernel'][:44], r['launches_per_step'], r['avg_us'], r['frac']))
"
