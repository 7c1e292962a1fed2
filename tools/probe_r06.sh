#!/bin/bash
# round 6: A/B runs of variant builds of the library (tools/build_variant.sh NAME flags...) on the headline step
#   tools/probe_r06.sh [-q] NAME[:ENV=V,...] ...   -> gpurun_out/r06_probe_NAME.json + one table line per run
#   -q: step times only (no rocprofv3 child)
V=$PWD/bm-nas_amd/bmnas/variants
extra=""
if [ "$1" = "-q" ]; then extra="--no-roofline"; shift; fi
for spec in "$@"; do
  name=${spec%%:*}; envs=""
  [ "$spec" != "$name" ] && envs=$(echo "${spec#*:}" | tr ',' ' ')
  lib=${name%%+*}
  env BMNAS_LIB=$V/libbmnas_$lib.so $envs python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-full-step $extra \
      > gpurun_out/r06_probe_$name.json 2> gpurun_out/r06_probe_$name.err
  python - "$name" <<'PY'
import json, sys
n = sys.argv[1]
d = json.loads(open(f'gpurun_out/r06_probe_{n}.json').read().strip().splitlines()[-1])
ks = {k['kernel'].split('<')[0]: k['avg_us'] for k in d.get('roofline_kernels', [])}
print(f"{n:28s} single {d['step_shapes']['single']['ms_per_step_median']:.4f} k4 {d['ms_per_step']:.4f} | " +
      ' '.join(f'{k}={v}' for k, v in ks.items()), flush=True)
PY
done
