#!/bin/bash
# A/B of variant builds (tools/build_variant.sh) over configurations of BASELINE.json, alternating runs:
#   tools/ab_r06.sh "CFG ARGS;CFG ARGS;..." VARIANT...   ("prod" = the production library)  -> one table row per configuration
IFS=';' read -ra cfgs <<< "$1"; shift
for cfg in "${cfgs[@]}"; do
  row=$(printf '%-44s' "$cfg")
  for rep in 1 2; do
  for v in "$@"; do
    lib="X=1"; [ "$v" != prod ] && lib="BMNAS_LIB=bm-nas_amd/bmnas/variants/libbmnas_$v.so"
    env $lib python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-full-step --no-roofline $cfg > /tmp/ab.json 2> /tmp/ab.err
    ms=$(python -c "import json; print('%.4f' % json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1])['ms_per_step'])" 2>/dev/null || echo FAIL)
    row="$row $v=$ms"
  done
  done
  echo "$row"
done
