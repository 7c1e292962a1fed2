#!/bin/bash
# A/B of variant builds (tools/build_variant.sh) over the configurations of BASELINE.json:
#   tools/ab_r06.sh VARIANT...   ("prod" = the production library)  -> one table row per configuration
cfgs=("--config mmimdb --batch 128" "--config ntu --batch 8" "--config ntu --batch 64" "--config ego --batch 6" "--config ego --batch 48"
      "--config ntu --batch 64 --stage found" "--config mmimdb --batch 128 --stage found" "--tier R")
for cfg in "${cfgs[@]}"; do
  row=$(printf '%-44s' "$cfg")
  for v in "$@"; do
    lib="X=1"; [ "$v" != prod ] && lib="BMNAS_LIB=bm-nas_amd/bmnas/variants/libbmnas_$v.so"
    env $lib python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-full-step --no-roofline $cfg > /tmp/ab.json 2> /tmp/ab.err
    ms=$(python -c "import json; print('%.4f' % json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1])['ms_per_step'])" 2>/dev/null || echo FAIL)
    row="$row $v=$ms"
  done
  echo "$row"
done
