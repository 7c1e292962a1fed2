#!/bin/bash
# HIP runtime knobs against the captured step (one line per setting): tools/env_sweep.sh
run() {
  label=$1; shift
  row=$(printf '%-44s' "$label")
  for cfg in "--config mmimdb --batch 128" "--config ntu --batch 8"; do
    env "$@" python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-full-step --no-roofline $cfg > /tmp/es.json 2> /tmp/es.err
    ms=$(python -c "import json; d=json.loads(open('/tmp/es.json').read().strip().splitlines()[-1]); print('%.4f/%.4f' % (d['ms_per_step'], d.get('step_shapes',{}).get('single',{}).get('ms_per_step_median',0)))" 2>/dev/null || echo FAIL)
    row="$row  $ms"
  done
  echo "$row"
}
run "default" X=1
run "HIP_FORCE_DEV_KERNARG=1" HIP_FORCE_DEV_KERNARG=1
run "HIP_FORCE_DEV_KERNARG=0" HIP_FORCE_DEV_KERNARG=0
run "ROC_USE_FGS_KERNARG=0" ROC_USE_FGS_KERNARG=0
run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run "AMD_OPT_FLUSH=0" AMD_OPT_FLUSH=0
run "AMD_OPT_FLUSH=1" AMD_OPT_FLUSH=1
run "DEBUG_HIP_KERNARG_COPY_OPT=0" DEBUG_HIP_KERNARG_COPY_OPT=0
run "GPU_FLUSH_ON_EXECUTION=1" GPU_FLUSH_ON_EXECUTION=1
run "ROC_SYSTEM_SCOPE_SIGNAL=0" ROC_SYSTEM_SCOPE_SIGNAL=0
run "DEBUG_CLR_KERNARG_HDP_FLUSH_WA=0" DEBUG_CLR_KERNARG_HDP_FLUSH_WA=0
run "default again" X=1
