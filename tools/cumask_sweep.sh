#!/bin/bash
# ROC_GLOBAL_CU_MASK against the small-shard configurations (is one XCD's L2 locality worth its 32 CUs?)
run() {
  label=$1; shift
  row=$(printf '%-50s' "$label")
  for cfg in "--config ntu --batch 8" "--config ego --batch 6" "--config ntu --batch 64"; do
    timeout -k 5 120 env "$@" python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-full-step --no-roofline $cfg > /tmp/es.json 2> /tmp/es.err
    ms=$(python -c "import json; d=json.loads(open('/tmp/es.json').read().strip().splitlines()[-1]); print('%.4f' % d['ms_per_step'])" 2>/dev/null || echo FAIL)
    row="$row  $ms"
  done
  echo "$row"
}
run "default" X=1
run "mask 32 low bits" ROC_GLOBAL_CU_MASK=0xffffffff
run "mask 64 low bits" ROC_GLOBAL_CU_MASK=0xffffffffffffffff
run "mask every 8th bit (32 CUs)" ROC_GLOBAL_CU_MASK=0x0101010101010101010101010101010101010101010101010101010101010101
run "mask every 4th bit (64 CUs)" ROC_GLOBAL_CU_MASK=0x1111111111111111111111111111111111111111111111111111111111111111
run "mask 128 low bits" ROC_GLOBAL_CU_MASK=0xffffffffffffffffffffffffffffffff
run "default again" X=1
