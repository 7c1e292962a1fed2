#!/bin/bash
# every launch-family switch against the NTU / Ego configurations (is a default still the faster branch?): ms per step
cfgs=("--config ntu --batch 8" "--config ntu --batch 64" "--config ego --batch 6" "--config ego --batch 48")
for sw in X=1 BMNAS_FUSE_NEXT_PAIR=0 BMNAS_FUSE_BN_TAIL=0 BMNAS_FUSE_INNER_SUM=0 BMNAS_FUSE_MIX_GEMM=0 BMNAS_FUSE_MIX_EPILOGUE=0 BMNAS_FUSE_BWD_PAIR=0 BMNAS_FUSE_ATTN_GEMM=0 BMNAS_FUSE_BN_APPLY=0 BMNAS_FUSE_PROLOGUE_PAIR=0 BMNAS_FUSE_HEAD=0 X=2; do
  row=$(printf '%-30s' "$sw")
  for cfg in "${cfgs[@]}"; do
    env $sw python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-full-step --no-roofline $cfg > /tmp/sw.json 2>/tmp/sw.err
    ms=$(python -c "import json; print('%.4f' % json.loads(open('/tmp/sw.json').read().strip().splitlines()[-1])['ms_per_step'])" 2>/dev/null || echo FAIL)
    row="$row $ms"
  done
  echo "$row"
done
