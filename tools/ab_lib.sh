#!/bin/bash
# same-box A/B of two builds: bash tools/ab_lib.sh <other.so> [reps]
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=$(realpath $1); R=${2:-3}
for i in $(seq $R); do
  for tag in new old; do
    if [ $tag = old ]; then export BMNAS_LIB=$O; else unset BMNAS_LIB; fi
    timeout 200 python bench.py --no-cpu-baseline --no-roofline --no-full-step --steps 300 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'): print('$tag', json.loads(l)['ms_per_step'])"
  done
done
