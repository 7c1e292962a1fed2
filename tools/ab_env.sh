#!/bin/bash
# A/B on one box: bash tools/ab_env.sh VAR [reps]  -> bench ms/step with VAR=1 and VAR=0, alternating
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
V=$1; R=${2:-2}
for i in $(seq $R); do for x in 1 0; do
  env $V=$x timeout 200 python bench.py --no-cpu-baseline --no-roofline --no-full-step --steps 300 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'): print('$V=$x ms_per_step', json.loads(l)['ms_per_step'])"
done; done
