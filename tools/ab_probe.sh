#!/bin/bash
# bash tools/ab_probe.sh VAR "v1 v2 ..." [bench args]: bench ms/step for each value of VAR on one box
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
V=$1; VALS=$2; shift 2
for x in $VALS; do
  env $V=$x timeout 200 python bench.py --no-cpu-baseline --no-roofline --no-full-step --steps 300 "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'): print('$V=$x ms_per_step', json.loads(l)['ms_per_step'])"
done
