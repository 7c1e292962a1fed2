#!/bin/bash
# usage (gpurun): bash tools/timeline.sh [ENV=VAL ...]  -> gpurun_out/timeline.txt
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for kv in "$@"; do export "$kv"; done
rm -rf gpurun_out/tl
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-full-step > gpurun_out/tl.log 2>&1
f=$(find gpurun_out/tl -name "*kernel_trace.csv" | head -1)
python3 tools/timeline.py $f > gpurun_out/timeline.txt
rm -rf gpurun_out/tl
cat gpurun_out/timeline.txt
