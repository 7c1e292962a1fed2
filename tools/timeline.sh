#!/bin/bash
# usage (gpurun): bash tools/timeline.sh OUT.txt [bench args ...]   (kernel timeline of one graph replay of bench.py's step)
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
out=${1:-gpurun_out/timeline.txt}; shift
rm -rf gpurun_out/tl
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-full-step "$@" > gpurun_out/tl.log 2>&1
f=$(find gpurun_out/tl -name "*kernel_trace.csv" | head -1)
python3 tools/timeline.py $f > $out
rm -rf gpurun_out/tl
tail -3 $out
