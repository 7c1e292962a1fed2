#!/bin/bash
# Collect the judged artefacts on the MI355X box: rocprofv3 kernel stats (eager + graph), the two PMC
# passes for HBM traffic, and the bench line.  Usage (via gpurun): bash tools/collect_profiles.sh <tag>
set -u
TAG=${1:-r01}
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
B="--no-cpu-baseline --no-roofline --no-full-step"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_eager -- python3 bench.py --mode eager --steps 50 --warmup 5 $B > $OUT/eager.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_graph -- python3 bench.py --steps 50 --warmup 5 $B > $OUT/graph.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --mode eager --steps 6 --warmup 2 $B > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --mode eager --steps 6 --warmup 2 $B > $OUT/pmc_write.log 2>&1
for k in eager graph; do
  f=$(find $OUT/stats_$k -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $OUT/${TAG}_kernel_stats_$k.csv
done
ff=$(find $OUT/pmc_fetch -name '*counter_collection.csv' | head -1)
fw=$(find $OUT/pmc_write -name '*counter_collection.csv' | head -1)
[ -n "$ff" ] && [ -n "$fw" ] && python3 tools/traffic_from_pmc.py "$ff" "$fw" $OUT/${TAG}_traffic.json > $OUT/traffic.log 2>&1
cp $OUT/${TAG}_traffic.json profiles/r01_traffic.json 2>/dev/null
timeout 400 python3 bench.py 2> $OUT/bench.log | tail -1 > $OUT/${TAG}_bench.json
# keep the merge-back small: drop the raw traces
rm -rf $OUT/stats_eager $OUT/stats_graph $OUT/pmc_fetch $OUT/pmc_write
ls -la $OUT; tail -3 $OUT/traffic.log; python3 tools/ktable.py $OUT/${TAG}_bench.json | head -30
