#!/bin/bash
# Evidence for the BASELINE configs (MI355X box, via gpurun): bench lines with roofline for MM-IMDB b128
# (headline), NTU b8, Ego b6 (per-GPU batches of configs 4/5), tier R, plus rocprofv3 kernel stats of the
# same commands.  Usage: bash tools/collect_r02.sh <tag>
TAG=${1:-r02}
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
B="--no-cpu-baseline --no-roofline --no-full-step"
run() {  # name, bench args
  local n=$1; shift
  timeout 400 python3 bench.py "$@" 2> $OUT/$n.log | tail -1 > $OUT/${TAG}_bench_$n.json
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st_$n -- python3 bench.py "$@" --steps 50 --warmup 5 $B > $OUT/st_$n.log 2>&1
  f=$(find $OUT/st_$n -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $OUT/${TAG}_kernel_stats_$n.csv
  rm -rf $OUT/st_$n
  python3 tools/ktable.py $OUT/${TAG}_bench_$n.json | head -45
}
run mmimdb_b128 --steps 200
run ntu_b8 --config ntu --batch 8 --steps 200 --no-cpu-baseline
run ego_b6 --config ego --batch 6 --steps 200 --no-cpu-baseline
run mmimdb_b128_tierR --tier R --steps 200
run ntu_b64_tierR --config ntu --batch 64 --tier R --steps 100
ls -la $OUT
