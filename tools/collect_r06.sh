#!/bin/bash
# Round 6.  Evidence for the BASELINE configs (MI355X box, via gpurun): bench lines with roofline for MM-IMDB b128
# (headline), NTU b8 / b64, Ego b6 / b48 (configs 4/5 per GPU and whole), tier R, rocprofv3 kernel stats of
# the same commands, the two PMC passes for HBM traffic and the K1 batch sweep.
# Usage: bash tools/collect_r06.sh <tag>
TAG=${1:-r06}
PART=${2:-all}        # a | b | all: one gpurun call is limited to 20 minutes
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
  echo "== $n"; python3 tools/ktable.py $OUT/${TAG}_bench_$n.json | head -40
}
if [ $PART != b ]; then
# PMC passes first: bench.py's roofline rows read profiles/<tag>_traffic.json
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --mode eager --steps 6 --warmup 2 $B > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --mode eager --steps 6 --warmup 2 $B > $OUT/pmc_write.log 2>&1
ff=$(find $OUT/pmc_fetch -name '*counter_collection.csv' | head -1)
fw=$(find $OUT/pmc_write -name '*counter_collection.csv' | head -1)
[ -n "$ff" ] && [ -n "$fw" ] && python3 tools/traffic_two_pass.py "$ff" "$fw" $OUT/${TAG}_traffic.json > $OUT/traffic.log 2>&1 && cp $OUT/${TAG}_traffic.json profiles/r06_traffic.json
rm -rf $OUT/pmc_fetch $OUT/pmc_write
# (every line carries cpu_baseline now: the oracle timed on THIS host; tier R has no CPU port
# of the reshape layers in the timed oracle step, so its lines skip it)
run mmimdb_b128 --steps 200
run ntu_b8 --config ntu --batch 8 --steps 200
run ntu_b64 --config ntu --batch 64 --steps 200
run ego_b6 --config ego --batch 6 --steps 200
run ego_b48 --config ego --batch 48 --steps 200
run mmimdb_b128_tierR --tier R --steps 200
fi
if [ $PART != a ]; then
run ntu_b64_tierR --config ntu --batch 64 --tier R --steps 100
run ntu_b8_tierR --config ntu --batch 8 --tier R --steps 100
run ego_b48_tierR --config ego --batch 48 --tier R --steps 100
run mmimdb_b1024 --batch 1024 --steps 100
# row f3: the found-stage training step of a fixed genotype (x != y kernels) + its evaluation forward
timeout 400 python3 bench.py --stage found --steps 200 2> $OUT/found_mm.log | tail -1 > $OUT/${TAG}_bench_found_mmimdb_b128.json
timeout 400 python3 bench.py --stage found --config ntu --batch 64 --steps 200 2> $OUT/found_ntu.log | tail -1 > $OUT/${TAG}_bench_found_ntu_b64.json
# configs 4 / 5 with their GLOBAL batch fixed (one GPU here: the sharded figure equals the one-GPU full-batch figure)
timeout 400 python3 bench.py --scaling strong --config ntu --steps 200 --no-full-step 2> $OUT/strong_ntu.log | tail -1 > $OUT/${TAG}_bench_strong_ntu.json
timeout 400 python3 bench.py --scaling strong --config ego --steps 200 --no-full-step 2> $OUT/strong_ego.log | tail -1 > $OUT/${TAG}_bench_strong_ego.json
# the N > 1 step shapes through a world-size-1 communicator (what one GPU can show of them)
timeout 300 python3 bench.py --dp-selftest --steps 100 --no-full-step --no-roofline --no-cpu-baseline 2> $OUT/dp.log | tail -1 > $OUT/${TAG}_bench_dp_selftest.json
# K1 (the MixedOp kernel) against the HBM roofline over the batch: launch-inclusive rocprofv3 durations
echo "# K1 = mixsum_pair_{fwd,bwd}_k: algorithmic bytes / rocprofv3 duration / 8 TB/s, per-GPU batch sweep (MM-IMDB shapes)" > $OUT/${TAG}_k1_batch_sweep.txt
for b in 32 64 128 256 512 1024; do
  timeout 400 python3 bench.py --batch $b --steps 100 --no-cpu-baseline --no-full-step 2>/dev/null | tail -1 > $OUT/sweep_$b.json
  python3 - $OUT/sweep_$b.json $b >> $OUT/${TAG}_k1_batch_sweep.txt <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
rows = [r for r in d.get('roofline_kernels', []) if r['kernel'].startswith(('mixsum', 'cell_prologue_pair')) and r.get('frac') is not None]
print(f"batch {int(sys.argv[2]):5d}: step {d['ms_per_step']:.4f} ms  " + '  '.join(
    f"{r['kernel'].replace('mixsum_pair_', '').replace('cell_prologue_pair_k', 'fwd_k+prologue')}: {r['avg_us']:.2f} us {r['algorithmic_units_per_launch'] / 1e6:.1f} MB frac {r['frac']:.3f}"
    for r in sorted(rows, key=lambda r: r['kernel'])))
PY
done
cat $OUT/${TAG}_k1_batch_sweep.txt
fi
ls -la $OUT
