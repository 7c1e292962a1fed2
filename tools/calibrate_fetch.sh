#!/bin/bash
# FETCH_SIZE read-width calibration on the MI355X box (one --pmc pass, kernel trace only).
# Usage (via gpurun): bash tools/calibrate_fetch.sh   -> gpurun_out/fetch_calibration.txt
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
OUT=gpurun_out/calib
rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -- python3 tools/calibrate_fetch.py run > $OUT/run.log 2>&1
f=$(find $OUT -name '*counter_collection.csv' | head -1)
[ -n "$f" ] && python3 tools/calibrate_fetch.py report "$f" gpurun_out/fetch_calibration.txt
rm -rf $OUT
