#!/bin/bash
# The GPU parity suite under every kernel-selection switch that is left: the merged / pipelined / lazy-LayerNorm
# launches are the default, the switches force the paths that shapes outside them take (larger batches, found nets,
# heads that cannot be fused, more than eight NodeMixedOps ...).
# EXCLUDED from the per-switch runs: the subprocess-driven tests (-k "driver or two_ranks or bench_ or found_stage or
# outer_loop": search drivers, two-rank runs, bench children, the outer-loop goldens) — half of the suite's time.  They
# run in full under the two configurations that change the LOOP's own code path: BMNAS_DEFAULT=1 (hipGraph steps,
# GraphedForward in run() / evaluate(), the all-ranks-agree capture fallback) and BMNAS_HIP_GRAPH=0 (the eager loop,
# reducer.plan() on the eager path).
# Usage (via gpurun; one call may run 20 minutes — pass a subset):  bash tools/test_matrix.sh "TOGGLE=0 TOGGLE=0 ..."
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ALL="BMNAS_DEFAULT=1 BMNAS_HIP_GRAPH=0 BMNAS_FOUND_FUSE_TAIL=0 BMNAS_FOUND_THRU=0 BMNAS_LAZY_LN=0 BMNAS_WRITE_ONCE=0 BMNAS_CONV_PIPE=0 BMNAS_FUSE_ATTN_GEMM=0 BMNAS_FUSE_PROLOGUE=0 BMNAS_FUSE_EPILOGUE=0 BMNAS_FUSE_BN_FINALIZE=0 BMNAS_FUSE_HEAD=0 BMNAS_FUSE_PROLOGUE_PAIR=0 BMNAS_FUSE_BN_APPLY=0 BMNAS_FUSE_BWD_PAIR=0 BMNAS_FUSE_BN_TAIL=0 BMNAS_FUSE_INNER_SUM=0 BMNAS_FUSE_LN_BWD=0 BMNAS_FUSE_NEXT_PAIR=0 BMNAS_FUSE_MIX_EPILOGUE=0 BMNAS_FUSE_MIX_GEMM=0"
for e in ${1:-$ALL}; do
  echo "== $e"
  case $e in
    BMNAS_DEFAULT=1|BMNAS_HIP_GRAPH=0) sel="" ;;          # the whole suite, subprocess-driven tests included
    *) sel="not driver and not two_ranks and not bench_ and not found_stage and not outer_loop" ;;
  esac
  env $e timeout 1100 python -m pytest tests -m gpu -q ${sel:+-k "$sel"} 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed" | tail -4
done
