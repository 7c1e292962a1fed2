#!/bin/bash
# The GPU parity suite under every kernel-selection switch that is left: the merged / pipelined launches are the
# default, the switches force the paths that shapes outside them take (larger batches, found nets, heads that cannot
# be fused, more than eight NodeMixedOps ...).  The subprocess-driven tests (drivers, two-rank runs, bench children)
# are left out: they do not depend on the switches and take half of the suite's time.
# Usage (via gpurun; one call may run 20 minutes — pass a subset):  bash tools/test_matrix.sh "TOGGLE=0 TOGGLE=0 ..."
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ALL="BMNAS_DEFAULT=1 BMNAS_CONV_PIPE=0 BMNAS_FUSE_ATTN_GEMM=0 BMNAS_FUSE_PROLOGUE=0 BMNAS_FUSE_EPILOGUE=0 BMNAS_FUSE_BN_FINALIZE=0 BMNAS_FUSE_HEAD=0 BMNAS_FUSE_PROLOGUE_PAIR=0 BMNAS_FUSE_BN_APPLY=0 BMNAS_FUSE_BWD_PAIR=0 BMNAS_FUSE_BN_TAIL=0 BMNAS_FUSE_INNER_SUM=0 BMNAS_FUSE_LN_BWD=0 BMNAS_FUSE_NEXT_PAIR=0 BMNAS_FUSE_MIX_EPILOGUE=0 BMNAS_KSPLIT_MULTI=0 BMNAS_FUSE_MIX_GEMM=0 BMNAS_HIP_GRAPH=0"
for e in ${1:-$ALL}; do
  echo "== $e"
  env $e timeout 900 python -m pytest tests -m gpu -q -k "not driver and not two_ranks and not bench_ and not found_stage" 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed" | tail -4
done
