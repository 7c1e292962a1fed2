#!/bin/bash
# The GPU parity suite under every kernel-selection toggle: the merged / pipelined launches are the
# default, the toggles force the paths that shapes outside them take.  Usage (via gpurun):
#   bash tools/test_matrix.sh
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for e in "BMNAS_DEFAULT=1" "BMNAS_CONV_PIPE=0" "BMNAS_FUSE_ATTN_GEMM=0" "BMNAS_FUSE_PROLOGUE=0" "BMNAS_FUSE_EPILOGUE=0" "BMNAS_FUSE_BN_FINALIZE=0" "BMNAS_FUSE_HEAD=0" "BMNAS_FUSE_PROLOGUE_PAIR=0" "BMNAS_FUSE_BN_APPLY=0" "BMNAS_FUSE_BWD_PAIR=0" "BMNAS_FUSE_BN_TAIL=0" "BMNAS_FUSE_INNER_SUM=0" "BMNAS_FUSE_LN_BWD=0" "BMNAS_FUSE_NEXT_PAIR=0" "BMNAS_FUSE_MIX_EPILOGUE=0" "BMNAS_KSPLIT_MULTI=0" \
         "BMNAS_HIP_GRAPH=0"; do
  echo "== $e"
  env $e timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed" | tail -4
done
