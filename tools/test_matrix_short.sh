#!/bin/bash
# A shorter matrix for the end of a round (gpurun calls are capped at 20 minutes): the toggles whose paths the
# latest changes touch.  Usage (via gpurun): bash tools/test_matrix_short.sh "TOGGLE=0 TOGGLE=0 ..."
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for e in ${1:-BMNAS_FUSE_NEXT_PAIR=0 BMNAS_FUSE_MIX_EPILOGUE=0 BMNAS_FUSE_BWD_PAIR=0 BMNAS_FUSE_BN_TAIL=0 BMNAS_FUSE_INNER_SUM=0 BMNAS_FUSE_PROLOGUE_PAIR=0 BMNAS_HIP_GRAPH=0 BMNAS_CONV_PIPE=0}; do
  echo "== $e"
  env $e timeout 600 python -m pytest tests -m gpu -q 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed" | tail -4
done
