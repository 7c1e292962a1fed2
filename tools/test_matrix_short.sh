cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for e in "BMNAS_FUSE_BWD_ALL=0" "BMNAS_FUSE_ATTN_GEMM=0" "BMNAS_CONV_PIPE=0" "BMNAS_FUSE_BN_FINALIZE=0" "BMNAS_FUSE_LN_BWD=0" "BMNAS_FUSE_BN_APPLY=0" "BMNAS_FUSE_BWD_PAIR=0" "BMNAS_FUSE_HEAD=0"; do
  echo "== $e"
  env $e timeout 600 python -m pytest tests -m gpu -q 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed" | tail -4
done
