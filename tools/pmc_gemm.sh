#!/bin/bash
# PMC passes over the isolated GEMM launches (tools/gemm_probe.py); prints per-kernel counter averages
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
OUT=gpurun_out/pmc_gemm; rm -rf $OUT; mkdir -p $OUT
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" \
           "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 tools/gemm_probe.py > $OUT/p$i.log 2>&1
  f=$(find $OUT/p$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 tools/pmc.py "$f" conv | grep -v "^$" > $OUT/p$i.txt
  rm -rf $OUT/p$i
  cat $OUT/p$i.txt
done
