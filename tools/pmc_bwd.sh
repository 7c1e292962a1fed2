#!/bin/bash
# PMC passes over the merged backward GEMM launch of the real step, one block class at a time
# (BMNAS_CONV_PROBE 48 = data-gradient tiles only, 80 = weight-gradient tiles only, 0 = all).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
OUT=gpurun_out/pmc_bwd; rm -rf $OUT; mkdir -p $OUT
B="--mode eager --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-full-step"
for probe in ${PROBES:-48 80 0}; do
  i=0
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
             "SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_WR" \
             "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    BMNAS_CONV_PROBE=$probe timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 bench.py $B > $OUT/p$i.log 2>&1
    f=$(find $OUT/p$i -name '*counter_collection.csv' | head -1)
    echo "== probe $probe set $i" >> $OUT/summary.txt
    [ -n "$f" ] && python3 tools/pmc.py "$f" ${KPAT:-conv_bwd_all_pipe} | grep -v "^$" >> $OUT/summary.txt
    rm -rf $OUT/p$i
  done
done
cat $OUT/summary.txt
