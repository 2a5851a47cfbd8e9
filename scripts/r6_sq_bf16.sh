#!/bin/bash
# SQ counters of bf16_append_kernel (BASELINE.json configs[4] on one GPU's share: 12.5 M x 1024, batch 1 024): separate rocprofv3 --pmc passes.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/sq_bf16
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --workload bf16 --n-vectors ${NVEC:-12500000} --dim 1024 --cpu-queries 0 --steps 3 --warmup 1"
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d /tmp/sqbf$i -- $BENCH > $OUT/bench_$i.json 2>/dev/null
  db=$(ls /tmp/sqbf$i/*/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python $ROOT/scripts/prof_summary.py $db "pmc pass $i: $set -- $BENCH" 2>&1 | grep -E "^#|bf16" > $OUT/summary_$i.txt
  rm -rf /tmp/sqbf$i
done
cat $OUT/summary_*.txt
