#!/bin/bash
# SQ counters of the BM25 kernels on the bench.py --workload bm25 batch (separate passes; --kernel-trace only with --pmc).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_bm25
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --workload bm25 --steps 3 --warmup 1 --cpu-queries 0 ${BM25_ARGS:-}"
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace -d $OUT/p$i -- $BENCH > /dev/null 2>&1
  db=$(ls $OUT/p$i/*/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python $ROOT/scripts/prof_summary.py $db "pmc pass $i: $set" 2>&1 | grep -E "^#|bm25" > $OUT/summary_$i.txt
  rm -rf $OUT/p$i
done
cat $OUT/summary_*.txt
