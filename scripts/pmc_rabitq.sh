#!/bin/bash
# SQ counters of the RaBitQ walk on the bench.py --workload rabitq batch (separate passes; --kernel-trace only with --pmc).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_rabitq
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --workload rabitq --n-vectors 1000000 --steps 3 --warmup 1 --cpu-queries 0 --batches-in-flight 1 ${RQ_ARGS:-}"
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" "TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace -d $OUT/p$i -- $BENCH > /dev/null 2>&1
  db=$(ls $OUT/p$i/*/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python $ROOT/scripts/prof_summary.py $db "pmc pass $i: $set" 2>&1 | grep -E "^#|rabitq" > $OUT/summary_$i.txt
  rm -rf $OUT/p$i
done
cat $OUT/summary_*.txt
