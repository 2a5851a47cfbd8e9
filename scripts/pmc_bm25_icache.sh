#!/bin/bash
# instruction-cache counters of the BM25 union kernels (stream: NIDX_GPU_BM25_UNION=1, lockstep: =3) on the bench batch
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_bm25_ic
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --workload bm25 --steps 3 --warmup 1 --cpu-queries 0"
for mode in 1 3; do
  i=0
  for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD"; do
    i=$((i+1))
    NIDX_GPU_BM25_UNION=$mode timeout 200 rocprofv3 --pmc $set --kernel-trace -d $OUT/m${mode}p$i -- $BENCH > $OUT/log_m${mode}p$i.txt 2>&1
    db=$(ls $OUT/m${mode}p$i/*/*.db 2>/dev/null | head -1)
    [ -n "$db" ] && python $ROOT/scripts/prof_summary.py $db "mode $mode pass $i: $set" 2>&1 | grep -E "^#|bm25_(union|stream)" > $OUT/summary_m${mode}_$i.txt
    rm -rf $OUT/m${mode}p$i
  done
done
cat $OUT/summary_*.txt
