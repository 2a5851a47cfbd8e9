#!/bin/bash
# SQ instruction counters of the BM25 scoring kernel on the bench batch (one launch at a time), after the score floors
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/bm25_insts
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export NIDX_BENCH_BM25_SEGMENTS=0 NIDX_BENCH_BM25_DEPTH=1 NIDX_BENCH_BM25_THREADS=1
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  for fl in 1 0; do
    NIDX_GPU_BM25_FLOOR=$fl timeout 300 rocprofv3 --pmc $set --kernel-trace -d /tmp/bi_${i}_$fl -- python $ROOT/bench.py --workload bm25 --steps 4 --warmup 1 --cpu-queries 0 > /dev/null 2>&1
    db=$(ls /tmp/bi_${i}_$fl/*/*.db 2>/dev/null | head -1)
    [ -n "$db" ] && python $ROOT/scripts/prof_summary.py $db "floor=$fl rocprofv3 --pmc $set" 2>&1 | grep -E "^#|bm25_stream|bm25_merge" > $OUT/s_${i}_$fl.txt
    rm -rf /tmp/bi_${i}_$fl
    echo "== floor=$fl $set"; grep "|" $OUT/s_${i}_$fl.txt | grep -v "^#" | cut -c1-40,90-200
  done
done
