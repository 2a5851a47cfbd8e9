#!/bin/bash
# SQ instruction counters of the BM25 scoring kernel on the bench batch (one launch at a time): the latency shape and the throughput shape
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/bm25_insts
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export NIDX_BENCH_BM25_SEGMENTS=0 NIDX_BENCH_BM25_DEPTH=1 NIDX_BENCH_BM25_THREADS=1
for shape in ${SHAPES:-0 1}; do
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES"; do
    tag=$(echo $set | cut -c1-12)
    NIDX_GPU_BM25_CROWDED=$shape timeout 300 rocprofv3 --pmc $set --kernel-trace -d /tmp/bi_$shape -- python $ROOT/bench.py --workload bm25 --steps 4 --warmup 1 --cpu-queries 0 > /dev/null 2>&1
    db=$(ls /tmp/bi_$shape/*/*.db 2>/dev/null | head -1)
    [ -n "$db" ] && python $ROOT/scripts/prof_summary.py $db "NIDX_GPU_BM25_CROWDED=$shape rocprofv3 --pmc $set" 2>&1 | grep -E "^#|bm25_stream" > $OUT/s_${shape}_$tag.txt
    rm -rf /tmp/bi_$shape
    echo "== crowded=$shape $set"; grep "|" $OUT/s_${shape}_$tag.txt | grep -v "^#" | cut -c1-30,100-200
  done
done
