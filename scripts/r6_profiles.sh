#!/bin/bash
# Round 6: the evidence behind DESIGN.md's [measured] figures.  bash scripts/r6_profiles.sh [part ...]
# parts: headline bm25 bf16 rabitq exchange crowded   (default: all but bf16).  Summaries land in gpurun_out/final6/ (scripts/r6_collect.sh -> profiles/r06_*).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/final6
mkdir -p $OUT
PARTS=${@:-headline bm25 rabitq exchange crowded}
cd /tmp && export TMPDIR=/tmp
prof() {  # name, bench args...
  local name=$1; shift
  timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -- python $ROOT/bench.py "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err < /dev/null
  local db=$(ls /tmp/prof_$name/*/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python $ROOT/scripts/prof_summary.py $db "rocprofv3 --kernel-trace --stats -- python bench.py $*" > $OUT/kernel_stats_$name.txt 2>&1
  rm -rf /tmp/prof_$name
  tail -c 300 $OUT/bench_$name.json; echo
}
pmc() {  # name, counter, bench args...
  local name=$1 c=$2; shift; shift
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_${name}_$c -- python $ROOT/bench.py "$@" > /dev/null 2>&1 < /dev/null
  local db=$(ls /tmp/pmc_${name}_$c/*/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python $ROOT/scripts/prof_summary.py $db "rocprofv3 --pmc $c --kernel-trace -- python bench.py $*" > $OUT/pmc_${name}_$c.txt 2>&1
  rm -rf /tmp/pmc_${name}_$c
}
HNSW_OFF="--cpu-queries 0 --parity-queries 0 --scan-check-queries 0 --ref-build-n 0 --single-query-calls 0 --bf16-block-n 0 --bm25-block 0 --iso-recall 0 --segment-regime 0 --corpus clustered"
for PART in $PARTS; do
case $PART in
headline)
  ( cd $ROOT && timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err )
  tail -c 300 $OUT/bench_default.json; echo ;;
bm25)
  export NIDX_BENCH_BM25_SEGMENTS=0
  NIDX_BENCH_BM25_DEPTH=1 NIDX_BENCH_BM25_THREADS=1 prof bm25_one_at_a_time --workload bm25 --cpu-queries 0 --steps 200
  NIDX_GPU_BM25_FUSED_MERGE=0 NIDX_BENCH_BM25_DEPTH=1 NIDX_BENCH_BM25_THREADS=1 prof bm25_one_at_a_time_two_launches --workload bm25 --cpu-queries 0 --steps 200
  prof bm25_pipelined --workload bm25 --cpu-queries 0 --steps 200
  pmc bm25 FETCH_SIZE --workload bm25 --steps 4 --warmup 1 --cpu-queries 0
  pmc bm25 WRITE_SIZE --workload bm25 --steps 4 --warmup 1 --cpu-queries 0
  ( cd $ROOT && timeout 500 bash scripts/bm25_batch_curve.sh 256 1024 4096 16384 < /dev/null > $OUT/bm25_batch_curve.txt 2>&1 )
  unset NIDX_BENCH_BM25_SEGMENTS
  ( cd $ROOT && timeout 400 python bench.py --workload bm25 --steps 200 > $OUT/bench_bm25.json 2> $OUT/bench_bm25.err )
  tail -c 300 $OUT/bench_bm25.json; echo ;;
bf16)
  prof bf16_12m5x1024 --workload bf16 --n-vectors 12500000 --dim 1024 --cpu-queries 0 --steps 5 --warmup 2 ;;
rabitq)
  prof rabitq_1m --workload rabitq --n-vectors 1000000 --steps 5 --warmup 1 ;;
exchange)
  # the N > 1 timed path at the shape each rank of the 8-GPU run will see, on one GPU: device entry + RCCL world-of-one all-gather on the side stream
  ( cd $ROOT && NIDX_BENCH_FORCE_EXCHANGE=1 timeout 900 python bench.py --gpus 1 --n-vectors 12500000 $HNSW_OFF > $OUT/bench_force_exchange_12m5.json 2> $OUT/bench_force_exchange_12m5.err )
  tail -c 300 $OUT/bench_force_exchange_12m5.json; echo
  ( cd $ROOT && timeout 900 python bench.py --n-vectors 12500000 $HNSW_OFF > $OUT/bench_pipeline_12m5.json 2> $OUT/bench_pipeline_12m5.err )
  tail -c 300 $OUT/bench_pipeline_12m5.json; echo ;;
crowded)
  # the kernel that is actually timed when batches overlap: hnsw_search_kernel<3,2,5,1> (tunable launch_shape = 1), one launch at a time for the counters
  export NIDX_BENCH_TUNABLES=launch_shape=1
  B="--n-vectors 10000000 $HNSW_OFF --steps 10 --warmup 2 --recall-queries 0 --batches-in-flight 1 --graph-cache /tmp/nidx_graphs"
  prof hnsw10m_crowded_shape $B
  pmc hnsw10m_crowded_shape FETCH_SIZE $B
  pmc hnsw10m_crowded_shape WRITE_SIZE $B
  unset NIDX_BENCH_TUNABLES ;;
esac
done
ls -la $OUT | tail -40
