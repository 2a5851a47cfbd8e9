#!/bin/bash
# Round 5: the evidence behind DESIGN.md's [measured] figures, one GPU call: bash scripts/r5_profiles.sh [part ...]
# parts: headline bm25 hybrid rabitq others  (default: all).  Summaries land in gpurun_out/final/; copy them to profiles/r05_* afterwards
# (scripts/r5_collect.sh).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/final
mkdir -p $OUT
PARTS=${@:-headline bm25 hybrid rabitq others}
cd /tmp && export TMPDIR=/tmp
prof() {  # name, bench args...
  local name=$1; shift
  timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof_$name -- python $ROOT/bench.py "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err < /dev/null
  local db=$(ls $OUT/prof_$name/*/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python $ROOT/scripts/prof_summary.py $db "rocprofv3 --kernel-trace --stats -- python bench.py $*" > $OUT/kernel_stats_$name.txt 2>&1
  rm -rf $OUT/prof_$name
  tail -c 300 $OUT/bench_$name.json; echo
}
pmc() {  # name, counter, bench args...
  local name=$1 c=$2; shift; shift
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d $OUT/pmc_${name}_$c -- python $ROOT/bench.py "$@" > /dev/null 2>&1 < /dev/null
  local db=$(ls $OUT/pmc_${name}_$c/*/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python $ROOT/scripts/prof_summary.py $db "rocprofv3 --pmc $c --kernel-trace -- python bench.py $*" > $OUT/pmc_${name}_$c.txt 2>&1
  rm -rf $OUT/pmc_${name}_$c
}
for PART in $PARTS; do
case $PART in
headline)
  timeout 900 python $ROOT/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
  tail -c 300 $OUT/bench_default.json; echo
  bash $ROOT/scripts/refresh_profiles.sh ;;
bm25)
  export NIDX_BENCH_BM25_SEGMENTS=0
  NIDX_BENCH_BM25_DEPTH=1 NIDX_BENCH_BM25_THREADS=1 prof bm25_one_at_a_time --workload bm25 --cpu-queries 0 --steps 200
  prof bm25_pipelined --workload bm25 --cpu-queries 0 --steps 200
  timeout 400 bash $ROOT/scripts/pmc_bm25_traffic.sh < /dev/null > $OUT/pmc_bm25_traffic.log 2>&1
  timeout 500 bash $ROOT/scripts/bm25_batch_curve.sh 256 1024 4096 16384 < /dev/null > $OUT/bm25_batch_curve.txt 2>&1
  unset NIDX_BENCH_BM25_SEGMENTS
  timeout 400 python $ROOT/bench.py --workload bm25 --steps 200 > $OUT/bench_bm25.json 2> $OUT/bench_bm25.err
  tail -c 300 $OUT/bench_bm25.json; echo ;;
hybrid)
  prof hybrid --workload hybrid --steps 400 --warmup 10 --cpu-queries 0
  NIDX_BENCH_HYBRID_SHAPE=0 prof hybrid_default_shape --workload hybrid --steps 400 --warmup 10 --cpu-queries 0 ;;
rabitq)
  prof rabitq_1m --workload rabitq --n-vectors 1000000 --steps 5 --warmup 1
  pmc rabitq_1m FETCH_SIZE --workload rabitq --n-vectors 1000000 --steps 3 --warmup 1 --cpu-queries 0 --batches-in-flight 1
  pmc rabitq_1m WRITE_SIZE --workload rabitq --n-vectors 1000000 --steps 3 --warmup 1 --cpu-queries 0 --batches-in-flight 1
  NIDX_GPU_RABITQ_WAVES=2 prof rabitq_1m_two_waves --workload rabitq --n-vectors 1000000 --steps 5 --warmup 1 --cpu-queries 0 ;;
hnsw1m)
  prof hnsw1m --n-vectors 1000000 --corpus clustered --bf16-block-n 0 --bm25-block 0 --single-query-calls 0 --segment-regime 0 --ref-build-n 0 ;;
others)
  prof hnsw1m --n-vectors 1000000 --corpus clustered --bf16-block-n 0 --bm25-block 0 --single-query-calls 0 --segment-regime 0 --ref-build-n 0
  prof scan_1m --workload scan --n-vectors 1000000 --steps 3 --warmup 1
  prof mfma_1m_k10 --workload mfma --n-vectors 1000000 --steps 5 --warmup 1
  prof mfma_1m_k64 --workload mfma --n-vectors 1000000 --steps 5 --warmup 1 --k 64
  prof bf16_12m5x1024 --workload bf16 --n-vectors 12500000 --dim 1024 --steps 5 --warmup 1 --cpu-queries 0 ;;
esac
done
ls -la $OUT | tail -50
