#!/bin/bash
# Round 4-5: every bench.py workload once on the GPU box, each under rocprofv3 --kernel-trace --stats, summaries under gpurun_out/final/.
# Usage (repo root, GPU box): bash scripts/refresh_all.sh [part]   part = headline | bf16 | hybrid | bm25stats | bm25 | others | all
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
PART=${1:-all}
OUT=$ROOT/gpurun_out/final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
prof() {  # name, bench args...
  local name=$1; shift
  timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof_$name -- python $ROOT/bench.py "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err < /dev/null
  local db=$(ls $OUT/prof_$name/*/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python $ROOT/scripts/prof_summary.py $db "rocprofv3 --kernel-trace --stats -- python bench.py $*" > $OUT/kernel_stats_$name.txt 2>&1
  rm -rf $OUT/prof_$name
  tail -c 400 $OUT/bench_$name.json; echo
}
if [ "$PART" = headline ] || [ "$PART" = all ]; then
  timeout 1500 python $ROOT/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
  tail -c 300 $OUT/bench_default.json; echo
  bash $ROOT/scripts/refresh_profiles.sh
fi
if [ "$PART" = bf16 ]; then   # BASELINE configs[4] on one GPU's share: per-kernel durations of the bf16 fallback
  prof bf16_12m5x1024 --workload bf16 --n-vectors 12500000 --dim 1024 --steps 5 --warmup 1 --cpu-queries 0 --recall-queries 16
fi
if [ "$PART" = hybrid ]; then   # BASELINE configs[2]: the BM25 launches beside the walks (in-contention trace)
  prof hybrid --workload hybrid --steps 10 --warmup 2 --cpu-queries 0
fi
if [ "$PART" = bm25stats ]; then   # only the per-kernel durations of the BM25 launches, one batch at a time
  NIDX_BENCH_BM25_DEPTH=1 prof bm25_one_at_a_time --workload bm25 --cpu-queries 0 --steps 200
fi
if [ "$PART" = bm25 ] || [ "$PART" = all ]; then
  # the BM25 evidence of DESIGN 4.4: kernel stats one batch at a time (what the bench line's HIP events time), HBM traffic, the batch curve,
  # SQ / instruction-cache counters of both union kernels  (every command bounded: a stalled step must not eat the GPU budget)
  NIDX_BENCH_BM25_DEPTH=1 prof bm25_one_at_a_time --workload bm25 --cpu-queries 0 --steps 200
  timeout 400 bash $ROOT/scripts/pmc_bm25_traffic.sh < /dev/null > $OUT/pmc_bm25_traffic.log 2>&1
  timeout 500 bash $ROOT/scripts/bm25_batch_curve.sh 256 1024 4096 16384 < /dev/null > $OUT/bm25_batch_curve.txt 2>&1
  timeout 500 bash $ROOT/scripts/pmc_bm25_icache.sh < /dev/null > $OUT/pmc_bm25_counters.txt 2>&1
fi
if [ "$PART" = others ] || [ "$PART" = all ]; then
  prof bm25 --workload bm25 --steps 10 --warmup 2
  prof hybrid --workload hybrid --steps 10 --warmup 2
  prof scan_1m --workload scan --n-vectors 1000000 --steps 3 --warmup 1
  prof mfma_1m --workload mfma --n-vectors 1000000 --steps 5 --warmup 1
  prof bf16_12m5x1024 --workload bf16 --n-vectors 12500000 --dim 1024 --steps 5 --warmup 1 --cpu-queries 0
  prof rabitq_1m --workload rabitq --n-vectors 1000000 --steps 5 --warmup 1
fi
