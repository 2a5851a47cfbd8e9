#!/bin/bash
# Re-measures the headline workload on the GPU box and writes everything under gpurun_out/final/:
#   bench line (default bench.py), rocprofv3 --kernel-trace --stats summary, and the two PMC passes (FETCH_SIZE, WRITE_SIZE)
# of the same command.  Usage (from the repo root on the GPU box): bash scripts/refresh_profiles.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/final
mkdir -p $OUT
cd $ROOT
python bench.py > $OUT/bench_hnsw_1m.json 2> $OUT/bench_hnsw_1m.err
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 5 --warmup 1 --cpu-queries 0 --recall-queries 0 --clustered-n 0"
rocprofv3 --kernel-trace --stats -d $OUT/prof_trace -- $BENCH > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/prof_fetch -- $BENCH > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/prof_write -- $BENCH > /dev/null 2>&1
cd $ROOT
for p in trace fetch write; do
  db=$(ls $OUT/prof_$p/*/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python scripts/prof_summary.py $db "$p pass: $BENCH" > $OUT/summary_$p.txt 2>&1
done
rm -rf $OUT/prof_trace $OUT/prof_fetch $OUT/prof_write
tail -c 600 $OUT/bench_hnsw_1m.json
grep -h "hnsw_search_kernel" $OUT/summary_*.txt | cut -c1-200
