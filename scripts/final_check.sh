#!/bin/bash
# Round-end check on the GPU box: the whole -m gpu suite, smoke(), the default bench line, and rocprofv3 kernel stats of the headline
# kernel (clustered corpus, one launch at a time) and of the BM25 workload.  Writes gpurun_out/final/.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/final
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 200 $OUT/bench_default.json; echo
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --corpus clustered --steps 10 --warmup 2 --cpu-queries 0 --parity-queries 0 --scan-check-queries 0 --ref-build-n 0 --single-query-calls 0 --recall-queries 0 --segment-regime 0 --batches-in-flight 1"
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_c -- $BENCH > $OUT/bench_hnsw10m_clustered_profiled.json 2> /dev/null
db=$(ls $OUT/prof_c/*/*.db 2>/dev/null | head -1)
[ -n "$db" ] && python $ROOT/scripts/prof_summary.py $db "clustered corpus, trace pass: $BENCH" > $OUT/summary_trace_clustered.txt 2>&1
rm -rf $OUT/prof_c
grep -h "hnsw_search_kernel" $OUT/summary_trace_clustered.txt | cut -c1-200
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_b -- python $ROOT/bench.py --workload bm25 --steps 10 --warmup 2 > $OUT/bench_bm25.json 2> $OUT/bench_bm25.err
db=$(ls $OUT/prof_b/*/*.db 2>/dev/null | head -1)
[ -n "$db" ] && python $ROOT/scripts/prof_summary.py $db "rocprofv3 --kernel-trace --stats -- python bench.py --workload bm25 --steps 10 --warmup 2" > $OUT/kernel_stats_bm25.txt 2>&1
rm -rf $OUT/prof_b
grep -h "bm25_" $OUT/kernel_stats_bm25.txt | cut -c1-160
