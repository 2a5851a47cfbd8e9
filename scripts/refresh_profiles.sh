#!/bin/bash
# Round 4-5: re-measures the headline workload (bench.py default: 10 M x 768 cosine HNSW, batch 1024) on the GPU box and writes under
# gpurun_out/final/: the bench line, and per corpus (clustered = the timed one, uniform = the second figure) the rocprofv3
# --kernel-trace --stats summary plus the two PMC passes (FETCH_SIZE, WRITE_SIZE; separate passes, --kernel-trace only) of the same
# command with the side legs switched off.  scripts/make_pmc_traffic.py turns the summaries into profiles/r05_pmc_traffic.json.
# Usage (repo root, GPU box): bash scripts/refresh_profiles.sh [n_vectors]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
N=${1:-10000000}
OUT=$ROOT/gpurun_out/final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for corpus in clustered uniform; do
  BENCH="python $ROOT/bench.py --n-vectors $N --corpus $corpus --steps 10 --warmup 2 --cpu-queries 0 --parity-queries 0 --scan-check-queries 0 --ref-build-n 0 --single-query-calls 0 --recall-queries 0 --bf16-block-n 0 --bm25-block 0 --iso-recall 0"
  # the PMC passes load the graph the trace pass built (rocprofv3 --pmc segfaults over the thousands of dispatches of a 10 M build)
  BENCH="$BENCH --graph-cache /tmp/nidx_graphs"
  [ $corpus = uniform ] && BENCH="$BENCH --batches-in-flight 1"   # the default run times this corpus one launch at a time
  PMCBENCH="$BENCH --batches-in-flight 1"   # counter collection serialises dispatches; concurrent streams crash rocprofv3 --pmc on this box
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_trace_$corpus -- $PMCBENCH > $OUT/bench_${corpus}_profiled.json 2> /dev/null
  if [ $corpus = clustered ]; then   # and with the default number of batches in flight (durations then overlap)
    timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_trace3_$corpus -- $BENCH > $OUT/bench_${corpus}_profiled_in_flight.json 2> /dev/null
    db=$(ls $OUT/prof_trace3_$corpus/*/*.db 2>/dev/null | head -1)
    [ -n "$db" ] && python $ROOT/scripts/prof_summary.py $db "$corpus corpus, trace pass with the default batches in flight: $BENCH" > $OUT/summary_trace_in_flight_$corpus.txt 2>&1
    rm -rf $OUT/prof_trace3_$corpus
  fi
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/prof_fetch_$corpus -- $PMCBENCH > /dev/null 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/prof_write_$corpus -- $PMCBENCH > /dev/null 2>&1
  for p in trace fetch write; do
    db=$(ls $OUT/prof_${p}_$corpus/*/*.db 2>/dev/null | head -1)
    [ -n "$db" ] && python $ROOT/scripts/prof_summary.py $db "$corpus corpus, $p pass: $BENCH" > $OUT/summary_${p}_$corpus.txt 2>&1
    rm -rf $OUT/prof_${p}_$corpus
  done
done
grep -h "hnsw_search_kernel" $OUT/summary_*.txt | cut -c1-220
