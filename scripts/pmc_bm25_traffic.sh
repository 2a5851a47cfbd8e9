#!/bin/bash
# HBM traffic of the BM25 kernels on the bench batch: FETCH_SIZE and WRITE_SIZE in separate passes (--kernel-trace only with --pmc).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_bm25
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --workload bm25 --steps 4 --warmup 1 --cpu-queries 0"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/p_$c
  timeout 200 rocprofv3 --pmc $c --kernel-trace -d /tmp/p_$c -- $BENCH > /dev/null 2>&1
  db=$(ls /tmp/p_$c/*/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python $ROOT/scripts/prof_summary.py $db "rocprofv3 --pmc $c --kernel-trace -- $BENCH" 2>&1 | grep -E "^#|bm25" > $OUT/traffic_$c.txt
done
cat $OUT/traffic_FETCH_SIZE.txt $OUT/traffic_WRITE_SIZE.txt
