#!/bin/bash
# -m gpu suite + smoke + BM25 / hybrid bench lines under rocprofv3 --kernel-trace --stats -> gpurun_out/final/
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/final
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
prof() {
  local name=$1; shift
  rm -rf /tmp/prof_$name
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -- python $ROOT/bench.py "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  local db=$(ls /tmp/prof_$name/*/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python $ROOT/scripts/prof_summary.py $db "rocprofv3 --kernel-trace --stats -- python bench.py $*" > $OUT/kernel_stats_$name.txt 2>&1
  tail -c 300 $OUT/bench_$name.json; echo
}
prof bm25 --workload bm25 --steps 10 --warmup 2
prof hybrid --workload hybrid --steps 10 --warmup 2
