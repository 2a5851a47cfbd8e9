#!/bin/bash
# end-to-end BM25 rate against submitting threads x tickets per thread (native threads), after the score floors
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/bm25_threads
mkdir -p $OUT
cd $ROOT
for cfg in ${CFGS:-12x1 8x2 16x1 10x1 14x1 6x2 8x1 12x1}; do
  th=${cfg%x*}; dp=${cfg#*x}
  NIDX_BENCH_BM25_SEGMENTS=0 NIDX_BENCH_BM25_THREADS=$th NIDX_BENCH_BM25_DEPTH=$dp timeout 600 python bench.py --workload bm25 --cpu-queries 0 > $OUT/b_$cfg.json 2> $OUT/b_$cfg.err
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/b_$cfg.json").read().strip().splitlines()[-1])
    c=d["config"]
    print("$cfg value=%.1f G postings/s host_cores_busy=%.1f throttled=%s one_thread=%.1f" % (d["value"]/1e9, c["host_load"]["host_cores_busy"], c["host_load"]["cgroup_throttled_periods"], (c.get("one_submitting_thread") or {}).get("postings_per_s",0)/1e9))
except Exception as e:
    print("$cfg FAILED", e); print(open("$OUT/b_$cfg.err").read()[-1500:])
PY
done
