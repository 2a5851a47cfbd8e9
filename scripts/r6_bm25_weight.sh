#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/bm25_weight
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_bm25_gpu.py tests/test_bm25_segments_gpu.py -x -q -m gpu > $OUT/tests.log 2>&1
echo "tests: $(tail -1 $OUT/tests.log)"
for w in ${WEIGHTS:-1 3 2 4 1 3}; do
  NIDX_GPU_BM25_SHORT_WEIGHT=$w timeout 600 python bench.py --workload bm25 --cpu-queries 0 > $OUT/bench_$w.json 2> $OUT/bench_$w.err
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_$w.json").read().strip().splitlines()[-1])
    r=d.get("roofline",{})
    print("weight=$w value=%.1f G postings/s kernel_ms=%.4f frac=%.4f sync_ms=%s" % (d["value"]/1e9, r.get("kernel_ms"), r.get("frac"), d["config"].get("synchronous_entry_ms_per_batch")))
except Exception as e:
    print("weight=$w FAILED", e); print(open("$OUT/bench_$w.err").read()[-2000:])
PY
done
