#!/bin/bash
# BM25 score floors: parity suites, then the bench workload with and without them (kernel us by HIP events, end to end, one at a time)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/bm25_floor
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_bm25_gpu.py tests/test_bm25_segments_gpu.py tests/test_bm25_aux_gpu.py tests/test_text_gpu.py -x -q -m gpu > $OUT/tests.log 2>&1
echo "tests: $(tail -1 $OUT/tests.log)"
for fl in 1 0 1 0; do
  NIDX_GPU_BM25_FLOOR=$fl timeout 600 python bench.py --workload bm25 --cpu-queries 0 > $OUT/bench_$fl.json 2> $OUT/bench_$fl.err
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_$fl.json").read().strip().splitlines()[-1])
    r=d.get("roofline",{})
    print("floor=$fl value=%.1f G postings/s kernel_ms=%s frac=%s parity=%s" % (d["value"]/1e9, r.get("kernel_ms"), r.get("frac"), (d["config"].get("parity") or {}).get("status")))
except Exception as e:
    print("floor=$fl FAILED", e); print(open("$OUT/bench_$fl.err").read()[-2000:])
PY
done
