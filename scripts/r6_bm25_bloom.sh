#!/bin/bash
# bitmap A as a blocked Bloom filter (three bits per document inside its word): parity suites, then the bench workload
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/bm25_bloom
mkdir -p $OUT
cd $ROOT
timeout 1200 python -m pytest tests/test_bm25_gpu.py tests/test_bm25_segments_gpu.py tests/test_bm25_aux_gpu.py tests/test_text_gpu.py tests/test_scale_parity_gpu.py -x -q -m gpu > $OUT/tests.log 2>&1
echo "tests: $(tail -1 $OUT/tests.log)"
for i in 1 2; do
  NIDX_BENCH_BM25_SEGMENTS=0 timeout 600 python bench.py --workload bm25 --cpu-queries 0 > $OUT/bench_$i.json 2> $OUT/bench_$i.err
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_$i.json").read().strip().splitlines()[-1])
    print("run $i value=%.1f G kernel_ms=%.4f frac=%.4f parity=%s" % (d["value"]/1e9, d["roofline"]["kernel_ms"], d["roofline"]["frac"], (d["config"].get("parity") or {}).get("status")))
except Exception as e:
    print("run $i FAILED", e); print(open("$OUT/bench_$i.err").read()[-800:])
PY
done
