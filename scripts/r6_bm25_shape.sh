#!/bin/bash
# end-to-end BM25 rate against the throughput shape's slice length (the planner's collision estimate follows the three-bit filter)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/bm25_shape
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_bm25_gpu.py tests/test_scale_parity_gpu.py -x -q -m gpu > $OUT/tests.log 2>&1
echo "tests: $(tail -1 $OUT/tests.log)"
run() {
  NIDX_BENCH_BM25_SEGMENTS=0 timeout 600 python bench.py --workload bm25 --cpu-queries 0 > $OUT/b_$1.json 2> $OUT/b_$1.err
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/b_$1.json").read().strip().splitlines()[-1])
    r=d.get("roofline",{})
    print("$1 value=%.1f G postings/s kernel_ms=%.4f" % (d["value"]/1e9, r.get("kernel_ms")))
except Exception as e:
    print("$1 FAILED", e); print(open("$OUT/b_$1.err").read()[-1500:])
PY
}
run auto
NIDX_GPU_BM25_CROWDED_SLICE=4096 run crowded4096
NIDX_GPU_BM25_CROWDED_SLICE=16384 run crowded16384
NIDX_GPU_BM25_CROWDED_SLICE=32768 run crowded32768
NIDX_GPU_BM25_CROWDED=1 run always_crowded
run auto2
