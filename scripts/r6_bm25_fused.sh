#!/bin/bash
# the fused merge: parity suites, then the pipelined bench with and without it
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/bm25_fused
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests/test_bm25_gpu.py tests/test_bm25_segments_gpu.py tests/test_bm25_aux_gpu.py tests/test_text_gpu.py tests/test_scale_parity_gpu.py -x -q -m gpu > $OUT/tests.log 2>&1
echo "tests: $(tail -1 $OUT/tests.log)"; grep -E "^FAILED|^ERROR|Error|assert" $OUT/tests.log | head -10
for v in 1 0 1 0; do
  NIDX_GPU_BM25_FUSED_MERGE=$v NIDX_BENCH_BM25_SEGMENTS=0 timeout 600 python bench.py --workload bm25 --cpu-queries 0 > $OUT/b_$v.json 2> $OUT/b_$v.err
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/b_$v.json").read().strip().splitlines()[-1])
    print("fused=$v value=%.1f G kernel_ms=%.4f sync_ms=%.4f parity=%s" % (d["value"]/1e9, d["roofline"]["kernel_ms"], d["config"]["synchronous_entry_ms_per_batch"], (d["config"].get("parity") or {}).get("status")))
except Exception as e:
    print("fused=$v FAILED", e); print(open("$OUT/b_$v.err").read()[-800:])
PY
done
