#!/bin/bash
# A/B of two builds of the library on the BM25 bench workload: _ab/libnidx_head.so (the previous commit) against the tree's
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p gpurun_out/ab_lib
timeout 900 python -m pytest tests/test_bm25_gpu.py tests/test_bm25_segments_gpu.py -x -q -m gpu > gpurun_out/ab_lib/tests.log 2>&1
echo "tests: $(tail -1 gpurun_out/ab_lib/tests.log)"
for v in head tree head tree head tree; do
  if [ $v = head ]; then export NIDX_GPU_LIB=$ROOT/_ab/libnidx_head.so; else unset NIDX_GPU_LIB; fi
  NIDX_BENCH_BM25_SEGMENTS=0 timeout 600 python bench.py --workload bm25 --cpu-queries 0 > gpurun_out/ab_lib/$v.json 2> gpurun_out/ab_lib/$v.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/ab_lib/$v.json").read().strip().splitlines()[-1])
    print("$v value=%.1f G kernel_ms=%.4f sync_ms=%.4f" % (d["value"]/1e9, d["roofline"]["kernel_ms"], d["config"]["synchronous_entry_ms_per_batch"]))
except Exception as e:
    print("$v FAILED", e); print(open("gpurun_out/ab_lib/$v.err").read()[-800:])
PY
done
