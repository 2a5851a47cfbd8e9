#!/bin/bash
# the pipelined BM25 rate with the runtime's transfers on the SDMA engines (default) and on shader blit kernels (HSA_ENABLE_SDMA=0)
set -u
cd ${GRAFT_REPO_ROOT:-.}
for v in 1 0 1 0; do
  HSA_ENABLE_SDMA=$v NIDX_BENCH_BM25_SEGMENTS=0 timeout 600 python bench.py --workload bm25 --cpu-queries 0 > gpurun_out/sdma_$v.json 2> gpurun_out/sdma_$v.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/sdma_$v.json").read().strip().splitlines()[-1])
    print("HSA_ENABLE_SDMA=$v value=%.1f G kernel_ms=%.4f sync_ms=%.4f one_thread=%.1f G" % (d["value"]/1e9, d["roofline"]["kernel_ms"], d["config"]["synchronous_entry_ms_per_batch"], d["config"]["one_submitting_thread"]["postings_per_s"]/1e9))
except Exception as e:
    print("HSA_ENABLE_SDMA=$v FAILED", e); print(open("gpurun_out/sdma_$v.err").read()[-800:])
PY
done
