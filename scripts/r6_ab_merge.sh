#!/bin/bash
# what bm25_merge_kernel costs the pipelined BM25 rate: NIDX_GPU_BM25_ABLATE_MERGE = 0 (product), 1 (no launch), 2 (an empty launch), 3 (merge, no output)
set -u
cd ${GRAFT_REPO_ROOT:-.}
for v in ${VARIANTS:-0 1 2 3 0}; do
  NIDX_GPU_BM25_ABLATE_MERGE=$v NIDX_BENCH_BM25_SEGMENTS=0 timeout 600 python bench.py --workload bm25 --cpu-queries 0 > gpurun_out/abm_$v.json 2> gpurun_out/abm_$v.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/abm_$v.json").read().strip().splitlines()[-1])
    print("ablate=$v us_per_batch=%.2f (= %.1f G postings/s at 8.4 M postings per batch) one_thread_us=%.1f" % (d["ms_per_step"]*1e3, 8.404/d["ms_per_step"]/1e3*1e3, d["config"]["one_submitting_thread"]["ms_per_step"]*1e3))
except Exception as e:
    print("ablate=$v FAILED", e); print(open("gpurun_out/abm_$v.err").read()[-800:])
PY
done
