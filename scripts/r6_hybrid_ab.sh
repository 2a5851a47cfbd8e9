#!/bin/bash
# the hybrid block with the BM25 merge fused into the scoring launch and in a launch of its own
set -u
cd ${GRAFT_REPO_ROOT:-.}
for v in 1 0 1 0; do
  NIDX_GPU_BM25_FUSED_MERGE=$v timeout 900 python bench.py --workload hybrid --cpu-queries 0 > gpurun_out/hyb_$v.json 2> gpurun_out/hyb_$v.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/hyb_$v.json").read().strip().splitlines()[-1])
    c=d.get("config",{})
    print("fused=$v hybrid=%.3f M q/s parts=%s bm25_kernel_ms=%s" % (d["value"]/1e6, json.dumps(c.get("ms_per_step_parts") or d.get("ms_per_step_parts")), c.get("bm25_kernel_ms") or d.get("bm25_kernel_ms")))
except Exception as e:
    print("fused=$v FAILED", e); print(open("gpurun_out/hyb_$v.err").read()[-800:])
PY
done
