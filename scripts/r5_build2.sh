#!/bin/bash
# build-time A/B on the 10 M clustered shard (side blocks off): the cap on a batch of insertions
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out/r5build
ARGS="--corpus clustered --parity-queries 0 --scan-check-queries 0 --segment-regime 0 --bf16-block-n 0 --ref-build-n 0 --single-query-calls 0 --cpu-queries 0 --bm25-block 0 --iso-recall 0 --steps 5 --min-timed-s 0.2"
for v in ${@:-8192 16384 32768}; do
  env NIDX_GPU_BUILD_MAX_BATCH=$v timeout 300 python bench.py $ARGS > gpurun_out/r5build/mb_$v.json 2> gpurun_out/r5build/mb_$v.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r5build/mb_$v.json")); c=d["config"]; b=c["build"]
print("max_batch=$v build_s=%.2f kernels=%.2f frac=%.3f recall=%.4f evals/insert=%.0f value=%.4g" % (c["hnsw_build_s"], b["seconds_of_kernels"], b["roofline"]["frac"], c["recall_at_10"], b["search_distance_evals_per_insert"], d["value"]))
PY
done
