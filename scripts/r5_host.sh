#!/bin/bash
# host-buffer pipeline A/B: batches in flight x staging helpers (10 M clustered shard, side blocks off)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out/r5host
ARGS="--corpus clustered --parity-queries 0 --scan-check-queries 0 --segment-regime 0 --bf16-block-n 0 --ref-build-n 0 --single-query-calls 0 --cpu-queries 0 --bm25-block 0 --iso-recall 0 --graph-cache gpurun_out/r5host/cache"
for v in "3 3" "4 3" "5 3" "6 3" "8 3"; do
  set -- $v
  NIDX_BENCH_HOST_IN_FLIGHT=$1 NIDX_GPU_STAGE_THREADS=$2 timeout 300 python bench.py $ARGS > gpurun_out/r5host/h_$1_$2.json 2> gpurun_out/r5host/h_$1_$2.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r5host/h_$1_$2.json")); c=d["config"]
print("in_flight=$1 helpers=$2 value=%.4g host=%.4g frac=%.3f blocking=%.4g" % (d["value"], c["host_buffer_queries_per_s"], c["host_buffer_fraction_of_value"], c["host_buffer_blocking_queries_per_s"]))
PY
done
rm -rf gpurun_out/r5host/cache
