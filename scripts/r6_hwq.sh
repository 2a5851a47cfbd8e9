#!/bin/bash
# the reference-constants leg (ef_upper = 1: a launch lasts as long as its longest walk) against the number of hardware queues the HIP runtime uses
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/hwq
mkdir -p $OUT
cd $ROOT
for hq in ${HQS:-4 8 16}; do
  GPU_MAX_HW_QUEUES=$hq timeout 900 python bench.py --bm25-block 0 --bf16-block-n 0 --segment-regime 0 --cpu-queries 0 --parity-queries 0 --single-query-calls 0 --ref-build-n 0 --corpus clustered --scan-check-queries 0 --iso-target 0.9 > $OUT/b_$hq.json 2> $OUT/b_$hq.err
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/b_$hq.json").read().strip().splitlines()[-1])
    rc=d["config"].get("reference_constants") or {}
    print("hwq=$hq value=%.3f M q/s host_buffer=%s ref_consts=%s" % (d["value"]/1e6, d["config"].get("host_buffer_queries_per_s"), json.dumps(rc.get("by_batches_in_flight"))))
except Exception as e:
    print("hwq=$hq FAILED", e); print(open("$OUT/b_$hq.err").read()[-1500:])
PY
done
