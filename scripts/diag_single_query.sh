#!/bin/bash
# The single-query (coalescer) latency leg a few times over, with the slow-search trace on.  Run on the GPU box.
mkdir -p gpurun_out/sq
for i in 1 2 3 4 5; do
    NIDX_GPU_TRACE_SLOW_US=${TRACE_US:-5000} timeout 200 python bench.py --corpus clustered --steps 2 --warmup 1 --batches-in-flight 1 \
        --parity-queries 0 --scan-check-queries 0 --segment-regime 0 --ref-build-n 0 --cpu-queries 0 --recall-queries 8 \
        --single-query-calls 2048 --graph-cache /tmp/gc > gpurun_out/sq/run$i.json 2> gpurun_out/sq/run$i.err
    python - $i <<'P'
import json, sys
d = json.load(open("gpurun_out/sq/run%s.json" % sys.argv[1]))["config"]["single_query"]
print(sys.argv[1], {k: {a: round(b, 3) for a, b in v.items()} for k, v in d.items()})
P
    grep "slow search" gpurun_out/sq/run$i.err | head -5
done
