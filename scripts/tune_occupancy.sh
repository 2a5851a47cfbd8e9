#!/bin/bash
# Walks resident per CU vs throughput: hnsw_search_kernel register class (NIDX_GPU_MIN_WAVES), visited-table size
# (NIDX_GPU_VIS_LOG2) and batches in flight on the 10 M clustered corpus.  Run on the GPU box; writes gpurun_out/occ/.
mkdir -p gpurun_out/occ
ONLY="$*"
run() {  # name, batches in flight, env...
    local name=$1 f=$2; shift 2
    if [ -n "$ONLY" ] && ! echo " $ONLY " | grep -q " $name "; then return; fi
    env "$@" timeout 240 python bench.py --corpus clustered --steps 40 --warmup 5 --batches-in-flight $f \
        --parity-queries 0 --scan-check-queries 0 --segment-regime 0 --ref-build-n 0 --single-query-calls 0 --cpu-queries 0 \
        --recall-queries 64 --graph-cache /tmp/gc > gpurun_out/occ/$name.json 2> gpurun_out/occ/$name.err
    python - "$name" <<'P'
import json, sys
try:
    d = json.load(open("gpurun_out/occ/%s.json" % sys.argv[1]))
    r, c = d["roofline"], d["config"]
    print("%-16s value %.0f ms/step %.4f kernel_ms %.4f frac %.3f sustained %.3f recall %.3f evals %.0f edge_hits %.1f flags %s/%s" % (
        sys.argv[1], d["value"], d["ms_per_step"], r["kernel_ms"], r["frac"], r["sustained"]["frac"], c["recall_at_10"],
        c["distance_evals_per_query"], c.get("expansions_with_edge_record_fetched_ahead_per_query", -1), c["kernel_flags"], c["timed_launch_flags"]))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
P
}
run base_f3 3 A=1
run vis12_f3 3 NIDX_GPU_VIS_LOG2=12
run w5_vis12_f3 3 NIDX_GPU_MIN_WAVES=5 NIDX_GPU_VIS_LOG2=12
run w5_vis12_f4 4 NIDX_GPU_MIN_WAVES=5 NIDX_GPU_VIS_LOG2=12
run w6_vis12_f3 3 NIDX_GPU_MIN_WAVES=6 NIDX_GPU_VIS_LOG2=12
run w6_vis12_f5 5 NIDX_GPU_MIN_WAVES=6 NIDX_GPU_VIS_LOG2=12
run base_f2 2 A=1
run base_f4 4 A=1
run rows2_f3 3 NIDX_GPU_EVAL_ROWS=2
run rows3_f3 3 NIDX_GPU_EVAL_ROWS=3
run wpq2_f3 3 NIDX_GPU_WAVES_PER_QUERY=2
run wpq2_f6 6 NIDX_GPU_WAVES_PER_QUERY=2
