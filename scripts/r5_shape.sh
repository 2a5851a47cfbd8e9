#!/bin/bash
# launch shape of hnsw_search_kernel on the 10 M clustered shard (side blocks off): register class x visited-table size
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out/r5shape
ARGS="--corpus clustered --parity-queries 64 --scan-check-queries 0 --segment-regime 0 --bf16-block-n 0 --ref-build-n 0 --single-query-calls 0 --cpu-queries 0 --bm25-block 0 --iso-recall 0 --graph-cache /tmp/nidx_graphs"
for v in ${@:-"" "min_waves=5,vis_log2=12" "min_waves=5" "vis_log2=12" "min_waves=6,vis_log2=12"}; do
  name=$(echo "x$v" | tr ',=' '__')
  env NIDX_BENCH_TUNABLES="$v" timeout 400 python bench.py $ARGS > gpurun_out/r5shape/$name.json 2> gpurun_out/r5shape/$name.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r5shape/$name.json")); c=d["config"]; r=d["roofline"]
    print("tunables='$v' value=%.4g ms_per_step=%.4f kernel_ms=%.4f frac=%.3f sustained=%.3f recall=%.4f host=%.4g parity=%s" % (d["value"], d["ms_per_step"], r["kernel_ms"], r["frac"], r["sustained"]["frac"], c["recall_at_10"], c.get("host_buffer_queries_per_s") or 0, (c.get("parity") or {}).get("hnsw_vs_oracle", {}).get("status")))
except Exception as e:
    print("tunables='$v' failed:", e)
PY
done
