#!/bin/bash
# Round 5 A/B runs on the GPU box: one line per variant under gpurun_out/r5ab/.  Usage: bash scripts/r5_ab.sh [rabitq|bm25|hybrid|all]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
PART=${1:-all}
OUT=$ROOT/gpurun_out/r5ab
mkdir -p $OUT
cd $ROOT
run() {  # name, env..., -- bench args
  local name=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  env "${envs[@]}" timeout 400 python bench.py "$@" > $OUT/$name.json 2> $OUT/$name.err < /dev/null
  echo "$name rc=$? $(python - <<PY
import json
try:
    d = json.load(open("$OUT/$name.json"))
    c = d.get("config", {})
    r = d.get("roofline") or {}
    print("value=%.4g ms_per_step=%.4g frac=%s kernel_ms=%s pipelined=%s recall=%s one_thread=%s multi=%s" % (d["value"], d["ms_per_step"], r.get("frac"), r.get("kernel_ms"), c.get("pipelined_queries_per_s"),
          c.get("recall_at_10"), (c.get("one_submitting_thread") or {}).get("postings_per_s"), (c.get("multi_segment") or {}).get("value")))
except Exception as e:
    print("no line:", e)
PY
)"
}
if [ "$PART" = rabitq ] || [ "$PART" = all ]; then
  run rabitq_pipe -- --workload rabitq --n-vectors 1000000 --steps 5 --warmup 1 --cpu-queries 256
  run rabitq_no_seen NIDX_GPU_RABITQ_SEEN=0 -- --workload rabitq --n-vectors 1000000 --steps 5 --warmup 1 --cpu-queries 0
  run rabitq_plain NIDX_GPU_RABITQ_PIPE=0 -- --workload rabitq --n-vectors 1000000 --steps 5 --warmup 1 --cpu-queries 0
  run rabitq_2w_barrier NIDX_GPU_RABITQ_WAVES=2 -- --workload rabitq --n-vectors 1000000 --steps 5 --warmup 1 --cpu-queries 0
fi
if [ "$PART" = bm25 ] || [ "$PART" = all ]; then
  run bm25_default -- --workload bm25 --cpu-queries 0 --steps 200
  run bm25_prio0 NIDX_GPU_BM25_PRIORITY=0 NIDX_BENCH_BM25_SEGMENTS=0 -- --workload bm25 --cpu-queries 0 --steps 200
  run bm25_copy_out NIDX_GPU_BM25_ZERO_COPY_OUT=0 NIDX_BENCH_BM25_SEGMENTS=0 -- --workload bm25 --cpu-queries 0 --steps 200
  run bm25_depth4 NIDX_BENCH_BM25_DEPTH=4 NIDX_BENCH_BM25_SEGMENTS=0 -- --workload bm25 --cpu-queries 0 --steps 200
fi
if [ "$PART" = bm25t ]; then
  for t in ${BM25_T:-2 3 4}; do
    run bm25_threads$t NIDX_BENCH_BM25_THREADS=$t NIDX_BENCH_BM25_SEGMENTS=0 -- --workload bm25 --cpu-queries 0 --steps 200
  done
  run bm25_threads4_depth1 NIDX_BENCH_BM25_THREADS=4 NIDX_BENCH_BM25_DEPTH=1 NIDX_BENCH_BM25_SEGMENTS=0 -- --workload bm25 --cpu-queries 0 --steps 200
  run bm25_batch2048 NIDX_BENCH_BM25_SEGMENTS=0 -- --workload bm25 --batch 2048 --cpu-queries 0 --steps 200
fi
if [ "$PART" = hybrid ]; then
  run hybrid_default -- --workload hybrid --steps 2000 --warmup 20 --cpu-queries 0
  run hybrid_w5_v12 NIDX_BENCH_TUNABLES=min_waves=5,vis_log2=12 -- --workload hybrid --steps 2000 --warmup 20 --cpu-queries 0
  run hybrid_prio0 NIDX_GPU_BM25_PRIORITY=0 -- --workload hybrid --steps 2000 --warmup 20 --cpu-queries 0
  run hybrid_w5_v13_prio0 NIDX_GPU_BM25_PRIORITY=0 NIDX_BENCH_TUNABLES=min_waves=5 -- --workload hybrid --steps 2000 --warmup 20 --cpu-queries 0
  run hybrid_w5_v12_prio0 NIDX_GPU_BM25_PRIORITY=0 NIDX_BENCH_TUNABLES=min_waves=5,vis_log2=12 -- --workload hybrid --steps 2000 --warmup 20 --cpu-queries 0
fi
if [ "$PART" = rqbatch ]; then
  for b in 128 256 512 1024 2048; do
    run rabitq_b$b -- --workload rabitq --n-vectors 1000000 --batch $b --steps 5 --warmup 1 --cpu-queries 0
  done
fi
if [ "$PART" = bm25slice ]; then
  for sl in 2048 2560 3072 4096 6144; do
    run bm25_slice$sl NIDX_GPU_BM25_SLICE=$sl NIDX_BENCH_BM25_SEGMENTS=0 -- --workload bm25 --cpu-queries 0 --steps 200
  done
fi
if [ "$PART" = rqseen ]; then
  for v in 0 9 10 11 12; do
    run rabitq_seen$v NIDX_GPU_RABITQ_SEEN=$v -- --workload rabitq --n-vectors 1000000 --steps 5 --warmup 1 --cpu-queries 0
  done
fi
if [ "$PART" = rqflight ]; then
  for v in 0 9 10; do for f in 4 6; do
    run rabitq_seen${v}_fl$f NIDX_GPU_RABITQ_SEEN=$v -- --workload rabitq --n-vectors 1000000 --steps 5 --warmup 1 --cpu-queries 0 --batches-in-flight $f
  done; done
fi
if [ "$PART" = rqlist ]; then
  for v in 8 9 10; do for f in 3 6; do
    run rabitq_seen${v}_fl$f NIDX_GPU_RABITQ_SEEN=$v -- --workload rabitq --n-vectors 1000000 --steps 5 --warmup 1 --cpu-queries 0 --batches-in-flight $f --rabitq-segments 0
  done; done
fi
