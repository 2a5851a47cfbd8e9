#!/bin/bash
# Every bench line + profile quoted in DESIGN.md, re-measured in one go on the GPU box (writes gpurun_out/final/).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/final
mkdir -p $OUT
cd $ROOT
bash scripts/refresh_profiles.sh > $OUT/refresh_profiles.log 2>&1
python bench.py --n-vectors 10000000 > $OUT/bench_hnsw_10m.json 2> $OUT/bench_hnsw_10m.err
python bench.py --workload hybrid --n-vectors 10000000 --cpu-queries 0 > $OUT/bench_hybrid_10m.json 2> $OUT/hybrid.err
python bench.py --workload scan --cpu-queries 0 --clustered-n 0 --steps 5 --recall-queries 0 > $OUT/bench_scan_1m.json 2>/dev/null
python bench.py --workload rabitq > $OUT/bench_rabitq_1m.json 2> $OUT/rabitq.err
python bench.py --workload bf16 --n-vectors 12500000 --dim 1024 --clustered-n 0 --steps 10 --recall-queries 256 --cpu-queries 64 > $OUT/bench_bf16_12m5x1024.json 2> $OUT/bf16.err
python bench.py --workload bf16 --clustered-n 0 --steps 10 --recall-queries 256 --cpu-queries 0 > $OUT/bench_bf16_1m.json 2>> $OUT/bf16.err
python bench.py --workload bm25 > $OUT/bench_bm25_10m.json 2> $OUT/bm25.err
for f in $OUT/bench_*.json; do echo "$(basename $f): $(tail -1 $f | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], (d.get("roofline") or {}).get("frac"))')"; done
grep -h "hnsw_search_kernel" $OUT/summary_*.txt | cut -c1-160
