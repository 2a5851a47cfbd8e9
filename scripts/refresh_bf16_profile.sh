#!/bin/bash
# bench line + kernel trace of the bf16 fallback on the per-GPU shard of BASELINE configs[4] (12.5M x 1024, batch 1024)
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/bf16
mkdir -p $OUT
python $ROOT/bench.py --workload bf16 --n-vectors 12500000 --dim 1024 --clustered-n 0 --steps 10 --recall-queries 256 --cpu-queries 64 > $OUT/bench_bf16_12m5x1024.json 2> $OUT/bench.err
tail -1 $OUT/bench_bf16_12m5x1024.json | cut -c1-600
python $ROOT/bench.py --workload bf16 --clustered-n 0 --steps 10 --recall-queries 256 --cpu-queries 0 > $OUT/bench_bf16_1m.json 2>> $OUT/bench.err
tail -1 $OUT/bench_bf16_1m.json | cut -c1-300
rocprofv3 --kernel-trace --stats -d $OUT/trace -- python $ROOT/bench.py --workload bf16 --n-vectors 12500000 --dim 1024 --clustered-n 0 --steps 5 --recall-queries 0 --cpu-queries 0 > /dev/null 2>&1
python3 $ROOT/scripts/prof_summary.py $(ls $OUT/trace/*/*.db | head -1) > $OUT/bf16_kernel_stats.txt 2>&1
head -12 $OUT/bf16_kernel_stats.txt
bash $ROOT/scripts/pmc_bf16.sh > $OUT/pmc_bf16.txt 2>&1
cat $OUT/pmc_bf16.txt
