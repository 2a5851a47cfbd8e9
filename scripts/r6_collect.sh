#!/bin/bash
# gpurun_out/final6 (scripts/r6_profiles.sh) -> profiles/r06_*
set -u
cd "$(dirname "$0")/.."
F=gpurun_out/final6; P=profiles
cp $F/bench_default.json $P/r06_bench_default.json
cp $F/bench_bm25.json $P/r06_bench_bm25.json
cp $F/bench_bm25_one_at_a_time.json $P/r06_bench_bm25_one_at_a_time.json
cat $F/kernel_stats_bm25_one_at_a_time.txt $F/kernel_stats_bm25_one_at_a_time_two_launches.txt $F/kernel_stats_bm25_pipelined.txt > $P/r06_kernel_stats_bm25.txt
cat $F/pmc_bm25_FETCH_SIZE.txt $F/pmc_bm25_WRITE_SIZE.txt > $P/r06_pmc_bm25.txt
cp $F/bm25_batch_curve.txt $P/r06_bm25_batch_curve.txt
cp $F/bench_rabitq_1m.json $P/r06_bench_rabitq_1m.json
cp $F/kernel_stats_rabitq_1m.txt $P/r06_kernel_stats_rabitq_1m.txt
cp $F/bench_force_exchange_12m5.json $P/r06_bench_force_exchange_12m5.json
cp $F/bench_pipeline_12m5.json $P/r06_bench_pipeline_12m5.json
cp $F/bench_hnsw10m_crowded_shape.json $P/r06_bench_hnsw10m_crowded_shape.json
cp $F/kernel_stats_hnsw10m_crowded_shape.txt $P/r06_kernel_stats_hnsw10m_crowded_shape.txt
cat $F/pmc_hnsw10m_crowded_shape_FETCH_SIZE.txt $F/pmc_hnsw10m_crowded_shape_WRITE_SIZE.txt > $P/r06_pmc_hnsw10m_crowded_shape.txt
[ -f $F/bench_bf16_12m5x1024.json ] && cp $F/bench_bf16_12m5x1024.json $P/r06_bench_bf16_12m5x1024.json && cp $F/kernel_stats_bf16_12m5x1024.txt $P/r06_kernel_stats_bf16_12m5x1024.txt
ls $P | grep r06
