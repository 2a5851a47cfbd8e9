#!/bin/bash
# gpurun_out/final (scripts/r5_profiles.sh) -> profiles/r05_*
set -u
cd "$(dirname "$0")/.."
F=gpurun_out/final; P=profiles
cp $F/bench_default.json $P/r05_bench_default.json
cp $F/bench_clustered_profiled.json $P/r05_bench_hnsw10m_clustered_profiled.json
cp $F/bench_clustered_profiled_in_flight.json $P/r05_bench_hnsw10m_clustered_profiled_in_flight.json
cp $F/bench_uniform_profiled.json $P/r05_bench_hnsw10m_uniform_profiled.json
cat $F/summary_trace_clustered.txt $F/summary_trace_in_flight_clustered.txt > $P/r05_kernel_stats_hnsw10m_clustered.txt
cp $F/summary_trace_uniform.txt $P/r05_kernel_stats_hnsw10m_uniform.txt
cat $F/summary_fetch_clustered.txt $F/summary_write_clustered.txt > $P/r05_pmc_hnsw10m_clustered.txt
cat $F/summary_fetch_uniform.txt $F/summary_write_uniform.txt > $P/r05_pmc_hnsw10m_uniform.txt
cp $F/bench_bm25.json $P/r05_bench_bm25.json
cp $F/bench_bm25_one_at_a_time.json $P/r05_bench_bm25_one_at_a_time.json
cat $F/kernel_stats_bm25_one_at_a_time.txt $F/kernel_stats_bm25_pipelined.txt > $P/r05_kernel_stats_bm25.txt
cp $F/bm25_batch_curve.txt $P/r05_bm25_batch_curve.txt
cat gpurun_out/pmc_bm25/traffic_FETCH_SIZE.txt gpurun_out/pmc_bm25/traffic_WRITE_SIZE.txt > $P/r05_pmc_bm25.txt 2>/dev/null
cp $F/bench_hybrid.json $P/r05_bench_hybrid.json
cp $F/bench_hybrid_default_shape.json $P/r05_bench_hybrid_default_shape.json
cat $F/kernel_stats_hybrid.txt $F/kernel_stats_hybrid_default_shape.txt > $P/r05_kernel_stats_hybrid.txt
cp $F/bench_rabitq_1m.json $P/r05_bench_rabitq_1m.json
cp $F/bench_rabitq_1m_two_waves.json $P/r05_bench_rabitq_1m_two_waves.json
cat $F/kernel_stats_rabitq_1m.txt $F/kernel_stats_rabitq_1m_two_waves.txt > $P/r05_kernel_stats_rabitq_1m.txt
cat $F/pmc_rabitq_1m_FETCH_SIZE.txt $F/pmc_rabitq_1m_WRITE_SIZE.txt > $P/r05_pmc_rabitq_1m.txt
cp $F/bench_hnsw1m.json $P/r05_bench_hnsw1m.json
cp $F/kernel_stats_hnsw1m.txt $P/r05_kernel_stats_hnsw1m.txt
for n in scan_1m mfma_1m_k10 mfma_1m_k64 bf16_12m5x1024; do cp $F/bench_$n.json $P/r05_bench_$n.json; cp $F/kernel_stats_$n.txt $P/r05_kernel_stats_$n.txt; done
python scripts/make_pmc_traffic.py > /dev/null
ls $P | grep r05 | wc -l
