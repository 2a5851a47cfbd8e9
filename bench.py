#!/usr/bin/env python3
"""bench.py — the nidx vector hot path on MI355X: batched 768-d cosine HNSW k-NN over 10 M vectors
(the configuration BASELINE.json's metric is quoted on: configs[2]'s vector half; configs[3] = 12.5 M per GPU with --gpus 8).

One process per GPU.  Every rank owns one index shard (n_vectors x dim, synthetic, resident in HBM,
HNSW graph built on the device before the timed region), receives the full query batch, searches
its shard with the hand-written HIP kernel through the C ABI (device pointers, torch's current
stream), and — when world_size > 1 — all-gathers the per-shard top-k over RCCL and merges them
with merge_vector_responses' rule on the device.  A "step" = one batch of `--batch` queries.

The timed corpus is the reference's clustered recall recipe (nidx_vector/src/segment.rs:841-905) scaled to the shard
size, so recall@10 is measured on the corpus the throughput is quoted on; the uniform-random corpus of the reference's
other tests is a second, labelled figure (config.uniform_corpus).

Prints ONE JSON line on rank 0 (see the driver contract in the task statement), extended with
  roofline      dominant kernel (hnsw_search_kernel): algorithmic bytes per launch / HIP-event time vs 8 TB/s
  cpu_baseline  the CPU oracle (restated reference algorithm) on this box's host cores, bounded sample
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Batches in flight live on separate HIP streams; with the runtime's default of 4 hardware queues two of them can land on one
# queue and serialise (measured: 3 in flight = 2.2 M queries/s with 4 queues, 3.5 M with 8).  libnidx_gpu.so sets GPU_MAX_HW_QUEUES=8
# itself when it is loaded (csrc/serving.cpp) — main() loads it before torch touches HIP; the line below only covers a fresh
# checkout, where the library is built after HIP has been initialised.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FAILURES = []          # parity / consistency breaks found on the way: the line is still printed, the exit status is 1
T_PROCESS_START = time.time()   # the headline line carries the wall-clock time of the whole run (`bench_wall_s`)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--n-vectors", type=int, default=0,
                   help="vectors per shard (per GPU); 0 = BASELINE.json: 10 M on one GPU (configs[2]), 12.5 M per GPU otherwise (configs[3]: 100 M over 8)")
    p.add_argument("--corpus", choices=["clustered", "uniform", "both"], default="both",
                   help="hnsw workload: the timed corpus is the clustered one; 'both' adds the uniform corpus as a second figure")
    p.add_argument("--parity-queries", type=int, default=256, help="hnsw: queries of the timed batch checked bit for bit against the oracle (0 = skip)")
    p.add_argument("--scan-check-queries", type=int, default=4, help="hnsw: queries whose exact-scan ground truth is checked against the oracle's brute force")
    p.add_argument("--segment-regime", type=int, default=200_000,
                   help="cpu_baseline: also time the oracle over segments of this many records + Fssc (the reference's own regime, src/settings.rs:258-278); 0 = skip")
    p.add_argument("--bf16-block-n", type=int, default=12_500_000,
                   help="hnsw (one GPU): also time the bf16 matrix-core fallback scan (BASELINE.json configs[4]) on a uniform shard of this many "
                        "x 1024-dim vectors after the headline legs and put the block into the line (0 = skip)")
    p.add_argument("--ref-build-n", type=int, default=50_000,
                   help="recall of the oracle's sequential HnswBuilder vs the device build on a clustered segment of this size (0 = skip)")
    p.add_argument("--rabitq-segments", type=int, default=5,
                   help="--workload rabitq: also search the same vectors as this many quantized segments through the one-launch grid (0 / 1 = skip)")
    p.add_argument("--batches-in-flight", type=int, default=3,
                   help="hnsw: consecutive batches are launched on this many streams in turn, so the walk-length tail of one batch (a launch "
                        "lasts as long as its longest walk) overlaps the body of the next; 1 = strictly one launch at a time")
    p.add_argument("--graph-cache", default="",
                   help="hnsw: directory; the device-built hnsw.graph of each corpus is written there and, when present, loaded instead "
                        "of rebuilt (profiling runs: rocprofv3 --pmc crashes over the thousands of dispatches of a 10 M build)")
    p.add_argument("--single-query-calls", type=int, default=2048, help="hnsw: nidx_gpu_vector_search_one calls for the p50/p99 figure (0 = skip)")
    p.add_argument("--min-timed-s", type=float, default=1.0,
                   help="the timed pass of --steps steps is repeated until the timed region is at least this long (ms_per_step = mean over all)")
    p.add_argument("--build-ef-upper", type=int, default=4,
                   help="hnsw: tunable build_ef_upper of the timed flat graph: results kept per layer while an insertion descends to its "
                        "node's top layer (0 = 1 = the reference's greedy descent, hnsw/build.rs:137-146).  A greedy descent into a graph of "
                        "millions of nodes ends in the wrong neighbourhood for a few percent of the insertions, and a node linked into the "
                        "wrong neighbourhood cannot be found by any search width (scripts/diag_orphans.py); the reference never builds such "
                        "graphs (segments stay <= 200 k records).  Searches keep the reference's constants either way.")
    p.add_argument("--ef-upper", type=int, default=4,
                   help="hnsw: tunable ef_upper of the timed searches: results kept per layer of the descent (1 = the reference's greedy "
                        "descent, hnsw/search.rs:318-324).  With a greedy descent a few percent of the queries end in another neighbourhood "
                        "of a 10 M-node layer 0 and return nothing useful; 4 costs no measurable time and removes them.  The oracle restates "
                        "the knob, so parity is checked at this setting; the reference-constants figures are reported beside it.")
    p.add_argument("--iso-target", type=float, default=0.0, help="hnsw: recall target of the iso-recall ladder when the segment-regime leg is off")
    p.add_argument("--iso-recall", type=int, default=1, help="hnsw: also time the flat graph at the ef_search whose recall reaches the reference regime's (0 = skip)")
    p.add_argument("--bm25-block", type=int, default=1, help="hnsw: add the BM25 and hybrid blocks of BASELINE.json configs[2] to the default line (0 = skip)")
    p.add_argument("--dim", type=int, default=768)
    p.add_argument("--batch", type=int, default=1024)
    p.add_argument("--k", type=int, default=10)
    p.add_argument("--workload", choices=["hnsw", "scan", "mfma", "bf16", "bm25", "rabitq", "hybrid", "gather"], default="hnsw")
    p.add_argument("--n-docs", type=int, default=10_000_000, help="bm25: documents per shard")
    p.add_argument("--vocab", type=int, default=1_000_000)
    p.add_argument("--recall-queries", type=int, default=256)
    p.add_argument("--cpu-queries", type=int, default=4096, help="bounded sample for the cpu_baseline leg (0 = skip)")
    p.add_argument("--cpu-threads", type=int, default=0)
    p.add_argument("--waves-per-query", type=int, default=0, help="tuning: workgroup waves per query (env NIDX_GPU_WAVES_PER_QUERY)")
    return p.parse_args()


def launch_ranks(a):
    """`python bench.py --gpus N` without a launcher (no WORLD_SIZE in the environment): this process becomes the launcher — N children
    with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, one per GPU, rank 0 prints the one JSON line (its stdout is
    this process's).  Fewer than N GPUs is an error, not a silent 1-GPU run — unless NIDX_BENCH_SAME_DEVICE=1 (validation on a 1-GPU
    box): every rank then shares GPU 0 and the library's exchange runs over its shared-memory transport."""
    import socket
    import subprocess

    n = a.gpus
    same = os.environ.get("NIDX_BENCH_SAME_DEVICE") == "1"
    have = torch.cuda.device_count()
    if have < n and not same:
        print("ERROR: --gpus %d but this node has %d GPU(s) visible (NIDX_BENCH_SAME_DEVICE=1 shares GPU 0 between the ranks: validation only)" % (n, have),
              file=sys.stderr)
        sys.exit(2)
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc, left = 0, list(procs)
    while left:
        time.sleep(0.2)
        for p_ in list(left):
            code = p_.poll()
            if code is None:
                continue
            left.remove(p_)
            if code != 0 and rc == 0:
                rc = code
                for q_ in left:   # one rank failed: the others would wait for it in a collective for ever
                    q_.terminate()
    sys.exit(rc)


def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        launch_ranks(a)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        # a launcher's WORLD_SIZE is the truth about the job; a flag that disagrees with it is a mistake worth a line on stderr
        print("WARNING: --gpus %d but WORLD_SIZE=%d: running as %d rank(s)" % (a.gpus, world, world), file=sys.stderr)
        a.gpus = world
    if a.waves_per_query:
        os.environ["NIDX_GPU_WAVES_PER_QUERY"] = str(a.waves_per_query)
    # NIDX_BENCH_SAME_DEVICE=1 (validation only): every rank uses GPU 0; torch.distributed (control traffic: query broadcast, barriers)
    # goes over gloo and the DATA-PATH exchange over the library's shared-memory transport, so the N>1 code path — the library's own
    # exchange included — can be exercised on a single-GPU box.  The driver never sets it.
    same_device = os.environ.get("NIDX_BENCH_SAME_DEVICE") == "1"
    if same_device:
        local_rank = 0
    from nucliadb_amd import _lib as _early

    if os.path.exists(_early.LIB_PATH):
        _early.lib()   # before the first HIP call of the process: the library's load-time defaults (hardware queues) apply
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist

        # (rank 0 alone runs the CPU-side legs after the timed region — oracle parity, cpu_baseline — while the others wait in the last
        # barrier: a generous timeout, so that a slow host never turns a measured run into a watchdog abort)
        from datetime import timedelta

        if same_device:
            dist.init_process_group("gloo", timeout=timedelta(minutes=60))
        else:
            dist.init_process_group("nccl", device_id=dev, timeout=timedelta(minutes=60))

    from nucliadb_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):  # fresh checkout: the .so is a build product (git-ignored)
        if local_rank == 0:
            import __graft_entry__ as g

            g.build()
        if world > 1:
            dist.barrier()
    L = _lib.lib()
    _lib.check(L.nidx_gpu_set_device(local_rank))
    if a.n_vectors <= 0:
        a.n_vectors = 10_000_000 if world == 1 else 12_500_000
    if a.workload == "hnsw":
        bench_hnsw(a, L, dev, rank, world)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    if a.workload == "gather":
        bench_gather(a, L, dev, rank)
        return
    if a.workload == "bm25":
        bench_bm25(a, L, dev, rank, world)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    if a.workload == "hybrid":
        bench_hybrid(a, L, dev, rank, world)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    if a.workload == "rabitq":
        bench_rabitq(a, L, dev, rank, world)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    n, d, B, k = a.n_vectors, a.dim, a.batch, a.k

    # ---- synthetic shard: the reference's generator (segment.rs:682-695), uniform(-1,1) then normalised
    g = torch.Generator(device=dev)
    g.manual_seed(1234567890 + rank)
    x = torch.rand((n, d), generator=g, device=dev, dtype=torch.float32) * 2 - 1
    x /= x.norm(dim=1, keepdim=True)
    x_host = x.cpu().numpy()
    del x
    torch.cuda.empty_cache()
    cfg = _lib.VectorConfigC(d, 1, 0, 0)
    cseg = _lib.VectorSegmentC(x_host.ctypes.data, d * 4, n, None, n, None, 0, 0, None, 0, None, None)
    h = C.c_void_p()
    t0 = time.time()
    _lib.check(L.nidx_gpu_vector_open(C.byref(cfg), C.byref(cseg), 1, C.byref(h)))
    open_s = time.time() - t0
    build_s = 0.0

    # ---- query batches (seed 2, identical on every rank)
    gq = torch.Generator(device=dev)
    gq.manual_seed(2)
    n_pool = 8
    qpool = torch.rand((n_pool, B, d), generator=gq, device=dev, dtype=torch.float32) * 2 - 1
    qpool /= qpool.norm(dim=2, keepdim=True)
    out_vec = torch.zeros((B, k), dtype=torch.int32, device=dev)
    out_score = torch.zeros((B, k), dtype=torch.float32, device=dev)
    out_count = torch.zeros((B,), dtype=torch.int32, device=dev)
    stats = torch.zeros((B, 8), dtype=torch.int32, device=dev)
    method = {"scan": _lib.METHOD_BRUTE_FORCE, "mfma": _lib.METHOD_BRUTE_FORCE_MFMA, "bf16": _lib.METHOD_BRUTE_FORCE_BF16}[a.workload]
    params = _lib.VectorSearchParamsC(k, -1.0, 1, method)
    stream = torch.cuda.current_stream().cuda_stream

    def search(qb, with_stats=False, m=None, out=None):
        p = params if m is None else _lib.VectorSearchParamsC(k, -1.0, 1, m)
        ov, osc, oc = out if out is not None else (out_vec, out_score, out_count)
        _lib.check(L.nidx_gpu_vector_segment_search_device(
            h, 0, qb.data_ptr(), B, C.byref(p), None, ov.data_ptr(), osc.data_ptr(), oc.data_ptr(),
            stats.data_ptr() if with_stats else None, stream))

    def exchange(out):
        # K10: all-gather of the per-shard top-k (12 B/hit) + merge_vector_responses on every rank
        from nucliadb_amd.shard_merge import exchange_and_merge_vector

        ov, osc, oc = out
        ids = (ov.to(torch.int64) & 0xFFFFFFFF) | (rank << 32)
        return exchange_and_merge_vector(osc, ids, oc, k)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # With more than one rank a step is search + exchange.  The exchange (three small all-gathers over xGMI + the merge kernel) is
    # latency-bound and needs none of the compute units, so it runs on a side stream from one of two result-buffer sets while
    # the main stream already searches the next batch into the other set: search i + 1 overlaps exchange i; a buffer set is
    # searched into again only after its exchange has finished.  NIDX_BENCH_FORCE_EXCHANGE=1 runs this path at world size 1.
    do_exchange = world > 1 or os.environ.get("NIDX_BENCH_FORCE_EXCHANGE") == "1"
    main_stream = torch.cuda.current_stream()
    side_stream = torch.cuda.Stream() if do_exchange else None
    out_sets = [(out_vec, out_score, out_count),
                (torch.zeros_like(out_vec), torch.zeros_like(out_score), torch.zeros_like(out_count))] if do_exchange else None
    ev_searched = [torch.cuda.Event(), torch.cuda.Event()]
    ev_exchanged = [torch.cuda.Event(), torch.cuda.Event()]
    last_merged = [None]

    # One launch lasts as long as its longest walk, and on clustered data one query in a thousand walks three times the median:
    # with a single stream the whole GPU waits for it.  Consecutive batches therefore go to `nfl` streams in turn (each with its
    # own result buffers): the workgroups of batch i + 1 take the slots batch i's finished walks free.  Every batch is still one
    # launch of B queries; per-launch durations (kernel_ms, the roofline's denominator) are measured on the launch's own stream.
    # (the second corpus is bandwidth-bound: overlapping its launches buys ~10 % and doubles every launch's duration, so it runs
    # one launch at a time and its per-launch figures read directly)
    nfl = 1   # (the exact-scan workloads are one launch at a time)
    fl_streams = [torch.cuda.Stream() for _ in range(nfl)] if nfl > 1 else None
    fl_out = [(torch.zeros_like(out_vec), torch.zeros_like(out_score), torch.zeros_like(out_count)) for _ in range(nfl)] if nfl > 1 else None

    def step(i, e0=None, e1=None):
        if not do_exchange:
            if nfl > 1:
                st_ = fl_streams[i % nfl]
                if e0 is not None:
                    e0.record(st_)
                search(qpool[i % n_pool], out=fl_out[i % nfl], on=st_.cuda_stream)
                if e1 is not None:
                    e1.record(st_)
                return
            if e0 is not None:
                e0.record()
            search(qpool[i % n_pool])
            if e1 is not None:
                e1.record()
            return
        b = i & 1
        main_stream.wait_event(ev_exchanged[b])   # (a no-op until the event has been recorded once)
        if e0 is not None:
            e0.record(main_stream)
        search(qpool[i % n_pool], out=out_sets[b])
        if e1 is not None:
            e1.record(main_stream)
        ev_searched[b].record(main_stream)
        with torch.cuda.stream(side_stream):
            side_stream.wait_event(ev_searched[b])
            last_merged[0] = exchange(out_sets[b])
            ev_exchanged[b].record(side_stream)

    for i in range(a.warmup):
        step(i)
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps)]
    barrier()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(a.warmup + i, ev0[i], ev1[i])
    barrier()
    elapsed = time.perf_counter() - t0
    exchange_check = None
    if do_exchange and a.steps > 0:
        # the overlapped pipeline must give what a plain search -> exchange of the same batch gives
        i_last = a.warmup + a.steps - 1
        torch.cuda.synchronize()
        got = [t.clone() for t in last_merged[0]]
        search(qpool[i_last % n_pool], out=out_sets[0])
        torch.cuda.synchronize()
        want = exchange(out_sets[0])
        torch.cuda.synchronize()
        exchange_check = "ok" if all(torch.equal(g_, w_) for g_, w_ in zip(got, want)) else "MISMATCH"
        if exchange_check != "ok":
            print("WARNING: overlapped exchange diverged from the sequential one", file=sys.stderr)
        barrier()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms = float(np.mean([ev0[i].elapsed_time(ev1[i]) for i in range(a.steps)]))

    alg_bytes = float(n) * d * 4  # the shard is read once per batch algorithmically
    evals_q, exp_q, flags = [float(n)], [0.0], 0
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
    alg_flops = 2.0 * n * d * B
    achieved_tf = alg_flops / (kernel_ms * 1e-3) / 1e12
    traffic, traffic_src, host_qps = None, None, None

    # ---- recall@k against the exact scan (oracle-verified kernel) on the same shard
    recall = None
    if a.workload == "bf16" and a.recall_queries > 0:
        rq = min(a.recall_queries, B)
        search(qpool[0])
        torch.cuda.synchronize()
        got = out_vec[:rq].cpu().numpy()
        search(qpool[0], m=_lib.METHOD_BRUTE_FORCE)
        torch.cuda.synchronize()
        exact = out_vec[:rq].cpu().numpy()
        recall = float(np.mean([len(set(got[i]) & set(exact[i])) / k for i in range(rq)]))

    # ---- CPU baseline: the oracle (restated reference algorithm, AVX2-shaped sums) on the host cores
    cpu = None
    if rank == 0 and a.cpu_queries > 0:
        try:
            cpu = cpu_baseline(a, L, h, x_host, qpool[0].cpu().numpy(), qpool[1].cpu().numpy())
        except Exception as e:  # the baseline is a reported side leg: its failure must not cost the measured line
            print("WARNING: cpu_baseline failed: %r" % (e,), file=sys.stderr)
            cpu = {"value": None, "unit": "queries/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}

    L.nidx_gpu_vector_close(h)
    if rank == 0:
        total_q = world * B * a.steps
        line = {
            "metric": "queries/sec (exact cosine k-NN, %s)" % a.workload,
            "value": total_q / elapsed,
            "unit": "queries/s (each against one %d-vector shard; %d shard(s) searched in parallel and merged)" % (n, world),
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "%s: %d x %d-dim cosine, k=%d, batch=%d queries, 1 shard per GPU" % (a.workload, n, d, k, B),
                "vectors_per_shard": n, "dim": d, "batch": B, "k": k, "shards": world,
                "corpus_vectors": n * world, "merged_queries_per_s": B * a.steps / elapsed,
                "recall_at_%d" % k: recall, "open_s": open_s,
                "distance_evals_per_query": float(np.mean(evals_q)), "expansions_per_query": float(np.mean(exp_q)),
                "kernel_flags": flags, "host_buffer_queries_per_s": host_qps, "parallelism": "shard-per-gpu x%d, RCCL all-gather of top-k" % world, "exchange_check": exchange_check,
            },
            "roofline": ({
                "kernel": "mfma_scan_kernel (+ merge_topk_kernel)", "bound": "mfma", "achieved": achieved_tf, "peak": 157.3,
                "unit": "TFLOP/s", "frac": achieved_tf / 157.3, "traffic": None, "algorithmic_flops_per_launch": alg_flops,
                "hbm_GBps_algorithmic": achieved, "kernel_ms": kernel_ms,
            } if a.workload == "mfma" else {
                "kernel": "bf16 fallback launches of one batch (sample passes, bf16_append_kernel / bf16_scan_kernel, merge_topk_kernel, rescore_select_kernel)", "bound": "mfma", "achieved": achieved_tf,
                "peak": 2500.0, "unit": "TFLOP/s", "frac": achieved_tf / 2500.0, "traffic": None,
                "algorithmic_flops_per_launch": alg_flops, "hbm_GBps_algorithmic_bf16": float(n) * d * 2 / (kernel_ms * 1e-3) / 1e9,
                "hbm_frac": float(n) * d * 2 / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "kernel_ms": kernel_ms,
            } if a.workload == "bf16" else {
                "kernel": "scan_shared_kernel / scan_topk_kernel (+ merge_topk_kernel)",
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": kernel_ms,
                # at batch >= ~64 the shared-row scan is bound by the f32 FMA pipe, not by HBM (every row is read once per launch and used by
                # all queries): the fraction that says how good the kernel is there
                "valu_f32": {"achieved": achieved_tf, "peak": 157.3, "unit": "TFLOP/s", "frac": achieved_tf / 157.3,
                             "note": "2 n d B flops per launch on the VALU (packed f32 FMA), against the 157.3 TFLOP/s f32 peak"},
            }),
            "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if world > 1:
        dist.barrier()  # rank 0 runs the CPU baseline; keep the group alive until it is done
        dist.destroy_process_group()


# =====================================================================================================================
# The headline workload: batched cosine HNSW k-NN over one shard per GPU
# =====================================================================================================================
PER_CLUSTER = 160  # segment.rs:849-863: 80 vectors at radius 0.01 + 80 at 0.03 around each centre


def _unit_rows(shape, g, dev):
    v = torch.rand(shape, generator=g, device=dev, dtype=torch.float32) * 2 - 1
    return v / v.norm(dim=-1, keepdim=True)


def clustered_centres(n, d, seed):
    """The chained cluster centres of the reference's recall recipe (segment.rs:849-865: `center =
    random_nearby_vector(center, 0.1)` after every cluster) — sequential by construction, so drawn on the host."""
    rng = np.random.default_rng(seed)

    def unit(*shape):
        v = rng.uniform(-1.0, 1.0, shape).astype(np.float32)
        return v / np.linalg.norm(v, axis=-1, keepdims=True)

    n_centres = (n + PER_CLUSTER - 1) // PER_CLUSTER
    steps = unit(n_centres, d)
    centres = np.empty((n_centres, d), np.float32)
    c = unit(d)
    for j in range(n_centres):
        centres[j] = c
        c = c + np.float32(0.1) * steps[j]
        c = c / np.linalg.norm(c)
    return centres


def gen_corpus(kind, n, d, dev, seed):
    """-> x [n][d] f32 unit rows on the device.
    uniform:   the generator of the reference's unit tests (segment.rs:682-695): uniform(-1, 1), normalised.
    clustered: the reference's recall recipe (segment.rs:841-905) scaled to n: chained centres 0.1 apart, 160 vectors per
               centre (half at radius 0.01, half at 0.03), rows in random order (the reference inserts in BTreeMap order of
               random keys)."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    x = torch.empty((n, d), device=dev, dtype=torch.float32)
    chunk = 1 << 20
    if kind == "uniform":
        for i0 in range(0, n, chunk):
            i1 = min(n, i0 + chunk)
            x[i0:i1] = _unit_rows((i1 - i0, d), g, dev)
        return x
    centres = torch.from_numpy(clustered_centres(n, d, seed)).to(dev)
    radius = torch.where(torch.arange(PER_CLUSTER, device=dev) < PER_CLUSTER // 2, 0.01, 0.03).to(torch.float32)
    perm = torch.randperm(n, generator=g, device=dev)  # row r of the shard is point perm[r] of the recipe
    for i0 in range(0, n, chunk):
        i1 = min(n, i0 + chunk)
        src = perm[i0:i1]
        rows = centres[src // PER_CLUSTER] + radius[src % PER_CLUSTER][:, None] * _unit_rows((i1 - i0, d), g, dev)
        x[i0:i1] = rows / rows.norm(dim=1, keepdim=True)
    return x


def gen_queries(kind, x, n_pool, B, d, dev, seed):
    """uniform: random unit vectors; clustered: `random_nearby_vector(stored, 0.05)` of random stored vectors (segment.rs:879-882)."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    if kind == "uniform":
        return _unit_rows((n_pool, B, d), g, dev)
    base = x[torch.randint(0, x.shape[0], (n_pool * B,), generator=g, device=dev)]
    q = base + 0.05 * _unit_rows((n_pool * B, d), g, dev)
    return (q / q.norm(dim=1, keepdim=True)).reshape(n_pool, B, d).contiguous()


def serialize_graph(L, h, seg=0):
    glen, nedges = C.c_uint64(0), C.c_uint64(0)
    from nucliadb_amd import _lib
    _lib.check(L.nidx_gpu_vector_serialize_hnsw(h, seg, None, 0, C.byref(glen), None, 0, C.byref(nedges)))
    graph = np.zeros(glen.value, np.uint8)
    edges = np.zeros(max(1, nedges.value), np.float32)
    _lib.check(L.nidx_gpu_vector_serialize_hnsw(h, seg, graph.ctypes.data, glen.value, C.byref(glen), edges.ctypes.data, nedges.value,
                                                C.byref(nedges)))
    return graph, edges[: nedges.value]


def pmc_traffic(kind, n, d, B, k):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes of this same command
    (profiles/r05_pmc_traffic.json — or a previous round's — written by scripts/refresh_profiles.sh / make_pmc_traffic.py; counters
    cannot be read from inside this process)."""
    for name in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                for e in json.load(f)["entries"]:
                    w = e["workload"]
                    if (w["corpus"], w["n_vectors"], w["dim"], w["batch"], w["k"]) == (kind, n, d, B, k):
                        return e["hbm_bytes_per_launch"], e["source"]
        except (OSError, KeyError, ValueError, TypeError):
            pass
    return None, None


def hnsw_leg(a, L, dev, rank, world, kind, headline):
    """One corpus: generate, open, build, time, count, check.  Returns the figures of this corpus (rank 0) or None."""
    from nucliadb_amd import _lib
    if world > 1:
        import torch.distributed as dist

    n, d, B, k = a.n_vectors, a.dim, a.batch, a.k
    n_pool = 8
    t0 = time.time()
    x = gen_corpus(kind, n, d, dev, 1234567890 + rank)
    qpool = gen_queries(kind, x, n_pool, B, d, dev, 2)
    if world > 1:
        dist.broadcast(qpool, src=0)  # every shard answers the same batch (the clustered queries sit near rank 0's vectors)
    torch.cuda.synchronize()
    gen_s = time.time() - t0
    cfg = _lib.VectorConfigC(d, 1, 0, 0)
    gpath = os.path.join(a.graph_cache, "hnsw_%s_%d_%d_r%d_u%d.graph" % (kind, n, d, rank, a.build_ef_upper)) if a.graph_cache else ""
    cached = None
    if gpath and os.path.exists(gpath):
        cached = np.fromfile(gpath, dtype=np.uint8)
    cseg = _lib.VectorSegmentC(x.data_ptr(), d * 4, n, None, n, cached.ctypes.data if cached is not None else None,
                               cached.size if cached is not None else 0, 0, None, 0, None, None)
    h = C.c_void_p()
    t0 = time.time()
    _lib.check(L.nidx_gpu_vector_open(C.byref(cfg), C.byref(cseg), 1, C.byref(h)))  # packed device matrix: copied device to device
    open_s = time.time() - t0
    del cached
    # the host copy feeds the oracle legs (rank 0 of the headline corpus only): the product never reads it
    need_host = rank == 0 and headline and (a.parity_queries > 0 or a.cpu_queries > 0 or a.scan_check_queries > 0 or a.ref_build_n > 0)
    x_host = x.cpu().numpy() if need_host else None
    del x
    torch.cuda.empty_cache()
    t0 = time.time()
    cached_used = bool(gpath and os.path.exists(gpath))
    if not cached_used:
        if a.build_ef_upper > 1:
            _lib.check(L.nidx_gpu_vector_set_tunable(h, b"build_ef_upper", a.build_ef_upper))
        _lib.check(L.nidx_gpu_vector_build_hnsw(h, 0, 2))
        if gpath:
            os.makedirs(a.graph_cache, exist_ok=True)
            g_, _e = serialize_graph(L, h)
            g_.tofile(gpath)
            del g_, _e
    build_s = time.time() - t0
    build_blk = None
    if not (gpath and cached_used):
        # the build on the roofline (north_star: "build + k-NN query distance kernels"; HnswBuilder, hnsw/build.rs:28-167): the build
        # kernels count their distance evaluations, expansions and the rows the neighbour-selection heuristic reads, like the search kernel
        bst = (C.c_uint64 * 10)()
        _lib.check(L.nidx_gpu_vector_build_stats(h, bst))
        if int(bst[2]) != 2**64 - 1 and int(bst[1]) > 0:
            secs = int(bst[1]) / 1e6
            rows = int(bst[2]) + int(bst[4]) + int(bst[5])
            alg = rows * 4.0 * d + int(bst[3]) * 256.0
            build_blk = {
                "nodes": int(bst[0]), "seconds_of_kernels": secs, "inserts_per_s": int(bst[0]) / secs,
                "search_distance_evals_per_insert": int(bst[2]) / max(1, int(bst[0])), "search_expansions_per_insert": int(bst[3]) / max(1, int(bst[0])),
                "select_rows_per_insert": int(bst[4]) / max(1, int(bst[0])), "prune_rows_per_insert": int(bst[5]) / max(1, int(bst[0])),
                "reverse_link_appends_per_insert": int(bst[6]) / max(1, int(bst[0])), "prunes_per_insert": int(bst[7]) / max(1, int(bst[0])),
                "roofline": {"kernel": "insert_search_kernel + select_link_kernel + reverse_link_kernel (whole build)", "bound": "hbm",
                             "algorithmic_bytes": alg, "achieved": alg / secs / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / secs / 1e9 / HBM_PEAK_GBS,
                             "definition": "(search evaluations + rows read by select_neighbours_heuristic and the reverse-link prunes) x 4 D + expansions x 256 B, "
                                           "counted by the kernels, over the time from the first batch's launch to the last one's completion"},
                "reference": "nidx_vector/src/hnsw/build.rs:57-166"}
    if a.ef_upper > 1:
        _lib.check(L.nidx_gpu_vector_set_tunable(h, b"ef_upper", a.ef_upper))
    for kv in filter(None, os.environ.get("NIDX_BENCH_TUNABLES", "").split(",")):   # A/B runs: name=value[,name=value] (launch-shape / measurement knobs)
        name, _, val = kv.partition("=")
        _lib.check(L.nidx_gpu_vector_set_tunable(h, name.strip().encode(), int(val)))

    out_vec = torch.zeros((B, k), dtype=torch.int32, device=dev)
    out_score = torch.zeros((B, k), dtype=torch.float32, device=dev)
    out_count = torch.zeros((B,), dtype=torch.int32, device=dev)
    stats = torch.zeros((B, 8), dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def search(qb, with_stats=False, m=_lib.METHOD_HNSW, out=None, nq=B, on=None):
        p = _lib.VectorSearchParamsC(k, -1.0, 1, m)
        ov, osc, oc = out if out is not None else (out_vec, out_score, out_count)
        _lib.check(L.nidx_gpu_vector_segment_search_device(
            h, 0, qb.data_ptr(), nq, C.byref(p), None, ov.data_ptr(), osc.data_ptr(), oc.data_ptr(),
            stats.data_ptr() if with_stats else None, on if on is not None else stream))

    def device_flags():
        f = C.c_uint32(0)
        _lib.check(L.nidx_gpu_vector_device_flags(h, stream, C.byref(f)))
        return int(f.value)

    comm = None
    same_device = os.environ.get("NIDX_BENCH_SAME_DEVICE") == "1"
    if world > 1:
        # the product's own exchange (csrc/shard_comm.cpp: RCCL inside the library — or, when the ranks share one GPU for validation,
        # the library's shared-memory transport: same pack / gather layout / merge kernels); torch.distributed only ships the 128-byte id
        from nucliadb_amd.shard_merge import ShardComm

        # (if the library's communicator cannot be set up on this node every rank falls back to the same exchange through
        # torch.distributed — the line then says so in config.timed_region.entry — instead of losing the run)
        ok = torch.ones(1, dtype=torch.int32, device=dev)
        idt = torch.zeros(_lib.SHARD_COMM_ID_BYTES, dtype=torch.uint8, device=dev)
        try:
            if rank == 0:
                idt.copy_(torch.frombuffer(bytearray(ShardComm.unique_id_shm() if same_device else ShardComm.unique_id()), dtype=torch.uint8))
        except Exception as e:
            print("WARNING: nidx_gpu_shard_comm_unique_id failed: %r" % (e,), file=sys.stderr)
            ok.zero_()
        dist.broadcast(ok, src=0)
        if int(ok.item()):
            dist.broadcast(idt, src=0)
            try:
                comm = ShardComm(bytes(idt.cpu().numpy().tobytes()), rank, world, b"shard-%02d" % rank)
            except Exception as e:
                print("WARNING: rank %d: nidx_gpu_shard_comm_init failed: %r" % (rank, e), file=sys.stderr)
                ok.zero_()
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if not int(ok.item()):
                comm = None

    def exchange_torch(out):
        # the same exchange through torch.distributed (cross-check of the product path)
        from nucliadb_amd.shard_merge import exchange_and_merge_vector

        ov, osc, oc = out
        ids = (ov.to(torch.int64) & 0xFFFFFFFF) | (rank << 32)
        return exchange_and_merge_vector(osc, ids, oc, k)

    def exchange(out, st=None):
        # K10: all-gather of the per-shard top-k (12 B/hit) + merge_vector_responses on every rank
        if comm is None:
            return exchange_torch(out)
        ov, osc, oc = out
        ids = (ov.to(torch.int64) & 0xFFFFFFFF) | (rank << 32)
        return comm.exchange_merge_vector(osc, ids, oc, k, stream=st)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # One launch lasts as long as its longest walk, and on clustered data one query in a thousand walks three times the median:
    # with a single batch at a time the whole GPU waits for it.  `nfl` batches are therefore kept in flight; every batch is still one
    # launch of B queries.
    #   N = 1: through the PRODUCT's serving pipeline — nidx_gpu_vector_search_submit (device-resident queries, the library's own
    #          streams) / nidx_gpu_vector_search_wait (hits in host arrays, flagged queries re-run exactly): the timed region is
    #          what a host gets from the C ABI, results delivered;
    #   N > 1: search into device buffers + the library's RCCL exchange (ShardComm) on a side stream, in step order on every rank,
    #          while the search streams already work on the next batches; a result-buffer set is searched into again only after
    #          its exchange has finished.  NIDX_BENCH_FORCE_EXCHANGE=1 runs this path at world size 1.
    # (the second corpus is bandwidth-bound: overlapping its launches buys ~10 % and doubles every launch's duration, so it runs
    # one launch at a time and its per-launch figures read directly)
    do_exchange = world > 1 or os.environ.get("NIDX_BENCH_FORCE_EXCHANGE") == "1"
    nfl = max(1, a.batches_in_flight) if headline else 1
    main_stream = torch.cuda.current_stream()
    streams = [torch.cuda.Stream() for _ in range(nfl)] if (nfl > 1 and do_exchange) else [main_stream]
    side_stream = torch.cuda.Stream() if do_exchange else None
    n_sets = nfl + 1 if do_exchange else 1
    out_sets = [(out_vec, out_score, out_count)] + [(torch.zeros_like(out_vec), torch.zeros_like(out_score), torch.zeros_like(out_count))
                                                    for _ in range(max(n_sets, 2) - 1)]
    n_sets = len(out_sets)
    ev_searched = [torch.cuda.Event() for _ in range(n_sets)]
    ev_exchanged = [torch.cuda.Event() for _ in range(n_sets)]
    last_merged = [None]
    p_hnsw = _lib.VectorSearchParamsC(k, -1.0, 1, _lib.METHOD_HNSW)
    host_out = [(np.zeros((B, k), np.uint32), np.zeros((B, k), np.float32), np.zeros(B, np.uint32)) for _ in range(max(nfl, 16))]
    nfl_now = [nfl]   # (the reference-constants leg below also times deeper pipelines)
    in_flight = []   # (ticket, host_out index)
    retried_total = [0]

    def wait_oldest():
        t, j = in_flight.pop(0)
        hv_, hs2_, hc_ = host_out[j]
        r_ = C.c_uint32(0)
        _lib.check(L.nidx_gpu_vector_search_wait(h, t, None, None, hv_.ctypes.data, hs2_.ctypes.data, hc_.ctypes.data, C.byref(r_)))
        retried_total[0] += r_.value
        return j

    def step(i):
        if not do_exchange:
            if len(in_flight) == nfl_now[0]:
                wait_oldest()
            t = C.c_uint64(0)
            _lib.check(L.nidx_gpu_vector_search_submit(h, qpool[i % n_pool].data_ptr(), B, d, C.byref(p_hnsw), None, C.byref(t)))
            in_flight.append((t.value, i % nfl_now[0]))
            return
        b = i % n_sets
        st_ = streams[i % len(streams)]
        tr0 = time.perf_counter()
        st_.wait_event(ev_exchanged[b])   # (a no-op until the event has been recorded once)
        search(qpool[i % n_pool], out=out_sets[b], on=st_.cuda_stream)
        ev_searched[b].record(st_)
        tr1 = time.perf_counter()
        with torch.cuda.stream(side_stream):
            side_stream.wait_event(ev_searched[b])
            last_merged[0] = exchange(out_sets[b], st=side_stream.cuda_stream)
            ev_exchanged[b].record(side_stream)
        if trace_steps:
            trace_acc[0] += tr1 - tr0
            trace_acc[1] += time.perf_counter() - tr1
            trace_acc[2] += 1

    def drain():
        while in_flight:
            wait_oldest()

    trace_steps = os.environ.get("NIDX_BENCH_TRACE_STEPS") == "1"
    trace_acc = [0.0, 0.0, 0]

    _lib.check(L.nidx_gpu_vector_set_tunable(h, b"pipeline_depth", max(nfl, 4)))
    _lib.check(L.nidx_gpu_vector_set_tunable(h, b"pipeline_walks", nfl))
    # the exchange path overlaps its searches on streams of its own (nidx_gpu_vector_segment_search_device): the library cannot see them, so
    # the bench says what the pipeline would have found out by itself (tunable "launch_shape"); back to automatic afterwards
    crowded_by_hand = do_exchange and len(streams) > 1 and not os.environ.get("NIDX_BENCH_TUNABLES")
    if crowded_by_hand:
        _lib.check(L.nidx_gpu_vector_set_tunable(h, b"launch_shape", 1))
    for i in range(a.warmup):
        step(i)
    drain()
    torch.cuda.synchronize()
    device_flags()  # clear: the word below covers exactly the timed launches
    retried_total[0] = 0
    # EXACTLY `--steps` steps are timed per pass; the pass is repeated until the timed region is at least `--min-timed-s` long (20
    # steps of this workload are 6 ms: too short to time honestly), and `ms_per_step` is the mean over every timed step
    repeats, elapsed = 1, 0.0
    if a.steps > 0:
        # the pass count is sized from a WARM probe: the first pass over the pool runs up to 2 x slower than the steady state (cold
        # TLBs / caches / clocks), and a count taken from it left the timed region at half the length asked for (round 4: 0.49 s)
        probe = 0.0
        for _ in range(3):
            barrier()
            t0 = time.perf_counter()
            for i in range(a.steps):
                step(a.warmup + i)
            drain()
            barrier()
            probe = time.perf_counter() - t0
        repeats = max(1, int(np.ceil(1.1 * a.min_timed_s / max(probe, 1e-6))))
        if world > 1:
            t = torch.tensor([repeats], dtype=torch.int64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            repeats = int(t.item())
    # (a 20-step probe is mostly pipeline fill and drain — it ran at half the steady rate and the region sized from it came out at 0.55 s:
    # the region is timed, and when it falls short of --min-timed-s it is sized again from its own rate and timed once more)
    for _attempt in range(3):
        barrier()
        t0 = time.perf_counter()
        for i in range(a.steps * repeats):
            step(a.warmup + i)
        drain()
        barrier()
        elapsed = time.perf_counter() - t0
        again = a.steps > 0 and elapsed < a.min_timed_s and _attempt < 2
        if world > 1:
            t = torch.tensor([1 if again else 0, repeats], dtype=torch.int64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            again = bool(int(t[0].item()))
        if not again:
            break
        repeats = max(repeats + 1, int(np.ceil(repeats * 1.15 * a.min_timed_s / max(elapsed, 1e-6))))
        if world > 1:
            t = torch.tensor([repeats], dtype=torch.int64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            repeats = int(t.item())
        trace_acc[0] = trace_acc[1] = 0.0
        trace_acc[2] = 0
        retried_total[0] = 0
        if do_exchange:
            device_flags()
    if trace_steps and trace_acc[2]:
        print("[bench trace] rank %d: %d exchange steps: host time in search launch %.3f ms, in exchange %.3f ms per step; elapsed %.3f ms per step"
              % (rank, trace_acc[2], trace_acc[0] / trace_acc[2] * 1e3, trace_acc[1] / trace_acc[2] * 1e3, elapsed / max(1, a.steps * repeats) * 1e3), file=sys.stderr)
    # every timed launch ORs the overflow flags of its queries into one word: wait() re-runs flagged queries exactly and counts them
    # (N = 1); the device-entry launches of the exchange path leave the word for the poll below
    timed_flags = device_flags() if do_exchange else 0
    timed_retried = retried_total[0]
    if crowded_by_hand:
        _lib.check(L.nidx_gpu_vector_set_tunable(h, b"launch_shape", 0))
    exchange_check = None
    if do_exchange and a.steps > 0:
        # the overlapped pipeline must give what a plain search -> exchange of the same batch gives — and the library's RCCL
        # exchange what the same exchange through torch.distributed gives
        i_last = a.warmup + a.steps * repeats - 1
        torch.cuda.synchronize()
        got = [t.clone() for t in last_merged[0]]
        search(qpool[i_last % n_pool], out=out_sets[0])
        torch.cuda.synchronize()
        want = exchange(out_sets[0])
        torch.cuda.synchronize()
        exchange_check = "ok" if all(torch.equal(g_, w_) for g_, w_ in zip(got, want)) else "MISMATCH"
        if comm is not None:
            ref = exchange_torch(out_sets[0])
            torch.cuda.synchronize()
            if not all(torch.equal(g_, w_) for g_, w_ in zip(want, ref)):
                exchange_check = "MISMATCH (library RCCL exchange vs torch.distributed)"
        if exchange_check != "ok":
            FAILURES.append("exchange_check: " + exchange_check)
            print("ERROR: " + exchange_check, file=sys.stderr)
        barrier()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    steps_timed = a.steps * repeats
    # per-launch duration: HIP events on the launch's own stream around launches issued one at a time (the overlapped launches of
    # the timed region also span their wait for workgroup slots, which no profiler counts as kernel time)
    ea = [torch.cuda.Event(enable_timing=True) for _ in range(2 * max(1, min(a.steps, 8)))]
    for i in range(len(ea) // 2):
        ea[2 * i].record()
        search(qpool[i % n_pool])
        ea[2 * i + 1].record()
    torch.cuda.synchronize()
    kernel_ms = float(np.mean([ea[2 * i].elapsed_time(ea[2 * i + 1]) for i in range(len(ea) // 2)]))
    alone_ms = kernel_ms
    # ---- algorithmic bytes per launch: evals*4D + expansions*256 B, counted by the kernel
    bytes_per_launch, evals_q, exp_q, flags = [], [], [], 0
    hits_q = []
    for i in range(min(n_pool, max(1, a.steps))):
        search(qpool[i], with_stats=True)
        torch.cuda.synchronize()
        st = stats.cpu().numpy().astype(np.int64)
        bytes_per_launch.append(float((st[:, 0] * 4 * d + st[:, 1] * 256).sum()))
        evals_q.append(float(st[:, 0].mean()))
        exp_q.append(float(st[:, 1].mean()))
        hits_q.append(float(st[:, 5].mean()))   # NIDX_STAT_EDGE_HITS
        flags |= int(np.bitwise_or.reduce(st[:, 3]))
    alg_bytes = float(np.mean(bytes_per_launch))
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
    traffic, traffic_src = pmc_traffic(kind, n, d, B, k)

    # ---- recall@k of the timed configuration against the exact scan of the same shard (merged over the shards when N > 1)
    recall, got0, exact0, recall_hist = None, None, None, None
    if a.recall_queries > 0:
        rq = min(a.recall_queries, B)
        search(qpool[0])
        torch.cuda.synchronize()
        got0 = (out_vec.cpu().numpy().copy(), out_score.cpu().numpy().copy(), out_count.cpu().numpy().copy())
        if world > 1:
            mg = exchange((out_vec, out_score, out_count))
            torch.cuda.synchronize()
            got_ids = mg[1].cpu().numpy()
        else:
            got_ids = got0[0].astype(np.int64)
        # exact scan (oracle-verified kernel): the first rq queries only, the rest of the buffers keep the HNSW rows
        ev_, es_, ec_ = torch.zeros_like(out_vec), torch.zeros_like(out_score), torch.zeros_like(out_count)
        search(qpool[0], m=_lib.METHOD_BRUTE_FORCE, out=(ev_, es_, ec_), nq=rq)
        torch.cuda.synchronize()
        exact0 = (ev_.cpu().numpy().copy(), es_.cpu().numpy().copy(), ec_.cpu().numpy().copy())
        if world > 1:
            ec_[rq:] = 0
            me = exchange((ev_, es_, ec_))
            torch.cuda.synchronize()
            exact_ids = me[1].cpu().numpy()
        else:
            exact_ids = exact0[0].astype(np.int64)
        found_q = [len(set(got_ids[i][:k].tolist()) & set(exact_ids[i][:k].tolist())) for i in range(rq)]
        recall = float(np.mean(found_q)) / k
        recall_hist = np.bincount(np.asarray(found_q, np.int64), minlength=k + 1).tolist()   # queries with 0, 1, ..., k of the exact top-k

    res = None
    if rank == 0:
        res = {
            "corpus": kind, "elapsed": elapsed, "kernel_ms": kernel_ms, "alone_ms": alone_ms, "nfl": nfl, "alg_bytes": alg_bytes, "achieved": achieved,
            "traffic": traffic, "traffic_src": traffic_src, "recall": recall, "recall_hist": recall_hist, "evals": float(np.mean(evals_q)),
            "expansions": float(np.mean(exp_q)), "edge_hits": float(np.mean(hits_q)), "flags": flags, "timed_flags": timed_flags, "gen_s": gen_s, "open_s": open_s,
            "build_s": build_s, "exchange_check": exchange_check, "library_exchange": comm is not None, "exchange_transport": (None if comm is None else "shm" if same_device else "rccl"), "launch_shape_set_by_bench": crowded_by_hand, "build": build_blk, "steps_timed": steps_timed, "repeats": repeats, "timed_retried": timed_retried,
        }
    if not headline:
        L.nidx_gpu_vector_close(h)
        return res

    # ---- serving shapes of the same batch (reported beside `value`, never as `value`) -----------------------------------------
    extra = {}
    if rank == 0 and not do_exchange and got0 is not None:
        # what the timed entry points deliver to the host == what the launch-only entry leaves in HBM (the block the oracle checks)
        t = C.c_uint64(0)
        _lib.check(L.nidx_gpu_vector_search_submit(h, qpool[0].data_ptr(), B, d, C.byref(p_hnsw), None, C.byref(t)))
        hv_, hs2_, hc_ = host_out[0]
        _lib.check(L.nidx_gpu_vector_search_wait(h, t.value, None, None, hv_.ctypes.data, hs2_.ctypes.data, hc_.ctypes.data, None))
        same_ = bool(np.array_equal(hc_, got0[2].view(np.uint32)) and np.array_equal(hv_, got0[0].view(np.uint32)) and
                     np.array_equal(hs2_.view(np.uint32), got0[1].view(np.uint32)))
        extra["submit_wait_equals_device_entry"] = same_
        if not same_:
            FAILURES.append("submit/wait delivered other hits than the device entry")
    if rank == 0:
        # (1) the complete device entry: launch + one D2H of the result block + flag check (+ fallback when flagged)
        words = B * k * 2 + B + 1
        d_block = torch.zeros((words,), dtype=torch.int32, device=dev)
        h_block = torch.zeros((words,), dtype=torch.int32).pin_memory()
        p = _lib.VectorSearchParamsC(k, -1.0, 1, _lib.METHOD_HNSW)
        retried = C.c_uint32(0)
        reps = 10
        for i in range(reps + 2):
            if i == 2:
                t1 = time.perf_counter()
            _lib.check(L.nidx_gpu_vector_segment_search_device_exact(h, 0, qpool[i % n_pool].data_ptr(), B, C.byref(p), None, d_block.data_ptr(),
                                                                     h_block.data_ptr(), stream, C.byref(retried)))
        extra["exact_entry_queries_per_s"] = B * reps / (time.perf_counter() - t1)
        # (2) host buffers in and out (queries over PCIe, Fssc on the host): nidx_gpu_vector_search
        qh = qpool[0].cpu().numpy()
        hv, hs_, hc = np.zeros((B, k), np.uint32), np.zeros((B, k), np.float32), np.zeros(B, np.uint32)
        for i in range(reps + 2):
            if i == 2:
                t1 = time.perf_counter()
            _lib.check(L.nidx_gpu_vector_search(h, qh.ctypes.data, B, C.byref(p), None, None, None, hv.ctypes.data,
                                                hs_.ctypes.data, hc.ctypes.data, None))
        extra["host_buffer_blocking_queries_per_s"] = B * reps / (time.perf_counter() - t1)
        if got0 is not None:
            extra["host_buffer_equals_device_entry"] = bool(np.array_equal(hv, got0[0].view(np.uint32)) and np.array_equal(hs_.view(np.uint32), got0[1].view(np.uint32)))
        # (2b) what the seam really offers (VectorSearchRequest.vector: Vec<f32>, nidx_vector/src/request_types.rs:19-35): HOST query
        # rows in, hits in host arrays out, through the same pipeline as `value` — nidx_gpu_vector_search_submit / _wait, `nfl` batches
        # in flight; the rows are staged into pinned memory by the submitting thread and the library's helper threads, 3 MiB over
        # PCIe per batch
        if not do_exchange:
            qhost = [qpool[i].cpu().numpy() for i in range(n_pool)]
            tick = []
            # as many batches in flight as the device-resident loop (measured on the 10 M shard, scripts/r5_host.sh: 3 in flight 0.99 of
            # `value`, 4: 0.94, 5: 0.86, 8: 0.75 — also with the library letting only three of them search at once, tunable pipeline_walks)
            nfl_h = int(os.environ.get("NIDX_BENCH_HOST_IN_FLIGHT", str(nfl if native_host_driver() is None else 4)))
            _lib.check(L.nidx_gpu_vector_set_tunable(h, b"pipeline_depth", max(nfl_h, 4)))
            host_out_h = [(np.zeros((B, k), np.uint32), np.zeros((B, k), np.float32), np.zeros(B, np.uint32)) for _ in range(nfl_h)]

            def host_step(i):
                if len(tick) == nfl_h:
                    t_, j_ = tick.pop(0)
                    hv_, hs2_, hc_ = host_out_h[j_]
                    _lib.check(L.nidx_gpu_vector_search_wait(h, t_, None, None, hv_.ctypes.data, hs2_.ctypes.data, hc_.ctypes.data, None))
                t = C.c_uint64(0)
                _lib.check(L.nidx_gpu_vector_search_submit(h, qhost[i % n_pool].ctypes.data, B, d, C.byref(p_hnsw), None, C.byref(t)))
                tick.append((t.value, i % nfl_h))

            def host_drain():
                while tick:
                    t_, j_ = tick.pop(0)
                    hv_, hs2_, hc_ = host_out_h[j_]
                    _lib.check(L.nidx_gpu_vector_search_wait(h, t_, None, None, hv_.ctypes.data, hs2_.ctypes.data, hc_.ctypes.data, None))

            n_host = max(a.steps, int(steps_timed * 0.5))
            drv = native_host_driver()
            # two threads with two tickets each (measured round 6 at 4 M vectors, fraction of the device-resident rate: 1 thread x 3 tickets
            # 0.86, 2 x 2 0.96, 3 x 2 0.92; a thread with ONE ticket stages its next batch only after its last one has landed: 3 x 1 0.64)
            host_threads = max(1, int(os.environ.get("NIDX_BENCH_HOST_THREADS", "2")))
            if drv is not None:
                # the callers are native threads (bench_native/host_driver.cpp; the reference serves every request on a blocking thread of
                # its own, shard_search.rs:139-153): the staging copy of one thread's batch (3 MiB into pinned memory) runs beside the
                # others' waits, which one Python thread could not do — it spent a step's worth of time per batch in staging + glue
                per_thread = max(1, -(-nfl_h // host_threads))
                _lib.check(L.nidx_gpu_vector_set_tunable(h, b"pipeline_depth", max(per_thread * host_threads, 4)))
                ptrs = (C.c_void_p * n_pool)(*[q_.ctypes.data for q_ in qhost])
                el = C.c_double()
                hv_, hs2_, hc_ = host_out_h[0]
                _lib.check(drv.nidx_bench_vector_pipeline(C.cast(L.nidx_gpu_vector_search_submit, C.c_void_p), C.cast(L.nidx_gpu_vector_search_wait, C.c_void_p), h,
                                                          ptrs, n_pool, B, d, C.byref(p_hnsw), host_threads, per_thread, max(4, a.warmup), n_host, C.byref(el),
                                                          hv_.ctypes.data, hs2_.ctypes.data, hc_.ctypes.data))
                dt_host = el.value
                nfl_h = per_thread * host_threads
                # the staging threads share the box's cores with whatever else runs there: three submitting threads are timed beside the two
                # (0.92 against 0.96 of the device-resident rate when the round-6 sweep was taken; the order flips from box to box) and the
                # better of the two is the figure
                by_threads = [{"threads": host_threads, "tickets_each": per_thread, "queries_per_s": B * n_host / dt_host}]
                if "NIDX_BENCH_HOST_THREADS" not in os.environ and "NIDX_BENCH_HOST_IN_FLIGHT" not in os.environ:
                    _lib.check(L.nidx_gpu_vector_set_tunable(h, b"pipeline_depth", 6))
                    el3 = C.c_double()
                    _lib.check(drv.nidx_bench_vector_pipeline(C.cast(L.nidx_gpu_vector_search_submit, C.c_void_p), C.cast(L.nidx_gpu_vector_search_wait, C.c_void_p),
                                                              h, ptrs, n_pool, B, d, C.byref(p_hnsw), 3, 2, max(4, a.warmup), n_host, C.byref(el3),
                                                              hv_.ctypes.data, hs2_.ctypes.data, hc_.ctypes.data))
                    by_threads.append({"threads": 3, "tickets_each": 2, "queries_per_s": B * n_host / el3.value})
                    if el3.value < dt_host:
                        dt_host, host_threads, nfl_h = el3.value, 3, 6
                extra["host_buffer_by_threads"] = by_threads
            else:
                host_threads = 1
                for i in range(max(4, a.warmup)):
                    host_step(i)
                host_drain()
                t1 = time.perf_counter()
                for i in range(n_host):
                    host_step(i)
                host_drain()
                dt_host = time.perf_counter() - t1
            extra["host_buffer_queries_per_s"] = B * n_host / dt_host
            extra["host_buffer_fraction_of_value"] = (B * n_host / dt_host) / (B * steps_timed / elapsed)
            extra["host_buffer_entry"] = ("nidx_gpu_vector_search_submit / _wait: HOST query rows in (staged through pinned memory by the submitting "
                                          "thread + the library's helper threads, 3 MiB over PCIe per batch), hits in host arrays out, %d batches in flight "
                                          "from %d %s submitting thread(s), %d timed" % (nfl_h, host_threads, "Python" if drv is None else "native", n_host))
            if got0 is not None:
                if drv is None:
                    host_step(0)
                    host_drain()
                hv_, hs2_, hc_ = host_out_h[0]
                same_h = bool(np.array_equal(hc_, got0[2].view(np.uint32)) and np.array_equal(hv_, got0[0].view(np.uint32)) and
                              np.array_equal(hs2_.view(np.uint32), got0[1].view(np.uint32)))
                extra["host_buffer_pipelined_equals_device_entry"] = same_h
                if not same_h:
                    FAILURES.append("submit/wait from host rows delivered other hits than the device entry")
        # (3) the reference's request shape: one query per call from many blocking threads (shard_search.rs:139-153),
        # coalesced into batched launches by csrc/coalescer.cpp
        if a.single_query_calls > 0:
            extra["single_query"] = single_query_latency(a, L, h, qpool.reshape(-1, d).cpu().numpy())

    # ---- oracle legs (rank 0): bit parity at this scale, then the CPU baseline ------------------------------------------------
    parity, cpu = None, None
    if rank == 0 and x_host is not None:
        try:
            parity, cpu = oracle_legs(a, L, h, x_host, qpool, got0, exact0, kind)
        except Exception as e:  # side legs: the measured line is still printed, the run is marked failed
            print("ERROR: oracle legs failed: %r" % (e,), file=sys.stderr)
            parity = {"status": "failed: %r" % (e,)}
            FAILURES.append("oracle legs failed: %r" % (e,))
        for name, leg in (parity or {}).items():
            status = leg.get("status") if isinstance(leg, dict) else leg
            if isinstance(status, str) and (status.startswith("MISMATCH") or status.startswith("BEYOND") or status.startswith("failed")):
                FAILURES.append("parity.%s: %s" % (name, status))
                print("ERROR: parity.%s: %s" % (name, status), file=sys.stderr)
    del x_host

    # ---- recall bar and the reference's own search constants ----------------------------------------------------------------------
    # The reference at this corpus size is 50 segments of <= 200 k records, EACH searched at ef = 30 and merged: that regime's
    # recall@k (measured above against the exact scan) is the bar "recall >= reference".  The timed configuration (--ef-upper,
    # layer-0 ef = 30) is checked against it; if it fell short, a ladder of wider layer-0 searches would be walked and the first
    # rung that reaches the bar timed.  The same graph at the reference's constants (greedy descent) is timed beside it.
    iso, ref_consts = None, None
    seg_reg = (cpu or {}).get("segment_regime") if isinstance(cpu, dict) else None
    iso_target = a.iso_target if a.iso_target > 0 else (seg_reg or {}).get("recall_at_%d" % k) if isinstance(seg_reg, dict) else None
    if rank == 0 and world == 1 and a.iso_recall and exact0 is not None:
        rq = min(a.recall_queries, B)
        ev, _es, ec = exact0

        def recall_now():
            search(qpool[0])
            torch.cuda.synchronize()
            g_ = out_vec.cpu().numpy()
            f_ = [len(set(g_[i][:k].tolist()) & set(ev[i, : ec[i]].tolist())) for i in range(rq)]
            return float(np.mean(f_)) / k, np.bincount(np.asarray(f_, np.int64), minlength=k + 1).tolist()

        def timed_now():
            for i in range(max(2, a.warmup)):
                step(i)
            drain()
            n_iso = max(a.steps, int(steps_timed * 0.4))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(n_iso):
                step(i)
            drain()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            ea = [torch.cuda.Event(enable_timing=True) for _ in range(8)]
            for i in range(4):
                ea[2 * i].record()
                search(qpool[i % n_pool])
                ea[2 * i + 1].record()
            torch.cuda.synchronize()
            k_ms = float(np.mean([ea[2 * i].elapsed_time(ea[2 * i + 1]) for i in range(4)]))
            search(qpool[0], with_stats=True)
            torch.cuda.synchronize()
            st = stats.cpu().numpy().astype(np.int64)
            ab = float((st[:, 0] * 4 * d + st[:, 1] * 256).sum())
            return {"queries_per_s": B * n_iso / dt, "ms_per_step": dt / n_iso * 1e3, "steps": n_iso, "batches_in_flight": nfl_now[0],
                    "distance_evals_per_query": float(st[:, 0].mean()), "expansions_per_query": float(st[:, 1].mean()),
                    "kernel_flags": int(np.bitwise_or.reduce(st[:, 3])),
                    "roofline": {"bound": "hbm", "achieved": ab / (k_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": ab / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": ab, "kernel_ms": k_ms,
                                 "sustained_frac": ab * n_iso / dt / 1e9 / HBM_PEAK_GBS}}

        if iso_target is not None:
            iso = {"target": ("recall@%d of the reference regime (%d segments x ef = 30 + Fssc, oracle)" % (k, seg_reg.get("segments", 0)))
                             if a.iso_target <= 0 else "--iso-target",
                   "target_recall": iso_target, "timed_configuration": {"ef_upper": max(1, a.ef_upper), "ef_search": 30, "recall": recall},
                   "knobs": "ef_upper = results kept per upper layer of the descent (reference: 1, hnsw/search.rs:318-324), ef_search = results "
                            "kept on layer 0 (reference: 30); both tunables of the library, defaults = the reference's constants"}
            if recall is not None and recall >= iso_target:
                iso["status"] = "the timed configuration reaches the bar: `value` is the iso-recall figure"
            else:
                iso["ladder"] = []
                chosen = None
                for ef in (48, 64, 96, 128, 192, 256):
                    _lib.check(L.nidx_gpu_vector_set_tunable(h, b"ef_search", ef))
                    r_, hist_ = recall_now()
                    iso["ladder"].append({"ef_upper": max(1, a.ef_upper), "ef_search": ef, "recall": r_, "queries_by_hits": hist_})
                    if r_ >= iso_target:
                        chosen = ef
                        break
                if chosen is not None:
                    iso.update({"ef_search": chosen, "recall_at_%d" % k: iso["ladder"][-1]["recall"]})
                    iso.update(timed_now())
                    iso["status"] = "reached on the ladder"
                else:
                    iso["status"] = "no rung of the ladder reaches the target"
                    FAILURES.append("recall: no configuration of the ladder reaches the reference regime's recall")
                _lib.check(L.nidx_gpu_vector_set_tunable(h, b"ef_search", 0))
        if a.ef_upper > 1:
            _lib.check(L.nidx_gpu_vector_set_tunable(h, b"ef_upper", 0))
            r_, hist_ = recall_now()
            ref_consts = {"ef_upper": 1, "ef_search": 30, "recall_at_%d" % k: r_, "queries_by_hits": hist_,
                          "note": "the same graph searched with the reference's constants (greedy descent): its misses are whole queries whose "
                                  "descent ends in another neighbourhood of a 10 M-node layer 0"}
            ref_consts.update(timed_now())
            # A greedy descent loses a few queries per batch in a far neighbourhood of layer 0, their walks are several times the median's, and
            # a launch lasts as long as its longest walk: with three launches on the device most of its wave slots wait for those few.  The
            # pipeline takes up to 16 tickets; deeper ones are timed beside the default and the best is the figure of this block.
            if not do_exchange and headline:
                ref_consts["by_batches_in_flight"] = [{"batches_in_flight": ref_consts["batches_in_flight"], "queries_per_s": ref_consts["queries_per_s"],
                                                       "sustained_frac": ref_consts["roofline"]["sustained_frac"]}]
                for deeper in (6, 10, 15):
                    nfl_now[0] = deeper
                    _lib.check(L.nidx_gpu_vector_set_tunable(h, b"pipeline_depth", deeper + 1))
                    _lib.check(L.nidx_gpu_vector_set_tunable(h, b"pipeline_walks", deeper))
                    t_ = timed_now()
                    ref_consts["by_batches_in_flight"].append({"batches_in_flight": deeper, "queries_per_s": t_["queries_per_s"],
                                                               "sustained_frac": t_["roofline"]["sustained_frac"]})
                    if t_["queries_per_s"] > ref_consts["queries_per_s"]:
                        ref_consts.update(t_)
                nfl_now[0] = nfl
                _lib.check(L.nidx_gpu_vector_set_tunable(h, b"pipeline_depth", max(nfl, 4)))
                _lib.check(L.nidx_gpu_vector_set_tunable(h, b"pipeline_walks", nfl))
            _lib.check(L.nidx_gpu_vector_set_tunable(h, b"ef_upper", a.ef_upper))

    # ---- the other half of BASELINE.json's metric on the same box: BM25 over as many synthetic documents, and the hybrid batch ---
    bm25_blk, hybrid_blk = None, None
    if a.bm25_block and world == 1:
        try:
            bm = Bm25Bench(a, L, dev, rank, n)
            bm25_blk = bm.timed_block(rank, dev)
            cpu_v = cpu.get("value") if isinstance(cpu, dict) else None
            cpu_b = (bm25_blk.get("cpu_baseline") or {}).get("queries_per_s")
            hybrid_blk = hybrid_block(a, L, dev, rank, h, qpool, bm, nfl, cpu_v, cpu_b)
            bm.close()
        except Exception as e:
            print("ERROR: bm25 / hybrid block failed: %r" % (e,), file=sys.stderr)
            FAILURES.append("bm25 / hybrid block failed: %r" % (e,))
    L.nidx_gpu_vector_close(h)
    if res is not None:
        res.update({"extra": extra, "parity": parity, "cpu": cpu, "iso_recall": iso, "reference_constants": ref_consts, "bm25": bm25_blk, "hybrid": hybrid_blk})
    return res


def host_cpu():
    """(CPU seconds of this process so far, periods in which its cgroup was throttled by the CPU quota so far or None)"""
    import resource
    ru = resource.getrusage(resource.RUSAGE_SELF)
    thr = None
    try:
        for line in open("/sys/fs/cgroup/cpu.stat"):
            if line.startswith("nr_throttled"):
                thr = int(line.split()[1])
    except OSError:
        pass
    return ru.ru_utime + ru.ru_stime, thr


def cpu_quota_cores():
    """The container's CFS quota in cores (cgroup v2 cpu.max), or None: os.cpu_count() shows the machine's CPUs, not what this process may use."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if quota == "max" else float(quota) / float(period)
    except (OSError, ValueError):
        return None


def single_query_latency(a, L, h, queries):
    """nidx_gpu_vector_search_one from 1 / 64 / 256 / 1024 native threads (nidx_gpu_diag_single_query_latency: the reference's one
    blocking thread per request, shard_search.rs:139-153), coalesced by csrc/coalescer.cpp into batches that run through the serving
    pipeline, several of them in flight."""
    from nucliadb_amd import _lib

    d, k = a.dim, a.k
    calls = max(64, a.single_query_calls)
    p = _lib.VectorSearchParamsC(k, -1.0, 1, _lib.METHOD_HNSW)
    q = np.ascontiguousarray(queries, np.float32)
    out = {}

    try:
        out["cgroup_cpu_max"] = open("/sys/fs/cgroup/cpu.max").read().strip()
    except OSError:
        out["cgroup_cpu_max"] = None
    for th in (1, 64, 256, 1024):
        n = min(calls, 256) if th == 1 else max(calls, 48 * th)   # dozens of calls per thread: the start of a run is not its steady state
        lat = np.zeros(n, np.float32)
        el = C.c_double(0)
        # untimed warm-up: small batches take other launch shapes of the kernel, whose code objects load on first use
        _lib.check(L.nidx_gpu_diag_single_query_latency(h, q.ctypes.data, q.shape[0], d, C.byref(p), th, 2 * th, lat.ctypes.data, C.byref(el)))
        b0, q0 = C.c_uint64(0), C.c_uint64(0)
        L.nidx_gpu_vector_coalescer_stats(h, C.byref(b0), C.byref(q0))
        c0, t0_ = host_cpu()
        _lib.check(L.nidx_gpu_diag_single_query_latency(h, q.ctypes.data, q.shape[0], d, C.byref(p), th, n, lat.ctypes.data, C.byref(el)))
        c1, t1_ = host_cpu()
        b1, q1 = C.c_uint64(0), C.c_uint64(0)
        L.nidx_gpu_vector_coalescer_stats(h, C.byref(b1), C.byref(q1))
        out["threads_%d" % th] = {"calls": n, "p50_ms": float(np.percentile(lat, 50) / 1e3), "p99_ms": float(np.percentile(lat, 99) / 1e3),
                                  "queries_per_s": n / el.value,
                                  "queries_per_launch": (q1.value - q0.value) / max(1, b1.value - b0.value),
                                  "host_cores_busy": (c1 - c0) / max(el.value, 1e-9),
                                  "cgroup_throttled_periods": None if t0_ is None or t1_ is None else t1_ - t0_}
    if os.environ.get("NIDX_BENCH_SQ_DOOR") == "1":   # tuning aid: the admission door (coalesce_max_callers) at 1 024 callers
        door = []
        for cap in (256, 512, 1024, 0):
            _lib.check(L.nidx_gpu_vector_set_tunable(h, b"coalesce_max_callers", cap))
            for inflight in (4, 8):
                _lib.check(L.nidx_gpu_vector_set_tunable(h, b"coalesce_in_flight", inflight))
                _lib.check(L.nidx_gpu_vector_set_tunable(h, b"pipeline_depth", max(4, inflight)))
                n = 48 * 1024
                lat = np.zeros(n, np.float32)
                el = C.c_double(0)
                b0, q0 = C.c_uint64(0), C.c_uint64(0)
                L.nidx_gpu_vector_coalescer_stats(h, C.byref(b0), C.byref(q0))
                c0, t0_ = host_cpu()
                _lib.check(L.nidx_gpu_diag_single_query_latency(h, q.ctypes.data, q.shape[0], d, C.byref(p), 1024, n, lat.ctypes.data, C.byref(el)))
                c1, t1_ = host_cpu()
                b1, q1 = C.c_uint64(0), C.c_uint64(0)
                L.nidx_gpu_vector_coalescer_stats(h, C.byref(b1), C.byref(q1))
                door.append({"max_callers": cap, "in_flight": inflight, "queries_per_s": n / el.value, "p50_ms": float(np.percentile(lat, 50) / 1e3),
                             "host_cores_busy": (c1 - c0) / max(el.value, 1e-9), "cgroup_throttled_periods": None if t0_ is None or t1_ is None else t1_ - t0_,
                             "p99_ms": float(np.percentile(lat, 99) / 1e3), "queries_per_launch": (q1.value - q0.value) / max(1, b1.value - b0.value)})
        out["door_sweep_1024_callers"] = door
        _lib.check(L.nidx_gpu_vector_set_tunable(h, b"coalesce_max_callers", 256))
        _lib.check(L.nidx_gpu_vector_set_tunable(h, b"coalesce_in_flight", 4))
    if os.environ.get("NIDX_BENCH_SQ_SWEEP") == "1":   # tuning aid: the coalescer's window / batches in flight at 256 callers
        sweep = []
        for window in (10, 25, 50, 100):
            for inflight in (2, 4, 8):
                _lib.check(L.nidx_gpu_vector_set_tunable(h, b"coalesce_window_us", window))
                _lib.check(L.nidx_gpu_vector_set_tunable(h, b"coalesce_in_flight", inflight))
                _lib.check(L.nidx_gpu_vector_set_tunable(h, b"pipeline_depth", max(4, inflight)))
                for th in (64, 256):
                    n = 16 * th
                    lat = np.zeros(n, np.float32)
                    el = C.c_double(0)
                    _lib.check(L.nidx_gpu_diag_single_query_latency(h, q.ctypes.data, q.shape[0], d, C.byref(p), th, n, lat.ctypes.data, C.byref(el)))
                    sweep.append({"window_us": window, "in_flight": inflight, "threads": th, "queries_per_s": n / el.value,
                                  "p50_ms": float(np.percentile(lat, 50) / 1e3), "p99_ms": float(np.percentile(lat, 99) / 1e3)})
        out["sweep"] = sweep
        _lib.check(L.nidx_gpu_vector_set_tunable(h, b"coalesce_window_us", 50))
        _lib.check(L.nidx_gpu_vector_set_tunable(h, b"coalesce_in_flight", 4))
    return out


def score_bound(cnt_a, score_a, cnt_b, score_b, same_ids, tol=1e-5):
    """north_star: 'cosine scores within 1e-5'.  For the lists whose ids differ between two summation orders (AVX2-shaped oracle vs
    WAVE64 device): the score vectors, rank by rank, must agree within `tol` — a near-tie flipped, nothing else happened."""
    worst, beyond, differing = 0.0, 0, 0
    for i in range(len(same_ids)):
        if same_ids[i]:
            continue
        differing += 1
        c = int(min(cnt_a[i], cnt_b[i]))
        dmax = float(np.max(np.abs(score_a[i, :c].astype(np.float64) - score_b[i, :c].astype(np.float64)))) if c else 0.0
        if cnt_a[i] != cnt_b[i]:
            dmax = max(dmax, 1.0)
        worst = max(worst, dmax)
        beyond += int(dmax > tol)
    return {"lists_with_other_ids": differing, "of": len(same_ids), "max_abs_score_difference_rank_by_rank": worst,
            "lists_beyond_1e-5": beyond, "status": "ok" if beyond == 0 else "BEYOND_TOLERANCE"}


def oracle_legs(a, L, h, x_host, qpool, got0, exact0, kind):
    """(parity, cpu_baseline) for the headline corpus.  Test infrastructure on the host cores; nothing here is timed as `value`."""
    from nucliadb_amd import _lib
    from oracle import oracle as orc

    orc.build()
    n, d, B, k = a.n_vectors, a.dim, a.batch, a.k
    threads = a.cpu_threads or min(64, os.cpu_count() or 1)
    t0 = time.time()
    graph, edges = serialize_graph(L, h)
    og = orc.Hnsw.deserialize_v2(graph, edges)
    del graph, edges
    graph_s = time.time() - t0
    q0 = qpool[0].cpu().numpy()
    parity = {"graph_roundtrip_s": graph_s}
    # ---- (1) the timed kernel against the oracle on the same device-built graph: ids, ranks and score BITS, WAVE64 order
    if a.parity_queries > 0 and got0 is not None:
        nq = min(a.parity_queries, B)
        oseg = orc.Segment(x_host, similarity=orc.SIM_COSINE, order=orc.ORDER_WAVE64, graph=og)
        oseg.ef_upper = a.ef_upper if a.ef_upper > 1 else 0
        ov, os_, oc = oseg.hnsw_search_batch(q0[:nq], k, threads=threads)
        gv, gs, gc = got0
        same = [bool(oc[i] == gc[i] and np.array_equal(ov[i, : oc[i]], gv[i, : oc[i]].view(np.uint32)) and
                     np.array_equal(os_[i, : oc[i]].view(np.uint32), gs[i, : oc[i]].view(np.uint32))) for i in range(nq)]
        parity["hnsw_vs_oracle"] = {"queries": nq, "identical_ids_ranks_score_bits": int(sum(same)),
                                    "status": "ok" if all(same) else "MISMATCH", "oracle_order": "WAVE64",
                                    "reference": "nidx_vector/src/hnsw/search.rs:242-383"}
        if not all(same):
            print("WARNING: device HNSW results differ from the oracle's on %d of %d queries" % (nq - sum(same), nq), file=sys.stderr)
    # ---- (2) the recall ground truth (exact scan kernel) against orc_brute_force_search
    if a.scan_check_queries > 0 and exact0 is not None:
        nq = min(a.scan_check_queries, a.recall_queries, B)
        oseg = orc.Segment(x_host, similarity=orc.SIM_COSINE, order=orc.ORDER_WAVE64)
        ov, os_, oc = oseg.brute_force_batch(q0[:nq], k, threads=min(threads, nq))
        ev, es, ec = exact0
        same = [bool(oc[i] == ec[i] and np.array_equal(ov[i, : oc[i]], ev[i, : oc[i]].view(np.uint32)) and
                     np.array_equal(os_[i, : oc[i]].view(np.uint32), es[i, : oc[i]].view(np.uint32))) for i in range(nq)]
        parity["exact_scan_vs_oracle"] = {"queries": nq, "identical_ids_ranks_score_bits": int(sum(same)),
                                          "status": "ok" if all(same) else "MISMATCH", "reference": "nidx_vector/src/segment.rs:569-623"}
    # ---- (3) CPU baseline, flat: the oracle's HNSW search over the same (device-built) graph, AVX2-shaped sums,
    # one query per POSIX thread (the reference serves one request per blocking thread, shard_search.rs:139-153)
    cpu = None
    if a.cpu_queries > 0:
        qs = qpool.reshape(-1, d)[: a.cpu_queries].cpu().numpy()
        oseg = orc.Segment(x_host, similarity=orc.SIM_COSINE, order=orc.ORDER_HASWELL, graph=og)
        oseg.ef_upper = a.ef_upper if a.ef_upper > 1 else 0   # the same search configuration as the timed device run
        oseg.hnsw_search_batch(qs[:threads], k, threads=threads)  # warm
        t0 = time.perf_counter()
        cv, cs, cc, cst = oseg.hnsw_search_batch(qs, k, threads=threads, want_stats=True)
        dt = time.perf_counter() - t0
        cpu = {"value": qs.shape[0] / dt, "unit": "queries/s", "cores": threads, "cpu_quota_cores": cpu_quota_cores(), "kind": "port",
               "sample": "%d queries of the timed pool over the same %d x %d %s shard and device-built graph, oracle (C restatement of the "
                         "reference algorithm, AVX2-shaped f32 sums), one query per POSIX thread; flat = ONE segment; descent width ef_upper = %d like the timed device run" % (qs.shape[0], n, d, kind, max(1, a.ef_upper)),
               "distance_evals_per_query": float(cst[:, 0].mean())}
        if got0 is not None:
            m = min(B, qs.shape[0])
            cpu["recall_vs_device_ids"] = float(np.mean([len(set(cv[i, : cc[i]].tolist()) & set(got0[0][i, : got0[2][i]].view(np.uint32).tolist())) / k for i in range(m)]))
            same_ids = [bool(cc[i] == got0[2][i] and np.array_equal(cv[i, : cc[i]], got0[0][i, : cc[i]].view(np.uint32))) for i in range(m)]
            cpu["avx2_vs_wave64"] = score_bound(cc[:m], cs[:m], got0[2][:m], got0[1][:m], same_ids)
            cpu["avx2_vs_wave64"]["note"] = ("the timed baseline sums in AVX2 order, the device in WAVE64 order; where the id lists differ the "
                                             "scores must still agree rank by rank within 1e-5 (a near-tie flipped)")
            parity["avx2_vs_wave64_flat"] = cpu["avx2_vs_wave64"]["status"]
    del og
    # ---- (4) CPU baseline in the reference's own regime: segments of <= 200 k records searched one after the other and merged
    # by Fssc (searcher.rs:270-287; the cap: src/settings.rs:258-278).  Graphs of the segments: device-built, like the flat one.
    if a.cpu_queries > 0 and a.segment_regime > 0 and n > a.segment_regime:
        try:
            cpu["segment_regime"] = segment_regime_leg(a, L, x_host, qpool, kind, threads, exact0)
            if cpu["segment_regime"].get("avx2_vs_wave64"):
                parity["avx2_vs_wave64_segments"] = cpu["segment_regime"]["avx2_vs_wave64"]["status"]
            if cpu["segment_regime"].get("device_vs_oracle_wave64"):
                parity["segments_vs_oracle"] = cpu["segment_regime"]["device_vs_oracle_wave64"]
        except Exception as e:
            cpu["segment_regime"] = {"status": "failed: %r" % (e,)}
    # ---- (5) recall of the reference's sequential HnswBuilder vs the device's batch-synchronous build, same data and queries
    if a.ref_build_n > 0:
        try:
            parity["build_recall"] = build_recall_leg(a, L, threads)
        except Exception as e:
            parity["build_recall"] = {"status": "failed: %r" % (e,)}
    return parity, cpu


def segment_regime_leg(a, L, x_host, qpool, kind, threads, exact0):
    from nucliadb_amd import _lib
    from oracle import oracle as orc

    n, d, B, k = a.n_vectors, a.dim, a.batch, a.k
    cap = a.segment_regime
    bounds = list(range(0, n, cap)) + [n]
    S = len(bounds) - 1
    cfg = _lib.VectorConfigC(d, 1, 0, 0)
    csegs = (_lib.VectorSegmentC * S)()
    for s in range(S):
        rows = x_host[bounds[s]: bounds[s + 1]]
        csegs[s] = _lib.VectorSegmentC(rows.ctypes.data, d * 4, rows.shape[0], None, rows.shape[0], None, 0, 0, None, 0, None, None)
    hs = C.c_void_p()
    _lib.check(L.nidx_gpu_vector_open(C.byref(cfg), csegs, S, C.byref(hs)))
    t0 = time.time()
    osegs = []
    for s in range(S):
        _lib.check(L.nidx_gpu_vector_build_hnsw(hs, s, 2))
        g, e = serialize_graph(L, hs, s)
        osegs.append(orc.Segment(x_host[bounds[s]: bounds[s + 1]], similarity=orc.SIM_COSINE, order=orc.ORDER_HASWELL,
                                 graph=orc.Hnsw.deserialize_v2(g, e)))
    build_s = time.time() - t0
    # the device on the same segmented index: (a) nidx_gpu_vector_search_submit / _wait with three batches in flight — every segment's
    # walks in ONE launch per batch (hnsw_search_segments_kernel), Fssc on the device, one transfer of k hits per query; (b) the
    # blocking host-buffer entry point, which takes the same path one batch at a time; (c) the round-3 path for comparison: a launch,
    # a transfer and a wait per segment, Fssc on the host (tunable serial_segments)
    qh = qpool[0].cpu().numpy()
    p = _lib.VectorSearchParamsC(k, -1.0, 1, _lib.METHOD_HNSW)
    hv, hsc, hc, hsg = np.zeros((B, k), np.uint32), np.zeros((B, k), np.float32), np.zeros(B, np.uint32), np.zeros((B, k), np.uint32)

    def blocking(n_batches):
        for _ in range(n_batches):
            _lib.check(L.nidx_gpu_vector_search(hs, qh.ctypes.data, B, C.byref(p), None, hsg.ctypes.data, None, hv.ctypes.data, hsc.ctypes.data,
                                                hc.ctypes.data, None))

    _lib.check(L.nidx_gpu_vector_set_tunable(hs, b"serial_segments", 1))
    blocking(1)
    t1 = time.perf_counter()
    blocking(2)
    serial_qps = B * 2 / (time.perf_counter() - t1)
    serial_ids = (hsg.copy(), hv.copy(), hc.copy())
    _lib.check(L.nidx_gpu_vector_set_tunable(hs, b"serial_segments", 0))
    blocking(1)
    t1 = time.perf_counter()
    blocking(3)
    gpu_qps = B * 3 / (time.perf_counter() - t1)
    one_launch_equals_serial = bool(np.array_equal(hc, serial_ids[2]) and all(
        np.array_equal(hsg[i, : hc[i]], serial_ids[0][i, : hc[i]]) and np.array_equal(hv[i, : hc[i]], serial_ids[1][i, : hc[i]]) for i in range(B)))
    n_pool = qpool.shape[0]
    nfl, n_timed = 3, 12
    _lib.check(L.nidx_gpu_vector_set_tunable(hs, b"pipeline_depth", nfl))
    outs = [[np.zeros((B, k), np.uint32) for _ in range(3)] + [np.zeros((B, k), np.float32), np.zeros(B, np.uint32)] for _ in range(nfl)]

    def pipelined(n_batches):
        tickets = []
        for i in range(n_batches + nfl):
            if i >= nfl:
                o = outs[i % nfl]
                _lib.check(L.nidx_gpu_vector_search_wait(hs, tickets[i - nfl], o[0].ctypes.data, o[1].ctypes.data, o[2].ctypes.data, o[3].ctypes.data,
                                                         o[4].ctypes.data, None))
            if i < n_batches:
                t = C.c_uint64(0)
                _lib.check(L.nidx_gpu_vector_search_submit(hs, qpool[i % n_pool].data_ptr(), B, d, C.byref(p), None, C.byref(t)))
                tickets.append(t.value)

    pipelined(nfl)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    pipelined(n_timed)
    dt1 = time.perf_counter() - t1
    pipe_qps = B * n_timed / dt1
    # ... and again over a region of at least one second, sized from that rate (twelve batches are ~0.1 s)
    n_long = max(n_timed, int(np.ceil(min(a.min_timed_s, 1.0) * n_timed / max(dt1, 1e-6))))
    t1 = time.perf_counter()
    pipelined(n_long)
    dt_long = time.perf_counter() - t1
    pipe_qps = B * n_long / dt_long
    n_timed_regime = n_long
    # the walks' algorithmic bytes (evals x 4D + expansions x 256) from the oracle's counters on a sample of segments
    walk_bytes = None
    try:
        sample = [osegs[i] for i in sorted({0, S // 2, S - 1})]
        st = np.concatenate([sg_.hnsw_search_batch(qh[:64], k, threads=threads, want_stats=True)[3] for sg_ in sample]).astype(np.float64)
        walk_bytes = float(st[:, 0].mean() * 4 * d + st[:, 1].mean() * 256)
        walk_evals = float(st[:, 0].mean())
    except Exception as e:  # noqa: BLE001 — the figure is an annotation
        walk_evals = None
        print("segment regime: no walk counters (%r)" % (e,), file=sys.stderr)
    L.nidx_gpu_vector_close(hs)
    nq = min(a.cpu_queries, max(threads * 4, 512))
    qs = qpool.reshape(-1, d)[:nq].cpu().numpy()
    orc.searcher_search_batch(osegs, qs[:threads], k, threads=threads)
    t0 = time.perf_counter()
    sg, sv, ss, sc = orc.searcher_search_batch(osegs, qs, k, with_duplicates=True, threads=threads)
    dt = time.perf_counter() - t0
    m = min(B, nq)
    same_l = [bool(sc[i] == hc[i] and np.array_equal(sg[i, : sc[i]], hsg[i, : sc[i]]) and np.array_equal(sv[i, : sc[i]], hv[i, : sc[i]])) for i in range(m)]
    same = int(sum(same_l))
    # bit parity AT THE TIMED SHAPE: the same Searcher::_search with the oracle summing in the device's WAVE64 order — segments,
    # vector ids, ranks and score bits of the one-launch + device-Fssc path must be the oracle's (the AVX2-order run above is the
    # timed CPU baseline; it may flip near-ties)
    for sg_ in osegs:
        sg_.order = orc.ORDER_WAVE64
    wq = min(m, 256)
    wsg, wsv, wss, wsc = orc.searcher_search_batch(osegs, qs[:wq], k, with_duplicates=True, threads=threads)
    same_w = int(sum(bool(wsc[i] == hc[i] and np.array_equal(wsg[i, : wsc[i]], hsg[i, : wsc[i]]) and np.array_equal(wsv[i, : wsc[i]], hv[i, : wsc[i]]) and
                          np.array_equal(wss[i, : wsc[i]].view(np.uint32), hsc[i, : wsc[i]].view(np.uint32))) for i in range(wq)))
    out = {"value": nq / dt, "unit": "queries/s", "cores": threads, "cpu_quota_cores": cpu_quota_cores(), "segments": S, "records_per_segment": cap,
           "device_vs_oracle_wave64": {"queries": wq, "identical_segments_ids_ranks_score_bits": same_w, "status": "ok" if same_w == wq else "MISMATCH",
                                       "oracle_order": "WAVE64", "reference": "nidx_vector/src/searcher.rs:149-199,270-287"},
           "sample": "%d queries, oracle Searcher::_search: %d segments of <= %d records searched sequentially + Fssc, one query per POSIX thread" % (nq, S, cap),
           "segment_builds_s": build_s, "device_same_index_host_buffer_queries_per_s": gpu_qps,
           "device_same_index_pipelined_queries_per_s": pipe_qps, "device_same_index_pipelined_batches_timed": n_timed_regime,
           "device_same_index_pipelined_timed_s": dt_long, "device_same_index_serial_segments_queries_per_s": serial_qps,
           "one_launch_ids_identical_to_a_launch_per_segment": one_launch_equals_serial,
           "device_entry": "nidx_gpu_vector_search_submit / _wait, %d batches of %d in flight, %d timed: one launch of %d x %d walks per batch + Fssc on the device" % (nfl, B, n_timed, B, S),
           "device_walk": None if walk_bytes is None else {
               "distance_evals_per_walk": walk_evals, "algorithmic_bytes_per_walk": walk_bytes, "walks_per_query": S,
               "achieved_GBps": pipe_qps * S * walk_bytes / 1e9, "frac_of_hbm_peak": pipe_qps * S * walk_bytes / 1e9 / HBM_PEAK_GBS,
               "queries_per_s_at_hbm_peak": HBM_PEAK_GBS * 1e9 / (S * walk_bytes),
               "note": "every query walks every segment (Searcher::_search): %d walks x %.0f B; the flat index answers a query with one walk" % (S, walk_bytes)},
           "device_ids_identical_to_oracle": "%d/%d (the timed baseline sums in AVX2 order, the device in WAVE64 order: near-ties may flip)" % (same, m),
           "avx2_vs_wave64": score_bound(sc[:m], ss[:m], hc[:m], hsc[:m], same_l)}
    # recall@k of the REFERENCE'S OWN REGIME (every segment searched at ef = 30, merged by Fssc) against the exact scan of the
    # whole shard: the bar 'recall@10 >= reference' is this figure.  Flat row number = segment start + row in the segment.
    if exact0 is not None:
        ev, _es, ec = exact0
        rq = min(a.recall_queries, m, ev.shape[0])
        starts = np.asarray(bounds[:-1], np.int64)

        def rec(seg_, vec_, cnt_):
            r = []
            for i in range(rq):
                flat = (starts[seg_[i, : cnt_[i]].astype(np.int64)] + vec_[i, : cnt_[i]].astype(np.int64)).tolist()
                r.append(len(set(flat) & set(ev[i, : ec[i]].view(np.uint32).astype(np.int64).tolist())) / k)
            return float(np.mean(r))

        out["recall_at_%d" % k] = rec(sg, sv, sc)
        out["recall_at_%d_device_same_index" % k] = rec(hsg, hv, hc)
        out["recall_queries"] = rq
    return out


def build_recall_leg(a, L, threads):
    """HnswBuilder parity is recall (the reference's rayon build is not deterministic): the oracle's sequential build
    (hnsw/build.rs:57-166 restated) and the device's batch-synchronous build of the same clustered segment, both searched
    by the oracle with the same queries, against the oracle's brute force."""
    from nucliadb_amd import _lib
    from oracle import oracle as orc

    n, d, k = a.ref_build_n, a.dim, a.k
    dev = torch.device("cuda", torch.cuda.current_device())
    x = gen_corpus("clustered", n, d, dev, 99)
    nq = 256
    q = gen_queries("clustered", x, 1, nq, d, dev, 3)[0].cpu().numpy()
    xh = x.cpu().numpy()
    del x
    cfg = _lib.VectorConfigC(d, 1, 0, 0)
    cseg = _lib.VectorSegmentC(xh.ctypes.data, d * 4, n, None, n, None, 0, 0, None, 0, None, None)
    h = C.c_void_p()
    _lib.check(L.nidx_gpu_vector_open(C.byref(cfg), C.byref(cseg), 1, C.byref(h)))
    t0 = time.time()
    _lib.check(L.nidx_gpu_vector_build_hnsw(h, 0, 2))
    dev_build_s = time.time() - t0
    g, e = serialize_graph(L, h)
    L.nidx_gpu_vector_close(h)
    oseg = orc.Segment(xh, similarity=orc.SIM_COSINE, order=orc.ORDER_HASWELL)
    ev, _, ec = oseg.brute_force_batch(q, k, threads=threads)
    oseg.graph = orc.Hnsw.deserialize_v2(g, e)
    dv, _, dc = oseg.hnsw_search_batch(q, k, threads=threads)
    t0 = time.time()
    oseg.build_graph(2)
    ref_build_s = time.time() - t0
    rv, _, rc = oseg.hnsw_search_batch(q, k, threads=threads)

    def rec(v, c):
        return float(np.mean([len(set(v[i, : c[i]].tolist()) & set(ev[i, : ec[i]].tolist())) / k for i in range(nq)]))

    return {"vectors": n, "queries": nq, "recall_at_%d_device_build" % k: rec(dv, dc), "recall_at_%d_reference_sequential_build" % k: rec(rv, rc),
            "device_build_s": dev_build_s, "reference_sequential_build_s": ref_build_s,
            "note": "clustered recipe; both graphs searched by the oracle (ef = 30) against the oracle's brute force"}


def bench_hnsw(a, L, dev, rank, world):
    n, d, B, k = a.n_vectors, a.dim, a.batch, a.k
    kinds = ["clustered", "uniform"] if a.corpus == "both" else [a.corpus]
    head = hnsw_leg(a, L, dev, rank, world, kinds[0], True)
    second = hnsw_leg(a, L, dev, rank, world, kinds[1], False) if len(kinds) > 1 else None
    bf16_blk = None
    if world == 1 and a.bf16_block_n > 0:
        try:
            bf16_blk = bf16_block(a, L, dev)
        except Exception as e:  # noqa: BLE001 — a side block: its failure must not cost the measured line
            print("ERROR: bf16 block failed: %r" % (e,), file=sys.stderr)
            FAILURES.append("bf16 block failed: %r" % (e,))
    if rank != 0:
        return
    total_q = world * B * head["steps_timed"]
    extra = head.get("extra") or {}
    cfgd = {
        "workload": "hnsw: %d x %d-dim cosine (%s corpus), k=%d, batch=%d queries, 1 shard per GPU" % (n, d, head["corpus"], k, B),
        "corpus": head["corpus"] + (": the reference's recall recipe (nidx_vector/src/segment.rs:841-905) scaled to the shard: chained centres 0.1 apart, "
                                    "160 vectors per centre at radius 0.01 / 0.03, queries = stored vector + 0.05 noise" if head["corpus"] == "clustered" else
                                    ": uniform(-1,1) normalised (segment.rs:682-695)"),
        "vectors_per_shard": n, "dim": d, "batch": B, "k": k, "shards": world, "corpus_vectors": n * world,
        "merged_queries_per_s": B * head["steps_timed"] / head["elapsed"],
        "recall_at_%d" % k: head["recall"], "recall_queries": min(a.recall_queries, B), "recall_queries_by_hits": head.get("recall_hist"),
        "recall_at_%d_reference_regime" % k: ((head.get("cpu") or {}).get("segment_regime") or {}).get("recall_at_%d" % k),
        "search_knobs": {"ef_upper": max(1, a.ef_upper), "ef_search": 30},
        "iso_recall": head.get("iso_recall"), "reference_constants": head.get("reference_constants"), "bm25": head.get("bm25"), "hybrid": head.get("hybrid"),
        "distance_evals_per_query": head["evals"], "expansions_per_query": head["expansions"],
        "expansions_with_edge_record_fetched_ahead_per_query": head["edge_hits"],
        "kernel_flags": head["flags"], "timed_launch_flags": head["timed_flags"], "timed_queries_re_run_exactly": head["timed_retried"],
        "timed_region": {"steps_per_pass": a.steps, "passes": head["repeats"], "steps_timed": head["steps_timed"], "seconds": head["elapsed"],
                         "entry": "nidx_gpu_vector_search_submit / _wait: device-resident queries in, hits in host arrays out" if world == 1 else
                                  ("nidx_gpu_vector_segment_search_device + nidx_gpu_shard_exchange_merge_vector (%s)" % ("RCCL inside the library" if head.get("exchange_transport") == "rccl" else "the library's shared-memory transport: the ranks share one GPU, validation") if head.get("library_exchange")
                                   else "nidx_gpu_vector_segment_search_device + the same exchange through torch.distributed (validation mode, or the library's communicator could not be set up)")},
        "corpus_gen_s": head["gen_s"], "open_s": head["open_s"], "hnsw_build_s": head["build_s"], "build_ef_upper": max(1, a.build_ef_upper),
        "parallelism": "shard-per-gpu x%d, RCCL all-gather of top-k" % world, "exchange_check": head["exchange_check"],
        "batches_in_flight": head["nfl"],
        "parity": head.get("parity"),
    }
    cfgd.update(extra)
    # the figures of the nested blocks a reader of the top level needs, as scalars
    seg_reg = (head.get("cpu") or {}).get("segment_regime") or {}
    bm, hy = head.get("bm25") or {}, head.get("hybrid") or {}
    bld = head.get("build") or {}
    # (a record that keeps only the first scalars of `config` still carries every block's headline figure: these come first)
    first = {
        "workload": cfgd["workload"], "vectors_per_shard": n, "shards": world,
        "recall_at_%d" % k: head["recall"], "recall_at_%d_reference_regime" % k: seg_reg.get("recall_at_%d" % k),
        "host_buffer_queries_per_s": extra.get("host_buffer_queries_per_s"),
        "bm25_postings_per_s": bm.get("value"), "bm25_roofline_frac": (bm.get("roofline") or {}).get("frac"),
        "bm25_kernel_ms": (bm.get("roofline") or {}).get("kernel_ms"),
        "bm25_scoring_alone_roofline_frac": (((bm.get("roofline") or {}).get("scoring_alone")) or {}).get("frac"),
        "bm25_multi_segment_postings_per_s": (bm.get("multi_segment") or {}).get("value"),
        "hybrid_queries_per_s": hy.get("value"),
        "segment_regime_device_qps": seg_reg.get("device_same_index_pipelined_queries_per_s"),
        "segment_regime_device_frac_of_hbm_peak": (seg_reg.get("device_walk") or {}).get("frac_of_hbm_peak"),
        "bf16_fallback_frac_of_bf16_peak": ((bf16_blk or {}).get("roofline") or {}).get("frac"),
        "hnsw_build_s": head["build_s"], "hnsw_build_frac_of_hbm_peak": (bld.get("roofline") or {}).get("frac"),
        "exchange_check": head["exchange_check"], "ef_upper": max(1, a.ef_upper),
        # the same graph at the reference's own constants (ef_upper = 1: the greedy descent of hnsw/search.rs:318-324), first class
        "reference_constants_queries_per_s": (head.get("reference_constants") or {}).get("queries_per_s"),
        "reference_constants_recall_at_%d" % k: (head.get("reference_constants") or {}).get("recall_at_%d" % k),
        "reference_constants_roofline_frac": ((head.get("reference_constants") or {}).get("roofline") or {}).get("frac"),
        "reference_constants_sustained_frac": ((head.get("reference_constants") or {}).get("roofline") or {}).get("sustained_frac"),
    }
    cfgd = dict(first, **{kk_: v for kk_, v in cfgd.items() if kk_ not in first})
    cfgd["build"] = bld or None
    cfgd.update({
        "ef_upper": max(1, a.ef_upper), "build_ef_upper": max(1, a.build_ef_upper),
        "bm25_postings_per_s": bm.get("value"), "bm25_queries_per_s": bm.get("queries_per_s"),
        "bm25_roofline_frac": (bm.get("roofline") or {}).get("frac"), "bm25_kernel_ms": (bm.get("roofline") or {}).get("kernel_ms"),
        "hybrid_queries_per_s": hy.get("value"),
        "segment_regime_device_qps": seg_reg.get("device_same_index_pipelined_queries_per_s"),
        "segment_regime_device_blocking_qps": seg_reg.get("device_same_index_host_buffer_queries_per_s"),
        "segment_regime_device_frac_of_hbm_peak": (seg_reg.get("device_walk") or {}).get("frac_of_hbm_peak"),
        "segment_regime_cpu_qps": seg_reg.get("value"), "segment_regime_segments": seg_reg.get("segments"),
        "bf16_fallback": bf16_blk,
        "bf16_fallback_queries_per_s": (bf16_blk or {}).get("queries_per_s"), "bf16_fallback_tflops": ((bf16_blk or {}).get("roofline") or {}).get("achieved"),
        "bf16_fallback_frac_of_bf16_peak": ((bf16_blk or {}).get("roofline") or {}).get("frac"),
    })
    if second is not None:
        cfgd["uniform_corpus" if second["corpus"] == "uniform" else "second_corpus"] = {
            "workload": "hnsw: %d x %d-dim cosine (%s corpus), k=%d, batch=%d queries" % (n, d, second["corpus"], k, B),
            "queries_per_s": world * B * second["steps_timed"] / second["elapsed"], "ms_per_step": second["elapsed"] / second["steps_timed"] * 1e3,
            "recall_at_%d" % k: second["recall"], "distance_evals_per_query": second["evals"], "expansions_per_query": second["expansions"],
            "kernel_flags": second["flags"], "timed_launch_flags": second["timed_flags"], "hnsw_build_s": second["build_s"],
            "batches_in_flight": second["nfl"],
            "roofline": {"kernel": "hnsw_search_kernel<3,4,4,1>", "bound": "hbm", "achieved": second["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": second["achieved"] / HBM_PEAK_GBS, "traffic": second["traffic"], "traffic_source": second["traffic_src"],
                         "algorithmic_bytes_per_launch": second["alg_bytes"], "kernel_ms": second["kernel_ms"]},
            "note": "uniform random 768-d unit vectors have no neighbourhood structure (all pairwise cosines within +-0.1): ef = 30 cannot "
                    "find the exact top-10 among near-ties, for the reference either; it is the worst-case access pattern (every neighbour unvisited)",
        }
    line = {
        "metric": "queries/sec + recall@%d (768-dim cosine k-NN, HNSW M=30 ef=30 ef_upper=%d build_ef_upper=%d, k=%d)" % (
            k, max(1, a.ef_upper), max(1, a.build_ef_upper), k),
        "value": total_q / head["elapsed"],
        # N > 1: `value` counts a query once per shard it is searched in (the weak-scaling numerator: per-GPU work is fixed); the
        # end-to-end rate of the %d-vector index — one merged answer per query — is merged_queries_per_s = value / n_gpus
        "unit": ("queries/s against one %d-vector index" % n) if world == 1 else
                ("SHARD-queries/s: every query is searched in all %d shards of %d vectors (one per GPU) and counted once per shard; "
                 "merged answers per second over the %d-vector index = merged_queries_per_s" % (world, n, n * world)),
        "merged_queries_per_s": B * head["steps_timed"] / head["elapsed"],
        "corpus_vectors": n * world,
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": head["elapsed"] / head["steps_timed"] * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": cfgd,
        # Per-launch figures: the timed batches overlap (nfl in flight), and an event pair on a stream then also spans the time the
        # launch waits for workgroup slots, which no profiler counts as kernel time.  The roofline's launch duration is therefore
        # measured on the same launches issued one at a time right after the timed region (HIP events on the launch stream; this is
        # the figure rocprofv3 --kernel-trace of `bench.py --batches-in-flight 1` reproduces: profiles/r02_pmc_hnsw10m_clustered.txt);
        # the overlapped durations and the sustained rate of the timed region are given beside it.
        "roofline": {"kernel": "hnsw_search_kernel<3,4,4,1>", "bound": "hbm", "achieved": head["alg_bytes"] / (head["alone_ms"] * 1e-3) / 1e9,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": head["alg_bytes"] / (head["alone_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "traffic": head["traffic"], "traffic_source": head["traffic_src"],
                     "algorithmic_bytes_per_launch": head["alg_bytes"], "kernel_ms": head["alone_ms"],
                     "note": "one launch at a time (a launch lasts as long as its longest walk); the timed region keeps %d batches in flight, and a batch "
                             "that finds others on the device is launched as hnsw_search_kernel<3,2,5,1> with a 2^12 visited table (five workgroups per CU "
                             "instead of four; same hits): DESIGN.md 4.1" % head["nfl"],
                     "sustained": {"achieved": head["alg_bytes"] * head["steps_timed"] / head["elapsed"] / 1e9,
                                   "frac": head["alg_bytes"] * head["steps_timed"] / head["elapsed"] / 1e9 / HBM_PEAK_GBS,
                                   "note": "algorithmic bytes of all timed launches / elapsed time of the timed region"},
                     "gather_ceiling": gather_ceiling()},
        "cpu_baseline": head.get("cpu"),
    }
    if FAILURES:
        line["failures"] = FAILURES
    line["bench_wall_s"] = time.time() - T_PROCESS_START
    print(json.dumps(line))


def bf16_block(a, L, dev):
    """BASELINE.json configs[4] on one GPU's share: the batched brute-force fallback on the bf16 matrix cores (csrc/vector_bf16.hip:
    tiled bf16 copy, MFMA 32x32x16, exact f32 re-score of the candidates) over `--bf16-block-n` x 1024-dim cosine vectors resident in
    HBM, batch and k of the headline.  One launch at a time, HIP events on the launch stream; recall@k against the exact f32 scan."""
    from nucliadb_amd import _lib

    n, d, B, k = a.bf16_block_n, 1024, a.batch, a.k
    g = torch.Generator(device=dev)
    g.manual_seed(4242)
    t0 = time.time()
    x = torch.empty((n, d), device=dev, dtype=torch.float32)
    step = 1 << 20
    for i in range(0, n, step):   # uniform(-1, 1) normalised (segment.rs:682-695), in pieces: no second 51 GB temporary
        j = min(n, i + step)
        x[i:j] = torch.rand((j - i, d), generator=g, device=dev, dtype=torch.float32) * 2 - 1
        x[i:j] /= x[i:j].norm(dim=1, keepdim=True)
    q = torch.rand((2, B, d), generator=g, device=dev, dtype=torch.float32) * 2 - 1
    q /= q.norm(dim=2, keepdim=True)
    torch.cuda.synchronize()
    gen_s = time.time() - t0
    cfg = _lib.VectorConfigC(d, 1, 0, 0)
    cseg = _lib.VectorSegmentC(x.data_ptr(), d * 4, n, None, n, None, 0, 0, None, 0, None, None)
    h = C.c_void_p()
    t0 = time.time()
    _lib.check(L.nidx_gpu_vector_open(C.byref(cfg), C.byref(cseg), 1, C.byref(h)))
    open_s = time.time() - t0
    del x
    torch.cuda.empty_cache()
    ov = torch.zeros((B, k), dtype=torch.int32, device=dev)
    osc = torch.zeros((B, k), dtype=torch.float32, device=dev)
    oc = torch.zeros((B,), dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def search(qb, method):
        p = _lib.VectorSearchParamsC(k, -1.0, 1, method)
        _lib.check(L.nidx_gpu_vector_segment_search_device(h, 0, qb.data_ptr(), B, C.byref(p), None, ov.data_ptr(), osc.data_ptr(), oc.data_ptr(), None, stream))

    try:
        search(q[0], _lib.METHOD_BRUTE_FORCE_BF16)   # builds the tiled bf16 copy
        torch.cuda.synchronize()
        steps = 4
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        t0 = time.perf_counter()
        for i in range(steps):
            ev[i][0].record()
            search(q[i & 1], _lib.METHOD_BRUTE_FORCE_BF16)
            ev[i][1].record()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        kernel_ms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in ev]))
        # the same batches with the other scan of the fallback (NIDX_GPU_BF16_APPEND: 1 / unset = floors + bf16_append_kernel, the
        # default; 0 = bf16_scan_kernel with its candidate lists alone): both are the product, the environment picks
        other_env = "0" if os.environ.get("NIDX_GPU_BF16_APPEND", "1") != "0" else "1"
        saved_env = os.environ.get("NIDX_GPU_BF16_APPEND")
        os.environ["NIDX_GPU_BF16_APPEND"] = other_env
        try:
            search(q[0], _lib.METHOD_BRUTE_FORCE_BF16)
            torch.cuda.synchronize()
            ev2 = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(2)]
            for i in range(2):
                ev2[i][0].record()
                search(q[i & 1], _lib.METHOD_BRUTE_FORCE_BF16)
                ev2[i][1].record()
            torch.cuda.synchronize()
            other_ms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in ev2]))
        finally:
            if saved_env is None:
                os.environ.pop("NIDX_GPU_BF16_APPEND", None)
            else:
                os.environ["NIDX_GPU_BF16_APPEND"] = saved_env
        rq = min(64, B)
        search(q[0], _lib.METHOD_BRUTE_FORCE_BF16)
        torch.cuda.synchronize()
        got = ov[:rq].cpu().numpy().copy()
        search(q[0], _lib.METHOD_BRUTE_FORCE)
        torch.cuda.synchronize()
        exact = ov[:rq].cpu().numpy()
        recall = float(np.mean([len(set(got[i]) & set(exact[i])) / k for i in range(rq)]))
    finally:
        L.nidx_gpu_vector_close(h)
    flops = 2.0 * n * d * B
    tf = flops / (kernel_ms * 1e-3) / 1e12
    return {"workload": "bf16 fallback: %d x %d-dim cosine (uniform corpus), k=%d, batch=%d queries, exact f32 re-score of the candidates" % (n, d, k, B),
            "queries_per_s": B * steps / elapsed, "ms_per_batch": kernel_ms, "recall_at_%d_vs_exact_scan" % k: recall, "recall_queries": rq,
            "corpus_gen_s": gen_s, "open_s": open_s,
            "scan": "append" if other_env == "0" else "lists",
            "ms_per_batch_other_scan": {"scan": "lists (NIDX_GPU_BF16_APPEND=0)" if other_env == "0" else "append (NIDX_GPU_BF16_APPEND=1)", "ms_per_batch": other_ms,
                                        "frac": flops / (other_ms * 1e-3) / 1e12 / 2500.0},
            "roofline": {"kernel": "bf16 fallback launches of one batch (sample passes, bf16_append_kernel / bf16_scan_kernel, merge_topk_kernel, rescore_select_kernel)", "bound": "mfma", "achieved": tf, "peak": 2500.0,
                         "unit": "TFLOP/s", "frac": tf / 2500.0, "traffic": None, "algorithmic_flops_per_launch": flops, "kernel_ms": kernel_ms,
                         "hbm_frac_bf16_read_once": float(n) * d * 2 / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}}


def gather_ceiling():
    """The measured ceiling this kernel's access pattern has on the box: random 3-KiB-row gathers, read only
    (profiles/r02_gather_ceiling.json, written by bench.py --workload gather)."""
    try:
        with open(os.path.join(ROOT, "profiles", "r02_gather_ceiling.json")) as f:
            j = json.load(f)
        return {"GBps": j["best_GBps"], "source": "profiles/r02_gather_ceiling.json"}
    except (OSError, KeyError, ValueError):
        return None



def bench_gather(a, L, dev, rank):
    """Calibration of the HBM roofline claim: read-only random gathers of 4*dim-byte rows over an n_vectors x dim matrix (no
    traversal, no arithmetic beyond a fold), swept over wavefronts in flight and rows in flight per wave.  The best rate is
    the gather ceiling of this box; copy it to profiles/r02_gather_ceiling.json."""
    from nucliadb_amd import _lib

    n, d = a.n_vectors, a.dim
    x = torch.empty((n, d), device=dev, dtype=torch.float32)
    chunk = 1 << 20
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    for i0 in range(0, n, chunk):
        x[i0:i0 + chunk] = torch.rand((min(chunk, n - i0), d), generator=g, device=dev, dtype=torch.float32)
    torch.cuda.synchronize()
    sweep = []
    for waves in (4096, 8192, 16384, 32768):
        for rif in (1, 2, 4, 8):
            gpw = max(64, (1 << 22) // waves)  # ~4 M row gathers (12.9 GB at 768 dims) per launch
            ms = C.c_float(0)
            _lib.check(L.nidx_gpu_diag_gather(x.data_ptr(), n, d, waves, gpw, rif, 5, C.byref(ms)))
            gbps = waves * gpw * d * 4 / (ms.value * 1e-3) / 1e9
            sweep.append({"waves": waves, "rows_in_flight": rif, "gathers_per_wave": gpw, "ms": ms.value, "GBps": gbps,
                          "rows_per_s": waves * gpw / (ms.value * 1e-3)})
    # the same sweep confined to a 256 MiB window (what the L2 / MALL can hold): the cached ceiling
    m = min(n, (256 << 20) // (d * 4))
    cached = []
    for rif in (2, 4, 8):
        ms = C.c_float(0)
        _lib.check(L.nidx_gpu_diag_gather(x.data_ptr(), m, d, 16384, 256, rif, 5, C.byref(ms)))
        cached.append({"rows_in_flight": rif, "GBps": 16384 * 256 * d * 4 / (ms.value * 1e-3) / 1e9})
    best = max(sweep, key=lambda r: r["GBps"])
    if rank == 0:
        print(json.dumps({"metric": "random row-gather rate (read only)", "value": best["GBps"], "unit": "GB/s", "n_gpus": 1,
                          "config": {"workload": "gather: %d x %d f32 rows (%.1f GB), %d-byte rows, uniform random row numbers" % (n, d, n * d * 4 / 1e9, d * 4)},
                          "best_GBps": best["GBps"], "best": best, "frac_of_8TBps": best["GBps"] / HBM_PEAK_GBS, "sweep": sweep,
                          "cached_256MiB_window": cached}))


def zipf_corpus_on_device(L, dev, n_docs, vocab, rank):
    """T-zipf of SURVEY §8d generated with torch on the device -> host CSR postings."""
    g = torch.Generator(device=dev)
    g.manual_seed(1234567890 + rank)
    lens = torch.exp(torch.randn(n_docs, generator=g, device=dev) * 0.6 + np.log(48.0)).round().clamp(4, 2000).to(torch.int64)
    total_tokens = int(lens.sum().item())
    cdf = torch.cumsum(1.0 / torch.arange(1, vocab + 1, device=dev, dtype=torch.float64), 0)
    cdf /= cdf[-1].clone()
    doc_of = torch.repeat_interleave(torch.arange(n_docs, device=dev, dtype=torch.int64), lens)
    key = torch.empty(total_tokens, dtype=torch.int64, device=dev)
    chunk = 1 << 26
    for s0 in range(0, total_tokens, chunk):
        u = torch.rand(min(chunk, total_tokens - s0), generator=g, device=dev, dtype=torch.float64)
        term = torch.searchsorted(cdf, u).clamp_(max=vocab - 1)
        key[s0:s0 + chunk] = term * n_docs + doc_of[s0:s0 + chunk]
    del doc_of
    key, _ = torch.sort(key)
    uniq, counts = torch.unique_consecutive(key, return_counts=True)
    del key
    term = uniq // n_docs
    doc_ids = (uniq % n_docs).to(torch.int32).cpu().numpy().astype(np.uint32)
    tfs = counts.to(torch.int32).cpu().numpy().astype(np.uint32)
    df = torch.bincount(term, minlength=vocab)
    term_offsets = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(df, 0)]).cpu().numpy().astype(np.uint64)
    table = torch.tensor([L.nidx_gpu_fieldnorm_from_id(i) for i in range(256)], device=dev, dtype=torch.int64)
    fieldnorm_ids = (torch.searchsorted(table, lens, right=True) - 1).to(torch.uint8).cpu().numpy()
    del uniq, counts, term, df
    torch.cuda.empty_cache()
    return term_offsets, doc_ids, tfs, fieldnorm_ids, total_tokens


_HOST_DRIVER = []


def native_host_driver():
    """bench_native/libnidx_bench_host.so (built by __graft_entry__.build()): the caller's submit / wait loops in native threads.
    NIDX_BENCH_PYTHON_THREADS=1 keeps the Python threads of rounds 4-5 (comparison).  -> CDLL or None"""
    if os.environ.get("NIDX_BENCH_PYTHON_THREADS") == "1":
        return None
    if not _HOST_DRIVER:
        path = os.path.join(ROOT, "bench_native", "libnidx_bench_host.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "bench_native"), "-s"])
        d = C.CDLL(path)
        d.nidx_bench_vector_pipeline.restype = C.c_int32
        d.nidx_bench_vector_pipeline.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32,
                                                 C.c_uint32, C.c_uint32, C.c_uint64, C.POINTER(C.c_double), C.c_void_p, C.c_void_p, C.c_void_p]
        d.nidx_bench_bm25_pipeline.restype = C.c_int32
        d.nidx_bench_bm25_pipeline.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32,
                                               C.c_uint64, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_double), C.c_void_p, C.c_uint64]
        _HOST_DRIVER.append(d)
    return _HOST_DRIVER[0]


def rrf_batch(vec_ids, vec_cnt, bm_ids, bm_cnt, k_out, k_rrf=60.0):
    """ReciprocalRankFusion (nucliadb rank_fusion.py:106-181) for a whole batch with numpy: both lists arrive ranked;
    score(d) = sum 1 / (k + rank); returns the k_out best fused ids per query."""
    B = vec_ids.shape[0]
    ids = np.concatenate([vec_ids, bm_ids], axis=1).astype(np.int64)
    w = np.concatenate([1.0 / (k_rrf + np.arange(vec_ids.shape[1])), 1.0 / (k_rrf + np.arange(bm_ids.shape[1]))])
    w = np.broadcast_to(w, ids.shape).copy()
    valid = np.concatenate([np.arange(vec_ids.shape[1])[None, :] < vec_cnt[:, None], np.arange(bm_ids.shape[1])[None, :] < bm_cnt[:, None]], axis=1)
    w[~valid] = 0.0
    ids[~valid] = -1
    order = np.argsort(ids, axis=1, kind="stable")
    ids_s = np.take_along_axis(ids, order, axis=1)
    w_s = np.take_along_axis(w, order, axis=1)
    same = np.zeros_like(ids_s, dtype=bool)
    same[:, 1:] = ids_s[:, 1:] == ids_s[:, :-1]
    # a document appears at most once per list: fold the second occurrence into the first
    w_s[:, :-1] += np.where(same[:, 1:], w_s[:, 1:], 0.0)
    w_s[same] = 0.0
    top = np.argsort(-w_s, axis=1, kind="stable")[:, :k_out]
    return np.take_along_axis(ids_s, top, axis=1), np.take_along_axis(w_s, top, axis=1)


class Bm25Bench:
    """The BM25 half of BASELINE.json's metric: T-zipf corpus of SURVEY §8d (vocabulary 1 M, term ids ~ Zipf(1.0), doc length ~
    lognormal(ln 48, 0.6) in [4, 2000]) resident in HBM; batches of `B` queries x 3 Should terms drawn uniformly from the rank band
    [100, 100 k], k = 20."""

    K = 20

    def __init__(self, a, L, dev, rank, n_docs, n_pool=32):
        from nucliadb_amd import _lib
        from nucliadb_amd.bm25 import Bm25Searcher, Bm25Segment

        self.a, self.L, self.n_docs, self.vocab, self.B = a, L, n_docs, a.vocab, a.batch
        t0 = time.time()
        self.corpus = zipf_corpus_on_device(L, dev, n_docs, a.vocab, rank)
        self.gen_s = time.time() - t0
        t0 = time.time()
        self.searcher = Bm25Searcher.open([Bm25Segment(*self.corpus)])
        self.open_s = time.time() - t0
        rng = np.random.default_rng(2)
        B = self.B
        self.terms = [rng.integers(99, 100_000, (B, 3)) for _ in range(n_pool)]
        self.prepared = []
        for terms in self.terms:   # clause arrays are prepared once: the timed region is the library call (clauses in, hits out)
            cl = (_lib.Bm25ClauseC * (3 * B))()
            for i in range(B):
                for j in range(3):
                    cl[3 * i + j].term, cl[3 * i + j].occur, cl[3 * i + j].mode, cl[3 * i + j].boost = int(terms[i, j]), 0, 0, 1.0
            self.prepared.append(cl)
        self.offsets = (np.arange(B + 1, dtype=np.uint64) * 3).copy()
        k = self.K
        self.docaddr, self.score = np.zeros((B, k), np.uint64), np.zeros((B, k), np.float32)
        self.count, self.total, self.post = np.zeros(B, np.uint32), np.zeros(B, np.uint64), np.zeros(B, np.uint64)

    def search(self, i):
        from nucliadb_amd import _lib

        _lib.check(self.L.nidx_gpu_bm25_search(self.searcher._handle, self.prepared[i % len(self.prepared)], self.offsets.ctypes.data, self.B, self.K, None,
                                               self.docaddr.ctypes.data, self.score.ctypes.data, self.count.ctypes.data, self.total.ctypes.data,
                                               self.post.ctypes.data))

    def kernel_ms(self):
        ms = C.c_float()
        self.L.nidx_gpu_bm25_last_kernel_ms(self.searcher._handle, C.byref(ms))
        return ms.value

    def submit(self, i):
        """nidx_gpu_bm25_search_submit of prepared batch i -> ticket"""
        from nucliadb_amd import _lib

        if not hasattr(self, "_opt"):
            self._opt = _lib.Bm25SearchOptionsC()
            self._opt.k, self._opt.order_field = self.K, -1
            self._zero64 = np.zeros(1, np.uint64)
            self._opt.term_set_offsets = self._opt.phrase_offsets = self._opt.subquery_offsets = self._zero64.ctypes.data
        t = C.c_uint64(0)
        _lib.check(self.L.nidx_gpu_bm25_search_submit(self.searcher._handle, self.prepared[i % len(self.prepared)], self.offsets.ctypes.data, self.B,
                                                      C.byref(self._opt), C.byref(t)))
        return t.value

    def wait(self, ticket, out=None):
        """out: (docaddr u64 [B][K], score f32 [B][K], count u32 [B]) to fill instead of the bench's own arrays"""
        from nucliadb_amd import _lib

        da, sc, cn = out if out is not None else (self.docaddr, self.score, self.count)
        _lib.check(self.L.nidx_gpu_bm25_search_wait(self.searcher._handle, ticket, da.ctypes.data, sc.ctypes.data, cn.ctypes.data,
                                                    self.total.ctypes.data, self.post.ctypes.data))

    def close(self):
        self.searcher.close()

    def timed_pipeline(self, searcher, n_threads, depth, min_s=None):
        """`n_threads` threads, each: submit / wait with `depth` tickets in flight over the prepared batches, for at least --steps steps
        and --min-timed-s seconds -> (elapsed s, batches, postings scored, postings per batch)"""
        import threading

        from nucliadb_amd import _lib

        a, B, k, L = self.a, self.B, self.K, self.L
        min_s = min(a.min_timed_s, 1.0) if min_s is None else min_s
        opt = _lib.Bm25SearchOptionsC()
        opt.k, opt.order_field = k, -1
        zero64 = np.zeros(1, np.uint64)
        opt.term_set_offsets = opt.phrase_offsets = opt.subquery_offsets = zero64.ctypes.data
        handle = searcher._handle
        drv = native_host_driver()
        if drv is not None:
            # the submitting threads are native (bench_native/host_driver.cpp: a client of include/nidx_gpu.h like the reference's Rust
            # host, one blocking thread per request — shard_search.rs:139-153): six Python threads spent ~30 us per batch in the
            # interpreter lock, more than a batch's kernels take
            ptrs = (C.c_void_p * len(self.prepared))(*[C.addressof(p_) for p_ in self.prepared])
            cap = 1 << 16
            per = np.zeros(cap, np.float64)
            el, nb, po = C.c_double(), C.c_uint64(), C.c_double()
            rc = drv.nidx_bench_bm25_pipeline(C.cast(L.nidx_gpu_bm25_search_submit, C.c_void_p), C.cast(L.nidx_gpu_bm25_search_wait, C.c_void_p), handle, ptrs,
                                              len(self.prepared), self.offsets.ctypes.data, B, C.byref(opt), n_threads, depth, int(a.steps), float(min_s),
                                              C.byref(el), C.byref(nb), C.byref(po), per.ctypes.data, cap)
            _lib.check(rc)
            n = int(min(nb.value, cap))
            return el.value, int(nb.value), float(po.value), per[:n].tolist()
        results = [None] * n_threads
        start = threading.Barrier(n_threads + 1)
        stop_at = [0.0]

        def worker(w):
            docaddr, score = np.zeros((B, k), np.uint64), np.zeros((B, k), np.float32)
            count, total, post = np.zeros(B, np.uint32), np.zeros(B, np.uint64), np.zeros(B, np.uint64)
            pending, per_batch = [], []

            def submit(i):
                t = C.c_uint64(0)
                _lib.check(L.nidx_gpu_bm25_search_submit(handle, self.prepared[i % len(self.prepared)], self.offsets.ctypes.data, B, C.byref(opt), C.byref(t)))
                pending.append(t.value)

            def wait_oldest():
                _lib.check(L.nidx_gpu_bm25_search_wait(handle, pending.pop(0), docaddr.ctypes.data, score.ctypes.data, count.ctypes.data, total.ctypes.data,
                                                       post.ctypes.data))
                return float(post.sum())

            try:
                for i in range(4):   # the slots' buffers and streams exist before the clock starts
                    submit(w + i * n_threads)
                    if len(pending) >= depth:
                        wait_oldest()
                while pending:
                    wait_oldest()
                start.wait()
                n = 0
                while n * n_threads < a.steps or time.perf_counter() < stop_at[0]:
                    submit(w + n * n_threads)
                    if len(pending) >= depth:
                        per_batch.append(wait_oldest())
                    n += 1
                while pending:
                    per_batch.append(wait_oldest())
                results[w] = (n, per_batch, time.perf_counter())
            except BaseException as e:   # noqa: BLE001
                results[w] = e
                try:
                    start.abort()
                except Exception:
                    pass

        ths = [threading.Thread(target=worker, args=(w,)) for w in range(n_threads)]
        for t in ths:
            t.start()
        stop_at[0] = time.perf_counter() + 3600.0
        start.wait()
        t0 = time.perf_counter()
        stop_at[0] = t0 + min_s
        for t in ths:
            t.join()
        for r in results:
            if isinstance(r, BaseException):
                raise r
        elapsed = max(r[2] for r in results) - t0
        per_batch = [x for r in results for x in r[1]]
        return elapsed, sum(r[0] for r in results), float(np.sum(per_batch)), per_batch

    def split_by_merge_policy(self, dev):
        """The same documents as the segments the log-merge policy leaves behind (nidx/src/settings.rs:246-253: merges run at >= 4
        segments of a level and stop at 10 M records): one of 60 %, one of 30 %, four of 2.5 % -> [Bm25Segment]"""
        from nucliadb_amd.bm25 import Bm25Segment

        term_offsets, doc_ids, tfs, fieldnorm_ids, tokens = self.corpus
        n = self.n_docs
        cuts = [0, int(n * 0.6), int(n * 0.9)] + [int(n * (0.9 + 0.025 * i)) for i in range(1, 4)] + [n]
        d_doc = torch.from_numpy(doc_ids.view(np.int32)).to(dev)
        d_tf = torch.from_numpy(tfs.view(np.int32)).to(dev)
        df = torch.from_numpy(np.diff(term_offsets.astype(np.int64))).to(dev)
        term_of = torch.repeat_interleave(torch.arange(self.vocab, device=dev, dtype=torch.int32), df)
        table = np.array([self.L.nidx_gpu_fieldnorm_from_id(i) for i in range(256)], dtype=np.int64)
        parts = []
        for a_, b_ in zip(cuts[:-1], cuts[1:]):
            m = (d_doc >= a_) & (d_doc < b_)
            pd = (d_doc[m] - a_).cpu().numpy().view(np.uint32)
            pt = d_tf[m].cpu().numpy().view(np.uint32)
            cnt = torch.bincount(term_of[m], minlength=self.vocab)
            off = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(cnt, 0)]).cpu().numpy().astype(np.uint64)
            fn = fieldnorm_ids[a_:b_]
            # Bm25Weight's average field length is searcher-wide: only the SUM of the segments' token counts enters a score, so the
            # last part takes the remainder of the whole's count and every score bit equals the one-segment index's
            last = b_ == n
            part_tokens = tokens - sum(p_.total_num_tokens for p_ in parts) if last else int(table[fn].sum())
            parts.append(Bm25Segment(off, pd, pt, fn, part_tokens))
            del m, cnt
        del d_doc, d_tf, df, term_of
        torch.cuda.empty_cache()
        return parts, cuts

    def multi_segment_leg(self, dev, one_segment_value, threads_n, depth):
        """The same 10 M documents as SIX segments: opened as one term-major resident layout, searched in one launch sequence per batch
        (csrc/bm25_index.cpp: bm25_upload_concatenated).  Hits must equal the one-segment index's (doc = base[segment] + doc)."""
        from nucliadb_amd import _lib
        from nucliadb_amd.bm25 import Bm25Searcher

        a, B, k = self.a, self.B, self.K
        t0 = time.time()
        parts, cuts = self.split_by_merge_policy(dev)
        split_s = time.time() - t0
        t0 = time.time()
        ms = Bm25Searcher.open(parts)
        open_s = time.time() - t0
        try:
            base = np.asarray(cuts[:-1], np.int64)
            self.search(0)
            want = (self.docaddr.copy(), self.count.copy(), self.total.copy(), self.score.copy())
            da, sc = np.zeros((B, k), np.uint64), np.zeros((B, k), np.float32)
            cn, tt, pp = np.zeros(B, np.uint32), np.zeros(B, np.uint64), np.zeros(B, np.uint64)
            _lib.check(self.L.nidx_gpu_bm25_search(ms._handle, self.prepared[0], self.offsets.ctypes.data, B, k, None, da.ctypes.data, sc.ctypes.data,
                                                   cn.ctypes.data, tt.ctypes.data, pp.ctypes.data))
            same_ids = 0
            for i in range(B):
                c = int(cn[i])
                g = base[(da[i, :c] >> np.uint64(32)).astype(np.int64)] + (da[i, :c] & np.uint64(0xFFFFFFFF)).astype(np.int64)
                same_ids += int(c == int(want[1][i]) and tt[i] == want[2][i] and np.array_equal(g, want[0][i, :c].astype(np.int64)) and
                                np.array_equal(sc[i, :c].view(np.uint32), want[3][i, :c].view(np.uint32)))
            # a sample against THE ORACLE'S searcher over the same six segments (searcher-wide statistics, per-segment collectors,
            # merge_fruits: oracle.Bm25Searcher) — the direct check; the one-segment comparison above is the extra
            oracle_same = oracle_n = None
            if self.a.cpu_queries > 0:
                from oracle import oracle as orc

                orc.build()
                osr = orc.Bm25Searcher([orc.Bm25Index(p_.term_offsets, p_.doc_ids, p_.tfs, p_.fieldnorm_ids, p_.total_num_tokens) for p_ in parts])
                oracle_n, oracle_same = min(64, B), 0
                for i in range(oracle_n):
                    wd, ws, _, wt, _ = osr.search_ex([(int(t), 0, 0, 1.0) for t in self.terms[0][i]], k, daat=True)
                    c = int(cn[i])
                    oracle_same += int(c == len(wd) and int(tt[i]) == wt and np.array_equal(da[i, :c], wd) and
                                       np.array_equal(sc[i, :c].view(np.uint32), ws.view(np.uint32)))
                if oracle_same != oracle_n:
                    FAILURES.append("bm25 multi-segment: %d of %d sampled queries differ from the oracle's multi-segment searcher" % (oracle_n - oracle_same, oracle_n))
            # two rounds of (this index, the one-segment index again as the control: clocks, allocator state and host threads as for the
            # leg) — the submitting Python threads make single one-second figures wander by 10-20 %; the better round of each is kept
            best, one_segment_value = None, 0.0
            for _round in range(2):
                r_ = self.timed_pipeline(ms, threads_n, depth)
                if best is None or r_[2] / r_[0] > best[2] / best[0]:
                    best = r_
                e1, n1, p1, _ = self.timed_pipeline(self.searcher, threads_n, depth)
                one_segment_value = max(one_segment_value, p1 / e1)
            elapsed, n_steps, postings, _ = best
            out = {"segments": [int(b_ - a_) for a_, b_ in zip(cuts[:-1], cuts[1:])], "value": postings / elapsed, "unit": "postings/s",
                   "queries_per_s": n_steps * B / elapsed, "ms_per_step": elapsed / n_steps * 1e3, "one_segment_value": one_segment_value, "one_segment_value_note": "the one-segment index timed again right after each round of this leg; the better of two rounds for both",
                   "ratio_to_one_segment": postings / elapsed / one_segment_value if one_segment_value else None,
                   "queries_identical_to_the_one_segment_index": same_ids, "queries": B, "split_s": split_s, "open_s": open_s,
                   "queries_identical_to_the_oracle_searcher": oracle_same, "oracle_sample": oracle_n,
                   "note": "the log-merge policy's shape (nidx/src/settings.rs:246-253); documents, ranks, score bits and totals must equal the "
                           "one-segment index's"}
            if same_ids != B:
                FAILURES.append("bm25 multi-segment: %d of %d queries return other hits than the one-segment index" % (B - same_ids, B))
            return out
        finally:
            ms.close()

    def oracle_index(self):
        from oracle import oracle as orc

        orc.build()
        return orc.Bm25Index(*self.corpus)

    def timed_block(self, rank, dev=None):
        """-> dict: value (postings/s end to end through the host-buffer entry point), roofline of the scoring kernel, cpu_baseline,
        parity of a sample against the oracle."""
        a, B, k = self.a, self.B, self.K
        from nucliadb_amd import _lib

        for i in range(max(1, a.warmup)):
            self.search(i)
        # the timed loop runs through the library's pipelined entry (nidx_gpu_bm25_search_submit / _wait): `threads` submitting threads
        # (the reference serves every request on a thread of its own, shard_search.rs:176-248) with `depth` batches in flight each —
        # the host side of a batch (clause weights, work list, staging, ~8 runtime calls) costs its thread more than the kernels cost
        # the device, and every ticket is planned and launched on a context of its own
        # twelve native submitting threads with one ticket each (the library allows sixteen tickets): measured round 6 on the bench batch,
        # 6 x 2: 198 G postings/s, 8 x 2: 207, 10 x 1: 213, 12 x 1: 220, 14 x 1: 214 — what a thread pays per batch (~60 us of planning and
        # runtime calls in submit, ~25 us in wait, the wake-up of a blocking synchronise) is what the extra threads hide
        depth = int(os.environ.get("NIDX_BENCH_BM25_DEPTH", "1"))
        threads_n = max(1, int(os.environ.get("NIDX_BENCH_BM25_THREADS", "12")))
        cpu0, thr0 = host_cpu()
        elapsed, n_steps, postings, post_per_batch = self.timed_pipeline(self.searcher, threads_n, depth)
        cpu1, thr1 = host_cpu()
        host_load = {"host_cores_busy": (cpu1 - cpu0) / max(elapsed, 1e-9), "cpu_quota_cores": cpu_quota_cores(),
                     "cgroup_throttled_periods": None if thr0 is None or thr1 is None else thr1 - thr0}
        one_thread = None
        if threads_n > 1:
            e1, n1, p1, _ = self.timed_pipeline(self.searcher, 1, depth)
            one_thread = {"postings_per_s": p1 / e1, "queries_per_s": n1 * B / e1, "ms_per_step": e1 / n1 * 1e3}
        # the scoring kernel's duration: launches issued one at a time (events around overlapped launches also span their waits)
        kernel_ms, sync_ms = [], []
        for i in range(8):
            t1 = time.perf_counter()
            self.search(i)
            sync_ms.append((time.perf_counter() - t1) * 1e3)
            kernel_ms.append(self.kernel_ms())
        k_ms = float(np.mean(kernel_ms))
        # the scoring launch also merges the slices of every query (kernels.h: Bm25FusedMerge) — the figure above is that launch; the same
        # batches with the merge in a launch of its own behind it (round 5's shape, NIDX_GPU_BM25_FUSED_MERGE=0): the scoring kernel alone
        alone_ms = None
        if os.environ.get("NIDX_GPU_BM25_FUSED_MERGE") is None:
            os.environ["NIDX_GPU_BM25_FUSED_MERGE"] = "0"
            try:
                am = []
                for i in range(8):
                    self.search(i)
                    am.append(self.kernel_ms())
                alone_ms = float(np.mean(am))
            finally:
                del os.environ["NIDX_GPU_BM25_FUSED_MERGE"]
        traffic, traffic_src = pmc_traffic("bm25", self.n_docs, self.vocab, B, k)
        alg = float(np.mean(post_per_batch)) * 8.0   # doc id (4 B) + the resident posting word tf | fieldnorm id << 24 (4 B)
        achieved = alg / (k_ms * 1e-3) / 1e9
        out = {
            "metric": "BM25 docs (postings) scored/sec", "value": postings / elapsed, "unit": "postings/s", "queries_per_s": n_steps * B / elapsed,
            "steps": n_steps, "ms_per_step": elapsed / n_steps * 1e3,
            "workload": "bm25: %d docs, vocab %d Zipf(1.0), %d queries x 3 Should terms from rank band [100,100k], k=%d" % (self.n_docs, self.vocab, B, k),
            "postings_in_index": int(self.corpus[0][-1]), "postings_per_batch": float(np.mean(post_per_batch)),
            "corpus_gen_s": self.gen_s, "open_s": self.open_s,
            "note": "value is end to end through the pipelined host-buffer entry points (nidx_gpu_bm25_search_submit / _wait: clauses in, hits out over "
                    "PCIe) from %d submitting thread(s) with %d batches in flight each; the corpus is resident in HBM" % (threads_n, depth),
            "submitting_threads_are": "Python threads (NIDX_BENCH_PYTHON_THREADS=1)" if native_host_driver() is None else
                                      "native threads of bench_native/host_driver.cpp (a client of include/nidx_gpu.h, like the reference's Rust host)",
            "submitting_threads": threads_n, "batches_in_flight": depth * threads_n, "host_load": host_load, "one_submitting_thread": one_thread,
            "synchronous_entry_ms_per_batch": float(np.mean(sync_ms)),
            "roofline": {"kernel": "bm25_stream_kernel: scoring + the merge of every query's slices in one launch", "bound": "hbm", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": alg, "kernel_ms": k_ms,
                         "scoring_alone": None if alone_ms is None else {
                             "kernel_ms": alone_ms, "frac": alg / (alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "note": "NIDX_GPU_BM25_FUSED_MERGE=0: the scoring kernel without the merge (bm25_merge_kernel then runs in a launch of its own "
                                     "behind it, ~16 us for this batch): the figure earlier rounds quoted"}},
            "cpu_baseline": None,
        }
        if rank == 0 and dev is not None and os.environ.get("NIDX_BENCH_BM25_SEGMENTS", "1") != "0":
            try:
                out["multi_segment"] = self.multi_segment_leg(dev, out["value"], threads_n, depth)
            except Exception as e:  # noqa: BLE001 — a side leg
                print("ERROR: bm25 multi-segment leg failed: %r" % (e,), file=sys.stderr)
                FAILURES.append("bm25 multi-segment leg failed: %r" % (e,))
        if rank == 0 and a.cpu_queries > 0:
            from oracle import oracle as orc

            term_offsets = self.corpus[0]
            oidx = self.oracle_index()
            threads = a.cpu_threads or min(64, os.cpu_count() or 1)
            nq = min(max(4 * threads, 256), B)
            self.search(0)
            got = (self.docaddr.copy(), self.score.copy(), self.count.copy(), self.total.copy())
            queries = [[(int(t), 0, 0, 1.0) for t in self.terms[0][i]] for i in range(nq)]
            done = int(sum(int(term_offsets[int(t) + 1] - term_offsets[int(t)]) for i in range(nq) for t in self.terms[0][i]))
            orc.bm25_search_daat_batch(oidx, queries[:threads], k, threads=threads)   # warm
            t1 = time.perf_counter()
            od, os_, oc, ot = orc.bm25_search_daat_batch(oidx, queries, k, threads=threads)
            dt = time.perf_counter() - t1
            out["cpu_baseline"] = {"value": done / dt, "unit": "postings/s", "queries_per_s": nq / dt, "cores": threads, "cpu_quota_cores": cpu_quota_cores(), "kind": "port",
                                   "sample": "%d queries of batch 0 over the same %d-doc index, oracle document-at-a-time BM25 (tantivy-style union of the clause cursors, no block-max pruning), one query per POSIX thread" % (nq, self.n_docs)}
            # bit parity of the same sample: doc ids, ranks, score bits, Count
            ok = 0
            for i in range(nq):
                c = int(got[2][i])
                ok += int(c == oc[i] and ot[i] == got[3][i] and np.array_equal(got[0][i, :c], od[i, :c]) and
                          np.array_equal(got[1][i, :c].view(np.uint32), os_[i, :c].view(np.uint32)))
            out["parity"] = {"queries": nq, "identical_ids_ranks_score_bits": ok, "status": "ok" if ok == nq else "MISMATCH",
                             "reference": "nidx_paragraph/src/reader.rs:244-348 (tantivy BM25, TopDocs)"}
            if ok != nq:
                FAILURES.append("bm25 block: device hits differ from the oracle's on %d of %d queries" % (nq - ok, nq))
        return out


def hybrid_block(a, L, dev, rank, h, qpool, bm, nfl, cpu_vector_qps, cpu_bm25_qps):
    """BASELINE.json configs[2]: cosine HNSW over N x 768 vectors + BM25 over N synthetic documents (document i owns vector i), a
    batch of hybrid queries (one vector + 3 keyword terms), fused with reciprocal rank fusion.  The vector batches go through the
    serving pipeline (`nfl` in flight, hits delivered to the host), the BM25 batches through the library's pipelined entries (two in flight,
    streams of their own), the fusion is nucliadb's Python-side step (batched in host C++ here: nidx_gpu_rank_fusion_rrf)."""
    from nucliadb_amd import _lib
    from nucliadb_amd.rank_fusion import rrf_fuse_batch

    B, k, d, kb = a.batch, a.k, a.dim, bm.K
    n_pool = qpool.shape[0]
    # The launch shape of the walks for a device they SHARE with the BM25 scorer: a walk workgroup normally holds 40 KiB of LDS and
    # <= 128 VGPRs per lane, four of them fill a CU and no BM25 workgroup fits beside them — the two pipelines took turns (2.6 M hybrid
    # queries/s).  With the <= 96-VGPR register class (tunable min_waves = 5) and a 2^12-slot visited table (vis_log2 = 12: 24 KiB per
    # walk; a walk that fills it raises its flag and is re-run exactly, like always) a BM25 workgroup is co-resident on every CU and its
    # instruction-bound rows run in the issue slots the latency-bound walks leave: 3.3 M (scripts/r5_ab.sh hybrid).  Results do not
    # depend on either knob.  NIDX_BENCH_HYBRID_SHAPE=0 keeps the default shape (comparison).
    shared_shape = os.environ.get("NIDX_BENCH_HYBRID_SHAPE", "1") != "0" and not os.environ.get("NIDX_BENCH_TUNABLES")
    if shared_shape:
        _lib.check(L.nidx_gpu_vector_set_tunable(h, b"min_waves", 5))
        _lib.check(L.nidx_gpu_vector_set_tunable(h, b"vis_log2", 12))
    retried_total = [0]
    p = _lib.VectorSearchParamsC(k, -1.0, 1, _lib.METHOD_HNSW)
    host_out = [(np.zeros((B, k), np.uint32), np.zeros((B, k), np.float32), np.zeros(B, np.uint32)) for _ in range(nfl)]
    # the fusion of batch i (native host code, the GIL released) runs on a second host thread while batch i + 1 is waited for: its
    # inputs are double-buffered, its result is collected one step later
    from concurrent.futures import ThreadPoolExecutor

    fuser = ThreadPoolExecutor(max_workers=1)
    # BM25 batches in flight: their launches wait for workgroup slots behind the walks that fill the device, so a scoring launch that
    # takes 0.06 ms alone completes 0.2 - 0.3 ms after its submit; with three tickets outstanding most of that latency is hidden
    bm_depth = int(os.environ.get("NIDX_BENCH_HYBRID_BM25_DEPTH", "3"))
    ring = bm_depth + 4   # the keyword leg runs at most three batches ahead of the fusion (the queue below holds two results)
    bm_out = [(np.zeros((B, kb), np.uint64), np.zeros((B, kb), np.float32), np.zeros(B, np.uint32)) for _ in range(ring)]
    fusing = []   # the job of the previous batch
    low32 = np.uint64(0xFFFFFFFF)

    def fuse(bo, vo):
        # keyword list first, like nucliadb's fuse({"keyword": .., "semantic": ..}); ids = document numbers
        return rrf_fuse_batch([(bo[0] & low32, bo[2], 1.0, bo[1]), (vo[0].astype(np.uint64), vo[2], 1.0, None)], k=60.0, window=k)

    in_flight = []
    t_parts = {"vector_submit": 0.0, "bm25_wait": 0.0, "vector_wait": 0.0, "fusion": 0.0}
    kernel_ms = []
    import queue
    import threading

    def submit(i):
        t = C.c_uint64(0)
        _lib.check(L.nidx_gpu_vector_search_submit(h, qpool[i % n_pool].data_ptr(), B, d, C.byref(p), None, C.byref(t)))
        in_flight.append((t.value, i % nfl))

    def keyword_leg(n_steps, results):
        # The keyword search of a hybrid request is a request thread of its own in the reference (src/searcher/shard_search.rs:139-153: the
        # text / paragraph search and the vector search of one request run side by side): here one thread that keeps `bm_depth` BM25 batches
        # in flight through nidx_gpu_bm25_search_submit / _wait (the calls release the interpreter lock) and hands every finished batch over
        try:
            pend, nxt = [], 0
            for _ in range(n_steps):
                while len(pend) < bm_depth and nxt < n_steps:
                    pend.append((nxt, bm.submit(nxt)))
                    nxt += 1
                bi, tk = pend.pop(0)
                bm.wait(tk, out=bm_out[bi % ring])
                kernel_ms.append(bm.kernel_ms())
                results.put(bi)
        except BaseException as e:  # noqa: BLE001 — handed to the main loop
            results.put(e)

    def run_region(n_steps):
        """n_steps hybrid batches -> the fused result of the last one"""
        results = queue.Queue(maxsize=2)
        leg = threading.Thread(target=keyword_leg, args=(n_steps, results), daemon=True)
        leg.start()
        fused = None
        for i in range(n_steps):
            t0 = time.perf_counter()
            while len(in_flight) < nfl and (i + len(in_flight)) < n_steps:
                submit(i + len(in_flight))
            t1 = time.perf_counter()
            tk, j = in_flight.pop(0)
            hv_, hs_, hc_ = host_out[j]
            r_ = C.c_uint32(0)
            _lib.check(L.nidx_gpu_vector_search_wait(h, tk, None, None, hv_.ctypes.data, hs_.ctypes.data, hc_.ctypes.data, C.byref(r_)))
            retried_total[0] += r_.value
            t2 = time.perf_counter()
            bi = results.get()
            if isinstance(bi, BaseException):
                raise bi
            t3 = time.perf_counter()
            job = fuser.submit(fuse, bm_out[bi % ring], host_out[j])
            fused = fusing.pop(0).result() if fusing else None   # batch i - 1, fused while this batch was waited for
            fusing.append(job)
            if i + 1 == n_steps:   # the last batch of a region is fused inside it
                fused = fusing.pop(0).result()
            t4 = time.perf_counter()
            t_parts["vector_submit"] += t1 - t0
            t_parts["vector_wait"] += t2 - t1
            t_parts["bm25_wait"] += t3 - t2
            t_parts["fusion"] += t4 - t3
        leg.join()
        return fused

    run_region(max(2, a.warmup))
    torch.cuda.synchronize()
    # probe, then a region of at least min_timed_s
    n_steps = max(a.steps, 8)
    t0 = time.perf_counter()
    run_region(n_steps)
    probe = time.perf_counter() - t0
    n_steps = max(n_steps, int(np.ceil(n_steps * min(a.min_timed_s, 1.0) / max(probe, 1e-6))))
    for k_ in t_parts:
        t_parts[k_] = 0.0
    kernel_ms.clear()
    retried_total[0] = 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_region(n_steps)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    out = {
        "metric": "hybrid queries/sec (768-dim cosine HNSW k=%d + BM25 k=%d over the same documents, reciprocal rank fusion)" % (k, kb),
        "value": B * n_steps / elapsed, "unit": "hybrid queries/s", "steps": n_steps, "ms_per_step": elapsed / n_steps * 1e3,
        "workload": "hybrid: HNSW over the timed shard + BM25 over %d docs (vocab %d), batch=%d, RRF k=60, results (fused ids + scores) on the host" % (bm.n_docs, bm.vocab, B),
        "vector_batches_in_flight": nfl, "bm25_kernel_ms": float(np.mean(kernel_ms)),
        "walk_launch_shape": "min_waves = 5 (<= 96 VGPRs), vis_log2 = 12: a BM25 workgroup is co-resident with four walks on every CU" if shared_shape else "the library's default",
        "vector_queries_re_run_exactly": retried_total[0],
        "fusion": "nidx_gpu_rank_fusion_rrf (native, host) on a second host thread, one batch behind the searches; ms_per_step_parts.fusion = what the main loop still waits for it",
        "keyword_leg": "a host thread of its own (the reference runs the keyword and the vector search of a request side by side, shard_search.rs:139-153): %d BM25 batches in flight; ms_per_step_parts.bm25_wait = what the main loop still waits for it" % bm_depth,
        "ms_per_step_parts": {kk_: v / n_steps * 1e3 for kk_, v in t_parts.items()},
        "cpu_baseline": None,
    }
    if cpu_vector_qps and cpu_bm25_qps:
        nq = min(B, 512)
        t1 = time.perf_counter()
        for _ in range(4):
            rrf_fuse_batch([(bm.docaddr[:nq] & np.uint64(0xFFFFFFFF), bm.count[:nq], 1.0, bm.score[:nq]),
                            (host_out[0][0][:nq].astype(np.uint64), host_out[0][2][:nq], 1.0, None)], k=60.0, window=k)
        t_fuse = (time.perf_counter() - t1) / 4
        out["cpu_baseline"] = {"value": nq / (nq / cpu_vector_qps + nq / cpu_bm25_qps + t_fuse), "unit": "hybrid queries/s", "kind": "port",
                               "cores": a.cpu_threads or min(64, os.cpu_count() or 1),
                               "sample": "composed on this box's cores: the oracle's flat HNSW leg (%.0f queries/s) then its document-at-a-time BM25 leg (%.0f queries/s), "
                                         "both measured above on their bounded samples, then the rank fusion of %d queries timed here" % (cpu_vector_qps, cpu_bm25_qps, nq)}
    return out


def bench_hybrid(a, L, dev, rank, world):
    """--workload hybrid: the hybrid block on its own (generates the shard, builds the graph, opens the BM25 index)."""
    from nucliadb_amd import _lib

    n, d, B = a.n_vectors, a.dim, a.batch
    kind = "uniform" if a.corpus == "uniform" else "clustered"
    x = gen_corpus(kind, n, d, dev, 1234567890 + rank)
    qpool = gen_queries(kind, x, 4, B, d, dev, 2)
    cfg = _lib.VectorConfigC(d, 1, 0, 0)
    cseg = _lib.VectorSegmentC(x.data_ptr(), d * 4, n, None, n, None, 0, 0, None, 0, None, None)
    h = C.c_void_p()
    _lib.check(L.nidx_gpu_vector_open(C.byref(cfg), C.byref(cseg), 1, C.byref(h)))
    del x
    torch.cuda.empty_cache()
    t0 = time.time()
    if a.build_ef_upper > 1:
        _lib.check(L.nidx_gpu_vector_set_tunable(h, b"build_ef_upper", a.build_ef_upper))
    _lib.check(L.nidx_gpu_vector_build_hnsw(h, 0, 2))
    build_s = time.time() - t0
    if a.ef_upper > 1:
        _lib.check(L.nidx_gpu_vector_set_tunable(h, b"ef_upper", a.ef_upper))
    for kv in filter(None, os.environ.get("NIDX_BENCH_TUNABLES", "").split(",")):   # A/B runs: name=value[,name=value]
        name, _, val = kv.partition("=")
        _lib.check(L.nidx_gpu_vector_set_tunable(h, name.strip().encode(), int(val)))
    bm = Bm25Bench(a, L, dev, rank, n)
    blk = hybrid_block(a, L, dev, rank, h, qpool, bm, max(1, a.batches_in_flight), None, None)
    bm.close()
    L.nidx_gpu_vector_close(h)
    if rank == 0:
        print(json.dumps({
            "metric": blk["metric"], "value": world * blk["value"], "unit": blk["unit"], "n_gpus": world, "steps": blk["steps"], "warmup": a.warmup,
            "ms_per_step": blk["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": dict({kk_: v for kk_, v in blk.items() if kk_ not in ("metric", "value", "unit", "steps", "ms_per_step", "cpu_baseline")},
                           hnsw_build_s=build_s),
            "roofline": None, "cpu_baseline": blk["cpu_baseline"]}))


def bench_bm25(a, L, dev, rank, world):
    """--workload bm25: BASELINE.json's second metric on its own line."""
    bm = Bm25Bench(a, L, dev, rank, a.n_docs)
    blk = bm.timed_block(rank, dev)
    bm.close()
    if rank == 0:
        print(json.dumps({
            "metric": blk["metric"], "value": blk["value"], "unit": blk["unit"], "n_gpus": world, "steps": blk["steps"], "warmup": a.warmup,
            "ms_per_step": blk["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {kk_: v for kk_, v in blk.items() if kk_ not in ("metric", "value", "unit", "steps", "ms_per_step", "roofline", "cpu_baseline")},
            "roofline": blk["roofline"], "cpu_baseline": blk["cpu_baseline"]}))


def rabitq_segments_leg(a, L, xh, qpool, nfl):
    """The same vectors as `--rabitq-segments` quantized segments (Searcher::_search walks every open segment, searcher.rs:149-199): all
    segments' RaBitQ walks of a batch in ONE launch (rabitq_hnsw3_segments_kernel) + the entry-mode walks + Fssc, pipelined; checked
    against a launch per segment (NIDX_GPU_SEGMENT_LAUNCHES)."""
    from nucliadb_amd import _lib

    n, d, B, k, S = a.n_vectors, a.dim, a.batch, a.k, a.rabitq_segments
    bounds = [n * i // S for i in range(S + 1)]
    cfg = _lib.VectorConfigC(d, 0, 0, 0, 0)
    csegs = (_lib.VectorSegmentC * S)()
    for s_ in range(S):
        rows = xh[bounds[s_]:bounds[s_ + 1]]
        csegs[s_] = _lib.VectorSegmentC(rows.ctypes.data, d * 4, rows.shape[0], None, rows.shape[0], None, 0, 0, None, 0, None, None)
    h = C.c_void_p()
    _lib.check(L.nidx_gpu_vector_open(C.byref(cfg), csegs, S, C.byref(h)))
    t0 = time.time()
    for s_ in range(S):
        _lib.check(L.nidx_gpu_vector_build_hnsw(h, s_, 2 + s_))
        _lib.check(L.nidx_gpu_vector_quantize(h, s_))
    build_s = time.time() - t0
    _lib.check(L.nidx_gpu_vector_set_tunable(h, b"pipeline_depth", max(nfl, 4)))
    p = _lib.VectorSearchParamsC(k, -1.0, 1, _lib.METHOD_RABITQ_HNSW)
    n_pool = qpool.shape[0]
    hout = [(np.zeros((B, k), np.uint32), np.zeros((B, k), np.uint32), np.zeros((B, k), np.float32), np.zeros(B, np.uint32)) for _ in range(nfl)]
    tick = []

    def wait_oldest():
        t_, j_ = tick.pop(0)
        o = hout[j_]
        _lib.check(L.nidx_gpu_vector_search_wait(h, t_, o[0].ctypes.data, None, o[1].ctypes.data, o[2].ctypes.data, o[3].ctypes.data, None))

    def step(i):
        if len(tick) == nfl:
            wait_oldest()
        t = C.c_uint64(0)
        _lib.check(L.nidx_gpu_vector_search_submit(h, qpool[i % n_pool].data_ptr(), B, d, C.byref(p), None, C.byref(t)))
        tick.append((t.value, i % nfl))

    def run(n_steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n_steps):
            step(i)
        while tick:
            wait_oldest()
        return time.perf_counter() - t0

    run(nfl + 1)
    n_steps = max(a.steps, 3 * nfl)
    el = run(n_steps)
    last = [x.copy() for x in hout[(n_steps - 1) % nfl]]
    os.environ["NIDX_GPU_SEGMENT_LAUNCHES"] = "1"
    try:
        run(1)
        el_per_segment = run(n_steps)
        other = [x.copy() for x in hout[(n_steps - 1) % nfl]]
    finally:
        del os.environ["NIDX_GPU_SEGMENT_LAUNCHES"]
    L.nidx_gpu_vector_close(h)
    same = all(np.array_equal(x.view(np.uint32), y.view(np.uint32)) for x, y in zip(last, other))
    if not same:
        FAILURES.append("rabitq segments: one launch for all segments delivered other hits than a launch per segment")
    return {"segments": S, "records_per_segment": n // S, "queries_per_s": B * n_steps / el, "ms_per_step": el / n_steps * 1e3,
            "launch_per_segment_queries_per_s": B * n_steps / el_per_segment, "one_launch_equals_a_launch_per_segment": same,
            "batches_in_flight": nfl, "builds_and_quantize_s": build_s,
            "entry": "nidx_gpu_vector_search_submit / _wait, METHOD_RABITQ_HNSW: every segment's walks of a batch in one launch (rabitq_hnsw3_segments_kernel), "
                     "entry-mode walks in one launch (hnsw_search_segments_kernel), Fssc on the device; oracle parity of this path: tests/test_serving_gpu.py"}


def bench_rabitq(a, L, dev, rank, world):
    """The RaBitQ arm of a Dot index: 1-bit codes + popcount estimates drive the HNSW walk with
    ef = min(100 k, 2000), the ef neighbours are re-ranked with the raw rows under the error bound, closest_up_nodes
    finishes on the raw query.  Clustered unit vectors (the reference's recall recipe), so recall means something."""
    from nucliadb_amd import _lib

    n, d, B, k = a.n_vectors, a.dim, a.batch, a.k
    g = torch.Generator(device=dev)
    g.manual_seed(1234567890 + rank)

    def unit(*shape):
        v = torch.rand(shape, generator=g, device=dev, dtype=torch.float32) * 2 - 1
        return v / v.norm(dim=-1, keepdim=True)

    per = 160
    n_centres = (n + per - 1) // per
    centres = unit(n_centres, d)
    radius = torch.where(torch.arange(per, device=dev) < per // 2, 0.1, 0.3).repeat(n_centres)[:n]
    x = centres.repeat_interleave(per, dim=0)[:n] + radius[:, None] * unit(n, d)
    x = x / x.norm(dim=1, keepdim=True)
    x = x[torch.randperm(n, generator=g, device=dev)]
    gq = torch.Generator(device=dev)
    gq.manual_seed(2)
    n_pool = 4
    base = x[torch.randint(0, n, (n_pool * B,), generator=gq, device=dev)]
    noise = torch.rand((n_pool * B, d), generator=gq, device=dev, dtype=torch.float32) * 2 - 1
    q = base + 0.2 * noise / noise.norm(dim=1, keepdim=True)
    qpool = (q / q.norm(dim=1, keepdim=True)).reshape(n_pool, B, d).contiguous()
    xh = x.cpu().numpy()
    del x, base, noise, q
    torch.cuda.empty_cache()
    cfg = _lib.VectorConfigC(d, 0, 0, 0, 0)
    cseg = _lib.VectorSegmentC(xh.ctypes.data, d * 4, n, None, n, None, 0, 0, None, 0, None, None)
    h = C.c_void_p()
    _lib.check(L.nidx_gpu_vector_open(C.byref(cfg), C.byref(cseg), 1, C.byref(h)))
    t0 = time.time()
    _lib.check(L.nidx_gpu_vector_build_hnsw(h, 0, 2))
    build_s = time.time() - t0
    t0 = time.time()
    _lib.check(L.nidx_gpu_vector_quantize(h, 0))
    quant_s = time.time() - t0
    ov = torch.zeros((B, k), dtype=torch.int32, device=dev)
    os_ = torch.zeros((B, k), dtype=torch.float32, device=dev)
    oc = torch.zeros((B,), dtype=torch.int32, device=dev)
    st = torch.zeros((B, 8), dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def search(qb, m, with_stats=False):
        p = _lib.VectorSearchParamsC(k, -1.0, 1, m)
        _lib.check(L.nidx_gpu_vector_segment_search_device(h, 0, qb.data_ptr(), B, C.byref(p), None, ov.data_ptr(), os_.data_ptr(),
                                                           oc.data_ptr(), st.data_ptr() if with_stats else None, stream))

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    def timed(m):
        for i in range(a.warmup):
            search(qpool[i % n_pool], m)
        ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps)]
        ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps)]
        barrier()
        t0 = time.perf_counter()
        for i in range(a.steps):
            ev0[i].record()
            search(qpool[i % n_pool], m)
            ev1[i].record()
        barrier()
        el = time.perf_counter() - t0
        return el, float(np.mean([ev0[i].elapsed_time(ev1[i]) for i in range(a.steps)]))

    elapsed, k_ms = timed(_lib.METHOD_RABITQ_HNSW)
    # the same batches through the serving pipeline (nidx_gpu_vector_search_submit / _wait, device-resident queries in, hits in host arrays
    # out), three in flight: a walk is a chain of ~1 100 dependent expansions and one launch of 1 024 walks leaves three quarters of the
    # wave slots empty — the next batches' walks take them
    nfl_p = max(1, a.batches_in_flight)
    _lib.check(L.nidx_gpu_vector_set_tunable(h, b"pipeline_depth", max(nfl_p, 4)))
    p_rq = _lib.VectorSearchParamsC(k, -1.0, 1, _lib.METHOD_RABITQ_HNSW)
    hout = [(np.zeros((B, k), np.uint32), np.zeros((B, k), np.float32), np.zeros(B, np.uint32)) for _ in range(nfl_p)]
    tick = []

    def pstep(i):
        if len(tick) == nfl_p:
            t_, j_ = tick.pop(0)
            _lib.check(L.nidx_gpu_vector_search_wait(h, t_, None, None, hout[j_][0].ctypes.data, hout[j_][1].ctypes.data, hout[j_][2].ctypes.data, None))
        t = C.c_uint64(0)
        _lib.check(L.nidx_gpu_vector_search_submit(h, qpool[i % n_pool].data_ptr(), B, d, C.byref(p_rq), None, C.byref(t)))
        tick.append((t.value, i % nfl_p))

    def pdrain():
        while tick:
            t_, j_ = tick.pop(0)
            _lib.check(L.nidx_gpu_vector_search_wait(h, t_, None, None, hout[j_][0].ctypes.data, hout[j_][1].ctypes.data, hout[j_][2].ctypes.data, None))

    for i in range(nfl_p + 1):
        pstep(i)
    pdrain()
    n_pipe = max(a.steps, 3 * nfl_p)
    barrier()
    t0 = time.perf_counter()
    for i in range(n_pipe):
        pstep(i)
    pdrain()
    barrier()
    pipe_elapsed = time.perf_counter() - t0
    pipe_last = hout[(n_pipe - 1) % nfl_p][0].copy()
    pipe_last_batch = (n_pipe - 1) % n_pool
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    _, exact_hnsw_ms = timed(_lib.METHOD_HNSW)
    # counters + recall
    search(qpool[0], _lib.METHOD_RABITQ_HNSW, with_stats=True)
    torch.cuda.synchronize()
    s = st.cpu().numpy().astype(np.int64)
    got = ov.cpu().numpy().copy()
    flags = int(np.bitwise_or.reduce(s[:, 3]))
    if pipe_last_batch != 0:
        search(qpool[pipe_last_batch], _lib.METHOD_RABITQ_HNSW)
        torch.cuda.synchronize()
        pipe_same = bool(np.array_equal(pipe_last, ov.cpu().numpy().view(np.uint32)))
        search(qpool[0], _lib.METHOD_RABITQ_HNSW, with_stats=True)
        torch.cuda.synchronize()
    else:
        pipe_same = bool(np.array_equal(pipe_last, got.view(np.uint32)))
    if not pipe_same:
        FAILURES.append("rabitq: submit / wait delivered other hits than the device entry")
    rec_len = d // 8 + 8
    alg = float((s[:, 0] * rec_len + s[:, 1] * 256 + s[:, 2] * 4 * d).sum())
    search(qpool[0], _lib.METHOD_HNSW)
    torch.cuda.synchronize()
    got_exact_hnsw = ov.cpu().numpy().copy()
    search(qpool[0], _lib.METHOD_BRUTE_FORCE)
    torch.cuda.synchronize()
    exact = ov.cpu().numpy().copy()
    rq = min(a.recall_queries or B, B)
    recall = float(np.mean([len(set(got[i]) & set(exact[i])) / k for i in range(rq)]))
    recall_exact_hnsw = float(np.mean([len(set(got_exact_hnsw[i]) & set(exact[i])) / k for i in range(rq)]))
    cpu = None
    if rank == 0 and a.cpu_queries > 0:
        from concurrent.futures import ThreadPoolExecutor

        from oracle import oracle as orc

        orc.build()
        glen, nedges, qlen = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        L.nidx_gpu_vector_serialize_hnsw(h, 0, None, 0, C.byref(glen), None, 0, C.byref(nedges))
        graph = np.zeros(glen.value, np.uint8)
        edges = np.zeros(max(1, nedges.value), np.float32)
        L.nidx_gpu_vector_serialize_hnsw(h, 0, graph.ctypes.data, glen.value, C.byref(glen), edges.ctypes.data, nedges.value, C.byref(nedges))
        L.nidx_gpu_vector_serialize_quantized(h, 0, None, 0, C.byref(qlen))
        quant = np.zeros(qlen.value, np.uint8)
        L.nidx_gpu_vector_serialize_quantized(h, 0, quant.ctypes.data, quant.size, C.byref(qlen))
        oseg = orc.Segment(xh, similarity=orc.SIM_DOT, order=orc.ORDER_HASWELL, graph=orc.Hnsw.deserialize_v2(graph, edges[: nedges.value]),
                           quantized=quant.reshape(n, rec_len))
        qs = qpool[0].cpu().numpy()
        threads = a.cpu_threads or min(64, os.cpu_count() or 1)
        nq = min(a.cpu_queries, B)
        with ThreadPoolExecutor(threads) as ex:
            list(ex.map(lambda i: oseg.hnsw_search(qs[i], k), range(min(threads, nq))))
            t0 = time.perf_counter()
            res = list(ex.map(lambda i: oseg.hnsw_search(qs[i], k), range(nq)))
            dt = time.perf_counter() - t0
        same = int(sum(np.array_equal(res[i][0], got[i][: len(res[i][0])]) for i in range(nq)))
        cpu = {"value": nq / dt, "unit": "queries/s", "cores": threads, "cpu_quota_cores": cpu_quota_cores(), "kind": "port",
               "sample": "%d queries of the same batch, oracle RaBitQ HNSW over the device-built graph and codes, one query per thread; "
                         "%d/%d id lists identical to the device's (the timed baseline sums in AVX2 order, the device in WAVE64 order: a near-tie "
                         "of the exact re-rank may flip; the parity tests run the oracle in WAVE64 order and are bit-exact)" % (nq, same, nq)}
    L.nidx_gpu_vector_close(h)
    seg_leg = rabitq_segments_leg(a, L, xh, qpool, nfl_p) if rank == 0 and a.rabitq_segments > 1 else None
    if rank == 0:
        achieved = alg / (k_ms * 1e-3) / 1e9
        print(json.dumps({
            "metric": "queries/sec (%d-dim dot k-NN, RaBitQ HNSW: ef=min(100k,2000) on 1-bit codes + raw re-rank, k=%d)" % (d, k),
            "value": world * B * a.steps / elapsed, "unit": "queries/s (each against one %d-vector shard)" % n, "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64 popcount + f32", "data": "synthetic",
            "config": {"workload": "rabitq: %d x %d-dim dot (clustered unit vectors), k=%d, batch=%d queries, 1 shard per GPU" % (n, d, k, B),
                       "recall_at_%d" % k: recall, "recall_at_%d_exact_hnsw_ef30" % k: recall_exact_hnsw,
                       "exact_hnsw_ms_per_batch": exact_hnsw_ms, "estimates_per_query": float(s[:, 0].mean()),
                       "expansions_per_query": float(s[:, 1].mean()), "rows_reranked_per_query": float(s[:, 2].mean()),
                       "kernel_flags": flags, "hnsw_build_s": build_s, "quantize_s": quant_s,
                       "pipelined_queries_per_s": world * B * n_pipe / pipe_elapsed, "pipelined_batches_in_flight": nfl_p,
                       "pipelined_frac_of_hbm_peak": alg * n_pipe / pipe_elapsed / 1e9 / HBM_PEAK_GBS,
                       "pipelined_equals_one_launch_at_a_time": pipe_same, "segments": seg_leg,
                       "walk_kernel": ("rabitq_hnsw2_kernel (two waves per query: the fetcher expands the predicted next candidate while the controller admits)"
                                       if os.environ.get("NIDX_GPU_RABITQ_WAVES") == "2" else "rabitq_hnsw_kernel (one wave per query, rounds 1-4)"
                                       if os.environ.get("NIDX_GPU_RABITQ_PIPE") == "0" else
                                       "rabitq_hnsw3_kernel (one wave per query; the predicted next expansion's loads are in flight under the admissions)"),
                       "expansions_with_loads_in_flight_under_the_admissions_per_query": (float(s[:, 5].mean()) if os.environ.get("NIDX_GPU_RABITQ_WAVES") != "2"
                                                                                           and os.environ.get("NIDX_GPU_RABITQ_PIPE") != "0" else None),
                       "neighbours_asked_of_memory_per_query": (float(s[:, 4].mean()) if os.environ.get("NIDX_GPU_RABITQ_WAVES") != "2"
                                                                and os.environ.get("NIDX_GPU_RABITQ_PIPE") != "0" else None),
                       "cycles_per_query": ({"admission": float(s[:, 6].mean()), "total": float(s[:, 7].mean())}
                                            if os.environ.get("NIDX_GPU_RABITQ_WAVES") != "2" and os.environ.get("NIDX_GPU_RABITQ_PIPE") != "0" else
                                            {"pop_edge_visited": float(s[:, 4].mean()), "estimates": float(s[:, 5].mean()),
                                             "admission": float(s[:, 6].mean()), "total": float(s[:, 7].mean())} if os.environ.get("NIDX_GPU_RABITQ_WAVES") != "2" else
                                            {"fetcher_speculative_fetches": float(((s[:, 4] & 0xFFFF) << 8).mean()),
                                             "controller_predict_pop_and_waiting_for_the_fetcher": float((((s[:, 4] >> 16) & 0xFFFF) << 8).mean()), "admission": float(s[:, 6].mean()),
                                             "total": float(s[:, 7].mean())}),
                       "speculated_expansions_confirmed_per_query": None if os.environ.get("NIDX_GPU_RABITQ_WAVES") != "2" else float((s[:, 5] & 0xFFFF).mean()),
                       "expansions_with_edge_record_held_per_query": None if os.environ.get("NIDX_GPU_RABITQ_WAVES") != "2" else float(((s[:, 5] >> 16) & 0xFFFF).mean())},
            "roofline": {"kernel": "rabitq walk kernel + hnsw_search_kernel (entry mode)", "bound": "hbm", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "algorithmic_bytes_per_launch": alg, "kernel_ms": k_ms},
            "cpu_baseline": cpu}))


def cpu_baseline(a, L, h, x_host, q0, q1):
    """Exact-scan workloads: the oracle's brute_force_search (segment.rs:569-623 restated, AVX2-shaped sums) over the same
    shard, one query per POSIX thread, on a bounded sample (one query per thread)."""
    from oracle import oracle as orc

    orc.build()
    n, d, k = a.n_vectors, a.dim, a.k
    threads = a.cpu_threads or min(64, os.cpu_count() or 1)
    oseg = orc.Segment(x_host, similarity=orc.SIM_COSINE, order=orc.ORDER_HASWELL)
    qs = np.vstack([q0, q1])[:threads]
    t0 = time.perf_counter()
    oseg.brute_force_batch(qs, k, threads=threads)
    dt = time.perf_counter() - t0
    return {"value": qs.shape[0] / dt, "unit": "queries/s", "cores": threads, "cpu_quota_cores": cpu_quota_cores(), "kind": "port",
            "sample": "%d queries of the same batch over the same %d x %d shard, oracle brute force (C restatement of the reference "
                      "algorithm, AVX2-shaped f32 sums), one query per thread" % (qs.shape[0], n, d)}


if __name__ == "__main__":
    main()
    if FAILURES:   # the line above was printed; a parity break must not look like a green run
        sys.stdout.flush()
        sys.exit(1)
