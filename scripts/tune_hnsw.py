#!/usr/bin/env python3
"""Sweep the launch-shape knobs of hnsw_search_kernel on one device-built graph (tuning aid, GPU box)."""
import argparse, ctypes as C, itertools, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from nucliadb_amd import _lib

p = argparse.ArgumentParser()
p.add_argument("--n", type=int, default=1_000_000)
p.add_argument("--d", type=int, default=768)
p.add_argument("--batch", type=int, default=1024)
p.add_argument("--k", type=int, default=10)
p.add_argument("--reps", type=int, default=8)
p.add_argument("--corpus", default="uniform")
p.add_argument("--grid", default="waves_per_query=4,2;eval_rows=4,2;min_waves=2,4;vis_log2=13,12")
a = p.parse_args()
L = _lib.lib()
dev = torch.device("cuda", 0)
import bench
x = bench.gen_corpus(a.corpus, a.n, a.d, dev, 1234567890)
q = bench.gen_queries(a.corpus, x, 4, a.batch, a.d, dev, 2)
cfg = _lib.VectorConfigC(a.d, 1, 0, 0)
seg = _lib.VectorSegmentC(x.data_ptr(), a.d * 4, a.n, None, a.n, None, 0, 0, None, 0, None, None)
h = C.c_void_p()
_lib.check(L.nidx_gpu_vector_open(C.byref(cfg), C.byref(seg), 1, C.byref(h)))
del x
t0 = time.time(); _lib.check(L.nidx_gpu_vector_build_hnsw(h, 0, 2)); print("build_s", time.time() - t0, flush=True)
B, k = a.batch, a.k
ov = torch.zeros((B, k), dtype=torch.int32, device=dev); os_ = torch.zeros((B, k), device=dev); oc = torch.zeros(B, dtype=torch.int32, device=dev)
st = torch.zeros((B, 8), dtype=torch.int32, device=dev)
params = _lib.VectorSearchParamsC(k, -1.0, 1, _lib.METHOD_HNSW)
stream = torch.cuda.current_stream().cuda_stream
def run(qb, stats=False):
    _lib.check(L.nidx_gpu_vector_segment_search_device(h, 0, qb.data_ptr(), B, C.byref(params), None, ov.data_ptr(), os_.data_ptr(), oc.data_ptr(), st.data_ptr() if stats else None, stream))
names, vals = [], []
for part in a.grid.split(";"):
    n_, v_ = part.split("="); names.append(n_); vals.append([int(t) for t in v_.split(",")])
for combo in itertools.product(*vals):
    for n_, v_ in zip(names, combo):
        _lib.check(L.nidx_gpu_vector_set_tunable(h, n_.encode(), v_))
    run(q[0]); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(a.reps): run(q[r % 4])
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.reps
    run(q[0], stats=True); torch.cuda.synchronize()
    s = st.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    byt = float((s[:, 0] * 4 * a.d + s[:, 1] * 256).sum())
    cyc = s[:, 4:8].mean(0)
    print(dict(zip(names, combo)), "ms=%.3f qps=%.0f GB/s=%.0f frac=%.3f evals=%.0f exp=%.1f flags=%d cyc(ctl,EDGE_HITS,ins,total)=%s us_total=%.0f" % (
        ms, B / ms * 1e3, byt / ms / 1e6, byt / ms / 1e6 / 8000, s[:, 0].mean(), s[:, 1].mean(), int(np.bitwise_or.reduce(s[:, 3])),
        np.round(cyc).astype(int).tolist(), cyc[3] / 100.0), flush=True)
    tot = s[:, 7].astype(np.float64)
    print("   total cycles per query: min %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f ; evals p50 %.0f p99 %.0f max %.0f ; exp p50 %.0f max %.0f" % (
        tot.min(), np.percentile(tot, 50), np.percentile(tot, 90), np.percentile(tot, 99), tot.max(), np.percentile(s[:, 0], 50),
        np.percentile(s[:, 0], 99), s[:, 0].max(), np.percentile(s[:, 1], 50), s[:, 1].max()), flush=True)
L.nidx_gpu_vector_close(h)
