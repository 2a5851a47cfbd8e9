"""Round 6 measurement: what one submitting thread pays per BM25 batch (submit call, wait call, Python glue)."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    sys.argv = [sys.argv[0], "--workload", "bm25", "--cpu-queries", "0"]
    a = bench.parse()
    import torch

    from nucliadb_amd import _lib

    L = _lib.lib()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    bm = bench.Bm25Bench(a, L, dev, 0, a.n_docs, n_pool=32)
    B, k = bm.B, bm.K
    opt = _lib.Bm25SearchOptionsC()
    opt.k, opt.order_field = k, -1
    zero64 = np.zeros(1, np.uint64)
    opt.term_set_offsets = opt.phrase_offsets = opt.subquery_offsets = zero64.ctypes.data
    docaddr, score = np.zeros((B, k), np.uint64), np.zeros((B, k), np.float32)
    count, total, post = np.zeros(B, np.uint32), np.zeros(B, np.uint64), np.zeros(B, np.uint64)
    h = bm.searcher._handle
    for depth in (1, 2, 4):
        pending = []
        ts, tw, n = 0.0, 0.0, 0
        t0 = time.perf_counter()
        for i in range(400):
            t = C.c_uint64(0)
            t1 = time.perf_counter()
            _lib.check(L.nidx_gpu_bm25_search_submit(h, bm.prepared[i % 32], bm.offsets.ctypes.data, B, C.byref(opt), C.byref(t)))
            t2 = time.perf_counter()
            ts += t2 - t1
            pending.append(t.value)
            if len(pending) >= depth:
                t3 = time.perf_counter()
                _lib.check(L.nidx_gpu_bm25_search_wait(h, pending.pop(0), docaddr.ctypes.data, score.ctypes.data, count.ctypes.data, total.ctypes.data, post.ctypes.data))
                tw += time.perf_counter() - t3
                n += 1
        while pending:
            _lib.check(L.nidx_gpu_bm25_search_wait(h, pending.pop(0), docaddr.ctypes.data, score.ctypes.data, count.ctypes.data, total.ctypes.data, post.ctypes.data))
        el = time.perf_counter() - t0
        print("depth %d: %.1f us per batch; submit call %.1f us, wait call %.1f us" % (depth, el / 400 * 1e6, ts / 400 * 1e6, tw / max(n, 1) * 1e6), flush=True)
    bm.close()


if __name__ == "__main__":
    main()
