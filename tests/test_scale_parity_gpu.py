"""Parity at BENCHMARK scale (VERDICT r01 "weak" #1): the shapes bench.py times, not the small fixtures.

  * HNSW: 1 M x 768 cosine, the DEVICE-built graph, a batch of 1024 queries launched exactly like bench.py does
    (nidx_gpu_vector_segment_search_device, the 4-workgroup-per-CU launch shape, the 2^13-slot LDS visited table): the graph is
    serialised, the oracle walks it in ORDER_WAVE64 for 256 of the queries — ids, ranks and score BITS must be identical and no
    kernel flag may be raised (hnsw/search.rs:242-383).  Both corpora of bench.py: clustered (the reference's recall recipe) and uniform.
  * exact scan at 1 M x 768, batch 1024 (the shared-row scan route) vs orc_brute_force_search on 8 queries (segment.rs:569-623).
  * BM25: the 10 M-document T-zipf index of bench.py --workload bm25, 1024 queries x 3 Should terms, vs orc_bm25_search_daat on 64
    queries: doc ids, ranks, score bits, Count (nidx_paragraph/src/reader.rs:244-348).
"""
import ctypes as C
import os

import numpy as np
import pytest

from nucliadb_amd import _lib

pytestmark = pytest.mark.gpu

N, D, B, K = 1_000_000, 768, 1024, 10


@pytest.fixture(scope="module")
def torch_dev():
    import torch

    if not torch.cuda.is_available():
        if _lib.device_count() > 0:  # the HIP library sees a device: torch not seeing it is a failure, not a skip
            torch.cuda.init()
            pytest.fail("torch.cuda.is_available() is False although libnidx_gpu sees %d device(s)" % _lib.device_count())
        pytest.skip("needs a GPU")
    return torch.device("cuda", 0)


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("kind", ["clustered", "uniform"])
def test_hnsw_and_scan_match_oracle_at_1m_x_768(orc, torch_dev, kind):
    import torch

    import bench

    L = _lib.lib()
    x = bench.gen_corpus(kind, N, D, torch_dev, 1234567890)
    q = bench.gen_queries(kind, x, 1, B, D, torch_dev, 2)[0].contiguous()
    cfg = _lib.VectorConfigC(D, 1, 0, 0)
    cseg = _lib.VectorSegmentC(x.data_ptr(), D * 4, N, None, N, None, 0, 0, None, 0, None, None)
    h = C.c_void_p()
    _lib.check(L.nidx_gpu_vector_open(C.byref(cfg), C.byref(cseg), 1, C.byref(h)))
    try:
        xh = x.cpu().numpy()
        del x
        _lib.check(L.nidx_gpu_vector_build_hnsw(h, 0, 2))
        ov = torch.zeros((B, K), dtype=torch.int32, device=torch_dev)
        os_ = torch.zeros((B, K), dtype=torch.float32, device=torch_dev)
        oc = torch.zeros((B,), dtype=torch.int32, device=torch_dev)
        st = torch.zeros((B, 8), dtype=torch.int32, device=torch_dev)
        stream = torch.cuda.current_stream().cuda_stream
        flags = C.c_uint32(0)
        _lib.check(L.nidx_gpu_vector_device_flags(h, stream, C.byref(flags)))  # clear

        def run(method, nq=B):
            p = _lib.VectorSearchParamsC(K, -1.0, 1, method)
            _lib.check(L.nidx_gpu_vector_segment_search_device(h, 0, q.data_ptr(), nq, C.byref(p), None, ov.data_ptr(), os_.data_ptr(),
                                                               oc.data_ptr(), st.data_ptr(), stream))
            torch.cuda.synchronize()
            return ov.cpu().numpy().view(np.uint32).copy(), os_.cpu().numpy().copy(), oc.cpu().numpy().view(np.uint32).copy()

        gv, gs, gc = run(_lib.METHOD_HNSW)
        stats = st.cpu().numpy()
        assert int(np.bitwise_or.reduce(stats[:, 3])) == 0, "kernel flags raised at benchmark scale"
        _lib.check(L.nidx_gpu_vector_device_flags(h, stream, C.byref(flags)))
        assert flags.value == 0
        graph, edges = bench.serialize_graph(L, h)
        qh = q.cpu().numpy()
        threads = min(32, os.cpu_count() or 1)
        oseg = orc.Segment(xh, similarity=orc.SIM_COSINE, order=orc.ORDER_WAVE64, graph=orc.Hnsw.deserialize_v2(graph, edges))
        nq = 256
        wv, ws, wc, wst = oseg.hnsw_search_batch(qh[:nq], K, threads=threads, want_stats=True)
        for i in range(nq):
            assert wc[i] == gc[i], (i, wc[i], gc[i])
            assert np.array_equal(wv[i, : wc[i]], gv[i, : wc[i]]), (i, wv[i], gv[i])
            assert np.array_equal(_bits(ws[i, : wc[i]]), _bits(gs[i, : wc[i]])), i
        # the kernel's own traffic counters are the oracle's (what the roofline's algorithmic bytes are made of)
        assert np.array_equal(stats[:nq, 0].astype(np.uint64), wst[:, 0]), "distance evaluations differ from the oracle's"
        assert np.array_equal(stats[:nq, 1].astype(np.uint64), wst[:, 1]), "expansions differ from the oracle's"
        # the complete entry point (launch + one D2H + flag check) returns the same block
        words = B * K * 2 + B + 1
        d_block = torch.zeros((words,), dtype=torch.int32, device=torch_dev)
        h_block = torch.zeros((words,), dtype=torch.int32).pin_memory()
        retried = C.c_uint32(7)
        p = _lib.VectorSearchParamsC(K, -1.0, 1, _lib.METHOD_HNSW)
        _lib.check(L.nidx_gpu_vector_segment_search_device_exact(h, 0, q.data_ptr(), B, C.byref(p), None, d_block.data_ptr(), h_block.data_ptr(),
                                                                 stream, C.byref(retried)))
        hb = h_block.numpy().view(np.uint32)
        assert retried.value == 0 and hb[-1] == 0
        assert np.array_equal(hb[: B * K].reshape(B, K), gv) and np.array_equal(hb[B * K: 2 * B * K].reshape(B, K), _bits(gs))
        assert np.array_equal(hb[2 * B * K: 2 * B * K + B], gc)
        # exact scan, full batch (the shared-row route), checked on 8 queries
        ev, es, ec = run(_lib.METHOD_BRUTE_FORCE)
        oseg.graph = None
        bv, bs, bc = oseg.brute_force_batch(qh[:8], K, threads=8)
        for i in range(8):
            assert bc[i] == ec[i]
            assert np.array_equal(bv[i, : bc[i]], ev[i, : bc[i]]), (i, bv[i], ev[i])
            assert np.array_equal(_bits(bs[i, : bc[i]]), _bits(es[i, : bc[i]])), i
        if kind == "clustered":
            rec = np.mean([len(set(gv[i, :K].tolist()) & set(ev[i, :K].tolist())) / K for i in range(B)])
            assert rec >= 0.9, rec
    finally:
        L.nidx_gpu_vector_close(h)


def test_bm25_matches_oracle_on_the_10m_doc_zipf_index(orc, torch_dev, monkeypatch):
    import bench
    from nucliadb_amd.bm25 import Bm25Searcher, Bm25Segment

    L = _lib.lib()
    n_docs, vocab, k = 10_000_000, 1_000_000, 20
    term_offsets, doc_ids, tfs, fieldnorm_ids, total_tokens = bench.zipf_corpus_on_device(L, torch_dev, n_docs, vocab, 0)
    searcher = Bm25Searcher.open([Bm25Segment(term_offsets, doc_ids, tfs, fieldnorm_ids, total_tokens)])
    try:
        rng = np.random.default_rng(2)
        terms = rng.integers(99, 100_000, (B, 3))
        cl = (_lib.Bm25ClauseC * (3 * B))()
        for i in range(B):
            for j in range(3):
                cl[3 * i + j].term, cl[3 * i + j].occur, cl[3 * i + j].mode, cl[3 * i + j].boost = int(terms[i, j]), 0, 0, 1.0
        offsets = (np.arange(B + 1, dtype=np.uint64) * 3).copy()
        docaddr, score = np.zeros((B, k), np.uint64), np.zeros((B, k), np.float32)
        count, total, post = np.zeros(B, np.uint32), np.zeros(B, np.uint64), np.zeros(B, np.uint64)
        _lib.check(L.nidx_gpu_bm25_search(searcher._handle, cl, offsets.ctypes.data, B, k, None, docaddr.ctypes.data, score.ctypes.data,
                                          count.ctypes.data, total.ctypes.data, post.ctypes.data))
        # the work list's throughput shape (what a batch gets when other tickets are out: the fewest slices the bitmaps' collision
        # estimate allows, csrc/bm25_index.cpp crowded_shape) must give the same bits
        monkeypatch.setenv("NIDX_GPU_BM25_CROWDED", "1")
        d2, s2 = np.zeros((B, k), np.uint64), np.zeros((B, k), np.float32)
        c2, t2, p2 = np.zeros(B, np.uint32), np.zeros(B, np.uint64), np.zeros(B, np.uint64)
        _lib.check(L.nidx_gpu_bm25_search(searcher._handle, cl, offsets.ctypes.data, B, k, None, d2.ctypes.data, s2.ctypes.data, c2.ctypes.data, t2.ctypes.data,
                                          p2.ctypes.data))
        monkeypatch.delenv("NIDX_GPU_BM25_CROWDED")
        assert np.array_equal(c2, count) and np.array_equal(t2, total) and np.array_equal(p2, post)
        for i in range(B):
            assert np.array_equal(d2[i, : count[i]], docaddr[i, : count[i]]) and np.array_equal(_bits(s2[i, : count[i]]), _bits(score[i, : count[i]])), i
        oidx = orc.Bm25Index(term_offsets, doc_ids, tfs, fieldnorm_ids, total_tokens)
        nq = 64
        queries = [[(int(t), 0, 0, 1.0) for t in terms[i]] for i in range(nq)]
        od, os_, oc, ot = orc.bm25_search_daat_batch(oidx, queries, k, threads=min(32, os.cpu_count() or 1))
        for i in range(nq):
            assert oc[i] == count[i] and ot[i] == total[i], (i, oc[i], count[i], ot[i], total[i])
            assert np.array_equal(od[i, : oc[i]], docaddr[i, : oc[i]]), i
            assert np.array_equal(_bits(os_[i, : oc[i]]), _bits(score[i, : oc[i]])), i
    finally:
        searcher.close()
