"""tests/golden/*.npz (scripts/make_golden.py): the oracle must still reproduce them (CPU, drift guard)
and the HIP path must reproduce them bit for bit (GPU, through the C ABI / host mirrors)."""
import ctypes as C
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def bits(a):
    return np.asarray(a, np.float32).view(np.uint32)


def load(name):
    return np.load(os.path.join(GOLD, name))


def _clauses(qrow):
    return [(int(c[0]), int(c[1]), int(c[2]), float(c[3])) for c in qrow if c[0] >= 0]


# ---------------------------------------------------------------- oracle drift guards (CPU)
def test_oracle_reproduces_similarity_golden(orc):
    g = load("similarity.npz")
    for name, order in (("wave64", orc.ORDER_WAVE64), ("serial_fma", orc.ORDER_SERIAL_FMA)):
        for sname, sim in (("dot", 0), ("cosine", 1)):
            got = np.array([orc.similarity(g["x"][i], g["y"][i], sim, order) for i in range(len(g["x"]))], np.float32)
            assert np.array_equal(bits(got), bits(g[f"{sname}_{name}"]))


def test_oracle_reproduces_vector_search_golden(orc):
    g = load("vector_search.npz")
    k = int(g["k"])
    graph = orc.Hnsw.deserialize_v2(g["graph"])
    seg = orc.Segment(g["vectors"], similarity=orc.SIM_COSINE, graph=graph)
    for name, fn, kw in (("bf", seg.brute_force, {}), ("hnsw_nodup", seg.hnsw_search, {"with_duplicates": False}),
                         ("hnsw_filter", seg.hnsw_search, {"filter_bits": g["filter"], "min_score": 0.05})):
        for i, q in enumerate(g["queries"]):
            v, s = fn(q, k, **kw)
            assert len(v) == g[name + "_count"][i]
            assert np.array_equal(v, g[name + "_vec"][i, : len(v)]) and np.array_equal(bits(s), bits(g[name + "_score"][i, : len(v)]))


def test_oracle_reproduces_bm25_golden(orc):
    g = load("bm25.npz")
    idx = orc.Bm25Index(g["term_offsets"], g["doc_ids"], g["tfs"], g["fieldnorm_ids"], int(g["total_num_tokens"]), g["alive"])
    for i, qrow in enumerate(g["queries"]):
        d, s, tot = idx.search(_clauses(qrow), int(g["k"]))
        assert tot == g["total"][i] and len(d) == g["count"][i]
        assert np.array_equal(d, g["docaddr"][i, : len(d)]) and np.array_equal(bits(s), bits(g["score"][i, : len(d)]))


# ---------------------------------------------------------------- HIP path vs golden (GPU)
@pytest.mark.gpu
def test_gpu_reproduces_similarity_golden():
    from nucliadb_amd import _lib

    g = load("similarity.npz")
    x, y = np.ascontiguousarray(g["x"]), np.ascontiguousarray(g["y"])
    for sname, sim in (("dot", 0), ("cosine", 1)):
        out = np.zeros(len(x), np.float32)
        _lib.check(_lib.lib().nidx_gpu_similarity(x.ctypes.data, y.ctypes.data, len(x), x.shape[1], sim, _lib.ORDER_WAVE64, out.ctypes.data))
        assert np.array_equal(bits(out), bits(g[f"{sname}_wave64"]))


@pytest.mark.gpu
def test_gpu_reproduces_vector_search_golden():
    from nucliadb_amd import _lib
    from nucliadb_amd.vector import Similarity, VectorConfig, VectorSearcher, VectorSearchRequest, VectorSegment

    g = load("vector_search.npz")
    x, q, k = g["vectors"], g["queries"], int(g["k"])
    n = len(x)
    seg = VectorSegment([f"k{i}" for i in range(n)], x, [[] for _ in range(n)], [b""] * n, graph=g["graph"].tobytes())
    s = VectorSearcher.open(VectorConfig(x.shape[1], Similarity.Cosine), [(seg, 1)])
    L = _lib.lib()
    filt = np.ascontiguousarray(g["filter"])

    def run(method, with_dup, min_score, use_filter):
        ov, osc, oc = np.zeros((len(q), k), np.uint32), np.zeros((len(q), k), np.float32), np.zeros(len(q), np.uint32)
        params = _lib.VectorSearchParamsC(k, min_score, int(with_dup), method)
        fp = (C.c_void_p * 1)(filt.ctypes.data) if use_filter else None
        qq = np.ascontiguousarray(q)
        _lib.check(L.nidx_gpu_vector_search(s._handle, qq.ctypes.data, len(q), C.byref(params), fp, None, None, ov.ctypes.data,
                                            osc.ctypes.data, oc.ctypes.data, None))
        return ov, osc, oc

    cases = (("bf", _lib.METHOD_BRUTE_FORCE, True, -1.0, False), ("bf_filter", _lib.METHOD_BRUTE_FORCE, True, 0.05, True),
             ("hnsw_dup", _lib.METHOD_HNSW, True, -1.0, False), ("hnsw_nodup", _lib.METHOD_HNSW, False, -1.0, False),
             ("hnsw_filter", _lib.METHOD_HNSW, True, 0.05, True), ("mfma", _lib.METHOD_BRUTE_FORCE_MFMA, True, -1.0, False))
    for name, method, dup, ms, use_filter in cases:
        ov, osc, oc = run(method, dup, ms, use_filter)
        assert np.array_equal(oc, g[name + "_count"]), name
        for i in range(len(q)):
            c = oc[i]
            assert np.array_equal(ov[i, :c], g[name + "_vec"][i, :c]), (name, i)
            assert np.array_equal(bits(osc[i, :c]), bits(g[name + "_score"][i, :c])), (name, i)
    s.close()


@pytest.mark.gpu
def test_gpu_reproduces_bm25_golden():
    from nucliadb_amd.bm25 import Bm25Searcher, Bm25Segment, Clause

    g = load("bm25.npz")
    seg = Bm25Segment(g["term_offsets"], g["doc_ids"], g["tfs"], g["fieldnorm_ids"], int(g["total_num_tokens"]), g["alive"])
    s = Bm25Searcher.open([seg])
    queries = [[Clause(*c) for c in _clauses(qrow)] for qrow in g["queries"]]
    d, sc, cnt, tot, _ = s.search_batch(queries, int(g["k"]))
    assert np.array_equal(cnt, g["count"]) and np.array_equal(tot, g["total"])
    for i in range(len(queries)):
        assert np.array_equal(d[i, : cnt[i]], g["docaddr"][i, : cnt[i]])
        assert np.array_equal(bits(sc[i, : cnt[i]]), bits(g["score"][i, : cnt[i]]))
    s.close()
