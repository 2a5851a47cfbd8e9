"""GPU parity tests of the vector path: HIP kernels (through the C ABI) vs the CPU oracle on the same
seeded inputs.  Bar: bit-exact vector addresses, ranks AND score bit patterns (both sides sum in the
WAVE64 order); SURVEY §8c asks for cosine within 1e-5, which bit-exactness implies."""
import ctypes as C

import numpy as np
import pytest

from nucliadb_amd import _lib
from nucliadb_amd.vector import (Similarity, VectorConfig, VectorSearcher, VectorSearchRequest, VectorSegment)

pytestmark = pytest.mark.gpu


def unit_rows(rng, n, d):
    # the reference's own generator (segment.rs:682-695): uniform(-1,1) then L2-normalised
    x = rng.uniform(-1, 1, (n, d)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True).astype(np.float32)
    return x


def bits(a):
    return np.asarray(a, dtype=np.float32).view(np.uint32)


def make_segment(x, graph=None):
    n = x.shape[0]
    return VectorSegment([f"k{i}" for i in range(n)], x, [[] for _ in range(n)], [b""] * n, graph=graph)


def gpu_search(x, sim, queries, k, *, method, graph=None, min_score=-1.0, with_duplicates=True, filter_bits=None, alive=None, info=None, tunables=None):
    """Single segment straight through the C ABI (no Python filter logic in between)."""
    L = _lib.lib()
    n, d = x.shape
    cfg = _lib.VectorConfigC(d, sim, 0, 0)
    g = np.frombuffer(graph, np.uint8) if graph is not None else None
    seg = _lib.VectorSegmentC(x.ctypes.data, d * 4, n, None, n, g.ctypes.data if g is not None else None,
                              len(graph) if graph is not None else 0, 0, None, 0, alive.ctypes.data if alive is not None else None, None)
    h = C.c_void_p()
    _lib.check(L.nidx_gpu_vector_open(C.byref(cfg), C.byref(seg), 1, C.byref(h)))
    try:
        for name, value in (tunables or {}).items():
            _lib.check(L.nidx_gpu_vector_set_tunable(h, name.encode(), value))
        q = np.ascontiguousarray(queries, np.float32)
        B = q.shape[0]
        ov, osc, oc = np.zeros((B, k), np.uint32), np.zeros((B, k), np.float32), np.zeros(B, np.uint32)
        params = _lib.VectorSearchParamsC(k, min_score, int(with_duplicates), method)
        fp = None
        if filter_bits is not None:
            fp = (C.c_void_p * 1)(filter_bits.ctypes.data)
        _lib.check(L.nidx_gpu_vector_search(h, q.ctypes.data, B, C.byref(params), fp, None, None, ov.ctypes.data,
                                            osc.ctypes.data, oc.ctypes.data, None))
        if info is not None:
            n_spill = C.c_uint64(0)
            _lib.check(L.nidx_gpu_vector_spill_stats(h, C.byref(n_spill)))
            info["spill_queries"] = n_spill.value
        return ov, osc, oc
    finally:
        L.nidx_gpu_vector_close(h)


# ---- a1: dense_f32::{dot,cosine}_similarity -----------------------------------------------------------
@pytest.mark.parametrize("d", [3, 10, 64, 256, 758, 768, 1024, 1536])
def test_similarity_bits_match_oracle(orc, d):
    rng = np.random.default_rng(d)
    n = 257
    x = rng.normal(size=(n, d)).astype(np.float32)
    y = rng.normal(size=(n, d)).astype(np.float32)
    x[0] = 0  # zero-norm edge cases of SimSIMD's cosine
    y[1] = 0
    x[2] = 0
    y[2] = 0
    y[3] = x[3]
    L = _lib.lib()
    for sim in (0, 1):
        out = np.zeros(n, np.float32)
        _lib.check(L.nidx_gpu_similarity(x.ctypes.data, y.ctypes.data, n, d, sim, _lib.ORDER_WAVE64, out.ctypes.data))
        want = np.array([orc.similarity(x[i], y[i], sim) for i in range(n)], np.float32)
        assert np.array_equal(bits(out), bits(want)), np.nonzero(bits(out) != bits(want))
    # and the reference's own accuracy bar (dense_f32.rs:66-84): |ours - naive| < 0.01
    naive = np.einsum("ij,ij->i", x.astype(np.float64), y.astype(np.float64))
    out = np.zeros(n, np.float32)
    _lib.check(L.nidx_gpu_similarity(x.ctypes.data, y.ctypes.data, n, d, 0, _lib.ORDER_WAVE64, out.ctypes.data))
    assert np.max(np.abs(out - naive)) < 0.01


# ---- a7: brute_force_search ----------------------------------------------------------------------------
@pytest.mark.parametrize("sim", [0, 1])
@pytest.mark.parametrize("n,d,nq,k", [(1, 8, 1, 3), (5, 3, 2, 10), (4000, 64, 9, 10), (20000, 768, 17, 10), (3000, 758, 5, 7), (6000, 128, 5, 500), (2500, 768, 3, 301),
                                      (2500, 1024, 8, 64), (1200, 1536, 3, 5), (4000, 64, 9, 200), (300, 32, 1, 256),
                                      (9000, 768, 6, 100)])
def test_brute_force_matches_oracle(orc, sim, n, d, nq, k):
    rng = np.random.default_rng(n * 31 + d)
    x = unit_rows(rng, n, d) if d > 3 else rng.normal(size=(n, d)).astype(np.float32)
    q = rng.normal(size=(nq, d)).astype(np.float32)
    ov, osc, oc = gpu_search(x, sim, q, k, method=_lib.METHOD_BRUTE_FORCE)
    oseg = orc.Segment(x, similarity=sim)
    for i in range(nq):
        wv, ws = oseg.brute_force(q[i], k)
        assert oc[i] == len(wv)
        assert np.array_equal(ov[i, : oc[i]], wv), (i, ov[i], wv)
        assert np.array_equal(bits(osc[i, : oc[i]]), bits(ws))


@pytest.mark.parametrize("sim", [0, 1])
@pytest.mark.parametrize("n,d,nq,k", [(20003, 768, 200, 10), (5001, 256, 64, 64), (9000, 1024, 130, 7), (4100, 512, 65, 33)])
def test_shared_row_scan_matches_oracle_and_register_tile_scan(orc, monkeypatch, sim, n, d, nq, k):
    """Large batches take the shared-row form of the exact scan (vector_scan_shared.hip: rows staged once per 64 queries through
    LDS): same arithmetic, so ids, ranks and score bits equal the oracle's and the register-tile scan's — with dead rows, a
    filter, a min_score cut, identical rows (address-ordered ties) and n not a multiple of the 8-row tile."""
    rng = np.random.default_rng(n + d + sim)
    x = unit_rows(rng, n, d)
    x[200:230] = x[11]                     # exact ties
    q = rng.normal(size=(nq, d)).astype(np.float32)
    q[0] = x[11]
    alive = orc.bitset(n, fill=True)
    for dead in (11, 200, n - 1, 4097):
        alive[dead >> 6] &= ~np.uint64(1 << (dead & 63))
    filt = orc.bitset(n, ones=np.nonzero(rng.random(n) < 0.6)[0].tolist() + list(range(200, 230)))
    oseg = orc.Segment(x, similarity=sim, alive=alive)
    for kwargs, obits, ms in (({}, None, -1.0), ({"alive": alive, "filter_bits": filt}, alive & filt, 0.02)):
        monkeypatch.setenv("NIDX_GPU_SCAN_SHARED", "1")
        ov, osc, oc = gpu_search(x, sim, q, k, method=_lib.METHOD_BRUTE_FORCE, min_score=ms, **kwargs)
        monkeypatch.setenv("NIDX_GPU_SCAN_SHARED", "0")
        rv, rsc, rc = gpu_search(x, sim, q, k, method=_lib.METHOD_BRUTE_FORCE, min_score=ms, **kwargs)
        assert np.array_equal(oc, rc) and np.array_equal(ov, rv) and np.array_equal(bits(osc), bits(rsc))
        oref = oseg if obits is not None else orc.Segment(x, similarity=sim)
        for i in sorted({0, 1, 2, 3, 4, 5, 63, min(64, nq - 1), nq - 1}):
            wv, ws = oref.brute_force(q[i], k, min_score=ms, filter_bits=obits)
            assert oc[i] == len(wv), (i, oc[i], len(wv))
            assert np.array_equal(ov[i, : oc[i]], wv), (i, ov[i, : oc[i]], wv)
            assert np.array_equal(bits(osc[i, : oc[i]]), bits(ws))


def test_brute_force_filters_min_score_and_ties(orc):
    rng = np.random.default_rng(99)
    n, d, k = 6000, 128, 10
    x = unit_rows(rng, n, d)
    x[100:140] = x[7]          # 41 identical rows -> 41-way exact score ties: order must be address asc
    q = np.vstack([x[7][None, :], rng.normal(size=(4, d)).astype(np.float32)])
    alive = orc.bitset(n, fill=True)
    for dead in (7, 100, 101, 5999):
        alive[dead >> 6] &= ~np.uint64(1 << (dead & 63))
    filt = orc.bitset(n, ones=np.nonzero(rng.random(n) < 0.3)[0].tolist() + list(range(100, 140)))
    both = alive & filt
    for sim in (0, 1):
        oseg = orc.Segment(x, similarity=sim, alive=alive)
        for kwargs, obits in (({"alive": alive}, alive), ({"alive": alive, "filter_bits": filt}, both)):
            for ms in (-1.0, 0.1, 0.99):
                ov, osc, oc = gpu_search(x, sim, q, k, method=_lib.METHOD_BRUTE_FORCE, min_score=ms, **kwargs)
                for i in range(q.shape[0]):
                    wv, ws = oseg.brute_force(q[i], k, min_score=ms, filter_bits=obits)
                    assert oc[i] == len(wv), (sim, ms, i)
                    assert np.array_equal(ov[i, : oc[i]], wv)
                    assert np.array_equal(bits(osc[i, : oc[i]]), bits(ws))


# ---- a3/a4/a5: HNSW search over an oracle-built graph (DiskHnswV2 image) --------------------------------
@pytest.fixture(scope="module")
def hnsw_case(orc):
    rng = np.random.default_rng(1234567890)
    n, d = 3000, 96
    x = unit_rows(rng, n, d)
    x[50:58] = x[49]  # duplicate vectors: exercises RepCounter (with_duplicates=false)
    oseg = orc.Segment(x, similarity=orc.SIM_COSINE)
    g = oseg.build_graph(seed=2)
    gbytes, _ = g.serialize_v2(n)
    return x, oseg, bytes(gbytes)


@pytest.mark.parametrize("k", [1, 10, 30, 50, 64, 100, 200, 300, 500])   # 500: MAX_RANK_FUSION_WINDOW (result_per_page = max(top_k, windows))
@pytest.mark.parametrize("with_dup", [True, False])
def test_hnsw_search_matches_oracle(orc, hnsw_case, k, with_dup):
    x, oseg, gbytes = hnsw_case
    rng = np.random.default_rng(k)
    q = np.vstack([x[49][None, :], x[1234][None, :], unit_rows(rng, 30, x.shape[1])])
    ov, osc, oc = gpu_search(x, 1, q, k, method=_lib.METHOD_HNSW, graph=gbytes, with_duplicates=with_dup)
    for i in range(q.shape[0]):
        wv, ws = oseg.hnsw_search(q[i], k, with_duplicates=with_dup)
        assert oc[i] == len(wv), (i, oc[i], len(wv))
        assert np.array_equal(ov[i, : oc[i]], wv), (i, ov[i, : oc[i]], wv)
        assert np.array_equal(bits(osc[i, : oc[i]]), bits(ws))


@pytest.mark.parametrize("ef_upper,ef_search", [(4, 0), (1, 48), (8, 100), (64, 30), (16, 300)])
def test_hnsw_search_knobs_match_the_oracles(orc, hnsw_case, ef_upper, ef_search):
    """The two knobs that change results — "ef_search" (layer-0 width, reference constant 30) and "ef_upper" (results kept per
    upper layer of the descent, reference: 1, hnsw/search.rs:318-345) — are restated in the oracle: ids, ranks and score bits."""
    x, oseg, gbytes = hnsw_case
    rng = np.random.default_rng(ef_upper * 1000 + ef_search)
    q = np.vstack([x[49][None, :], unit_rows(rng, 24, x.shape[1])])
    oseg.ef_upper, oseg.ef_search = ef_upper, ef_search
    try:
        for k, with_dup in ((10, True), (10, False), (70, True)):
            ov, osc, oc = gpu_search(x, 1, q, k, method=_lib.METHOD_HNSW, graph=gbytes, with_duplicates=with_dup,
                                     tunables={"ef_upper": ef_upper, "ef_search": ef_search})
            for i in range(q.shape[0]):
                wv, ws = oseg.hnsw_search(q[i], k, with_duplicates=with_dup)
                assert oc[i] == len(wv), (i, oc[i], len(wv))
                assert np.array_equal(ov[i, : oc[i]], wv), (i, ov[i, : oc[i]], wv)
                assert np.array_equal(bits(osc[i, : oc[i]]), bits(ws))
    finally:
        oseg.ef_upper, oseg.ef_search = 0, 0


def test_hnsw_search_filter_and_min_score(orc, hnsw_case):
    x, oseg, gbytes = hnsw_case
    n = x.shape[0]
    rng = np.random.default_rng(3)
    q = unit_rows(rng, 16, x.shape[1])
    for sel in (0.5, 0.05):
        filt = orc.bitset(n, ones=np.nonzero(rng.random(n) < sel)[0].tolist())
        for ms in (-1.0, 0.2):
            ov, osc, oc = gpu_search(x, 1, q, 10, method=_lib.METHOD_HNSW, graph=gbytes, filter_bits=filt, min_score=ms)
            for i in range(q.shape[0]):
                wv, ws = oseg.hnsw_search(q[i], 10, min_score=ms, filter_bits=filt)
                assert oc[i] == len(wv), (sel, ms, i, oc[i], len(wv))
                assert np.array_equal(ov[i, : oc[i]], wv)
                assert np.array_equal(bits(osc[i, : oc[i]]), bits(ws))


def test_hnsw_walk_larger_than_the_on_chip_pool_matches_oracle(orc, hnsw_case):
    """closest_up_nodes (search.rs:188-240) keeps an unbounded heap and visited set: under a filter that lets one row in
    a hundred (or a thousand) through, the walk pops far more nodes than the kernel's LDS pool holds.  Those queries are
    re-run by the HBM-resident fallback (hnsw_spill.hip) and must still equal the oracle bit for bit."""
    x, oseg, gbytes = hnsw_case
    n = x.shape[0]
    rng = np.random.default_rng(31)
    q = np.vstack([x[49][None, :], unit_rows(rng, 23, x.shape[1])])
    spilled = 0
    for sel, k, with_dup, ms in ((0.01, 10, True, -1.0), (0.003, 10, False, -1.0), (0.02, 40, True, -1.0), (0.01, 10, True, 0.05), (0.0, 5, True, -1.0)):
        ones = np.nonzero(rng.random(n) < sel)[0].tolist() if sel else [7]      # sel 0: a single admissible row
        if sel == 0.003:
            ones = sorted(set(ones) | set(range(49, 58)))                        # the duplicated rows: RepCounter in the fallback
        filt = orc.bitset(n, ones=ones)
        info = {}
        ov, osc, oc = gpu_search(x, 1, q, k, method=_lib.METHOD_HNSW, graph=gbytes, filter_bits=filt, min_score=ms, with_duplicates=with_dup,
                                 info=info)
        spilled += info["spill_queries"]
        for i in range(q.shape[0]):
            wv, ws = oseg.hnsw_search(q[i], k, min_score=ms, filter_bits=filt, with_duplicates=with_dup)
            assert oc[i] == len(wv), (sel, k, i, oc[i], len(wv))
            assert np.array_equal(ov[i, : oc[i]], wv), (sel, k, i)
            assert np.array_equal(bits(osc[i, : oc[i]]), bits(ws))
    assert spilled > 0  # the fallback really ran


def test_auto_routing_follows_use_hnsw(orc, hnsw_case):
    x, oseg, gbytes = hnsw_case
    n = x.shape[0]
    rng = np.random.default_rng(4)
    q = unit_rows(rng, 4, x.shape[1])
    # unfiltered: HNSW; 20 matching rows: brute force (segment.rs:626-660)
    few = orc.bitset(n, ones=list(range(0, n, n // 20))[:20])
    for filt in (None, few):
        ov, osc, oc = gpu_search(x, 1, q, 10, method=_lib.METHOD_AUTO, graph=gbytes, filter_bits=filt)
        for i in range(q.shape[0]):
            wv, ws, method = oseg.search(q[i], 10, filter_bits=filt)
            assert method == ("hnsw" if filt is None else "brute force")
            assert np.array_equal(ov[i, : oc[i]], wv) and np.array_equal(bits(osc[i, : oc[i]]), bits(ws))


def test_hnsw_dot_d768(orc):
    rng = np.random.default_rng(77)
    n, d = 1500, 768
    x = unit_rows(rng, n, d)
    oseg = orc.Segment(x, similarity=orc.SIM_DOT)
    g = oseg.build_graph(seed=2)
    gbytes, _ = g.serialize_v2(n)
    q = unit_rows(rng, 12, d)
    ov, osc, oc = gpu_search(x, 0, q, 10, method=_lib.METHOD_HNSW, graph=bytes(gbytes))
    for i in range(q.shape[0]):
        wv, ws = oseg.hnsw_search(q[i], 10)
        assert np.array_equal(ov[i, : oc[i]], wv) and np.array_equal(bits(osc[i, : oc[i]]), bits(ws))


def test_graph_image_round_trip(orc, hnsw_case):
    """hnsw.graph in -> HBM layout -> hnsw.graph out must be byte identical (disk/v2.rs:339-473)."""
    x, oseg, gbytes = hnsw_case
    s = VectorSearcher.open(VectorConfig(x.shape[1], Similarity.Cosine), [(make_segment(x, gbytes), 1)])
    out, _edges = s.serialize_hnsw(0)
    s.close()
    assert out == gbytes


# ---- a9: multi-segment Fssc merge ----------------------------------------------------------------------------
def test_multi_segment_fssc_matches_oracle(orc):
    rng = np.random.default_rng(11)
    d, k = 64, 10
    xs = [unit_rows(rng, n, d) for n in (700, 300, 1200)]
    xs[1][5] = xs[0][17]      # same vector bytes in two segments: dropped unless with_duplicates
    xs[2][9] = xs[0][17]
    q = np.vstack([xs[0][17][None, :], unit_rows(rng, 6, d)])
    osegs = [orc.Segment(x, similarity=orc.SIM_DOT) for x in xs]
    keys, base = [], 0
    for x in xs:
        keys.append(np.arange(base, base + x.shape[0], dtype=np.uint64))
        base += x.shape[0]
    segs = [VectorSegment([f"s{si}-{i}" for i in range(x.shape[0])], x, [[] for _ in range(x.shape[0])], [b""] * x.shape[0])
            for si, x in enumerate(xs)]
    # open() searches newest first: give the oracle the same order
    searcher = VectorSearcher.open(VectorConfig(d, Similarity.Dot), [(segs[0], 3), (segs[1], 2), (segs[2], 1)])
    for with_dup in (False, True):
        req = VectorSearchRequest(result_per_page=k, min_score=-1.0, with_duplicates=with_dup)
        seg, par, vec, score, count = searcher.search_batch(req, q)
        for i in range(q.shape[0]):
            want = orc.searcher_search(osegs, keys, q[i], k, with_duplicates=with_dup)
            got = [(int(seg[i, j]), int(vec[i, j])) for j in range(count[i])]
            assert got == [(w[2], w[3]) for w in want], (with_dup, i)
            assert np.array_equal(bits(score[i, : count[i]]), bits([w[1] for w in want]))
    searcher.close()


# ---- error behaviour (searcher.rs:255-262) -----------------------------------------------------------------
def test_inconsistent_dimensions():
    x = np.eye(8, dtype=np.float32)
    s = VectorSearcher.open(VectorConfig(8), [(make_segment(x), 1)])
    with pytest.raises(_lib.NidxGpuError) as e:
        s.search(VectorSearchRequest(vector=[0.0] * 7, result_per_page=3, min_score=-1.0))
    assert e.value.code == _lib.NIDX_ERR_INCONSISTENT_DIMENSIONS
    assert "Inconsistent dimensions. Index=8 Vector=7" in e.value.message
    s.close()


# ---- a7 batched: the f32-MFMA GEMM scan (SERIAL_FMA order) ---------------------------------------------------
@pytest.mark.parametrize("sim", [0, 1])
@pytest.mark.parametrize("n,d,nq,k", [(1, 8, 1, 3), (130, 20, 3, 10), (4000, 64, 9, 10), (20000, 768, 200, 10), (3000, 758, 5, 7),
                                      (2500, 1024, 130, 16), (1200, 1536, 3, 5), (5000, 96, 140, 17), (9000, 768, 33, 64), (40, 32, 3, 64)])
def test_mfma_scan_matches_oracle_serial_fma(orc, sim, n, d, nq, k):
    rng = np.random.default_rng(n * 17 + d)
    x = unit_rows(rng, n, d) if d > 8 else rng.normal(size=(n, d)).astype(np.float32)
    q = rng.normal(size=(nq, d)).astype(np.float32)
    ov, osc, oc = gpu_search(x, sim, q, k, method=_lib.METHOD_BRUTE_FORCE_MFMA)
    oseg = orc.Segment(x, similarity=sim, order=orc.ORDER_SERIAL_FMA)
    for i in list(range(min(nq, 12))) + [nq - 1]:
        wv, ws = oseg.brute_force(q[i], k)
        assert oc[i] == len(wv), (i, oc[i], len(wv))
        assert np.array_equal(ov[i, : oc[i]], wv), (i, ov[i], wv)
        assert np.array_equal(bits(osc[i, : oc[i]]), bits(ws))


def test_mfma_scan_filters_min_score_and_ties(orc):
    rng = np.random.default_rng(98)
    n, d, k = 6000, 128, 10
    x = unit_rows(rng, n, d)
    x[100:140] = x[7]
    q = np.vstack([x[7][None, :], rng.normal(size=(4, d)).astype(np.float32)])
    alive = orc.bitset(n, fill=True)
    for dead in (7, 100, 101, 5999):
        alive[dead >> 6] &= ~np.uint64(1 << (dead & 63))
    filt = orc.bitset(n, ones=np.nonzero(rng.random(n) < 0.3)[0].tolist() + list(range(100, 140)))
    both = alive & filt
    for sim in (0, 1):
        oseg = orc.Segment(x, similarity=sim, alive=alive, order=orc.ORDER_SERIAL_FMA)
        for kwargs, obits in (({"alive": alive}, alive), ({"alive": alive, "filter_bits": filt}, both)):
            for ms in (-1.0, 0.1, 0.99):
                ov, osc, oc = gpu_search(x, sim, q, k, method=_lib.METHOD_BRUTE_FORCE_MFMA, min_score=ms, **kwargs)
                for i in range(q.shape[0]):
                    wv, ws = oseg.brute_force(q[i], k, min_score=ms, filter_bits=obits)
                    assert oc[i] == len(wv), (sim, ms, i)
                    assert np.array_equal(ov[i, : oc[i]], wv)
                    assert np.array_equal(bits(osc[i, : oc[i]]), bits(ws))


def test_mfma_scan_page_bound():
    """Pages above a full wave of ranks are refused, not truncated."""
    rng = np.random.default_rng(3)
    x = unit_rows(rng, 500, 32)
    with pytest.raises(_lib.NidxGpuError) as e:
        gpu_search(x, 0, x[:2], 65, method=_lib.METHOD_BRUTE_FORCE_MFMA)
    assert e.value.code == _lib.NIDX_ERR_UNSUPPORTED


def test_mfma_and_wave64_scans_agree_within_1e5():
    """Two summation orders, one answer up to f32 round-off: same ids wherever the score gap exceeds
    the round-off, scores within the 1e-5 the north star allows for cosine."""
    rng = np.random.default_rng(5)
    x = unit_rows(rng, 30000, 768)
    q = unit_rows(rng, 64, 768)
    v1, s1, c1 = gpu_search(x, 1, q, 10, method=_lib.METHOD_BRUTE_FORCE)
    v2, s2, c2 = gpu_search(x, 1, q, 10, method=_lib.METHOD_BRUTE_FORCE_MFMA)
    assert np.array_equal(c1, c2)
    assert np.max(np.abs(s1 - s2)) < 1e-5
    assert np.mean(v1 == v2) > 0.99


# ---- batched fallback on the bf16 matrix cores with exact re-scoring (BASELINE configs[4]) -------------------
@pytest.mark.parametrize("sim", [0, 1])
@pytest.mark.parametrize("n,d,nq,k", [(500, 64, 7, 10), (30000, 768, 130, 10), (9000, 1024, 40, 16), (2000, 100, 5, 32),
                                      (150000, 64, 140, 10), (140000, 200, 33, 32)])  # the last two take the two-pass (sample bound + append) route
def test_bf16_fallback_scores_exact_and_recall(orc, sim, n, d, nq, k):
    """Returned scores must be bit-identical to the exact (WAVE64) similarity of the returned ids and
    sorted by (score desc, address asc); the id set may differ from the exact top-k only through bf16
    ranking error, which on these data must stay below 1 %."""
    rng = np.random.default_rng(n + d)
    x = unit_rows(rng, n, d)
    q = unit_rows(rng, nq, d)
    v_ex, s_ex, c_ex = gpu_search(x, sim, q, k, method=_lib.METHOD_BRUTE_FORCE)
    v_bf, s_bf, c_bf = gpu_search(x, sim, q, k, method=_lib.METHOD_BRUTE_FORCE_BF16)
    assert np.array_equal(c_ex, c_bf)
    hits = 0
    for i in range(nq):
        hits += len(set(v_ex[i, : c_ex[i]].tolist()) & set(v_bf[i, : c_bf[i]].tolist()))
        for j in range(min(c_bf[i], 4)):
            want = orc.similarity(x[v_bf[i, j]], q[i], sim)
            assert bits([s_bf[i, j]])[0] == bits([want])[0]
        keys = [(-float(s_bf[i, j]), int(v_bf[i, j])) for j in range(c_bf[i])]
        assert keys == sorted(keys)
    assert hits / float(c_ex.sum()) >= 0.99


@pytest.mark.parametrize("case", ["one_pass", "sampled_floor", "sampled_then_crowded", "crowded_sample", "crowded_stripe", "filter_empties_the_floor"])
def test_bf16_append_scan_equals_the_list_scan(orc, case):
    """bf16_append_kernel (candidates appended under a per-query floor, no lists in LDS) against bf16_scan_kernel alone
    (NIDX_GPU_BF16_APPEND=0): the same candidates, hence the same ids, counts and score bits — with one append pass, with the
    floor tightened on a strided sample first, when a stripe holds more than 32 rows above the floor (the block falls back to the
    list kernel) and when a filter leaves the sampled prefix without 32 rows (no floor: the block is left to the list kernel)."""
    import os
    rng = np.random.default_rng(77)
    kwargs = {}
    if case == "one_pass":
        n, d, nq, k = 150000, 64, 140, 10
    elif case in ("sampled_floor", "sampled_then_crowded", "crowded_sample"):
        # 4 query blocks -> 64 stripes: the 64 k-row prefix floor carries 1.3 M rows, so every 12th round is sampled first and the full
        # pass goes on from that sample's slots (and skips its rounds)
        n, d, nq, k = 1500000, 64, 800, 10
    elif case == "crowded_stripe":
        n, d, nq, k = 120000, 64, 300, 32
    else:
        n, d, nq, k = 120000, 64, 70, 10
    x = unit_rows(rng, n, d)
    q = unit_rows(rng, nq, d)
    crowd = {"crowded_stripe": 60000, "sampled_then_crowded": 700000, "crowded_sample": 197000}.get(case)
    if crowd is not None:
        # 3 000 near-copies of query 0 in consecutive rows behind the sampled prefix: in rounds the full pass scans (its slots fill up:
        # the list kernel takes the block), or — rows 197 k.. = round 12 — in a sampled round (the sample's slots fill up: the full
        # pass drops them and scans every round)
        x[crowd:crowd + 3000] = q[0] + rng.normal(size=(3000, d)).astype(np.float32) * np.float32(0.01)
        x[crowd:crowd + 3000] /= np.linalg.norm(x[crowd:crowd + 3000], axis=1, keepdims=True)
    if case == "filter_empties_the_floor":
        ones = np.nonzero(rng.random(n) < 0.3)[0]
        ones = ones[ones >= 20000]                  # nothing of the sampled prefix passes
        kwargs["filter_bits"] = orc.bitset(n, ones=ones.tolist())
    old = os.environ.get("NIDX_GPU_BF16_APPEND")
    try:
        os.environ["NIDX_GPU_BF16_APPEND"] = "0"
        v0, s0, c0 = gpu_search(x, 1, q, k, method=_lib.METHOD_BRUTE_FORCE_BF16, **kwargs)
        os.environ["NIDX_GPU_BF16_APPEND"] = "1"
        v1, s1, c1 = gpu_search(x, 1, q, k, method=_lib.METHOD_BRUTE_FORCE_BF16, **kwargs)
        # the other shape of the append kernel's operand ring (nine chunks, a barrier per chunk): the same candidates again
        os.environ["NIDX_GPU_BF16_MAINLOOP"] = "1"
        v2, s2, c2 = gpu_search(x, 1, q, k, method=_lib.METHOD_BRUTE_FORCE_BF16, **kwargs)
    finally:
        os.environ.pop("NIDX_GPU_BF16_MAINLOOP", None)
        if old is None:
            os.environ.pop("NIDX_GPU_BF16_APPEND", None)
        else:
            os.environ["NIDX_GPU_BF16_APPEND"] = old
    assert np.array_equal(c0, c1) and np.array_equal(c0, c2)
    for i in range(nq):
        assert np.array_equal(v0[i, : c0[i]], v1[i, : c1[i]]), (case, i)
        assert np.array_equal(bits(s0[i, : c0[i]]), bits(s1[i, : c1[i]]))
        assert np.array_equal(v0[i, : c0[i]], v2[i, : c2[i]]), (case, i)
        assert np.array_equal(bits(s0[i, : c0[i]]), bits(s2[i, : c2[i]]))
    if crowd is not None:
        assert set(v1[0, : c1[0]].tolist()) <= set(range(crowd, crowd + 3000))


def test_bf16_fallback_filter_and_min_score(orc):
    rng = np.random.default_rng(21)
    n, d, k = 8000, 128, 10
    x = unit_rows(rng, n, d)
    q = unit_rows(rng, 9, d)
    filt = orc.bitset(n, ones=np.nonzero(rng.random(n) < 0.2)[0].tolist())
    v_ex, s_ex, c_ex = gpu_search(x, 1, q, k, method=_lib.METHOD_BRUTE_FORCE, filter_bits=filt, min_score=0.15)
    v_bf, s_bf, c_bf = gpu_search(x, 1, q, k, method=_lib.METHOD_BRUTE_FORCE_BF16, filter_bits=filt, min_score=0.15)
    assert np.array_equal(c_ex, c_bf)
    for i in range(len(q)):
        assert np.array_equal(v_ex[i, : c_ex[i]], v_bf[i, : c_bf[i]])
        assert np.array_equal(bits(s_ex[i, : c_ex[i]]), bits(s_bf[i, : c_bf[i]]))


@pytest.mark.parametrize("d", [3500, 4096])
def test_wide_dimensions_match_oracle(orc, d):
    """Dimensions above 3 072 (round 5: up to 4 096 — sixteen 16-byte pieces of a row per lane): the exact scan, the HNSW search over
    an oracle-built graph and a filter-starved walk through the HBM-resident fallback equal the oracle bit for bit; beyond 4 096 the
    open is refused.  The reference has no bound (nidx_vector/src/config.rs:170-173)."""
    rng = np.random.default_rng(d)
    n, nq, k = 1500, 6, 10
    x = unit_rows(rng, n, d)
    q = np.vstack([x[7][None, :], unit_rows(rng, nq - 1, d)])
    oseg = orc.Segment(x, similarity=orc.SIM_COSINE, order=orc.ORDER_WAVE64)
    gbytes = bytes(oseg.build_graph(seed=2).serialize_v2(n)[0])
    for method, fn in ((_lib.METHOD_BRUTE_FORCE, oseg.brute_force), (_lib.METHOD_HNSW, oseg.hnsw_search)):
        ov, osc, oc = gpu_search(x, 1, q, k, method=method, graph=gbytes)
        for i in range(nq):
            wv, ws = fn(q[i], k)
            assert oc[i] == len(wv), (method, i)
            assert np.array_equal(ov[i, : oc[i]], wv), (method, i, ov[i, : oc[i]], wv)
            assert np.array_equal(bits(osc[i, : oc[i]]), bits(ws)), (method, i)
    filt = orc.bitset(n, ones=list(range(0, n, 150)))
    ov, osc, oc = gpu_search(x, 1, q, k, method=_lib.METHOD_HNSW, graph=gbytes, filter_bits=filt)
    for i in range(nq):
        wv, ws = oseg.hnsw_search(q[i], k, filter_bits=filt)
        assert oc[i] == len(wv) and np.array_equal(ov[i, : oc[i]], wv) and np.array_equal(bits(osc[i, : oc[i]]), bits(ws)), i
    L = _lib.lib()
    cfg = _lib.VectorConfigC(4100, 1, 0, 0)
    xs = np.zeros((2, 4100), np.float32)
    seg = _lib.VectorSegmentC(xs.ctypes.data, 4100 * 4, 2, None, 2, None, 0, 0, None, 0, None, None)
    h = C.c_void_p()
    assert L.nidx_gpu_vector_open(C.byref(cfg), C.byref(seg), 1, C.byref(h)) == _lib.NIDX_ERR_UNSUPPORTED
