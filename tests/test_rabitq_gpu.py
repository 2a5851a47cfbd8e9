"""GPU parity tests of the RaBitQ arms: device encode / query codes / popcount estimates /
error-bounded re-rank / RaBitQ HNSW + brute force through the C ABI vs the CPU oracle on the same seeded inputs.
Bar: bit-exact record bytes, vector addresses, ranks and score bit patterns."""
import ctypes as C

import numpy as np
import pytest

from nucliadb_amd import _lib
from nucliadb_amd.vector import Similarity, VectorConfig, VectorSearcher, VectorSearchRequest, VectorSegment

pytestmark = pytest.mark.gpu


def unit_rows(rng, n, d):
    x = rng.uniform(-1, 1, (n, d)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True).astype(np.float32)
    return x


def clustered(rng, n, d, clusters=40, spread=0.05):
    centers = unit_rows(rng, clusters, d)
    x = centers[rng.integers(0, clusters, n)] + rng.normal(size=(n, d)).astype(np.float32) * np.float32(spread / np.sqrt(d))
    x /= np.linalg.norm(x, axis=1, keepdims=True).astype(np.float32)
    return x.astype(np.float32)


def bits(a):
    return np.asarray(a, dtype=np.float32).view(np.uint32)


class Index:
    """One Dot segment through the C ABI, optionally with vectors.quant / hnsw.graph handed in."""

    def __init__(self, x, graph=None, quantized=None, alive=None, flags=0):
        L = _lib.lib()
        n, d = x.shape
        self.x, self.n, self.d = x, n, d
        cfg = _lib.VectorConfigC(d, 0, 0, 0, flags)
        self._keep = [x, graph, quantized, alive]
        g = np.frombuffer(graph, np.uint8) if graph is not None else None
        self._keep.append(g)
        seg = _lib.VectorSegmentC(x.ctypes.data, d * 4, n, None, n, g.ctypes.data if g is not None else None,
                                  len(graph) if graph is not None else 0, 0, None, 0,
                                  alive.ctypes.data if alive is not None else None, None,
                                  quantized.ctypes.data if quantized is not None else None,
                                  quantized.size if quantized is not None else 0)
        self.h = C.c_void_p()
        _lib.check(L.nidx_gpu_vector_open(C.byref(cfg), C.byref(seg), 1, C.byref(self.h)))

    def close(self):
        _lib.lib().nidx_gpu_vector_close(self.h)

    def quantize(self):
        _lib.check(_lib.lib().nidx_gpu_vector_quantize(self.h, 0))
        n = C.c_uint64()
        _lib.check(_lib.lib().nidx_gpu_vector_serialize_quantized(self.h, 0, None, 0, C.byref(n)))
        out = np.zeros(n.value, np.uint8)
        _lib.check(_lib.lib().nidx_gpu_vector_serialize_quantized(self.h, 0, out.ctypes.data, out.size, C.byref(n)))
        return out.reshape(self.n, self.d // 8 + 8)

    def build(self):
        L = _lib.lib()
        _lib.check(L.nidx_gpu_vector_build_hnsw(self.h, 0, 2))
        glen, elen = C.c_uint64(), C.c_uint64()
        _lib.check(L.nidx_gpu_vector_serialize_hnsw(self.h, 0, None, 0, C.byref(glen), None, 0, C.byref(elen)))
        g, e = np.zeros(glen.value, np.uint8), np.zeros(elen.value, np.float32)
        _lib.check(L.nidx_gpu_vector_serialize_hnsw(self.h, 0, g.ctypes.data, g.size, C.byref(glen), e.ctypes.data, e.size, C.byref(elen)))
        return g, e

    def search(self, q, k, method, min_score=-1.0, with_duplicates=True, filter_bits=None):
        L = _lib.lib()
        q = np.ascontiguousarray(q, np.float32)
        B = q.shape[0]
        ov, osc, oc = np.zeros((B, k), np.uint32), np.zeros((B, k), np.float32), np.zeros(B, np.uint32)
        om = np.zeros(1, np.int32)
        params = _lib.VectorSearchParamsC(k, min_score, int(with_duplicates), method)
        fp = (C.c_void_p * 1)(filter_bits.ctypes.data) if filter_bits is not None else None
        _lib.check(L.nidx_gpu_vector_search(self.h, q.ctypes.data, B, C.byref(params), fp, None, None, ov.ctypes.data,
                                            osc.ctypes.data, oc.ctypes.data, om.ctypes.data))
        return ov, osc, oc, int(om[0])


def same_hits(got, want, i):
    ov, osc, oc = got
    wv, ws = want
    assert oc[i] == len(wv), (i, oc[i], len(wv), ov[i], wv)
    assert np.array_equal(ov[i, : oc[i]], wv), (i, ov[i, : oc[i]], wv)
    assert np.array_equal(bits(osc[i, : oc[i]]), bits(ws)), (i, osc[i, : oc[i]], ws)


# ---- EncodedVector::encode ----------------------------------------------------------------------------------
@pytest.mark.parametrize("d", [64, 128, 192, 448, 768, 1024, 1536, 2048])
def test_encode_bytes_match_oracle(orc, d):
    rng = np.random.default_rng(d)
    x = unit_rows(rng, 301, d)
    x[3, 5] = 0.0
    x[4, :9] = -0.0
    x[5] = np.abs(x[5])
    x[6] = -np.abs(x[6])
    idx = Index(x)
    try:
        got = idx.quantize()
    finally:
        idx.close()
    want = orc.rabitq_encode(x, orc.ORDER_WAVE64)
    assert np.array_equal(got, want), np.nonzero((got != want).any(axis=1))[0][:10]


def test_quantize_needs_a_quantizable_config():
    x = unit_rows(np.random.default_rng(1), 10, 100)  # 100 % 64 != 0 (config.rs:170-173)
    idx = Index(x)
    try:
        with pytest.raises(_lib.NidxGpuError) as e:
            idx.quantize()
        assert e.value.code == _lib.NIDX_ERR_INVALID_CONFIGURATION
    finally:
        idx.close()


# ---- brute force, RaBitQ arm --------------------------------------------------------------------------------
@pytest.mark.parametrize("n,d,nq,k", [(1, 64, 1, 3), (70, 64, 3, 100), (5000, 128, 9, 10), (20000, 768, 12, 10),
                                      (6000, 1024, 5, 1), (3000, 192, 4, 64), (2000, 2048, 3, 7), (4000, 256, 3, 256),
                                      (5000, 128, 4, 300), (3000, 768, 3, 512)])   # pages up to NIDX_K_MAX (rabitq.rs:30-36 re-ranks up to 2 000)
def test_rabitq_brute_force_matches_oracle(orc, n, d, nq, k):
    rng = np.random.default_rng(n + d)
    x = clustered(rng, n, d) if n > 100 else unit_rows(rng, n, d)
    q = np.vstack([x[rng.integers(n)] + rng.normal(size=d).astype(np.float32) * np.float32(0.3 / np.sqrt(d)) for _ in range(nq)])
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    q = q.astype(np.float32)
    oseg = orc.Segment(x, similarity=orc.SIM_DOT)
    quant = oseg.quantize()
    idx = Index(x, quantized=quant)
    try:
        got = idx.search(q, k, _lib.METHOD_RABITQ_BRUTE_FORCE)[:3]
    finally:
        idx.close()
    for i in range(nq):
        same_hits(got, oseg.brute_force(q[i], k), i)


def test_rabitq_brute_force_filters_min_score_ties(orc):
    rng = np.random.default_rng(77)
    n, d, k = 7000, 128, 10
    x = clustered(rng, n, d)
    x[200:230] = x[9]  # identical rows: identical codes, estimates and real scores
    q = np.vstack([x[9][None, :], unit_rows(rng, 4, d)]).astype(np.float32)
    alive = orc.bitset(n, fill=True)
    for dead in (9, 200, 201, 6999):
        alive[dead >> 6] &= ~np.uint64(1 << (dead & 63))
    filt = orc.bitset(n, ones=np.nonzero(rng.random(n) < 0.2)[0].tolist() + list(range(200, 230)))
    oseg = orc.Segment(x, similarity=orc.SIM_DOT, alive=alive)
    quant = oseg.quantize()
    idx = Index(x, quantized=quant, alive=alive)
    try:
        for fb, ob in ((None, alive), (filt, alive & filt)):
            for ms in (-1.0, 0.3, 0.95, 0.9999):
                got = idx.search(q, k, _lib.METHOD_RABITQ_BRUTE_FORCE, min_score=ms, filter_bits=fb)[:3]
                for i in range(q.shape[0]):
                    same_hits(got, oseg.brute_force(q[i], k, min_score=ms, filter_bits=ob), i)
    finally:
        idx.close()


# ---- HNSW, RaBitQ arm ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,d,k", [(3000, 128, 10), (20000, 768, 10), (8000, 256, 1), (8000, 256, 30), (6000, 1024, 5), (9000, 128, 300), (6000, 768, 512)])
@pytest.mark.parametrize("walk", ["pipelined", "pipelined_no_seen_cache", "plain", "two_waves"])
def test_rabitq_hnsw_matches_oracle(orc, monkeypatch, n, d, k, walk):
    """The three walk kernels give the same bits: `pipelined` (default: one wave, the predicted next expansion's loads in flight under
    the admissions, rollback on a mispredict; neighbours known to be visited — an LDS cache of ids — ask neither the bitset nor their
    code; NIDX_GPU_RABITQ_SEEN=0 switches that cache off), `plain` (rounds 1-4, NIDX_GPU_RABITQ_PIPE=0), `two_waves`
    (NIDX_GPU_RABITQ_WAVES=2: a fetcher wave runs the predicted expansion)."""
    if walk == "pipelined_no_seen_cache":
        monkeypatch.setenv("NIDX_GPU_RABITQ_SEEN", "0")
    elif walk == "plain":
        monkeypatch.setenv("NIDX_GPU_RABITQ_PIPE", "0")
    elif walk == "two_waves":
        if not _lib.lib().nidx_gpu_build_features() & _lib.FEATURE_RABITQ_EXPERIMENTS:
            pytest.skip("the two-wave walk is measurement material: only in a `make EXPERIMENTS=1` library")
        monkeypatch.setenv("NIDX_GPU_RABITQ_WAVES", "2")
    rng = np.random.default_rng(n * 7 + d + k)
    x = clustered(rng, n, d, clusters=60, spread=0.3)
    nq = 12
    q = np.vstack([x[rng.integers(n)] + rng.normal(size=d).astype(np.float32) * np.float32(0.2 / np.sqrt(d)) for _ in range(nq)])
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    q = q.astype(np.float32)
    idx = Index(x)
    try:
        graph, edges = idx.build()
        quant = idx.quantize()
        got = idx.search(q, k, _lib.METHOD_RABITQ_HNSW)[:3]
        alive = orc.bitset(n, fill=True)
        filt = orc.bitset(n, ones=np.nonzero(rng.random(n) < 0.5)[0].tolist())
        got_f = idx.search(q, k, _lib.METHOD_RABITQ_HNSW, min_score=0.2, with_duplicates=False, filter_bits=filt)[:3]
    finally:
        idx.close()
    oseg = orc.Segment(x, similarity=orc.SIM_DOT, graph=orc.Hnsw.deserialize_v2(graph, edges), quantized=quant)
    for i in range(nq):
        same_hits(got, oseg.hnsw_search(q[i], k), i)
        same_hits(got_f, oseg.hnsw_search(q[i], k, min_score=0.2, with_duplicates=False, filter_bits=alive & filt), i)


def test_auto_routes_like_the_reference_and_recall(orc):
    """OpenSegment::_search (segment.rs:506-555): with a quantized store AUTO takes the RaBitQ arm the cost model picks;
    DISABLE_RABITQ_SEARCH turns it off; a store handed in (any summation order) is used as it is."""
    rng = np.random.default_rng(2024)
    n, d, k = 60000, 128, 10
    x = clustered(rng, n, d, clusters=200, spread=0.4)
    nq = 64
    q = np.vstack([x[rng.integers(n)] + rng.normal(size=d).astype(np.float32) * np.float32(0.2 / np.sqrt(d)) for _ in range(nq)])
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    q = q.astype(np.float32)
    quant = orc.rabitq_encode(x, orc.ORDER_HASWELL)  # as a CPU writer would have produced it
    plain = Index(x)
    try:
        graph, edges = plain.build()
        exact = plain.search(q, k, _lib.METHOD_BRUTE_FORCE)
    finally:
        plain.close()
    assert orc.use_hnsw(n, n, k, True)
    idx = Index(x, graph=graph.tobytes(), quantized=quant)
    try:
        ov, osc, oc, method = idx.search(q, k, _lib.METHOD_AUTO)
        assert method == _lib.METHOD_RABITQ_HNSW
        sparse = orc.bitset(n, ones=np.nonzero(rng.random(n) < 0.02)[0].tolist())
        fv, fs, fc, fmethod = idx.search(q, k, _lib.METHOD_AUTO, filter_bits=sparse)
        assert fmethod == _lib.METHOD_RABITQ_BRUTE_FORCE
    finally:
        idx.close()
    oseg = orc.Segment(x, similarity=orc.SIM_DOT, graph=orc.Hnsw.deserialize_v2(graph, edges), quantized=quant)
    hit = 0
    for i in range(nq):
        wv, ws, m = oseg.search(q[i], k)
        assert m == "hnsw"
        same_hits((ov, osc, oc), (wv, ws), i)
        wv, ws, m = oseg.search(q[i], k, filter_bits=sparse)
        assert m == "brute force"
        same_hits((fv, fs, fc), (wv, ws), i)
        hit += len(set(ov[i, : oc[i]].tolist()) & set(exact[0][i, : exact[2][i]].tolist()))
    assert hit / (nq * k) >= 0.95, hit / (nq * k)  # ef = 1000 estimates + raw re-rank: recall vs the exact scan
    off = Index(x, graph=graph.tobytes(), quantized=quant, flags=_lib.CONFIG_DISABLE_RABITQ_SEARCH)
    try:
        assert off.search(q[:2], k, _lib.METHOD_AUTO)[3] == _lib.METHOD_HNSW
    finally:
        off.close()


def spill_count(idx):
    n = C.c_uint64()
    _lib.check(_lib.lib().nidx_gpu_vector_spill_stats(idx.h, C.byref(n)))
    return n.value


@pytest.mark.parametrize("k", [1, 3])
def test_more_evicted_ties_than_the_lds_list_holds(orc, monkeypatch, k):
    """The reference's candidate heap is an unbounded BinaryHeap (hnsw/search.rs:252-299): an entry evicted from the result set
    stays a candidate while its score still EQUALS the worst result's (`cs < ws` does not stop on it, search.rs:271-277).  The walk
    keeps 64 such entries in LDS; identical vectors are common in NucliaDB corpora (the reference carries RepCounter for them,
    search.rs:386-412), and a group of hundreds of identical vectors at the bottom of a full result set produces more.  They go to
    a per-query region in HBM (ef entries always suffice: csrc/rabitq.hip RqLayer) — the answer is the oracle's bit for bit,
    without the exact fallback.

    The graph is made by hand so that the walk (ef = 100 k) depends on the LATE ties: the result set fills with identical copies
    A_1 .. A_ef (equal estimates; among equal scores the lower address ranks first), 70 better nodes H then evict A_ef .. A_(ef-69)
    one by one — all still tie with the worst result — and only the six evicted last (the 65th .. 70th tie) have an edge to z,
    the true nearest neighbour.  NIDX_GPU_RABITQ_TIE_SPILL=0 takes the HBM region away: those queries then raise
    NIDX_FLAG_POOL_INEXACT (counted by nidx_gpu_vector_spill_stats), which shows that the scenario does overflow the list."""
    rng = np.random.default_rng(64 + k)
    d, ef = 768, 100 * k
    m = -(-(ef - 60 + 8) // 58)                  # fill expansions: A_1 .. A_m reveal 58 more copies each
    n_a = 60 + 58 * m
    q0 = unit_rows(rng, 1, d)[0]

    def with_dot(t):                              # a unit vector with <q0, v> = t
        r = unit_rows(rng, 1, d)[0]
        r -= np.float32(r @ q0) * q0
        r /= np.linalg.norm(r)
        return (np.float32(t) * q0 + np.float32(np.sqrt(1.0 - t * t)) * r).astype(np.float32)

    E, A0, H0, Z = 0, 1, 1 + n_a, 1 + n_a + 70    # addresses: e | A_1 .. A_n_a | H_1 .. H_70 | z
    a_vec = with_dot(0.25)
    x = np.vstack([with_dot(-0.3)[None], np.repeat(a_vec[None], n_a, axis=0), np.vstack([with_dot(0.55 + 0.002 * j) for j in range(70)]),
                   with_dot(0.97)[None]]).astype(np.float32)
    n = x.shape[0]
    A = lambda i: A0 + i - 1                      # noqa: E731 — A_i, 1-based
    edges = {v: [] for v in range(n)}
    edges[E] = [A(i) for i in range(1, 61)]
    for i in range(1, m + 1):
        edges[A(i)] = [A(j) for j in range(60 + 58 * (i - 1) + 1, 60 + 58 * i + 1)]
    edges[A(m + 1)] = [H0 + j for j in range(60)]
    edges[A(m + 2)] = [H0 + j for j in range(60, 70)]
    for i in range(m + 3, n_a + 1):
        edges[A(i)] = [A(1), A(2)]                # nothing new
    for i in range(ef - 69, ef - 63):             # the six copies evicted last
        edges[A(i)] = [A(1), Z]
    assert m + 2 < ef - 69
    g = orc.Hnsw.new()
    for v in range(n):
        g.add_node(v, 0)
    for v, e_ in edges.items():
        g.set_edges(0, v, e_, [float(x[v] @ x[t]) for t in e_])
    g.set_entry_point(E, 0)
    graph, gedges = g.serialize_v2(n)
    q = np.vstack([q0, q0 + unit_rows(rng, 1, d)[0] * np.float32(0.02), q0 + unit_rows(rng, 1, d)[0] * np.float32(0.02)]).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    nq = q.shape[0]
    idx = Index(x, graph=bytes(graph))
    try:
        quant = idx.quantize()
        before = spill_count(idx)
        got = idx.search(q, k, _lib.METHOD_RABITQ_HNSW)[:3]
        assert spill_count(idx) == before         # no query needed the fallback
        monkeypatch.setenv("NIDX_GPU_RABITQ_TIE_SPILL", "0")
        idx.search(q, k, _lib.METHOD_RABITQ_HNSW)
        overflowed = spill_count(idx) - before
        monkeypatch.delenv("NIDX_GPU_RABITQ_TIE_SPILL")
    finally:
        idx.close()
    oseg = orc.Segment(x, similarity=orc.SIM_DOT, graph=orc.Hnsw.deserialize_v2(graph, gedges), quantized=quant)
    for i in range(nq):
        want = oseg.hnsw_search(q[i], k)
        assert want[0][0] == Z, "the hand-made scenario does not lead the oracle to z: estimates out of the designed order?"
        same_hits(got, want, i)
    assert overflowed == nq, "the scenario does not overflow the 64-entry tie list: the test is vacuous"
