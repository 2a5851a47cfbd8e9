"""VectorCardinality::Multi: a paragraph owns several contiguous vectors; every search returns one
hit per paragraph — its best vector.  Device (C ABI) vs the CPU oracle, bit-exact ids / ranks / scores:
brute force = "best vector match per paragraph" (segment.rs:582-593), HNSW = NodeFilter::paragraphs (hnsw/search.rs:159-164),
RaBitQ brute force = best ESTIMATE per paragraph, then the re-rank (segment.rs:586-611)."""
import ctypes as C

import numpy as np
import pytest

from nucliadb_amd import _lib

pytestmark = pytest.mark.gpu


def bits(a):
    return np.asarray(a, dtype=np.float32).view(np.uint32)


def make(rng, n_para, d, vmax):
    num = rng.integers(1, vmax + 1, n_para).astype(np.uint32)
    first = np.concatenate([[0], np.cumsum(num)[:-1]]).astype(np.uint32)
    n = int(num.sum())
    centers = rng.normal(size=(n_para, d)).astype(np.float32)
    pov = np.repeat(np.arange(n_para, dtype=np.uint32), num)
    x = centers[pov] + rng.normal(size=(n, d)).astype(np.float32) * np.float32(0.3)
    x /= np.linalg.norm(x, axis=1, keepdims=True).astype(np.float32)
    return x.astype(np.float32), pov, first, num


class Index:
    def __init__(self, x, pov, n_para, sim, cardinality=1, alive=None, quantized=None):
        L = _lib.lib()
        n, d = x.shape
        cfg = _lib.VectorConfigC(d, sim, 0, cardinality, 0)
        self._keep = [x, pov, alive, quantized]
        seg = _lib.VectorSegmentC(x.ctypes.data, d * 4, n, pov.ctypes.data, n_para, None, 0, 0, None, 0,
                                  alive.ctypes.data if alive is not None else None, None,
                                  quantized.ctypes.data if quantized is not None else None, quantized.size if quantized is not None else 0)
        self.h = C.c_void_p()
        _lib.check(L.nidx_gpu_vector_open(C.byref(cfg), C.byref(seg), 1, C.byref(self.h)))

    def close(self):
        _lib.lib().nidx_gpu_vector_close(self.h)

    def build(self):
        L = _lib.lib()
        _lib.check(L.nidx_gpu_vector_build_hnsw(self.h, 0, 2))
        glen, elen = C.c_uint64(), C.c_uint64()
        _lib.check(L.nidx_gpu_vector_serialize_hnsw(self.h, 0, None, 0, C.byref(glen), None, 0, C.byref(elen)))
        g, e = np.zeros(glen.value, np.uint8), np.zeros(max(elen.value, 1), np.float32)
        _lib.check(L.nidx_gpu_vector_serialize_hnsw(self.h, 0, g.ctypes.data, g.size, C.byref(glen), e.ctypes.data, e.size, C.byref(elen)))
        return g, e[: elen.value]

    def search(self, q, k, method, min_score=-1.0, with_duplicates=True, filter_bits=None):
        L = _lib.lib()
        q = np.ascontiguousarray(q, np.float32)
        B = q.shape[0]
        op, ov, osc, oc = np.zeros((B, k), np.uint32), np.zeros((B, k), np.uint32), np.zeros((B, k), np.float32), np.zeros(B, np.uint32)
        params = _lib.VectorSearchParamsC(k, min_score, int(with_duplicates), method)
        fp = (C.c_void_p * 1)(filter_bits.ctypes.data) if filter_bits is not None else None
        _lib.check(L.nidx_gpu_vector_search(self.h, q.ctypes.data, B, C.byref(params), fp, None, op.ctypes.data, ov.ctypes.data,
                                            osc.ctypes.data, oc.ctypes.data, None))
        return op, ov, osc, oc


def check(got, want, pov, i):
    op, ov, osc, oc = got
    wv, ws = want
    assert oc[i] == len(wv), (i, oc[i], len(wv), ov[i], wv)
    assert np.array_equal(ov[i, : oc[i]], wv), (i, ov[i, : oc[i]], wv)
    assert np.array_equal(bits(osc[i, : oc[i]]), bits(ws))
    assert np.array_equal(op[i, : oc[i]], pov[wv])                      # the paragraph of every hit
    assert len(set(op[i, : oc[i]].tolist())) == oc[i]                    # one hit per paragraph


@pytest.mark.parametrize("sim", [0, 1])
def test_multi_vector_brute_force_and_hnsw_match_oracle(orc, sim):
    rng = np.random.default_rng(17 + sim)
    n_para, d, vmax, k = 5000, 64, 4, 10
    x, pov, first, num = make(rng, n_para, d, vmax)
    p7 = int(np.nonzero(num >= 2)[0][3])
    x[first[p7] + 1] = x[first[p7]]  # two identical vectors in one paragraph: max_by keeps the LAST of equal maxima
    nq = 10
    q = np.vstack([x[first[p7]][None, :], x[rng.integers(0, x.shape[0], nq - 1)] + rng.normal(size=(nq - 1, d)).astype(np.float32) * np.float32(0.1)])
    alive = orc.bitset(n_para, ones=np.nonzero(rng.random(n_para) < 0.9)[0].tolist() + [p7])
    filt = orc.bitset(n_para, ones=np.nonzero(rng.random(n_para) < 0.5)[0].tolist())
    idx = Index(x, pov, n_para, sim, alive=alive)
    try:
        graph, edges = idx.build()
        bf = idx.search(q, k, _lib.METHOD_BRUTE_FORCE)
        bf_f = idx.search(q, k, _lib.METHOD_BRUTE_FORCE, min_score=0.3, filter_bits=filt)
        hn = idx.search(q, k, _lib.METHOD_HNSW)
        hn_f = idx.search(q, k, _lib.METHOD_HNSW, min_score=0.3, with_duplicates=False, filter_bits=filt)
        bf100 = idx.search(q, 100, _lib.METHOD_BRUTE_FORCE)      # 100 x 4 vectors per paragraph = 400 candidate vectors (<= 512)
        with pytest.raises(_lib.NidxGpuError):
            idx.search(q, 200, _lib.METHOD_BRUTE_FORCE)          # 200 x 4 > 512
    finally:
        idx.close()
    oseg = orc.Segment(x, similarity=sim, vec_paragraph=pov, para_first_vec=first, para_num_vec=num, alive=alive, n_paragraphs=n_para,
                       graph=orc.Hnsw.deserialize_v2(graph, edges))
    for i in range(nq):
        check(bf100, oseg.brute_force(q[i], 100), pov, i)
        check(bf, oseg.brute_force(q[i], k), pov, i)
        check(bf_f, oseg.brute_force(q[i], k, min_score=0.3, filter_bits=alive & filt), pov, i)
        check(hn, oseg.hnsw_search(q[i], k, multi=True), pov, i)
        check(hn_f, oseg.hnsw_search(q[i], k, min_score=0.3, with_duplicates=False, filter_bits=alive & filt, multi=True), pov, i)
    assert bf[1][0, 0] == first[p7] + 1  # the later of the two identical vectors represents the paragraph


@pytest.mark.parametrize("sim", [0, 1])
def test_multi_vector_matrix_core_scans(orc, sim):
    """The f32-MFMA scan (SERIAL_FMA order) on a multi-vector segment: k x vmax vectors, one hit per paragraph, bit-exact vs the
    oracle in that order; the bf16 fallback: exact (WAVE64) scores of the vectors it returns, one hit per paragraph, the exact
    scan's paragraphs up to bf16 ranking error."""
    rng = np.random.default_rng(41 + sim)
    n_para, d, vmax, k = 4000, 96, 3, 10
    x, pov, first, num = make(rng, n_para, d, vmax)
    p7 = int(np.nonzero(num >= 2)[0][5])
    x[first[p7] + 1] = x[first[p7]]
    nq = 140   # two query tiles of the MFMA scan
    q = np.vstack([x[first[p7]][None, :], x[rng.integers(0, x.shape[0], nq - 1)] + rng.normal(size=(nq - 1, d)).astype(np.float32) * np.float32(0.1)])
    alive = orc.bitset(n_para, ones=np.nonzero(rng.random(n_para) < 0.9)[0].tolist() + [p7])
    filt = orc.bitset(n_para, ones=np.nonzero(rng.random(n_para) < 0.5)[0].tolist())
    idx = Index(x, pov, n_para, sim, alive=alive)
    try:
        mf = idx.search(q, k, _lib.METHOD_BRUTE_FORCE_MFMA)
        mf_f = idx.search(q, k, _lib.METHOD_BRUTE_FORCE_MFMA, min_score=0.3, filter_bits=filt)
        mf21 = idx.search(q, 21, _lib.METHOD_BRUTE_FORCE_MFMA)   # 21 x 3 = 63 vectors: the wide lists
        ex = idx.search(q, k, _lib.METHOD_BRUTE_FORCE)
        bf = idx.search(q, k, _lib.METHOD_BRUTE_FORCE_BF16)      # 10 x 3 = 30 of the 32 candidates
        with pytest.raises(_lib.NidxGpuError):
            idx.search(q, 22, _lib.METHOD_BRUTE_FORCE_MFMA)      # 66 > 64
        with pytest.raises(_lib.NidxGpuError):
            idx.search(q, 11, _lib.METHOD_BRUTE_FORCE_BF16)      # 33 > 32
    finally:
        idx.close()
    oseg = orc.Segment(x, similarity=sim, vec_paragraph=pov, para_first_vec=first, para_num_vec=num, alive=alive, n_paragraphs=n_para,
                       order=orc.ORDER_SERIAL_FMA)
    for i in list(range(12)) + [nq - 1]:
        check(mf, oseg.brute_force(q[i], k), pov, i)
        check(mf_f, oseg.brute_force(q[i], k, min_score=0.3, filter_bits=alive & filt), pov, i)
        check(mf21, oseg.brute_force(q[i], 21), pov, i)
    assert mf[1][0, 0] == first[p7] + 1
    # bf16: exact scores of what it returns, sorted, distinct paragraphs, and the exact scan's paragraphs
    op, ov, osc, oc = bf
    assert np.array_equal(oc, ex[3])
    same = 0
    for i in range(nq):
        c = int(oc[i])
        assert len(set(op[i, :c].tolist())) == c and np.array_equal(op[i, :c], pov[ov[i, :c]])
        keys = [(-float(osc[i, j]), int(ov[i, j])) for j in range(c)]
        assert keys == sorted(keys)
        for j in range(min(c, 3)):
            assert bits([osc[i, j]])[0] == bits([orc.similarity(x[ov[i, j]], q[i], sim)])[0]
        same += len(set(op[i, :c].tolist()) & set(ex[0][i, : ex[3][i]].tolist()))
    assert same / float(ex[3].sum()) >= 0.99


def test_multi_vector_rabitq_brute_force_matches_oracle(orc):
    rng = np.random.default_rng(23)
    n_para, d, vmax, k = 3000, 128, 3, 10
    x, pov, first, num = make(rng, n_para, d, vmax)
    nq = 8
    q = x[rng.integers(0, x.shape[0], nq)] + rng.normal(size=(nq, d)).astype(np.float32) * np.float32(0.05)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    q = q.astype(np.float32)
    oseg = orc.Segment(x, similarity=orc.SIM_DOT, vec_paragraph=pov, para_first_vec=first, para_num_vec=num, n_paragraphs=n_para)
    quant = oseg.quantize()
    idx = Index(x, pov, n_para, 0, quantized=quant)
    try:
        got = idx.search(q, k, _lib.METHOD_RABITQ_BRUTE_FORCE)
    finally:
        idx.close()
    for i in range(nq):
        check(got, oseg.brute_force(q[i], k), pov, i)


def test_single_cardinality_rejects_multi_vector_paragraphs():
    x = np.eye(4, 8, dtype=np.float32)
    pov = np.array([0, 0, 1, 2], np.uint32)
    with pytest.raises(_lib.NidxGpuError) as e:
        Index(x, pov, 3, 0, cardinality=0)
    assert e.value.code == _lib.NIDX_ERR_INVALID_CONFIGURATION


def test_maxsim_reference_case():
    """nidx_vector/tests/test_maxsim.rs:23-140: three paragraphs of three one-hot vectors, a two-vector query; the score is
    the number of query vectors a paragraph contains, and min_score applies to the maxsim score, not to single vectors."""
    import uuid

    from nucliadb_amd.vector import Elem, PrefilterResult, VectorCardinality, VectorConfig, VectorSearcher, VectorSearchRequest, segment_create

    def onehots(*idx):
        m = np.zeros((len(idx), 5), np.float32)
        for r, i in enumerate(idx):
            m[r, i] = 1.0
        return m.reshape(-1).tolist()

    config = VectorConfig.for_paragraphs(5)
    config.vector_cardinality = VectorCardinality.Multi
    rid = uuid.uuid4().hex
    seg = segment_create([Elem(f"{rid}/f/d0/0-123", onehots(1, 2, 4)), Elem(f"{rid}/f/d1/0-123", onehots(0, 1, 2)),
                          Elem(f"{rid}/f/d2/0-123", onehots(0, 2, 3))], config)
    s = VectorSearcher.open(config, [(seg, 1)])
    query = onehots(0, 3)
    r = s.search(VectorSearchRequest(vector=query, result_per_page=1, min_score=-10.0), PrefilterResult.All)
    assert [d.doc_id for d in r.documents] == [f"{rid}/f/d2/0-123"] and r.documents[0].score == 2.0
    r = s.search(VectorSearchRequest(vector=query, result_per_page=10, min_score=1.5), PrefilterResult.All)
    assert [(d.doc_id, d.score) for d in r.documents] == [(f"{rid}/f/d2/0-123", 2.0)]
    r = s.search(VectorSearchRequest(vector=query, result_per_page=10, min_score=0.5), PrefilterResult.All)
    assert [(d.doc_id, d.score) for d in r.documents] == [(f"{rid}/f/d2/0-123", 2.0), (f"{rid}/f/d1/0-123", 1.0)]
    r = s.search(VectorSearchRequest(vector=query, result_per_page=10, min_score=-10.0), PrefilterResult.All)
    assert [(d.doc_id, d.score) for d in r.documents] == [(f"{rid}/f/d2/0-123", 2.0), (f"{rid}/f/d1/0-123", 1.0), (f"{rid}/f/d0/0-123", 0.0)]
    s.close()


@pytest.mark.parametrize("sim", [0, 1])
def test_maxsim_matches_oracle(orc, sim):
    """search_multi_vector (searcher.rs:345-394) composed from oracle pieces: per query vector the oracle's segment search
    (max(k, 10) hits, one per paragraph), the union of the paragraphs, orc_maxsim, `> min_score`, top k."""
    rng = np.random.default_rng(31 + sim)
    n_para, d, vmax, k = 1500, 32, 4, 5
    x, pov, first, num = make(rng, n_para, d, vmax)
    idx = Index(x, pov, n_para, sim)
    oseg = orc.Segment(x, similarity=sim, vec_paragraph=pov, para_first_vec=first, para_num_vec=num, n_paragraphs=n_para)
    L = _lib.lib()
    try:
        for trial in range(6):
            nqv = int(rng.integers(1, 5))
            qv = (x[rng.integers(0, x.shape[0], nqv)] + rng.normal(size=(nqv, d)).astype(np.float32) * np.float32(0.2)).astype(np.float32)
            min_score = float(rng.choice([-10.0, 0.5, 1.2]))
            qoff = np.array([0, nqv], np.uint64)
            oseg_, opar, osc, ocnt = np.zeros((1, k), np.uint32), np.zeros((1, k), np.uint32), np.zeros((1, k), np.float32), np.zeros(1, np.uint32)
            params = _lib.VectorSearchParamsC(k, min_score, 0, _lib.METHOD_AUTO)
            _lib.check(L.nidx_gpu_vector_search_maxsim(idx.h, qv.ctypes.data, qoff.ctypes.data, 1, C.byref(params), None, oseg_.ctypes.data,
                                                       opar.ctypes.data, osc.ctypes.data, ocnt.ctypes.data))
            paras = set()
            for v in qv:
                wv, _, _ = oseg.search(v, max(k, 10), min_score=-3.0e38, with_duplicates=True)
                paras |= {int(pov[a]) for a in wv}
            scored = []
            for p in sorted(paras):
                sc = orc.maxsim(qv, x[first[p]: first[p] + num[p]], sim)
                if sc > min_score:
                    scored.append((-sc, p))
            scored.sort()
            want = scored[:k]
            n = int(ocnt[0])
            assert n == len(want)
            assert opar[0, :n].tolist() == [p for _, p in want]
            assert np.array_equal(bits(osc[0, :n]), bits([-s for s, _ in want]))
    finally:
        idx.close()
