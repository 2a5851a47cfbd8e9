"""The reference's own nidx_vector tests, replayed through the host mirror (same names and argument
meaning) on the GPU path: nidx/nidx_vector/tests/{test_basic_search,test_min_score}.rs,
nidx/tests/integration/vector_normalization.rs, nidx_vector/src/searcher.rs test_key_prefix_search."""
import uuid

import numpy as np
import pytest

from nucliadb_amd import _lib
from nucliadb_amd.vector import (And, Elem, FieldId, FilterOperator, Literal, Not, Or, PrefilterResult, Similarity,
                                 VectorConfig, VectorSearcher, VectorSearchRequest, segment_create)

pytestmark = pytest.mark.gpu
DIMENSION = 64


def sentence(i, d=DIMENSION):
    v = [0.0] * d
    v[i] = 1.0
    return v


@pytest.mark.parametrize("similarity", [Similarity.Dot, Similarity.Cosine])
def test_basic_search(similarity):
    """test_basic_search.rs:38-145: 64 orthogonal one-hot vectors."""
    config = VectorConfig.for_paragraphs(DIMENSION)
    config.similarity = similarity
    rid = str(uuid.uuid4())
    segment = segment_create([Elem(f"{rid}/a/title/0-{i}", sentence(i)) for i in range(DIMENSION)], config)
    searcher = VectorSearcher.open(config, [(segment, 1)])
    results = searcher.search(VectorSearchRequest(vector=sentence(5), result_per_page=10, min_score=-1.0), PrefilterResult.All)
    assert len(results.documents) == 10
    assert results.documents[0].doc_id == f"{rid}/a/title/0-5"
    assert results.documents[0].score > 0.9999
    assert results.documents[1].score < 0.0001
    vector = [0.0] * DIMENSION
    vector[42], vector[43], vector[44], vector[45] = 0.7, 0.59, 0.35, 0.2
    results = searcher.search(VectorSearchRequest(vector=vector, result_per_page=10, min_score=-1.0), PrefilterResult.All)
    assert len(results.documents) == 10
    for rank, (idx, floor) in enumerate([(42, 0.6), (43, 0.5), (44, 0.3), (45, 0.15)]):
        assert results.documents[rank].doc_id == f"{rid}/a/title/0-{idx}"
        assert results.documents[rank].score > floor
    assert results.documents[5].score == 0.0
    searcher.close()


_rand = np.random.default_rng(7)


def _resource(dim, labels=(), value=None):
    """tests/common/mod.rs:46-80: one sentence `[0.5, 0.5, 0.5, rand::random()]` per resource."""
    rid = str(uuid.uuid4())
    v = [0.5] * (dim - 1) + [float(_rand.random())] if value is None else value
    return rid, [Elem(f"{rid}/a/title/0-5", v, labels=list(labels))]


def test_deletions():
    """test_basic_search.rs:147-216: resource-level and field-level deletions with seq ordering."""
    config = VectorConfig.for_paragraphs(4)
    r1, e1 = _resource(4)
    r2, e2 = _resource(4)
    s1, s2 = segment_create(e1, config), segment_create(e2, config)
    request = VectorSearchRequest(vector=[0.0] * 4, result_per_page=10, min_score=-1.0)
    searcher = VectorSearcher.open(config, [(s1, 1), (s2, 2)])
    assert len(searcher.search(request, PrefilterResult.All).documents) == 2
    searcher = VectorSearcher.open(config, [(s1, 1), (s2, 2)], [(r1, 3)])
    docs = searcher.search(request, PrefilterResult.All).documents
    assert len(docs) == 1 and docs[0].doc_id.startswith(r2)
    searcher = VectorSearcher.open(config, [(s1, 1), (s2, 2)], [(f"{r2}/a/title", 3)])
    docs = searcher.search(request, PrefilterResult.All).documents
    assert len(docs) == 1 and docs[0].doc_id.startswith(r1)
    # a deletion older than the segment does not apply (lib.rs:166-200: seq > segment seq)
    searcher = VectorSearcher.open(config, [(s1, 5), (s2, 6)], [(r1, 3)])
    assert len(searcher.search(request, PrefilterResult.All).documents) == 2


def test_filtered_search():
    """test_basic_search.rs:218-400: boolean label formulas + prefilter AND/OR."""
    config = VectorConfig.for_paragraphs(4)
    L = lambda i: f"/l/labelset/label_{i}"  # noqa: E731  (test_basic_search.rs:232-241)
    work = [[L(i), L((i % 2) + 8)] for i in range(4)]
    rids, segs = [], []
    for i, labels in enumerate(work):
        rid, elems = _resource(4, labels)
        rids.append(rid)
        segs.append((segment_create(elems, config), i + 1))
    searcher = VectorSearcher.open(config, segs)

    def search(formula=None, prefilter=PrefilterResult.All, op=FilterOperator.And):
        req = VectorSearchRequest(vector=[0.0] * 4, result_per_page=10, min_score=-1.0, filtering_formula=formula, filter_operator=op)
        return {d.doc_id.split("/")[0] for d in searcher.search(req, prefilter).documents}

    assert search() == set(rids)
    assert search(Literal(L(0))) == {rids[0]}
    assert search(Literal(L(8))) == {rids[0], rids[2]}
    assert search(And([Literal(L(8)), Literal(L(2))])) == {rids[2]}
    assert search(Or([Literal(L(0)), Literal(L(3))])) == {rids[0], rids[3]}
    assert search(Not(Literal(L(9)))) == {rids[0], rids[2]}
    assert search(And([Literal(L(8)), Not(Literal(L(0)))])) == {rids[2]}
    assert search(Literal("/l/labelset")) == set(rids)          # label.fst prefix search: a parent matches its children
    assert search(Literal("/l/labelset/label")) == set()        # ...but only at a path boundary
    some = PrefilterResult.some([FieldId(uuid.UUID(rids[1]), "/a/title"), FieldId(uuid.UUID(rids[2]), None)])
    assert search(prefilter=some) == {rids[1], rids[2]}
    assert search(Literal(L(8)), some) == {rids[2]}
    assert search(Literal(L(0)), some, FilterOperator.Or) == {rids[0], rids[1], rids[2]}
    wrong_field = PrefilterResult.some([FieldId(uuid.UUID(rids[1]), "/a/other")])
    assert search(prefilter=wrong_field) == set()


def test_min_score():
    """test_min_score.rs:61-164: 5 one-hot vectors, min_score 0.5 keeps exactly the matching one."""
    config = VectorConfig.for_paragraphs(5)
    rid = str(uuid.uuid4())
    segment = segment_create([Elem(f"{rid}/a/title/0-{i}", sentence(i, 5)) for i in range(5)], config)
    searcher = VectorSearcher.open(config, [(segment, 1)])
    q = sentence(2, 5)
    assert len(searcher.search(VectorSearchRequest(vector=q, result_per_page=10, min_score=0.5), PrefilterResult.All).documents) == 1
    assert len(searcher.search(VectorSearchRequest(vector=q, result_per_page=10, min_score=-1.0), PrefilterResult.All).documents) == 5
    # and through the HNSW path (closest_up_nodes cuts at min_score, hnsw/search.rs:205-216)
    searcher.build_hnsw(0)
    for ms, n in ((0.5, 1), (-1.0, 5)):
        req = VectorSearchRequest(vector=q, result_per_page=10, min_score=ms, with_duplicates=True)
        assert len(searcher.search(req, PrefilterResult.All, method=_lib.METHOD_HNSW).documents) == n


def test_min_score_respected_with_rabitq_brute_force():
    """test_min_score.rs:61-164 as written: DIMENSION = 64 and Dot make the index quantizable, five one-hot vectors keep the
    segment on the RaBitQ brute-force arm; the huge error bound of a one-hot code lets every candidate through to the
    re-rank, which must re-apply min_score (the bug the reference test pins)."""
    config = VectorConfig.for_paragraphs(64)
    assert config.similarity == Similarity.Dot and config.quantizable_vectors()
    rid = str(uuid.uuid4())
    segment = segment_create([Elem(f"{rid}/a/title/0-{i}", sentence(i, 64)) for i in range(5)], config)
    searcher = VectorSearcher.open(config, [(segment, 1)])
    results = searcher.search(VectorSearchRequest(vector=sentence(0, 64), result_per_page=5, min_score=0.5), PrefilterResult.All)
    assert searcher.last_methods == [_lib.METHOD_RABITQ_BRUTE_FORCE]
    assert len(results.documents) == 1
    assert results.documents[0].score > 0.99
    results = searcher.search(VectorSearchRequest(vector=sentence(0, 64), result_per_page=5, min_score=-1.0), PrefilterResult.All)
    assert len(results.documents) == 5
    searcher.close()


def test_vector_normalization():
    """nidx/tests/integration/vector_normalization.rs:31-91: normalised index, Dot, 20 colinear vectors."""
    config = VectorConfig(dimension=10, similarity=Similarity.Dot, normalize_vectors=True)
    rid = str(uuid.uuid4())
    elems = []
    for i in range(1, 21):
        v = np.full(10, float(i), np.float32)
        out = np.empty_like(v)
        _lib.check(_lib.lib().nidx_gpu_normalize(v.ctypes.data, 1, 10, out.ctypes.data))  # indexer.rs:107-111
        elems.append(Elem(f"{rid}/a/title/0-{i}", out.tolist()))
    searcher = VectorSearcher.open(config, [(segment_create(elems, config), 1)])
    req = VectorSearchRequest(vector=[500.0] * 10, result_per_page=20, min_score=0.999, with_duplicates=True)
    docs = searcher.search(req, PrefilterResult.All).documents
    assert len(docs) == 20 and all(d.score >= 0.999 for d in docs)
    # with_duplicates = false (the proto default): hits are de-duplicated by vector BYTES
    # (Fssc.seen, searcher.rs:175-181) — one hit per distinct bit pattern after normalisation
    req.with_duplicates = False
    distinct = len({np.asarray(e.vector, np.float32).tobytes() for e in elems})
    assert len(searcher.search(req, PrefilterResult.All).documents) == distinct < 20


def test_metadata_and_labels_round_trip():
    """test_basic_search.rs:402-470."""
    config = VectorConfig.for_paragraphs(4)
    rid = str(uuid.uuid4())
    elems = [Elem(f"{rid}/a/title/0-{i}", sentence(i, 4), labels=[f"l{i}"], metadata=bytes([i, i + 1])) for i in range(4)]
    searcher = VectorSearcher.open(config, [(segment_create(elems, config), 1)])
    docs = searcher.search(VectorSearchRequest(vector=sentence(2, 4), result_per_page=1, min_score=-1.0), PrefilterResult.All).documents
    assert docs[0].doc_id.endswith("0-2") and docs[0].labels == ["l2"] and docs[0].metadata == bytes([2, 3])
    assert searcher.space_usage() > 0


def oracle_formula_mask(orc, seg, expr):
    """The formula evaluated by the ORACLE (oracle/nidx_oracle.c: orc_formula_filter — ParagraphInvertedIndexes::filter restated
    document at a time over the paragraph keys and labels; no posting lists, nothing from the product package).  The
    translation below is the test's own: expression tree -> the oracle's postfix program over label / field-id strings."""
    from nucliadb_amd.vector import _KeyPrefixSet

    ops, atoms = [], []

    def emit(e):
        if isinstance(e, Literal):
            ops.append((orc.FORMULA_LABEL, len(atoms), 0))
            atoms.append(e.label)
        elif isinstance(e, _KeyPrefixSet):
            ops.append((orc.FORMULA_KEYSET, len(atoms), len(atoms) + len(e.prefixes)))
            atoms.extend(e.prefixes)
        elif isinstance(e, Not):
            emit(e.operand)
            ops.append((orc.FORMULA_NOT, 0, 0))
        elif isinstance(e, (And, Or)):
            if not e.operands:
                ops.append((orc.FORMULA_ALL if isinstance(e, And) else orc.FORMULA_NONE, 0, 0))
                return
            emit(e.operands[0])
            for o in e.operands[1:]:
                emit(o)
                ops.append((orc.FORMULA_AND if isinstance(e, And) else orc.FORMULA_OR, 0, 0))
        else:
            raise TypeError(e)

    emit(expr)
    # resource-granular prefilter entries: the product follows the intent documented at searcher.rs:300-313 (every field of
    # the resource), DESIGN.md §4.9
    return orc.formula_filter(seg.keys, seg.labels, ops, atoms, resource_prefix=True)


def test_device_filter_programs_match_host_bitsets(orc):
    """Random label / key-prefix formulas: the postfix program evaluated on the GPU (filter.hip) and the
    numpy bitset route must select the same paragraphs, report the same matching count and route alike — and that count
    must be what the ORACLE's document-at-a-time evaluation of the same formula gives."""
    rng = np.random.default_rng(3)
    d, n = 16, 700
    config = VectorConfig.for_paragraphs(d)
    rids = [str(uuid.uuid4()) for _ in range(40)]
    labels_pool = ["/l/a", "/l/a/x", "/l/b", "/l/c", "/e/person/ann", "/e/person", "/t/z"]
    elems = []
    for i in range(n):
        rid = rids[int(rng.integers(0, len(rids)))]
        field = ["a/title", "t/body", "f/file"][int(rng.integers(0, 3))]
        labs = [l for l in labels_pool if rng.random() < 0.25]
        v = rng.normal(size=d).astype(np.float32)
        elems.append(Elem(f"{rid}/{field}/{i}-{i + 5}", v.tolist(), labels=labs))
    segs = [(segment_create(elems[:400], config), 1), (segment_create(elems[400:], config), 2)]
    searcher = VectorSearcher.open(config, segs)
    q = rng.normal(size=(3, d)).astype(np.float32)

    def rand_expr(depth=0):
        r = rng.random()
        if depth > 2 or r < 0.4:
            return Literal(labels_pool[int(rng.integers(0, len(labels_pool)))])
        if r < 0.55:
            return Not(rand_expr(depth + 1))
        ops = [rand_expr(depth + 1) for _ in range(int(rng.integers(1, 4)))]
        return And(ops) if r < 0.8 else Or(ops)

    for trial in range(40):
        formula = rand_expr() if trial % 5 else None
        pre = PrefilterResult.All
        if trial % 3 == 0:
            pre = PrefilterResult.some([FieldId(uuid.UUID(rids[int(rng.integers(0, 40))]), None if rng.random() < 0.5 else "/a/title")
                                        for _ in range(int(rng.integers(1, 6)))])
        op = FilterOperator.Or if trial % 4 == 0 else FilterOperator.And
        req = VectorSearchRequest(result_per_page=10, min_score=-10.0, with_duplicates=True, filtering_formula=formula, filter_operator=op)
        a = searcher.search_batch(req, q, pre, device_filter=True)
        methods_dev, matching_dev = list(searcher.last_methods), list(searcher.last_matching)
        b = searcher.search_batch(req, q, pre, device_filter=False)
        assert methods_dev == searcher.last_methods
        for x, y in zip(a, b):
            assert np.array_equal(x, y), trial
        if formula is not None or pre.kind == "some":
            want = [int(oracle_formula_mask(orc, seg, searcher._formula(req, pre)).sum()) for seg in searcher._segments]
            assert matching_dev == want, (trial, matching_dev, want)
            # and paragraph by paragraph: a search that returns every matching paragraph returns exactly the oracle's set
            big = VectorSearchRequest(result_per_page=200, min_score=-1e30, with_duplicates=True, filtering_formula=formula, filter_operator=op)
            hits = searcher.search(VectorSearchRequest(**{**big.__dict__, "vector": q[0].tolist()}), pre).documents
            if sum(want) <= 200:
                got = sorted(h.doc_id for h in hits)
                exp = sorted(seg.keys[i] for seg in searcher._segments for i in np.nonzero(oracle_formula_mask(orc, seg, searcher._formula(req, pre)))[0])
                assert got == exp, trial
    searcher.close()


def test_single_query_callers_are_coalesced():
    """One query per call from many threads (the reference's request shape): every caller gets exactly
    the rows a batch-of-one search returns, and the calls are served in fewer launches than queries."""
    import ctypes as C
    import threading

    rng = np.random.default_rng(9)
    n, d, k, T = 4000, 64, 10, 48
    x = rng.normal(size=(n, d)).astype(np.float32)
    config = VectorConfig(d, Similarity.Cosine)
    seg = segment_create([Elem(f"{uuid.uuid4()}/a/t/{i}", x[i].tolist()) for i in range(n)], config)
    s = VectorSearcher.open(config, [(seg, 1)])
    s.build_hnsw(0)
    L = _lib.lib()
    _lib.check(L.nidx_gpu_vector_set_tunable(s._handle, b"coalesce_window_us", 20000))
    q = rng.normal(size=(T, d)).astype(np.float32)
    req = VectorSearchRequest(result_per_page=k, min_score=-1.0, with_duplicates=True)
    _, _, want_vec, want_score, want_count = s.search_batch(req, q)
    got = [None] * T
    params = _lib.VectorSearchParamsC(k, -1.0, 1, _lib.METHOD_AUTO)

    def one(i):
        ov, os_, oc = np.zeros(k, np.uint32), np.zeros(k, np.float32), C.c_uint32()
        rc = L.nidx_gpu_vector_search_one(s._handle, q[i].ctypes.data, d, C.byref(params), None, None, ov.ctypes.data,
                                          os_.ctypes.data, C.byref(oc))
        got[i] = (rc, ov, os_, oc.value)

    threads = [threading.Thread(target=one, args=(i,)) for i in range(T)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    for i in range(T):
        rc, ov, os_, oc = got[i]
        assert rc == 0 and oc == want_count[i]
        assert np.array_equal(ov[:oc], want_vec[i, :oc]) and np.array_equal(os_[:oc].view(np.uint32), want_score[i, :oc].view(np.uint32))
    b, nq = C.c_uint64(), C.c_uint64()
    _lib.check(L.nidx_gpu_vector_coalescer_stats(s._handle, C.byref(b), C.byref(nq)))
    assert nq.value == T and b.value < T // 2, (b.value, nq.value)
    # a wrong dimension is an error for that caller only
    oc = C.c_uint32()
    assert L.nidx_gpu_vector_search_one(s._handle, q[0].ctypes.data, d - 1, C.byref(params), None, None, None, None, C.byref(oc)) == _lib.NIDX_ERR_INCONSISTENT_DIMENSIONS
    s.close()


def test_hidden_search():
    """test_hidden.rs:25-78: segment tags (the resource labels the indexer extracts, indexer.rs:27) and
    segment_filtering_formula: Not("/q/h") skips the segment of the hidden resource."""
    config = VectorConfig.for_paragraphs(4)
    hid, hidden = _resource(4)
    vid, visible = _resource(4)
    hidden_segment = segment_create(hidden, config, tags={"/q/h"})
    visible_segment = segment_create(visible, config)
    searcher = VectorSearcher.open(config, [(hidden_segment, 1), (visible_segment, 2)])
    request = VectorSearchRequest(vector=[0.5, 0.5, 0.5, 0.5], min_score=-1.0, result_per_page=10)
    everything = searcher.search(request, PrefilterResult.All)
    assert {d.doc_id for d in everything.documents} == {f"{hid}/a/title/0-5", f"{vid}/a/title/0-5"}
    request.segment_filtering_formula = Not(Literal("/q/h"))
    seen = searcher.search(request, PrefilterResult.All)
    assert [d.doc_id for d in seen.documents] == [f"{vid}/a/title/0-5"]
    searcher.close()


def test_paragraph_merge_with_deletions():
    """tests/test_paragraph_merge.rs:69-173: two single-resource segments (seq 2 and 4) merged under deletions that share
    their seq; the merged segment holds all four vectors and every one of them is its own top hit."""
    from test_vector_indexer_cpu import UUID1, UUID2, make_vector, two_segments
    from nucliadb_amd.vector import VectorIndexer

    config, s1, s2 = two_segments()
    deletions = [(f"{UUID1}/a/title", 2), (f"{UUID1}/t/body", 2), (f"{UUID2}/a/title", 4), (f"{UUID2}/t/body", 4)]
    merged = VectorIndexer().merge(config, [(s1, 2), (s2, 4)], deletions)
    assert merged.records == 4
    searcher = VectorSearcher.open(config, [(merged, 5)])
    owner = {0: f"{UUID1}/a/title/0-10", 1: f"{UUID1}/t/body/0-10", 2: f"{UUID2}/a/title/0-10", 3: f"{UUID2}/t/body/0-10"}
    for axis in (0, 2, 3):
        results = searcher.search(VectorSearchRequest(vector=make_vector(axis), result_per_page=10, with_duplicates=True),
                                  PrefilterResult.All)
        assert len(results.documents) == 4
        assert results.documents[0].score > 0.9999 and results.documents[0].doc_id == owner[axis]
    searcher.close()
    # the same index searched unmerged, with a deletion that IS newer than segment 1
    searcher = VectorSearcher.open(config, [(s1, 2), (s2, 4)], deletions + [(f"{UUID1}/a/title", 3)])
    results = searcher.search(VectorSearchRequest(vector=make_vector(0), result_per_page=10, with_duplicates=True), PrefilterResult.All)
    assert len(results.documents) == 3 and owner[0] not in {d.doc_id for d in results.documents}
    searcher.close()
