"""Pins the CPU oracle against every golden the reference's own tests hold for the hot path
(SURVEY.md §8c).  Each test cites the reference test it restates (paths relative to
/root/reference/nidx).  CPU only."""
import math

import numpy as np
import pytest

ORDERS = [0, 1, 2, 3]


def one_hot(dim, i):
    v = np.zeros(dim, np.float32)
    v[i] = 1.0
    return v


# --- nidx_vector/src/vector_types/dense_f32.rs:66-84 ---------------------------------------
@pytest.mark.parametrize("order", ORDERS)
def test_cosine_vs_naive(orc, order):
    v0 = np.arange(758, dtype=np.float32) * 2.0
    v1 = np.arange(758, dtype=np.float32) * 1.0 + 1.0

    def naive(a, b):
        ab = np.float32(0); aa = np.float32(0); bb = np.float32(0)
        for x, y in zip(a, b):
            ab += x * y; aa += x * x; bb += y * y
        return ab / (np.sqrt(aa) * np.sqrt(bb))

    assert abs(naive(v0, v0) - orc.cosine(v0, v0, order)) < 0.01
    assert abs(naive(v0, v1) - orc.cosine(v0, v1, order)) < 0.01


@pytest.mark.parametrize("order", ORDERS)
def test_dot_vs_naive(orc, order):
    v0 = np.arange(758, dtype=np.float32) * np.float32(0.002)
    v1 = np.arange(758, dtype=np.float32) * np.float32(0.002) + np.float32(0.05)

    def naive(a, b):
        s = np.float32(0)
        for x, y in zip(a, b):
            s += x * y
        return s

    assert abs(naive(v0, v0) - orc.dot(v0, v0, order)) < 0.01
    assert abs(naive(v0, v1) - orc.dot(v0, v1, order)) < 0.01


def test_orders_agree_within_tolerance(orc):
    """north_star: cosine within 1e-5. All four summation orders must sit inside that band."""
    rng = np.random.default_rng(7)
    for dim in (3, 64, 255, 256, 768, 1024):
        x = rng.uniform(-1, 1, dim).astype(np.float32)
        y = rng.uniform(-1, 1, dim).astype(np.float32)
        x /= np.linalg.norm(x); y /= np.linalg.norm(y)
        ref = float(np.dot(x.astype(np.float64), y.astype(np.float64)))
        refc = ref / (np.linalg.norm(x.astype(np.float64)) * np.linalg.norm(y.astype(np.float64)))
        for o in ORDERS:
            assert abs(orc.dot(x, y, o) - ref) < 1e-5
            assert abs(orc.cosine(x, y, o) - refc) < 1e-5


def test_simsimd_cosine_edge_cases(orc):
    z = np.zeros(8, np.float32)
    a = one_hot(8, 1)
    b = one_hot(8, 2)
    for o in ORDERS:
        assert orc.cosine(z, z, o) == 1.0  # a2 == b2 == 0 -> distance 0
        assert orc.cosine(a, b, o) == 0.0  # ab == 0 -> distance 1
        assert orc.cosine(z, a, o) == 0.0
        assert orc.cosine(a, a, o) == 1.0
        assert orc.cosine(a, -a, o) == -1.0  # distance 2 is not clamped from above


# --- nidx_vector/src/utils.rs:140-155 --------------------------------------------------------
def test_vector_normalization(orc):
    assert orc.normalize(np.zeros(0, np.float32)).size == 0
    np.testing.assert_array_equal(orc.normalize([3.0, 0.0, 4.0, 0.0]), np.array([3.0 / 5.0, 0.0, 4.0 / 5.0, 0.0], np.float32))
    np.testing.assert_array_equal(orc.normalize([-1.0, -1.0, 0.0, 1.0, 1.0]), np.array([-0.5, -0.5, 0.0, 0.5, 0.5], np.float32))
    big = orc.normalize(np.full(10000, 100.0, np.float32))
    assert big[0] == np.float32(0.01)
    np.testing.assert_array_equal(big, np.full(10000, 0.01, np.float32))


# --- nidx_vector/tests/test_basic_search.rs:38-145 -------------------------------------------
@pytest.mark.parametrize("sim", [0, 1])
@pytest.mark.parametrize("order", ORDERS)
def test_basic_search(orc, sim, order):
    dim = 64
    vecs = np.stack([one_hot(dim, i) for i in range(dim)])
    seg = orc.Segment(vecs, similarity=sim, order=order)
    seg.build_graph()
    ids, scores, method = seg.search(one_hot(dim, 5), 10, min_score=-1.0, with_duplicates=False)
    assert method == "brute force"  # 64 records: the cost model picks brute force
    assert len(ids) == 10
    assert ids[0] == 5
    assert scores[0] > 0.9999
    assert scores[1] < 0.0001

    q = np.zeros(dim, np.float32)
    q[42], q[43], q[44], q[45] = 0.7, 0.59, 0.35, 0.2
    ids, scores, _ = seg.search(q, 10, min_score=-1.0, with_duplicates=False)
    assert len(ids) == 10
    assert list(ids[:4]) == [42, 43, 44, 45]
    assert scores[0] > 0.6 and scores[1] > 0.5 and scores[2] > 0.3 and scores[3] > 0.15
    assert scores[5] == 0.0
    # the same through the multi-segment Searcher + Fssc (with_duplicates defaults to false)
    res = orc.searcher_search([seg], [np.arange(dim, dtype=np.uint64)], q, 10, min_score=-1.0, with_duplicates=False)
    assert [r[0] for r in res[:4]] == [42, 43, 44, 45]


# --- nidx_vector/tests/test_min_score.rs:61-164 (exact path; RaBitQ is out of the first slice) --
def test_min_score(orc):
    dim, n = 64, 5
    vecs = np.stack([one_hot(dim, i) for i in range(n)])
    seg = orc.Segment(vecs, similarity=0)
    seg.build_graph()
    ids, scores, method = seg.search(one_hot(dim, 0), n, min_score=0.5)
    assert method == "brute force"
    assert len(ids) == 1 and scores[0] > 0.99 and scores[0] >= 0.5
    ids, scores, _ = seg.search(one_hot(dim, 0), n, min_score=-1.0)
    assert len(ids) == n


# --- nidx_vector/src/segment.rs:626-660 cost model ---------------------------------------------
def test_use_hnsw_cost_model(orc):
    assert not orc.use_hnsw(64, 64, 10)
    assert not orc.use_hnsw(5, 5, 5)
    assert orc.use_hnsw(640, 640, 5)
    assert orc.use_hnsw(100_000, 100_000, 10)
    # a selective filter flips it back to brute force: k*M*N/matching
    assert not orc.use_hnsw(100_000, 100, 10)

    def model(n, m, k):
        l = np.float32(math.log(np.float32(n))) - np.float32(2.0)
        rq = np.float32(l * l) * np.float32(math.log(np.float32(k)))
        return int(rq) + (k * 30 * n // m) < m

    for n, m, k in [(200, 200, 10), (330, 330, 10), (340, 340, 10), (1000, 500, 10), (5000, 300, 20)]:
        assert orc.use_hnsw(n, m, k) == model(n, m, k)


# --- nidx_vector/src/segment.rs:841-912 recall floor ------------------------------------------
def _random_vector(rng, dim):
    v = rng.uniform(-1.0, 1.0, dim).astype(np.float32)
    return (v / np.float32(np.sqrt(np.sum(v * v, dtype=np.float32)))).astype(np.float32)


def _nearby(rng, close_to, distance):
    fuzz = _random_vector(rng, close_to.size)
    v = close_to + fuzz * np.float32(distance)
    return (v / np.float32(np.sqrt(np.sum(v * v, dtype=np.float32)))).astype(np.float32)


def clustered(rng, dim=256, clusters=4):
    elems = []
    center = _random_vector(rng, dim)
    for _ in range(clusters):
        for _ in range(80):
            elems.append(_nearby(rng, center, 0.01))
        for _ in range(80):
            elems.append(_nearby(rng, center, 0.03))
        center = _nearby(rng, center, 0.1)
    return np.stack(elems)


def test_recall_clustered_data(orc):
    rng = np.random.default_rng(1234567890)
    elems = clustered(rng)
    # the reference stores them in BTreeMap (random key) order
    elems = elems[rng.permutation(len(elems))]
    seg = orc.Segment(elems, similarity=0, order=orc.ORDER_HASWELL)
    seg.build_graph(seed=2)
    correct = 0.0
    for _ in range(100):
        base = elems[rng.integers(0, len(elems))]
        q = _nearby(rng, base, 0.05)
        sims = elems @ q
        brute = np.argsort(-sims, kind="stable")[:5]
        ids, _, method = seg.search(q, 5, min_score=0.0, with_duplicates=False)
        assert method == "hnsw"
        correct += 0.2 * len(set(brute.tolist()) & set(ids.tolist()))
    recall = correct / 100.0
    assert recall >= 0.95, recall


# --- nidx_vector/src/segment/tests.rs:379-477: exact self match after build ---------------------
def test_self_match_scores(orc):
    rng = np.random.default_rng(5)
    vecs = np.stack([_random_vector(rng, 128) for _ in range(800)])
    seg = orc.Segment(vecs, similarity=0)
    seg.build_graph()
    for i in (0, 17, 799):
        ids, scores, method = seg.search(vecs[i], 1, min_score=-1.0)
        assert method == "hnsw"
        assert ids[0] == i and scores[0] >= 0.999


# --- nidx/tests/integration/vector_normalization.rs:31-91 ---------------------------------------
def test_vector_normalization_index(orc):
    dim = 10
    stored = np.stack([orc.normalize(np.full(dim, float(i + 1), np.float32)) for i in range(20)])
    seg = orc.Segment(stored, similarity=0)
    magnitude = np.float32(math.sqrt(17.0**2 * dim))
    q = np.full(dim, np.float32(17.0) / magnitude, np.float32)
    res = orc.searcher_search([seg], [np.arange(20, dtype=np.uint64)], q, 30, min_score=0.9, with_duplicates=True,
                              normalize_query=True)
    assert len(res) == 20
    assert all(r[1] >= 0.999 for r in res)


# --- nidx_vector/src/searcher.rs:411+ (3-d Dot, explicit vectors; the duplicate 3rd/4th vectors) --
def test_explicit_vectors_and_duplicates(orc):
    vecs = np.array([[1, 3, 4], [2, 4, 5], [3, 5, 6], [3, 5, 6]], np.float32)
    seg = orc.Segment(vecs, similarity=0)
    keys = [np.arange(4, dtype=np.uint64)]
    q = np.array([4, 6, 7], np.float32)
    res = orc.searcher_search([seg], keys, q, 20, with_duplicates=True)
    assert len(res) == 4
    assert [r[1] for r in res] == [84.0, 84.0, 67.0, 50.0]
    # with_duplicates = false suppresses byte-identical vectors (hnsw/search.rs:155-168, searcher.rs:175-181)
    res = orc.searcher_search([seg], keys, q, 20, with_duplicates=False)
    assert len(res) == 3  # Fssc.seen drops the byte-identical 4th vector even inside one segment
    seg2 = orc.Segment(vecs[2:3], similarity=0)
    res = orc.searcher_search([seg, seg2], [keys[0], np.array([99], np.uint64)], q, 20, with_duplicates=False)
    assert 99 not in [r[0] for r in res]


# --- Fssc quirks (searcher.rs:183-197) ---------------------------------
def test_fssc_same_paragraph_collapses(orc):
    a = np.array([[1, 0, 0], [0, 1, 0]], np.float32)
    b = np.array([[0.9, 0.1, 0], [0, 0, 1]], np.float32)
    s0, s1 = orc.Segment(a, similarity=0), orc.Segment(b, similarity=0)
    # paragraph key 7 appears in both segments -> second insert is a no-op
    res = orc.searcher_search([s0, s1], [np.array([7, 8], np.uint64), np.array([7, 9], np.uint64)],
                              np.array([1, 0, 0], np.float32), 10, with_duplicates=True)
    keys = [r[0] for r in res]
    assert keys.count(7) == 1 and res[0][0] == 7 and res[0][1] == 1.0


# --- nidx_vector/src/hnsw/disk/v2.rs:16-49 (documented example) and :339-473 --------------------
def test_disk_hnsw_v2_documented_layout(orc):
    g = orc.Hnsw.new()
    g.add_node(0, 2)
    g.set_edges(0, 0, [1, 17, 5433, 45, 667])
    g.set_edges(1, 0, [45, 666, 22])
    g.set_entry_point(0, 2)
    graph, edges = g.serialize_v2(1)
    words = np.frombuffer(graph.tobytes(), dtype="<u4").tolist()
    # the module comment's example prints "16 32 52" for the offsets; serialize_node (v2.rs:109-140)
    # computes node_len - layer_start = 56 for layer 0 (14 words), which is what the reader needs.
    assert words == [5, 1, 17, 5433, 45, 667, 3, 45, 666, 22, 0, 16, 32, 56, 56, 2, 0]
    assert edges.size == 8


def _three_layer_graph(orc):
    g = orc.Hnsw.new()
    cnx0 = [[(1, 1.0)], [(2, 2.0)], [(3, 3.0)]]
    cnx1 = [[(1, 4.0)], [(2, 5.0)]]
    cnx2 = [[(1, 6.0)]]
    for i in range(3):
        g.add_node(i, 0)
    for i in range(2):
        g.add_node(i, 1)
    g.add_node(0, 2)
    for layer, cnx in enumerate([cnx0, cnx1, cnx2]):
        for node, edges in enumerate(cnx):
            g.set_edges(layer, node, [e[0] for e in edges], [e[1] for e in edges])
    g.set_entry_point(0, 2)
    return g, [cnx0, cnx1, cnx2]


def test_disk_hnsw_v2_roundtrip(orc):
    empty = orc.Hnsw.new()
    graph, _ = empty.serialize_v2(0)
    assert graph.size == 0

    g, cnx = _three_layer_graph(orc)
    graph, edges = g.serialize_v2(3)
    assert orc.disk_v2_entry_point(graph) == (0, 2)
    for layer, layer_cnx in enumerate(cnx):
        for node in range(3):
            expected = [e[0] for e in layer_cnx[node]] if node < len(layer_cnx) else []
            assert orc.disk_v2_edges(graph, layer, node).tolist() == expected
    # deserialize -> serialize is byte exact (hnsw_deserialize_test)
    ram = orc.Hnsw.deserialize_v2(graph, edges)
    graph2, edges2 = ram.serialize_v2(3)
    assert graph.tobytes() == graph2.tobytes()
    assert edges.tobytes() == edges2.tobytes()
    assert edges.tolist() == [1.0, 4.0, 6.0, 2.0, 5.0, 3.0]

    one = orc.Hnsw.new()
    one.add_node(0, 0)
    one.update_entry_point()
    graph, edges = one.serialize_v2(1)
    ram = orc.Hnsw.deserialize_v2(graph, edges)
    ram.fix_broken_graph()
    assert ram.num_layers == 1 and ram.num_nodes == 1 and ram.edges(0, 0)[0].size == 0


# --- nidx_vector/src/hnsw/ram_hnsw.rs:177-198 ---------------------------------------------------
def test_fix_broken_links(orc):
    g = orc.Hnsw.new()
    g.add_node(0, 1)
    g.add_node(1, 0)
    g.set_edges(0, 0, [1], [0.5])
    g.set_edges(0, 1, [0], [0.5])
    g.set_edges(1, 0, [1], [0.5])  # node 1 is not in layer 1: broken link
    g.fix_broken_graph()
    assert g.edges(1, 0)[0].size == 0
    assert g.edges(0, 0)[0].tolist() == [1]


# --- level draw (hnsw/build.rs:97-101, params.rs:20-22) -----------------------------------------
def test_level_distribution(orc):
    lv = orc.hnsw_levels(2, 200_000)
    # P(level >= 1) = P(-ln(u)/ln 30 >= 0.5) = 30^-0.5 ; P(level >= 2) = 30^-1.5   (round, not floor)
    assert abs((lv >= 1).mean() - 30 ** -0.5) < 0.01
    assert abs((lv >= 2).mean() - 30 ** -1.5) < 0.002
    assert np.array_equal(lv, orc.hnsw_levels(2, 200_000))


# --- nidx/src/searcher/shard_merge.rs:1200-1243 -------------------------------------------------
def test_merge_vector_results(orc):
    a, b, c, d = 1, 2, 3, 4
    shards = [[(1.0, a), (0.5, b)], [(3.2, c), (0.9, d)]]
    assert [i for _, i in orc.merge_vector(shards, 20)] == [c, a, d, b]
    assert [i for _, i in orc.merge_vector(shards, 2)] == [c, a]
    assert orc.merge_vector([], 20) == []
    assert orc.merge_vector([[], []], 20) == []


# --- nidx/src/searcher/shard_merge.rs:613-761 (documents) and :985-1135 (paragraphs) -----------
def test_merge_bm25_by_score_and_tiebreaks(orc):
    foo, bar, baz, quux = 1, 2, 3, 4
    merged = orc.merge_bm25([[(3.0, 2, b"", foo), (2.0, 1, b"", bar)], [(4.0, 2, b"", baz), (2.0, 2, b"", quux)]], 20)
    assert [m[3] for m in merged] == [baz, foo, bar, quux]

    A = bytes.fromhex("aa" * 16)
    B = bytes.fromhex("bb" * 16)
    merged = orc.merge_bm25([[(2.0, 1, B, foo)], [(2.0, 1, A, bar)]], 20)
    assert [m[3] for m in merged] == [foo, bar] and merged[0][2] == B
    merged = orc.merge_bm25([[(2.0, 1, A, foo)], [(2.0, 1, B, bar)]], 20)
    assert [m[3] for m in merged] == [bar, foo] and merged[0][2] == B
    merged = orc.merge_bm25([[(2.0, 2, A, foo)], [(2.0, 1, A, bar)]], 20)
    assert [m[3] for m in merged] == [bar, foo]
    # with_limit: 2 shards x 20 equal hits
    shard = [(0.0, 0, b"", i) for i in range(20)]
    assert len(orc.merge_bm25([shard, shard], 50)) == 40
    assert len(orc.merge_bm25([shard, shard], 20)) == 20


# --- tantivy fieldnorm table + BM25 formula ------------------------------------------------------
def test_fieldnorm_table(orc):
    t = orc.fieldnorm_table()
    assert t[:41].tolist() == list(range(41))
    assert t[41:49].tolist() == [42, 44, 46, 48, 50, 52, 54, 56]
    assert t[49:57].tolist() == [60, 64, 68, 72, 76, 80, 84, 88]
    assert t[57] == 96 and t[65] == 168
    assert t[255] == 2_013_265_944
    assert np.all(np.diff(t.astype(np.int64)) > 0)
    for n in (0, 1, 39, 40, 41, 42, 43, 57, 1000, 2_013_265_944, 4_000_000_000):
        i = orc.fieldnorm_to_id(n)
        assert t[i] <= n and (i == 255 or t[i + 1] > n)


def test_bm25_formula(orc):
    # idf = ln(1 + (N - n + 0.5)/(n + 0.5)) in f32
    assert orc.bm25_idf(1, 1) == float(np.float32(math.log(np.float32(1.0) + np.float32(0.5) / np.float32(1.5))))
    cache = orc.bm25_tf_cache(10.0)
    assert cache[10] == np.float32(1.2) * (np.float32(1.0) - np.float32(0.75) + np.float32(0.75) * np.float32(10.0) / np.float32(10.0))
    # single doc, single term, tf=1, fieldnorm 1 -> weight * 1/(1 + K1)
    idx = orc.Bm25Index([0, 1], [0], [1], [1], 1)
    docs, scores, total = idx.search([(0, orc.OCCUR_SHOULD, orc.TF_FREQ, 1.0)], 10)
    w = np.float32(orc.bm25_idf(1, 1)) * np.float32(2.2)
    assert total == 1 and docs.tolist() == [0]
    assert scores[0] == w * (np.float32(1.0) / (np.float32(1.0) + np.float32(1.2)))


def test_bm25_counts_min_score_and_order(orc):
    """nidx_text/tests/test_search.rs:311-332 and nidx_paragraph/tests/reader.rs:315-342 pin only
    counts: a matching query returns hits at min_score 0 and none at a high min_score while
    `total` stays.  Tie order = (score desc, docaddr asc)."""
    # 4 docs containing term 0 with equal tf / length -> 4 equal scores
    idx = orc.Bm25Index([0, 4, 5], [0, 1, 2, 3, 2], [1, 1, 1, 1, 3], [8, 8, 8, 8], 32)
    docs, scores, total = idx.search([(0, orc.OCCUR_SHOULD, orc.TF_FREQ, 1.0)], 10)
    assert total == 4 and docs.tolist() == [0, 1, 2, 3]
    assert len(set(scores.tolist())) == 1 and scores[0] < 30.0
    kept = [s for s in scores if s >= 30.0]
    assert kept == [] and total == 4
    # search_after (nidx_paragraph/src/reader.rs:379-390): page one hit at a time
    seq = []
    after = None
    for _ in range(4):
        d, s, _ = idx.search([(0, orc.OCCUR_SHOULD, orc.TF_FREQ, 1.0)], 1, after=after)
        seq.append(int(d[0]))
        after = (float(s[0]), 1, int(d[0]))
    assert seq == [0, 1, 2, 3]
    # two SHOULD terms: doc 2 has both -> ranks first
    docs, scores, total = idx.search([(0, 0, 0, 1.0), (1, 0, 0, 1.0)], 10)
    assert docs[0] == 2 and total == 4
    # MUST + MUST_NOT
    docs, _, total = idx.search([(0, orc.OCCUR_MUST, 0, 1.0), (1, orc.OCCUR_MUST_NOT, 0, 1.0)], 10)
    assert docs.tolist() == [0, 1, 3] and total == 3


def test_bm25_daat_equals_term_at_a_time(orc):
    """The document-at-a-time form used as bench.py's CPU baseline gives bit-identical hits, order and totals."""
    rng = np.random.default_rng(31)
    vocab, n_docs = 120, 3000
    lens = rng.integers(2, 25, n_docs)
    p = 1.0 / np.arange(1, vocab + 1)
    p /= p.sum()
    flat = rng.choice(vocab, size=int(lens.sum()), p=p)
    doc_of = np.repeat(np.arange(n_docs), lens)
    uniq, counts = np.unique(flat.astype(np.int64) * (n_docs + 1) + doc_of, return_counts=True)
    t, d = uniq // (n_docs + 1), uniq % (n_docs + 1)
    off = np.zeros(vocab + 1, np.uint64)
    np.add.at(off, t + 1, 1)
    off = np.cumsum(off).astype(np.uint64)
    ids = np.array([orc.fieldnorm_to_id(int(x)) for x in lens], np.uint8)
    alive = orc.bitset(n_docs, ones=np.nonzero(rng.random(n_docs) < 0.8)[0].tolist())
    idx = orc.Bm25Index(off, d.astype(np.uint32), counts.astype(np.uint32), ids, int(lens.sum()), alive)
    for _ in range(60):
        q = [(int(rng.integers(0, 50)), int(rng.choice([0, 0, 1, 2, 3])), int(rng.choice([0, 1, 2])), float(rng.choice([1.0, 0.5])))
             for _ in range(int(rng.integers(1, 6)))]
        after = None if rng.random() < 0.7 else (float(rng.random() * 3), int(rng.integers(0, 3)), int(rng.integers(0, n_docs)))
        a = idx.search(q, 15, after=after, segment_ord=2)
        b = idx.search(q, 15, after=after, segment_ord=2, daat=True)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)) and a[2] == b[2]


# ---- FuzzyTermQuery's automaton (levenshtein_automata 0.2.1, restated) ----------------------------------------------
def _osa(a, b):
    d = [[0] * (len(b) + 1) for _ in range(len(a) + 1)]
    for i in range(len(a) + 1):
        d[i][0] = i
    for j in range(len(b) + 1):
        d[0][j] = j
    for i in range(1, len(a) + 1):
        for j in range(1, len(b) + 1):
            d[i][j] = min(d[i - 1][j] + 1, d[i][j - 1] + 1, d[i - 1][j - 1] + (a[i - 1] != b[j - 1]))
            if i > 1 and j > 1 and a[i - 1] == b[j - 2] and a[i - 2] == b[j - 1]:
                d[i][j] = min(d[i][j], d[i - 2][j - 2] + 1)
    return d


def test_fuzzy_automaton_reference_cases(orc):
    """nidx_paragraph/tests/reader.rs:262-300: a typo of distance 1 (a transposition counts one) still matches, distance 2
    does not; fuzzy_parser.rs:35-42: distance 1, prefix DFA for the last literal."""
    assert orc.fuzzy_match("shoupd", "should") and orc.fuzzy_match("enaugh", "enough")
    assert not orc.fuzzy_match("sJoupd", "should") and not orc.fuzzy_match("enaugJ", "enough")
    assert orc.fuzzy_match("enoguh", "enough") and not orc.fuzzy_match("eonguh", "enough")
    assert orc.fuzzy_match("shoul", "shoulder", prefix=True) and not orc.fuzzy_match("shoul", "shoulder")
    import random

    rnd = random.Random(5)
    for _ in range(3000):
        a = "".join(rnd.choice("abcñ道") for _ in range(rnd.randint(1, 7)))
        b = "".join(rnd.choice("abcñ道") for _ in range(rnd.randint(0, 9)))
        d = _osa(a, b)
        assert orc.fuzzy_match(a, b) == (d[len(a)][len(b)] <= 1), (a, b)
        assert orc.fuzzy_match(a, b, prefix=True) == (min(d[len(a)]) <= 1), (a, b)


def test_prefilter_reference_cases(orc):
    """nidx_text/tests/test_search.rs:75-128,355-443 on the reference's test resource (tests/common/mod.rs:60-111: two fields,
    labels /l/mylabel + /e/myentity on the title, /f/body + /l/mylabel2 on the body, created = modified = now): the
    counts those tests assert, and a pure-Python evaluation of random expressions."""
    T, AND, OR, NOT, ALL, NONE, RANGE, PHRASE = range(8)
    # terms: 0 /l/mylabel 1 /e/myentity 2 /f/body 3 /l/mylabel2 4 /l 5 uuid 6 first 7 document 8 tantivy 9 this
    docs = [np.array([9, 6, 7, 0, 1, 4, 5]), np.array([9, 8, 2, 3, 4, 5])]
    now = 1_700_000_000
    created = modified = np.array([now, now], np.int64)

    def build(docs, alive=None):
        n_terms = 10
        post = [[] for _ in range(n_terms)]
        pos = [[] for _ in range(n_terms)]
        for d, toks in enumerate(docs):
            for t in sorted(set(toks.tolist())):
                post[t].append(d)
                pos[t].append([i for i, x in enumerate(toks) if x == t])
        offs = np.cumsum([0] + [len(p) for p in post]).astype(np.uint64)
        ids = np.array([d for p in post for d in p], np.uint32)
        tfs = np.array([len(q) for p in pos for q in p], np.uint32)
        po = np.cumsum([0] + [len(q) for p in pos for q in p]).astype(np.uint64)
        pp = np.array([x for p in pos for q in p for x in q], np.uint32)
        return orc.Bm25Index(offs, ids, tfs, np.zeros(len(docs), np.uint8), sum(len(d) for d in docs), alive, po, pp)

    idx = build(docs)
    run = lambda ops, lists=(), ranges=(), phrases=(): idx.prefilter(ops, lists, ranges, created, modified, phrases)
    assert run([])[0].tolist() == [0, 1] and run([])[1] == 2
    assert run([(T, 0, 1), (NOT, 0, 0)], [0])[0].tolist() == [1]                      # test_prefilter_not_search: 1 field
    assert run([(T, 0, 1)], [0])[0].tolist() == [0]                                   # test_labels_prefilter_search: 1 field
    assert run([(RANGE, 0, 0)], ranges=[(0, now - 100, now + 100)])[0].size == 2      # test_timestamp_filtering
    assert run([(RANGE, 0, 0)], ranges=[(1, now + 100, None)])[0].size == 0
    assert run([(RANGE, 0, 0)], ranges=[(1, now, now)])[0].size == 2                  # inclusive on both sides
    assert run([(RANGE, 0, 0)], ranges=[(0, None, None)])[0].size == 2                # no bound = AllQuery
    assert run([(T, 0, 1)], [5])[0].size == 2                                         # test_key_filtering: 2 fields
    assert run([(PHRASE, 0, 0)], phrases=[[6, 7]])[0].tolist() == [0]                 # "first document"
    assert run([(PHRASE, 0, 0)], phrases=[[7, 6]])[0].size == 0
    assert run([(T, 0, 2), (T, 2, 3), (AND, 0, 0)], [0, 3, 8])[0].tolist() == [1]     # (mylabel | mylabel2) & tantivy
    # deleted documents never match and do not count as live
    dead = build(docs, np.array([0b10], np.uint64))
    got, live = dead.prefilter([(ALL, 0, 0)], (), (), created, modified, ())
    assert got.tolist() == [1] and live == 1
    # random expressions vs a set-based evaluation
    rng = np.random.default_rng(3)
    rdocs = [rng.integers(0, 10, int(rng.integers(1, 12))) for _ in range(300)]
    ridx = build(rdocs)
    cr, mo = rng.integers(0, 50, 300), rng.integers(0, 50, 300)
    universe = set(range(300))
    for _ in range(200):
        ops, lists, ranges, phrases, stack = [], [], [], [], []
        for _ in range(int(rng.integers(1, 8))):
            r = rng.random()
            if r < 0.4 or len(stack) == 0:
                kind = rng.random()
                if kind < 0.5:
                    ts = rng.integers(0, 10, int(rng.integers(0, 3))).tolist()
                    ops.append((T, len(lists), len(lists) + len(ts)))
                    lists += ts
                    stack.append({d for d in universe if any(t in rdocs[d] for t in ts)})
                elif kind < 0.8:
                    lo, hi = (int(x) if rng.random() < 0.7 else None for x in rng.integers(0, 50, 2))
                    f = int(rng.integers(0, 2))
                    ops.append((RANGE, len(ranges), 0))
                    ranges.append((f, lo, hi))
                    v = cr if f == 0 else mo
                    stack.append({d for d in universe if (lo is None or v[d] >= lo) and (hi is None or v[d] <= hi)})
                else:
                    ph = rng.integers(0, 10, 2).tolist()
                    ops.append((PHRASE, len(phrases), 0))
                    phrases.append(ph)
                    stack.append({d for d in universe if any(rdocs[d][i] == ph[0] and rdocs[d][i + 1] == ph[1] for i in range(len(rdocs[d]) - 1))})
            elif r < 0.55:
                ops.append((NOT, 0, 0))
                stack.append(universe - stack.pop())
            elif len(stack) >= 2:
                b, a = stack.pop(), stack.pop()
                if rng.random() < 0.5:
                    ops.append((AND, 0, 0)); stack.append(a & b)
                else:
                    ops.append((OR, 0, 0)); stack.append(a | b)
        while len(stack) > 1:
            b, a = stack.pop(), stack.pop()
            ops.append((AND, 0, 0)); stack.append(a & b)
        got, live = ridx.prefilter(ops, lists, ranges, cr, mo, phrases)
        assert live == 300 and got.tolist() == sorted(stack[0]), ops


def test_nested_query_evaluator_equals_the_c_oracle_on_flat_queries():
    """oracle.bm25_nested_search (numpy, used to check nested BooleanQuerys on the device) evaluates flat queries exactly like
    orc_bm25_search: same doc addresses, same f32 score bits, same Count — Must / MustNot / Should / required groups, the three
    score modes, boosts."""
    import numpy as np

    from oracle import oracle as orc

    orc.build()
    rng = np.random.default_rng(5)
    n_docs, vocab = 4000, 300
    lens = np.clip(np.round(rng.lognormal(np.log(20), 0.6, n_docs)), 2, 400).astype(np.int64)
    p = 1.0 / np.arange(1, vocab + 1)
    p /= p.sum()
    flat = rng.choice(vocab, size=int(lens.sum()), p=p)
    doc_of = np.repeat(np.arange(n_docs), lens)
    key = flat.astype(np.int64) * (n_docs + 1) + doc_of
    uniq, counts = np.unique(key, return_counts=True)
    t, d = uniq // (n_docs + 1), uniq % (n_docs + 1)
    term_offsets = np.zeros(vocab + 1, np.uint64)
    np.add.at(term_offsets, t + 1, 1)
    term_offsets = np.cumsum(term_offsets).astype(np.uint64)
    table = orc.fieldnorm_table().astype(np.int64)
    fn_ids = (np.searchsorted(table, lens, side="right") - 1).astype(np.uint8)
    idx = orc.Bm25Index(term_offsets, d.astype(np.uint32), counts.astype(np.uint32), fn_ids, int(lens.sum()))
    for _ in range(60):
        q = [(int(rng.integers(0, 120)), int(rng.choice([0, 0, 1, 2, 3, 4])), int(rng.choice([0, 1, 2])), float(rng.choice([1.0, 0.5, 2.0])))
             for _ in range(int(rng.integers(1, 6)))]
        wd, ws, wt = idx.search(q, 25)
        gd, gs, gt = orc.bm25_nested_search(idx, q, 25)
        assert gt == wt and np.array_equal(gd, wd) and np.array_equal(gs.view(np.uint32), ws.view(np.uint32)), q


def test_nested_evaluator_sets_and_phrases_equal_the_c_oracle_and_slop_contract():
    """The leaf kinds round 4 added to oracle.bm25_nested_search — ("set", ..) ConstScorer unions / complements and ("phrase", ..)
    PhraseQuerys — evaluate at the top level exactly like the C oracle's term-set and phrase clauses (slop 0); the slop matcher is
    then checked against PhraseQuery::set_slop's documented contract (a budget of moves shared by the terms, both directions:
    "A B C"~1 matches "A X B C" and "A B X C", not "A X B X C"; "A B"~1 does not match "B A", ~2 does)."""
    import numpy as np

    from nucliadb_amd.bm25 import Bm25Segment
    from oracle import oracle as orc

    orc.build()
    rng = np.random.default_rng(15)
    vocab = 40
    docs = [rng.integers(0, vocab, int(rng.integers(3, 30))) for _ in range(1500)]
    seg = Bm25Segment.from_term_docs(docs, vocab, with_positions=True)
    idx = orc.Bm25Index(seg.term_offsets, seg.doc_ids, seg.tfs, seg.fieldnorm_ids, seg.total_num_tokens, None, seg.pos_offsets, seg.positions)
    for _ in range(40):
        flat, nested = [], []
        for _ in range(int(rng.integers(1, 4))):
            kind = int(rng.integers(0, 3))
            occur, boost = int(rng.choice([0, 0, 1, 2, 3])), float(rng.choice([1.0, 0.5, 2.0]))
            if kind == 0:
                t, mode = int(rng.integers(0, vocab)), int(rng.choice([0, 1, 2]))
                flat.append((t, occur, mode, boost))
                nested.append((t, occur, mode, boost))
            elif kind == 1:
                terms, comp = sorted(set(rng.integers(0, vocab, int(rng.integers(1, 4))).tolist())), bool(rng.random() < 0.3)
                flat.append((0, occur, 2, boost, terms, comp))
                nested.append(("set", occur, boost, terms, comp))
            else:
                terms = rng.integers(0, vocab, int(rng.integers(2, 4))).tolist()
                flat.append((0, occur, 0, boost, terms, False, True))
                nested.append(("phrase", occur, boost, terms, 0))
        wd, ws, _, wt, _ = idx.search_ex(flat, 30)
        gd, gs, gt = orc.bm25_nested_search(idx, nested, 30)
        assert gt == wt and np.array_equal(gd, wd) and np.array_equal(gs.view(np.uint32), ws.view(np.uint32)), nested

    def count(text, phrase, slop):
        toks = text.split()
        return orc.phrase_count_with_slop([[i for i, w in enumerate(toks) if w == p] for p in phrase.split()], slop)

    assert count("a x b c", "a b c", 1) == 1 and count("a b x c", "a b c", 1) == 1 and count("a x b x c", "a b c", 1) == 0
    assert count("a x b x c", "a b c", 2) == 1
    assert count("b a", "a b", 1) == 0 and count("b a", "a b", 2) == 1 and count("a b", "a b", 0) == 1
    assert count("a b a b", "a b", 0) == 2 and count("a c b", "a b", 1) == 1 and count("a c c b", "a b", 1) == 0
    # slop 0 is the exact phrase for random documents
    for _ in range(200):
        toks = rng.integers(0, 4, int(rng.integers(2, 12))).tolist()
        ph = rng.integers(0, 4, int(rng.integers(2, 4))).tolist()
        exact = sum(1 for s in range(len(toks) - len(ph) + 1) if toks[s:s + len(ph)] == ph)
        lists = [[i for i, w in enumerate(toks) if w == p] for p in ph]
        assert orc.phrase_count_with_slop(lists, 0) == (exact if all(lists) else 0)
