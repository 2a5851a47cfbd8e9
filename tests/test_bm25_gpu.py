"""GPU parity tests of the BM25 path: the HIP term-at-a-time kernel (through the C ABI) vs the CPU
oracle (tantivy's formulas restated) — bit-exact scores, doc addresses, ranks and totals.
The reference's own tests pin only counts and score thresholds (BM25 parity unpinned);
those are mirrored at the bottom."""
import numpy as np
import pytest

from nucliadb_amd import _lib
from nucliadb_amd.bm25 import Bm25Searcher, Bm25Segment, Clause, SearchAfter, tokenize

pytestmark = pytest.mark.gpu
S, M, N = _lib.OCCUR_SHOULD, _lib.OCCUR_MUST, _lib.OCCUR_MUST_NOT
FREQ, BASIC, CONST = _lib.TF_FREQ, _lib.TF_BASIC, _lib.CONST_SCORE


def zipf_corpus(rng, n_docs, vocab, mean_len=48):
    """T-zipf of SURVEY §8d, scaled: term ids ~ Zipf(1.0), doc length ~ lognormal clipped to [4, 2000]."""
    lens = np.clip(np.round(rng.lognormal(np.log(mean_len), 0.6, n_docs)), 4, 2000).astype(np.int64)
    p = 1.0 / np.arange(1, vocab + 1)
    p /= p.sum()
    flat = rng.choice(vocab, size=int(lens.sum()), p=p)
    return np.split(flat, np.cumsum(lens)[:-1])


def bits(a):
    return np.asarray(a, np.float32).view(np.uint32)


def compare(orc, seg, searcher, queries, k, after=None, segment_ord=0):
    docaddr, score, count, total, postings = searcher.search_batch(queries, k, after)
    oidx = orc.Bm25Index(seg.term_offsets, seg.doc_ids, seg.tfs, seg.fieldnorm_ids, seg.total_num_tokens, seg.alive)
    for i, q in enumerate(queries):
        a = None if after is None or after[i] is None else (after[i].score, after[i].tie_break, after[i].docaddr)
        wd, ws, wt = oidx.search([(c.term, c.occur, c.mode, c.boost) for c in q], k, after=a, segment_ord=segment_ord)
        assert total[i] == wt, (i, total[i], wt)
        assert count[i] == len(wd), (i, count[i], len(wd))
        assert np.array_equal(docaddr[i, : count[i]], wd), (i, docaddr[i, : count[i]], wd)
        assert np.array_equal(bits(score[i, : count[i]]), bits(ws)), (i, score[i, : count[i]], ws)
        assert postings[i] == sum(int(seg.term_offsets[c.term + 1] - seg.term_offsets[c.term]) for c in q)


@pytest.fixture(scope="module")
def corpus():
    rng = np.random.default_rng(1234567890)
    vocab = 5000
    docs = zipf_corpus(rng, 60000, vocab)
    return Bm25Segment.from_term_docs(docs, vocab), vocab


def test_fieldnorm_table_and_idf(orc):
    L = _lib.lib()
    table = orc.fieldnorm_table()
    assert [L.nidx_gpu_fieldnorm_from_id(i) for i in range(256)] == table.tolist()
    for n in list(range(0, 3000)) + [2**20, 2**31, 2**32 - 1]:
        assert L.nidx_gpu_fieldnorm_to_id(n) == orc.fieldnorm_to_id(n)
    rng = np.random.default_rng(0)
    for _ in range(500):
        N_ = int(rng.integers(1, 20_000_000))
        n_ = int(rng.integers(0, N_ + 1))
        assert np.float32(L.nidx_gpu_bm25_idf(n_, N_)).view(np.uint32) == np.float32(orc.bm25_idf(n_, N_)).view(np.uint32)


def test_or_queries_match_oracle(orc, corpus):
    seg, vocab = corpus
    rng = np.random.default_rng(1)
    s = Bm25Searcher.open([seg])
    queries = [[Clause(int(t)) for t in rng.integers(0, vocab, int(rng.integers(1, 6)))] for _ in range(64)]
    queries += [[Clause(0), Clause(1), Clause(2)], [Clause(3, boost=2.5), Clause(4000)], []]  # very dense lists; empty query
    compare(orc, seg, s, queries, 20)
    compare(orc, seg, s, queries, 1)
    compare(orc, seg, s, queries, 64)
    compare(orc, seg, s, queries, 201)   # paragraph search asks for k+1 with result_per_page up to 200
    compare(orc, seg, s, queries, 501)   # ... or max(top_k, rank-fusion window, reranker window) = 500
    s.close()


def test_boolean_mix_and_modes(orc, corpus):
    seg, vocab = corpus
    rng = np.random.default_rng(2)
    s = Bm25Searcher.open([seg])
    queries = []
    for _ in range(80):
        q = []
        for _ in range(int(rng.integers(1, 7))):
            q.append(Clause(int(rng.integers(0, 300)), int(rng.choice([S, S, M, N])), int(rng.choice([FREQ, BASIC, CONST])),
                            float(rng.choice([1.0, 0.5, 2.0]))))
        queries.append(q)
    # the paragraph shape (nidx_paragraph/src/search_query.rs:185-243): Should Basic terms + Must const filters
    queries.append([Clause(10, S, BASIC), Clause(11, S, BASIC), Clause(1, M, CONST, 1.0), Clause(0, M, BASIC)])
    queries.append([Clause(5, N), Clause(6, N)])          # only MustNot: nothing matches
    queries.append([Clause(7, M), Clause(7, M)])          # the same term twice
    compare(orc, seg, s, queries, 20)
    s.close()


def test_alive_bitset_and_search_after(orc):
    rng = np.random.default_rng(3)
    vocab = 800
    docs = zipf_corpus(rng, 20000, vocab, mean_len=12)
    alive = orc.bitset(20000, ones=np.nonzero(rng.random(20000) < 0.7)[0].tolist())
    seg = Bm25Segment.from_term_docs(docs, vocab, alive=alive)
    s = Bm25Searcher.open([seg])
    queries = [[Clause(int(t), S, BASIC) for t in rng.integers(0, 60, 3)] for _ in range(24)]  # tf == 1: many exact ties
    compare(orc, seg, s, queries, 20)
    # page through with the (score, docaddr) cursor like nidx/tests/integration/search_after.rs
    docaddr, score, count, total, _ = s.search_batch(queries, 20)
    after = [SearchAfter(float(score[i, 7]), 1, int(docaddr[i, 7])) if count[i] > 7 else None for i in range(len(queries))]
    compare(orc, seg, s, queries, 20, after=after)
    after = [SearchAfter(float(score[i, 3]), t, int(docaddr[i, 3])) if count[i] > 3 else None for i, t in zip(range(len(queries)), [0, 1, 2] * 8)]
    compare(orc, seg, s, queries, 20, after=after)
    s.close()


def test_multi_segment_equals_one_segment():
    """Statistics are searcher-wide (tantivy Bm25Weight::for_terms): splitting the corpus into segments
    must not change any score, and DocAddress = (segment_ord << 32) | doc orders ties."""
    rng = np.random.default_rng(4)
    vocab = 1000
    docs = zipf_corpus(rng, 9000, vocab, mean_len=10)
    whole = Bm25Segment.from_term_docs(docs, vocab)
    parts = [Bm25Segment.from_term_docs(docs[:4000], vocab), Bm25Segment.from_term_docs(docs[4000:], vocab)]
    queries = [[Clause(int(t)) for t in rng.integers(0, 100, 3)] for _ in range(32)]
    s1, s2 = Bm25Searcher.open([whole]), Bm25Searcher.open(parts)
    d1, sc1, c1, t1, _ = s1.search_batch(queries, 20)
    d2, sc2, c2, t2, _ = s2.search_batch(queries, 20)
    assert np.array_equal(t1, t2) and np.array_equal(c1, c2)
    for i in range(len(queries)):
        g = [(int(a) >> 32) * 4000 + (int(a) & 0xFFFFFFFF) for a in d2[i, : c2[i]]]
        assert g == [int(a) for a in d1[i, : c1[i]]]
        assert np.array_equal(bits(sc1[i, : c1[i]]), bits(sc2[i, : c2[i]]))
    s1.close()
    s2.close()


def test_reference_min_score_counts():
    """nidx_text/tests/test_search.rs:311-332 and nidx_paragraph/tests/reader.rs:315-342: a one-word
    query scores well below 30 — hits survive min_score 0 and vanish at 30/100 while `total` stays."""
    texts = ["This is one of the best ways to test", "should enough", "shoupd enough test", "enough test for it to be a test",
             "some other text that does not match"]
    vocab: dict = {}
    docs = [np.array([vocab.setdefault(t, len(vocab)) for t in tokenize(x)], dtype=np.int64) for x in texts]
    seg = Bm25Segment.from_term_docs(docs, len(vocab))
    s = Bm25Searcher.open([seg])
    docaddr, score, count, total, _ = s.search_batch([[Clause(vocab["should"])], [Clause(vocab["enough"]), Clause(vocab["test"])]], 10)
    assert count[0] == 1 and total[0] == 1 and 0 < score[0, 0] < 30
    assert count[1] == 4 and total[1] == 4 and (score[1, :4] < 30).all() and (score[1, :4] > 0).all()
    assert sum(1 for x in score[1, :4] if x >= 30) == 0
    s.close()


def test_required_should_groups_and_wide_queries(orc, corpus):
    """Several required Should groups (occur = OCCUR_SHOULD_GROUP + g: the keyword group, an Or formula, the prefilter's
    SetQuery pair — nidx_paragraph/src/search_query.rs:88-143), queries with more clauses than a window has rows, and
    queries with more than 32 clauses (64-bit hit masks)."""
    seg, vocab = corpus
    rng = np.random.default_rng(7)
    s = Bm25Searcher.open([seg])
    G = _lib.OCCUR_SHOULD_GROUP
    queries = []
    for _ in range(60):
        q = []
        for g in range(int(rng.integers(1, 4))):           # 1..3 groups of 1..3 clauses over dense terms
            for _ in range(int(rng.integers(1, 4))):
                q.append(Clause(int(rng.integers(0, 120)), G + g, int(rng.choice([FREQ, BASIC, CONST])), float(rng.choice([1.0, 0.5]))))
        if rng.random() < 0.5:
            q.append(Clause(int(rng.integers(0, 40)), M, BASIC))
        if rng.random() < 0.3:
            q.append(Clause(int(rng.integers(0, 200)), N))
        if rng.random() < 0.3:
            q.append(Clause(int(rng.integers(0, 500)), S))
        order = rng.permutation(len(q))
        queries.append([q[i] for i in order])              # groups interleaved with the other clauses
    queries.append([Clause(3, G), Clause(4, G + 1), Clause(5, G + 7)])   # the highest group id
    compare(orc, seg, s, queries, 20)
    wide = [[Clause(int(t), int(rng.choice([S, S, S, M])) if t < 30 else S) for t in rng.permutation(400)[: int(n)]] for n in (5, 9, 12, 20, 33, 40, 64)]
    compare(orc, seg, s, wide, 20)
    compare(orc, seg, s, wide + queries[:8], 10)            # one launch mixing 64-bit-mask and narrow queries
    s.close()


def test_general_kernel_on_narrow_queries(orc, corpus, monkeypatch):
    """Narrow queries normally take bm25_fast_kernel; NIDX_GPU_BM25_WIDE routes them through the general kernel
    (bm25_rows_kernel, FAST and PACKED windows): both must give the oracle's answer."""
    seg, vocab = corpus
    rng = np.random.default_rng(11)
    monkeypatch.setenv("NIDX_GPU_BM25_WIDE", "1")
    s = Bm25Searcher.open([seg])
    queries = [[Clause(int(t), int(rng.choice([S, S, M, N])), int(rng.choice([FREQ, BASIC, CONST]))) for t in rng.integers(0, 400, int(rng.integers(1, 8)))]
               for _ in range(64)]
    compare(orc, seg, s, queries, 20)
    compare(orc, seg, s, queries, 100)
    s.close()


@pytest.mark.parametrize("union_mode", ["2", "2-throughput-shape", "4", "0"])
def test_union_kernel_and_hash_kernel_agree_with_the_oracle(orc, corpus, monkeypatch, union_mode):
    """The union kernels (postings are final unless a bitmap filter says their document may occur twice; those are resolved
    exactly) against the oracle, on query families that make their slow paths the common ones: NIDX_GPU_BM25_UNION=2 sends every
    query of <= 8 plain term clauses through bm25_stream_kernel (4: through bm25_union_kernel) — dense terms (term 0 is in nearly
    every document: the involved list overflows, the doc range is cut in half and retried), Must / MustNot / required Should
    groups, constant scores, negative and zero boosts, k from 1 to 501, the alive bitset and the search-after cursor.  Mode 0 never
    uses them: the same answers from the hash kernels."""
    seg, vocab = corpus
    rng = np.random.default_rng(21)
    if union_mode == "2-throughput-shape":
        # the slicing a batch gets when other tickets are out (csrc/bm25_index.cpp crowded_shape: the fewest slices the bitmaps'
        # collision estimate allows — on this small corpus whole queries in one item, the overflow / halve / retry path the rule)
        monkeypatch.setenv("NIDX_GPU_BM25_CROWDED", "1")
        union_mode = "2"
    monkeypatch.setenv("NIDX_GPU_BM25_UNION", union_mode)
    s = Bm25Searcher.open([seg])
    G = _lib.OCCUR_SHOULD_GROUP
    sparse = [[Clause(int(t)) for t in rng.integers(200, vocab, int(rng.integers(1, 9)))] for _ in range(48)]
    dense = [[Clause(0), Clause(1), Clause(2)], [Clause(0, M), Clause(1, M), Clause(5)], [Clause(0, M, CONST, 0.5), Clause(300), Clause(301)],
             [Clause(0), Clause(0)], [Clause(1, N), Clause(0), Clause(7)], [Clause(2, boost=-1.5), Clause(3, boost=0.0), Clause(4, boost=-0.0)],
             [Clause(0, G), Clause(1, G), Clause(2, G + 1), Clause(3, G + 1), Clause(4, M, BASIC)], []]
    mixed = []
    for _ in range(64):
        q = []
        for _ in range(int(rng.integers(1, 9))):
            q.append(Clause(int(rng.integers(0, 400)), int(rng.choice([S, S, M, N, G, G + 1])), int(rng.choice([FREQ, BASIC, CONST])),
                            float(rng.choice([1.0, 0.5, 2.0]))))
        mixed.append(q)
    for k in (20, 1, 64, 201, 501):
        compare(orc, seg, s, sparse + dense, k)
    compare(orc, seg, s, mixed, 20)
    compare(orc, seg, s, mixed + sparse, 10)
    s.close()
    # alive bitset + search-after (many exact score ties: tf == 1)
    rng = np.random.default_rng(3)
    docs = zipf_corpus(rng, 20000, 800, mean_len=12)
    alive = orc.bitset(20000, ones=np.nonzero(rng.random(20000) < 0.7)[0].tolist())
    seg2 = Bm25Segment.from_term_docs(docs, 800, alive=alive)
    s = Bm25Searcher.open([seg2])
    queries = [[Clause(int(t), S, BASIC) for t in rng.integers(0, 60, 3)] for _ in range(24)]
    compare(orc, seg2, s, queries, 20)
    docaddr, score, count, total, _ = s.search_batch(queries, 20)
    after = [SearchAfter(float(score[i, 3]), t, int(docaddr[i, 3])) if count[i] > 3 else None for i, t in zip(range(len(queries)), [0, 1, 2] * 8)]
    compare(orc, seg2, s, queries, 20, after=after)
    s.close()


def _nested_to_oracle(q):
    return [("sub", c.occur, c.boost, [(l.term, l.occur, l.mode, l.boost) for l in c.subquery]) if c.subquery is not None
            else (c.term, c.occur, c.mode, c.boost) for c in q]


def test_nested_boolean_queries_match_the_oracle(orc, corpus):
    """BooleanQuerys inside the BooleanQuery (NIDX_BM25_SUBQUERY): an AND inside an OR, a negated conjunction, a boosted
    conjunction, a conjunction with its own required Should group and optional Should leaves — what tantivy's QueryParser builds
    for `a OR (b AND c)`, `x AND NOT (a AND b)`, `(a AND b)^2.5` (nidx_text/src/reader.rs:357-376) and what a conjunction or a
    negation inside an `Or` filtering formula is (nidx_paragraph/src/search_query.rs:88-143).  The nested query is materialised on
    the device as a pre-scored posting list; the oracle evaluates the tree document at a time (orc.bm25_nested_search, itself
    pinned to the C oracle on flat queries in tests/test_oracle_golden.py)."""
    seg, vocab = corpus
    rng = np.random.default_rng(31)
    s = Bm25Searcher.open([seg])
    G = _lib.OCCUR_SHOULD_GROUP
    oidx = orc.Bm25Index(seg.term_offsets, seg.doc_ids, seg.tfs, seg.fieldnorm_ids, seg.total_num_tokens, seg.alive)

    def leaf(lo, hi, occur, mode=None, boost=None):
        return Clause(int(rng.integers(lo, hi)), occur, int(rng.choice([FREQ, BASIC, CONST])) if mode is None else mode,
                      float(rng.choice([1.0, 0.5, 2.0])) if boost is None else boost)

    queries = [
        [Clause(300), Clause(0, S, subquery=[Clause(1, M), Clause(2, M)])],                                # a OR (b AND c)
        [Clause(5, M), Clause(0, N, subquery=[Clause(1, M), Clause(2, M)])],                                # x AND NOT (a AND b)
        [Clause(0, S, boost=2.5, subquery=[Clause(3, M), Clause(4, M, BASIC)]), Clause(900)],               # (a AND b)^2.5 OR c
        [Clause(0, M, subquery=[Clause(0, M, CONST, 1.0), Clause(7, N)]), Clause(20, S)],                   # Not(l) inside a formula: AllQuery-like Must + MustNot
        [Clause(0, G, subquery=[Clause(10, M), Clause(11, G), Clause(12, G), Clause(400, S)]), Clause(13, G), Clause(2, M, BASIC)],
        [Clause(0, S, subquery=[Clause(4000, M), Clause(4001, M)])],                                         # a conjunction that matches (almost) nothing
    ]
    for _ in range(40):
        q = []
        for _ in range(int(rng.integers(1, 4))):
            q.append(leaf(0, 500, int(rng.choice([S, S, M, N]))))
        for _ in range(int(rng.integers(1, 3))):
            sub = [leaf(0, 60, M)] + [leaf(0, 200, int(rng.choice([M, N, S, G, G + 1]))) for _ in range(int(rng.integers(1, 5)))]
            order = rng.permutation(len(sub))
            q.append(Clause(0, int(rng.choice([S, S, M, N, G])), boost=float(rng.choice([1.0, 0.5, 3.0])), subquery=[sub[i] for i in order]))
        order = rng.permutation(len(q))
        queries.append([q[i] for i in order])
    for k in (20, 3, 120):
        docaddr, score, count, total, _ = s.search_batch(queries, k)
        for i, q in enumerate(queries):
            wd, ws, wt = orc.bm25_nested_search(oidx, _nested_to_oracle(q), k)
            assert total[i] == wt, (i, total[i], wt)
            assert count[i] == len(wd), (i, count[i], len(wd))
            assert np.array_equal(docaddr[i, : count[i]], wd), (i, docaddr[i, : count[i]], wd)
            assert np.array_equal(bits(score[i, : count[i]]), bits(ws)), (i, score[i, : count[i]], ws)
    s.close()


def _tree_to_oracle(c):
    if c.subquery is not None:
        return ("sub", c.occur, c.boost, [_tree_to_oracle(l) for l in c.subquery])
    if c.term_set is not None and c.phrase:
        return ("phrase", c.occur, c.boost, [int(t) for t in c.term_set], c.slop)
    if c.term_set is not None:
        return ("set", c.occur, c.boost, [int(t) for t in c.term_set], c.complement)
    return (c.term, c.occur, c.mode, c.boost)


def test_boolean_trees_of_any_depth_match_the_oracle(orc):
    """Round 4: a leaf of a nested BooleanQuery may be a term set, a phrase (with or without slop) or ANOTHER nested query, and a
    nested query needs no Must leaf (its candidates are then the union of one required Should group, or of its Should leaves) —
    every shape tantivy's QueryParser can build for a parenthesised body (nidx_text/src/reader.rs:357-376) and every nesting of a
    filtering formula (nidx_paragraph/src/search_query.rs:88-143, query_io.rs:28-66).  Random trees up to four levels deep,
    materialised on the device children first, against the oracle's recursive document-at-a-time evaluation: doc addresses, ranks,
    f32 score bits and Count."""
    rng = np.random.default_rng(91)
    vocab = 120
    p = 1.0 / np.arange(1, vocab + 1)
    p /= p.sum()
    docs = [rng.choice(vocab, size=int(rng.integers(3, 40)), p=p) for _ in range(8000)]
    alive = orc.bitset(8000, ones=np.nonzero(rng.random(8000) < 0.9)[0].tolist())
    seg = Bm25Segment.from_term_docs(docs, vocab, alive=alive, with_positions=True)
    s = Bm25Searcher.open([seg])
    oidx = orc.Bm25Index(seg.term_offsets, seg.doc_ids, seg.tfs, seg.fieldnorm_ids, seg.total_num_tokens, seg.alive, seg.pos_offsets, seg.positions)
    G = _lib.OCCUR_SHOULD_GROUP

    def leaf(occur):
        kind = rng.random()
        boost = float(rng.choice([1.0, 0.5, 2.0, 3.0]))
        if kind < 0.55:
            return Clause(int(rng.integers(0, vocab)), occur, int(rng.choice([FREQ, BASIC, CONST])), boost)
        if kind < 0.75:
            return Clause(0, occur, CONST, boost, term_set=sorted(set(rng.integers(0, vocab, int(rng.integers(1, 5))).tolist())), complement=bool(rng.random() < 0.2))
        return Clause(0, occur, FREQ, boost, term_set=rng.integers(0, 12, int(rng.integers(2, 4))).tolist(), phrase=True, slop=int(rng.choice([0, 0, 1, 3])))

    def tree(depth, occur):
        shape = rng.random()
        if shape < 0.3:     # a conjunction: Must leaves, maybe exclusions
            occurs = [M] * int(rng.integers(1, 4)) + [N] * int(rng.integers(0, 2))
        elif shape < 0.55:  # a disjunction: only Should leaves (no list to walk: the union is materialised)
            occurs = [S] * int(rng.integers(1, 5)) + [N] * int(rng.integers(0, 2))
        elif shape < 0.75:  # required groups without a Must leaf
            occurs = [G] * int(rng.integers(1, 3)) + [G + 1] * int(rng.integers(1, 3)) + [S] * int(rng.integers(0, 2))
        else:               # everything
            occurs = [int(rng.choice([S, M, N, G, G + 3])) for _ in range(int(rng.integers(2, 7)))]
        kids = [tree(depth - 1, o) if depth > 0 and rng.random() < 0.45 else leaf(o) for o in occurs]
        order = rng.permutation(len(kids))
        return Clause(0, occur, boost=float(rng.choice([1.0, 0.5, 2.5])), subquery=[kids[i] for i in order])

    queries = [
        [Clause(0, M, subquery=[Clause(1, S), Clause(2, S)])],                                                     # (b OR c) as a nested query
        [Clause(5, S), Clause(0, S, subquery=[Clause(1, M), Clause(0, M, subquery=[Clause(2, S), Clause(0, S, subquery=[Clause(3, M), Clause(4, M)])])])],
        [Clause(0, S, subquery=[Clause(0, G, subquery=[Clause(1, M), Clause(2, M)]), Clause(0, G, subquery=[Clause(3, M), Clause(4, M)])])],
        [Clause(0, M, subquery=[Clause(7, N)]), Clause(3, S)],                                                     # only exclusions: matches nothing
        [Clause(0, S, subquery=[Clause(0, M, FREQ, 1.0, term_set=[0, 1], phrase=True, slop=1), Clause(0, M, CONST, 2.0, term_set=[5, 6, 7])]), Clause(90)],
        [Clause(0, N, subquery=[Clause(0, S, subquery=[Clause(1, M), Clause(2, M)]), Clause(9, S)]), Clause(0, M, CONST, 1.0, term_set=[3], complement=True)],
        [Clause(0, S, FREQ, 2.0, term_set=[0, 1, 2], phrase=True, slop=2), Clause(0, S, FREQ, 1.0, term_set=[1, 0], phrase=True, slop=2)],
    ]
    for _ in range(50):
        q = [leaf(int(rng.choice([S, S, M, N]))) for _ in range(int(rng.integers(0, 3)))]
        q += [tree(int(rng.integers(1, 4)), int(rng.choice([S, S, M, N, G]))) for _ in range(int(rng.integers(1, 3)))]
        order = rng.permutation(len(q))
        queries.append([q[i] for i in order])
    for k in (20, 3):
        r = s.search_batch_ex(queries, k)
        for i, q in enumerate(queries):
            wd, ws, wt = orc.bm25_nested_search(oidx, [_tree_to_oracle(c) for c in q], k)
            n = int(r["count"][i])
            assert r["total"][i] == wt, (i, r["total"][i], wt)
            assert n == len(wd), (i, n, len(wd))
            assert np.array_equal(r["docaddr"][i, :n], wd), (i, r["docaddr"][i, :n], wd)
            assert np.array_equal(bits(r["score"][i, :n]), bits(ws)), (i, r["score"][i, :n], ws)
    # the limits are errors, not wrong answers: 33 leaves in one nested query; a nested query that refers to a later one
    with pytest.raises(_lib.NidxGpuError):
        s.search_batch_ex([[Clause(0, S, subquery=[Clause(i, S) for i in range(33)])]], 5)
    s.close()


def test_pipelined_search_equals_the_synchronous_one(orc, corpus):
    """nidx_gpu_bm25_search_submit / _wait: several batches in flight, waited for out of order, give what nidx_gpu_bm25_search gives
    for each; a seventeenth outstanding ticket is NIDX_ERR_BUSY; a ticket is waited for once; requests the pipeline does not cover (term
    sets) run inside submit and still come back through wait."""
    import ctypes as C

    seg, vocab = corpus
    rng = np.random.default_rng(41)
    s = Bm25Searcher.open([seg])
    batches = [[[Clause(int(t), int(rng.choice([S, S, M, N]))) for t in rng.integers(0, vocab, int(rng.integers(1, 6)))] for _ in range(int(n))]
               for n in (64, 1, 200, 33)]
    want = [s.search_batch(b, 20) for b in batches]
    for _ in range(3):
        tickets = [s.submit(b, 20) for b in batches * 4]
        with pytest.raises(_lib.NidxGpuError) as e:
            s.submit(batches[0], 20)
        assert "not been waited" in str(e.value)
        for i in (2, 0, 7, 13, 3, 15, 5, 1, 9, 6, 4, 8, 12, 10, 14, 11):
            got = s.wait(tickets[i])
            for g, w in zip(got, want[i % 4]):
                assert np.array_equal(g.view(np.uint32) if g.dtype == np.float32 else g, w.view(np.uint32) if w.dtype == np.float32 else w), i
        out = np.zeros(1, np.uint32)
        assert _lib.lib().nidx_gpu_bm25_search_wait(s._handle, tickets[0], None, None, out.ctypes.data, None, None) == _lib.NIDX_ERR_INVALID_ARGUMENT
    # a term-set query through the pipeline entry points (runs inside submit)
    q = [[Clause(0, S, CONST, 0.5, term_set=[3, 4, 5]), Clause(7)]]
    ref = s.search_batch_ex(q, 10)
    cl = (_lib.Bm25ClauseC * 2)()
    cl[0].term, cl[0].occur, cl[0].mode, cl[0].boost = _lib.BM25_TERM_SET | 0, S, CONST, 0.5
    cl[1].term, cl[1].occur, cl[1].mode, cl[1].boost = 7, S, FREQ, 1.0
    st, so, offs = np.array([3, 4, 5], np.uint32), np.array([0, 3], np.uint64), np.array([0, 2], np.uint64)
    po = np.zeros(1, np.uint64)
    opt = _lib.Bm25SearchOptionsC()
    opt.k, opt.order_field = 10, -1
    opt.term_set_terms, opt.term_set_offsets, opt.n_term_sets = st.ctypes.data, so.ctypes.data, 1
    opt.phrase_offsets = po.ctypes.data
    opt.subquery_offsets = po.ctypes.data
    t = C.c_uint64(0)
    _lib.check(_lib.lib().nidx_gpu_bm25_search_submit(s._handle, cl, offs.ctypes.data, 1, C.byref(opt), C.byref(t)))
    d, sc, cnt = np.zeros((1, 10), np.uint64), np.zeros((1, 10), np.float32), np.zeros(1, np.uint32)
    _lib.check(_lib.lib().nidx_gpu_bm25_search_wait(s._handle, t.value, d.ctypes.data, sc.ctypes.data, cnt.ctypes.data, None, None))
    assert cnt[0] == ref["count"][0] and np.array_equal(d[0, : cnt[0]], ref["docaddr"][0, : cnt[0]]) and np.array_equal(bits(sc[0, : cnt[0]]), bits(ref["score"][0, : cnt[0]]))
    s.close()


def test_score_floor_changes_no_result(orc, corpus, monkeypatch):
    """The streaming scorer starts every work item's bar at the query's score floor (bm25_index.cpp / bm25_aux.hip: at least k documents are
    known to reach it).  Queries of Should terms with and without it, through the stream kernel for every query (NIDX_GPU_BM25_UNION=2) and
    by the planner's own routing, against the oracle and against a searcher opened without floors."""
    seg, vocab = corpus
    rng = np.random.default_rng(77)
    queries = [[Clause(int(t), S, int(rng.choice([FREQ, BASIC])), float(rng.choice([1.0, 0.25, 4.0]))) for t in rng.integers(0, vocab, int(rng.integers(1, 7)))]
               for _ in range(96)]
    queries += [[Clause(0), Clause(1)], [Clause(4999), Clause(4998), Clause(2)], [Clause(17), Clause(17)]]
    monkeypatch.setenv("NIDX_GPU_BM25_FLOOR", "0")
    plain = Bm25Searcher.open([seg])
    monkeypatch.delenv("NIDX_GPU_BM25_FLOOR")
    floors = Bm25Searcher.open([seg])
    for union in ("1", "2"):
        monkeypatch.setenv("NIDX_GPU_BM25_UNION", union)
        for k in (1, 20, 64, 201, 501):
            compare(orc, seg, floors, queries, k)
            a, b = floors.search_batch(queries, k), plain.search_batch(queries, k)
            for x, y in zip(a, b):
                assert np.array_equal(np.asarray(x).view(np.uint8), np.asarray(y).view(np.uint8))
    plain.close()
    floors.close()


def test_score_floor_with_every_posting_at_the_floor(orc, monkeypatch):
    """documents of one length, tf = 1: every posting of a term scores exactly the floor; nothing may be dropped and the lowest doc ids win"""
    rng = np.random.default_rng(5)
    vocab = 40
    docs = [rng.choice(vocab, size=8, replace=False) for _ in range(30000)]
    seg = Bm25Segment.from_term_docs(docs, vocab)
    s = Bm25Searcher.open([seg])
    queries = [[Clause(int(t), S, BASIC)] for t in range(0, 40, 3)] + [[Clause(1, S, BASIC), Clause(2, S, BASIC), Clause(3, S, BASIC)], [Clause(5), Clause(6, boost=2.0)]]
    monkeypatch.setenv("NIDX_GPU_BM25_UNION", "2")
    for k in (1, 20, 64):
        compare(orc, seg, s, queries, k)
    s.close()


def test_fused_merge_changes_no_result(orc, corpus, monkeypatch):
    """k <= 64 and every query a term union: the scoring launch merges the slices itself (the last wave of every group of eight slices, then the
    last group: kernels.h Bm25FusedMerge) instead of bm25_merge_kernel in a launch of its own.  One slice, a few, two levels (slices pinned to
    256 postings: ~150 of them for the densest term), with and without extras (deletions), against the oracle and against the two-launch path."""
    seg, vocab = corpus
    rng = np.random.default_rng(99)
    queries = [[Clause(int(t), S, int(rng.choice([FREQ, BASIC]))) for t in rng.integers(0, vocab, int(rng.integers(1, 5)))] for _ in range(60)]
    queries += [[Clause(0), Clause(1), Clause(2)], [Clause(0)], [Clause(4999)], [Clause(3, boost=2.0), Clause(0)], []]
    monkeypatch.setenv("NIDX_GPU_BM25_UNION", "2")   # every query through the stream kernel: the condition of the fused merge
    s = Bm25Searcher.open([seg])
    for slice_env in (None, "256", "1024"):
        if slice_env is None:
            monkeypatch.delenv("NIDX_GPU_BM25_SLICE", raising=False)
        else:
            monkeypatch.setenv("NIDX_GPU_BM25_SLICE", slice_env)
        for k in (1, 20, 64):
            monkeypatch.delenv("NIDX_GPU_BM25_FUSED_MERGE", raising=False)
            compare(orc, seg, s, queries, k)
            a = s.search_batch(queries, k)
            a2 = s.search_batch(queries, k)   # the arrival counters are back at zero after a launch
            monkeypatch.setenv("NIDX_GPU_BM25_FUSED_MERGE", "0")
            b = s.search_batch(queries, k)
            for x, y, z in zip(a, b, a2):
                assert np.array_equal(np.asarray(x).view(np.uint8), np.asarray(y).view(np.uint8))
                assert np.array_equal(np.asarray(x).view(np.uint8), np.asarray(z).view(np.uint8))
    s.close()
    monkeypatch.delenv("NIDX_GPU_BM25_FUSED_MERGE", raising=False)
    monkeypatch.setenv("NIDX_GPU_BM25_SLICE", "512")
    alive = orc.bitset(seg.n_docs, ones=np.nonzero(rng.random(seg.n_docs) < 0.8)[0].tolist())
    dead = Bm25Segment(seg.term_offsets, seg.doc_ids, seg.tfs, seg.fieldnorm_ids, seg.total_num_tokens, alive)
    s = Bm25Searcher.open([dead])
    compare(orc, dead, s, queries, 20)
    s.close()
