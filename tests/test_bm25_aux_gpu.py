"""GPU parity tests of the pieces around the BM25 scorer: the fuzzy automaton over the term
dictionary, term-set (fuzzy) clauses, TopDocs ordered by a fast field and facet counts — device vs the CPU oracle."""
import numpy as np
import pytest

from nucliadb_amd import _lib
from nucliadb_amd.bm25 import Bm25Searcher, Bm25Segment, Clause
from test_bm25_gpu import bits, zipf_corpus

pytestmark = pytest.mark.gpu
S, M, N, G = _lib.OCCUR_SHOULD, _lib.OCCUR_MUST, _lib.OCCUR_MUST_NOT, _lib.OCCUR_SHOULD_GROUP
FREQ, BASIC, CONST = _lib.TF_FREQ, _lib.TF_BASIC, _lib.CONST_SCORE


def random_words(rng, n):
    alphabet = list("abcdefghijklmnopqrstuvwxyz") + ["ñ", "é", "ü", "ß", "道"]
    words = set()
    while len(words) < n:
        ln = int(rng.integers(1, 11))
        words.add("".join(rng.choice(alphabet, ln)))
    return sorted(words)


def test_fuzzy_automaton_over_the_dictionary(orc):
    rng = np.random.default_rng(7)
    terms = random_words(rng, 6000)
    # neighbours of a few probe words, so that every edit kind occurs
    probes = ["should", "enough", "niño", "ab", "abc", "道路", "tantivy", "a"]
    extra = set()
    for w in probes:
        for i in range(len(w) + 1):
            extra.add(w[:i] + "x" + w[i:])            # insertion
            if i < len(w):
                extra.add(w[:i] + w[i + 1:])          # deletion
                extra.add(w[:i] + "y" + w[i + 1:])    # substitution
            if i + 1 < len(w):
                extra.add(w[:i] + w[i + 1] + w[i] + w[i + 2:])  # transposition
        extra.add(w + "zz")
        extra.add(w + "suffix")
        extra.add("q" + w + "q")
    terms = sorted(set(terms) | {e for e in extra if e} | set(probes))
    vocab = len(terms)
    seg = Bm25Segment.from_term_docs([np.array([0, 1], np.int64)], vocab)
    s = Bm25Searcher.open([seg])
    try:
        s.set_dictionary(terms)
        queries = probes + ["shoupd", "sJoupd", "enaugh", "enaugJ", "enoguh", "eonguh"] + [terms[i] for i in rng.integers(0, vocab, 40)]
        for w in queries:
            for prefix in (False, True):
                got = s.fuzzy_terms(w, prefix)
                want = orc.fuzzy_terms(terms, w, 1, prefix)
                assert np.array_equal(got, want), (w, prefix, [terms[i] for i in set(got) ^ set(want)][:10])
        # the reference's own cases (nidx_paragraph/tests/reader.rs:262-275): one typo matches, two do not
        idx = {t: i for i, t in enumerate(terms)}
        assert idx["should"] in s.fuzzy_terms("shoupd") and idx["should"] not in s.fuzzy_terms("sJoupd")
        assert idx["enough"] in s.fuzzy_terms("enoguh") and idx["enough"] not in s.fuzzy_terms("eonguh")
    finally:
        s.close()


@pytest.fixture(scope="module")
def corpus():
    rng = np.random.default_rng(99)
    vocab = 3000
    docs = zipf_corpus(rng, 40000, vocab, mean_len=24)
    alive = None
    return Bm25Segment.from_term_docs(docs, vocab, alive=alive), vocab


def oracle_index(orc, seg):
    return orc.Bm25Index(seg.term_offsets, seg.doc_ids, seg.tfs, seg.fieldnorm_ids, seg.total_num_tokens, seg.alive)


def oclause(c):
    return (c.term, c.occur, c.mode, c.boost, None if c.term_set is None else list(c.term_set))


def test_term_set_clauses_match_oracle(orc, corpus):
    """FuzzyTermQuery = ConstScorer over the union of its terms' documents (fuzzy_query.rs:90-125), under the shape of
    the paragraph fuzzy query (search_query.rs:200-240): Must-grouped fuzzy Shoulds, Must filters, boost 0.5."""
    seg, vocab = corpus
    rng = np.random.default_rng(5)
    s = Bm25Searcher.open([seg])
    queries = []
    for _ in range(48):
        q = []
        for _ in range(int(rng.integers(1, 4))):
            members = rng.integers(0, vocab, int(rng.integers(1, 40))).tolist()
            if rng.random() < 0.3:
                members += rng.integers(0, 30, 3).tolist()  # dense lists inside the union
            q.append(Clause(0, G, CONST, 0.5, term_set=members))
        if rng.random() < 0.7:
            q.append(Clause(int(rng.integers(0, 20)), M, BASIC, 0.5))
        if rng.random() < 0.3:
            q.append(Clause(int(rng.integers(0, 200)), N))
        queries.append(q)
    queries.append([Clause(0, S, CONST, 1.0, term_set=[5, 5, 5])])              # the same term three times: counted once
    queries.append([Clause(0, M, CONST, 2.0, term_set=[7]), Clause(7, S, FREQ)])  # a set next to a plain clause
    oidx = oracle_index(orc, seg)
    try:
        for k in (20, 1):
            r = s.search_batch_ex(queries, k)
            for i, q in enumerate(queries):
                wd, ws, _, wt, _ = oidx.search_ex([oclause(c) for c in q], k)
                assert r["total"][i] == wt, (i, r["total"][i], wt)
                n = int(r["count"][i])
                assert n == len(wd)
                assert np.array_equal(r["docaddr"][i, :n], wd), (i, r["docaddr"][i, :n], wd)
                assert np.array_equal(bits(r["score"][i, :n]), bits(ws)), (i, r["score"][i, :n], ws)
    finally:
        s.close()


def test_order_by_fast_field_and_facets_match_oracle(orc, corpus):
    seg, vocab = corpus
    rng = np.random.default_rng(11)
    n = seg.n_docs
    created = rng.integers(1_600_000_000, 1_600_000_000 + 5000, n).astype(np.int64)   # many ties
    modified = created + rng.integers(0, 10**6, n)
    alive = orc.bitset(n, ones=np.nonzero(rng.random(n) < 0.8)[0].tolist())
    seg2 = Bm25Segment(seg.term_offsets, seg.doc_ids, seg.tfs, seg.fieldnorm_ids, seg.total_num_tokens, alive=alive)
    s = Bm25Searcher.open([seg2])
    oidx = oracle_index(orc, seg2)
    queries = [[Clause(int(t), S, BASIC) for t in rng.integers(0, 80, 3)] for _ in range(20)]
    queries.append([Clause(3, M, BASIC), Clause(4, M, BASIC)])
    queries.append([])
    facet_terms = [rng.integers(0, vocab, int(rng.integers(0, 12))).tolist() for _ in queries]
    facet_terms[1] = []                # a query without facets in a batch with facets
    facet_terms[2] = [0, 1, 2, 0]      # dense facets, one asked twice
    try:
        s.set_fast_field(0, 0, created)
        s.set_fast_field(0, 1, modified)
        for field, vals in ((0, created), (1, modified)):
            for desc in (True, False):
                r = s.search_batch_ex(queries, 20, order_field=field, order_desc=desc, facets=facet_terms)
                for i, q in enumerate(queries):
                    wd, _, wv, wt, mb = oidx.search_ex([oclause(c) for c in q], 20, order_values=vals, order_desc=desc, want_match_bits=True)
                    cnt = int(r["count"][i])
                    assert r["total"][i] == wt and cnt == len(wd)
                    assert np.array_equal(r["docaddr"][i, :cnt], wd), (field, desc, i)
                    assert np.array_equal(r["order_value"][i, :cnt], wv)
                    match = np.unpackbits(mb.view(np.uint8), bitorder="little")[:n].astype(bool)
                    want = [int(match[seg.doc_ids[int(seg.term_offsets[t]): int(seg.term_offsets[t + 1])]].sum()) for t in facet_terms[i]]
                    assert r["facet_counts"][i].tolist() == want, (i, r["facet_counts"][i], want)
        # facets only (only_faceted, reader.rs:403-410): k = 0
        r = s.search_batch_ex(queries, 0, facets=facet_terms)
        for i, q in enumerate(queries):
            _, _, _, wt, mb = oidx.search_ex([oclause(c) for c in q], 0, want_match_bits=True)
            match = np.unpackbits(mb.view(np.uint8), bitorder="little")[:n].astype(bool)
            want = [int(match[seg.doc_ids[int(seg.term_offsets[t]): int(seg.term_offsets[t + 1])]].sum()) for t in facet_terms[i]]
            assert r["total"][i] == wt and r["count"][i] == 0 and r["facet_counts"][i].tolist() == want
    finally:
        s.close()


def test_two_segments_merge_by_fast_field(orc):
    rng = np.random.default_rng(21)
    vocab = 500
    segs, vals = [], []
    for n in (9000, 4000):
        segs.append(Bm25Segment.from_term_docs(zipf_corpus(rng, n, vocab, mean_len=10), vocab))
        vals.append(rng.integers(0, 300, n).astype(np.int64))
    s = Bm25Searcher.open(segs)
    queries = [[Clause(int(t), S, BASIC) for t in rng.integers(0, 40, 2)] for _ in range(10)]
    try:
        for i, v in enumerate(vals):
            s.set_fast_field(i, 0, v)
        r = s.search_batch_ex(queries, 15, order_field=0, order_desc=True)
        for qi, q in enumerate(queries):
            hits = []
            for si, seg in enumerate(segs):
                # statistics are searcher-wide (BM25 idf), but the order by a fast field does not depend on scores
                wd, _, wv, wt, _ = oracle_index(orc, seg).search_ex([oclause(c) for c in q], 15, segment_ord=si, order_values=vals[si], order_desc=True)
                hits += list(zip((-wv).tolist(), wd.tolist()))
            hits.sort()
            cnt = int(r["count"][qi])
            assert cnt == min(15, len(hits))
            assert r["docaddr"][qi, :cnt].tolist() == [h[1] for h in hits[:cnt]]
            assert r["order_value"][qi, :cnt].tolist() == [-h[0] for h in hits[:cnt]]
    finally:
        s.close()


def test_phrase_clauses_match_oracle(orc):
    """PhraseQuery with slop 0 (tantivy PhraseWeight / PhraseScorer restated; nidx_paragraph keyword_parser.rs:69-91, nidx_text's
    QueryParser): consecutive positions in order, tf = occurrences, Bm25Weight::for_terms (idf summed over the terms)."""
    rng = np.random.default_rng(41)
    vocab = 40                     # small vocabulary: phrases of 2-4 terms really occur
    docs = [rng.integers(0, vocab, int(rng.integers(3, 40))) for _ in range(6000)]
    docs[10] = np.array([1, 2, 3, 1, 2, 3, 1, 2], np.int64)    # repeated phrase: tf 2 / 3
    docs[11] = np.array([5, 5, 5, 5], np.int64)                # the same term at consecutive positions
    seg = Bm25Segment.from_term_docs(docs, vocab, with_positions=True)
    s = Bm25Searcher.open([seg])
    oidx = orc.Bm25Index(seg.term_offsets, seg.doc_ids, seg.tfs, seg.fieldnorm_ids, seg.total_num_tokens, seg.alive, seg.pos_offsets, seg.positions)
    queries = []
    for _ in range(40):
        m = int(rng.integers(2, 5))
        q = [Clause(0, int(rng.choice([S, M, G])), FREQ, float(rng.choice([1.0, 0.5])), term_set=rng.integers(0, vocab, m).tolist(), phrase=True)]
        if rng.random() < 0.5:
            q.append(Clause(int(rng.integers(0, vocab)), S, BASIC))
        if rng.random() < 0.3:
            q.append(Clause(0, N, FREQ, 1.0, term_set=rng.integers(0, vocab, 2).tolist(), phrase=True))
        queries.append(q)
    queries.append([Clause(0, M, FREQ, 1.0, term_set=[1, 2, 3], phrase=True)])
    queries.append([Clause(0, M, FREQ, 1.0, term_set=[5, 5], phrase=True)])
    queries.append([Clause(0, M, FREQ, 1.0, term_set=[1, 2, 3, 1, 2, 3, 1, 2], phrase=True)])   # 8 terms: the maximum
    try:
        r = s.search_batch_ex(queries, 20)
        for i, q in enumerate(queries):
            oc = [(c.term, c.occur, c.mode, c.boost, None if c.term_set is None else list(c.term_set), False, c.phrase) for c in q]
            wd, ws, _, wt, _ = oidx.search_ex(oc, 20)
            n = int(r["count"][i])
            assert r["total"][i] == wt, (i, r["total"][i], wt)
            assert n == len(wd) and np.array_equal(r["docaddr"][i, :n], wd), (i, r["docaddr"][i, :n], wd)
            assert np.array_equal(bits(r["score"][i, :n]), bits(ws)), (i, r["score"][i, :n], ws)
        assert r["total"][-3] >= 1 and 10 in (r["docaddr"][-3, : r["count"][-3]] & 0xFFFFFFFF).tolist()
    finally:
        s.close()


def random_filter_program(rng, vocab, n_ranges, n_phrases, depth=0):
    """A random boolean expression in postfix form: ([(op, a, b)], [term ids])."""
    ops, lists = [], []

    def leaf():
        kind = rng.random()
        if kind < 0.55:
            m = int(rng.integers(0, 4))  # 0 terms: the empty union
            ops.append((_lib.FILTER_PUSH_LISTS, len(lists), len(lists) + m))
            lists.extend(int(t) for t in rng.integers(0, vocab, m))
        elif kind < 0.75 and n_ranges:
            ops.append((_lib.FILTER_PUSH_RANGE, int(rng.integers(0, n_ranges)), 0))
        elif kind < 0.9 and n_phrases:
            ops.append((_lib.FILTER_PUSH_PHRASE, int(rng.integers(0, n_phrases)), 0))
        else:
            ops.append((_lib.FILTER_PUSH_ALL if rng.random() < 0.5 else _lib.FILTER_PUSH_NONE, 0, 0))

    def expr(d):
        if d >= 3 or rng.random() < 0.3:
            leaf()
        else:
            r = rng.random()
            if r < 0.25:
                expr(d + 1)
                ops.append((_lib.FILTER_NOT, 0, 0))
            else:
                n = int(rng.integers(2, 4))
                for i in range(n):
                    expr(d + 1)
                    if i:
                        ops.append((_lib.FILTER_AND if r < 0.65 else _lib.FILTER_OR, 0, 0))

    expr(depth)
    return ops, lists


def test_prefilter_matches_oracle(orc):
    """TextReaderService::prefilter (nidx_text/src/reader.rs:148-180): nested And / Or / Not over term, date-range and phrase
    leaves, two segments with deletions — the device's bitset algebra vs the oracle's document-at-a-time evaluation."""
    rng = np.random.default_rng(2024)
    vocab = 60
    segs, oidx, fast = [], [], []
    for n_docs in (20011, 777):
        docs = [rng.integers(0, vocab, int(rng.integers(1, 30))) for _ in range(n_docs)]
        alive = rng.random(n_docs) > 0.1
        words = np.zeros((n_docs + 63) // 64, np.uint64)
        for d in np.flatnonzero(alive):
            words[d >> 6] |= np.uint64(1) << np.uint64(d & 63)
        seg = Bm25Segment.from_term_docs(docs, vocab, alive=words, with_positions=True)
        segs.append(seg)
        oidx.append(orc.Bm25Index(seg.term_offsets, seg.doc_ids, seg.tfs, seg.fieldnorm_ids, seg.total_num_tokens, seg.alive, seg.pos_offsets, seg.positions))
        fast.append((rng.integers(1000, 1200, n_docs), rng.integers(-50, 50, n_docs)))
    s = Bm25Searcher.open(segs)
    try:
        for i, (cr, mo) in enumerate(fast):
            s.set_fast_field(i, 0, cr)
            s.set_fast_field(i, 1, mo)
        ranges = [(0, 1050, 1100), (0, 1100, None), (1, None, 0), (1, 10, 10), (0, 5000, None), (1, None, None), (0, 1100, 1050), (1, -50, 49)]
        phrases = [rng.integers(0, vocab, int(rng.integers(2, 4))).tolist() for _ in range(6)] + [[3, 3]]
        live_want = None
        for trial in range(60):
            ops, lists = random_filter_program(rng, vocab, len(ranges), len(phrases))
            got, live = s.prefilter(ops, lists, ranges, phrases)
            want, lw = [], 0
            for i, oi in enumerate(oidx):
                d, l = oi.prefilter(ops, lists, ranges, fast[i][0], fast[i][1], phrases)
                want.append((np.uint64(i) << np.uint64(32)) | d.astype(np.uint64))
                lw += l
            want = np.concatenate(want)
            assert live == lw
            assert np.array_equal(got, want), (trial, ops, got.size, want.size)
            live_want = lw
        # no expression at all: every live document; a capacity smaller than the result still reports the full count
        got, live = s.prefilter([], [])
        assert got.size == live == live_want
        import ctypes as C
        req = _lib.Bm25PrefilterC()
        out = np.zeros(10, np.uint64)
        n, lv = C.c_uint64(0), C.c_uint64(0)
        _lib.check(_lib.lib().nidx_gpu_bm25_prefilter(s._handle, C.byref(req), out.ctypes.data, 10, C.byref(n), C.byref(lv)))
        assert n.value == live_want and np.array_equal(out, got[:10])
        # errors: stack underflow, unknown range, out-of-range term
        for bad in ([(_lib.FILTER_AND, 0, 0)], [(_lib.FILTER_PUSH_RANGE, 99, 0)], [(_lib.FILTER_PUSH_ALL, 0, 0), (_lib.FILTER_PUSH_ALL, 0, 0)]):
            with pytest.raises(_lib.NidxGpuError):
                s.prefilter(bad, [], ranges, phrases)
        with pytest.raises(_lib.NidxGpuError):
            s.prefilter([(_lib.FILTER_PUSH_LISTS, 0, 1)], [vocab + 5])
    finally:
        s.close()
