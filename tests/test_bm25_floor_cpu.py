"""The REASONING behind the streaming BM25 scorer's per-query score floor (csrc/bm25_aux.hip: bm25_term_floor_kernel, csrc/bm25_index.cpp:
the floor of a query, csrc/bm25_stream.hip: the bar starts at it), checked against the oracle on the CPU.

For a query of Should term clauses the library takes, per clause, the smallest fieldnorm id f such that at least R >= k of the term's
(first 65 536) postings belong to documents of fieldnorm id <= f, and claims that at least k documents of the query score
>= weight * quotient(tf = 1, f).  Postings strictly below the largest such bound never become candidates — so the claim must hold
in f32 exactly as the kernels compute scores, for every k the rank grid serves, and a posting AT the bound must survive.  The model
below restates the host's arithmetic (numpy f32); the oracle is tantivy's scorer restated (oracle/nidx_oracle.c)."""
import numpy as np
import pytest

from nucliadb_amd import _lib
from nucliadb_amd.bm25 import Bm25Segment

RANKS = [1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256, 384, 512]   # csrc/kernels.h: BM25_FLOOR_RANKS
K1, B = np.float32(1.2), np.float32(0.75)
FREQ, BASIC = _lib.TF_FREQ, _lib.TF_BASIC


def corpus(rng, n_docs, vocab, mean_len):
    lens = np.clip(np.round(rng.lognormal(np.log(mean_len), 0.6, n_docs)), 2, 400).astype(np.int64)
    p = 1.0 / np.arange(1, vocab + 1)
    p /= p.sum()
    flat = rng.choice(vocab, size=int(lens.sum()), p=p)
    return np.split(flat, np.cumsum(lens)[:-1])


def quot1_table(orc, seg):
    """1 / (1 + K1 (1 - B + B fieldnorm / avg)) per fieldnorm id, in the two-step f32 arithmetic of bm25_index.cpp"""
    table = orc.fieldnorm_table().astype(np.float32)
    avg = np.float32(seg.total_num_tokens) / np.float32(seg.n_docs)
    kcache = (K1 * (np.float32(1.0) - B + B * table / avg)).astype(np.float32)
    return (np.float32(1.0) / (np.float32(1.0) + kcache)).astype(np.float32)


def floor_of(orc, seg, q1, clauses, k, cap=1 << 16):
    """the host's floor for a query of (term, mode, boost) Should clauses; -inf = none"""
    j = next((i for i, r in enumerate(RANKS) if r >= k), None)
    if j is None:
        return -np.inf
    best = -np.inf
    for term, _mode, boost in clauses:
        b, e = int(seg.term_offsets[term]), int(seg.term_offsets[term + 1])
        df = e - b
        e = min(e, b + cap)
        fn = np.sort(seg.fieldnorm_ids[seg.doc_ids[b:e]])
        if len(fn) < RANKS[j]:
            continue
        f = int(fn[RANKS[j] - 1])          # smallest id with >= RANKS[j] postings at or under it
        if f == 255:
            continue
        w = np.float32(np.float32(orc.bm25_idf(df, seg.n_docs)) * (np.float32(1.0) + K1) * np.float32(boost))
        best = max(best, float(np.float32(w * q1[f])))
    return best


@pytest.mark.parametrize("seed,mean_len", [(1, 6), (2, 24), (3, 60)])
def test_the_floor_is_never_above_the_kth_best_score(orc, seed, mean_len):
    rng = np.random.default_rng(seed)
    vocab = 600
    docs = corpus(rng, 12000, vocab, mean_len)
    seg = Bm25Segment.from_term_docs(docs, vocab)
    q1 = quot1_table(orc, seg)
    assert np.all(np.diff(q1) <= 0)   # the monotonicity the library checks at open
    oidx = orc.Bm25Index(seg.term_offsets, seg.doc_ids, seg.tfs, seg.fieldnorm_ids, seg.total_num_tokens, seg.alive)
    tight = 0
    for _ in range(120):
        n = int(rng.integers(1, 5))
        clauses = [(int(rng.integers(0, vocab)), int(rng.choice([FREQ, BASIC])), float(rng.choice([1.0, 0.5, 3.0]))) for _ in range(n)]
        for k in (1, 5, 20, 33, 64, 201, 501):
            f = floor_of(orc, seg, q1, clauses, k)
            _d, s, _t = oidx.search([(t, _lib.OCCUR_SHOULD, m, b) for t, m, b in clauses], k)
            if f == -np.inf:
                continue
            assert len(s) == k, "a floor promises k documents"
            assert np.float32(f) <= np.float32(s[k - 1]), (clauses, k, f, s[k - 1])
            tight += np.float32(f) >= np.float32(0.8) * np.float32(s[k - 1])
    assert tight > 100   # (and it is a useful bound, not a trivial one)


def test_documents_of_one_length_tie_with_the_floor(orc):
    """every document the same length and tf = 1: all postings of a term score alike, the floor EQUALS the k-th best score — the scorer
    keeps postings at the floor (`!(score < floor)`), and the oracle's k best are the lowest doc ids"""
    rng = np.random.default_rng(9)
    vocab = 40
    docs = [rng.choice(vocab, size=8, replace=False) for _ in range(3000)]
    seg = Bm25Segment.from_term_docs(docs, vocab)
    q1 = quot1_table(orc, seg)
    oidx = orc.Bm25Index(seg.term_offsets, seg.doc_ids, seg.tfs, seg.fieldnorm_ids, seg.total_num_tokens, seg.alive)
    for term in (0, 7, 39):
        f = floor_of(orc, seg, q1, [(term, BASIC, 1.0)], 20)
        d, s, _t = oidx.search([(term, _lib.OCCUR_SHOULD, BASIC, 1.0)], 20)
        assert np.float32(f).view(np.uint32) == np.float32(s[19]).view(np.uint32)
        b = int(seg.term_offsets[term])
        assert np.array_equal(d & 0xFFFFFFFF, seg.doc_ids[b : b + 20])
