"""Rank fusion mirror vs the cases of the reference's own unit tests (nucliadb/tests/search/unit/test_rank_fusion.py:
test_reciprocal_rank_fusion_algorithm :150-312, _boosting :315-404, test_weighted_comb_sum_rank_fusion :409-455; RRF_TEST_K = 2,
FAKE_GRAPH_SCORE = 1.0): same inputs, same (id, rounded score, score type) sequences."""
from nucliadb_amd.rank_fusion import BM25, BOTH, RELATION_RELEVANCE, VECTOR, ReciprocalRankFusion, ScoredItem, WeightedCombSum

K = 2


def kw(score, pid):
    return ScoredItem(pid, score, BM25)


def sem(score, pid):
    return ScoredItem(pid, score, VECTOR)


def graph(pid):
    return ScoredItem(pid, 1.0, RELATION_RELEVANCE)


def run(algo, keyword, semantic, graph_):
    merged = algo.fuse({"keyword": keyword, "semantic": semantic, "graph": graph_})
    return [(x.paragraph_id, round(x.score, 6), x.score_type) for x in merged]


def test_rrf_mix_of_sources():
    got = run(ReciprocalRankFusion(k=K, window=20),
              [kw(0.1, "k-1"), kw(0.5, "k-2"), kw(0.3, "k-3")],
              [sem(0.2, "s-1"), sem(0.3, "s-2"), sem(0.6, "s-3"), sem(0.4, "s-4")],
              [graph("g-1"), graph("g-2")])
    r = lambda i: round(1 / (i + K), 6)
    assert got == [("k-2", r(0), BM25), ("s-3", r(0), VECTOR), ("g-1", r(0), RELATION_RELEVANCE), ("k-3", r(1), BM25), ("s-4", r(1), VECTOR),
                   ("g-2", r(1), RELATION_RELEVANCE), ("k-1", r(2), BM25), ("s-2", r(2), VECTOR), ("s-1", r(3), VECTOR)]


def test_single_source_keeps_its_scores():
    assert run(ReciprocalRankFusion(k=K, window=20), [kw(1, "k-1"), kw(3, "k-2"), kw(4, "k-3")], [], []) == [
        ("k-3", 4.0, BM25), ("k-2", 3.0, BM25), ("k-1", 1.0, BM25)]
    assert run(ReciprocalRankFusion(k=K, window=20), [], [sem(0.2, "s-1"), sem(0.3, "s-2"), sem(0.6, "s-3"), sem(0.4, "s-4")], []) == [
        ("s-3", 0.6, VECTOR), ("s-4", 0.4, VECTOR), ("s-2", 0.3, VECTOR), ("s-1", 0.2, VECTOR)]
    assert run(ReciprocalRankFusion(k=K, window=20), [], [], [graph("g-1"), graph("g-2")]) == [
        ("g-1", 1.0, RELATION_RELEVANCE), ("g-2", 1.0, RELATION_RELEVANCE)]


def test_rrf_ties_and_multi_match():
    r = lambda i: 1 / (i + K)
    got = run(ReciprocalRankFusion(k=K, window=20),
              [kw(0.1, "k-1"), kw(0.5, "k-2"), kw(0.3, "k-3"), kw(0.6, "k-4"), kw(0.6, "k-5")],
              [sem(2, "s-1"), sem(3, "s-2"), sem(6, "s-3")], [])
    assert [x[0] for x in got] == ["k-4", "s-3", "k-5", "s-2", "k-2", "s-1", "k-3", "k-1"]
    got = run(ReciprocalRankFusion(k=K, window=20),
              [kw(0.1, "r-1"), kw(0.5, "r-2"), kw(0.3, "r-4")],
              [sem(2, "r-1"), sem(3, "r-3"), sem(6, "r-4"), sem(6, "r-5")], [])
    assert got == [("r-4", round(r(1) + r(0), 6), BOTH), ("r-2", round(r(0), 6), BM25), ("r-1", round(r(2) + r(3), 6), BOTH),
                   ("r-5", round(r(1), 6), VECTOR), ("r-3", round(r(2), 6), VECTOR)]


def test_rrf_boosting():
    r = lambda i: 1 / (i + K)
    got = run(ReciprocalRankFusion(k=K, window=20, weights={"keyword": 2, "semantic": 0.5}, default_weight=1.0),
              [kw(0.1, "r-1"), kw(0.5, "r-2"), kw(0.3, "r-4")],
              [sem(2, "r-1"), sem(3, "r-3"), sem(6, "r-4"), sem(6, "r-5")],
              [graph("r-1"), graph("r-6")])
    assert got == [("r-1", round(r(2) * 2 + r(3) * 0.5 + r(0) * 1.0, 6), BOTH), ("r-2", round(r(0) * 2, 6), BM25),
                   ("r-4", round(r(1) * 2 + r(0) * 0.5, 6), BOTH), ("r-6", round(r(1) * 1.0, 6), RELATION_RELEVANCE),
                   ("r-5", round(r(1) * 0.5, 6), VECTOR), ("r-3", round(r(2) * 0.5, 6), VECTOR)]


def test_weighted_comb_sum():
    got = run(WeightedCombSum(window=20, weights={"keyword": 2, "semantic": 0.5}, default_weight=1.5),
              [kw(0.1, "r-1"), kw(0.5, "r-2"), kw(0.3, "r-4")],
              [sem(2, "r-1"), sem(3, "r-3"), sem(6, "r-4"), sem(6, "r-5")],
              [graph("r-1"), graph("r-6")])
    assert got == [("r-4", round(0.3 * 2.0 + 6 * 0.5, 6), BOTH), ("r-5", round(6 * 0.5, 6), VECTOR),
                   ("r-1", round(0.1 * 2.0 + 2 * 0.5 + 1.0 * 1.5, 6), BOTH), ("r-3", round(3 * 0.5, 6), VECTOR),
                   ("r-6", round(1.0 * 1.5, 6), RELATION_RELEVANCE), ("r-2", round(0.5 * 2.0, 6), BM25)]
