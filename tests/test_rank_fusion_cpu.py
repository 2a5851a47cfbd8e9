"""Rank fusion mirror vs the cases of the reference's own unit tests (nucliadb/tests/search/unit/test_rank_fusion.py:
test_reciprocal_rank_fusion_algorithm :150-312, _boosting :315-404, test_weighted_comb_sum_rank_fusion :409-455; RRF_TEST_K = 2,
FAKE_GRAPH_SCORE = 1.0): same inputs, same (id, rounded score, score type) sequences."""
import numpy as np
import pytest

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


def test_native_batched_rrf_equals_the_mirror():
    """nidx_gpu_rank_fusion_rrf (host C++, batched) against ReciprocalRankFusion on random ranked lists: overlapping ids, weights,
    empty lists (single-source rule), windows shorter than the union — same ids, same f64 scores, same order."""
    import numpy as np

    from nucliadb_amd.rank_fusion import rrf_fuse_batch

    rng = np.random.default_rng(12)
    B, sa, sb, sc, window = 300, 20, 10, 4, 12
    def make(stride, lo):
        ids = np.zeros((B, stride), np.uint64)
        scores = np.zeros((B, stride), np.float32)
        counts = rng.integers(lo, stride + 1, B).astype(np.uint32)
        for q in range(B):
            ids[q, : counts[q]] = rng.choice(40, counts[q], replace=False)          # small id space: lists overlap
            scores[q, : counts[q]] = np.sort(rng.random(counts[q]).astype(np.float32))[::-1]
        return ids, counts, scores
    a, b, c = make(sa, 0), make(sb, 0), make(sc, 0)
    b[1][:40] = 0                                                                     # many single-source queries
    c[1][:200] = 0
    weights = {"keyword": 1.0, "semantic": 2.5, "graph": 0.5}
    got_ids, got_scores, got_counts = rrf_fuse_batch([(a[0], a[1], weights["keyword"], a[2]), (b[0], b[1], weights["semantic"], b[2]),
                                                      (c[0], c[1], weights["graph"], c[2])], k=K, window=window)
    algo = ReciprocalRankFusion(k=K, window=window, weights=weights)
    for q in range(B):
        src = {"keyword": [ScoredItem(str(int(i)), float(s), BM25) for i, s in zip(a[0][q, : a[1][q]], a[2][q])],
               "semantic": [ScoredItem(str(int(i)), float(s), VECTOR) for i, s in zip(b[0][q, : b[1][q]], b[2][q])],
               "graph": [ScoredItem(str(int(i)), float(s), RELATION_RELEVANCE) for i, s in zip(c[0][q, : c[1][q]], c[2][q])]}
        if not any(src.values()):
            assert got_counts[q] == 0
            continue
        want = algo.fuse(src)[:window]
        n = int(got_counts[q])
        assert n == len(want), q
        assert [int(x) for x in got_ids[q, :n]] == [int(w.paragraph_id) for w in want], q
        assert [float(x) for x in got_scores[q, :n]] == [float(w.score) for w in want], q


def test_rrf_ranks_scored_lists_by_score_and_dedups_large_windows():
    """Ranks are positions AFTER `sorted(values, key=score, reverse=True)` (rank_fusion.py:139-147: every source is sorted first,
    stably): a list that carries scores in another order — BM25 hits ordered by a date field — is ranked by them, not refused
    (ADVICE r02); windows of 500 hits per source are merged through a hash map."""
    from nucliadb_amd.rank_fusion import rrf_fuse_batch

    ids = np.array([[5, 6, 7, 8]], np.uint64)
    cnt = np.array([4], np.uint32)
    unsorted = np.array([[1.0, 3.0, 2.0, 3.0]], np.float32)    # ranks by score, stable: 6, 8, 7, 5
    other = np.array([[9, 5, 6, 0]], np.uint64)
    f_ids, f_scores, n = rrf_fuse_batch([(ids, cnt, 1.0, unsorted), (other, np.array([3], np.uint32), 2.0, None)], k=60.0, window=8)
    want = {}
    for r, i in enumerate([6, 8, 7, 5]):
        want[i] = want.get(i, 0.0) + 1 / (60.0 + r) * 1.0
    for r, i in enumerate([9, 5, 6]):
        want[i] = want.get(i, 0.0) + 1 / (60.0 + r) * 2.0
    first_seen = [6, 8, 7, 5, 9]
    order = sorted(first_seen, key=lambda i: -want[i])          # stable: ties keep first-seen order
    assert n[0] == 5 and [int(i) for i in f_ids[0, :5]] == order
    assert [float(s) for s in f_scores[0, :5]] == [want[i] for i in order]
    rng = np.random.default_rng(5)
    a = rng.permutation(2000)[:500].astype(np.uint64)[None, :]
    b = rng.permutation(2000)[:500].astype(np.uint64)[None, :]
    c500 = np.array([500], np.uint32)
    fused_ids, fused_scores, n = rrf_fuse_batch([(a, c500, 1.0, None), (b, c500, 2.0, None)], k=60.0, window=1000)
    want = {}
    for r, i in enumerate(a[0]):
        want[int(i)] = want.get(int(i), 0.0) + 1.0 / (60.0 + r)
    for r, i in enumerate(b[0]):
        want[int(i)] = want.get(int(i), 0.0) + 2.0 / (60.0 + r)
    assert n[0] == len(want)
    got = {int(i): float(s) for i, s in zip(fused_ids[0, : n[0]], fused_scores[0, : n[0]])}
    assert got.keys() == want.keys() and all(abs(got[i] - want[i]) < 1e-12 for i in want)
    assert list(fused_scores[0, : n[0]]) == sorted(fused_scores[0, : n[0]], reverse=True)


def _reference_cases():
    import gzip
    import json
    import os

    with gzip.open(os.path.join(os.path.dirname(__file__), "golden", "rank_fusion_reference.json.gz"), "rt", encoding="utf-8") as f:
        return json.load(f)["cases"]


def test_mirror_matches_outputs_of_the_reference_module():
    """tests/golden/rank_fusion_reference.json.gz holds outputs of the REFERENCE's rank_fusion.py itself (imported from /root/reference by
    scripts/make_reference_golden.py) on seeded random cases: three sources, overlapping ids, ties, weights, empty sources.  The
    mirror must give the same order, the same f64 score bits, the same score types and the same score history per hit."""
    from nucliadb_amd.rank_fusion import BM25, ReciprocalRankFusion, ScoredItem, WeightedCombSum

    cases = _reference_cases()
    assert len(cases) >= 140
    seen = {"rrf": 0, "wcombsum": 0, "single": 0, "both": 0}
    for n, c in enumerate(cases):
        if c["algorithm"] == "rrf":
            algo = ReciprocalRankFusion(k=c["k"], window=20, weights=c["weights"], default_weight=c["default_weight"])
        else:
            algo = WeightedCombSum(window=20, weights=c["weights"], default_weight=c["default_weight"])
        src = {name: [ScoredItem(pid, float.fromhex(s), kind) for pid, s, kind in hits] for name, hits in c["sources"].items()}
        got = algo.fuse(src)
        assert [(g.paragraph_id, float(g.score).hex(), g.score_type, [float(h).hex() for h in g.history]) for g in got] == [
            (pid, s, kind, hist) for pid, s, kind, hist in c["expected"]], n
        seen[c["algorithm"]] += 1
        seen["single"] += sum(1 for h in c["sources"].values() if h) == 1
        seen["both"] += any(e[2] == "BOTH" for e in c["expected"])
    assert min(seen.values()) > 0, seen


def test_native_batched_rrf_matches_outputs_of_the_reference_module():
    """The same reference outputs against nidx_gpu_rank_fusion_rrf: every source pre-sorted by score (stable, as
    rank_fusion.py:139-147 does) and handed over as ranked id lists."""
    from nucliadb_amd.rank_fusion import rrf_fuse_batch

    n_checked = 0
    for n, c in enumerate(_reference_cases()):
        if c["algorithm"] != "rrf":
            continue
        ids_of, lists = {}, []
        for name in ("keyword", "semantic", "graph"):
            hits = sorted(c["sources"][name], key=lambda h: float.fromhex(h[1]), reverse=True)
            row = np.zeros((1, 32), np.uint64)
            for j, (pid, _, _) in enumerate(hits):
                row[0, j] = ids_of.setdefault(pid, len(ids_of) + 1)
            lists.append((row, np.array([len(hits)], np.uint32), c["weights"].get(name, c["default_weight"]), None))
        if sum(1 for l in lists if l[1][0]) < 2:
            continue   # a single source keeps its own scores: nothing to fuse
        got_ids, got_scores, got_counts = rrf_fuse_batch(lists, k=c["k"], window=64)
        name_of = {v: k for k, v in ids_of.items()}
        m = int(got_counts[0])
        assert [(name_of[int(i)], float(s).hex()) for i, s in zip(got_ids[0, :m], got_scores[0, :m])] == [(e[0], e[1]) for e in c["expected"]], n
        n_checked += 1
    assert n_checked >= 20


def test_native_batched_wcombsum_matches_outputs_of_the_reference_module():
    """nidx_gpu_rank_fusion_wcombsum against the reference module's WeightedCombSum outputs: the cases of the fixture whose
    scores an f32 holds exactly (the product's lists carry f32 scores), hits in the order the retriever gave them."""
    from nucliadb_amd.rank_fusion import wcombsum_fuse_batch

    n_checked = 0
    for n, c in enumerate(_reference_cases()):
        if c["algorithm"] != "wcombsum" or not c.get("f32_scores"):
            continue
        ids_of, lists = {}, []
        for name in ("keyword", "semantic", "graph"):
            hits = c["sources"][name]
            row = np.zeros((1, 32), np.uint64)
            sc = np.zeros((1, 32), np.float32)
            for j, (pid, s, _) in enumerate(hits):
                row[0, j] = ids_of.setdefault(pid, len(ids_of) + 1)
                sc[0, j] = np.float32(float.fromhex(s))
                assert float(sc[0, j]) == float.fromhex(s)
            lists.append((row, np.array([len(hits)], np.uint32), c["weights"].get(name, c["default_weight"]), sc))
        got_ids, got_scores, got_counts = wcombsum_fuse_batch(lists, window=64)
        name_of = {v: k for k, v in ids_of.items()}
        m = int(got_counts[0])
        assert [(name_of[int(i)], float(s).hex()) for i, s in zip(got_ids[0, :m], got_scores[0, :m])] == [(e[0], e[1]) for e in c["expected"]], n
        n_checked += 1
    assert n_checked >= 40


def test_native_rrf_is_thread_safe_and_deterministic():
    """The native routine spreads a batch over a persistent host-thread pool (one job at a time): concurrent callers with batches of
    different sizes must each get exactly what a lone caller gets."""
    import threading

    from nucliadb_amd.rank_fusion import rrf_fuse_batch

    rng = np.random.default_rng(77)

    def make(B):
        a = np.stack([rng.choice(500, 20, replace=False) for _ in range(B)]).astype(np.uint64)
        b = np.stack([rng.choice(500, 10, replace=False) for _ in range(B)]).astype(np.uint64)
        sa = np.sort(rng.random((B, 20)).astype(np.float32), axis=1)[:, ::-1].copy()
        return [(a, np.full(B, 20, np.uint32), 1.0, sa), (b, np.full(B, 10, np.uint32), 2.0, None)]

    jobs = [make(B) for B in (1, 63, 64, 65, 300, 1024, 2048, 130)]
    want = [rrf_fuse_batch(j, k=60.0, window=16) for j in jobs]
    errors = []

    def worker(i):
        for _ in range(40):
            got = rrf_fuse_batch(jobs[i], k=60.0, window=16)
            if not all(np.array_equal(g, w) for g, w in zip(got, want[i])):
                errors.append(i)
                return

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(len(jobs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors
