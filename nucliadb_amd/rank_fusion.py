"""Host mirror of nucliadb's rank fusion of the BM25 and vector lists:
`nucliadb/src/nucliadb/search/search/rank_fusion.py:60-254` — windows of a few hundred hits, so it stays on the host; the lists
it fuses come from the HIP kernels (ParagraphSearcher / VectorSearcher).  The classes below mirror the reference's objects (score
types, per-hit score history); rrf_fuse_batch / wcombsum_fuse_batch run whole batches through the library's native routines.

  fuse()                 one non-empty source => its hits unchanged, else the algorithm; then sort by score desc (:74-91)
  ReciprocalRankFusion   score(d) = sum over retrievers of weight(r) / (k + rank_r(d)), ranks from each list sorted by score
                         desc (stable), hits deduplicated by paragraph id, BM25 + VECTOR => BOTH (:106-181)
  WeightedCombSum        score(d) = sum over retrievers of weight(r) * score_r(d), the first occurrence is kept (:184-252)
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional

BM25, VECTOR, BOTH, RELATION_RELEVANCE = "BM25", "VECTOR", "BOTH", "RELATION_RELEVANCE"


@dataclass
class ScoredItem:
    paragraph_id: str
    score: float
    score_type: str
    history: List[float] = field(default_factory=list)  # ScoredTextBlock.scores: every score the hit carried

    def __post_init__(self):
        if not self.history:
            self.history = [self.score]


def _both(a: str, b: str) -> bool:
    return {a, b} == {BM25, VECTOR}


class RankFusionAlgorithm:
    def __init__(self, window: int):
        self.window = window

    def fuse(self, sources: Dict[str, List[ScoredItem]]) -> List[ScoredItem]:
        non_empty = [v for v in sources.values() if v]
        merged = list(non_empty[0]) if len(non_empty) == 1 else self._fuse(sources)
        merged.sort(key=lambda r: r.score, reverse=True)  # stable: ties keep the order the algorithm produced
        return merged

    def _fuse(self, sources):
        raise NotImplementedError


class ReciprocalRankFusion(RankFusionAlgorithm):
    def __init__(self, k: float = 60.0, *, window: int, weights: Optional[Dict[str, float]] = None, default_weight: float = 1.0):
        super().__init__(window)
        self.k, self.weights, self.default_weight = k, weights or {}, default_weight

    def _fuse(self, sources):
        acc: Dict[str, ScoredItem] = {}
        rrf: Dict[str, float] = {}
        for name, hits in sources.items():
            w = self.weights.get(name, self.default_weight)
            for rank, item in enumerate(sorted(hits, key=lambda r: r.score, reverse=True)):
                pid = item.paragraph_id
                if pid not in acc:
                    acc[pid] = item
                    rrf[pid] = 1 / (self.k + rank) * w
                else:
                    rrf[pid] += 1 / (self.k + rank) * w
                    acc[pid].history.append(item.score)
                    if _both(acc[pid].score_type, item.score_type):
                        acc[pid].score_type = BOTH
        out = []
        for pid, item in acc.items():
            item.history.append(rrf[pid])
            item.score = rrf[pid]
            out.append(item)
        return out


class WeightedCombSum(RankFusionAlgorithm):
    def __init__(self, *, window: int, weights: Optional[Dict[str, float]] = None, default_weight: float = 1.0):
        super().__init__(window)
        self.weights, self.default_weight = weights or {}, default_weight

    def _fuse(self, sources):
        first: Dict[str, ScoredItem] = {}
        total: Dict[str, float] = {}
        kind: Dict[str, str] = {}
        hist: Dict[str, List[float]] = {}
        for name, hits in sources.items():
            w = self.weights.get(name, self.default_weight)
            for item in hits:
                pid = item.paragraph_id
                if pid not in first:
                    first[pid], total[pid], kind[pid], hist[pid] = item, 0, item.score_type, []
                total[pid] += item.score * w
                hist[pid].append(item.score)
                if _both(kind[pid], item.score_type):
                    kind[pid] = BOTH
        out = []
        for pid, item in first.items():
            item.history = hist[pid] + [total[pid]]
            item.score, item.score_type = total[pid], kind[pid]
            out.append(item)
        return out


def wcombsum_fuse_batch(lists, window: int = 20):
    """WeightedCombSum for a whole batch through the native host routine (nidx_gpu_rank_fusion_wcombsum): `lists` as for
    rrf_fuse_batch, every one with its f32 scores, hits in the order the retriever returned them."""
    return rrf_fuse_batch(lists, k=0.0, window=window, _comb_sum=True)


def rrf_fuse_batch(lists, k: float = 60.0, window: int = 20, _comb_sum: bool = False):
    """ReciprocalRankFusion for a whole batch through the native host routine (nidx_gpu_rank_fusion_rrf): `lists` =
    [(ids u64 [B][stride], counts u32 [B], weight, scores f32 [B][stride] | None), ...] in source order.
    -> (ids u64 [B][window], scores f64 [B][window], counts u32 [B])"""
    import ctypes as C

    import numpy as np

    from . import _lib

    B = len(lists[0][1]) if lists else 0
    arr = (_lib.RankedListC * max(1, len(lists)))()
    keep = []
    for i, (ids, counts, weight, scores) in enumerate(lists):
        ids = np.ascontiguousarray(ids, dtype=np.uint64)
        counts = np.ascontiguousarray(counts, dtype=np.uint32)
        sc = None if scores is None else np.ascontiguousarray(scores, dtype=np.float32)
        keep += [ids, counts, sc]
        arr[i] = _lib.RankedListC(ids.ctypes.data, None if sc is None else sc.ctypes.data, counts.ctypes.data, ids.shape[1] if ids.ndim == 2 else 0, float(weight))
    out_ids = np.zeros((B, window), np.uint64)
    out_scores = np.zeros((B, window), np.float64)
    out_counts = np.zeros(B, np.uint32)
    if _comb_sum:
        _lib.check(_lib.lib().nidx_gpu_rank_fusion_wcombsum(arr, len(lists), B, window, out_ids.ctypes.data, out_scores.ctypes.data, out_counts.ctypes.data))
    else:
        _lib.check(_lib.lib().nidx_gpu_rank_fusion_rrf(arr, len(lists), B, float(k), window, out_ids.ctypes.data, out_scores.ctypes.data, out_counts.ctypes.data))
    return out_ids, out_scores, out_counts
