"""Host-side mirror of the BM25 scoring surface of `nidx_text` / `nidx_paragraph` over libnidx_gpu.

The reference delegates everything below `searcher.search(&query, &(TopDocs, Count))` to tantivy
(nidx_text/src/reader.rs:433-435, nidx_paragraph/src/reader.rs:290-292,330-332).  Here the postings
live in HBM and a batch of boolean term queries is scored by the HIP kernel; this module keeps what
stays on the host: the term dictionary, the tokenizer of the `text` field (tantivy's default:
split on non-alphanumeric, drop tokens > 40 bytes, lowercase) and the query -> clause mapping.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib

Occur = type("Occur", (), {"Should": _lib.OCCUR_SHOULD, "Must": _lib.OCCUR_MUST, "MustNot": _lib.OCCUR_MUST_NOT, "ShouldGroup": _lib.OCCUR_SHOULD_GROUP})
TfMode = type("TfMode", (), {"Freq": _lib.TF_FREQ, "Basic": _lib.TF_BASIC, "Const": _lib.CONST_SCORE})


@dataclass
class Clause:
    term: int
    occur: int = _lib.OCCUR_SHOULD
    mode: int = _lib.TF_FREQ
    boost: float = 1.0


@dataclass
class SearchAfter:
    """(score, docaddr) cursor of nidx_paragraph/src/reader.rs:350-390; tie_break 0 keeps every tie,
    1 keeps ties with a greater docaddr, 2 drops ties."""
    score: float
    tie_break: int
    docaddr: int


def tokenize(text: str) -> List[str]:
    """tantivy's "default" tokenizer: SimpleTokenizer + RemoveLongFilter(40) + LowerCaser."""
    out, cur = [], []
    for ch in text:
        if ch.isalnum():
            cur.append(ch)
        elif cur:
            out.append("".join(cur))
            cur = []
    if cur:
        out.append("".join(cur))
    return [t.lower() for t in out if len(t.encode("utf-8")) <= 40]


class Bm25Segment:
    """One tantivy segment's postings for the scored field, term-id resolved (CSR)."""

    def __init__(self, term_offsets, doc_ids, tfs, fieldnorm_ids, total_num_tokens, alive=None):
        self.term_offsets = np.ascontiguousarray(term_offsets, dtype=np.uint64)
        self.doc_ids = np.ascontiguousarray(doc_ids, dtype=np.uint32)
        self.tfs = np.ascontiguousarray(tfs, dtype=np.uint32)
        self.fieldnorm_ids = np.ascontiguousarray(fieldnorm_ids, dtype=np.uint8)
        self.total_num_tokens = int(total_num_tokens)
        self.alive = None if alive is None else np.ascontiguousarray(alive, dtype=np.uint64)

    @property
    def n_docs(self) -> int:
        return int(self.fieldnorm_ids.size)

    @classmethod
    def from_term_docs(cls, docs: Sequence[np.ndarray], n_terms: int, alive=None) -> "Bm25Segment":
        """docs[i] = array of term ids (with repeats) of document i — what the single-segment tantivy
        writer (nidx_tantivy/src/lib.rs:39-78) would index."""
        L = _lib.lib()
        lens = np.array([len(d) for d in docs], dtype=np.int64)
        doc_of = np.repeat(np.arange(len(docs), dtype=np.int64), lens)
        terms = np.concatenate(docs).astype(np.int64) if len(docs) else np.zeros(0, np.int64)
        key = terms * (len(docs) + 1) + doc_of
        uniq, counts = np.unique(key, return_counts=True)
        t = uniq // (len(docs) + 1)
        d = uniq % (len(docs) + 1)
        term_offsets = np.zeros(n_terms + 1, dtype=np.uint64)
        np.add.at(term_offsets, t + 1, 1)
        term_offsets = np.cumsum(term_offsets).astype(np.uint64)
        table = np.array([L.nidx_gpu_fieldnorm_from_id(i) for i in range(256)], dtype=np.int64)
        ids = (np.searchsorted(table, lens, side="right") - 1).astype(np.uint8)
        return cls(term_offsets, d.astype(np.uint32), counts.astype(np.uint32), ids, int(lens.sum()), alive)

    def to_c(self) -> _lib.Bm25SegmentC:
        return _lib.Bm25SegmentC(self.n_docs, self.total_num_tokens, self.term_offsets.size - 1, self.term_offsets.ctypes.data,
                                 self.doc_ids.ctypes.data, self.tfs.ctypes.data, self.fieldnorm_ids.ctypes.data,
                                 None if self.alive is None else self.alive.ctypes.data)


class Bm25Searcher:
    """The scoring core shared by TextSearcher::search and ParagraphSearcher::search."""

    def __init__(self):
        self._handle = C.c_void_p()
        self.segments: List[Bm25Segment] = []

    @classmethod
    def open(cls, segments: Sequence[Bm25Segment]) -> "Bm25Searcher":
        self = cls()
        self.segments = list(segments)
        arr = (_lib.Bm25SegmentC * max(1, len(segments)))(*[s.to_c() for s in segments])
        _lib.check(_lib.lib().nidx_gpu_bm25_open(arr, len(segments), C.byref(self._handle)))
        return self

    def close(self):
        if self._handle:
            _lib.lib().nidx_gpu_bm25_close(self._handle)
            self._handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def space_usage(self) -> int:
        out = C.c_uint64(0)
        _lib.check(_lib.lib().nidx_gpu_bm25_space_usage(self._handle, C.byref(out)))
        return out.value

    def search_batch(self, queries: Sequence[Sequence[Clause]], k: int, after: Optional[Sequence[Optional[SearchAfter]]] = None):
        """-> (docaddr [B][k] u64, score [B][k] f32, count [B], total [B], postings [B])"""
        B = len(queries)
        offsets = np.zeros(B + 1, dtype=np.uint64)
        flat = []
        for i, q in enumerate(queries):
            flat.extend(q)
            offsets[i + 1] = len(flat)
        cl = (_lib.Bm25ClauseC * max(1, len(flat)))()
        for i, c in enumerate(flat):
            cl[i].term, cl[i].occur, cl[i].mode, cl[i].boost = c.term, c.occur, c.mode, c.boost
        af = None
        if after is not None:
            af = (_lib.Bm25SearchAfterC * max(1, B))()
            for i, a in enumerate(after):
                if a is not None:
                    af[i].has_after, af[i].score, af[i].tie_break, af[i].docaddr = 1, a.score, a.tie_break, a.docaddr
        kk = max(1, k)
        docaddr = np.zeros((B, kk), dtype=np.uint64)
        score = np.zeros((B, kk), dtype=np.float32)
        count = np.zeros(B, dtype=np.uint32)
        total = np.zeros(B, dtype=np.uint64)
        postings = np.zeros(B, dtype=np.uint64)
        _lib.check(_lib.lib().nidx_gpu_bm25_search(self._handle, cl, offsets.ctypes.data, B, k, af, docaddr.ctypes.data,
                                                   score.ctypes.data, count.ctypes.data, total.ctypes.data, postings.ctypes.data))
        return docaddr, score, count, total, postings
