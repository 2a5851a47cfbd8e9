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
    # FuzzyTermQuery (fuzzy_query.rs:55-125): the clause is the UNION of these terms' documents, each scored
    # ConstScorer(boost) once; `term` and `mode` are ignored
    term_set: Optional[Sequence[int]] = None
    # parse_excluded (keyword_parser.rs:93-105): every document OUTSIDE the union, AllQuery's 1.0 * boost
    complement: bool = False
    # PhraseQuery(term_set): the terms at consecutive positions in this order (slop 0) or, with `slop` > 0, each within `slop`
    # extra positions of the match so far (PhraseQuery::set_slop); tf = occurrences, Bm25Weight::for_terms (the index must have
    # been opened with positions)
    phrase: bool = False
    slop: int = 0
    # a nested BooleanQuery (NIDX_BM25_SUBQUERY): its leaves — clauses of any kind, nested queries included; the outer clause
    # contributes boost x the nested query's own score; `term` and `mode` are ignored
    subquery: Optional[Sequence["Clause"]] = None


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

    def __init__(self, term_offsets, doc_ids, tfs, fieldnorm_ids, total_num_tokens, alive=None, pos_offsets=None, positions=None):
        # positions of every posting (WithFreqsAndPositions): posting i owns positions[pos_offsets[i] .. pos_offsets[i+1])
        self.pos_offsets = None if pos_offsets is None else np.ascontiguousarray(pos_offsets, dtype=np.uint64)
        self.positions = None if positions is None else np.ascontiguousarray(positions, dtype=np.uint32)
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
    def from_term_docs(cls, docs: Sequence[np.ndarray], n_terms: int, alive=None, with_positions: bool = False) -> "Bm25Segment":
        """docs[i] = array of term ids (with repeats) of document i — what the single-segment tantivy
        writer (nidx_tantivy/src/lib.rs:39-78) would index.  with_positions: token index of every occurrence."""
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
        pos_offsets = positions = None
        if with_positions:
            pos_in_doc = np.concatenate([np.arange(len(x), dtype=np.int64) for x in docs]) if len(docs) else np.zeros(0, np.int64)
            order = np.lexsort((pos_in_doc, key))  # by (term, doc), then position
            positions = pos_in_doc[order].astype(np.uint32)
            pos_offsets = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
        return cls(term_offsets, d.astype(np.uint32), counts.astype(np.uint32), ids, int(lens.sum()), alive, pos_offsets, positions)

    def to_c(self) -> _lib.Bm25SegmentC:
        return _lib.Bm25SegmentC(self.n_docs, self.total_num_tokens, self.term_offsets.size - 1, self.term_offsets.ctypes.data,
                                 self.doc_ids.ctypes.data, self.tfs.ctypes.data, self.fieldnorm_ids.ctypes.data,
                                 None if self.alive is None else self.alive.ctypes.data,
                                 None if self.pos_offsets is None else self.pos_offsets.ctypes.data,
                                 None if self.positions is None else self.positions.ctypes.data)


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

    def set_fast_field(self, segment: int, field: int, values) -> None:
        """`created` (0) / `modified` (1) fast field of one segment: int64 per document."""
        v = np.ascontiguousarray(values, dtype=np.int64)
        assert v.size == self.segments[segment].n_docs
        _lib.check(_lib.lib().nidx_gpu_bm25_set_fast_field(self._handle, segment, field, v.ctypes.data))

    def apply_deletions(self, segment: int, terms: Sequence[int]) -> int:
        """open_index_with_deletions' device half: the documents of these posting lists leave the segment's alive bitset.
        -> live documents of the segment afterwards."""
        t = np.ascontiguousarray(terms, dtype=np.uint32)
        n_alive = C.c_uint64(0)
        _lib.check(_lib.lib().nidx_gpu_bm25_apply_deletions(self._handle, segment, t.ctypes.data if t.size else None, t.size, C.byref(n_alive)))
        return int(n_alive.value)

    def set_dictionary(self, terms: Sequence[str]) -> None:
        """The term dictionary of the scored field, term id = position."""
        enc = [t.encode("utf-8") for t in terms]
        offs = np.zeros(len(enc) + 1, np.uint64)
        offs[1:] = np.cumsum([len(e) for e in enc])
        blob = np.frombuffer(b"".join(enc) or b"\0", np.uint8)
        _lib.check(_lib.lib().nidx_gpu_bm25_set_dictionary(self._handle, blob.ctypes.data, offs.ctypes.data))

    def fuzzy_terms(self, word: str, prefix: bool = False) -> np.ndarray:
        """Ids of the dictionary terms FuzzyTermQuery's automaton accepts (distance 1, transposition = one edit)."""
        q = word.encode("utf-8")
        n = C.c_uint32(0)
        cap = 1024
        while True:
            out = np.zeros(cap, np.uint32)
            _lib.check(_lib.lib().nidx_gpu_bm25_fuzzy_terms(self._handle, q, len(q), int(prefix), out.ctypes.data, cap, C.byref(n)))
            if n.value <= cap:
                return out[: n.value].copy()
            cap = n.value

    def prefilter(self, ops: Sequence[Tuple[int, int, int]], lists: Sequence[int] = (), ranges: Sequence[Tuple[int, Optional[int], Optional[int]]] = (),
                  phrases: Sequence[Sequence[int]] = ()) -> Tuple[np.ndarray, int]:
        """TextReaderService::prefilter (nidx_text/src/reader.rs:148-180) on the device: `ops` is a postfix filter program
        [(op, a, b)] over term ids `lists`, inclusive fast-field `ranges` [(field, since | None, until | None)] and
        `phrases` (term-id tuples).  Returns (ascending docaddrs of the live matching documents, live documents of the index)."""
        c_ops = (_lib.FilterOpC * max(1, len(ops)))(*[_lib.FilterOpC(*o) for o in ops])
        c_lists = np.ascontiguousarray(lists, dtype=np.uint32)
        c_ranges = (_lib.Bm25DateRangeC * max(1, len(ranges)))(
            *[_lib.Bm25DateRangeC(f, int(lo is not None), int(hi is not None), 0, int(lo or 0), int(hi or 0)) for f, lo, hi in ranges])
        p_terms = np.ascontiguousarray([t for ph in phrases for t in ph], dtype=np.uint32)
        p_offs = np.zeros(len(phrases) + 1, np.uint64)
        p_offs[1:] = np.cumsum([len(ph) for ph in phrases])
        req = _lib.Bm25PrefilterC(
            _lib.FilterProgramC(C.addressof(c_ops) if len(ops) else None, len(ops), c_lists.ctypes.data if c_lists.size else None, c_lists.size),
            C.addressof(c_ranges) if len(ranges) else None, len(ranges), len(phrases),
            p_terms.ctypes.data if p_terms.size else None, p_offs.ctypes.data)
        n, live = C.c_uint64(0), C.c_uint64(0)
        cap = 1 << 16
        while True:
            out = np.zeros(cap, np.uint64)
            _lib.check(_lib.lib().nidx_gpu_bm25_prefilter(self._handle, C.byref(req), out.ctypes.data, cap, C.byref(n), C.byref(live)))
            if n.value <= cap:
                return out[: n.value].copy(), live.value
            cap = n.value

    def search_batch_ex(self, queries: Sequence[Sequence[Clause]], k: int, after: Optional[Sequence[Optional[SearchAfter]]] = None,
                        order_field: int = -1, order_desc: bool = True, facets: Optional[Sequence[Sequence[int]]] = None):
        """The collectors around the scoring (nidx_text/src/reader.rs:367-451): term-set clauses, TopDocs ordered by a
        fast field, facet counts (facets[q] = term ids to count among query q's matching documents).
        -> dict(docaddr, score, count, total, postings, order_value, facet_counts[q] = counts aligned with facets[q])"""
        B = len(queries)
        offsets = np.zeros(B + 1, dtype=np.uint64)
        flat = []
        for i, q in enumerate(queries):
            flat.extend(q)
            offsets[i + 1] = len(flat)
        cl = (_lib.Bm25ClauseC * max(1, len(flat)))()
        set_terms: List[int] = []
        set_offsets = [0]
        set_comp: List[int] = []
        phrase_terms: List[int] = []
        phrase_offsets = [0]
        phrase_slops: List[int] = []
        sub_leaves: List[Tuple[int, int, int, float]] = []   # (term word, occur, mode, boost) of every nested query's leaves
        sub_offsets = [0]

        def lower(c: Clause) -> Tuple[int, int]:
            """-> (term word, mode) of one clause; a nested query's own leaves are lowered first, so the queries of a tree are
            listed children before parents (the order the library materialises them in)."""
            if c.subquery is not None:
                leaves = [(*lower(l), l) for l in c.subquery]
                sub_leaves.extend((t, l.occur, m, l.boost) for t, m, l in leaves)
                sub_offsets.append(len(sub_leaves))
                return _lib.BM25_SUBQUERY | (len(sub_offsets) - 2), c.mode
            if c.term_set is not None and c.phrase:
                phrase_terms.extend(int(t) for t in c.term_set)
                phrase_offsets.append(len(phrase_terms))
                phrase_slops.append(int(c.slop))
                return _lib.BM25_PHRASE | (len(phrase_offsets) - 2), c.mode
            if c.term_set is not None:
                set_terms.extend(int(t) for t in c.term_set)
                set_offsets.append(len(set_terms))
                set_comp.append(int(c.complement))
                return _lib.BM25_TERM_SET | (len(set_offsets) - 2), _lib.CONST_SCORE
            return c.term, c.mode

        for i, c in enumerate(flat):
            term, mode = lower(c)
            cl[i].term, cl[i].occur, cl[i].mode, cl[i].boost = term, c.occur, mode, c.boost
        af = None
        if after is not None:
            af = (_lib.Bm25SearchAfterC * max(1, B))()
            for i, a in enumerate(after):
                if a is not None:
                    af[i].has_after, af[i].score, af[i].tie_break, af[i].docaddr = 1, a.score, a.tie_break, a.docaddr
        kk = max(1, k)
        docaddr = np.zeros((B, kk), dtype=np.uint64)
        score = np.zeros((B, kk), dtype=np.float32)
        order_value = np.zeros((B, kk), dtype=np.int64)
        count = np.zeros(B, dtype=np.uint32)
        total = np.zeros(B, dtype=np.uint64)
        postings = np.zeros(B, dtype=np.uint64)
        st = np.ascontiguousarray(set_terms, dtype=np.uint32)
        so = np.ascontiguousarray(set_offsets, dtype=np.uint64)
        opt = _lib.Bm25SearchOptionsC()
        opt.k = k
        opt.after = C.cast(af, C.c_void_p) if af is not None else None
        opt.term_set_terms = st.ctypes.data if st.size else None
        opt.term_set_offsets = so.ctypes.data
        opt.n_term_sets = len(set_offsets) - 1
        sc = np.ascontiguousarray(set_comp, dtype=np.uint8)
        opt.term_set_complement = sc.ctypes.data if sc.size else None
        pt = np.ascontiguousarray(phrase_terms, dtype=np.uint32)
        po = np.ascontiguousarray(phrase_offsets, dtype=np.uint64)
        opt.phrase_terms = pt.ctypes.data if pt.size else None
        opt.phrase_offsets = po.ctypes.data
        opt.n_phrases = len(phrase_offsets) - 1
        ps = np.ascontiguousarray(phrase_slops, dtype=np.uint32)
        opt.phrase_slops = ps.ctypes.data if any(phrase_slops) else None
        sub_cl = (_lib.Bm25ClauseC * max(1, len(sub_leaves)))()
        for i, (t, o, m, b) in enumerate(sub_leaves):
            sub_cl[i].term, sub_cl[i].occur, sub_cl[i].mode, sub_cl[i].boost = t, o, m, b
        sub_off = np.ascontiguousarray(sub_offsets, dtype=np.uint64)
        opt.subquery_clauses = C.cast(sub_cl, C.c_void_p) if sub_leaves else None
        opt.subquery_offsets = sub_off.ctypes.data
        opt.n_subqueries = len(sub_offsets) - 1
        opt.order_field, opt.order_desc = order_field, int(order_desc)
        fo = ft = fc = None
        if facets is not None:
            fo = np.zeros(B + 1, np.uint64)
            fo[1:] = np.cumsum([len(f) for f in facets])
            ft = np.ascontiguousarray([t for f in facets for t in f], dtype=np.uint32)
            fc = np.zeros(max(int(fo[-1]), 1), np.uint64)
            opt.facet_terms = ft.ctypes.data if ft.size else None
            opt.facet_offsets = fo.ctypes.data
            opt.out_facet_counts = fc.ctypes.data
        opt.out_order_value = order_value.ctypes.data
        _lib.check(_lib.lib().nidx_gpu_bm25_search_ex(self._handle, cl, offsets.ctypes.data, B, C.byref(opt), docaddr.ctypes.data,
                                                      score.ctypes.data, count.ctypes.data, total.ctypes.data, postings.ctypes.data))
        facet_counts = None
        if facets is not None:
            facet_counts = [fc[int(fo[q]): int(fo[q + 1])].astype(np.int64) for q in range(B)]
        return {"docaddr": docaddr, "score": score, "count": count, "total": total, "postings": postings,
                "order_value": order_value, "facet_counts": facet_counts}

    def submit(self, queries: Sequence[Sequence[Clause]], k: int) -> int:
        """nidx_gpu_bm25_search_submit for a batch of plain term clauses -> ticket; the host side of this batch overlaps the kernels of
        the batches already in flight (at most NIDX_GPU_BM25_MAX_TICKETS = 16 tickets outstanding; several threads may submit at once)."""
        B = len(queries)
        offsets = np.zeros(B + 1, dtype=np.uint64)
        flat = []
        for i, q in enumerate(queries):
            flat.extend(q)
            offsets[i + 1] = len(flat)
        cl = (_lib.Bm25ClauseC * max(1, len(flat)))()
        for i, c in enumerate(flat):
            if c.term_set is not None or c.subquery is not None:
                raise ValueError("submit() takes plain term clauses; use search_batch_ex for the rest")
            cl[i].term, cl[i].occur, cl[i].mode, cl[i].boost = c.term, c.occur, c.mode, c.boost
        opt = _lib.Bm25SearchOptionsC()
        opt.k = k
        opt.order_field = -1
        t = C.c_uint64(0)
        _lib.check(_lib.lib().nidx_gpu_bm25_search_submit(self._handle, cl, offsets.ctypes.data, B, C.byref(opt), C.byref(t)))
        self._tickets = getattr(self, "_tickets", {})
        self._tickets[t.value] = (B, k)
        return t.value

    def wait(self, ticket: int):
        """-> (docaddr [B][k] u64, score [B][k] f32, count [B], total [B], postings [B]) of the batch submitted under `ticket`."""
        B, k = self._tickets.pop(ticket)
        kk = max(1, k)
        docaddr = np.zeros((B, kk), dtype=np.uint64)
        score = np.zeros((B, kk), dtype=np.float32)
        count = np.zeros(B, dtype=np.uint32)
        total = np.zeros(B, dtype=np.uint64)
        postings = np.zeros(B, dtype=np.uint64)
        _lib.check(_lib.lib().nidx_gpu_bm25_search_wait(self._handle, ticket, docaddr.ctypes.data, score.ctypes.data, count.ctypes.data, total.ctypes.data,
                                                        postings.ctypes.data))
        return docaddr, score, count, total, postings

    def search_batch(self, queries: Sequence[Sequence[Clause]], k: int, after: Optional[Sequence[Optional[SearchAfter]]] = None):
        """-> (docaddr [B][k] u64, score [B][k] f32, count [B], total [B], postings [B])"""
        if any(c.term_set is not None or c.subquery is not None for q in queries for c in q):
            r = self.search_batch_ex(queries, k, after)
            return r["docaddr"], r["score"], r["count"], r["total"], r["postings"]
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
