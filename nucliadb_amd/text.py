"""Host-side mirrors of `nidx_text::TextSearcher` and `nidx_paragraph::ParagraphSearcher` — the
request/response layer around the BM25 kernel.

What runs where:
  * tokenising, the term dictionary, query text -> term clauses, response assembly (`min_score`
    cut, `next_page`, `k + 1` over-fetch, search-after cursor): here, on the host, following
      nidx_text/src/reader.rs:289-451 (`do_search`, `convert_bm25_order`),
      nidx_text/src/search_query.rs:92-126 (`create_query`),
      nidx_paragraph/src/reader.rs:104-139,244-390 (`search`, `do_search`, `is_after`),
      nidx_paragraph/src/search_query.rs:185-243, query_parser/keyword_parser.rs:27-105,
      nidx_paragraph/src/search_response.rs:218-311;
  * every posting read, BM25 score, boolean combination, top-k and count: in the HIP kernel behind
    `nidx_gpu_bm25_search`.

Supported query shapes are the term-clause ones (the overwhelmingly common path): text index =
conjunction of the body's tokens (tantivy QueryParser with `set_conjunction_by_default`); paragraph
index = Should(term, IndexRecordOption::Basic) per literal token, wrapped with the Must clauses
`repeated_in_field:0` (unless with_duplicates) and label filters.  Quoted phrases, `-excluded`
terms, the fuzzy fallback and facets are not term clauses and raise NotImplementedError (SURVEY §8f
row 4, "next").
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import _lib
from .bm25 import Bm25Searcher, Bm25Segment, Clause, SearchAfter, tokenize

ALL_DOCS = "\x00all"            # pseudo term: every document (AllQuery, scored by ConstScorer)
NOT_REPEATED = "\x00repeated:0"  # pseudo term: paragraphs with repeated_in_field == 0


@dataclass
class ResultScore:
    bm25: float
    docaddr: int


@dataclass
class TextDocument:
    uuid: str
    field: str
    text: str
    labels: List[str] = field(default_factory=list)
    repeated_in_field: bool = False  # paragraph index only


class Vocabulary:
    """Term dictionary shared by every segment of an index (term ids must agree across segments)."""

    def __init__(self):
        self.ids: Dict[str, int] = {}

    def id(self, term: str) -> int:
        return self.ids.setdefault(term, len(self.ids))

    def lookup(self, term: str) -> Optional[int]:
        return self.ids.get(term)


class TextSegment:
    """One single-segment tantivy writer's output (nidx_tantivy/src/lib.rs:39-78), as token streams."""

    def __init__(self, docs: Sequence[TextDocument], vocab: Vocabulary):
        self.docs = list(docs)
        self.vocab = vocab
        self.streams = []
        for d in self.docs:
            toks = [vocab.id(t) for t in tokenize(d.text)]
            self.streams.append(toks)
            vocab.id(ALL_DOCS), vocab.id(NOT_REPEATED)
            for lab in d.labels:
                vocab.id("\x00label:" + lab)

    def to_bm25(self, n_terms: int, alive=None) -> Bm25Segment:
        """Postings of the text field, plus constant-frequency pseudo terms for AllQuery / labels /
        repeated_in_field.  Field norms come from the text field's token count only."""
        L = _lib.lib()
        n = len(self.docs)
        lens = np.array([len(s) for s in self.streams], dtype=np.int64)
        terms, docs = [], []
        for i, (d, s) in enumerate(zip(self.docs, self.streams)):
            terms.extend(s)
            docs.extend([i] * len(s))
            extra = [self.vocab.ids[ALL_DOCS]] + ([] if d.repeated_in_field else [self.vocab.ids[NOT_REPEATED]])
            extra += [self.vocab.ids["\x00label:" + lab] for lab in d.labels]
            terms.extend(extra)
            docs.extend([i] * len(extra))
        terms = np.array(terms, dtype=np.int64)
        docs = np.array(docs, dtype=np.int64)
        uniq, counts = np.unique(terms * (n + 1) + docs, return_counts=True)
        t, dd = uniq // (n + 1), uniq % (n + 1)
        term_offsets = np.zeros(n_terms + 1, dtype=np.uint64)
        np.add.at(term_offsets, t + 1, 1)
        term_offsets = np.cumsum(term_offsets).astype(np.uint64)
        table = np.array([L.nidx_gpu_fieldnorm_from_id(i) for i in range(256)], dtype=np.int64)
        ids = (np.searchsorted(table, lens, side="right") - 1).astype(np.uint8)
        return Bm25Segment(term_offsets, dd.astype(np.uint32), counts.astype(np.uint32), ids, int(lens.sum()), alive)


def _bitset(mask: np.ndarray) -> np.ndarray:
    n = mask.shape[0]
    words = (n + 63) // 64
    padded = np.zeros(words * 64, dtype=np.uint8)
    padded[:n] = mask
    return np.packbits(padded.reshape(words, 64), axis=1, bitorder="little").view(np.uint64).reshape(words).copy()


class _Index:
    def __init__(self, segments: Sequence[TextSegment], deleted: Sequence[set] = ()):
        self.segments = list(segments)
        self.vocab = segments[0].vocab if segments else Vocabulary()
        n_terms = len(self.vocab.ids) + 1  # + one always-empty term for words missing from the dictionary
        self.empty_term = n_terms - 1
        bsegs = []
        for i, seg in enumerate(self.segments):
            alive = None
            if i < len(deleted) and deleted[i]:
                m = np.ones(len(seg.docs), dtype=bool)
                m[list(deleted[i])] = False
                alive = _bitset(m)
            bsegs.append(seg.to_bm25(n_terms, alive))
        self.searcher = Bm25Searcher.open(bsegs)

    def term(self, word: str) -> int:
        t = self.vocab.lookup(word)
        return self.empty_term if t is None else t

    def doc(self, docaddr: int) -> TextDocument:
        return self.segments[docaddr >> 32].docs[docaddr & 0xFFFFFFFF]

    def close(self):
        self.searcher.close()


# =============================================================================== nidx_text
@dataclass
class DocumentSearchRequest:
    body: str = ""
    result_per_page: int = 0
    min_score: float = 0.0
    label_filter: Optional[List[str]] = None  # a conjunction of labels (filter_expression subset)


@dataclass
class DocumentResult:
    uuid: str
    field: str
    score: ResultScore
    labels: List[str]


@dataclass
class DocumentSearchResponse:
    total: int
    results: List[DocumentResult]
    next_page: bool
    query: str


class TextSearcher:
    """nidx_text::TextSearcher (lib.rs:178-237) — `search` only."""

    def __init__(self, index: _Index):
        self._index = index

    @classmethod
    def open(cls, segments: Sequence[TextSegment], deleted: Sequence[set] = ()) -> "TextSearcher":
        return cls(_Index(segments, deleted))

    def close(self):
        self._index.close()

    def _clauses(self, request: DocumentSearchRequest) -> List[Clause]:
        words = tokenize(request.body)
        if any(ch in request.body for ch in '"-+():^~*'):
            raise NotImplementedError("only plain conjunctive term queries are term-clause shaped")
        clauses = []
        if not words:  # create_query: empty text => AllQuery (search_query.rs:100-104)
            clauses.append(Clause(self._index.term(ALL_DOCS), _lib.OCCUR_MUST, _lib.CONST_SCORE, 1.0))
        for w in words:  # set_conjunction_by_default: every term is a Must TermQuery with frequencies
            clauses.append(Clause(self._index.term(w), _lib.OCCUR_MUST, _lib.TF_FREQ, 1.0))
        for lab in request.label_filter or []:
            # filter clauses score too in tantivy's BooleanQuery; facet TermQuerys carry no frequencies
            clauses.append(Clause(self._index.term("\x00label:" + lab), _lib.OCCUR_MUST, _lib.TF_BASIC, 1.0))
        return clauses

    def search(self, request: DocumentSearchRequest) -> DocumentSearchResponse:
        k = max(0, int(request.result_per_page))
        # TopDocs::with_limit(results + 1) (reader.rs:380,433)
        docaddr, score, count, total, _ = self._index.searcher.search_batch([self._clauses(request)], k + 1)
        total_ = int(total[0])
        results = []
        for i in range(min(int(count[0]), k)):  # .take(results_per_page), drop `score < min_score` (reader.rs:299-305)
            s = float(score[0, i])
            if s < request.min_score:
                continue
            d = self._index.doc(int(docaddr[0, i]))
            results.append(DocumentResult(d.uuid, d.field, ResultScore(s, int(docaddr[0, i])), [l for l in d.labels if l.startswith("/l/")]))
        return DocumentSearchResponse(total_, results, total_ > k, request.body)


# =============================================================================== nidx_paragraph
STOP_WORDS_MIN_TOKENS = 2


@dataclass
class ParagraphSearchRequest:
    body: str = ""
    result_per_page: int = 0
    with_duplicates: bool = False
    min_score: float = 0.0
    label_filter: Optional[List[str]] = None
    search_after: Optional[SearchAfter] = None


@dataclass
class ParagraphResult:
    uuid: str
    field: str
    paragraph: str
    score: ResultScore
    labels: List[str]


@dataclass
class ParagraphSearchResponse:
    total: int
    results: List[ParagraphResult]
    next_page: bool
    query: str


class ParagraphSearcher:
    """nidx_paragraph::ParagraphSearcher (lib.rs:117-169) — keyword `search` only."""

    def __init__(self, index: _Index):
        self._index = index

    @classmethod
    def open(cls, segments: Sequence[TextSegment], deleted: Sequence[set] = ()) -> "ParagraphSearcher":
        return cls(_Index(segments, deleted))

    def close(self):
        self._index.close()

    def _clauses(self, request: ParagraphSearchRequest) -> List[Clause]:
        if '"' in request.body or any(w.startswith("-") for w in request.body.split()):
            raise NotImplementedError("quoted phrases and -excluded terms are not term clauses")
        words = tokenize(request.body)
        clauses = []
        if not words:  # parse_keyword_query: no subqueries => AllQuery
            clauses.append(Clause(self._index.term(ALL_DOCS), _lib.OCCUR_MUST, _lib.CONST_SCORE, 1.0))
        # TermQuery(text, IndexRecordOption::Basic) per literal, Occur::Should (keyword_parser.rs:36-67)
        # as a required group: the keyword BooleanQuery sits under Occur::Must next to the filters
        # (search_query.rs:191-228), so a paragraph has to match at least one of its words
        should = [Clause(self._index.term(w), _lib.OCCUR_SHOULD_GROUP, _lib.TF_BASIC, 1.0) for w in words]
        musts = []
        for lab in request.label_filter or []:
            musts.append(Clause(self._index.term("\x00label:" + lab), _lib.OCCUR_MUST, _lib.TF_BASIC, 1.0))
        if not request.with_duplicates:  # Must TermQuery(repeated_in_field = 0, Basic) (search_query.rs:218-223)
            musts.append(Clause(self._index.term(NOT_REPEATED), _lib.OCCUR_MUST, _lib.TF_BASIC, 1.0))
        return clauses + should + musts

    def search(self, request: ParagraphSearchRequest) -> ParagraphSearchResponse:
        k = max(0, int(request.result_per_page))
        clauses = self._clauses(request)
        after = [request.search_after] if request.search_after is not None else None
        docaddr, score, count, total, _ = self._index.searcher.search_batch([clauses], k + 1, after)
        obtained = int(count[0])
        scores = [float(score[0, i]) for i in range(obtained)]
        # search_response.rs:218-311: next_page counts scores above min_score, results stop at the first below it
        next_page = sum(1 for s in scores if s > request.min_score) > k
        results = []
        for i in range(min(obtained, k)):
            if scores[i] < request.min_score:
                break
            d = self._index.doc(int(docaddr[0, i]))
            results.append(ParagraphResult(d.uuid, d.field, d.text, ResultScore(scores[i], int(docaddr[0, i])), list(d.labels)))
        return ParagraphSearchResponse(int(total[0]), results, next_page, request.body)
