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

Supported query shapes: text index = conjunction of the body's tokens (tantivy QueryParser with
`set_conjunction_by_default`); paragraph index = the nidx query grammar (query_parser/tokenizer.rs:49-135):
literals -> Should(term, IndexRecordOption::Basic), one-word quotes -> the same term query, wrapped with the
Must clauses `repeated_in_field:0` (unless with_duplicates) and label filters; the FUZZY FALLBACK
(reader.rs:128-133, fuzzy_parser.rs:35-93, search_query.rs:200-240): every literal of >= 3 characters becomes
a FuzzyTermQuery (Levenshtein 1, the last literal as a prefix when >= 4 characters) expanded against the term
dictionary on the device and scored ConstScorer(0.5).  Both searchers collect FACETS (FacetCollector, top 50
children per requested facet) and can ORDER by the created / modified fast fields.  `-excluded` words are
Should(everything but the word): the complement of the posting list, built on the device.  Multi-word quotes are
PhraseQuery's (positions in HBM, phrase lists materialised on the device).
"""
from __future__ import annotations

import bisect

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Set, Tuple

import numpy as np

from . import _lib
from .bm25 import Bm25Searcher, Bm25Segment, Clause, SearchAfter, tokenize

ALL_DOCS = "\x00all"            # pseudo term: every document (AllQuery, scored by ConstScorer)
NOT_REPEATED = "\x00repeated:0"  # pseudo term: paragraphs with repeated_in_field == 0
PUBLIC = "\x00public"            # pseudo term: groups_public == 1


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
    created: int = 0                 # fast fields (schema.rs:59-115), seconds
    modified: int = 0
    access_groups: Optional[List[str]] = None  # resource security; None / empty = public (resource_indexer.rs:49-63)


class Vocabulary:
    """Term dictionary shared by every segment of an index (term ids must agree across segments)."""

    def __init__(self):
        self.ids: Dict[str, int] = {}

    def id(self, term: str) -> int:
        return self.ids.setdefault(term, len(self.ids))

    def lookup(self, term: str) -> Optional[int]:
        return self.ids.get(term)

    def sorted_text_terms(self) -> Tuple[List[bytes], List[int]]:
        """(term bytes ascending, their ids) of the text field's terms — the order tantivy's term dictionary keeps them in."""
        cached = getattr(self, "_sorted", None)
        if cached is None or cached[0] != len(self.ids):
            rows = sorted((t.encode(), i) for t, i in self.ids.items() if not t.startswith("\x00"))
            cached = (len(self.ids), [r[0] for r in rows], [r[1] for r in rows])
            self._sorted = cached
        return cached[1], cached[2]


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
            for t in self._pseudo_terms(d):
                vocab.id(t)

    @staticmethod
    def _pseudo_terms(d: TextDocument) -> List[str]:
        """The non-text indexed fields of one document (nidx_text/src/resource_indexer.rs:34-110) as pseudo terms of the one
        term-id space: AllQuery, repeated_in_field, facets (every ancestor path, tantivy's FacetTokenizer), the resource
        uuid, the field facet, the encoded field id, and the security groups."""
        out = {ALL_DOCS}
        if not d.repeated_in_field:
            out.add(NOT_REPEATED)
        for lab in d.labels:
            out.update("\x00label:" + anc for anc in facet_ancestors(lab))
        out.add("\x00uuid:" + d.uuid)
        out.update("\x00field:" + anc for anc in facet_ancestors(d.field))
        out.add("\x00fid:" + d.uuid + "/" + d.field.lstrip("/"))
        if d.access_groups:
            for g in d.access_groups:
                out.update("\x00group:" + anc for anc in facet_ancestors(g if g.startswith("/") else "/" + g))
        else:
            out.add(PUBLIC)
        return sorted(out)

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
            extra = [self.vocab.ids[t] for t in self._pseudo_terms(d)]
            terms.extend(extra)
            docs.extend([i] * len(extra))
        terms = np.array(terms, dtype=np.int64)
        docs = np.array(docs, dtype=np.int64)
        # positions: token index inside the text field; pseudo terms (all / labels / repeated) sit at position 0
        pos = []
        for i, (d, st) in enumerate(zip(self.docs, self.streams)):
            pos.extend(range(len(st)))
            pos.extend([0] * len(self._pseudo_terms(d)))
        pos = np.array(pos, dtype=np.int64)
        key = terms * (n + 1) + docs
        uniq, counts = np.unique(key, return_counts=True)
        order = np.lexsort((pos, key))
        positions = pos[order].astype(np.uint32)
        pos_offsets = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
        t, dd = uniq // (n + 1), uniq % (n + 1)
        term_offsets = np.zeros(n_terms + 1, dtype=np.uint64)
        np.add.at(term_offsets, t + 1, 1)
        term_offsets = np.cumsum(term_offsets).astype(np.uint64)
        table = np.array([L.nidx_gpu_fieldnorm_from_id(i) for i in range(256)], dtype=np.int64)
        ids = (np.searchsorted(table, lens, side="right") - 1).astype(np.uint8)
        return Bm25Segment(term_offsets, dd.astype(np.uint32), counts.astype(np.uint32), ids, int(lens.sum()), alive, pos_offsets, positions)


def facet_ancestors(label: str) -> List[str]:
    """/a/b/c -> [/a, /a/b, /a/b/c]"""
    parts = [p for p in label.split("/") if p]
    return ["/" + "/".join(parts[: i + 1]) for i in range(len(parts))]


def is_valid_facet(facet: str) -> bool:
    """Facet::from_text(..).is_ok(): an absolute path (the root "" / "/" is skipped by the callers' tests)"""
    return facet.startswith("/") and len(facet) > 1


@dataclass
class OrderBy:
    """nodereader.proto OrderBy: sort_by Created (0) / Modified (1), desc unless type says asc."""
    field: int = 0
    desc: bool = True


@dataclass
class FacetResult:
    tag: str
    total: int


def _bitset(mask: np.ndarray) -> np.ndarray:
    n = mask.shape[0]
    words = (n + 63) // 64
    padded = np.zeros(words * 64, dtype=np.uint8)
    padded[:n] = mask
    return np.packbits(padded.reshape(words, 64), axis=1, bitorder="little").view(np.uint64).reshape(words).copy()


def terms_in_range(vocab: "Vocabulary", lo: Optional[str], lo_incl: bool, hi: Optional[str], hi_incl: bool) -> List[int]:
    """The ids of the text field's terms inside a range of tantivy's RangeQuery (term bytes in lexicographic order; None = open end).
    The pseudo terms of the other fields (their \x00 prefix) are not terms of the text field.  The dictionary's text terms are kept
    as one sorted table per vocabulary (rebuilt when it grew), the bounds are two bisections."""
    keys, ids = vocab.sorted_text_terms()
    i0 = 0 if lo is None else (bisect.bisect_left if lo_incl else bisect.bisect_right)(keys, lo.encode())
    i1 = len(keys) if hi is None else (bisect.bisect_right if hi_incl else bisect.bisect_left)(keys, hi.encode())
    return sorted(ids[i0:i1]) if i0 < i1 else []


def deletion_terms(vocab: "Vocabulary", segment_seq: int, deletions: Sequence[Tuple[str, int]]) -> List[int]:
    """The TermSetQuery of the DeletionQueryBuilders (nidx_text/src/lib.rs:95-128, nidx_paragraph/src/lib.rs:50-71) for one
    segment: of the (key, seq) deletions those NEWER than the segment (index_reader.rs:47-50: `Seq(segment) < del_seq`); a
    key longer than 32 bytes is a field id (`<32 hex>/<type>/<name>`), else a resource uuid; keys the dictionary does not
    know delete nothing."""
    terms = []
    for key, del_seq in deletions:
        if not segment_seq < del_seq:
            continue
        t = vocab.lookup(("\x00fid:" if len(key) > 32 else "\x00uuid:") + key)
        if t is not None:
            terms.append(t)
    return sorted(set(terms))


class _Index:
    def __init__(self, segments: Sequence[TextSegment], deleted: Sequence[set] = (), seqs: Optional[Sequence[int]] = None,
                 deletions: Sequence[Tuple[str, int]] = ()):
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
        # the term dictionary of the text field (pseudo terms excluded from fuzzy expansion by their \x00 prefix)
        self.terms = [None] * n_terms
        for t, i in self.vocab.ids.items():
            self.terms[i] = t
        self.terms[self.empty_term] = "\x00empty"
        self.searcher.set_dictionary(self.terms)
        # open_index_with_deletions (nidx_tantivy/src/index_reader.rs:39-74): deletions by key, resolved on the device
        self.live = [len(seg.docs) for seg in self.segments]
        if deletions:
            for i in range(len(self.segments)):
                terms = deletion_terms(self.vocab, seqs[i] if seqs is not None else 0, deletions)
                self.live[i] = self.searcher.apply_deletions(i, terms)
        for i, seg in enumerate(self.segments):
            if seg.docs:
                self.searcher.set_fast_field(i, 0, [d.created for d in seg.docs])
                self.searcher.set_fast_field(i, 1, [d.modified for d in seg.docs])

    def fuzzy_terms(self, word: str, prefix: bool) -> List[int]:
        """FuzzyTermQuery's automaton over the text field's dictionary, evaluated on the device."""
        return [int(t) for t in self.searcher.fuzzy_terms(word, prefix) if not self.terms[int(t)].startswith("\x00")]

    def facet_children(self, facet: str) -> List[Tuple[str, int]]:
        """(child path, term id) of every direct child of `facet` present in the index"""
        pre = "\x00label:" + facet.rstrip("/") + "/"
        out = []
        for t, i in self.vocab.ids.items():
            if t.startswith(pre) and "/" not in t[len(pre):]:
                out.append((t[len("\x00label:"):], i))
        return sorted(out)

    def facet_request(self, faceted: Optional[Sequence[str]]):
        """-> (requested valid facets, [(facet, child path, term id)])"""
        facets = [f for f in (faceted or []) if is_valid_facet(f)]
        pairs = [(f, child, tid) for f in facets for child, tid in self.facet_children(f)]
        return facets, pairs

    @staticmethod
    def produce_facets(facets, pairs, counts) -> Dict[str, List[FacetResult]]:
        """produce_facets / facet_count (nidx_text/src/reader.rs:43-62): the 50 most frequent children per facet,
        facets without any hit dropped."""
        out: Dict[str, List[FacetResult]] = {}
        for f in facets:
            rows = [(int(c), child) for (ff, child, _), c in zip(pairs, counts) if ff == f and int(c) > 0]
            rows.sort(key=lambda r: (-r[0], r[1]))
            if rows:
                out[f] = [FacetResult(child, c) for c, child in rows[:50]]
        return out

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
    faceted: Optional[List[str]] = None
    order: Optional[OrderBy] = None
    only_faceted: bool = False


@dataclass
class DocumentResult:
    uuid: str
    field: str
    score: Optional[ResultScore]
    labels: List[str]
    sort_value: Optional[int] = None  # SortValue::Date when ordered by a fast field


@dataclass
class DocumentSearchResponse:
    total: int
    results: List[DocumentResult]
    next_page: bool
    query: str
    facets: Dict[str, List[FacetResult]] = field(default_factory=dict)


# ---- filter expressions (nidx_protos FilterExpression; nidx_text/src/search_query.rs:156-223) -----------------------
@dataclass
class BoolAnd:
    operands: List["FilterExpression"]


@dataclass
class BoolOr:
    operands: List["FilterExpression"]


@dataclass
class BoolNot:
    operand: "FilterExpression"


@dataclass
class ResourceFilter:
    resource_id: str


@dataclass
class FieldFilter:
    field_type: str
    field_id: Optional[str] = None


@dataclass
class ResourceFieldPrefixFilter:
    resource_id: str
    field_type: str
    field_id_prefix: str


@dataclass
class KeywordFilter:
    keyword: str


@dataclass
class DateRangeFilter:
    field: int                    # OrderBy.CREATED / OrderBy.MODIFIED
    since: Optional[int] = None   # inclusive
    until: Optional[int] = None   # inclusive


@dataclass
class FacetFilter:
    facet: str


FilterExpression = object  # any of the classes above


@dataclass
class Security:
    access_groups: List[str] = field(default_factory=list)


@dataclass
class PreFilterRequest:
    security: Optional[Security] = None
    filter_expression: Optional[FilterExpression] = None


@dataclass
class PrefilterResult:
    """PrefilterResult::{All, None, Some(fields)} (nidx_types/src/prefilter.rs)"""
    kind: str                                            # "All" | "None" | "Some"
    fields: List[Tuple[str, str]] = field(default_factory=list)  # (resource uuid, field id) of every matching document


# ---- tantivy's query grammar (tantivy-query-grammar 0.26, restated for the shapes nidx_text sends) --------------------------
class QuerySyntaxError(ValueError):
    """QueryParserError::SyntaxError: the body is then searched as one phrase (adapt_text)."""


@dataclass
class QLeaf:
    text: str = ""
    boost: float = 1.0
    all: bool = False      # `*`
    # `[a TO b]` / `{a TO b}` / `[a TO *]`: (lower token or None, inclusive, upper token or None, inclusive) — a RangeQuery over the
    # text field's terms (byte order), scored ConstScorer(1.0)
    term_range: Optional[Tuple[Optional[str], bool, Optional[str], bool]] = None
    slop: int = 0          # `"a b"~2`


@dataclass
class QNode:
    op: str                # "and" | "or" | "not"
    children: list
    boost: float = 1.0     # `( ... )^boost`: BoostQuery over the group


_SPECIAL = set(' \t\n\r()"\':^{}[]')


def parse_text_query(body: str):
    """-> QLeaf | QNode.  Grammar: clause := ['+' | '-'] (word | "phrase" | '(' query ')' | '*') ['^' number], words may
    carry a `text:` field prefix; clauses are joined by AND / OR / NOT or by juxtaposition (= AND: conjunction by default);
    AND binds tighter than OR.  Raises QuerySyntaxError on what tantivy's parser rejects (unbalanced quotes / parentheses,
    a dangling operator, an unknown field, a range bound that is not one token).  `"a b"~2` is a phrase with slop.
    Ranges: `[a TO b]` inclusive, `{a TO b}` exclusive, mixed brackets, `*` for an open end."""
    pos, n = 0, len(body)

    def skip_ws():
        nonlocal pos
        while pos < n and body[pos].isspace():
            pos += 1

    def parse_boost() -> float:
        nonlocal pos
        if pos < n and body[pos] == "^":
            j = pos + 1
            while j < n and (body[j].isdigit() or body[j] == "."):
                j += 1
            if j == pos + 1:
                raise QuerySyntaxError("boost without a number")
            b = float(body[pos + 1:j])
            pos = j
            return b
        return 1.0

    def parse_atom():
        nonlocal pos
        skip_ws()
        if pos >= n:
            raise QuerySyntaxError("unexpected end")
        c = body[pos]
        if c == "(":
            pos += 1
            node = parse_or()
            skip_ws()
            if pos >= n or body[pos] != ")":
                raise QuerySyntaxError("unbalanced parenthesis")
            pos += 1
            b = parse_boost()
            if b != 1.0:
                if isinstance(node, QLeaf):
                    node = QLeaf(node.text, node.boost * b, node.all, node.term_range, node.slop)
                else:
                    node = QNode(node.op, node.children, node.boost * b)
            return node
        if c == '"':
            j = body.find('"', pos + 1)
            if j < 0:
                raise QuerySyntaxError("unbalanced quote")
            text = body[pos + 1:j]
            pos = j + 1
            slop = 0
            if pos < n and body[pos] == "~":   # "a b"~2: PhraseQuery::set_slop
                j = pos + 1
                while j < n and body[j].isdigit():
                    j += 1
                if j == pos + 1:
                    raise QuerySyntaxError("slop without a number")
                slop = int(body[pos + 1:j])
                pos = j
            return QLeaf(text, parse_boost(), slop=slop)
        if c == "*":
            pos += 1
            return QLeaf("", parse_boost(), all=True)
        if c in "[{":
            lo_incl = c == "["
            pos += 1

            def parse_bound():
                nonlocal pos
                skip_ws()
                if pos >= n:
                    raise QuerySyntaxError("unterminated range")
                if body[pos] == "*":
                    pos += 1
                    return None
                if body[pos] == '"':
                    j = body.find('"', pos + 1)
                    if j < 0:
                        raise QuerySyntaxError("unbalanced quote")
                    raw = body[pos + 1:j]
                    pos = j + 1
                else:
                    j = pos
                    while j < n and not body[j].isspace() and body[j] not in "]}":
                        j += 1
                    raw = body[pos:j]
                    pos = j
                toks = tokenize(raw)
                if len(toks) != 1:   # QueryParserError::RangeMustNotHavePhrase (and an empty bound)
                    raise QuerySyntaxError("a range bound must be exactly one token")
                return toks[0]

            lo = parse_bound()
            skip_ws()
            if not (body.startswith("TO", pos) and pos + 2 < n and body[pos + 2].isspace()):
                raise QuerySyntaxError("range without TO")
            pos += 2
            hi = parse_bound()
            skip_ws()
            if pos >= n or body[pos] not in "]}":
                raise QuerySyntaxError("unterminated range")
            hi_incl = body[pos] == "]"
            pos += 1
            return QLeaf("", parse_boost(), term_range=(lo, lo_incl, hi, hi_incl))
        if c in _SPECIAL or c in "+-":
            raise QuerySyntaxError(f"unexpected {c!r}")
        j = pos
        while j < n and body[j] not in _SPECIAL:
            j += 1
        word = body[pos:j]
        pos = j
        if pos < n and body[pos] == ":":  # field prefix
            if word != "text":
                raise QuerySyntaxError(f"field {word!r} does not exist")
            pos += 1
            return parse_atom()
        return QLeaf(word, parse_boost())

    def parse_clause():
        nonlocal pos
        skip_ws()
        if pos < n and body[pos] in "+-" and pos + 1 < n and not body[pos + 1].isspace():
            neg = body[pos] == "-"
            pos += 1
            atom = parse_atom()
            return QNode("not", [atom]) if neg else atom
        if body.startswith("NOT", pos) and pos + 3 < n and body[pos + 3].isspace():
            pos += 3
            return QNode("not", [parse_clause()])
        return parse_atom()

    def at_operator(word: str) -> bool:
        return body.startswith(word, pos) and (pos + len(word) == n or body[pos + len(word)].isspace() or body[pos + len(word)] == "(")

    def parse_and():
        nonlocal pos
        items = [parse_clause()]
        while True:
            skip_ws()
            if pos >= n or body[pos] == ")" or at_operator("OR"):
                break
            if at_operator("AND"):
                pos += 3
                skip_ws()
                if pos >= n:
                    raise QuerySyntaxError("dangling AND")
            items.append(parse_clause())
        return items[0] if len(items) == 1 else QNode("and", items)

    def parse_or():
        nonlocal pos
        items = [parse_and()]
        while True:
            skip_ws()
            if at_operator("OR"):
                pos += 2
                skip_ws()
                if pos >= n:
                    raise QuerySyntaxError("dangling OR")
                items.append(parse_and())
            else:
                break
        return items[0] if len(items) == 1 else QNode("or", items)

    skip_ws()
    if pos >= n:
        return None
    node = parse_or()
    skip_ws()
    if pos < n:
        raise QuerySyntaxError(f"unexpected {body[pos]!r}")
    return node


def _scaled(leaf: "QLeaf", factor: float) -> "QLeaf":
    return leaf if factor == 1.0 else QLeaf(leaf.text, leaf.boost * factor, leaf.all, leaf.term_range, leaf.slop)


def flatten_conjunction(node):
    """One BooleanQuery level of the boolean tree as (Must leaves, MustNot leaves, [required OR groups], MustNot sub-trees, Must
    sub-trees).  A group member is a leaf or — an AND inside an OR, `a OR (b AND c)` — a (node, boost) pair that becomes a nested
    BooleanQuery (NIDX_BM25_SUBQUERY); a negated expression `NOT (a AND b)`, `NOT (a OR (b c))` and a boosted conjunction
    `(a b)^2` are sub-trees too.  A nested query's own children go through this function again, so the tree may be of any depth.
    A boost on an OR group is carried by its members (tantivy multiplies the group's sum: equal for powers of two, else it may differ
    in the last bit — the oracle follows the mirror)."""
    musts, nots, groups, not_subs, must_subs = [], [], [], [], []

    def or_members(nd, factor, members):
        for c in nd.children:
            if isinstance(c, QLeaf):
                members.append(_scaled(c, factor))
            elif c.op == "or":
                or_members(c, factor * c.boost, members)   # an OR inside an OR is the same group
            else:   # a OR (b AND c), a OR (NOT b)
                members.append((c, factor * (c.boost if c.op == "and" else 1.0)))

    def walk(nd, factor=1.0):
        if isinstance(nd, QLeaf):
            musts.append(_scaled(nd, factor))
        elif nd.op == "and":
            if nd.boost * factor != 1.0:
                must_subs.append((nd, nd.boost * factor))   # (a b)^2: BoostQuery over the conjunction's own sum
            else:
                for ch in nd.children:
                    walk(ch)
        elif nd.op == "not":
            inner = nd.children[0]
            if isinstance(inner, QLeaf):
                nots.append(inner)
            elif inner.op == "or" and all(isinstance(c, QLeaf) for c in inner.children):
                nots.extend(inner.children)   # NOT (a OR b) = NOT a AND NOT b
            elif inner.op == "not":           # NOT (NOT x): a MustNot BooleanQuery[MustNot x], which matches nothing — excludes nothing
                pass
            else:
                not_subs.append(inner)        # NOT (a AND b), NOT (a OR (b AND c)): a MustNot nested BooleanQuery
        elif nd.op == "or":
            members = []
            or_members(nd, nd.boost * factor, members)
            groups.append(members)

    walk(node)
    return musts, nots, groups, not_subs, must_subs


MAX_NESTED_LEAVES = 32   # BM25_MAX_SUBQUERY_LEAVES


class TextSearcher:
    """nidx_text::TextSearcher (lib.rs:178-237) — `search` only."""

    def __init__(self, index: _Index):
        self._index = index

    @classmethod
    def open(cls, segments: Sequence[TextSegment], deleted: Sequence[set] = (), seqs: Optional[Sequence[int]] = None,
             deletions: Sequence[Tuple[str, int]] = ()) -> "TextSearcher":
        """segments + either explicit deleted doc sets, or the index's (key, seq) deletions with every segment's seq."""
        return cls(_Index(segments, deleted, seqs, deletions))

    def close(self):
        self._index.close()

    @staticmethod
    def adapt_text(text: str) -> str:
        """TextReaderService::adapt_text (reader.rs:357-365): a body the QueryParser rejects is searched as ONE phrase, its
        quotes dropped."""
        if text == "":
            return text
        try:
            parse_text_query(text)
            return text
        except QuerySyntaxError:
            return '"' + text.replace('"', "") + '"'

    def _leaf(self, leaf: "QLeaf", occur: int) -> Optional[Clause]:
        """One literal of the grammar through the text field's tokenizer: no token -> no clause, one -> TermQuery with
        frequencies, several -> PhraseQuery (QueryParser::compute_logical_ast_for_leaf)."""
        if leaf.all:
            return Clause(self._index.term(ALL_DOCS), occur, _lib.CONST_SCORE, leaf.boost)   # AllQuery
        if leaf.term_range is not None:
            # RangeQuery over the field's term dictionary: the union of the terms inside the bounds, ConstScorer(1.0) x boost
            ids = terms_in_range(self._index.vocab, *leaf.term_range)
            if not len(ids):
                return Clause(self._index.empty_term, occur, _lib.TF_FREQ, 1.0)   # matches nothing
            return Clause(0, occur, _lib.CONST_SCORE, leaf.boost, term_set=ids)
        words = tokenize(leaf.text)
        if not words:
            return None
        if len(words) > 1:
            return Clause(0, occur, _lib.TF_FREQ, leaf.boost, term_set=[self._index.term(w) for w in words], phrase=True, slop=leaf.slop)
        return Clause(self._index.term(words[0]), occur, _lib.TF_FREQ, leaf.boost)

    def _subquery(self, node, occur: int, boost: float) -> Optional[Clause]:
        """A boolean expression below the level it sits in as a nested BooleanQuery (NIDX_BM25_SUBQUERY) whose leaves are the
        clauses of ITS level — words, phrases, ranges and further nested queries."""
        inner = QNode(node.op, node.children) if isinstance(node, QNode) else node
        leaves, positive = self._boolean(inner)
        if not positive:   # `a OR (NOT b)`: BooleanQuery[MustNot b] matches nothing in tantivy
            return Clause(self._index.empty_term, occur, _lib.TF_FREQ, 1.0) if occur == _lib.OCCUR_MUST else None
        if len(leaves) > MAX_NESTED_LEAVES:
            raise ValueError(f"more than {MAX_NESTED_LEAVES} clauses in one nested boolean expression")
        return Clause(0, occur, _lib.TF_FREQ, boost, subquery=leaves)

    def _boolean(self, ast) -> Tuple[List[Clause], bool]:
        """The clauses of ONE BooleanQuery level (tantivy's QueryParser with set_conjunction_by_default, reader.rs:372-377):
        juxtaposed literals are Must, `+` / `-` prefixes, AND / OR / NOT, parentheses, "phrases"[~slop], ranges, `text:` field
        prefixes and `^boost`; flattened to Must / MustNot clauses and required Should groups (one per OR), whatever nests
        deeper becomes a nested query.  -> (clauses, has a positive clause)"""
        musts, nots, groups, not_subs, must_subs = flatten_conjunction(ast)
        clauses: List[Clause] = []
        positive = False
        for nd, b in must_subs:   # a boosted conjunction: BoostQuery(BooleanQuery[Must ...])
            c = self._subquery(QNode("and", nd.children), _lib.OCCUR_MUST, b)
            if c is not None:
                clauses.append(c)
                positive = True
        for leaf in musts:
            c = self._leaf(leaf, _lib.OCCUR_MUST)
            if c is not None:
                clauses.append(c)
                positive = True
        for g, leaves in enumerate(groups):
            occur = _lib.OCCUR_SHOULD_GROUP + g if g < 8 else _lib.OCCUR_SHOULD   # past 8 groups: a nested query of its own
            members = [self._subquery(l[0], occur, l[1]) if isinstance(l, tuple) else self._leaf(l, occur) for l in leaves]
            members = [m for m in members if m is not None]
            if members:
                clauses += members if g < 8 else [Clause(0, _lib.OCCUR_MUST, _lib.TF_FREQ, 1.0, subquery=members)]
                positive = True
        for leaf in nots:
            c = self._leaf(leaf, _lib.OCCUR_MUST_NOT)
            if c is not None:
                clauses.append(c)
        for nd in not_subs:   # NOT (a AND b)
            c = self._subquery(nd, _lib.OCCUR_MUST_NOT, 1.0)
            if c is not None:
                clauses.append(c)
        return clauses, positive

    def _clauses(self, request: DocumentSearchRequest) -> List[Clause]:
        """create_query (search_query.rs:92-126): Must(main query) + Must(filters).  The main query is the body through
        tantivy's QueryParser (see _boolean)."""
        body = self.adapt_text(request.body)
        clauses: List[Clause] = []
        if body == "":
            clauses.append(Clause(self._index.term(ALL_DOCS), _lib.OCCUR_MUST, _lib.CONST_SCORE, 1.0))  # AllQuery
        else:
            try:
                ast = parse_text_query(body)
            except QuerySyntaxError:  # `parse_query(text).unwrap_or_else(|_| AllQuery)`: cannot happen after adapt_text
                ast = None
            if ast is None:
                clauses.append(Clause(self._index.term(ALL_DOCS), _lib.OCCUR_MUST, _lib.CONST_SCORE, 1.0))
            else:
                clauses, positive = self._boolean(ast)
                if not positive:  # only exclusions (or nothing survived the tokenizer): a BooleanQuery without a positive clause matches nothing
                    clauses.append(Clause(self._index.empty_term, _lib.OCCUR_MUST, _lib.TF_FREQ, 1.0))
        for lab in request.label_filter or []:
            # filter clauses score too in tantivy's BooleanQuery; facet TermQuerys carry no frequencies
            clauses.append(Clause(self._index.term("\x00label:" + lab), _lib.OCCUR_MUST, _lib.TF_BASIC, 1.0))
        return clauses

    # ---- prefilter ------------------------------------------------------------------------------------------------
    def _filter_program(self, expr, ops, lists, ranges, phrases) -> None:
        """filter_to_query (search_query.rs:156-223) flattened to the postfix program of nidx_gpu_bm25_prefilter."""
        ix = self._index

        def push_terms(terms: Sequence[int]) -> None:
            ops.append((_lib.FILTER_PUSH_LISTS, len(lists), len(lists) + len(terms)))
            lists.extend(terms)

        if isinstance(expr, (BoolAnd, BoolOr)):
            if not expr.operands:  # BooleanQuery::intersection / union of nothing matches nothing
                ops.append((_lib.FILTER_PUSH_NONE, 0, 0))
                return
            for i, e in enumerate(expr.operands):
                self._filter_program(e, ops, lists, ranges, phrases)
                if i:
                    ops.append((_lib.FILTER_AND if isinstance(expr, BoolAnd) else _lib.FILTER_OR, 0, 0))
        elif isinstance(expr, BoolNot):
            self._filter_program(expr.operand, ops, lists, ranges, phrases)
            ops.append((_lib.FILTER_NOT, 0, 0))
        elif isinstance(expr, ResourceFilter):
            push_terms([ix.term("\x00uuid:" + expr.resource_id)])
        elif isinstance(expr, FieldFilter):  # field_key (search_query.rs:148-154)
            key = f"/{expr.field_type}/{expr.field_id}" if expr.field_id is not None else f"/{expr.field_type}"
            push_terms([ix.term("\x00field:" + key)])
        elif isinstance(expr, ResourceFieldPrefixFilter):
            # RangeQuery [prefix, prefix with its last byte + 1) over encoded_field_id_bytes: every indexed id with the prefix
            prefix = "\x00fid:" + expr.resource_id + "/" + expr.field_type + "/" + expr.field_id_prefix
            push_terms([i for t, i in ix.vocab.ids.items() if t.startswith(prefix)])
        elif isinstance(expr, KeywordFilter):  # translate_keyword_to_text_query (query_io.rs:22-42)
            words = tokenize(expr.keyword)
            if len(words) <= 1:
                push_terms([ix.term(words[0] if words else expr.keyword)])
            else:
                ops.append((_lib.FILTER_PUSH_PHRASE, len(phrases), 0))
                phrases.append([ix.term(w) for w in words])
        elif isinstance(expr, DateRangeFilter):
            ops.append((_lib.FILTER_PUSH_RANGE, len(ranges), 0))
            ranges.append((expr.field, expr.since, expr.until))
        elif isinstance(expr, FacetFilter):
            push_terms([ix.term("\x00label:" + expr.facet)])
        else:
            raise TypeError(f"not a filter expression: {expr!r}")

    def prefilter(self, request: PreFilterRequest) -> PrefilterResult:
        """TextReaderService::prefilter (reader.rs:148-180): which fields pass the security + filter expression."""
        ops, lists, ranges, phrases = [], [], [], []
        n_sub = 0
        if request.security is not None:  # security_query (search_query.rs:66-90): public OR any of the groups
            groups = [g if g.startswith("/") else "/" + g for g in request.security.access_groups]
            terms = [self._index.term(PUBLIC)] + [self._index.term("\x00group:" + g) for g in groups]
            ops.append((_lib.FILTER_PUSH_LISTS, len(lists), len(lists) + len(terms)))
            lists.extend(terms)
            n_sub += 1
        if request.filter_expression is not None:
            self._filter_program(request.filter_expression, ops, lists, ranges, phrases)
            n_sub += 1
            if n_sub == 2:
                ops.append((_lib.FILTER_AND, 0, 0))
        if n_sub == 0:
            return PrefilterResult("All")
        docaddr, live = self._index.searcher.prefilter(ops, lists, ranges, phrases)
        if docaddr.size == 0:
            return PrefilterResult("None")
        if docaddr.size == live:
            return PrefilterResult("All")
        docs = [self._index.doc(int(a)) for a in docaddr]
        return PrefilterResult("Some", [(d.uuid, d.field) for d in docs])

    def search(self, request: DocumentSearchRequest) -> DocumentSearchResponse:
        k = max(0, int(request.result_per_page))
        facets, pairs = self._index.facet_request(request.faceted)
        fterms = [[tid for _, _, tid in pairs]] if pairs else None
        clauses = self._clauses(request)
        if request.only_faceted:  # "Just a facet search" (reader.rs:403-410)
            r = self._index.searcher.search_batch_ex([clauses], 0, facets=fterms)
            return DocumentSearchResponse(0, [], False, "", self._index.produce_facets(facets, pairs, r["facet_counts"][0] if pairs else []))
        order = request.order
        # TopDocs::with_limit(results + 1) (reader.rs:380,433), by score or by a fast field
        r = self._index.searcher.search_batch_ex([clauses], k + 1, order_field=-1 if order is None else order.field,
                                                order_desc=True if order is None else order.desc, facets=fterms)
        docaddr, score, count, total = r["docaddr"], r["score"], r["count"], r["total"]
        total_ = int(total[0])
        results = []
        for i in range(min(int(count[0]), k)):  # .take(results_per_page), drop `score < min_score` (reader.rs:299-305)
            d = self._index.doc(int(docaddr[0, i]))
            labels = [l for l in d.labels if l.startswith("/l/")]
            if order is not None:  # convert_int_order: no min_score cut, the sort value instead of a score
                results.append(DocumentResult(d.uuid, d.field, None, labels, sort_value=int(r["order_value"][0, i])))
                continue
            s = float(score[0, i])
            if s < request.min_score:
                continue
            results.append(DocumentResult(d.uuid, d.field, ResultScore(s, int(docaddr[0, i])), labels))
        fc = self._index.produce_facets(facets, pairs, r["facet_counts"][0]) if pairs else {}
        return DocumentSearchResponse(total_, results, total_ > k, self.adapt_text(request.body), fc)


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
    faceted: Optional[List[str]] = None
    order: Optional[OrderBy] = None
    only_faceted: bool = False
    # BooleanExpression<String> over label facets (nidx_types/src/query_language.rs): FormulaLiteral / FormulaNot / FormulaOp
    filtering_formula: Optional[object] = None
    filter_or: bool = False   # FilterOperator::Or: the formula and the prefilter are alternatives (search_query.rs:94-98)


@dataclass
class FormulaLiteral:
    label: str


@dataclass
class FormulaNot:
    operand: object


@dataclass
class FormulaOp:
    operator: str              # "and" | "or"
    operands: List[object]


@dataclass
class ParagraphResult:
    uuid: str
    field: str
    paragraph: str
    score: Optional[ResultScore]
    labels: List[str]
    sort_value: Optional[int] = None


@dataclass
class ParagraphSearchResponse:
    total: int
    results: List[ParagraphResult]
    next_page: bool
    query: str
    facets: Dict[str, List[FacetResult]] = field(default_factory=dict)
    fuzzy: bool = False  # the hits come from the fuzzy fallback query


def parse_query(body: str, stop_words: Optional[Set[str]] = None) -> List[Tuple[str, str]]:
    """tokenize_query_infallible + remove_stop_words (query_parser/tokenizer.rs:49-188, query_parser.rs:60-62):
    [(kind, text)] with kind in literal / quoted / excluded.  Grammar: "quoted text", -excluded, literals are runs of
    non-space, non-control, non-quote characters; an unclosed quote is dropped.  Every token then goes through the
    index's tokenizer (punctuation splits, lower case); a quoted run keeps its words joined by one space."""
    raw: List[Tuple[str, str]] = []
    i, n = 0, len(body)

    def literal_char(c: str) -> bool:
        return c.isalnum() or (not c.isspace() and c != '"' and c.isprintable())

    while i < n:
        c = body[i]
        if c.isspace():
            i += 1
        elif c == '"':
            j = body.find('"', i + 1)
            if j > i + 1:  # "xxx"
                if body[i + 1:j].strip():
                    raw.append(("quoted", body[i + 1:j]))
                i = j + 1
            else:  # unclosed quote(s), or an empty pair: dropped
                while i < n and body[i] == '"':
                    i += 1
        elif c == "-" and i + 1 < n and literal_char(body[i + 1]):
            j = i + 1
            while j < n and literal_char(body[j]):
                j += 1
            raw.append(("excluded", body[i + 1:j]))
            i = j
        elif literal_char(c):
            j = i
            while j < n and literal_char(body[j]):
                j += 1
            raw.append(("literal", body[i:j]))
            i = j
        else:
            i += 1
    tokens: List[Tuple[str, str]] = []
    for kind, text in raw:
        words = tokenize(text)
        if kind == "quoted":
            if words:
                tokens.append((kind, " ".join(words)))
        else:
            tokens.extend((kind, w) for w in words)
    if stop_words:
        # remove_stop_words (stop_words.rs): literals that are stop words go, except the last token of the query
        kept = [t for i, t in enumerate(tokens) if t[0] != "literal" or t[1] not in stop_words or i == len(tokens) - 1]
        tokens = kept
    return tokens


MIN_FUZZY_LEN = 3          # fuzzy_parser.rs:35
MIN_FUZZY_PREFIX_LEN = 4   # fuzzy_parser.rs:39
FUZZY_BOOST = 0.5          # search_query.rs:235-239


class ParagraphSearcher:
    """nidx_paragraph::ParagraphSearcher (lib.rs:117-169) — `search`: keyword query, fuzzy fallback, facets, order."""

    def __init__(self, index: _Index, stop_words: Optional[Set[str]] = None):
        self._index = index
        # the reference removes the stop words of eight languages (query_parser/stop_words/*.json, data files that
        # are not copied here): pass the union of those lists to get the same query
        self.stop_words = stop_words

    @classmethod
    def open(cls, segments: Sequence[TextSegment], deleted: Sequence[set] = (), stop_words: Optional[Set[str]] = None,
             seqs: Optional[Sequence[int]] = None, deletions: Sequence[Tuple[str, int]] = ()) -> "ParagraphSearcher":
        return cls(_Index(segments, deleted, seqs, deletions), stop_words)

    def close(self):
        self._index.close()

    def _label(self, label: str, occur: int, boost: float) -> Clause:
        return Clause(self._index.term("\x00label:" + label), occur, _lib.TF_BASIC, boost)   # translate_literal (query_io.rs:22-26)

    def _filter_query(self, request: ParagraphSearchRequest, prefilter: Optional[PrefilterResult], boost: float) -> List[Clause]:
        """filter_query (search_query.rs:88-143): BooleanQuery[(occur, translate_expression(formula)), (occur,
        BooleanQuery[Should SetQuery(field_uuid), Should SetQuery(uuid)])] under Occur::Must, occur = Must (And) / Should (Or).
        Flattened to the kernel's clause family: And of literals -> Must each; Or of literals -> one required Should group;
        Not(literal) -> Must AllQuery + MustNot literal (query_io.rs:28-41); the prefilter's two SetQuerys -> one required
        group of constant-score term sets.  With FilterOperator::Or everything is ONE required group."""
        groups: List[List[Clause]] = []
        out: List[Clause] = []
        G = _lib.OCCUR_SHOULD_GROUP
        next_group = [1]   # group 0 is the keyword group

        def new_group() -> int:
            """-> the occur code of the next required Should group; past the eighth: OCCUR_SHOULD, and the caller wraps the
            members into a nested Should-only query under Occur::Must (which is what the group is in tantivy anyway)."""
            g = next_group[0]
            next_group[0] += 1
            return G + g if g < 8 else _lib.OCCUR_SHOULD

        def conj(expr):   # expr under Occur::Must
            if isinstance(expr, FormulaLiteral):
                out.append(self._label(expr.label, _lib.OCCUR_MUST, boost))
            elif isinstance(expr, FormulaNot):
                out.append(Clause(self._index.term(ALL_DOCS), _lib.OCCUR_MUST, _lib.CONST_SCORE, boost))
                out.extend(negated(expr.operand))
            elif isinstance(expr, FormulaOp) and expr.operator == "and":
                for e in expr.operands:
                    conj(e)
            elif isinstance(expr, FormulaOp) and expr.operator == "or":
                occur = new_group()
                members: List[Clause] = []
                disj(expr, occur, members)
                out.extend(members if occur != _lib.OCCUR_SHOULD else [Clause(0, _lib.OCCUR_MUST, _lib.TF_BASIC, 1.0, subquery=members)])
            else:
                raise TypeError(f"not a formula: {expr!r}")

        def negated(operand) -> List[Clause]:
            """The MustNot half of translate_not (query_io.rs:28-41: Not(x) = BooleanQuery[Must AllQuery, MustNot x])."""
            if isinstance(operand, FormulaLiteral):
                return [self._label(operand.label, _lib.OCCUR_MUST_NOT, boost)]
            if isinstance(operand, FormulaOp) and operand.operator == "or" and all(isinstance(e, FormulaLiteral) for e in operand.operands):
                return [self._label(e.label, _lib.OCCUR_MUST_NOT, boost) for e in operand.operands]   # Not(Or(..)) = none of them
            return [nested(operand, _lib.OCCUR_MUST_NOT)]   # Not(And(..)), Not(Not(..)), Not(Or(And(..), ..))

        def nested(expr, occur: int) -> Clause:
            """An expression below the level it sits in as a nested BooleanQuery (translate_expression recurses the same way): a
            conjunction's operands are Must leaves, a negation is Must AllQuery + MustNot, a disjunction is a required Should
            group; operands that are not literals nest again."""
            leaves: List[Clause] = []
            groups_in = [0]

            def inner(e):
                if isinstance(e, FormulaLiteral):
                    leaves.append(self._label(e.label, _lib.OCCUR_MUST, boost))
                elif isinstance(e, FormulaNot):
                    leaves.append(Clause(self._index.term(ALL_DOCS), _lib.OCCUR_MUST, _lib.CONST_SCORE, boost))
                    leaves.extend(negated(e.operand))
                elif isinstance(e, FormulaOp) and e.operator == "and":
                    for x in e.operands:
                        inner(x)
                elif isinstance(e, FormulaOp) and e.operator == "or":
                    g = groups_in[0]
                    groups_in[0] += 1
                    members: List[Clause] = []
                    disj(e, G + g if g < 8 else _lib.OCCUR_SHOULD, members)
                    leaves.extend(members if g < 8 else [Clause(0, _lib.OCCUR_MUST, _lib.TF_BASIC, 1.0, subquery=members)])
                else:
                    raise TypeError(f"not a formula: {e!r}")

            inner(expr)
            if len(leaves) > MAX_NESTED_LEAVES:
                raise ValueError(f"more than {MAX_NESTED_LEAVES} literals in one level of a nested formula")
            return Clause(0, occur, _lib.TF_BASIC, 1.0, subquery=leaves)

        def disj(expr, occur: int, members: List[Clause]):   # expr as members of one required group
            if isinstance(expr, FormulaLiteral):
                members.append(self._label(expr.label, occur, boost))
            elif isinstance(expr, FormulaOp) and expr.operator == "or":
                for e in expr.operands:
                    disj(e, occur, members)
            else:   # And(..) / Not(..) below an Or
                members.append(nested(expr, occur))

        def prefilter_sets(occur: int, out: List[Clause]):
            fields = sorted({"\x00fid:" + rid + "/" + fid.lstrip("/") for rid, fid in prefilter.fields if fid is not None})
            resources = sorted({"\x00uuid:" + rid for rid, fid in prefilter.fields if fid is None})
            for keys in (fields, resources):   # SetQuery = TermSetQuery: ConstScorer over the union of the terms
                if keys:
                    terms = [self._index.term(t) for t in keys]
                    out.append(Clause(0, occur, _lib.CONST_SCORE, boost, term_set=terms))

        some = prefilter is not None and prefilter.kind == "Some" and prefilter.fields
        if request.filter_or and (request.filtering_formula is not None or some):
            g = new_group()
            if request.filtering_formula is not None:
                disj(request.filtering_formula, g, out)
            if some:
                prefilter_sets(g, out)
        else:
            if request.filtering_formula is not None:
                conj(request.filtering_formula)
            if some:
                g = new_group()
                members: List[Clause] = []
                prefilter_sets(g, members)
                out.extend(members if g != _lib.OCCUR_SHOULD else [Clause(0, _lib.OCCUR_MUST, _lib.TF_BASIC, 1.0, subquery=members)])
        return out

    def _filters(self, request: ParagraphSearchRequest, prefilter: Optional[PrefilterResult], boost: float) -> List[Clause]:
        # the prefilter is per request (it carries the field / security verdict): it travels as an argument, never through
        # the searcher, which concurrent requests share
        musts = self._filter_query(request, prefilter, boost)
        for lab in request.label_filter or []:
            musts.append(Clause(self._index.term("\x00label:" + lab), _lib.OCCUR_MUST, _lib.TF_BASIC, boost))
        if not request.with_duplicates:  # Must TermQuery(repeated_in_field = 0, Basic) (search_query.rs:218-223)
            musts.append(Clause(self._index.term(NOT_REPEATED), _lib.OCCUR_MUST, _lib.TF_BASIC, boost))
        return musts

    def _tokens(self, request: ParagraphSearchRequest) -> List[Tuple[str, str]]:
        return parse_query(request.body, self.stop_words)

    def _word_or_phrase(self, kind: str, text: str, boost: float) -> Clause:
        """parse_literal / parse_quoted (keyword_parser.rs:62-91): a word is TermQuery(Basic); a quote of several words is
        PhraseQuery(words)"""
        words = text.split(" ")
        if kind == "quoted" and len(words) > 1:
            return Clause(0, _lib.OCCUR_SHOULD_GROUP, _lib.TF_FREQ, boost, term_set=[self._index.term(w) for w in words], phrase=True)
        return Clause(self._index.term(text), _lib.OCCUR_SHOULD_GROUP, _lib.TF_BASIC, boost)

    def _clauses(self, request: ParagraphSearchRequest, prefilter: Optional[PrefilterResult] = None) -> List[Clause]:
        """The keyword query (keyword_parser.rs:27-105 under search_query.rs:185-243)."""
        tokens = self._tokens(request)
        clauses = []
        if not tokens:  # parse_keyword_query: no subqueries => AllQuery
            clauses.append(Clause(self._index.term(ALL_DOCS), _lib.OCCUR_MUST, _lib.CONST_SCORE, 1.0))
        # TermQuery(text, IndexRecordOption::Basic) per literal / one-word quote, Occur::Should, as a required group: the
        # keyword BooleanQuery sits under Occur::Must next to the filters, so a paragraph has to match one of its words
        should = [self._excluded(w, 1.0) if kind == "excluded" else self._word_or_phrase(kind, w, 1.0) for kind, w in tokens]
        return clauses + should + self._filters(request, prefilter, 1.0)

    def _excluded(self, word: str, boost: float) -> Clause:
        """parse_excluded (keyword_parser.rs:93-105): Should(BooleanQuery[Must AllQuery, MustNot term]) = every paragraph
        that does not contain the word, scored AllQuery's 1.0: the complement of the term's posting list, on the device."""
        return Clause(0, _lib.OCCUR_SHOULD_GROUP, _lib.CONST_SCORE, boost, term_set=[self._index.term(word)], complement=True)

    def _fuzzy_clauses(self, request: ParagraphSearchRequest, prefilter: Optional[PrefilterResult] = None) -> List[Clause]:
        """The fuzzy query (fuzzy_parser.rs:52-123 under search_query.rs:200-240)."""
        tokens = self._tokens(request)
        some = prefilter is not None and prefilter.kind == "Some" and prefilter.fields
        filters_present = bool(request.label_filter) or not request.with_duplicates or request.filtering_formula is not None or bool(some)
        boost = FUZZY_BOOST if filters_present else 1.0  # BoostQuery(0.5) only wraps a multi-clause Boolean (:229-240)
        last_literal = max((i for i, t in enumerate(tokens) if t[0] == "literal"), default=None)
        clauses = []
        if not tokens:
            clauses.append(Clause(self._index.term(ALL_DOCS), _lib.OCCUR_MUST, _lib.CONST_SCORE, boost))
        for i, (kind, w) in enumerate(tokens):
            if kind == "excluded":
                clauses.append(self._excluded(w, boost))
                continue
            if kind == "quoted" or len(w.encode("utf-8")) < MIN_FUZZY_LEN:  # quotes stay exact; too short to be fuzzy
                clauses.append(self._word_or_phrase(kind, w, boost))
                continue
            prefix = i == last_literal and len(w.encode("utf-8")) >= MIN_FUZZY_PREFIX_LEN
            members = self._index.fuzzy_terms(w, prefix) or [self._index.empty_term]
            clauses.append(Clause(0, _lib.OCCUR_SHOULD_GROUP, _lib.CONST_SCORE, boost, term_set=members))
        return clauses + self._filters(request, prefilter, boost)

    def _run(self, request: ParagraphSearchRequest, clauses: List[Clause], fuzzy: bool) -> ParagraphSearchResponse:
        """Searcher::do_search (reader.rs:244-348) + the response assembly (search_response.rs:218-311)."""
        k = max(0, int(request.result_per_page))
        facets, pairs = self._index.facet_request(request.faceted)
        fterms = [[tid for _, _, tid in pairs]] if pairs else None
        if request.only_faceted:
            r = self._index.searcher.search_batch_ex([clauses], 0, facets=fterms)
            return ParagraphSearchResponse(0, [], False, "", self._index.produce_facets(facets, pairs, r["facet_counts"][0] if pairs else []), fuzzy)
        order = request.order
        after = [request.search_after] if request.search_after is not None and order is None else None
        r = self._index.searcher.search_batch_ex([clauses], k + 1, after, order_field=-1 if order is None else order.field,
                                                order_desc=True if order is None else order.desc, facets=fterms)
        docaddr, score, total = r["docaddr"], r["score"], r["total"]
        obtained = int(r["count"][0])
        fc = self._index.produce_facets(facets, pairs, r["facet_counts"][0]) if pairs else {}
        results = []
        if order is not None:  # SearchIntResponse: no min_score, next_page = total > requested
            for i in range(min(obtained, k)):
                d = self._index.doc(int(docaddr[0, i]))
                results.append(ParagraphResult(d.uuid, d.field, d.text, None, list(d.labels), sort_value=int(r["order_value"][0, i])))
            return ParagraphSearchResponse(int(total[0]), results, int(total[0]) > k, request.body, fc, fuzzy)
        scores = [float(score[0, i]) for i in range(obtained)]
        # search_response.rs:218-311: next_page counts scores above min_score, results stop at the first below it
        next_page = sum(1 for s in scores if s > request.min_score) > k
        for i in range(min(obtained, k)):
            if scores[i] < request.min_score:
                break
            d = self._index.doc(int(docaddr[0, i]))
            results.append(ParagraphResult(d.uuid, d.field, d.text, ResultScore(scores[i], int(docaddr[0, i])), list(d.labels)))
        return ParagraphSearchResponse(int(total[0]), results, next_page, request.body, fc, fuzzy)

    def search(self, request: ParagraphSearchRequest, prefilter: Optional[PrefilterResult] = None) -> ParagraphSearchResponse:
        """ParagraphReaderService::search (reader.rs:104-139): the keyword query first; when it finds nothing (and results
        were asked for, with min_score == 0) the fuzzy query is run instead.  `prefilter` = the text index's verdict on the
        request's field filters (PrefilterResult::{All, None, Some}); None finds nothing by construction."""
        if prefilter is not None and prefilter.kind == "None":
            return ParagraphSearchResponse(0, [], False, request.body, {}, False)
        response = self._run(request, self._clauses(request, prefilter), False)
        if not response.results and request.result_per_page > 0 and request.min_score == 0.0 and not request.only_faceted:
            response = self._run(request, self._fuzzy_clauses(request, prefilter), True)
        return response
