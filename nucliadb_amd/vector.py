"""Host-side mirror of the reference's `nidx_vector` public interface over libnidx_gpu.

Same names, argument meaning and error behaviour as the Rust crate so the parity tests read like
the reference's own (`nidx/nidx_vector/tests/*.rs`):

    VectorConfig            nidx_vector/src/config.rs:102-124
    Elem, segment_create    nidx_vector/src/data_types.rs / segment.rs:199-239 (in-memory segment)
    VectorSearchRequest     nidx_vector/src/request_types.rs:19-35
    PrefilterResult/FieldId nidx_types/src/prefilter.rs:23-43
    VectorSearcher          nidx_vector/src/lib.rs:120-148  (open / search / space_usage)

Everything that touches vector data (similarity, HNSW traversal, brute-force scan, top-k) runs in
the HIP kernels behind the C ABI; this module only keeps what the reference keeps host-side: keys,
labels, sentence metadata, deletions -> alive bitsets, label/key-prefix formulas -> filter bitsets.
"""
from __future__ import annotations

import ctypes as C
import uuid as _uuid
from dataclasses import dataclass, field
from enum import Enum
from typing import Iterable, List, Optional, Sequence, Tuple, Union

import numpy as np

from . import _lib
from ._lib import NidxGpuError  # noqa: F401  (re-exported)


class Similarity(Enum):
    Dot = 0
    Cosine = 1


class VectorCardinality(Enum):
    Single = 0
    Multi = 1


@dataclass
class VectorConfig:
    """nidx_vector::config::VectorConfig (config.rs:102-124): the fields the hot path reads."""

    dimension: int
    similarity: Similarity = Similarity.Dot
    normalize_vectors: bool = False
    vector_cardinality: VectorCardinality = VectorCardinality.Single
    flags: List[str] = field(default_factory=list)  # config.rs:25-30; "disable_rabitq_search" is the one the search path reads

    DISABLE_RABITQ_SEARCH = "disable_rabitq_search"

    @classmethod
    def for_paragraphs(cls, dimension: int) -> "VectorConfig":
        return cls(dimension=dimension)

    def quantizable_vectors(self) -> bool:
        """config.rs:170-173"""
        return self.similarity == Similarity.Dot and self.dimension % 64 == 0

    def to_c(self) -> _lib.VectorConfigC:
        flags = _lib.CONFIG_DISABLE_RABITQ_SEARCH if self.DISABLE_RABITQ_SEARCH in self.flags else 0
        return _lib.VectorConfigC(self.dimension, self.similarity.value, int(self.normalize_vectors), self.vector_cardinality.value, flags)


@dataclass
class Elem:
    """One indexed sentence (data_types / indexer.rs:94-145): key, vector, labels, metadata."""

    key: str
    vector: Sequence[float]
    labels: List[str] = field(default_factory=list)
    metadata: bytes = b""


# ---- boolean filter expressions (nidx_types/src/query_language.rs:23-40) ---------------------------
@dataclass
class Literal:
    label: str


@dataclass
class Not:
    operand: "BooleanExpression"


@dataclass
class And:
    operands: List["BooleanExpression"]


@dataclass
class Or:
    operands: List["BooleanExpression"]


BooleanExpression = Union[Literal, Not, And, Or]


class FilterOperator(Enum):
    And = 0
    Or = 1


@dataclass
class FieldId:
    resource_id: _uuid.UUID
    field_id: Optional[str] = None  # e.g. "/a/title"; None = every field of the resource


class PrefilterResult:
    """nidx_types::prefilter::PrefilterResult: None_ / All / Some(fields)."""

    def __init__(self, kind: str, fields: Optional[List[FieldId]] = None):
        self.kind = kind
        self.fields = fields or []

    @classmethod
    def none(cls) -> "PrefilterResult":
        return cls("none")

    @classmethod
    def all(cls) -> "PrefilterResult":
        return cls("all")

    @classmethod
    def some(cls, fields: List[FieldId]) -> "PrefilterResult":
        return cls("some", fields)


PrefilterResult.All = PrefilterResult.all()  # type: ignore[attr-defined]
PrefilterResult.None_ = PrefilterResult.none()  # type: ignore[attr-defined]


@dataclass
class VectorSearchRequest:
    vector: Sequence[float] = ()
    result_per_page: int = 0
    with_duplicates: bool = False
    vector_set: str = ""
    min_score: float = 0.0
    filtering_formula: Optional[BooleanExpression] = None
    segment_filtering_formula: Optional[BooleanExpression] = None
    filter_operator: FilterOperator = FilterOperator.And


@dataclass
class DocumentScored:
    """nodereader.proto DocumentScored (:130-136)."""

    doc_id: str
    score: float
    metadata: Optional[bytes]
    labels: List[str]


@dataclass
class VectorSearchResponse:
    documents: List[DocumentScored]


# ---- segments ------------------------------------------------------------------------------------------
def _key_norm(key: str) -> str:
    """Paragraph keys start with a hyphenated uuid; the inverted index stores it in simple form."""
    head, sep, rest = key.partition("/")
    try:
        head = _uuid.UUID(head).hex
    except ValueError:
        pass
    return head + sep + rest


def _field_key(key: str) -> Optional[str]:
    """FieldKey::from_field_id (utils.rs:84-115) in text form: `uuidhex/type/name`, a bare `uuidhex` for a key that is only
    a resource id, None when the uuid does not parse or a type comes without a name."""
    parts = key.split("/")
    try:
        rid = _uuid.UUID(parts[0]).hex
    except ValueError:
        return None
    if len(parts) == 1:
        return rid
    if len(parts) == 2:
        return None
    return f"{rid}/{parts[1]}/{parts[2]}"


class VectorSegment:
    """An in-memory vector segment: what segment::create writes to disk (segment.rs:199-239) minus
    the files.  `graph` is an hnsw.graph image (DiskHnswV2) or None."""

    def __init__(self, keys: List[str], vectors: np.ndarray, labels: List[List[str]], metadata: List[bytes],
                 tags: Optional[set] = None, graph: Optional[bytes] = None, graph_edges: Optional[np.ndarray] = None,
                 graph_nodes: int = 0, quantized: Optional[np.ndarray] = None, para_of_vec: Optional[np.ndarray] = None):
        # VectorCardinality::Multi: paragraph (= key index) of every vector row, non-decreasing; None = one vector per key
        self.para_of_vec = None if para_of_vec is None else np.ascontiguousarray(para_of_vec, dtype=np.uint32)
        # vectors.quant: [records][dimension/8 + 8] RaBitQ records (None = the store has no quantized vectors)
        self.quantized = None if quantized is None else np.ascontiguousarray(quantized, dtype=np.uint8)
        self.graph_edges = None if graph_edges is None else np.ascontiguousarray(graph_edges, dtype=np.float32)
        self.graph_nodes = graph_nodes  # 0 = the image covers every vector; else only the first graph_nodes (merge reuse)
        self.keys = keys
        self.vectors = np.ascontiguousarray(vectors, dtype=np.float32)
        self.labels = labels
        self.metadata = metadata
        self.tags = set(tags or ())
        self.graph = graph
        self.records = len(keys)
        self._norm_keys = [_key_norm(k) for k in keys]
        self._label_index: dict = {}
        for i, ls in enumerate(labels):
            for lab in ls:
                self._label_index.setdefault(lab, []).append(i)
        # posting lists as the reference's inverted indexes key them (inverted_index/paragraph.rs:63-103):
        #   labels:  labels_key(l) = l[1:] + "/"   (prefix-searchable: a label matches its children)
        #   fields:  FieldKey "uuid/type/name" of the paragraph key
        lists: dict = {}
        for i, ls in enumerate(labels):
            for lab in ls:
                lists.setdefault("L:" + lab[1:] + "/", []).append(i)
        self._field_keys = [_field_key(k) for k in keys]
        for i, fk in enumerate(self._field_keys):
            if fk is not None:
                lists.setdefault("F:" + fk, []).append(i)
        self.list_keys = sorted(lists)
        self.list_id = {k: j for j, k in enumerate(self.list_keys)}
        self.list_offsets = np.zeros(len(self.list_keys) + 1, dtype=np.uint64)
        for j, k in enumerate(self.list_keys):
            self.list_offsets[j + 1] = self.list_offsets[j] + len(lists[k])
        self.list_ids = np.array([i for k in self.list_keys for i in lists[k]], dtype=np.uint32)

    def lists_for(self, expr) -> List[int]:
        """The string -> posting-list lookup the FSTs do (label.fst prefix search, field.fst lookups)."""
        if isinstance(expr, Literal):
            p = "L:" + expr.label[1:] + "/"
            return [self.list_id[k] for k in self.list_keys if k.startswith(p)]
        out = []
        for pre in expr.prefixes:  # _KeyPrefixSet: "{uuid_simple}{field_id}" or "{uuid_simple}"
            key = "F:" + pre
            out += [self.list_id[k] for k in self.list_keys if k == key or k.startswith(key + "/")]
        return sorted(set(out))

    @staticmethod
    def atom_queries(e) -> List[Tuple[bytes, int]]:
        """The key-table lookups of one atom: (query bytes, is_prefix)."""
        if isinstance(e, Literal):
            return [(("L:" + e.label[1:] + "/").encode("utf-8"), 1)]
        out = []
        for pre in e.prefixes:  # "{uuid_simple}{field_id}": that field; "{uuid_simple}": every field of the resource
            out.append((("F:" + pre).encode("utf-8"), 0))
            out.append((("F:" + pre + "/").encode("utf-8"), 1))
        return out

    def compile(self, expr, lookup=None):
        """BooleanExpression -> postfix program for nidx_gpu_vector_search_filtered.  lookup: a function resolving a batch of
        (query bytes, is_prefix) against this segment's key table on the device (VectorSearcher._lookup); None = the host dict."""
        ops, lists = [], []
        resolved = {}
        if lookup is not None:
            atoms = []

            def collect(e):
                if isinstance(e, (Literal, _KeyPrefixSet)):
                    atoms.append(e)
                elif isinstance(e, Not):
                    collect(e.operand)
                elif isinstance(e, (And, Or)):
                    for o in e.operands:
                        collect(o)

            collect(expr)
            queries = [q for a in atoms for q in self.atom_queries(a)]
            ranges = lookup(queries)   # ONE device call for every atom of the formula
            at = 0
            for a in atoms:
                nq = len(self.atom_queries(a))
                ids = sorted({j for f, l in ranges[at: at + nq] for j in range(f, l)})
                at += nq
                resolved[id(a)] = ids

        def emit(e):
            if isinstance(e, (Literal, _KeyPrefixSet)):
                ids = resolved[id(e)] if lookup is not None else self.lists_for(e)
                ops.append((_lib.FILTER_PUSH_LISTS, len(lists), len(lists) + len(ids)))
                lists.extend(ids)
            elif isinstance(e, Not):
                emit(e.operand)
                ops.append((_lib.FILTER_NOT, 0, 0))
            elif isinstance(e, (And, Or)):
                if not e.operands:
                    ops.append((_lib.FILTER_PUSH_ALL if isinstance(e, And) else _lib.FILTER_PUSH_NONE, 0, 0))
                    return
                emit(e.operands[0])
                for o in e.operands[1:]:
                    emit(o)
                    ops.append((_lib.FILTER_AND if isinstance(e, And) else _lib.FILTER_OR, 0, 0))
            else:
                raise TypeError(f"unknown expression {e!r}")

        emit(expr)
        return ops, lists

    # ParagraphInvertedIndexes::filter (inverted_index/paragraph.rs:124-184) on the host
    def _eval(self, expr) -> np.ndarray:
        n = self.records
        if isinstance(expr, (Literal, _KeyPrefixSet)):
            # an atom is the union of the posting lists the FST lookup returns (label: prefix search at a
            # path boundary; field keys: the field, or every field of the resource)
            m = np.zeros(n, dtype=bool)
            for j in self.lists_for(expr):
                m[self.list_ids[int(self.list_offsets[j]): int(self.list_offsets[j + 1])]] = True
            return m
        if isinstance(expr, Not):
            return ~self._eval(expr.operand)
        if isinstance(expr, And):
            m = np.ones(n, dtype=bool)
            for o in expr.operands:
                m &= self._eval(o)
            return m
        if isinstance(expr, Or):
            m = np.zeros(n, dtype=bool)
            for o in expr.operands:
                m |= self._eval(o)
            return m
        raise TypeError(f"unknown expression {expr!r}")

    def ids_for_deletion_key(self, key: str) -> List[int]:
        """apply_deletions' lookup (segment.rs:428-445): a resource uuid or `uuid/type/name`, as
        field_index.get_prefix(FieldKey::from_field_id(key)) (inverted_index/paragraph.rs:118-120) — a BYTE prefix of the
        indexed `uuid type/name` keys, so deleting `uuid/t/title` also reaches a field named `title2` of that resource."""
        parts = key.split("/")
        try:
            rid = _uuid.UUID(parts[0]).hex
        except ValueError:
            return []
        if len(parts) == 1:
            prefix = rid + "/"
        elif len(parts) >= 3:
            prefix = f"{rid}/{parts[1]}/{parts[2]}"
        else:
            return []
        return [i for i, fk in enumerate(self._field_keys) if fk is not None and (fk + "/").startswith(prefix)]

    # ---- segment directories (data_store/v2, hnsw/disk/v2) ----------------------------------------------------------
    def save(self, path: str, dimension: Optional[int] = None) -> None:
        """segment::create's file output (segment.rs:199-239): vectors.bin, paragraphs.bin/.pos, and — when present —
        vectors.quant and hnsw.graph/.edges, through nidx_gpu_segment_dir_write."""
        if self.graph is not None and self.graph_nodes:
            raise NidxGpuError(_lib.NIDX_ERR_INVALID_ARGUMENT, "the graph covers only part of the segment: extend it before saving")
        D = int(dimension or self.vectors.shape[1])

        def table(items: List[bytes]):
            offs = np.zeros(len(items) + 1, np.uint64)
            offs[1:] = np.cumsum([len(b) for b in items])
            return np.frombuffer(b"".join(items) or b"\0", np.uint8), offs

        keys, key_offs = table([k.encode() for k in self.keys])
        labs, lab_offs = table([l.encode() for ls in self.labels for l in ls])
        pl_offs = np.zeros(self.records + 1, np.uint64)
        pl_offs[1:] = np.cumsum([len(ls) for ls in self.labels])
        meta, meta_offs = table([bytes(m or b"") for m in self.metadata])
        c = _lib.SegmentDirContentsC()
        c.dimension, c.n_vectors, c.n_paragraphs = D, self.vectors.shape[0], self.records
        c.vectors = self.vectors.ctypes.data if self.vectors.size else None
        c.paragraph_of_vector = None if self.para_of_vec is None else self.para_of_vec.ctypes.data
        c.keys, c.key_offsets = keys.ctypes.data, key_offs.ctypes.data
        c.labels, c.label_offsets, c.paragraph_label_offsets = labs.ctypes.data, lab_offs.ctypes.data, pl_offs.ctypes.data
        c.metadata, c.metadata_offsets = meta.ctypes.data, meta_offs.ctypes.data
        graph = np.frombuffer(self.graph, np.uint8) if self.graph else None
        if graph is not None:
            edges = self.graph_edges if self.graph_edges is not None else np.zeros(0, np.float32)
            c.hnsw_graph, c.hnsw_graph_len = graph.ctypes.data, graph.size
            c.hnsw_edges, c.n_hnsw_edges = (edges.ctypes.data if edges.size else None), edges.size
        if self.quantized is not None and self.quantized.size:
            c.quantized, c.quantized_len = self.quantized.ctypes.data, self.quantized.size
        _lib.check(_lib.lib().nidx_gpu_segment_dir_write(path.encode(), C.byref(c)))

    @classmethod
    def load(cls, path: str, dimension: int, tags: Optional[set] = None) -> "VectorSegment":
        """segment::open's file side (segment.rs:39-90) through nidx_gpu_segment_dir_open, materialised as host arrays."""
        with SegmentDir(path, dimension) as d:
            return d.to_segment(tags)


class SegmentDir:
    """A mapped segment directory (nidx_gpu_segment_dir_*): the zero-copy views nidx_gpu_vector_open consumes, the rebuilt
    inverted indexes and the paragraph records."""

    def __init__(self, path: str, dimension: int):
        self.dimension = dimension
        self._h = C.c_void_p()
        _lib.check(_lib.lib().nidx_gpu_segment_dir_open(path.encode(), dimension, C.byref(self._h)))

    def close(self):
        if self._h:
            _lib.lib().nidx_gpu_segment_dir_close(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def indexes_from_files(self) -> bool:
        """True: the posting lists were read from field.fst / label.fst / index.map; False: rebuilt from the paragraphs."""
        return _lib.lib().nidx_gpu_segment_dir_index_source(self._h) == 1

    def segment_c(self) -> "_lib.VectorSegmentC":
        s = _lib.VectorSegmentC()
        _lib.check(_lib.lib().nidx_gpu_segment_dir_segment(self._h, C.byref(s)))
        return s

    def filter_index_c(self) -> "_lib.FilterIndexC":
        fi = _lib.FilterIndexC()
        _lib.check(_lib.lib().nidx_gpu_segment_dir_filter_index(self._h, C.byref(fi)))
        return fi

    def lists(self, kind: int, key: str, prefix: bool = False) -> range:
        """Posting-list ids the FST lookup selects (label prefix search / field get / field get_prefix)."""
        k = key.encode()
        first, count = C.c_uint32(0), C.c_uint32(0)
        _lib.check(_lib.lib().nidx_gpu_segment_dir_lists(self._h, kind, k, len(k), int(prefix), C.byref(first), C.byref(count)))
        return range(first.value, first.value + count.value)

    def apply_deletions(self, keys: Sequence[str], alive: Optional[np.ndarray] = None) -> np.ndarray:
        """OpenSegment::apply_deletions through nidx_gpu_segment_dir_apply_deletions: the alive mask (bool per stored
        paragraph) after deleting `keys` (resource uuids or uuid/type/name field ids)."""
        n = self.segment_c().n_paragraphs
        bits = _bitset(np.ones(n, bool) if alive is None else np.asarray(alive, dtype=bool))
        if len(bits) == 0:
            return np.zeros(0, bool)
        enc = [k.encode() for k in keys]
        arr = (C.c_char_p * max(1, len(enc)))(*enc)
        lens = (C.c_uint32 * max(1, len(enc)))(*[len(e) for e in enc])
        cleared = C.c_uint32()
        _lib.check(_lib.lib().nidx_gpu_segment_dir_apply_deletions(self._h, arr, lens, len(enc), bits.ctypes.data, C.byref(cleared)))
        mask = np.unpackbits(bits.view(np.uint8), bitorder="little")[:n].astype(bool)
        assert int(cleared.value) == int((np.ones(n, bool) if alive is None else np.asarray(alive, dtype=bool)).sum() - mask.sum())
        return mask

    def posting_list(self, list_id: int) -> np.ndarray:
        fi = self.filter_index_c()
        offs = np.ctypeslib.as_array(C.cast(fi.list_offsets, C.POINTER(C.c_uint64)), (fi.n_lists + 1,))
        b, e = int(offs[list_id]), int(offs[list_id + 1])
        if e == b:
            return np.zeros(0, np.uint32)
        return np.ctypeslib.as_array(C.cast(fi.paragraph_ids, C.POINTER(C.c_uint32)), (int(offs[-1]),))[b:e].copy()

    def paragraph(self, addr: int) -> Tuple[str, List[str], bytes, int, int]:
        """(key, labels, metadata, first_vector, num_vectors) of one StoredParagraph."""
        p = _lib.ParagraphC()
        _lib.check(_lib.lib().nidx_gpu_segment_dir_paragraph(self._h, addr, C.byref(p)))
        labels = []
        for i in range(p.n_labels):
            ptr, n = C.c_void_p(), C.c_uint32(0)
            _lib.check(_lib.lib().nidx_gpu_segment_dir_paragraph_label(self._h, addr, i, C.byref(ptr), C.byref(n)))
            labels.append(C.string_at(ptr, n.value).decode())
        return (C.string_at(p.key, p.key_len).decode(), labels, C.string_at(p.metadata, p.metadata_len) if p.metadata_len else b"",
                p.first_vector, p.num_vectors)

    def to_segment(self, tags: Optional[set] = None) -> VectorSegment:
        s = self.segment_c()
        D = self.dimension
        n, npar = s.n_vectors, s.n_paragraphs
        if n:
            raw = np.ctypeslib.as_array(C.cast(s.vectors, C.POINTER(C.c_uint8)), (n * s.row_stride_bytes,)).reshape(n, s.row_stride_bytes)
            vectors = raw[:, : D * 4].copy().view(np.float32).reshape(n, D)
            pov = raw[:, D * 4: D * 4 + 4].copy().view(np.uint32).reshape(n)
        else:
            vectors, pov = np.zeros((0, D), np.float32), np.zeros(0, np.uint32)
        paras = [self.paragraph(a) for a in range(npar)]
        single = n == npar and bool(np.array_equal(pov, np.arange(n, dtype=np.uint32)))
        graph = C.string_at(s.hnsw_graph, s.hnsw_graph_len) if s.hnsw_graph_len else None
        edges = np.ctypeslib.as_array(C.cast(s.hnsw_edges, C.POINTER(C.c_float)), (s.n_hnsw_edges,)).copy() if s.n_hnsw_edges else None
        quant = None
        if s.quantized_len:
            quant = np.ctypeslib.as_array(C.cast(s.quantized, C.POINTER(C.c_uint8)), (s.quantized_len,)).copy().reshape(n, -1)
        return VectorSegment([p[0] for p in paras], vectors, [p[1] for p in paras], [p[2] for p in paras], tags, graph=graph,
                             graph_edges=edges, quantized=quant, para_of_vec=None if single else pov)


def segment_dir_merge(path: str, dimension: int, operands: Sequence[Tuple["SegmentDir", Optional[np.ndarray]]]):
    """segment::merge's file output through nidx_gpu_segment_dir_merge: `operands` = [(open SegmentDir, alive mask or None)].
    Returns (records, vectors, graph_nodes, has_quantized); graph_nodes > 0 = the largest operand's hnsw.graph was carried
    over and covers the first graph_nodes vectors (open with hnsw_graph_nodes, then VectorSearcher.extend_hnsw)."""
    ops = (_lib.MergeOperandC * max(1, len(operands)))()
    keep = []
    for i, (d, alive) in enumerate(operands):
        ops[i].dir = d._h
        if alive is not None:
            bits = _bitset(np.asarray(alive, dtype=bool))
            keep.append(bits)
            ops[i].alive_bitset = bits.ctypes.data
    rec, vec, gn, hq = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_int32()
    _lib.check(_lib.lib().nidx_gpu_segment_dir_merge(path.encode(), dimension, ops, len(operands), C.byref(rec), C.byref(vec),
                                                   C.byref(gn), C.byref(hq)))
    return rec.value, vec.value, gn.value, bool(hq.value)


@dataclass
class _KeyPrefixSet:
    prefixes: List[str]


def segment_create(elems: Iterable[Elem], config: VectorConfig, tags: Optional[set] = None) -> VectorSegment:
    """segment::create (segment.rs:199-239): dimension check, optional normalisation is the
    indexer's job (indexer.rs:107-111).  The HNSW graph is built later on the device
    (VectorSearcher.build_hnsw) or supplied as an hnsw.graph image."""
    elems = list(elems)
    if config.vector_cardinality == VectorCardinality.Multi:
        # indexer.rs:116-125: the sentence's vector is the concatenation of the paragraph's vectors
        rows, pov = [], []
        for i, e in enumerate(elems):
            if len(e.vector) == 0 or len(e.vector) % config.dimension:
                raise NidxGpuError(_lib.NIDX_ERR_INCONSISTENT_DIMENSIONS,
                                   f"Inconsistent dimensions. Index={config.dimension} Vector={len(e.vector)}")
            m = np.asarray(e.vector, dtype=np.float32).reshape(-1, config.dimension)
            rows.append(m)
            pov += [i] * m.shape[0]
        vectors = np.vstack(rows) if rows else np.zeros((0, config.dimension), np.float32)
        return VectorSegment([e.key for e in elems], vectors, [list(e.labels) for e in elems], [e.metadata for e in elems], tags,
                             para_of_vec=np.asarray(pov, dtype=np.uint32))
    for e in elems:
        if len(e.vector) != config.dimension:
            raise NidxGpuError(_lib.NIDX_ERR_INCONSISTENT_DIMENSIONS,
                               f"Inconsistent dimensions. Index={config.dimension} Vector={len(e.vector)}")
    vectors = np.array([e.vector for e in elems], dtype=np.float32).reshape(len(elems), config.dimension)
    return VectorSegment([e.key for e in elems], vectors, [list(e.labels) for e in elems], [e.metadata for e in elems], tags)


def segment_merge(operants: Sequence[Tuple[VectorSegment, Optional[np.ndarray]]], config: VectorConfig) -> VectorSegment:
    """segment::merge (segment.rs:92-197) minus the files: operands sorted largest first, the alive
    paragraphs of each concatenated in that order; when the largest operand has no deletions its HNSW
    graph (and edge weights) is carried over for reuse, otherwise the merged segment has no graph.
    `operants` = [(segment, alive mask or None)]."""
    if not operants:
        raise NidxGpuError(_lib.NIDX_ERR_EMPTY_MERGE, "Can not merge zero segments")
    ops = sorted(operants, key=lambda t: -t[0].records)
    tags = ops[0][0].tags
    for seg, _ in ops:
        if seg.tags != tags:
            raise NidxGpuError(_lib.NIDX_ERR_INVALID_ARGUMENT, "Not all of the merged segments have the same tags")
    keys, labels, metadata, rows, povs = [], [], [], [], []
    multi = any(seg.para_of_vec is not None for seg, _ in ops)

    def vector_mask(seg, alive):
        # a paragraph's vectors live and die together (data_store/v2.rs:89-103 copies alive paragraphs with their vectors)
        if alive is None:
            return None
        alive = np.asarray(alive, dtype=bool)
        return alive if seg.para_of_vec is None else alive[seg.para_of_vec]

    for seg, alive in ops:
        idx = range(seg.records) if alive is None else np.nonzero(alive)[0].tolist()
        if multi:
            pov = seg.para_of_vec if seg.para_of_vec is not None else np.arange(seg.records, dtype=np.uint32)
            renumber = np.full(seg.records, -1, np.int64)
            renumber[list(idx)] = len(keys) + np.arange(len(idx))
            vm = vector_mask(seg, alive)
            povs.append(renumber[pov if vm is None else pov[vm]].astype(np.uint32))
        for i in idx:
            keys.append(seg.keys[i])
            labels.append(list(seg.labels[i]))
            metadata.append(seg.metadata[i])
        vm = vector_mask(seg, alive)
        rows.append(seg.vectors if vm is None else seg.vectors[vm])
    vectors = np.vstack(rows) if rows else np.zeros((0, config.dimension), np.float32)
    para_of_vec = np.concatenate(povs) if multi else None
    # quantized vectors are copied when every operand has them (data_store/v2.rs:104-113); otherwise the caller
    # re-encodes the merged segment with VectorSearcher.quantize()
    quantized = None
    if config.quantizable_vectors() and all(seg.quantized is not None for seg, _ in ops):
        quantized = np.vstack([seg.quantized if alive is None else seg.quantized[vector_mask(seg, alive)] for seg, alive in ops])
    first, first_alive = ops[0]
    reuse = first.graph is not None and not first.graph_nodes and (first_alive is None or bool(np.all(first_alive)))
    if reuse and first.records < len(keys):
        return VectorSegment(keys, vectors, labels, metadata, tags, graph=first.graph, graph_edges=first.graph_edges,
                             graph_nodes=first.vectors.shape[0], quantized=quantized, para_of_vec=para_of_vec)
    if reuse:
        return VectorSegment(keys, vectors, labels, metadata, tags, graph=first.graph, graph_edges=first.graph_edges,
                             quantized=quantized, para_of_vec=para_of_vec)
    return VectorSegment(keys, vectors, labels, metadata, tags, quantized=quantized, para_of_vec=para_of_vec)


def _segments_with_deletions(segments: Sequence[Tuple[VectorSegment, int]],
                             deletions: Sequence[Tuple[str, int]]) -> List[Tuple[VectorSegment, np.ndarray]]:
    """segment_deletions + apply_deletions (lib.rs:169-200, segment.rs:428-445): segments and deletions sorted by seq, walked
    newest -> oldest; a segment loses the paragraphs of every deletion with seq > its own seq.  Returns [(segment, alive
    mask)] newest first — the order open_segments / VectorIndexer::merge push them in."""
    segs = sorted(segments, key=lambda t: t[1])
    dels = sorted(deletions, key=lambda t: t[1])
    ordered: List[Tuple[VectorSegment, np.ndarray]] = []
    so_far: List[str] = []
    di = len(dels) - 1
    for seg, seq in reversed(segs):
        while di >= 0 and dels[di][1] > seq:
            so_far.append(dels[di][0])
            di -= 1
        alive = np.ones(seg.records, dtype=bool)
        for key in so_far:
            alive[seg.ids_for_deletion_key(key)] = False
        ordered.append((seg, alive))
    return ordered


# ---- the indexer side (nidx_vector::VectorIndexer, lib.rs:65-118; indexer.rs:28-145) ------------------------------------------
SEGMENT_TAGS = ("/q/h",)  # indexer.rs:26


@dataclass
class VectorSentence:
    """noderesources.VectorSentence: the vector (a Multi index holds the paragraph's vectors concatenated) + metadata bytes."""

    vector: Sequence[float]
    metadata: Optional[bytes] = None


@dataclass
class IndexParagraph:
    """noderesources.IndexParagraph, the fields the vector indexer reads."""

    start: int = 0
    end: int = 0
    sentences: dict = field(default_factory=dict)             # sentence key -> VectorSentence (the default vectorset)
    vectorsets_sentences: dict = field(default_factory=dict)  # vectorset -> {sentence key -> VectorSentence}
    labels: List[str] = field(default_factory=list)


@dataclass
class Resource:
    """noderesources.Resource, the fields the vector indexer reads."""

    uuid: str
    labels: List[str] = field(default_factory=list)
    paragraphs: dict = field(default_factory=dict)            # field id -> {paragraph key -> IndexParagraph}
    vector_prefixes_to_delete: dict = field(default_factory=dict)
    vectors_to_delete_in_all_vectorsets: List[str] = field(default_factory=list)


class VectorIndexer:
    """nidx_vector::VectorIndexer for paragraph indexes (lib.rs:65-118).  Segments are VectorSegment objects (save() writes
    the directory the reference's index_resource / merge leave in `output_dir`)."""

    def index_resource(self, config: VectorConfig, resource: Resource, index_name: str = "default",
                       use_default_vectorset: bool = True) -> Optional[VectorSegment]:
        """indexer.rs:94-145 over ResourceWrapper::fields (:59-86): one Elem per sentence of the vectorset (falling back to
        the default sentences when asked to), normalised when the index says so; None when the resource has no vectors."""
        elems = []
        for _field_id, paragraphs in resource.paragraphs.items():
            for paragraph in paragraphs.values():
                sentences = paragraph.vectorsets_sentences.get(index_name)
                if sentences is None:
                    if not use_default_vectorset:
                        continue
                    sentences = paragraph.sentences
                for key, sentence in sentences.items():
                    v = np.ascontiguousarray(sentence.vector, dtype=np.float32)
                    if config.normalize_vectors and v.size:
                        out = np.empty_like(v)
                        _lib.check(_lib.lib().nidx_gpu_normalize(v.ctypes.data, 1, v.size, out.ctypes.data))
                        v = out
                    elems.append(Elem(key, v, list(paragraph.labels), sentence.metadata or b""))
        if not elems:
            return None
        tags = {t for t in resource.labels if t in SEGMENT_TAGS}
        return segment_create(elems, config, tags)

    def deletions_for_resource(self, resource: Resource, index_name: str) -> List[str]:
        """lib.rs:90-96"""
        if index_name in resource.vector_prefixes_to_delete:
            return list(resource.vector_prefixes_to_delete[index_name])
        return list(resource.vectors_to_delete_in_all_vectorsets)

    def merge(self, config: VectorConfig, segments: Sequence[Tuple[VectorSegment, int]],
              deletions: Sequence[Tuple[str, int]] = ()) -> VectorSegment:
        """lib.rs:98-118: every segment opened with the deletions newer than itself applied, then segment::merge."""
        return segment_merge(_segments_with_deletions(segments, deletions), config)


def _bitset(mask: np.ndarray) -> np.ndarray:
    n = mask.shape[0]
    words = (n + 63) // 64
    padded = np.zeros(words * 64, dtype=np.uint8)
    padded[:n] = mask
    return np.packbits(padded.reshape(words, 64), axis=1, bitorder="little").view(np.uint64).reshape(words).copy()


def _segment_matches(expr, tags: set) -> bool:
    """searcher.rs:206-219 segment_matches."""
    if isinstance(expr, Literal):
        return expr.label in tags
    if isinstance(expr, Not):
        return not _segment_matches(expr.operand, tags)
    if isinstance(expr, And):
        return all(_segment_matches(o, tags) for o in expr.operands)
    if isinstance(expr, Or):
        return any(_segment_matches(o, tags) for o in expr.operands)
    raise TypeError(expr)


class VectorSearcher:
    """nidx_vector::VectorSearcher (lib.rs:120-148)."""

    def __init__(self):
        self._handle = C.c_void_p()
        self._segments: List[VectorSegment] = []
        self._keep = []  # buffers referenced by the C structs during open
        self.config: Optional[VectorConfig] = None
        self.last_methods: List[int] = []

    @classmethod
    def open(cls, config: VectorConfig, segments: Sequence[Tuple[VectorSegment, int]],
             deletions: Sequence[Tuple[str, int]] = (), quantize: bool = True) -> "VectorSearcher":
        """VectorSearcher::open(config, impl OpenIndexMetadata) (lib.rs:126-200): segments sorted by
        seq, walked newest -> oldest accumulating the deletions with seq > segment seq."""
        self = cls()
        self.config = config
        ordered = _segments_with_deletions(segments, deletions)
        # open_segments pushes in that (newest first) order and _search walks them in it
        c_segs = (_lib.VectorSegmentC * max(1, len(ordered)))()
        key_table: dict = {}  # Fssc equates hits by paragraph id string (searcher.rs:67-96): intern the keys
        for i, (seg, alive) in enumerate(ordered):
            if seg.vectors.shape[1] != config.dimension and seg.records:
                raise NidxGpuError(_lib.NIDX_ERR_INCONSISTENT_DIMENSIONS,
                                   f"Inconsistent dimensions. Index={config.dimension} Vector={seg.vectors.shape[1]}")
            bits = _bitset(alive)
            graph = np.frombuffer(seg.graph, dtype=np.uint8) if seg.graph else None
            key_ids = np.array([key_table.setdefault(k, len(key_table)) for k in seg.keys], dtype=np.uint64)
            self._keep += [bits, graph, seg.vectors, key_ids]
            c_segs[i].vectors = seg.vectors.ctypes.data
            c_segs[i].row_stride_bytes = config.dimension * 4
            c_segs[i].n_vectors = seg.vectors.shape[0]
            c_segs[i].paragraph_of_vector = None if seg.para_of_vec is None else seg.para_of_vec.ctypes.data
            c_segs[i].n_paragraphs = seg.records
            c_segs[i].hnsw_graph = graph.ctypes.data if graph is not None else None
            c_segs[i].hnsw_graph_len = len(seg.graph) if seg.graph else 0
            c_segs[i].hnsw_graph_nodes = seg.graph_nodes if seg.graph else 0
            has_edges = seg.graph is not None and seg.graph_edges is not None and len(seg.graph_edges)
            c_segs[i].hnsw_edges = seg.graph_edges.ctypes.data if has_edges else None
            c_segs[i].n_hnsw_edges = len(seg.graph_edges) if has_edges else 0
            c_segs[i].alive_bitset = bits.ctypes.data
            c_segs[i].paragraph_key_ids = key_ids.ctypes.data if seg.records else None
            if seg.quantized is not None and seg.records:
                c_segs[i].quantized = seg.quantized.ctypes.data
                c_segs[i].quantized_len = seg.quantized.size
            self._segments.append(seg)
        cfg = config.to_c()
        _lib.check(_lib.lib().nidx_gpu_vector_open(C.byref(cfg), c_segs, len(ordered), C.byref(self._handle)))
        # DataStoreV2::create writes vectors.quant for every quantizable index (data_store/v2.rs:57-76): a segment that
        # arrives without its codes gets them encoded on the device, so AUTO routes like OpenSegment::_search
        if quantize and config.quantizable_vectors():
            for i, (seg, _alive) in enumerate(ordered):
                if seg.quantized is None and seg.records:
                    self.quantize(i)
        for i, seg in enumerate(self._segments):
            if len(seg.list_keys):
                fi = _lib.FilterIndexC(len(seg.list_keys), seg.list_offsets.ctypes.data, seg.list_ids.ctypes.data if len(seg.list_ids) else None)
                _lib.check(_lib.lib().nidx_gpu_vector_set_filter_index(self._handle, i, C.byref(fi)))
                # the lists' keys as a sorted table in HBM: label-prefix and field-key lookups of a request run on the device
                enc = [k.encode("utf-8") for k in seg.list_keys]
                assert enc == sorted(enc)
                offs = np.zeros(len(enc) + 1, np.uint64)
                offs[1:] = np.cumsum([len(e) for e in enc])
                blob = np.frombuffer(b"".join(enc) + b"\0", np.uint8)
                _lib.check(_lib.lib().nidx_gpu_vector_set_filter_keys(self._handle, i, blob.ctypes.data, offs.ctypes.data, len(enc)))
        self._keep = []  # everything was copied to HBM / host vectors by open
        return self

    def close(self):
        if self._handle:
            _lib.lib().nidx_gpu_vector_close(self._handle)
            self._handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _lookup(self, segment: int, queries) -> List[Tuple[int, int]]:
        """label.fst / field.fst lookups of one segment, batched on the device: [(first list, last list)] per query."""
        if not queries:
            return []
        blob = np.frombuffer(b"".join(q for q, _ in queries) + b"\0", np.uint8)
        offs = np.zeros(len(queries) + 1, np.uint64)
        offs[1:] = np.cumsum([len(q) for q, _ in queries])
        flags = np.array([p for _, p in queries], dtype=np.uint8)
        first, last = np.zeros(len(queries), np.uint32), np.zeros(len(queries), np.uint32)
        _lib.check(_lib.lib().nidx_gpu_vector_lookup_filter_keys(self._handle, segment, blob.ctypes.data, offs.ctypes.data, flags.ctypes.data,
                                                                 len(queries), first.ctypes.data, last.ctypes.data))
        return list(zip(first.tolist(), last.tolist()))

    def space_usage(self) -> int:
        out = C.c_uint64(0)
        _lib.check(_lib.lib().nidx_gpu_vector_space_usage(self._handle, C.byref(out)))
        return out.value

    def spill_queries(self) -> int:
        """Queries re-run by the exact HBM-resident closest_up_nodes fallback since open."""
        out = C.c_uint64(0)
        _lib.check(_lib.lib().nidx_gpu_vector_spill_stats(self._handle, C.byref(out)))
        return out.value

    def build_hnsw(self, segment: int = 0, level_seed: int = 2):
        """HnswBuilder on the device (hnsw/build.rs); the reference seeds the level RNG with 2."""
        _lib.check(_lib.lib().nidx_gpu_vector_build_hnsw(self._handle, segment, level_seed))

    def extend_hnsw(self, segment: int = 0, level_seed: int = 2):
        """The graph-reuse half of segment::merge (segment.rs:137-167): insert the vectors after the reused graph."""
        _lib.check(_lib.lib().nidx_gpu_vector_extend_hnsw(self._handle, segment, level_seed))

    def quantize(self, segment: int = 0):
        """DataStoreV2::create's quantized writer (data_store/v2.rs:57-76), on the device."""
        _lib.check(_lib.lib().nidx_gpu_vector_quantize(self._handle, segment))

    def serialize_quantized(self, segment: int = 0) -> np.ndarray:
        """The records of vectors.quant, [records][dimension/8 + 8] u8."""
        n = C.c_uint64()
        _lib.check(_lib.lib().nidx_gpu_vector_serialize_quantized(self._handle, segment, None, 0, C.byref(n)))
        out = np.zeros(n.value, np.uint8)
        _lib.check(_lib.lib().nidx_gpu_vector_serialize_quantized(self._handle, segment, out.ctypes.data, out.size, C.byref(n)))
        return out.reshape(-1, self.config.dimension // 8 + 8)

    def serialize_hnsw(self, segment: int = 0) -> Tuple[bytes, np.ndarray]:
        glen, nedges = C.c_uint64(0), C.c_uint64(0)
        L = _lib.lib()
        _lib.check(L.nidx_gpu_vector_serialize_hnsw(self._handle, segment, None, 0, C.byref(glen), None, 0, C.byref(nedges)))
        graph = np.zeros(max(1, glen.value), dtype=np.uint8)
        edges = np.zeros(max(1, nedges.value), dtype=np.float32)
        _lib.check(L.nidx_gpu_vector_serialize_hnsw(self._handle, segment, graph.ctypes.data, glen.value, C.byref(glen),
                                                    edges.ctypes.data, nedges.value, C.byref(nedges)))
        return graph[: glen.value].tobytes(), edges[: nedges.value]

    # -- Searcher::search (searcher.rs:292-343) ---------------------------------------------------------
    def _formula(self, request: VectorSearchRequest, prefilter: PrefilterResult):
        clauses = []
        if prefilter.kind == "some":
            prefixes = []
            for f in prefilter.fields:
                prefixes.append(f.resource_id.hex + (f.field_id or ""))
            clauses.append(_KeyPrefixSet(prefixes))
        if request.filtering_formula is not None:
            clauses.append(request.filtering_formula)
        if not clauses:
            return None
        return Or(clauses) if request.filter_operator == FilterOperator.Or else And(clauses)

    def _search_multi_vector(self, request: VectorSearchRequest, prefilter: PrefilterResult, method: int) -> "VectorSearchResponse":
        """Searcher::search_multi_vector (searcher.rs:345-394): the query is the concatenation of its vectors; every one is
        searched on its own, the paragraphs found are re-scored with maxsim_similarity and cut at min_score / top k."""
        d = self.config.dimension
        flat = np.ascontiguousarray(request.vector, dtype=np.float32).reshape(-1)
        if flat.size == 0 or flat.size % d:
            raise NidxGpuError(_lib.NIDX_ERR_INCONSISTENT_DIMENSIONS, f"Inconsistent dimensions. Index={d} Vector={flat.size}")
        k = max(0, int(request.result_per_page))
        S = len(self._segments)
        formula = self._formula(request, prefilter)
        filt_arrays, filt_ptrs = [], (C.c_void_p * max(1, S))()
        for s_, seg in enumerate(self._segments):
            skip = request.segment_filtering_formula is not None and not _segment_matches(request.segment_filtering_formula, seg.tags)
            bits = _bitset(np.zeros(seg.records, dtype=bool)) if skip else (_bitset(seg._eval(formula)) if formula is not None else None)
            filt_arrays.append(bits)
            filt_ptrs[s_] = bits.ctypes.data if bits is not None else None
        kk = max(1, k)
        out_seg, out_par = np.zeros((1, kk), np.uint32), np.zeros((1, kk), np.uint32)
        out_score, out_count = np.zeros((1, kk), np.float32), np.zeros(1, np.uint32)
        qoff = np.array([0, flat.size // d], dtype=np.uint64)
        params = _lib.VectorSearchParamsC(k, float(request.min_score), 1, method)
        _lib.check(_lib.lib().nidx_gpu_vector_search_maxsim(
            self._handle, flat.ctypes.data, qoff.ctypes.data, 1, C.byref(params),
            filt_ptrs if any(b is not None for b in filt_arrays) else None, out_seg.ctypes.data, out_par.ctypes.data,
            out_score.ctypes.data, out_count.ctypes.data))
        docs = []
        for i in range(int(out_count[0])):
            sg = self._segments[int(out_seg[0, i])]
            p = int(out_par[0, i])
            md = sg.metadata[p]
            docs.append(DocumentScored(sg.keys[p], float(out_score[0, i]), md if md else None, list(sg.labels[p])))
        return VectorSearchResponse(docs)

    def search_batch(self, request: VectorSearchRequest, queries: np.ndarray, prefilter: PrefilterResult = None,
                     method: int = _lib.METHOD_AUTO, device_filter: bool = True):
        """Batched form of search(): `queries` [B][D].  Returns (segment, paragraph, vector, score, count).
        device_filter=True sends the formula as a program evaluated on the GPU (filter.hip); False builds the
        bitsets with numpy and uploads them (kept for cross-checking the two routes)."""
        prefilter = prefilter or PrefilterResult.all()
        queries = np.ascontiguousarray(queries, dtype=np.float32)
        if queries.ndim != 2:
            raise ValueError("queries must be [B][D]")
        B, D = queries.shape
        k = max(0, int(request.result_per_page))
        S = len(self._segments)
        formula = self._formula(request, prefilter)
        kk = max(1, k)
        out_seg = np.zeros((B, kk), dtype=np.uint32)
        out_par = np.zeros((B, kk), dtype=np.uint32)
        out_vec = np.zeros((B, kk), dtype=np.uint32)
        out_score = np.zeros((B, kk), dtype=np.float32)
        out_count = np.zeros(B, dtype=np.uint32)
        out_method = np.zeros(max(1, S), dtype=np.int32)
        out_matching = np.zeros(max(1, S), dtype=np.uint64)
        params = _lib.VectorSearchParamsC(k, float(request.min_score), int(request.with_duplicates), method)
        skips = [request.segment_filtering_formula is not None and not _segment_matches(request.segment_filtering_formula, seg.tags)
                 for seg in self._segments]  # segment tag filter (searcher.rs:272-277): a non-matching segment is skipped
        if device_filter:
            progs = (_lib.FilterProgramC * max(1, S))()
            keep = []
            for s, seg in enumerate(self._segments):
                if skips[s]:
                    ops, lists = [(_lib.FILTER_PUSH_NONE, 0, 0)], []
                elif formula is not None:
                    ops, lists = seg.compile(formula, lookup=lambda queries, s=s: self._lookup(s, queries))
                else:
                    continue
                c_ops = (_lib.FilterOpC * len(ops))(*[_lib.FilterOpC(*o) for o in ops])
                c_lists = np.array(lists, dtype=np.uint32)
                keep += [c_ops, c_lists]
                progs[s] = _lib.FilterProgramC(C.addressof(c_ops), len(ops), c_lists.ctypes.data if len(lists) else None, len(lists))
            rc = _lib.lib().nidx_gpu_vector_search_filtered(
                self._handle, queries.ctypes.data, B, D, C.byref(params), progs if keep else None,
                out_seg.ctypes.data, out_par.ctypes.data, out_vec.ctypes.data, out_score.ctypes.data,
                out_count.ctypes.data, out_method.ctypes.data, out_matching.ctypes.data)
        else:
            filt_arrays, filt_ptrs = [], (C.c_void_p * max(1, S))()
            for s, seg in enumerate(self._segments):
                if skips[s]:
                    bits = _bitset(np.zeros(seg.records, dtype=bool))
                elif formula is not None:
                    bits = _bitset(seg._eval(formula))
                else:
                    bits = None
                filt_arrays.append(bits)
                filt_ptrs[s] = bits.ctypes.data if bits is not None else None
            any_filter = any(b is not None for b in filt_arrays)
            rc = _lib.lib().nidx_gpu_vector_search_dim(
                self._handle, queries.ctypes.data, B, D, C.byref(params), filt_ptrs if any_filter else None,
                out_seg.ctypes.data, out_par.ctypes.data, out_vec.ctypes.data, out_score.ctypes.data,
                out_count.ctypes.data, out_method.ctypes.data)
        _lib.check(rc)
        self.last_methods = out_method[:S].tolist()
        self.last_matching = out_matching[:S].tolist()
        return out_seg, out_par, out_vec, out_score, out_count

    def search(self, request: VectorSearchRequest, prefilter: PrefilterResult = None,
               method: int = _lib.METHOD_AUTO, device_filter: bool = True) -> VectorSearchResponse:
        prefilter = prefilter or PrefilterResult.all()
        if self.config.vector_cardinality == VectorCardinality.Multi:
            return self._search_multi_vector(request, prefilter, method)
        q = np.asarray(request.vector, dtype=np.float32).reshape(1, -1)
        seg, par, _vec, score, count = self.search_batch(request, q, prefilter, method, device_filter)
        docs = []
        for i in range(int(count[0])):
            s, p = int(seg[0, i]), int(par[0, i])
            sg = self._segments[s]
            md = sg.metadata[p]
            docs.append(DocumentScored(sg.keys[p], float(score[0, i]), md if md else None, list(sg.labels[p])))
        return VectorSearchResponse(docs)


def fst_map_build(entries: Sequence[Tuple[bytes, int]]) -> bytes:
    """fst::MapBuilder over strictly ascending (key, value) pairs (nidx_gpu_fst_map_build) -> the image."""
    keys = b"".join(k for k, _ in entries)
    offs = np.zeros(len(entries) + 1, np.uint64)
    offs[1:] = np.cumsum([len(k) for k, _ in entries], dtype=np.uint64)
    vals = np.array([v for _, v in entries], np.uint64)
    kb = np.frombuffer(keys, np.uint8) if keys else np.zeros(1, np.uint8)
    n = C.c_uint64(0)
    _lib.check(_lib.lib().nidx_gpu_fst_map_build(kb.ctypes.data, offs.ctypes.data, vals.ctypes.data, len(entries), None, 0, C.byref(n)))
    out = np.zeros(n.value, np.uint8)
    _lib.check(_lib.lib().nidx_gpu_fst_map_build(kb.ctypes.data, offs.ctypes.data, vals.ctypes.data, len(entries), out.ctypes.data, out.size, C.byref(n)))
    return out.tobytes()


def fst_map_get(image: bytes, key: bytes) -> Optional[int]:
    buf = np.frombuffer(image, np.uint8)
    v, found = C.c_uint64(0), C.c_int32(0)
    _lib.check(_lib.lib().nidx_gpu_fst_map_get(buf.ctypes.data, buf.size, key, len(key), C.byref(v), C.byref(found)))
    return v.value if found.value else None


def fst_map_entries(image: bytes) -> List[Tuple[bytes, int]]:
    """Every (key, value) of an fst::Map image in key order (nidx_gpu_fst_map_entries); raises on a malformed image."""
    buf = np.frombuffer(image, np.uint8)
    n, klen = C.c_uint32(0), C.c_uint64(0)
    _lib.check(_lib.lib().nidx_gpu_fst_map_entries(buf.ctypes.data, buf.size, None, 0, None, None, 0, C.byref(n), C.byref(klen)))
    keys = np.zeros(max(klen.value, 1), np.uint8)
    offs = np.zeros(n.value + 1, np.uint64)
    vals = np.zeros(max(n.value, 1), np.uint64)
    _lib.check(_lib.lib().nidx_gpu_fst_map_entries(buf.ctypes.data, buf.size, keys.ctypes.data, keys.size, offs.ctypes.data, vals.ctypes.data,
                                                  n.value, C.byref(n), C.byref(klen)))
    kb = keys.tobytes()
    return [(kb[int(offs[i]):int(offs[i + 1])], int(vals[i])) for i in range(n.value)]


def index_map_read(data: bytes, pos: int) -> np.ndarray:
    """InvertedMapReader::get (inverted_index/map.rs:63-70) through nidx_gpu_index_map_read."""
    buf = np.frombuffer(data, np.uint8)
    n = C.c_uint32(0)
    _lib.check(_lib.lib().nidx_gpu_index_map_read(buf.ctypes.data, buf.size, pos, None, 0, C.byref(n)))
    out = np.zeros(max(n.value, 1), np.uint32)
    _lib.check(_lib.lib().nidx_gpu_index_map_read(buf.ctypes.data, buf.size, pos, out.ctypes.data, out.size, C.byref(n)))
    return out[: n.value]
