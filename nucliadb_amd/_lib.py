"""ctypes binding of libnidx_gpu.so (include/nidx_gpu.h).

The library is the product: there is no Python / CPU fallback.  Importing this module on a machine
where the shared object is missing raises ImportError, and every compute entry point raises
NidxGpuError(NIDX_ERR_DEVICE) when no gfx950 device is present.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# NIDX_GPU_LIB: another build of the library (kernel A/B experiments on the GPU box); the product is the in-tree file
LIB_PATH = os.environ.get("NIDX_GPU_LIB") or os.path.join(_HERE, "libnidx_gpu.so")

NIDX_OK = 0
NIDX_ERR_IO = -1
NIDX_ERR_INCONSISTENT_DIMENSIONS = -2
NIDX_ERR_INVALID_CONFIGURATION = -3
NIDX_ERR_EMPTY_MERGE = -4
NIDX_ERR_INVALID_ARGUMENT = -5
NIDX_ERR_UNSUPPORTED = -6
NIDX_ERR_DEVICE = -7
NIDX_ERR_INVALID_GRAPH = -8
NIDX_ERR_INEXACT = -9
NIDX_ERR_OUT_OF_MEMORY = -10
NIDX_ERR_INTERNAL = -11
NIDX_ERR_BUSY = -12

SIMILARITY_DOT, SIMILARITY_COSINE = 0, 1
METHOD_AUTO, METHOD_HNSW, METHOD_BRUTE_FORCE, METHOD_BRUTE_FORCE_MFMA, METHOD_BRUTE_FORCE_BF16 = 0, 1, 2, 3, 4
METHOD_RABITQ_HNSW, METHOD_RABITQ_BRUTE_FORCE = 5, 6
CONFIG_DISABLE_RABITQ_SEARCH = 1
ORDER_WAVE64, ORDER_SERIAL_FMA = 3, 1
OCCUR_SHOULD, OCCUR_MUST, OCCUR_MUST_NOT, OCCUR_SHOULD_GROUP = 0, 1, 2, 3
TF_FREQ, TF_BASIC, CONST_SCORE = 0, 1, 2


class NidxGpuError(RuntimeError):
    """An error code returned across the C ABI (the analogue of nidx_vector::VectorErr)."""

    def __init__(self, code: int, message: str):
        super().__init__(f"[{code}] {message}")
        self.code = code
        self.message = message


class VectorConfigC(C.Structure):
    _fields_ = [
        ("dimension", C.c_uint32),
        ("similarity", C.c_int32),
        ("normalize_vectors", C.c_int32),
        ("vector_cardinality", C.c_int32),
        ("flags", C.c_uint32),
    ]


class VectorSegmentC(C.Structure):
    _fields_ = [
        ("vectors", C.c_void_p),
        ("row_stride_bytes", C.c_uint64),
        ("n_vectors", C.c_uint32),
        ("paragraph_of_vector", C.c_void_p),
        ("n_paragraphs", C.c_uint32),
        ("hnsw_graph", C.c_void_p),
        ("hnsw_graph_len", C.c_uint64),
        ("hnsw_graph_nodes", C.c_uint32),
        ("hnsw_edges", C.c_void_p),
        ("n_hnsw_edges", C.c_uint64),
        ("alive_bitset", C.c_void_p),
        ("paragraph_key_ids", C.c_void_p),
        ("quantized", C.c_void_p),
        ("quantized_len", C.c_uint64),
    ]


class Bm25SearchOptionsC(C.Structure):
    _fields_ = [
        ("k", C.c_uint32),
        ("after", C.c_void_p),
        ("term_set_terms", C.c_void_p),
        ("term_set_offsets", C.c_void_p),
        ("n_term_sets", C.c_uint32),
        ("term_set_complement", C.c_void_p),
        ("phrase_terms", C.c_void_p),
        ("phrase_offsets", C.c_void_p),
        ("n_phrases", C.c_uint32),
        ("order_field", C.c_int32),
        ("order_desc", C.c_int32),
        ("facet_terms", C.c_void_p),
        ("facet_offsets", C.c_void_p),
        ("out_facet_counts", C.c_void_p),
        ("out_order_value", C.c_void_p),
        ("subquery_clauses", C.c_void_p),
        ("subquery_offsets", C.c_void_p),
        ("n_subqueries", C.c_uint32),
        ("phrase_slops", C.c_void_p),
    ]


BM25_TERM_SET = 0x80000000
BM25_PHRASE = 0x40000000
BM25_SUBQUERY = 0x20000000


class FilterIndexC(C.Structure):
    _fields_ = [("n_lists", C.c_uint32), ("list_offsets", C.c_void_p), ("paragraph_ids", C.c_void_p)]


class ParagraphC(C.Structure):
    _fields_ = [("key", C.c_void_p), ("key_len", C.c_uint32), ("metadata", C.c_void_p), ("metadata_len", C.c_uint32),
                ("n_labels", C.c_uint32), ("first_vector", C.c_uint32), ("num_vectors", C.c_uint32)]


class MergeOperandC(C.Structure):
    _fields_ = [("dir", C.c_void_p), ("alive_bitset", C.c_void_p)]


class SegmentDirContentsC(C.Structure):
    _fields_ = [("dimension", C.c_uint32), ("n_vectors", C.c_uint32), ("n_paragraphs", C.c_uint32), ("vectors", C.c_void_p),
                ("paragraph_of_vector", C.c_void_p), ("keys", C.c_void_p), ("key_offsets", C.c_void_p), ("labels", C.c_void_p),
                ("label_offsets", C.c_void_p), ("paragraph_label_offsets", C.c_void_p), ("metadata", C.c_void_p),
                ("metadata_offsets", C.c_void_p), ("hnsw_graph", C.c_void_p), ("hnsw_graph_len", C.c_uint64),
                ("hnsw_edges", C.c_void_p), ("n_hnsw_edges", C.c_uint64), ("quantized", C.c_void_p), ("quantized_len", C.c_uint64)]


LIST_LABEL, LIST_FIELD = 0, 1


class RankedListC(C.Structure):
    _fields_ = [("ids", C.c_void_p), ("scores", C.c_void_p), ("counts", C.c_void_p), ("stride", C.c_uint32), ("weight", C.c_double)]


class FacetCountC(C.Structure):
    _fields_ = [("group", C.c_void_p), ("group_len", C.c_uint32), ("tag", C.c_void_p), ("tag_len", C.c_uint32), ("total", C.c_int32)]


MERGE_ORDER_SCORE, MERGE_ORDER_VALUE_DESC, MERGE_ORDER_VALUE_ASC = 0, 1, 2
SHARD_COMM_ID_BYTES = 128


class FilterOpC(C.Structure):
    _fields_ = [("op", C.c_int32), ("a", C.c_uint32), ("b", C.c_uint32)]


class FilterProgramC(C.Structure):
    _fields_ = [("ops", C.c_void_p), ("n_ops", C.c_uint32), ("lists", C.c_void_p), ("n_lists", C.c_uint32)]


FILTER_PUSH_LISTS, FILTER_AND, FILTER_OR, FILTER_NOT, FILTER_PUSH_ALL, FILTER_PUSH_NONE = 0, 1, 2, 3, 4, 5
FILTER_PUSH_RANGE, FILTER_PUSH_PHRASE = 6, 7  # text-index prefilter only


class Bm25DateRangeC(C.Structure):
    _fields_ = [("field", C.c_uint32), ("has_since", C.c_int32), ("has_until", C.c_int32), ("reserved", C.c_int32),
                ("since", C.c_int64), ("until", C.c_int64)]


class Bm25PrefilterC(C.Structure):
    _fields_ = [("program", FilterProgramC), ("ranges", C.c_void_p), ("n_ranges", C.c_uint32), ("n_phrases", C.c_uint32),
                ("phrase_terms", C.c_void_p), ("phrase_offsets", C.c_void_p)]


class VectorSearchParamsC(C.Structure):
    _fields_ = [
        ("k", C.c_uint32),
        ("min_score", C.c_float),
        ("with_duplicates", C.c_int32),
        ("method", C.c_int32),
    ]


class Bm25SegmentC(C.Structure):
    _fields_ = [
        ("n_docs", C.c_uint32),
        ("total_num_tokens", C.c_uint64),
        ("n_terms", C.c_uint32),
        ("term_offsets", C.c_void_p),
        ("doc_ids", C.c_void_p),
        ("tfs", C.c_void_p),
        ("fieldnorm_ids", C.c_void_p),
        ("alive_bitset", C.c_void_p),
        ("pos_offsets", C.c_void_p),
        ("positions", C.c_void_p),
    ]


class Bm25ClauseC(C.Structure):
    _fields_ = [("term", C.c_uint32), ("occur", C.c_int32), ("mode", C.c_int32), ("boost", C.c_float)]


class Bm25SearchAfterC(C.Structure):
    _fields_ = [("has_after", C.c_int32), ("score", C.c_float), ("tie_break", C.c_int32), ("docaddr", C.c_uint64)]


ABI_VERSION = 6   # include/nidx_gpu.h: NIDX_GPU_ABI_VERSION
FEATURE_RABITQ_EXPERIMENTS = 1   # nidx_gpu_build_features()

# name -> (restype, argtypes); the list every `-m "not gpu"` export test walks.
SIGNATURES = {
    "nidx_gpu_last_error": (C.c_int32, [C.c_char_p, C.c_size_t]),
    "nidx_gpu_abi_version": (C.c_int32, []),
    "nidx_gpu_build_features": (C.c_int32, []),
    "nidx_gpu_device_count": (C.c_int32, [C.POINTER(C.c_int32)]),
    "nidx_gpu_set_device": (C.c_int32, [C.c_int32]),
    "nidx_gpu_vector_open": (C.c_int32, [C.POINTER(VectorConfigC), C.POINTER(VectorSegmentC), C.c_uint32, C.POINTER(C.c_void_p)]),
    "nidx_gpu_vector_close": (None, [C.c_void_p]),
    "nidx_gpu_vector_set_tunable": (C.c_int32, [C.c_void_p, C.c_char_p, C.c_int32]),
    "nidx_gpu_vector_space_usage": (C.c_int32, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "nidx_gpu_vector_num_segments": (C.c_int32, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "nidx_gpu_vector_segment_records": (C.c_int32, [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]),
    "nidx_gpu_vector_search": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(VectorSearchParamsC), C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nidx_gpu_vector_search_dim": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(VectorSearchParamsC),
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nidx_gpu_vector_set_filter_index": (C.c_int32, [C.c_void_p, C.c_uint32, C.POINTER(FilterIndexC)]),
    "nidx_gpu_vector_search_filtered": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(VectorSearchParamsC),
                                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                    C.c_void_p]),
    "nidx_gpu_vector_segment_search_device": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                                                          C.POINTER(VectorSearchParamsC), C.c_void_p, C.c_void_p, C.c_void_p,
                                                          C.c_void_p, C.c_void_p, C.c_void_p]),
    "nidx_gpu_bm25_apply_deletions": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint64)]),
    "nidx_gpu_vector_set_filter_keys": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32]),
    "nidx_gpu_vector_lookup_filter_keys": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]),
    "nidx_gpu_diag_single_query_latency": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(VectorSearchParamsC), C.c_uint32,
                                                       C.c_uint32, C.c_void_p, C.POINTER(C.c_double)]),
    "nidx_gpu_diag_gather": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int32, C.c_uint32,
                                         C.POINTER(C.c_float)]),
    "nidx_gpu_vector_device_flags": (C.c_int32, [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]),
    "nidx_gpu_vector_segment_search_device_exact": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                                                                C.POINTER(VectorSearchParamsC), C.c_void_p, C.c_void_p, C.c_void_p,
                                                                C.c_void_p, C.POINTER(C.c_uint32)]),
    "nidx_gpu_vector_search_one": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(VectorSearchParamsC), C.c_void_p,
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]),
    "nidx_gpu_vector_search_submit": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(VectorSearchParamsC), C.c_void_p,
                                                  C.POINTER(C.c_uint64)]),
    "nidx_gpu_vector_search_wait": (C.c_int32, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.POINTER(C.c_uint32)]),
    "nidx_gpu_vector_spill_stats": (C.c_int32, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "nidx_gpu_vector_coalescer_stats": (C.c_int32, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "nidx_gpu_use_hnsw": (C.c_int32, [C.c_uint64, C.c_uint64, C.c_uint64, C.c_int32]),
    "nidx_gpu_similarity": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, C.c_void_p]),
    "nidx_gpu_normalize": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]),
    "nidx_gpu_vector_build_hnsw": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_uint64]),
    "nidx_gpu_vector_build_stats": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "nidx_gpu_vector_extend_hnsw": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_uint64]),
    "nidx_gpu_vector_search_maxsim": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(VectorSearchParamsC), C.c_void_p,
                                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nidx_gpu_vector_quantize": (C.c_int32, [C.c_void_p, C.c_uint32]),
    "nidx_gpu_vector_serialize_quantized": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]),
    "nidx_gpu_vector_serialize_hnsw": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64),
                                                   C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]),
    "nidx_gpu_bm25_open": (C.c_int32, [C.POINTER(Bm25SegmentC), C.c_uint32, C.POINTER(C.c_void_p)]),
    "nidx_gpu_bm25_close": (None, [C.c_void_p]),
    "nidx_gpu_bm25_space_usage": (C.c_int32, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "nidx_gpu_bm25_search": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nidx_gpu_bm25_search_ex": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(Bm25SearchOptionsC), C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nidx_gpu_bm25_search_submit": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(Bm25SearchOptionsC), C.POINTER(C.c_uint64)]),
    "nidx_gpu_bm25_search_wait": (C.c_int32, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nidx_gpu_bm25_set_fast_field": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]),
    "nidx_gpu_bm25_set_dictionary": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "nidx_gpu_segment_dir_open": (C.c_int32, [C.c_char_p, C.c_uint32, C.POINTER(C.c_void_p)]),
    "nidx_gpu_segment_dir_close": (None, [C.c_void_p]),
    "nidx_gpu_segment_dir_index_source": (C.c_int32, [C.c_void_p]),
    "nidx_gpu_fst_map_build": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]),
    "nidx_gpu_fst_map_get": (C.c_int32, [C.c_void_p, C.c_uint64, C.c_char_p, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_int32)]),
    "nidx_gpu_fst_map_entries": (C.c_int32, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32,
                                             C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]),
    "nidx_gpu_index_map_read": (C.c_int32, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]),
    "nidx_gpu_segment_dir_segment": (C.c_int32, [C.c_void_p, C.POINTER(VectorSegmentC)]),
    "nidx_gpu_segment_dir_filter_index": (C.c_int32, [C.c_void_p, C.POINTER(FilterIndexC)]),
    "nidx_gpu_segment_dir_lists": (C.c_int32, [C.c_void_p, C.c_int32, C.c_char_p, C.c_uint32, C.c_int32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "nidx_gpu_segment_dir_paragraph": (C.c_int32, [C.c_void_p, C.c_uint32, C.POINTER(ParagraphC)]),
    "nidx_gpu_segment_dir_paragraph_label": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_uint32)]),
    "nidx_gpu_segment_dir_write": (C.c_int32, [C.c_char_p, C.POINTER(SegmentDirContentsC)]),
    "nidx_gpu_hnsw_graph_check": (C.c_int32, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint32, C.POINTER(C.c_uint32),
                                              C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "nidx_gpu_segment_dir_apply_deletions": (C.c_int32, [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_uint32), C.c_uint32, C.c_void_p,
                                                         C.POINTER(C.c_uint32)]),
    "nidx_gpu_segment_dir_merge": (C.c_int32, [C.c_char_p, C.c_uint32, C.POINTER(MergeOperandC), C.c_uint32, C.POINTER(C.c_uint32),
                                               C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_int32)]),
    "nidx_gpu_bm25_prefilter": (C.c_int32, [C.c_void_p, C.POINTER(Bm25PrefilterC), C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "nidx_gpu_bm25_fuzzy_terms": (C.c_int32, [C.c_void_p, C.c_char_p, C.c_uint32, C.c_int32, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]),
    "nidx_gpu_bm25_last_kernel_ms": (C.c_int32, [C.c_void_p, C.POINTER(C.c_float)]),
    "nidx_gpu_bm25_idf": (C.c_float, [C.c_uint64, C.c_uint64]),
    "nidx_gpu_fieldnorm_from_id": (C.c_uint32, [C.c_uint8]),
    "nidx_gpu_fieldnorm_to_id": (C.c_uint8, [C.c_uint32]),
    "nidx_gpu_merge_vector": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.POINTER(C.c_uint32)]),
    "nidx_gpu_merge_vector_device": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nidx_gpu_merge_bm25_device": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                               C.c_uint32, C.c_uint32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_void_p]),
    "nidx_gpu_merge_vector_batch": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                                C.c_void_p, C.c_void_p, C.c_void_p]),
    "nidx_gpu_merge_bm25_batch": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                              C.c_uint32, C.c_uint32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nidx_gpu_merge_facets": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]),
    "nidx_gpu_shard_comm_unique_id": (C.c_int32, [C.c_void_p]),
    "nidx_gpu_shard_comm_unique_id_shm": (C.c_int32, [C.c_void_p]),
    "nidx_gpu_shard_comm_init": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_char_p, C.c_uint32, C.POINTER(C.c_void_p)]),
    "nidx_gpu_shard_comm_destroy": (None, [C.c_void_p]),
    "nidx_gpu_shard_exchange_merge_vector": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nidx_gpu_shard_exchange_merge_bm25": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                                       C.c_uint32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                       C.c_void_p]),
    "nidx_gpu_rank_fusion_rrf": (C.c_int32, [C.POINTER(RankedListC), C.c_uint32, C.c_uint32, C.c_double, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nidx_gpu_rank_fusion_wcombsum": (C.c_int32, [C.POINTER(RankedListC), C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nidx_gpu_merge_bm25": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]),
}

_lib = None


def _share_torch_hip_runtime() -> None:
    """One HIP/HSA runtime per process.  PyTorch-ROCm ships its own libamdhip64 / libhsa-runtime64 and asks for them by
    file name, so when libnidx_gpu.so (linked against /opt/rocm's libamdhip64.so.7) is loaded first, a later `import torch`
    maps a second runtime whose hsa_init finds no device ("No HIP GPUs are available").  The other order shares one
    runtime, because torch's copy carries the soname libnidx_gpu.so asks for.  So: when torch is installed (not necessarily
    imported), map its runtime first.  Processes without PyTorch (the Rust host) are not affected."""
    import importlib.util

    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    path = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(path):
        C.CDLL(path, mode=C.RTLD_GLOBAL)


def lib() -> C.CDLL:
    """Loads libnidx_gpu.so (once).  Raises ImportError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback."
            )
        _share_torch_hip_runtime()
        handle = C.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = restype
            fn.argtypes = argtypes
        got = handle.nidx_gpu_abi_version()
        if got != ABI_VERSION:   # a stale .so against newer struct layouts would read past the end of the caller's structs
            raise ImportError(f"{LIB_PATH} speaks ABI version {got}, these bindings were written against {ABI_VERSION} "
                              "(include/nidx_gpu.h: NIDX_GPU_ABI_VERSION): rebuild it with __graft_entry__.build()")
        _lib = handle
    return _lib


def last_error() -> str:
    buf = C.create_string_buffer(1024)
    lib().nidx_gpu_last_error(buf, len(buf))
    return buf.value.decode("utf-8", "replace")


def check(rc: int) -> None:
    if rc != NIDX_OK:
        raise NidxGpuError(rc, last_error())


def device_count() -> int:
    n = C.c_int32(0)
    rc = lib().nidx_gpu_device_count(C.byref(n))
    return n.value if rc == NIDX_OK else 0
