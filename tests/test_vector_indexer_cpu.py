"""VectorIndexer (nidx_vector/src/lib.rs:65-118, indexer.rs:28-145): index_resource, deletions_for_resource and merge with
the per-seq deletions of OpenIndexMetadata.  Host logic only; the search over a merged segment is in
tests/test_vector_reference_gpu.py::test_paragraph_merge_with_deletions."""
import numpy as np

from nucliadb_amd.vector import (IndexParagraph, Resource, Similarity, VectorCardinality, VectorConfig, VectorIndexer,
                                 VectorSentence)

DIMENSION = 4
UUID1 = "00112233445566778899aabbccddeeff"
UUID2 = "ffeeddccbbaa99887766554433221100"


def make_vector(index):
    v = [0.0] * DIMENSION
    v[index] = 1.0
    return v


def make_field(uuid, field_key, vector):
    para_key = f"{uuid}/{field_key}/0-10"
    return f"{uuid}/{field_key}", {para_key: IndexParagraph(0, 10, {para_key: VectorSentence(vector)})}


def build_resource(uuid, vector0, vector1):
    """tests/test_paragraph_merge.rs:35-66: two fields (a/title, t/body), one paragraph sentence each."""
    return Resource(uuid, paragraphs=dict([make_field(uuid, "a/title", vector0), make_field(uuid, "t/body", vector1)]))


def two_segments():
    config = VectorConfig.for_paragraphs(DIMENSION)
    s1 = VectorIndexer().index_resource(config, build_resource(UUID1, make_vector(0), make_vector(1)), "default", True)
    s2 = VectorIndexer().index_resource(config, build_resource(UUID2, make_vector(2), make_vector(3)), "default", True)
    assert s1.records == 2 and s2.records == 2
    return config, s1, s2


def test_paragraph_merge_with_deletions_keeps_same_seq_updates():
    """tests/test_paragraph_merge.rs:69-113: deletions that share the seq of their segment (an atomic update) remove
    nothing of it; the merge holds all four vectors."""
    config, s1, s2 = two_segments()
    deletions = [(f"{UUID1}/a/title", 2), (f"{UUID1}/t/body", 2), (f"{UUID2}/a/title", 4), (f"{UUID2}/t/body", 4)]
    merged = VectorIndexer().merge(config, [(s1, 2), (s2, 4)], deletions)
    assert merged.records == 4
    assert sorted(merged.keys) == sorted(s1.keys + s2.keys)
    assert sorted(map(tuple, merged.vectors.tolist())) == sorted(tuple(map(float, make_vector(i))) for i in range(4))


def test_merge_applies_newer_deletions_only():
    config, s1, s2 = two_segments()
    # seq 3 > segment 1's seq 2: its a/title goes; segment 2 (seq 4) is newer than the deletion and keeps everything
    merged = VectorIndexer().merge(config, [(s1, 2), (s2, 4)], [(f"{UUID1}/a/title", 3), (f"{UUID2}/a/title", 3)])
    assert merged.records == 3 and f"{UUID1}/a/title/0-10" not in merged.keys and f"{UUID2}/a/title/0-10" in merged.keys
    # a resource-level deletion (bare uuid) newer than both
    merged = VectorIndexer().merge(config, [(s1, 2), (s2, 4)], [(UUID2, 9)])
    assert sorted(merged.keys) == sorted(s1.keys)
    # a key that is no field id ("uuid/type") deletes nothing (FieldKey::from_field_id -> None, lib.rs:193-195)
    merged = VectorIndexer().merge(config, [(s1, 2), (s2, 4)], [(f"{UUID1}/a", 9), ("not-a-uuid/a/title", 9)])
    assert merged.records == 4
    # largest operand first, then the order the segments were opened in: newest first (lib.rs:104-109, segment.rs:92-94)
    big = VectorIndexer().merge(config, [(s1, 2), (s2, 4)], [])
    merged = VectorIndexer().merge(config, [(s1, 5), (big, 4)], [])
    assert merged.keys[:4] == big.keys and merged.records == 6


def test_index_resource_vectorsets_tags_and_normalisation():
    config = VectorConfig(dimension=DIMENSION, similarity=Similarity.Dot, normalize_vectors=True)
    key = f"{UUID1}/a/title/0-10"
    para = IndexParagraph(0, 10, {key: VectorSentence([3.0, 0.0, 4.0, 0.0], b"meta")},
                          {"multilingual": {key: VectorSentence([0.0, 2.0, 0.0, 0.0])}}, labels=["/l/set/x"])
    res = Resource(UUID1, labels=["/q/h", "/n/s/PROCESSED"], paragraphs={f"{UUID1}/a/title": {key: para}},
                   vector_prefixes_to_delete={"multilingual": [f"{UUID1}/a/title"]}, vectors_to_delete_in_all_vectorsets=[UUID1])
    ix = VectorIndexer()
    seg = ix.index_resource(config, res, "multilingual", False)
    assert seg.records == 1 and seg.vectors[0].tolist() == [0.0, 1.0, 0.0, 0.0]
    assert seg.tags == {"/q/h"} and seg.labels == [["/l/set/x"]] and seg.metadata == [b""]
    # an unknown vectorset: skipped, or the default sentences when asked to fall back (indexer.rs:64-76)
    assert ix.index_resource(config, res, "other", False) is None
    seg = ix.index_resource(config, res, "other", True)
    assert np.allclose(seg.vectors[0], [0.6, 0.0, 0.8, 0.0]) and seg.metadata == [b"meta"]
    assert ix.deletions_for_resource(res, "multilingual") == [f"{UUID1}/a/title"]
    assert ix.deletions_for_resource(res, "other") == [UUID1]


def test_multi_vector_merge_keeps_the_paragraph_of_every_vector():
    config = VectorConfig(dimension=2, vector_cardinality=VectorCardinality.Multi)
    ix = VectorIndexer()

    def res(uuid, sizes):
        paragraphs = {}
        for f, m in enumerate(sizes):
            key = f"{uuid}/t/f{f}/0-1"
            v = np.arange(2 * m, dtype=np.float32) + 100 * f
            paragraphs[f"{uuid}/t/f{f}"] = {key: IndexParagraph(0, 1, {key: VectorSentence(v.tolist())})}
        return Resource(uuid, paragraphs=paragraphs)

    a, b = ix.index_resource(config, res(UUID1, [2, 1, 3])), ix.index_resource(config, res(UUID2, [1, 2]))
    assert a.para_of_vec.tolist() == [0, 0, 1, 2, 2, 2] and b.para_of_vec.tolist() == [0, 1, 1]
    merged = ix.merge(config, [(a, 1), (b, 2)], [(f"{UUID1}/t/f1", 5)])
    assert merged.keys == [a.keys[0], a.keys[2]] + b.keys
    assert merged.para_of_vec.tolist() == [0, 0, 1, 1, 1, 2, 3, 3]
    assert np.array_equal(merged.vectors, np.vstack([a.vectors[[0, 1, 3, 4, 5]], b.vectors]))
