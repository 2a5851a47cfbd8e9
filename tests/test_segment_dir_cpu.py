"""Segment directories: the C-ABI reader / writer of vectors.bin, paragraphs.bin/.pos, vectors.quant and
hnsw.graph/.edges against the pure-Python restatement of the formats in oracle/oracle.py.  Host code only: no GPU."""
import os
import uuid

import numpy as np
import pytest

from nucliadb_amd import _lib
from nucliadb_amd.vector import SegmentDir, VectorSegment

RID = [uuid.UUID(int=0x1000 + i) for i in range(4)]


@pytest.fixture(autouse=True)
def _write_index_files(monkeypatch):
    """The writers leave field.fst / label.fst / index.map out unless asked (NIDX_GPU_SEGMENT_DIR_FST=1: their byte layouts are
    restated without the crates at hand): the tests of this module exercise them, so they ask."""
    monkeypatch.setenv("NIDX_GPU_SEGMENT_DIR_FST", "1")


def corpus(rng, dimension=8):
    keys, labels, metadata = [], [], []
    fields = ["t/title", "t/title2", "a/body", "f/file/extra"]
    for r, rid in enumerate(RID):
        for f in fields[: 2 + r % 3]:
            for p in range(2):
                keys.append(f"{rid}/{f}/{p * 10}-{p * 10 + 9}")
                labels.append([f"/l/set/label_{(r + p) % 3}"] + (["/l/set"] if p else []) + (["/e/entity/x"] if r == 1 else []))
                metadata.append(bytes(rng.integers(0, 256, int(rng.integers(0, 12)), dtype=np.uint8)))
    # varint boundaries: key lengths 250 / 251, a 70 000-byte metadata blob, many labels, no labels, a key without field
    keys += [f"{RID[0]}/t/" + "k" * (250 - 39), f"{RID[0]}/t/" + "k" * (251 - 39), "not-a-uuid/t/x/0-1", str(RID[3])]
    labels += [[], [f"/l/many/{i}" for i in range(300)], ["/l/set/label_0"], ["/l/set/label_1"]]
    metadata += [b"", bytes(70000), b"\x00\xff", b"m"]
    assert len(keys[-4]) == 250 and len(keys[-3]) == 251
    vectors = rng.standard_normal((len(keys), dimension)).astype(np.float32)
    return keys, labels, metadata, vectors


def read(path, name):
    with open(os.path.join(path, name), "rb") as f:
        return f.read()


def test_writer_bytes_equal_the_format_restatement(orc, tmp_path):
    rng = np.random.default_rng(5)
    keys, labels, metadata, vectors = corpus(rng)
    seg = VectorSegment(keys, vectors, labels, metadata)
    seg.save(str(tmp_path))
    want = orc.segment_dir_files(8, vectors, None, keys, labels, metadata)
    want.update(orc.segment_dir_index_files(keys, labels))
    for name, data in want.items():
        assert read(tmp_path, name) == data, name
    assert sorted(os.listdir(tmp_path)) == ["field.fst", "index.map", "label.fst", "paragraphs.bin", "paragraphs.pos", "vectors.bin"]
    # the record layout, spelled out once: key "ab", labels ["/l/x"], metadata 01 02, first_vector 0, num_vectors 1
    one = orc.segment_dir_files(2, [[1.0, 2.0]], None, ["ab"], [["/l/x"]], [b"\x01\x02"])
    assert one["paragraphs.bin"] == b"\x02ab" + b"\x01\x04/l/x" + b"\x02\x01\x02" + b"\x00" + b"\x01"
    assert one["vectors.bin"] == np.array([1.0, 2.0], "<f4").tobytes() + b"\x00\x00\x00\x00" and one["paragraphs.pos"] == b"\x00\x00\x00\x00"
    assert orc._varint(250) == b"\xfa" and orc._varint(251) == b"\xfb\xfb\x00" and orc._varint(65536) == b"\xfc\x00\x00\x01\x00"


def test_reader_decodes_reference_layout_and_rebuilds_the_inverted_indexes(orc, tmp_path):
    rng = np.random.default_rng(6)
    keys, labels, metadata, vectors = corpus(rng)
    # multi-vector paragraphs: paragraph a owns 1 + a % 3 contiguous rows
    pov = np.repeat(np.arange(len(keys), dtype=np.uint32), [1 + a % 3 for a in range(len(keys))])
    rows = rng.standard_normal((len(pov), 8)).astype(np.float32)
    for name, data in orc.segment_dir_files(8, rows, pov, keys, labels, metadata).items():
        with open(tmp_path / name, "wb") as f:
            f.write(data)
    with SegmentDir(str(tmp_path), 8) as d:
        assert not d.indexes_from_files   # no index.map: ParagraphInvertedIndexes::build's rebuild (segment.rs:49-67)
        s = d.segment_c()
        assert s.n_vectors == len(pov) and s.n_paragraphs == len(keys) and s.row_stride_bytes == 36
        assert not s.hnsw_graph_len and not s.quantized_len and not s.alive_bitset
        want = orc.parse_paragraphs(read(tmp_path, "paragraphs.bin"), read(tmp_path, "paragraphs.pos"))
        for a in range(len(keys)):
            assert d.paragraph(a) == want[a] == (keys[a], labels[a], metadata[a], int(np.flatnonzero(pov == a)[0]), 1 + a % 3)
        key_ids = np.ctypeslib.as_array(_lib.C.cast(s.paragraph_key_ids, _lib.C.POINTER(_lib.C.c_uint64)), (len(keys),))
        assert len(set(key_ids.tolist())) == len(set(keys))
        seg = d.to_segment()
        assert np.array_equal(seg.vectors, rows) and np.array_equal(seg.para_of_vec, pov) and seg.keys == keys and seg.labels == labels
        # label lookups = prefix search on labels_key (inverted_index/paragraph.rs:63-66,144-146)
        def label_paragraphs(label):
            ids = set()
            for l in d.lists(_lib.LIST_LABEL, label):
                ids.update(d.posting_list(l).tolist())
            return sorted(ids)
        for label in ["/l/set/label_0", "/l/set", "/l", "/e/entity", "/l/many/7", "/l/se", "/zzz"]:
            want_ids = [a for a in range(len(keys)) if any((l[1:] + "/").startswith(label[1:] + "/") for l in labels[a])]
            assert label_paragraphs(label) == want_ids, label
        # field lookups: exact get for KeyPrefixSet, byte-prefix get_prefix for deletions (title also reaches title2)
        def field_paragraphs(key, prefix):
            ids = set()
            for l in d.lists(_lib.LIST_FIELD, key, prefix):
                ids.update(d.posting_list(l).tolist())
            return sorted(ids)
        r0 = str(RID[0])
        title = [a for a, k in enumerate(keys) if k.startswith(f"{r0}/t/title/")]
        title2 = [a for a, k in enumerate(keys) if k.startswith(f"{r0}/t/title2/")]
        assert title and title2
        assert field_paragraphs(f"{r0}/t/title", False) == title
        assert field_paragraphs(f"{RID[0].hex}/t/title", False) == title          # simple form of the uuid
        assert field_paragraphs(f"{r0}/t/title", True) == sorted(title + title2)
        assert field_paragraphs(r0, True) == [a for a, k in enumerate(keys) if k.startswith(r0 + "/")]
        assert field_paragraphs(r0, False) == []                                   # a bare resource key is not stored
        assert field_paragraphs(f"{r0}/t", True) == [] and field_paragraphs("garbage/t/x", True) == []
        # the Python mirror's deletion lookup agrees
        for key in [f"{r0}/t/title", r0, f"{RID[2]}/a/body", "garbage", f"{r0}/t"]:
            assert seg.ids_for_deletion_key(key) == field_paragraphs(key, True), key
        # "not-a-uuid/…" is in no field list; a key that is a bare uuid is stored under the 16-byte resource key
        # (FieldKey::from_field_id accepts it, utils.rs:109-113)
        fi = d.filter_index_c()
        all_field = set()
        for l in range(fi.n_lists):
            if l not in set(d.lists(_lib.LIST_LABEL, "/l")) | set(d.lists(_lib.LIST_LABEL, "/e")):
                all_field.update(d.posting_list(l).tolist())
        assert keys.index("not-a-uuid/t/x/0-1") not in all_field and keys.index(str(RID[3])) in all_field
        assert field_paragraphs(str(RID[3]), False) == [keys.index(str(RID[3]))]


def test_round_trip_with_graph_and_quantized_store(tmp_path):
    rng = np.random.default_rng(8)
    n, D = 37, 64
    vectors = rng.standard_normal((n, D)).astype(np.float32)
    keys = [f"{RID[i % 4]}/a/body/{i}-{i + 1}" for i in range(n)]
    graph = bytes(rng.integers(0, 256, 997, dtype=np.uint8))   # opaque here
    edges = rng.random(211).astype(np.float32)
    quant = rng.integers(0, 256, (n, D // 8 + 8), dtype=np.uint8)
    seg = VectorSegment(keys, vectors, [[] for _ in keys], [b"" for _ in keys], graph=graph, graph_edges=edges, quantized=quant)
    seg.save(str(tmp_path))
    assert sorted(os.listdir(tmp_path)) == ["field.fst", "hnsw.edges", "hnsw.graph", "index.map", "label.fst", "paragraphs.bin", "paragraphs.pos",
                                            "vectors.bin", "vectors.quant"]
    assert read(tmp_path, "hnsw.graph") == graph and read(tmp_path, "hnsw.edges") == edges.tobytes() and read(tmp_path, "vectors.quant") == quant.tobytes()
    back = VectorSegment.load(str(tmp_path), D)
    assert back.keys == keys and np.array_equal(back.vectors, vectors) and back.graph == graph and back.para_of_vec is None
    assert np.array_equal(back.graph_edges, edges) and np.array_equal(back.quantized, quant)


def test_errors(tmp_path):
    with pytest.raises(_lib.NidxGpuError) as e:
        SegmentDir(str(tmp_path / "missing"), 8)
    assert e.value.code == _lib.NIDX_ERR_IO
    seg = VectorSegment(["k"], np.ones((1, 8), np.float32), [["/l/a"]], [b"meta"])
    seg.save(str(tmp_path))
    with pytest.raises(_lib.NidxGpuError) as e:      # dimension mismatch: the file is not a whole number of records
        SegmentDir(str(tmp_path), 7)
    assert e.value.code == _lib.NIDX_ERR_INCONSISTENT_DIMENSIONS
    data = read(tmp_path, "paragraphs.bin")
    with open(tmp_path / "paragraphs.bin", "wb") as f:
        f.write(data[:-3])
    with pytest.raises(_lib.NidxGpuError) as e:
        SegmentDir(str(tmp_path), 8)
    assert e.value.code == _lib.NIDX_ERR_IO and "truncated" in str(e.value)
    with open(tmp_path / "nodes.kv", "wb") as f:     # DataStoreV1 takes precedence (segment.rs:41); this one is not a store
        f.write(b"x")
    with pytest.raises(_lib.NidxGpuError) as e:
        SegmentDir(str(tmp_path), 8)
    assert e.value.code == _lib.NIDX_ERR_IO and "nodes.kv" in str(e.value)
    os.remove(tmp_path / "nodes.kv")
    # vectors of one paragraph must be contiguous
    bad = VectorSegment(["a", "b"], np.ones((3, 8), np.float32), [[], []], [b"", b""], para_of_vec=np.array([0, 1, 0], np.uint32))
    with pytest.raises(_lib.NidxGpuError):
        bad.save(str(tmp_path))


def test_corrupt_records_are_io_errors_never_reads_past_the_file(tmp_path):
    seg = VectorSegment(["k" * 20, "j" * 20], np.ones((2, 8), np.float32), [["/l/a", "/l/b"], []], [b"meta", b""])
    seg.save(str(tmp_path))
    good = read(tmp_path, "paragraphs.bin")
    SegmentDir(str(tmp_path), 8).close()
    # a length of 2^64 - 1 in front of the key / the label count / a label / the metadata (would wrap `at + n`)
    huge = b"\xfd" + b"\xff" * 8
    for cut in (0, 21, 22, 27, 32):
        with open(tmp_path / "paragraphs.bin", "wb") as f:
            f.write(good[:cut] + huge + good[cut + 1:])
        with pytest.raises(_lib.NidxGpuError) as e:
            SegmentDir(str(tmp_path), 8)
        assert e.value.code == _lib.NIDX_ERR_IO, cut
    # every single-byte corruption either still parses or is reported; nothing crashes
    rng = np.random.default_rng(11)
    for _ in range(300):
        b = bytearray(good)
        b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        with open(tmp_path / "paragraphs.bin", "wb") as f:
            f.write(bytes(b))
        try:
            SegmentDir(str(tmp_path), 8).close()
        except _lib.NidxGpuError as e:
            assert e.code == _lib.NIDX_ERR_IO
    # a record start beyond the file
    with open(tmp_path / "paragraphs.bin", "wb") as f:
        f.write(good)
    with open(tmp_path / "paragraphs.pos", "wb") as f:
        f.write(np.array([0, 1 << 31], "<u4").tobytes())
    with pytest.raises(_lib.NidxGpuError):
        SegmentDir(str(tmp_path), 8)


def test_writer_streams_vectors_bin_in_chunks(orc, tmp_path):
    # more rows than one 4 MiB write chunk holds (stride 36 B -> 116 508 rows per chunk)
    rng = np.random.default_rng(12)
    n = 250_000
    vectors = rng.standard_normal((n, 8)).astype(np.float32)
    pov = np.repeat(np.arange(n // 2, dtype=np.uint32), 2)
    keys = ["k%d" % i for i in range(n // 2)]
    VectorSegment(keys, vectors, [[] for _ in keys], [b"" for _ in keys], para_of_vec=pov).save(str(tmp_path))
    got = np.frombuffer(read(tmp_path, "vectors.bin"), np.uint8).reshape(n, 36)
    assert np.array_equal(got[:, :32].copy().view("<f4"), vectors)
    assert np.array_equal(got[:, 32:].copy().view("<u4")[:, 0], pov)


def test_native_directory_merge_equals_the_mirror_merge(tmp_path):
    """nidx_gpu_segment_dir_merge (segment::merge / DataStoreV2::merge file output) against segment_merge + save of the host
    mirror: same bytes in every file, same graph-reuse decision."""
    from nucliadb_amd.vector import VectorConfig, segment_dir_merge, segment_merge

    rng = np.random.default_rng(21)
    cfg = VectorConfig(dimension=64)

    def make(n, tag, quant=True, graph=False, multi=False):
        keys = [f"{RID[i % 4]}/t/{tag}{i}/0-{i}" for i in range(n)]
        labels = [[f"/l/{tag}/{i % 3}"] * (i % 2) + ["/e/x"] * (i % 5 == 0) for i in range(n)]
        metadata = [bytes(rng.integers(0, 256, i % 7, dtype=np.uint8)) for i in range(n)]
        pov = np.repeat(np.arange(n, dtype=np.uint32), 1 + (np.arange(n) % 3 == 0)) if multi else None
        nv = n if pov is None else len(pov)
        vectors = rng.standard_normal((nv, 64)).astype(np.float32)
        q = rng.integers(0, 256, (nv, 64 // 8 + 8), dtype=np.uint8) if quant else None
        g = (bytes(rng.integers(0, 256, 40, dtype=np.uint8)), rng.standard_normal(6).astype(np.float32)) if graph else (None, None)
        return VectorSegment(keys, vectors, labels, metadata, quantized=q, graph=g[0], graph_edges=g[1], para_of_vec=pov)

    def check(segs_alive, name):
        dirs = []
        for i, (seg, _) in enumerate(segs_alive):
            d = tmp_path / f"{name}_op{i}"
            d.mkdir()
            seg.save(str(d))
            dirs.append(SegmentDir(str(d), 64))
        want_dir, got_dir = tmp_path / f"{name}_want", tmp_path / f"{name}_got"
        want_dir.mkdir(), got_dir.mkdir()
        want = segment_merge(segs_alive, cfg)
        covered, want.graph_nodes = want.graph_nodes, 0   # the mirror refuses to save a partial graph; the files are the same
        want.save(str(want_dir))
        want.graph_nodes = covered
        rec, vec, gn, hq = segment_dir_merge(str(got_dir), 64, [(d, alive) for d, (_, alive) in zip(dirs, segs_alive)])
        assert rec == want.records and vec == want.vectors.shape[0]
        assert hq == (want.quantized is not None)
        # the mirror says "graph covers the first graph_nodes vectors (0 = all of them)"; the ABI reports the covered count
        assert gn == (0 if want.graph is None else (want.graph_nodes or want.vectors.shape[0]))
        assert sorted(os.listdir(got_dir)) == sorted(os.listdir(want_dir))
        for f in os.listdir(want_dir):
            assert read(got_dir, f) == read(want_dir, f), (name, f)
        for d in dirs:
            d.close()
        return want

    small, big, mid = make(5, "s"), make(9, "b", graph=True), make(7, "m")
    dead = np.ones(5, bool)
    dead[[1, 3]] = False
    # largest operand intact: its graph is carried over; a smaller operand loses two paragraphs
    w = check([(small, dead), (big, None), (mid, None)], "reuse")
    assert w.graph is not None and w.records == 3 + 9 + 7
    # the largest operand has a deletion: no graph
    hole = np.ones(9, bool)
    hole[0] = False
    w = check([(big, hole), (mid, None)], "rebuild")
    assert w.graph is None
    # one operand without codes: the merged directory has no vectors.quant (re-encoded on the device afterwards)
    w = check([(make(4, "q", quant=False), None), (mid, None)], "noquant")
    assert w.quantized is None
    # equal sizes keep their order; an operand that is entirely deleted; multi-vector paragraphs
    check([(make(6, "x"), None), (make(6, "y"), np.zeros(6, bool)), (make(6, "z"), None)], "ties")
    mdead = np.ones(8, bool)
    mdead[[0, 5]] = False
    check([(make(8, "u", multi=True), mdead), (make(10, "v", multi=True, graph=True), None)], "multi")
    # zero operands
    with pytest.raises(_lib.NidxGpuError) as e:
        segment_dir_merge(str(tmp_path), 64, [])
    assert e.value.code == _lib.NIDX_ERR_EMPTY_MERGE


def test_apply_deletions_equals_the_mirror(tmp_path):
    """nidx_gpu_segment_dir_apply_deletions (OpenSegment::apply_deletions, segment.rs:428-445) against
    VectorSegment.ids_for_deletion_key: resource keys, field keys, the byte-prefix reach (title -> title2), non-keys."""
    rng = np.random.default_rng(31)
    keys, labels, metadata, vectors = corpus(rng)
    seg = VectorSegment(keys, vectors, labels, metadata)
    seg.save(str(tmp_path))
    cases = [[str(RID[0])], [RID[1].hex], [f"{RID[2]}/t/title"], [f"{RID[2]}/t/title2"], [f"{RID[1]}/a/body", f"{RID[3]}/t/title"],
             [f"{RID[0]}/t"], ["not-a-uuid/t/title"], [f"{RID[3]}/f/file"], [f"{RID[3]}/f/file/extra"], [], [str(RID[0]), str(RID[0])]]
    with SegmentDir(str(tmp_path), 8) as d:
        for dels in cases:
            want = np.ones(seg.records, bool)
            for k in dels:
                want[seg.ids_for_deletion_key(k)] = False
            got = d.apply_deletions(dels)
            assert np.array_equal(got, want), dels
        # something was actually deleted in the interesting cases, and title reaches title2
        assert not d.apply_deletions([str(RID[0])]).all() and d.apply_deletions(["not-a-uuid/t/title"]).all()
        t1, t2 = d.apply_deletions([f"{RID[2]}/t/title"]), d.apply_deletions([f"{RID[2]}/t/title2"])
        assert (~t1).sum() > (~t2).sum() > 0 and not (t1 & ~t2).any()
        # an already-dead paragraph stays dead and is not counted twice
        start = np.ones(seg.records, bool)
        start[:5] = False
        assert not d.apply_deletions([str(RID[0])], start)[:5].any()


# ---- field.fst / label.fst / index.map ---------------------------------------------------------------------------------------
def all_lists(d):
    fi = d.filter_index_c()
    return [d.posting_list(l).tolist() for l in range(fi.n_lists)]


def lookups(d, keys):
    out = []
    for label in ["/l/set/label_0", "/l/set", "/l", "/e/entity", "/l/many/7", "/l/se", "/zzz"]:
        out.append([d.posting_list(l).tolist() for l in d.lists(_lib.LIST_LABEL, label)])
    for key in [f"{RID[0]}/t/title", str(RID[0]), f"{RID[2]}/a/body", "garbage", str(RID[3])]:
        for prefix in (False, True):
            out.append([d.posting_list(l).tolist() for l in d.lists(_lib.LIST_FIELD, key, prefix)])
    return out


def test_directory_opened_through_its_index_files_equals_the_rebuild(orc, tmp_path):
    rng = np.random.default_rng(31)
    keys, labels, metadata, vectors = corpus(rng)
    VectorSegment(keys, vectors, labels, metadata).save(str(tmp_path))
    with SegmentDir(str(tmp_path), 8) as d:
        assert d.indexes_from_files
        from_files = (all_lists(d), lookups(d, keys))
        dead = d.apply_deletions([f"{RID[0]}/t/title", str(RID[2])])
    # the same files through the independent restatement: every key -> its record in index.map
    files = orc.segment_dir_index_files(keys, labels)
    n_lists = 0
    for name in ("field.fst", "label.fst"):
        for key, pos in orc.fst_read(read(tmp_path, name)):
            assert orc.index_map_read(read(tmp_path, "index.map"), pos) == orc.index_map_read(files["index.map"], dict(orc.fst_read(files[name]))[key])
            n_lists += 1
    assert n_lists == len(from_files[0])
    os.remove(tmp_path / "index.map")   # InvertedIndexes::exists is false: rebuild
    with SegmentDir(str(tmp_path), 8) as d:
        assert not d.indexes_from_files
        assert (all_lists(d), lookups(d, keys)) == from_files
        assert np.array_equal(d.apply_deletions([f"{RID[0]}/t/title", str(RID[2])]), dead)


def test_damaged_index_files_fall_back_to_the_rebuild(tmp_path, monkeypatch):
    rng = np.random.default_rng(32)
    keys, labels, metadata, vectors = corpus(rng)
    VectorSegment(keys, vectors, labels, metadata).save(str(tmp_path))
    with SegmentDir(str(tmp_path), 8) as d:
        want = all_lists(d)
    good = {name: read(tmp_path, name) for name in ("field.fst", "label.fst", "index.map")}

    def reopen(exact=True):
        with SegmentDir(str(tmp_path), 8) as d:
            got = all_lists(d)
            assert got == want or (not exact and d.indexes_from_files and len(got) == len(want))
            return d.indexes_from_files

    for name, data in good.items():
        damaged = [data[: len(data) // 2], b"", data + b"\0"] if name != "index.map" else [data[: len(data) // 2], b""]
        for _ in range(40):
            b = bytearray(data)
            b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
            damaged.append(bytes(b))
        for bad in damaged:
            with open(tmp_path / name, "wb") as f:
                f.write(bad)
            # whichever way it was opened, the lists are the right ones — except that index.map carries no checksum (in the
            # reference neither): a flipped address bit that still decodes to an ascending list of stored paragraphs is read
            reopen(exact=name != "index.map")
            if name != "index.map" and bad != data:
                assert not reopen(), name   # an fst image with any byte changed fails its checksum
        with open(tmp_path / name, "wb") as f:
            f.write(data)
    assert reopen()
    # index.map that decodes, but to lists that are not lists of this store: an address beyond the paragraphs
    from nucliadb_amd.vector import fst_map_entries
    first_key, pos = fst_map_entries(good["label.fst"])[0]
    b = bytearray(good["index.map"])
    b[pos + 8 + 1] = 0xFF   # (the list's first address byte, behind the u64 count and one control byte)
    with open(tmp_path / "index.map", "wb") as f:
        f.write(bytes(b))
    assert not reopen()
    with open(tmp_path / "index.map", "wb") as f:
        f.write(good["index.map"])
    monkeypatch.setenv("NIDX_GPU_SEGMENT_DIR_FST", "0")
    assert not reopen()
    other = tmp_path / "plain"
    other.mkdir()
    VectorSegment(keys, vectors, labels, metadata).save(str(other))
    assert sorted(os.listdir(other)) == ["paragraphs.bin", "paragraphs.pos", "vectors.bin"]
    # the default (variable unset): files that are there are read, new directories are written without them
    monkeypatch.delenv("NIDX_GPU_SEGMENT_DIR_FST")
    assert reopen()
    third = tmp_path / "default"
    third.mkdir()
    VectorSegment(keys, vectors, labels, metadata).save(str(third))
    assert sorted(os.listdir(third)) == ["paragraphs.bin", "paragraphs.pos", "vectors.bin"]


def test_fst_and_index_map_containers(orc):
    from nucliadb_amd.vector import fst_map_build, fst_map_entries, fst_map_get, index_map_read

    assert orc._crc32c(b"123456789") == 0xE3069283   # CRC-32C check value
    # stream-vbyte: lengths 1,1,2,2 | 3,4,4,1 -> control bytes 0x50, 0x3e
    ids = [0, 255, 256, 65535, 65536, 1 << 24, (1 << 32) - 1, 7]
    rec = orc.index_map_record(ids)
    assert rec == bytes.fromhex("0800000000000000" "503e" "00ff" "0001ffff" "000001" "00000001" "ffffffff" "07")
    assert index_map_read(b"junk" + rec, 4).tolist() == ids == orc.index_map_read(b"junk" + rec, 4)
    assert index_map_read(orc.index_map_record([]), 0).tolist() == []
    with pytest.raises(_lib.NidxGpuError):
        index_map_read(rec[:-1], 0)
    # the writer's image of a small map, byte by byte (trie, general node encoding, value in the last node's final output)
    entries = [(b"a", 0), (b"ab", 5), (b"abc", 70000), (b"b", 1 << 40), (b"zzz", 3)]
    image = fst_map_build(entries)
    body = bytes.fromhex(
        "0300000000000000" "0000000000000000"      # version 3, type 0
        "701101" "03" "00" "40"                    # @21 "abc": final output 70000 (3 bytes), sizes 0|3, 0 transitions, final
        "05" "00" "01" "63" "11" "41"              # @27 "ab": final output 5, c -> delta 1 (out 0), sizes 1|1, final + 1
        "01" "62" "10" "41"                        # @31 "a": b -> delta 1, sizes 1|0, final + 1 (final output 0)
        "000000000001" "06" "00" "40"              # @40 "b": final output 2^40
        "03" "01" "00" "40"                        # @44 "zzz"
        "01" "7a" "10" "01" "01" "7a" "10" "01"    # @48 "zz", @52 "z"
        "010d16" "7a6261" "10" "03"                # @60 root: z, b, a (last first) -> deltas 1, 13, 22
        "0500000000000000" "3c00000000000000")     # 5 keys, root address 60
    s = orc._crc32c(body)
    assert image == body + ((((s >> 15) | (s << 17)) + 0xA282EAD8) & 0xFFFFFFFF).to_bytes(4, "little") == orc.fst_image(entries)
    assert fst_map_entries(image) == entries == orc.fst_read(image)
    assert [fst_map_get(image, k) for k in (b"ab", b"abc", b"b", b"", b"abd", b"zz", b"zzzz")] == [5, 70000, 1 << 40, None, None, None, None]
    with pytest.raises(_lib.NidxGpuError):
        fst_map_build([(b"b", 1), (b"a", 2)])
    with pytest.raises(_lib.NidxGpuError):
        fst_map_build([(b"a", 1), (b"a", 2)])
    # images the crate's own builder would write use its two compact single-transition encodings and put outputs on the
    # transitions (shared prefixes of the values); hand-assembled:
    def finish(nodes, n_keys, root):
        b = (3).to_bytes(8, "little") + bytes(8) + nodes + n_keys.to_bytes(8, "little") + root.to_bytes(8, "little")
        c = orc._crc32c(b)
        return b + ((((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF).to_bytes(4, "little")
    #   "ab" -> 7, "ac" -> 300: the node behind 'a' has b (out 0) and c (out 293), both to the empty final node (address 0);
    #   the root is a OneTrans node on the common input 'a' (index 5) with output 7
    img = finish(bytes.fromhex("2501" "0000" "00" "00" "63" "62" "12" "02") + bytes.fromhex("07" "01" "11" "85"), 2, 29)
    assert fst_map_entries(img) == [(b"ab", 7), (b"ac", 300)] == orc.fst_read(img)
    assert fst_map_get(img, b"ac") == 300 and fst_map_get(img, b"a") is None
    #   "xyz" -> 0: z = OneTrans to the empty final node, y and x = OneTransNext ("the node right below"), all common inputs
    img = finish(bytes.fromhex("00" "10" "b6") + b"\xdd" + b"\xea", 1, 20)
    assert fst_map_entries(img) == [(b"xyz", 0)] == orc.fst_read(img)
    #   the same with inputs that are not in the common table (stored in the byte below the state byte, the pack sizes below it): 01 02 03 -> 9
    img = finish(bytes.fromhex("09" "00" "11" "03" "80") + bytes.fromhex("02" "c0") + bytes.fromhex("01" "c0"), 1, 24)
    assert fst_map_entries(img) == [(b"\x01\x02\x03", 9)] == orc.fst_read(img)
    # random maps: both writers agree byte for byte, both readers read both; a root with all 256 first bytes (count stored as 1),
    # nodes with more than 32 transitions (the 256-byte index), values of every width, a value of 0 on a leaf (empty node)
    import random
    rng = random.Random(2)
    for trial in range(12):
        alphabet = b"abc/xyz" if trial % 2 else bytes(range(256))
        keys = sorted({bytes(rng.choice(alphabet) for _ in range(rng.randrange(1, 8))) for _ in range(rng.choice([0, 1, 5, 300, 3000]))})
        ents = [(k, rng.randrange(1 << rng.randrange(1, 64)) if rng.random() < 0.9 else 0) for k in keys]
        a, b = fst_map_build(ents), orc.fst_image(ents)
        assert a == b and orc.fst_read(a) == ents and fst_map_entries(b) == ents
        for k, v in ents[:50]:
            assert fst_map_get(a, k) == v
    # a damaged image is an error (or still a map), never a crash or an endless walk
    good = fst_map_build([(b"key%03d" % i, i * 1000) for i in range(200)])
    r = np.random.default_rng(5)
    for _ in range(400):
        b = bytearray(good)
        for _ in range(int(r.integers(1, 4))):
            b[int(r.integers(0, len(b)))] = int(r.integers(0, 256))
        try:
            fst_map_entries(bytes(b[: int(r.integers(0, len(b) + 1))] if r.random() < 0.2 else b))
        except _lib.NidxGpuError as e:
            assert e.code == _lib.NIDX_ERR_IO


# ---- pre-migration segments: nodes.kv (DataStoreV1) + index.hnsw (DiskHnswV1) ---------------------------------------------------
def v1_corpus(orc, rng, n=260, d=16):
    x = rng.standard_normal((n, d)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    keys = [f"{RID[i % 4]}/{'t/title' if i % 3 else 'a/body2'}/{i}-{i + 3}" for i in range(n)]
    labels = [[f"/l/set/{i % 5}"] * (i % 3 != 0) + ["/e/x/y"] * (i % 11 == 0) + ["/l/set"] * (i % 7 == 0) for i in range(n)]
    labels[5] = ["WORD1", "WORD2", "WORD3", "ORD1", "BAD", "GOOD"]   # the dictionary of the reference's own trie test (v1/trie.rs:113-134)
    metadata = [bytes(rng.integers(0, 256, i % 9, dtype=np.uint8)) for i in range(n)]
    g = orc.Segment(x, similarity=orc.SIM_DOT).build_graph(2)
    gbytes, edges = g.serialize_v2(n)
    return x, keys, labels, metadata, bytes(gbytes), np.asarray(edges, np.float32)


def write_v1(orc, path, x, keys, labels, metadata, gbytes, edges):
    with open(path / "nodes.kv", "wb") as f:
        f.write(orc.nodes_kv_bytes(x.shape[1], x, keys, labels, metadata))
    layers, entry = orc.parse_hnsw_v2(gbytes, edges, len(keys))
    with open(path / "index.hnsw", "wb") as f:
        f.write(orc.disk_hnsw_v1_bytes(len(keys), layers, entry))


def test_pre_migration_directory_opens_as_the_same_segment(orc, tmp_path):
    """A DataStoreV1 + DiskHnswV1 directory (segment::open's first branch, segment.rs:41-57; open_disk_hnsw's fallback, hnsw/disk.rs:25-32)
    against the same segment in the current formats: records, vectors, inverted indexes, deletions and the graph image byte for byte."""
    rng = np.random.default_rng(41)
    x, keys, labels, metadata, gbytes, edges = v1_corpus(orc, rng)
    v1, v2 = tmp_path / "v1", tmp_path / "v2"
    v1.mkdir(), v2.mkdir()
    write_v1(orc, v1, x, keys, labels, metadata, gbytes, edges)
    VectorSegment(keys, x, labels, metadata, graph=gbytes, graph_edges=edges).save(str(v2))
    # the layout, spelled out once: key "ab", labels ["L1"], metadata "M", one f32
    one = orc.node_v1_bytes("ab", np.array([1.0], "<f4").tobytes(), ["L1"], b"M")
    trie = orc.label_trie_bytes(["L1"])
    assert trie == (8 + 3 * 9 + 2 * 9 + 3 * 8).to_bytes(8, "little") + b"\0" + (1).to_bytes(8, "little") + b"L" + (1).to_bytes(8, "little") \
        + b"\0" + (1).to_bytes(8, "little") + b"1" + (2).to_bytes(8, "little") + b"\1" + bytes(8) \
        + (44).to_bytes(8, "little") + (26).to_bytes(8, "little") + (8).to_bytes(8, "little")
    assert one == b"".join(v.to_bytes(8, "little") for v in (32 + 1 + 3 + 8 + 4 + 8 + 2 + len(trie), 33, 33 + 3 + 12, 33 + 3 + 12 + 10)) + b"M" \
        + (4).to_bytes(4, "little") + (3).to_bytes(4, "little") + bytes(3) + np.array([1.0], "<f4").tobytes() + (2).to_bytes(8, "little") + b"ab" + trie
    with SegmentDir(str(v1), x.shape[1]) as a, SegmentDir(str(v2), x.shape[1]) as b:
        sa, sb = a.to_segment(), b.to_segment()
        assert sa.keys == sb.keys == keys and np.array_equal(sa.vectors, sb.vectors) and sa.para_of_vec is None
        assert [sorted(l) for l in sa.labels] == [sorted(set(l)) for l in labels] and sa.metadata == sb.metadata == metadata
        assert sorted(sa.labels[5]) == sorted(labels[5])
        assert sa.graph == sb.graph == gbytes and np.array_equal(sa.graph_edges, sb.graph_edges)
        assert not a.indexes_from_files and b.indexes_from_files   # (the V1 directory has no index files: rebuilt)
        assert all_lists(a) == all_lists(b)
        dead = [f"{RID[1]}/t/title", str(RID[2])]
        assert np.array_equal(a.apply_deletions(dead), b.apply_deletions(dead))
        for addr in (0, 5, len(keys) - 1):
            pa, pb = a.paragraph(addr), b.paragraph(addr)
            assert pa[0] == pb[0] and sorted(pa[1]) == sorted(set(pb[1])) and pa[2:] == pb[2:]
    # the reference's own migration test (hnsw/disk.rs:52-108): three nodes, node 0 alone (and without edges) in layer 1, entry (0, 1)
    layers = [{0: [(1, 0.5), (2, 0.2)], 1: [(0, 0.5), (1, 0.7)], 2: [(0, 0.2)]}, {0: []}]
    tiny = tmp_path / "tiny"
    tiny.mkdir()
    with open(tiny / "nodes.kv", "wb") as f:
        f.write(orc.nodes_kv_bytes(2, np.eye(3, 2, dtype=np.float32), ["a", "b", "c"], [[], [], []], [b"", b"", b""]))
    with open(tiny / "index.hnsw", "wb") as f:
        f.write(orc.disk_hnsw_v1_bytes(3, layers, (0, 1)))
    with SegmentDir(str(tiny), 2) as t:
        st = t.to_segment()
        got, entry = orc.parse_hnsw_v2(st.graph, st.graph_edges, 3)
        assert entry == (0, 1) and got[0] == {n: [(to, float(np.float32(w))) for to, w in es] for n, es in layers[0].items()} and got[1] == {}
    # an empty graph is an empty index.hnsw; a directory with nodes.kv only has no graph
    with open(tiny / "index.hnsw", "wb") as f:
        f.write(b"")
    with SegmentDir(str(tiny), 2) as t:
        assert t.to_segment().graph is None


def test_pre_migration_directory_merges_into_the_current_formats(orc, tmp_path):
    from nucliadb_amd.vector import segment_dir_merge

    rng = np.random.default_rng(42)
    x, keys, labels, metadata, gbytes, edges = v1_corpus(orc, rng, n=90)
    labels = [sorted(set(l)) for l in labels]   # (a V1 node stores its labels as a trie: a set, in trie order)
    y = rng.standard_normal((40, 16)).astype(np.float32)
    other = VectorSegment([f"{RID[3]}/f/file/{i}-{i + 1}" for i in range(40)], y, [["/l/other"]] * 40, [b"m"] * 40)
    dirs = {}
    for name in ("v1", "v2", "other", "out1", "out2"):
        dirs[name] = tmp_path / name
        dirs[name].mkdir()
    write_v1(orc, dirs["v1"], x, keys, labels, metadata, gbytes, edges)
    VectorSegment(keys, x, labels, metadata, graph=gbytes, graph_edges=edges).save(str(dirs["v2"]))
    other.save(str(dirs["other"]))
    alive = np.ones(90, bool)
    alive[[3, 50]] = False
    with SegmentDir(str(dirs["v1"]), 16) as a, SegmentDir(str(dirs["v2"]), 16) as b, SegmentDir(str(dirs["other"]), 16) as o:
        r1 = segment_dir_merge(str(dirs["out1"]), 16, [(a, alive), (o, None)])
        r2 = segment_dir_merge(str(dirs["out2"]), 16, [(b, alive), (o, None)])
    assert r1 == r2 and sorted(os.listdir(dirs["out1"])) == sorted(os.listdir(dirs["out2"]))
    with SegmentDir(str(dirs["out1"]), 16) as m1, SegmentDir(str(dirs["out2"]), 16) as m2:
        s1, s2 = m1.to_segment(), m2.to_segment()
        assert s1.keys == s2.keys and np.array_equal(s1.vectors, s2.vectors) and s1.metadata == s2.metadata
        assert [sorted(l) for l in s1.labels] == [sorted(l) for l in s2.labels] and all_lists(m1) == all_lists(m2)


def test_damaged_pre_migration_files_are_io_errors(orc, tmp_path):
    rng = np.random.default_rng(43)
    x, keys, labels, metadata, gbytes, edges = v1_corpus(orc, rng, n=60)
    write_v1(orc, tmp_path, x, keys, labels, metadata, gbytes, edges)
    SegmentDir(str(tmp_path), 16).close()
    with pytest.raises(_lib.NidxGpuError) as e:   # the vectors of the store do not have the index's dimension
        SegmentDir(str(tmp_path), 8)
    assert e.value.code == _lib.NIDX_ERR_IO
    for name in ("nodes.kv", "index.hnsw"):
        good = read(tmp_path, name)
        for trial in range(250):
            b = bytearray(good)
            for _ in range(int(rng.integers(1, 4))):
                at = int(rng.integers(0, len(b)))
                if trial % 3 == 0:   # a whole u64 field
                    b[at - at % 8: at - at % 8 + 8] = int(rng.integers(0, 1 << 62)).to_bytes(8, "little")
                else:
                    b[at] = int(rng.integers(0, 256))
            cut = int(rng.integers(0, len(b) + 1)) if trial % 5 == 0 else len(b)
            with open(tmp_path / name, "wb") as f:
                f.write(bytes(b[:cut]))
            try:
                SegmentDir(str(tmp_path), 16).close()
            except _lib.NidxGpuError as err:
                assert err.code in (_lib.NIDX_ERR_IO, _lib.NIDX_ERR_INCONSISTENT_DIMENSIONS)
        with open(tmp_path / name, "wb") as f:
            f.write(good)
