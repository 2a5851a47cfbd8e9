"""Device HNSW build (HnswBuilder, hnsw/build.rs): the graph is not unique in the reference either
(rayon inserts, segment.rs:908), so parity is structural invariants + recall, against the oracle's
sequential build and against the reference's own recall floor (segment.rs:841-912)."""
import numpy as np
import pytest

from nucliadb_amd import _lib
from nucliadb_amd.vector import Similarity, VectorConfig, VectorSearcher, VectorSearchRequest, VectorSegment

pytestmark = pytest.mark.gpu


def normalize(v):
    return (v / np.sqrt((v * v).sum(-1, keepdims=True))).astype(np.float32)


def random_vector(rng, d, n=None):
    return normalize(rng.uniform(-1, 1, (d,) if n is None else (n, d)).astype(np.float32))


def nearby(rng, close_to, distance):
    return normalize(close_to + random_vector(rng, close_to.shape[-1]) * np.float32(distance))


def clustered(rng, d, clusters, per):
    """The reference's recall recipe (segment.rs:849-862): chained centres, half the points at
    radius 0.01 and half at 0.03."""
    rows = []
    center = random_vector(rng, d)
    for _ in range(clusters):
        for _ in range(per // 2):
            rows.append(nearby(rng, center, 0.01))
        for _ in range(per // 2):
            rows.append(nearby(rng, center, 0.03))
        center = nearby(rng, center, 0.1)
    x = np.array(rows, np.float32)
    return x[rng.permutation(len(x))]  # the reference inserts in BTreeMap key order = random


def seg_of(x, graph=None):
    n = x.shape[0]
    return VectorSegment([f"k{i}" for i in range(n)], x, [[] for _ in range(n)], [b""] * n, graph=graph)


def recall_at(searcher, q, k, method):
    req = VectorSearchRequest(result_per_page=k, min_score=-1.0, with_duplicates=True)
    _, _, exact, _, ec = searcher.search_batch(req, q, method=_lib.METHOD_BRUTE_FORCE)
    _, _, got, _, gc = searcher.search_batch(req, q, method=method)
    hit = 0
    for i in range(q.shape[0]):
        hit += len(set(exact[i, : ec[i]].tolist()) & set(got[i, : gc[i]].tolist()))
    return hit / (q.shape[0] * k)


def check_invariants(orc, graph_bytes, n, levels):
    g = orc.Hnsw.deserialize_v2(np.frombuffer(graph_bytes, np.uint8))
    ep_node, ep_layer = orc.disk_v2_entry_point(np.frombuffer(graph_bytes, np.uint8))
    assert ep_layer == levels.max() and levels[ep_node] == ep_layer
    gb = np.frombuffer(graph_bytes, np.uint8)
    deg0 = np.zeros(n, np.int64)
    for layer in range(int(ep_layer) + 1):
        mmax = 60 if layer == 0 else 30
        for node in range(n):
            e = orc.disk_v2_edges(gb, layer, node)
            if levels[node] < layer:
                assert len(e) == 0, "edges on a layer the node is not in"
                continue
            assert len(e) <= mmax
            assert len(set(e.tolist())) == len(e), "duplicate edge"
            assert node not in e, "self loop"
            assert all(levels[t] >= layer for t in e), "edge to a node outside the layer (ram_hnsw.rs:123-128)"
            if layer == 0:
                deg0[node] = len(e)
    return deg0


def test_levels_match_reference_rng(orc):
    """The level draw is Xoshiro256++/SplitMix64 + round(-ln(u)/ln 30) (build.rs:97-101); the built
    graph's layer membership must equal the oracle's draw for the same seed."""
    rng = np.random.default_rng(5)
    n, d = 3000, 32
    x = random_vector(rng, d, n)
    s = VectorSearcher.open(VectorConfig(d, Similarity.Dot), [(seg_of(x), 1)])
    s.build_hnsw(0, level_seed=2)
    graph, edges = s.serialize_hnsw(0)
    s.close()
    levels = orc.hnsw_levels(2, n)
    deg0 = check_invariants(orc, graph, n, levels)
    assert (deg0 > 0).all(), "every node must be linked on layer 0"
    assert deg0.mean() > 20
    assert len(edges) > 0 and np.isfinite(edges).all()


def test_recall_clustered_data_reference_floor(orc):
    """segment.rs:841-912: 4 chained clusters x 160 vectors of 256-d, 100 nearby queries, recall@5 >= 0.95."""
    rng = np.random.default_rng(1234567890)
    d = 256
    x = clustered(rng, d, 4, 160)
    q = np.array([nearby(rng, x[rng.integers(0, len(x))], 0.05) for _ in range(100)], np.float32)
    s = VectorSearcher.open(VectorConfig(d, Similarity.Dot), [(seg_of(x), 1)])
    s.build_hnsw(0, level_seed=2)
    r = recall_at(s, q, 5, _lib.METHOD_HNSW)
    s.close()
    assert r >= 0.95, r


@pytest.mark.parametrize("sim", [Similarity.Cosine, Similarity.Dot])
def test_recall_not_below_oracle_build(orc, sim):
    """Same data, same search kernel: a graph built on the device must serve recall@10 at least as
    well (within 1 %) as the oracle's sequential reference-rule build."""
    rng = np.random.default_rng(42)
    d, k = 64, 10
    x = np.vstack([clustered(rng, d, 12, 160), random_vector(rng, d, 2000)])
    x = x[rng.permutation(len(x))]
    q = np.vstack([np.array([nearby(rng, x[rng.integers(0, len(x))], 0.05) for _ in range(150)], np.float32),
                   random_vector(rng, d, 106)])
    oseg = orc.Segment(x, similarity=sim.value)
    og, _ = oseg.build_graph(seed=2).serialize_v2(len(x))
    cfg = VectorConfig(d, sim)
    s_or = VectorSearcher.open(cfg, [(seg_of(x, bytes(og)), 1)])
    r_or = recall_at(s_or, q, k, _lib.METHOD_HNSW)
    s_or.close()
    s_gpu = VectorSearcher.open(cfg, [(seg_of(x), 1)])
    s_gpu.build_hnsw(0, level_seed=2)
    r_gpu = recall_at(s_gpu, q, k, _lib.METHOD_HNSW)
    graph, _ = s_gpu.serialize_hnsw(0)
    s_gpu.close()
    check_invariants(orc, graph, len(x), orc.hnsw_levels(2, len(x)))
    assert r_gpu >= r_or - 0.01, (r_gpu, r_or)
    assert r_gpu >= 0.9, r_gpu


def test_built_graph_is_searchable_by_the_oracle(orc):
    """Drop-in both ways: the hnsw.graph image written by the device build must be readable by the
    reference-format reader (oracle's DiskHnswV2 restatement) and give the same answers there."""
    rng = np.random.default_rng(8)
    n, d, k = 2500, 48, 10
    x = random_vector(rng, d, n)
    q = random_vector(rng, d, 20)
    s = VectorSearcher.open(VectorConfig(d, Similarity.Cosine), [(seg_of(x), 1)])
    s.build_hnsw(0, level_seed=2)
    graph, edges = s.serialize_hnsw(0)
    req = VectorSearchRequest(result_per_page=k, min_score=-1.0, with_duplicates=True)
    _, _, vec, score, count = s.search_batch(req, q, method=_lib.METHOD_HNSW)
    s.close()
    og = orc.Hnsw.deserialize_v2(np.frombuffer(graph, np.uint8), edges)
    oseg = orc.Segment(x, similarity=orc.SIM_COSINE, graph=og)
    for i in range(len(q)):
        wv, ws = oseg.hnsw_search(q[i], k)
        assert np.array_equal(vec[i, : count[i]], wv)
        assert np.array_equal(score[i, : count[i]].view(np.uint32), ws.view(np.uint32))


def test_device_graph_at_768d_is_searched_identically_by_the_oracle(orc):
    """The headline shape (D = 768, cosine) end to end: build on the device, serialise to hnsw.graph,
    load that image into the oracle, and the oracle's CPU search must return bit-identical hits."""
    rng = np.random.default_rng(77)
    n, d, k = 12000, 768, 10
    x = np.vstack([clustered(rng, d, 30, 160), random_vector(rng, d, n - 4800)])
    x = x[rng.permutation(len(x))]
    q = np.vstack([np.array([nearby(rng, x[rng.integers(0, n)], 0.05) for _ in range(12)], np.float32), random_vector(rng, d, 12)])
    s = VectorSearcher.open(VectorConfig(d, Similarity.Cosine), [(seg_of(x), 1)])
    s.build_hnsw(0, level_seed=2)
    graph, edges = s.serialize_hnsw(0)
    req = VectorSearchRequest(result_per_page=k, min_score=-1.0, with_duplicates=False)
    _, _, vec, score, count = s.search_batch(req, q, method=_lib.METHOD_HNSW)
    r = recall_at(s, q, k, _lib.METHOD_HNSW)
    s.close()
    oseg = orc.Segment(x, similarity=orc.SIM_COSINE, graph=orc.Hnsw.deserialize_v2(np.frombuffer(graph, np.uint8), edges))
    for i in range(len(q)):
        wv, ws = oseg.hnsw_search(q[i], k, with_duplicates=False)
        assert np.array_equal(vec[i, : count[i]], wv), i
        assert np.array_equal(score[i, : count[i]].view(np.uint32), ws.view(np.uint32))
    assert r >= 0.5  # half the queries are uniform random points (no structure to find)


def test_merge_with_graph_reuse(orc):
    """segment::merge (segment.rs:92-197) and its tests (segment/tests.rs:379-477): the largest operand's
    graph is reused when it has no deletions and only the appended vectors are inserted — every stored
    vector must still find itself (score >= 0.999), recall must hold, and reuse must beat a full rebuild."""
    import time

    from nucliadb_amd.vector import segment_merge

    rng = np.random.default_rng(91)
    d = 128
    big = np.vstack([clustered(rng, d, 40, 160), random_vector(rng, d, 13600)])   # 20000
    small = [random_vector(rng, d, 900), clustered(rng, d, 5, 160)]               # 900 + 800
    cfg = VectorConfig(d, Similarity.Dot)

    def seg(x, prefix):
        n = x.shape[0]
        return VectorSegment([f"{prefix}-{i}" for i in range(n)], x, [[] for _ in range(n)], [b""] * n)

    s_big = VectorSearcher.open(cfg, [(seg(big, "big"), 1)])
    s_big.build_hnsw(0)
    graph, edges = s_big.serialize_hnsw(0)
    s_big.close()
    big_seg = VectorSegment([f"big-{i}" for i in range(len(big))], big, [[] for _ in big], [b""] * len(big), graph=graph, graph_edges=edges)
    dead = np.ones(900, bool)
    dead[::7] = False                                             # deletions in a SMALL operand do not prevent reuse
    merged = segment_merge([(seg(small[0], "s0"), dead), (big_seg, None), (seg(small[1], "s1"), None)], cfg)
    n_total = len(big) + int(dead.sum()) + 800
    assert merged.records == n_total and merged.graph_nodes == len(big) and merged.keys[0] == "big-0"
    assert merged.keys[len(big)] == "s0-1"                        # largest first, then by size: s0 (900) before s1 (800)
    s = VectorSearcher.open(cfg, [(merged, 1)])
    with pytest.raises(_lib.NidxGpuError):                        # not searchable through HNSW before the new nodes are inserted
        s.search_batch(VectorSearchRequest(result_per_page=1, min_score=-1.0), merged.vectors[:1], method=_lib.METHOD_HNSW)
    t0 = time.time()
    s.extend_hnsw(0)
    t_reuse = time.time() - t0
    out_graph, _ = s.serialize_hnsw(0)
    levels = np.concatenate([orc.hnsw_levels(2, len(big)), orc.hnsw_levels(2, n_total - len(big))])  # a fresh RNG for the new nodes
    check_invariants(orc, out_graph, n_total, levels)
    # the reused part is untouched except for reverse links: node degrees of the big segment never shrink below what they were
    q = np.vstack([merged.vectors[rng.integers(0, len(big), 400)], merged.vectors[len(big):]])
    req = VectorSearchRequest(result_per_page=3, min_score=-1.0, with_duplicates=True)
    _, _, vec, score, count = s.search_batch(req, q, method=_lib.METHOD_HNSW)
    # self-match (segment/tests.rs:379-477).  HNSW at ef=30 misses ~0.7 % of uniform-random 128-d vectors
    # whichever way the graph was built (a full rebuild misses the same ones: scripts/diag_merge.py)
    assert (score[:400, 0] >= 0.999).mean() >= 0.98 and (score[400:, 0] >= 0.999).mean() >= 0.98
    r = recall_at(s, q, 3, _lib.METHOD_HNSW)
    s.close()
    full = VectorSearcher.open(cfg, [(seg(merged.vectors, "m"), 1)])
    full.build_hnsw(0)
    r_full = recall_at(full, q, 3, _lib.METHOD_HNSW)
    full.close()
    assert r >= r_full - 0.03, (r, r_full)                        # as good as a graph built from scratch on the same rows
    # a deletion in the LARGEST operand forbids reuse: the merged segment carries no graph (full rebuild)
    alive_big = np.ones(len(big), bool)
    alive_big[5] = False
    rebuilt = segment_merge([(big_seg, alive_big), (seg(small[1], "s1"), None)], cfg)
    assert rebuilt.graph is None and rebuilt.records == len(big) - 1 + 800
    s2 = VectorSearcher.open(cfg, [(rebuilt, 1)])
    t0 = time.time()
    s2.build_hnsw(0)
    t_full = time.time() - t0
    s2.close()
    assert t_full >= 1.5 * t_reuse, (t_full, t_reuse)             # segment/tests.rs:473-474


def test_build_work_counters(orc, monkeypatch):
    """nidx_gpu_vector_build_stats: the build kernels count their own work the way the search kernel does (the figures behind the
    build's roofline fraction in bench.py).  Counting must not change the graph: a build without counters (NIDX_GPU_BUILD_STATS=0)
    gives the same hnsw.graph bytes; the counters obey the structure of HnswBuilder::insert (hnsw/build.rs:97-167): one
    construction search per node reads at least its own candidates, every selected neighbour is one reverse-link request, a prune
    needs an append first."""
    import ctypes as C

    rng = np.random.default_rng(11)
    n, d = 6000, 96
    x = clustered(rng, d, n // 160 + 1, 160)[:n]

    def build(stats_on):
        if stats_on:
            monkeypatch.delenv("NIDX_GPU_BUILD_STATS", raising=False)
        else:
            monkeypatch.setenv("NIDX_GPU_BUILD_STATS", "0")
        s = VectorSearcher.open(VectorConfig(d, Similarity.Cosine), [(seg_of(x), 1)])
        s.build_hnsw(0, level_seed=2)
        st = (C.c_uint64 * 10)()
        _lib.check(_lib.lib().nidx_gpu_vector_build_stats(s._handle, st))
        graph, edges = s.serialize_hnsw(0)
        s.close()
        return [int(v) for v in st], bytes(graph), np.asarray(edges).tobytes()

    st, g1, e1 = build(True)
    st0, g0, e0 = build(False)
    assert g1 == g0 and e1 == e0, "counting the work changed the graph"
    assert st0[2] == 2**64 - 1 and st0[0] == n
    nodes, usec, evals, expansions, sel_rows, prune_rows, appends, prunes, escalated_at, end_flags = st
    assert escalated_at <= n and end_flags & ~1 == 0   # ([8]: first node inserted with the large visited table, [9]: the flags the build ended with)
    assert nodes == n and usec > 0
    assert evals >= expansions > n             # every insertion expands at least its entry point on layer 0
    assert evals / n > 100                     # ef_construction = 100 results are kept per layer: at least as many rows were scored
    assert sel_rows >= n                       # the heuristic reads at least the best candidate of every (node, layer) list
    slots = int((orc.hnsw_levels(2, n).astype(np.int64) + 1).sum())   # a node of level L is inserted into L + 1 layers
    assert 0 < appends <= 30 * slots           # at most M = 30 reverse-link requests per (node, layer)
    assert prunes <= appends and (prune_rows > 0) == (prunes > 0)
