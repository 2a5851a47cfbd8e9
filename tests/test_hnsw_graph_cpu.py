"""The hnsw.graph parser that runs in front of every upload (csrc/hnsw_graph.cpp) through its host-only entry point
nidx_gpu_hnsw_graph_check: oracle-written images are accepted with the right entry point / link counts, corrupt images are
refused with NIDX_ERR_INVALID_GRAPH — never read out of bounds (the files come from object storage)."""
import ctypes as C

import numpy as np
import pytest

from nucliadb_amd import _lib


def check(graph, edges, n):
    g = np.ascontiguousarray(graph, dtype=np.uint8)
    e = None if edges is None else np.ascontiguousarray(edges, dtype=np.float32)
    node, layer, links, broken = C.c_uint32(), C.c_uint32(), C.c_uint64(), C.c_uint64()
    rc = _lib.lib().nidx_gpu_hnsw_graph_check(g.ctypes.data if g.size else None, g.size, None if e is None else e.ctypes.data,
                                             0 if e is None else e.size, n, C.byref(node), C.byref(layer), C.byref(links), C.byref(broken))
    return rc, node.value, layer.value, links.value, broken.value


@pytest.fixture(scope="module")
def image(orc):
    rng = np.random.default_rng(3)
    x = rng.standard_normal((600, 16)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    seg = orc.Segment(x, similarity=orc.SIM_DOT)
    g = seg.build_graph(2)
    graph, edges = g.serialize_v2(600)
    return orc, g, graph, edges


def test_oracle_images_are_accepted(image):
    orc, g, graph, edges = image
    rc, node, layer, links, broken = check(graph, edges, 600)
    assert rc == 0 and (node, layer) == orc.disk_v2_entry_point(graph)
    assert links == edges.size and broken == 0          # one weight per link (hnsw/disk/v2.rs:46-49)
    assert check(graph, None, 600)[0] == 0             # hnsw.edges is optional
    assert check(np.zeros(0, np.uint8), None, 0)[0] == 0 and check(np.zeros(0, np.uint8), None, 5)[0] == 0   # empty graph
    # a link into a layer its target does not live on is what fix_broken_graph drops (ram_hnsw.rs:109-143)
    b = orc.Hnsw.new()
    b.add_node(0, 1)
    b.add_node(1, 0)
    b.set_edges(0, 0, [1], [0.5])
    b.set_edges(0, 1, [0], [0.5])
    b.set_edges(1, 0, [1], [0.5])
    b.set_entry_point(0, 1)
    bg, be = b.serialize_v2(2)
    assert check(bg, be, 2) == (0, 0, 1, 3, 1)


def test_malformed_images_are_refused(image):
    orc, g, graph, edges = image
    n = 600
    assert check(graph, edges, n + 1)[0] == _lib.NIDX_ERR_INVALID_GRAPH       # node index shorter than the segment
    assert check(graph[:-4], edges, n)[0] == _lib.NIDX_ERR_INVALID_GRAPH
    assert check(graph[: 4 * n], edges, n)[0] == _lib.NIDX_ERR_INVALID_GRAPH
    assert check(graph, edges[:-1], n)[0] == _lib.NIDX_ERR_INVALID_GRAPH      # fewer weights than links
    assert "hnsw" in _lib.last_error()
    words = graph.view("<u4").copy()
    for pos, value in ((-1, n), (-1, 0xffffffff), (-2, 64), (-3, 0xffffffff), (-3, 0), (0, 61), (0, 0x7fffffff)):
        w = words.copy()
        w[pos] = value                                                        # entry node, entry layer, a node offset, a degree
        assert check(w.view(np.uint8), None, n)[0] == _lib.NIDX_ERR_INVALID_GRAPH, (pos, value)
    w = words.copy()
    w[1] = n                                                                  # an edge target beyond the segment
    assert check(w.view(np.uint8), None, n)[0] == _lib.NIDX_ERR_INVALID_GRAPH


def test_random_corruption_never_crashes(image):
    orc, g, graph, edges = image
    rng = np.random.default_rng(4)
    words = graph.view("<u4")
    refused = 0
    for trial in range(3000):
        w = words.copy()
        for _ in range(int(rng.integers(1, 4))):
            # the trailer (node index + entry point) and the per-node layer offsets are where a wrong word can send a read astray
            pos = int(rng.integers(0, w.size)) if trial % 2 else int(w.size - 1 - rng.integers(0, 700))
            w[pos] = rng.choice([0, 1, 3, 4, 255, 1 << 16, 0x7fffffff, 0xffffffff, int(rng.integers(0, 1 << 32)), int(w[pos]) ^ (1 << int(rng.integers(0, 32)))])
        rc = check(w.view(np.uint8), edges if trial % 3 == 0 else None, 600)[0]
        assert rc in (0, _lib.NIDX_ERR_INVALID_GRAPH)
        refused += rc != 0
    assert refused > 500
    for cut in rng.integers(0, graph.size, 200):                              # truncations
        assert check(graph[: int(cut)], None, 600)[0] in (0, _lib.NIDX_ERR_INVALID_GRAPH)
