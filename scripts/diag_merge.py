import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from test_hnsw_build_gpu import *
from nucliadb_amd.vector import segment_merge
from nucliadb_amd import _lib
rng = np.random.default_rng(91)
d = 128
big = np.vstack([clustered(rng, d, 40, 160), random_vector(rng, d, 13600)])
small = np.vstack([random_vector(rng, d, 900), clustered(rng, d, 5, 160)])
cfg = VectorConfig(d, Similarity.Dot)
def seg(x, **kw):
    n = x.shape[0]
    return VectorSegment([f"k-{i}" for i in range(n)], x, [[] for _ in range(n)], [b""] * n, **kw)
s = VectorSearcher.open(cfg, [(seg(big), 1)]); s.build_hnsw(0); graph, edges = s.serialize_hnsw(0); s.close()
allv = np.vstack([big, small])
req = VectorSearchRequest(result_per_page=3, min_score=-1.0, with_duplicates=True)
def selfrate(s, name):
    _, _, vec, score, count = s.search_batch(req, allv, method=_lib.METHOD_HNSW)
    bad = np.nonzero(score[:, 0] < 0.999)[0]
    print(name, "self-miss", len(bad), "of", len(allv), "base-miss", (bad < len(big)).sum(), "new-miss", (bad >= len(big)).sum(), bad[:20])
m = seg(allv, graph=graph, graph_edges=edges, graph_nodes=len(big))
s = VectorSearcher.open(cfg, [(m, 1)]); t0=time.time(); s.extend_hnsw(0); print("extend", time.time()-t0); selfrate(s, "extend"); s.close()
s = VectorSearcher.open(cfg, [(seg(allv), 1)]); t0=time.time(); s.build_hnsw(0); print("full", time.time()-t0); selfrate(s, "full"); s.close()
