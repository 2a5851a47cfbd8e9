"""Stress of the fused merge's cross-workgroup hand-over (bm25_stream.hip: agent-scope atomic accesses, no fences): the bench corpus (10 M documents),
batches of 1 024 three-term queries, the fused launch against the two-launch path bit for bit, several slice lengths (1 .. ~180 slices per query),
repeated.  python scripts/r6_fused_stress.py [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0]] + sys.argv[1:]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
import importlib.util
import numpy as np
spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
b = importlib.util.module_from_spec(spec)
argv, sys.argv = sys.argv, ["bench.py", "--workload", "bm25", "--cpu-queries", "0"]
spec.loader.exec_module(b)
a = b.parse()
sys.argv = argv
import torch
from nucliadb_amd import _lib
L = _lib.lib()
bm = b.Bm25Bench(a, L, torch.device("cuda:0"), 0, a.n_docs)
bad = 0
checked = 0
for r in range(rounds):
    for sl in (None, "512", "4096", "1024"):
        if sl is None:
            os.environ.pop("NIDX_GPU_BM25_SLICE", None)
        else:
            os.environ["NIDX_GPU_BM25_SLICE"] = sl
        for i in range(len(bm.prepared)):
            os.environ.pop("NIDX_GPU_BM25_FUSED_MERGE", None)
            bm.search(i)
            f = (bm.docaddr.copy(), bm.score.copy(), bm.count.copy(), bm.total.copy(), bm.post.copy())
            os.environ["NIDX_GPU_BM25_FUSED_MERGE"] = "0"
            bm.search(i)
            u = (bm.docaddr, bm.score, bm.count, bm.total, bm.post)
            for x, y in zip(f, u):
                if not np.array_equal(np.asarray(x).view(np.uint8), np.asarray(y).view(np.uint8)):
                    bad += 1
            checked += 1
print("fused vs two launches: %d batches of %d queries compared, %d arrays differ" % (checked, bm.B, bad))
sys.exit(1 if bad else 0)
