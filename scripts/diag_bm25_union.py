"""Debug aid: the same queries through bm25_union_kernel (NIDX_GPU_BM25_UNION=2) and the hash kernels (=0); prints the first
differences with the per-clause membership of the documents involved."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from nucliadb_amd.bm25 import Bm25Searcher, Bm25Segment, Clause  # noqa: E402
from test_bm25_gpu import zipf_corpus  # noqa: E402

rng = np.random.default_rng(1234567890)
vocab = 5000
docs = zipf_corpus(rng, 60000, vocab)
seg = Bm25Segment.from_term_docs(docs, vocab)
rng = np.random.default_rng(1)
rng = np.random.default_rng(int(os.environ.get("DIAG_SEED", "21")))
queries = [[Clause(int(t)) for t in rng.integers(int(os.environ.get("DIAG_LO", "200")), vocab, int(rng.integers(1, 9)))] for _ in range(48)]
if os.environ.get("DIAG_N"):
    queries = queries[: int(os.environ["DIAG_N"])]
if os.environ.get("DIAG_ONE"):
    queries = [queries[int(os.environ["DIAG_ONE"])]]
s = Bm25Searcher.open([seg])
k = int(os.environ.get("DIAG_K", "64"))
os.environ["NIDX_GPU_BM25_UNION"] = "0"
d0, s0, c0, t0, p0 = s.search_batch(queries, k)
os.environ["NIDX_GPU_BM25_UNION"] = "2"
d1, s1, c1, t1, p1 = s.search_batch(queries, k)
nbad = 0
for i, q in enumerate(queries):
    same = c0[i] == c1[i] and np.array_equal(d0[i, : c0[i]], d1[i, : c1[i]]) and np.array_equal(s0[i, : c0[i]].view(np.uint32), s1[i, : c1[i]].view(np.uint32))
    if same and t0[i] == t1[i] and p0[i] == p1[i]:
        continue
    nbad += 1
    if nbad > 3:
        continue
    lists = {j: seg.doc_ids[seg.term_offsets[c.term]: seg.term_offsets[c.term + 1]] for j, c in enumerate(q)}
    print("query", i, "terms", [(c.term, len(lists[j])) for j, c in enumerate(q)], "total", t0[i], t1[i], "postings", p0[i], p1[i], "count", c0[i], c1[i])
    exp = {int(d): float(sc) for d, sc in zip(d0[i, : c0[i]], s0[i, : c0[i]])}
    got = {int(d): float(sc) for d, sc in zip(d1[i, : c1[i]], s1[i, : c1[i]])}
    for d in sorted(set(exp) | set(got)):
        if exp.get(d) != got.get(d):
            member = [int(np.searchsorted(l, d) < len(l) and l[np.searchsorted(l, d)] == d) for l in lists.values()]
            where = [int(np.searchsorted(l, d)) for l in lists.values()]
            print("   doc", d, "expected", exp.get(d), "got", got.get(d), "in clauses", member, "at positions", where)
print("queries that differ:", nbad, "of", len(queries))
