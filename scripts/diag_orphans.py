"""Why does the flat clustered graph stop below the reference regime's recall however wide the search is?

Builds the bench's clustered corpus at --n vectors on the device, serialises the graph, computes every node's layer-0
in-degree on the host (numpy over the DiskHnswV2 image), runs the ladder (ef_upper, ef_search) and reports, for the true
neighbours that are still missed at the widest setting, their in-degree and whether they sit in the tight (radius 0.01) or
the loose (0.03) half of their cluster.  Usage (GPU box, repo root): python scripts/diag_orphans.py --n 2000000"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from nucliadb_amd import _lib  # noqa: E402


def layer0_in_degree(graph_u8, n):
    """Layer-0 in- and out-degree of every node, through the oracle's DiskHnswV2 reader (hnsw/disk/v2.rs:159-174)."""
    from oracle import oracle as orc

    indeg = np.zeros(n, np.int64)
    outdeg = np.zeros(n, np.int64)
    g = orc.Hnsw.deserialize_v2(graph_u8)
    for i in range(n):
        e, _ = g.edges(0, i)
        outdeg[i] = e.size
        np.add.at(indeg, e.astype(np.int64), 1)
    return indeg, outdeg


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--n", type=int, default=1_000_000)
    p.add_argument("--dim", type=int, default=768)
    p.add_argument("--queries", type=int, default=512)
    p.add_argument("--build-ef-upper", type=int, default=0, help="tunable build_ef_upper (0 = the reference's greedy descent while inserting)")
    p.add_argument("--degrees", type=int, default=1, help="0 = skip the host-side in-degree pass")
    a = p.parse_args()
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    n, d, k, B = a.n, a.dim, 10, a.queries
    x = bench.gen_corpus("clustered", n, d, dev, 1234567890)
    # the recipe point of every row (bench.gen_corpus: row r is point perm[r]; the same generator state reproduces perm)
    g = torch.Generator(device=dev)
    g.manual_seed(1234567890)
    perm = torch.randperm(n, generator=g, device=dev).cpu().numpy()
    loose = (perm % bench.PER_CLUSTER) >= bench.PER_CLUSTER // 2
    q = bench.gen_queries("clustered", x, 1, B, d, dev, 2)[0].contiguous()
    cfg = _lib.VectorConfigC(d, 1, 0, 0)
    cseg = _lib.VectorSegmentC(x.data_ptr(), d * 4, n, None, n, None, 0, 0, None, 0, None, None)
    h = C.c_void_p()
    _lib.check(L.nidx_gpu_vector_open(C.byref(cfg), C.byref(cseg), 1, C.byref(h)))
    del x
    if a.build_ef_upper:
        _lib.check(L.nidx_gpu_vector_set_tunable(h, b"build_ef_upper", a.build_ef_upper))
    import time
    t0 = time.time()
    _lib.check(L.nidx_gpu_vector_build_hnsw(h, 0, 2))
    build_s = time.time() - t0
    if a.degrees:
        graph, _edges = bench.serialize_graph(L, h)
        indeg, outdeg = layer0_in_degree(graph, n)
    else:
        indeg, outdeg = np.full(n, 10, np.int64), np.zeros(n, np.int64)
    out = {"n": n, "build_ef_upper": a.build_ef_upper, "build_s": build_s, "orphans_layer0": int((indeg == 0).sum()), "in_degree_le_2": int((indeg <= 2).sum()),
           "orphans_loose": int(((indeg == 0) & loose).sum()), "orphans_tight": int(((indeg == 0) & ~loose).sum()),
           "mean_in_degree_loose": float(indeg[loose].mean()), "mean_in_degree_tight": float(indeg[~loose].mean()),
           "mean_out_degree": float(outdeg.mean())}
    ov = torch.zeros((B, k), dtype=torch.int32, device=dev)
    osc = torch.zeros((B, k), dtype=torch.float32, device=dev)
    oc = torch.zeros((B,), dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def search(method):
        pr = _lib.VectorSearchParamsC(k, -1.0, 1, method)
        _lib.check(L.nidx_gpu_vector_segment_search_device(h, 0, q.data_ptr(), B, C.byref(pr), None, ov.data_ptr(), osc.data_ptr(), oc.data_ptr(),
                                                           None, stream))
        torch.cuda.synchronize()
        return ov.cpu().numpy().astype(np.int64).copy()

    exact = search(_lib.METHOD_BRUTE_FORCE)
    out["ladder"] = []
    missed = None
    for efu, ef in ((1, 30), (4, 30), (4, 64), (4, 256), (16, 512)):
        _lib.check(L.nidx_gpu_vector_set_tunable(h, b"ef_upper", 0 if efu == 1 else efu))
        _lib.check(L.nidx_gpu_vector_set_tunable(h, b"ef_search", ef))
        got = search(_lib.METHOD_HNSW)
        hits = [len(set(got[i].tolist()) & set(exact[i].tolist())) for i in range(B)]
        out["ladder"].append({"ef_upper": efu, "ef_search": ef, "recall": float(np.mean(hits)) / k,
                              "queries_by_hits": np.bincount(hits, minlength=k + 1).tolist()})
        missed = np.asarray(sorted({int(v) for i in range(B) for v in set(exact[i].tolist()) - set(got[i].tolist())}), np.int64)
    out["missed_at_widest"] = {"n": int(missed.size), "in_degree_hist": np.bincount(np.minimum(indeg[missed], 10), minlength=11).tolist() if missed.size else [],
                               "loose": int(loose[missed].sum()) if missed.size else 0}
    print(json.dumps(out))
    L.nidx_gpu_vector_close(h)


if __name__ == "__main__":
    main()
