#!/usr/bin/env python3
"""Writes tests/golden/*.npz: small seeded input/output vectors of the hot path, produced by the CPU
oracle (oracle/nidx_oracle.c).  The reference itself is Rust and cannot run here, so these
fixtures pin the ORACLE's outputs: `-m "not gpu"` tests re-derive them with the oracle (drift guard),
`-m gpu` tests require the HIP path to reproduce them bit for bit.

    python scripts/make_golden.py        # regenerate (only when the oracle is deliberately changed)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def unit(rng, n, d):
    x = rng.uniform(-1, 1, (n, d)).astype(np.float32)
    return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)


def main():
    orc.build()
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(1234567890)

    # 1. similarity bits (a1), both summation orders of the kernels
    d = 300
    x, y = rng.normal(size=(48, d)).astype(np.float32), rng.normal(size=(48, d)).astype(np.float32)
    x[0] = 0
    y[1] = 0
    out = {}
    for name, order in (("wave64", orc.ORDER_WAVE64), ("serial_fma", orc.ORDER_SERIAL_FMA)):
        for sname, sim in (("dot", 0), ("cosine", 1)):
            out[f"{sname}_{name}"] = np.array([orc.similarity(x[i], y[i], sim, order) for i in range(len(x))], np.float32)
    np.savez_compressed(os.path.join(OUT, "similarity.npz"), x=x, y=y, **out)

    # 2. brute force + HNSW search over one graph image (a3-a7, a13)
    n, d, k = 1500, 40, 10
    xs = unit(rng, n, d)
    xs[20:24] = xs[19]
    q = np.vstack([xs[19][None, :], unit(rng, 15, d)])
    seg = orc.Segment(xs, similarity=orc.SIM_COSINE)
    graph, edges = seg.build_graph(seed=2).serialize_v2(n)
    filt = orc.bitset(n, ones=np.nonzero(rng.random(n) < 0.3)[0].tolist())
    res = {"vectors": xs, "queries": q, "graph": np.frombuffer(bytes(graph), np.uint8), "filter": filt, "k": np.array(k)}

    def pack(fn, **kw):
        vv, ss, cc = np.zeros((len(q), k), np.uint32), np.zeros((len(q), k), np.float32), np.zeros(len(q), np.uint32)
        for i in range(len(q)):
            v, s = fn(q[i], k, **kw)
            vv[i, : len(v)], ss[i, : len(v)], cc[i] = v, s, len(v)
        return vv, ss, cc

    for name, fn, kw in (("bf", seg.brute_force, {}), ("bf_filter", seg.brute_force, {"filter_bits": filt, "min_score": 0.05}),
                         ("hnsw_dup", seg.hnsw_search, {"with_duplicates": True}), ("hnsw_nodup", seg.hnsw_search, {"with_duplicates": False}),
                         ("hnsw_filter", seg.hnsw_search, {"filter_bits": filt, "min_score": 0.05})):
        res[name + "_vec"], res[name + "_score"], res[name + "_count"] = pack(fn, **kw)
    seg_mfma = orc.Segment(xs, similarity=orc.SIM_COSINE, order=orc.ORDER_SERIAL_FMA)
    res["mfma_vec"], res["mfma_score"], res["mfma_count"] = pack(seg_mfma.brute_force)
    np.savez_compressed(os.path.join(OUT, "vector_search.npz"), **res)

    # 3. BM25 (a15/a16)
    vocab, n_docs = 300, 4000
    lens = np.clip(np.round(rng.lognormal(np.log(12), 0.6, n_docs)), 2, 200).astype(np.int64)
    p = 1.0 / np.arange(1, vocab + 1)
    p /= p.sum()
    flat = rng.choice(vocab, size=int(lens.sum()), p=p)
    doc_of = np.repeat(np.arange(n_docs), lens)
    uniq, counts = np.unique(flat.astype(np.int64) * (n_docs + 1) + doc_of, return_counts=True)
    t, dd = uniq // (n_docs + 1), uniq % (n_docs + 1)
    term_offsets = np.zeros(vocab + 1, np.uint64)
    np.add.at(term_offsets, t + 1, 1)
    term_offsets = np.cumsum(term_offsets).astype(np.uint64)
    table = orc.fieldnorm_table().astype(np.int64)
    ids = (np.searchsorted(table, lens, side="right") - 1).astype(np.uint8)
    alive = orc.bitset(n_docs, ones=np.nonzero(rng.random(n_docs) < 0.9)[0].tolist())
    idx = orc.Bm25Index(term_offsets, dd.astype(np.uint32), counts.astype(np.uint32), ids, int(lens.sum()), alive)
    queries = []
    for _ in range(24):
        queries.append([(int(rng.integers(0, 60)), int(rng.choice([0, 0, 1, 2])), int(rng.choice([0, 1, 2])), float(rng.choice([1.0, 0.5])))
                        for _ in range(int(rng.integers(1, 5)))])
    kk = 20
    qa = np.full((len(queries), 4, 4), -1.0, np.float64)
    da, sa, ca, ta = np.zeros((len(queries), kk), np.uint64), np.zeros((len(queries), kk), np.float32), np.zeros(len(queries), np.uint32), np.zeros(len(queries), np.uint64)
    for i, qq in enumerate(queries):
        for j, c in enumerate(qq):
            qa[i, j] = c
        dv, sv, tot = idx.search(qq, kk)
        da[i, : len(dv)], sa[i, : len(dv)], ca[i], ta[i] = dv, sv, len(dv), tot
    np.savez_compressed(os.path.join(OUT, "bm25.npz"), term_offsets=term_offsets, doc_ids=dd.astype(np.uint32), tfs=counts.astype(np.uint32),
                        fieldnorm_ids=ids, total_num_tokens=np.array(int(lens.sum())), alive=alive, queries=qa, docaddr=da, score=sa, count=ca,
                        total=ta, k=np.array(kk))
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
