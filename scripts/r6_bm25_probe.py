"""Round 6 measurement: where does the BM25 union scorer spend a batch?  The bench batch (3 Should terms from the rank band
[100, 100 k] of a Zipf(1) vocabulary) mixes three regimes — whole short queries, slices of one long list with a few postings of
the short ones, single long lists.  This script times each regime on its own through the library's pipelined entry (six submitting
threads x two tickets, as bench.py) and one launch at a time (nidx_gpu_bm25_last_kernel_ms).

    python scripts/r6_bm25_probe.py [--n-docs 10000000] [regime ...]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--n-docs", type=int, default=10_000_000)
    p.add_argument("--batch", type=int, default=1024)
    p.add_argument("--min-s", type=float, default=1.0)
    p.add_argument("--threads", type=int, default=6)
    p.add_argument("--depth", type=int, default=2)
    p.add_argument("regimes", nargs="*")
    o = p.parse_args()
    sys.argv = [sys.argv[0], "--workload", "bm25", "--n-docs", str(o.n_docs), "--batch", str(o.batch), "--cpu-queries", "0"]
    a = bench.parse()
    a.min_timed_s = o.min_s
    import torch

    from nucliadb_amd import _lib

    L = _lib.lib()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    bm = bench.Bm25Bench(a, L, dev, 0, o.n_docs, n_pool=32)
    term_offsets = bm.corpus[0]
    df = np.diff(term_offsets.astype(np.int64))
    rng = np.random.default_rng(7)
    B = o.batch

    def prepared_from(term_lists):
        """term_lists: list (pool) of list (B) of term-id lists"""
        out = []
        for tl in term_lists:
            n = sum(len(q) for q in tl)
            cl = (_lib.Bm25ClauseC * max(n, 1))()
            offs = np.zeros(len(tl) + 1, np.uint64)
            j = 0
            for i, q in enumerate(tl):
                for t in q:
                    cl[j].term, cl[j].occur, cl[j].mode, cl[j].boost = int(t), 0, 0, 1.0
                    j += 1
                offs[i + 1] = j
            out.append((cl, offs))
        return out

    regimes = {
        "bench": lambda: [[list(rng.integers(99, 100_000, 3)) for _ in range(B)] for _ in range(16)],
        "short3": lambda: [[list(rng.integers(10_000, 100_000, 3)) for _ in range(B)] for _ in range(16)],           # 400 .. 4 k postings per list
        "mid3": lambda: [[list(rng.integers(1_000, 10_000, 3)) for _ in range(B // 4)] for _ in range(16)],          # 4 k .. 40 k
        "long1_short2": lambda: [[[int(rng.integers(99, 1_000))] + list(rng.integers(10_000, 100_000, 2)) for _ in range(B // 16)] for _ in range(16)],
        "long1": lambda: [[[int(rng.integers(99, 1_000))] for _ in range(B // 16)] for _ in range(16)],              # one list of 40 k .. 400 k
        "mid1": lambda: [[[int(rng.integers(1_000, 10_000))] for _ in range(B)] for _ in range(16)],
        "short1": lambda: [[[int(rng.integers(10_000, 100_000))] for _ in range(3 * B)] for _ in range(16)],
    }
    names = o.regimes or list(regimes)
    for name in names:
        tl = regimes[name]()
        prep = prepared_from(tl)
        nq = len(tl[0])
        posts = float(np.mean([sum(int(df[t]) for q in b for t in q) for b in tl]))
        # swap the bench object's batches
        bm.prepared = [c for c, _ in prep]
        bm.offsets = prep[0][1] if False else None
        # every batch of a regime has its own offsets: timed_pipeline reads self.offsets once per submit, so make them uniform in shape
        same_shape = all(np.array_equal(prep[0][1], pr[1]) for pr in prep)
        assert same_shape, name
        bm.offsets = prep[0][1]
        bm.B = nq
        bm.docaddr, bm.score = np.zeros((nq, bm.K), np.uint64), np.zeros((nq, bm.K), np.float32)
        bm.count, bm.total, bm.post = np.zeros(nq, np.uint32), np.zeros(nq, np.uint64), np.zeros(nq, np.uint64)
        for i in range(3):
            bm.search(i)
        k_ms = []
        for i in range(8):
            bm.search(i)
            k_ms.append(bm.kernel_ms())
        elapsed, n_steps, postings, _ = bm.timed_pipeline(bm.searcher, o.threads, o.depth, min_s=o.min_s)
        k = float(np.mean(k_ms))
        print(json.dumps({"regime": name, "queries": nq, "postings_per_batch": posts, "kernel_ms_one_at_a_time": k,
                          "kernel_frac_of_hbm": posts * 8 / (k * 1e-3) / 8e12, "sustained_postings_per_s": postings / elapsed,
                          "sustained_frac_of_hbm": postings / elapsed * 8 / 8e12, "ms_per_batch_sustained": elapsed / n_steps * 1e3}), flush=True)
    bm.close()


if __name__ == "__main__":
    main()
