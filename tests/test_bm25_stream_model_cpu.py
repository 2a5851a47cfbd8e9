"""The ALGORITHM of bm25_stream_kernel (csrc/bm25_stream.hip) as a plain Python model, checked against the oracle on the CPU.

The device tests (tests/test_bm25_gpu.py) check the kernel itself; this file checks the reasoning the kernel rests on, where a
GPU is not needed: that "final unless a bitmap filter says its document may occur twice" classifies every posting correctly
(both partners of a meeting end up in the involved list, whatever the phase order and whichever clause is the longest), that
resolving the involved postings clause by clause reproduces the oracle's f32 sums and boolean tests, that a candidate buffer
merged in bulk with a stale threshold gives the same top-k as one insertion per candidate, and that cutting the doc range in half
after an overflow and offering the already-final postings again is exact when later insertions ignore keys the list holds.

The model follows the kernel's control flow and constants (hashes, bitmap sizes, the 192-entry involved list, 64-posting rows,
four-row groups, the 128-entry candidate buffer drained after every row, the bar taken from the lane maxima of the first group) but
not its instruction stream; per-posting
scores come from the oracle itself (a one-clause query per term), so only the combination logic is under test."""
import numpy as np
import pytest

from nucliadb_amd import _lib
from nucliadb_amd.bm25 import Bm25Segment

S, M, N, G = _lib.OCCUR_SHOULD, _lib.OCCUR_MUST, _lib.OCCUR_MUST_NOT, _lib.OCCUR_SHOULD_GROUP
A_BITS, B_BITS, CAP = 1 << 15, 1 << 11, 192


def zipf_corpus(rng, n_docs, vocab, mean_len=24):
    lens = np.clip(np.round(rng.lognormal(np.log(mean_len), 0.6, n_docs)), 4, 400).astype(np.int64)
    p = 1.0 / np.arange(1, vocab + 1)
    p /= p.sum()
    flat = rng.choice(vocab, size=int(lens.sum()), p=p)
    return np.split(flat, np.cumsum(lens)[:-1])


def rank_key(score, doc):
    b = int(np.float32(score).view(np.int32))
    b ^= ((b >> 31) & 0xFFFFFFFF) >> 1
    return (((b & 0xFFFFFFFF) ^ 0x80000000) << 32) | (~doc & 0xFFFFFFFF)


class Model:
    def __init__(self, clauses, k):
        """clauses: [(docs u32 ascending, scores f32, occur)]"""
        self.cl, self.k = clauses, k
        C = len(clauses)
        occ = [c[2] for c in clauses]
        self.must = sum(1 << i for i in range(C) if occ[i] == M)
        self.nots = sum(1 << i for i in range(C) if occ[i] == N)
        self.should = sum(1 << i for i in range(C) if occ[i] == S)
        self.groups = [m for m in (sum(1 << i for i in range(C) if occ[i] == G + g) for g in range(8)) if m]
        self.single_ok = [self.mask_ok(1 << i) for i in range(C)]
        self.top, self.kth, self.thr = [], 0, 0      # sorted (descending) keys, at most 64
        self.cand, self.redo = [], False
        self.bar = None                              # a lower bound of the item's k-th best score (f32), set by the first group
        self.flushes = self.ranges = 0

    def mask_ok(self, m):
        any_required = self.must != 0 or len(self.groups) > 0
        ok = (m & self.must) == self.must and (m & self.nots) == 0 and (any_required or (m & self.should) != 0)
        return ok and all(m & g for g in self.groups)

    # ---- top-k: bulk merges of 64 buffered candidates with a threshold that is only refreshed at a merge ----
    def flush64(self):
        take, self.cand = self.cand[:64], self.cand[64:]
        self.top = sorted(self.top + take, reverse=True)[:64]
        self.kth = self.top[self.k - 1] if len(self.top) >= self.k else 0
        self.thr = max(self.thr, self.kth)
        self.flushes += 1

    def offer(self, keys):
        keys = [x for x in keys if x > self.thr]
        if not self.redo:
            self.cand += keys
            assert len(self.cand) <= 128
            return
        for x in keys:                       # after a retry: one by one, ignoring keys the list already holds
            if x > self.thr and x not in self.top:
                self.top = sorted(self.top + [x], reverse=True)[:64]
                self.kth = self.top[self.k - 1] if len(self.top) >= self.k else 0
                self.thr = max(self.thr, self.kth)

    def drain(self, everything=False):
        while len(self.cand) >= (1 if everything else 64):
            self.flush64()

    # ---- one work item: the doc range [lo, hi) ----
    def run(self, lo, hi):
        C = len(self.cl)
        total = 0
        pos = [int(np.searchsorted(c[0], lo)) for c in self.cl]
        item_end = [int(np.searchsorted(c[0], hi)) for c in self.cl]
        cur_lo, cur_hi, end = lo, hi, list(item_end)
        while True:
            n = [end[c] - pos[c] for c in range(C)]
            active = [c for c in range(C) if n[c] > 0]
            if active:
                L = max(active, key=lambda c: (n[c], -c))
                probe = len(active) > 1
                A, B = np.zeros(A_BITS // 32, np.uint32), np.zeros(B_BITS, bool)
                h_of = lambda d: (int(d) ^ (int(d) >> 15)) & 0x7FFF

                def a_mask(d):   # bs_a_mask: three bits of the document's word, from the top of a multiplicative hash
                    g = (int(d) * 0x9E3779B1) & 0xFFFFFFFF
                    return (1 << (g >> 27)) | (1 << ((g >> 22) & 31)) | (1 << ((g >> 17) & 31))

                in_a = lambda d: (int(A[h_of(d) >> 5]) & a_mask(d)) == a_mask(d)
                if probe:                                             # phase 1: mark
                    for c in active:
                        if c == L:
                            continue
                        for d in self.cl[c][0][pos[c]:end[c]]:
                            h = h_of(d)
                            if in_a(d):
                                B[h & 0x7FF] = True
                            A[h >> 5] |= np.uint32(a_mask(d))
                inv_short, inv_long, matched, overflow = [], [], 0, False
                for step, c in enumerate([L] + [c for c in active if c != L and probe]):   # phases 2 and 3
                    is_long = step == 0
                    docs, scores = self.cl[c][0][pos[c]:end[c]], self.cl[c][1][pos[c]:end[c]]
                    for g0 in range(0, len(docs), 256):
                        if self.bar is None and self.single_ok[c]:
                            # the bar before the list has one: the k-th largest of the 64 lane maxima of the first group's final postings
                            lane_max = {}
                            for j in range(g0, min(g0 + 256, len(docs))):
                                d = docs[j]
                                if not (probe and bool(in_a(d) if is_long else B[h_of(d) & 0x7FF])):
                                    lane_max[(j - g0) % 64] = max(lane_max.get((j - g0) % 64, np.float32(-np.inf)), np.float32(scores[j]))
                            best = sorted(lane_max.values(), reverse=True)
                            self.bar = best[self.k - 1] if len(best) >= self.k else np.float32(-np.inf)
                        for r in range(4):
                            row = slice(g0 + 64 * r, min(g0 + 64 * r + 64, len(docs)))
                            if row.start >= len(docs):
                                break
                            inv = [probe and bool(in_a(d) if is_long else B[h_of(d) & 0x7FF]) for d in docs[row]]
                            if sum(inv) and len(inv_short) + len(inv_long) + sum(inv) > CAP:
                                overflow = True
                                break
                            for d, s, i in zip(docs[row], scores[row], inv):
                                if i:
                                    if is_long:
                                        B[h_of(d) & 0x7FF] = True
                                    (inv_long if is_long else inv_short).append((int(d), np.float32(s), c))
                            if self.single_ok[c]:
                                singles = [(d, s) for d, s, i in zip(docs[row], scores[row], inv) if not i]
                                matched += len(singles)
                                self.offer([rank_key(s, int(d)) for d, s in singles if not (np.float32(s) < self.bar)])
                            if not self.redo:
                                self.drain()
                        if overflow:
                            break
                        if not self.redo:
                            self.drain()
                    if overflow:
                        break
                self.ranges += 1
                if overflow:    # halve the doc range and retry it; what was offered stays offered
                    self.drain(everything=True)
                    self.redo = True
                    cur_hi = cur_lo + max((cur_hi - cur_lo) // 2, 1)
                    end = [int(np.searchsorted(c[0][:end[i]], cur_hi)) for i, c in enumerate(self.cl)]
                    continue
                # phase 4: the involved postings in clause order (L's block between the short clauses in front of and behind it)
                involved = sorted(inv_short + inv_long, key=lambda e: e[2])
                by_doc = {}
                for d, s, c in involved:
                    acc, mask = by_doc.get(d, (np.float32(0.0), 0))
                    if self.cl[c][2] != N:
                        acc = np.float32(acc + s)
                    by_doc[d] = (acc, mask | (1 << c))
                keys = [rank_key(acc, d) for d, (acc, mask) in by_doc.items() if self.mask_ok(mask)]
                matched += len(keys)
                self.offer(keys)
                if not self.redo:
                    self.drain()
                total += matched
            if cur_hi >= hi:
                break
            cur_lo, cur_hi, pos, end = cur_hi, hi, list(end), list(item_end)
        self.drain(everything=True)
        return total


def model_search(oidx, seg, query, k, n_slices):
    """query: [(term, occur, mode, boost)] -> (docs, scores, total) through `n_slices` work items merged like bm25_merge_kernel"""
    n_docs = len(seg.fieldnorm_ids)
    clauses = []
    for t, o, m, b in query:
        d, s, _ = oidx.search([(t, S, m, b)], n_docs)      # the clause's own postings with the oracle's per-posting scores
        order = np.argsort(d, kind="stable")
        clauses.append((d[order].astype(np.uint32), s[order].astype(np.float32), o))
    keys, total, stats = [], 0, [0, 0]
    for sl in range(n_slices):
        m = Model(clauses, k)
        total += m.run(n_docs * sl // n_slices, n_docs * (sl + 1) // n_slices)
        keys += m.top[:k]
        stats[0] += m.ranges
        stats[1] += m.flushes
    keys = sorted(keys, reverse=True)[:k]
    docs = [(~x) & 0xFFFFFFFF for x in keys]
    sc = []
    for x in keys:
        b = ((x >> 32) ^ 0x80000000) & 0xFFFFFFFF
        b = b - (1 << 32) if b & 0x80000000 else b
        b ^= ((b >> 31) & 0xFFFFFFFF) >> 1
        sc.append(np.array([b & 0xFFFFFFFF], np.uint32).view(np.float32)[0])
    return np.array(docs, np.uint64), np.array(sc, np.float32), total, stats


@pytest.fixture(scope="module")
def index(orc):
    rng = np.random.default_rng(77)
    vocab = 600
    seg = Bm25Segment.from_term_docs(zipf_corpus(rng, 6000, vocab), vocab)
    return seg, orc.Bm25Index(seg.term_offsets, seg.doc_ids, seg.tfs, seg.fieldnorm_ids, seg.total_num_tokens, seg.alive), vocab


def check(index, query, k, n_slices):
    seg, oidx, _ = index
    wd, ws, wt = oidx.search(query, k)
    d, s, t, stats = model_search(oidx, seg, query, k, n_slices)
    assert t == wt, (query, t, wt)
    assert np.array_equal(d, wd) and np.array_equal(s.view(np.uint32), ws.view(np.uint32)), (query, d, wd)
    return stats


def test_the_stream_algorithm_equals_the_oracle_on_unions_and_boolean_mixes(index):
    _, _, vocab = index
    rng = np.random.default_rng(5)
    FREQ, BASIC, CONST = _lib.TF_FREQ, _lib.TF_BASIC, _lib.CONST_SCORE
    queries = [[(int(t), S, FREQ, 1.0) for t in rng.integers(20, vocab, int(rng.integers(1, 9)))] for _ in range(24)]
    for _ in range(24):   # Must / MustNot / required Should groups, constant scores, boosts
        queries.append([(int(rng.integers(0, 200)), int(rng.choice([S, S, M, N, G, G + 1])), int(rng.choice([FREQ, BASIC, CONST])),
                         float(rng.choice([1.0, 0.5, 2.0]))) for _ in range(int(rng.integers(1, 7)))])
    for q in queries:
        for k, n_slices in ((20, 1), (5, 3)):
            check(index, q, k, n_slices)


def test_overflowing_involved_lists_are_retried_exactly(index):
    """Dense and repeated terms: every posting of the shorter lists meets the longest one, the 192-entry list overflows, the doc range is
    halved again and again and what was offered before is offered again."""
    FREQ = _lib.TF_FREQ
    retried = 0
    for q in ([(0, S, FREQ, 1.0), (1, S, FREQ, 1.0), (2, S, FREQ, 1.0)], [(3, S, FREQ, 1.0), (3, S, FREQ, 2.0)],
              [(0, M, FREQ, 1.0), (1, M, FREQ, 1.0), (5, S, FREQ, 1.0)], [(1, N, FREQ, 1.0), (0, S, FREQ, 1.0), (7, S, FREQ, 1.0)],
              [(0, G, FREQ, 1.0), (1, G, FREQ, 1.0), (2, G + 1, FREQ, 1.0), (3, G + 1, FREQ, 1.0), (4, M, _lib.TF_BASIC, 1.0)]):
        for k in (1, 20, 64):
            ranges, _ = check(index, q, k, 1)
            retried += ranges > 1
    assert retried >= 10    # (the slow path was the common one here)
