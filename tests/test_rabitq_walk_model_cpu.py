"""A plain-Python model of the layer-0 walk of the RaBitQ arm as `rabitq_hnsw3_kernel` runs it (csrc/rabitq.hip), checked against the
walk it must equal: `layer_search` on estimates (nidx_vector/src/hnsw/search.rs:242-300) the way `rabitq_hnsw_kernel` (rounds 1-4) and the
oracle run it — pop the best candidate, test-and-set its neighbours in edge order, score the fresh ones, replay
`similarity.score > ws.score || len < k` in edge order.

What the model adds on top of that, exactly as the kernel does:
  * after the estimates of an expansion the NEXT pop is predicted — the best of {first unexpanded key of the result set, best new
    neighbour that will be admitted} — and, when the predicted node's edge record is held (the records of the three best candidates are
    requested one expansion ahead), its neighbours are test-and-set SPECULATIVELY before the admissions of the current expansion;
  * the pop verifies the prediction; a mismatch clears exactly the bits the speculation set and expands the popped node the plain way;
  * a direct-mapped table of ids KNOWN to be visited (entered only when an expansion is no longer speculative) answers most
    neighbours without touching the bitset.
The claims under test: the sequence of expansions, the fresh sets, the final result set and the final visited set are those of the
plain walk for any graph and any scores, exact ties included.  The model also shows WHY the roll-back is a safety net rather than a
path that runs: a speculation needs the predicted node's edge record in registers, only nodes that were candidates one expansion ago
have theirs there, and when the predicted node is such an existing candidate — i.e. it outranks every neighbour about to be admitted
— the admissions cannot put anything in front of it nor evict it.  The prediction only fails for a brand-new neighbour (ties at the
admission threshold), whose record is never held.  (The two-wave kernel's fetcher does fetch records of new nodes; there the
roll-back runs.)  The algorithm, not the instruction stream: the kernel itself is checked bit for bit against the oracle on the GPU,
tests/test_rabitq_gpu.py."""
import numpy as np
import pytest

NONE = 0xFFFFFFFF


def key(score, node):
    """rq_key's order: higher score first, then lower address (csrc/rabitq.hip:264-269)."""
    return (int(score), -int(node))


class ResultSet:
    """The result set with its `unexpanded` flags + the side list of evicted entries that still tie with the worst result
    (csrc/rabitq.hip: RqLayer / rq_admit / rq_pop)."""

    def __init__(self, cap):
        self.cap = cap
        self.items = []   # [key, node, unexpanded], best first
        self.ties = []    # (key, node)

    def worst_score(self):
        return self.items[-1][0][0]

    def full(self):
        return len(self.items) >= self.cap

    def admit(self, score, node):
        k = key(score, node)
        pos = 0
        while pos < len(self.items) and self.items[pos][0] > k:
            pos += 1
        self.items.insert(pos, [k, node, True])
        if len(self.items) > self.cap:
            ev = self.items.pop()
            if ev[2] and not (ev[0][0] < self.worst_score()):   # still a candidate only while it ties with the worst result
                self.ties.append((ev[0], ev[1]))

    def first_unexpanded(self, n=1):
        out = [it for it in self.items if it[2]][:n]
        return [(it[0], it[1]) for it in out]

    def pop(self):
        for it in self.items:
            if it[2]:
                it[2] = False
                return it[1]
        if not self.ties:
            return None
        best = max(self.ties)
        self.ties.remove(best)
        if best[0][0] < self.worst_score():   # `cs < ws => break`
            return None
        return best[1]


def plain_walk(edges, score, ep, ef):
    visited = {ep}
    res = ResultSet(ef)
    res.admit(score[ep], ep)
    trace = []
    while True:
        node = res.pop()
        if node is None:
            break
        fresh = []
        for w in edges[node]:
            if w not in visited:
                visited.add(w)
                fresh.append(w)
        trace.append((node, tuple(fresh)))
        for w in fresh:   # edge order
            if not res.full() or score[w] > res.worst_score():
                res.admit(score[w], w)
    return trace, [(it[0], it[1]) for it in res.items], visited


def speculative_walk(edges, score, ep, ef, seen_log2, stats):
    visited = {ep}
    seen = [NONE] * (1 << seen_log2) if seen_log2 else None

    def slot(v):
        return ((v * 2654435761) & 0xFFFFFFFF) >> (32 - seen_log2)

    res = ResultSet(ef)
    res.admit(score[ep], ep)
    held = []             # nodes whose edge record is in registers (at most three)
    spec = None           # (node, [(w, newly_set)]) of the speculated expansion
    trace = []
    while True:
        node = res.pop()
        if node is None:
            if spec is not None:   # (the kernel leaves the loop: nothing reads the bitset afterwards; the model keeps it comparable)
                for w, newly in spec[1]:
                    if newly:
                        visited.discard(w)
            break
        if spec is not None and spec[0] == node:
            stats["confirmed"] += 1
            fresh = [w for w, newly in spec[1] if newly]
        else:
            if spec is not None:
                stats["rolled_back"] += 1
                for w, newly in spec[1]:
                    if newly:
                        visited.discard(w)   # clear exactly the bits the speculative test-and-set set
            fresh = []
            for w in edges[node]:
                if seen is not None and seen[slot(w)] == w:
                    assert w in visited   # the table never claims an unvisited node
                    stats["answered_by_table"] += 1
                    continue
                stats["asked_memory"] += 1
                if w not in visited:
                    visited.add(w)
                    fresh.append(w)
        spec = None
        if seen is not None:
            for w in edges[node]:
                seen[slot(w)] = w   # every neighbour of an expanded node is visited from here on
        trace.append((node, tuple(fresh)))
        # ---- the next pop, predicted ----
        ws, full = (res.worst_score(), res.full()) if res.items else (None, False)
        new_keys = [key(score[w], w) for w in fresh if (not full or score[w] > ws)]
        best_new = max(new_keys) if new_keys else None
        peek = res.first_unexpanded(3)
        cands = sorted(([(best_new, -best_new[1])] if best_new else []) + peek, reverse=True)[:3]
        want = [c[1] for c in cands]
        if want and want[0] in held:
            p1 = want[0]
            tested = []
            for w in edges[p1]:
                if seen is not None and seen[slot(w)] == w:
                    assert w in visited
                    stats["answered_by_table"] += 1
                    continue
                stats["asked_memory"] += 1
                newly = w not in visited
                if newly:
                    visited.add(w)
                tested.append((w, newly))
            spec = (p1, tested)
        held = [h for h in held if h in want] + [w for w in want if w not in held]
        held = held[:3]
        # ---- admissions of the current expansion, in edge order ----
        for w in fresh:
            if not res.full() or score[w] > res.worst_score():
                res.admit(score[w], w)
    return trace, [(it[0], it[1]) for it in res.items], visited


def random_graph(rng, n, deg):
    edges = []
    for v in range(n):
        near = (v + rng.integers(1, 40, size=deg // 2)) % n          # a neighbourhood, so walks do wander
        far = rng.integers(0, n, size=deg - deg // 2)
        e = [int(x) for x in np.concatenate([near, far]) if x != v]
        edges.append(list(dict.fromkeys(e)))
    return edges


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("levels,ef,seen_log2", [(1 << 20, 64, 5), (8, 64, 5), (3, 40, 0), (3, 200, 4), (1, 30, 3)])
def test_speculative_walk_equals_the_plain_walk(seed, levels, ef, seen_log2):
    """`levels` distinct scores: 2^20 = practically no ties, 8 / 3 / 1 = ties everywhere (evicted ties, side-list pops, neighbours
    rejected at the threshold after the prediction counted on them)."""
    rng = np.random.default_rng(1000 * seed + levels + ef)
    n, deg = 1500, 24
    edges = random_graph(rng, n, deg)
    score = rng.integers(0, levels, size=n)
    for ep in rng.integers(0, n, size=4):
        stats = {"confirmed": 0, "rolled_back": 0, "answered_by_table": 0, "asked_memory": 0}
        want = plain_walk(edges, score, int(ep), ef)
        got = speculative_walk(edges, score, int(ep), ef, seen_log2, stats)
        assert got[0] == want[0], "the expansions (node, fresh neighbours) differ"
        assert got[1] == want[1], "the result sets differ"
        assert got[2] == want[2], "the visited sets differ (a roll-back left a bit behind or cleared one it did not set)"
        assert stats["confirmed"] > 0


def test_the_model_exercises_what_it_claims():
    """Speculations happen and are confirmed, with and without ties; none is ever rolled back (see the module docstring); the table of
    known-visited ids answers a share of the tests."""
    rng = np.random.default_rng(7)
    n, deg = 3000, 30
    edges = random_graph(rng, n, deg)
    for score in (rng.permutation(n), rng.integers(0, 3, size=n)):
        st = {"confirmed": 0, "rolled_back": 0, "answered_by_table": 0, "asked_memory": 0}
        t, _, _ = speculative_walk(edges, score, 5, 100, 9, st)
        # (random scores make a brand-new neighbour the best candidate more often than real estimates do — its record is never held —
        # so fewer expansions are speculated here than the 83 % the kernel measures on the bench corpus)
        assert st["rolled_back"] == 0 and st["confirmed"] > 0.25 * len(t)
        assert st["answered_by_table"] > 0.1 * (st["answered_by_table"] + st["asked_memory"])   # (half random edges: few re-visits; the bench corpus: 55 %)
