"""The bf16 fallback's list-free scan as a plain numpy model (no GPU): the ALGORITHM of `bf16_append_kernel` and of the host code
that drives it (`csrc/vector_bf16.hip`, `vector_index.cpp`: floors from a prefix, a sample of every rs-th round of the full pass's own
grid, slots of 32 per (query, stripe), the full pass that skips the sampled rounds and goes on from the sample's slots, the per-block
fall-back to the list kernel) — checked against the definition of what stage 1 must deliver: for every query the 32 best rows by
approximate score.  What the model pins down is the reasoning the kernel's comments make: a floor that 32 sampled rows reach never
cuts a true candidate, a group that fills up is never silently truncated, a discarded sample is never half-used."""
import numpy as np
import pytest

CAND, TILE, BLOCK = 32, 256, 256


def exact_candidates(score, ok):
    """per query: the CAND best (score desc, row asc) among the rows that take part"""
    out = []
    for q in range(score.shape[0]):
        rows = np.nonzero(ok)[0]
        order = np.lexsort((rows, -score[q, rows]))[:CAND]
        out.append([(float(score[q, rows[i]]), int(rows[i])) for i in order])
    return out


def list_kernel_stripe(score_q, rows, floor):
    """bf16_scan_kernel on one stripe: the CAND best of its rows at or above the floor"""
    rows = rows[score_q[rows] >= floor]
    order = np.lexsort((rows, -score_q[rows]))[:CAND]
    return [(float(score_q[rows[i]]), int(rows[i])) for i in order]


def merge(lists):
    """merge_topk_kernel: the CAND best of the stripes' entries"""
    allc = sorted((e for l in lists for e in l), key=lambda e: (-e[0], e[1]))
    return allc[:CAND]


def run_model(score, ok, stripes, prefix_tiles):
    nq, n = score.shape
    n_tiles = (n + TILE - 1) // TILE
    rounds = (n_tiles + stripes - 1) // stripes
    tile_rows = lambda t: np.arange(t * TILE, min(n, (t + 1) * TILE))
    stripe_round_rows = lambda x, i: (lambda r: r[ok[r]])(tile_rows(x + i * stripes)) if x + i * stripes < n_tiles else np.zeros(0, np.int64)
    # ---- sample A: the list kernel over the prefix ----
    pre = np.arange(0, min(n, prefix_tiles * TILE))
    pre = pre[ok[pre]]
    floor = np.full(nq, -np.inf)
    for q in range(nq):
        best = np.sort(score[q, pre])[::-1]
        if len(best) >= CAND:
            floor[q] = best[CAND - 1]
    floor_rows = prefix_tiles * TILE
    qblocks = (nq + BLOCK - 1) // BLOCK
    slots = [[[] for _ in range(stripes)] for _ in range(nq)]
    stats = {"sample_passes": 0, "sample_overflow_blocks": 0, "full_overflow_blocks": 0, "no_floor_blocks": 0}

    def append_pass(round_list_of, floor_now, keep):
        """bf16_append_kernel: every (query, stripe) appends the rows of its rounds that reach the floor; > CAND -> the block's flag"""
        flags = np.zeros(qblocks, bool)
        for b in range(qblocks):
            qs = range(b * BLOCK, min(nq, (b + 1) * BLOCK))
            if any(floor_now[q] == -np.inf for q in qs):   # a query without a floor: the block is left to the list kernel at once
                flags[b] = True
                stats["no_floor_blocks"] += 1
                continue
            for q in qs:
                for x in range(stripes):
                    group = [e for e in slots[q][x] if e[0] >= floor_now[q]] if keep[b] else []
                    for i in round_list_of(b):
                        rows = stripe_round_rows(x, i)
                        for r in rows[score[q, rows] >= floor_now[q]]:
                            if len(group) < CAND:
                                group.append((float(score[q, r]), int(r)))
                            else:
                                flags[b] = True
                    slots[q][x] = group
        return flags

    last_rs, flags_sample = 0, np.zeros(qblocks, bool)
    while True:
        cap_rows = floor_rows * stripes // 4
        if cap_rows + cap_rows // 4 >= n:
            break
        want = min(cap_rows, max(n * 4 // stripes, floor_rows * 2))
        rs = (n + want - 1) // want
        if rs < 2 or rounds < 2 * rs:
            break
        if ((rounds + rs - 1) // rs) * stripes * TILE > cap_rows:
            break
        for q in range(nq):
            slots[q] = [[] for _ in range(stripes)]
        flags_sample = append_pass(lambda b, rs=rs: range(0, rounds, rs), floor, np.zeros(qblocks, bool))
        stats["sample_passes"] += 1
        stats["sample_overflow_blocks"] += int(flags_sample.sum())
        for q in range(nq):
            if flags_sample[q // BLOCK]:
                continue   # bf16_floor_kernel: a sample that ran out of slots tells nothing
            m = merge(slots[q])
            if len(m) >= CAND:
                floor[q] = max(floor[q], m[CAND - 1][0])
        floor_rows = (rounds // rs) * stripes * TILE
        last_rs = rs
    if not last_rs:
        for q in range(nq):
            slots[q] = [[] for _ in range(stripes)]
    keep = ~flags_sample if last_rs else np.zeros(qblocks, bool)

    def full_rounds(b):
        if last_rs and keep[b]:
            return [i for i in range(rounds) if i % last_rs != 0]
        return range(rounds)

    flags_full = append_pass(full_rounds, floor, keep)
    stats["full_overflow_blocks"] = int(flags_full.sum())
    # ---- bf16_scan_kernel(run_if): the flagged blocks again, with lists, every round ----
    for b in np.nonzero(flags_full)[0]:
        for q in range(b * BLOCK, min(nq, (b + 1) * BLOCK)):
            for x in range(stripes):
                rows = np.concatenate([stripe_round_rows(x, i) for i in range(rounds)] + [np.zeros(0, np.int64)]).astype(np.int64)
                slots[q][x] = list_kernel_stripe(score[q], rows, floor[q])
    return [merge(slots[q]) for q in range(nq)], stats


def scores(rng, nq, n):
    return rng.normal(size=(nq, n)).astype(np.float32)


@pytest.mark.parametrize("case", ["plain", "sampled", "crowded_full_round", "crowded_sampled_round", "filtered_prefix", "ties"])
def test_append_scan_delivers_the_candidates_of_the_definition(case):
    rng = np.random.default_rng(5)
    nq, stripes, prefix_tiles = 3, 16, 8
    # 800 tiles in 50 rounds of 16: the 2 048-row prefix floor carries 8 k rows, so the floor is tightened on every 25th, 7th and 4th round
    # before the full pass (three sample passes: the loop of vector_index.cpp); 40 tiles: one pass, no sample
    n = {"plain": 40 * TILE, "filtered_prefix": 40 * TILE}.get(case, 800 * TILE)
    s = scores(rng, nq, n)
    ok = np.ones(n, bool)
    if case == "crowded_full_round":
        s[0, 212 * TILE + 10: 212 * TILE + 200] += 10.0    # tile 212 = stripe 4, round 13: never sampled
    if case == "crowded_sampled_round":
        s[0, 12 * TILE: 12 * TILE + 200] += 10.0           # tile 12 = stripe 12, round 0 (behind the 8-tile prefix): in every sample
    if case == "filtered_prefix":
        ok[: prefix_tiles * TILE] = False
    if case == "ties":
        s = np.round(s * 2) / 2                            # many equal scores: the floor is reached by more than 32 rows
    got, stats = run_model(s, ok, stripes, prefix_tiles)
    want = exact_candidates(s, ok)
    for q in range(nq):
        assert got[q] == want[q], (case, q, stats)
    if case == "sampled":
        assert stats["sample_passes"] >= 1 and stats["full_overflow_blocks"] == 0
    if case == "crowded_full_round":
        assert stats["sample_passes"] >= 1 and stats["full_overflow_blocks"] == 1
    if case == "crowded_sampled_round":
        assert stats["sample_overflow_blocks"] >= 1 and stats["full_overflow_blocks"] == 1
    if case == "filtered_prefix":
        assert stats["no_floor_blocks"] >= 1
