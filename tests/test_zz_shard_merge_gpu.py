"""merge_vector_responses on the device (csrc/shard_merge_device.hip, what bench.py runs after the RCCL all-gather at N > 1)
against the oracle's k-way merge (shard_merge.rs:332-348), tie order included, for 1 to 8 shard lists.  The same body runs on
CPU tensors (the library's host merge) in the "not gpu" suite, so the expectation itself is checked without a device."""
import numpy as np
import pytest
import torch


def _lists(P, B, k, seed):
    rng = np.random.default_rng(seed)
    score = np.zeros((P, B, k), np.float32)
    ident = np.zeros((P, B, k), np.int64)
    count = rng.integers(0, k + 1, (P, B)).astype(np.int32)
    if B > 1:
        count[:, 0] = 0                               # a query nobody has a hit for
        count[:, 1] = k                               # every list full
    for p in range(P):
        for q in range(B):
            c = int(count[p, q])
            # coarse scores: many ties across shards — the order kmerge's heap gives them is part of the contract
            score[p, q, :c] = np.sort(rng.integers(0, 6, c).astype(np.float32) / 4)[::-1]
            ident[p, q, :c] = (p << 32) | np.arange(c)
    return score, ident, count


def _check(orc, device):
    import __graft_entry__ as g
    from nucliadb_amd.shard_merge import merge_vector_lists

    g.build()
    for P, B, k, limit in ((1, 5, 10, 10), (2, 257, 10, 10), (4, 64, 10, 25), (8, 130, 10, 10), (8, 33, 20, 7), (3, 1, 4, 50)):
        score, ident, count = _lists(P, B, k, 1000 + P)
        ms, mi, mc = merge_vector_lists(torch.from_numpy(score).to(device), torch.from_numpy(ident).to(device),
                                        torch.from_numpy(count).to(device), limit)
        ms, mi, mc = ms.cpu().numpy(), mi.cpu().numpy(), mc.cpu().numpy()
        assert ms.shape == (B, limit) and mc.shape == (B,)
        for q in range(B):
            lists = [[(float(score[p, q, i]), int(ident[p, q, i])) for i in range(count[p, q])] for p in range(P)]
            want = orc.merge_vector(lists, limit)
            got = [(float(ms[q, i]), int(mi[q, i])) for i in range(mc[q])]
            assert got == want, (P, B, k, limit, q)
        if B > 1:
            assert mc[0] == 0 and mc[1] == min(limit, P * k)


def test_host_merge_matches_oracle(orc):
    _check(orc, torch.device("cpu"))


@pytest.mark.gpu
def test_device_merge_matches_oracle(orc):
    _check(orc, torch.device("cuda", 0))


# ---- BM25 merges: sort_documents_fn / sort_paragraphs_fn (shard_merge.rs:211-250,289-329) for a whole batch -----------------------
SHARD_IDS = [b"shard-b", b"shard-a", b"shard", b"shard-c", b"", b"t", b"shard-b", b"zz"]   # a prefix, an empty id, two equal ids


def _bm25_lists(P, B, k, seed):
    rng = np.random.default_rng(seed)
    score = np.zeros((P, B, k), np.float32)
    addr = np.zeros((P, B, k), np.int64)
    value = np.zeros((P, B, k), np.int64)
    count = rng.integers(0, k + 1, (P, B)).astype(np.int32)
    if B > 1:
        count[:, 0] = 0
        count[:, 1] = k
    for p in range(P):
        for q in range(B):
            c = int(count[p, q])
            score[p, q, :c] = np.sort(rng.integers(0, 5, c).astype(np.float32) / 2)[::-1]
            addr[p, q, :c] = np.sort(rng.integers(0, 40, c))            # same docaddr in several shards: the shard id decides
            value[p, q, :c] = np.sort(rng.integers(-3, 4, c))[::-1]     # dates, descending inside a shard
    return score, addr, value, count


def _check_bm25(orc, device):
    import __graft_entry__ as g
    from nucliadb_amd import _lib
    from nucliadb_amd.shard_merge import merge_bm25_lists

    g.build()
    for P, B, k, limit in ((1, 4, 10, 10), (2, 130, 10, 10), (4, 64, 10, 25), (8, 70, 20, 7), (3, 1, 4, 50)):
        score, addr, value, count = _bm25_lists(P, B, k, 2000 + P)
        ids = SHARD_IDS[:P]
        t = lambda a: torch.from_numpy(a).to(device)
        ms, ma, ml, mc = [x.cpu().numpy() for x in merge_bm25_lists(t(score), t(addr), t(count), ids, limit)]
        for q in range(B):
            lists = [[(float(score[p, q, i]), int(addr[p, q, i]), ids[p], p) for i in range(count[p, q])] for p in range(P)]
            want = orc.merge_bm25(lists, limit)
            got = [(float(ms[q, i]), int(ma[q, i]), ids[int(ml[q, i])]) for i in range(mc[q])]
            assert got == [(w[0], w[1], w[2]) for w in want], (P, B, k, limit, q)
        # SortExpr::Date: strictly greater (descending) / smaller (ascending) value first, ties in kmerge's heap order —
        # the expectation is the host per-query merge driven with the value as the only key (score = value, one shard id)
        for order, sign in ((_lib.MERGE_ORDER_VALUE_DESC, 1), (_lib.MERGE_ORDER_VALUE_ASC, -1)):
            v = value if sign == 1 else -value   # lists must arrive in the requested order
            vs, va, vl, vc, vv = [x.cpu().numpy() for x in merge_bm25_lists(t(score), t(addr), t(count), ids, limit, g_value=t(np.ascontiguousarray(v)), order=order)]
            for q in range(B):
                # kmerge with first(a, b) = a.value > b.value (desc) == merge_vector's `>=` only without ties; build the expectation
                # by a direct restatement of itertools' kmerge on the keys
                heads = [[int(v[p, q, i]) for i in range(count[p, q])] for p in range(P)]
                want = _kmerge_strict(heads, limit, desc=(sign == 1))
                got = [(int(vl[q, i]), int(vv[q, i])) for i in range(vc[q])]
                assert got == want, (order, P, B, q, got, want)


def _kmerge_strict(lists, limit, desc):
    """itertools::kmerge_by with first(a, b) = a > b (desc) / a < b (asc): heap of (list, pos) heads, sift_down after every pop."""
    heap = [[l, 0] for l in range(len(lists)) if lists[l]]
    val = lambda h: lists[h[0]][h[1]]
    first = (lambda a, b: val(a) > val(b)) if desc else (lambda a, b: val(a) < val(b))

    def sift_down(n, index):
        pos, child = index, 2 * index + 1
        while child + 1 < n:
            if first(heap[child + 1], heap[child]):
                child += 1
            if not first(heap[child], heap[pos]):
                return
            heap[pos], heap[child] = heap[child], heap[pos]
            pos, child = child, 2 * child + 1
        if child + 1 == n and first(heap[child], heap[pos]):
            heap[pos], heap[child] = heap[child], heap[pos]

    n = len(heap)
    for i in range(n // 2 - 1, -1, -1):
        sift_down(n, i)
    out = []
    while n > 0 and len(out) < limit:
        h = heap[0]
        out.append((h[0], val(h)))
        if h[1] + 1 < len(lists[h[0]]):
            h[1] += 1
        else:
            heap[0] = heap[n - 1]
            n -= 1
        sift_down(n, 0)
    return out


def test_host_bm25_batch_merge_matches_oracle(orc):
    _check_bm25(orc, torch.device("cpu"))


@pytest.mark.gpu
def test_device_bm25_merge_matches_oracle(orc):
    _check_bm25(orc, torch.device("cuda", 0))


def test_merge_facets_sums_per_group_and_tag():
    """merge_facets (shard_merge.rs:380-414): totals of equal (group, tag) summed across shards; the reference's own case
    (shard_merge.rs tests: two shards sharing one tag) plus disjoint tags, an empty shard, equal tags under different groups."""
    import __graft_entry__ as g
    from nucliadb_amd.shard_merge import merge_facets

    g.build()
    shards = [
        [(b"/l", b"/l/a", 2), (b"/l", b"/l/b", 1), (b"/t", b"/t/x", 7)],
        [],
        [(b"/l", b"/l/a", 3), (b"/n", b"/l/a", 5), (b"/t", b"/t/x", 1), (b"/t", b"/t/y", 4)],
    ]
    want = {}
    for s in shards:
        for grp, tag, total in s:
            want[(grp, tag)] = want.get((grp, tag), 0) + total
    got = merge_facets(shards)
    assert got == [(grp, tag, want[(grp, tag)]) for grp, tag in sorted(want)]
    assert merge_facets([]) == [] and merge_facets([[], []]) == []


@pytest.mark.gpu
def test_rccl_exchange_world_of_one(orc):
    """The product's own RCCL path (csrc/shard_comm.cpp) on the one GPU a test box has: communicator of size 1, the packed
    all-gather and the merge kernels on the gathered block.  A world of one must return every list cut to `limit`."""
    import __graft_entry__ as g
    from nucliadb_amd import _lib
    from nucliadb_amd.shard_merge import ShardComm

    g.build()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    _lib.check(_lib.lib().nidx_gpu_set_device(0))
    comm = ShardComm(ShardComm.unique_id(), 0, 1, b"shard-0")
    try:
        for B, k, limit in ((5, 10, 10), (257, 10, 4), (64, 7, 20)):
            score, ident, count = _lists(1, B, k, 77)
            t = lambda a: torch.from_numpy(a[0]).to(dev)
            ms, mi, mc = [x.cpu().numpy() for x in comm.exchange_merge_vector(t(score), t(ident), t(count), limit)]
            for q in range(B):
                n = min(int(count[0, q]), limit)
                assert mc[q] == n and np.array_equal(ms[q, :n], score[0, q, :n]) and np.array_equal(mi[q, :n], ident[0, q, :n])
            bs, ba, bv, bc = _bm25_lists(1, B, k, 78)
            os_, oa, orank, oc, ov = [x.cpu().numpy() for x in comm.exchange_merge_bm25(t(bs), t(ba), t(bc), limit, value=t(bv),
                                                                                        order=_lib.MERGE_ORDER_VALUE_DESC)]
            for q in range(B):
                n = min(int(bc[0, q]), limit)
                assert oc[q] == n and np.array_equal(ov[q, :n], bv[0, q, :n]) and np.array_equal(oa[q, :n], ba[0, q, :n]) and not orank[q, :n].any()
    finally:
        comm.close()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_library_exchange_between_processes(orc, world, tmp_path):
    """nidx_gpu_shard_exchange_merge_{vector,bm25} with a communicator of more than one rank: `world` PROCESSES on the one GPU of the
    box, the library's own exchange over its shared-memory transport (csrc/shard_comm.cpp: the same pack, in-place gather offsets,
    shard order and merge kernels as over RCCL — only ncclAllGather itself is replaced).  Every rank must come back with the
    merge of all ranks' lists: the oracle's merge_vector / merge_bm25 (shard_merge.rs:211-348) incl. tie order and shard-id order,
    the date orders against kmerge restated; ranks that bring different block shapes get an error, not a hang."""
    import os
    import subprocess
    import sys

    import __graft_entry__ as g
    from nucliadb_amd import _lib
    from nucliadb_amd.shard_merge import ShardComm

    g.build()
    ident = ShardComm.unique_id_shm()
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_shard_exchange_worker.py")
    outs = [str(tmp_path / ("rank%d.npz" % r)) for r in range(world)]
    procs = [subprocess.Popen([sys.executable, worker, ident.hex(), str(r), str(world), outs[r]], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(world)]
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for x in procs:
                x.kill()
            raise
        logs.append(o.decode(errors="replace"))
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    res = [np.load(o) for o in outs]
    ids = SHARD_IDS[:world]
    for case, (B, k, limit) in enumerate(((5, 10, 10), (257, 10, 4), (64, 7, 20))):
        score, idn, count = _lists(world, B, k, 500 + case)
        bs, ba, bv, bc = _bm25_lists(world, B, k, 600 + case)
        for r in res:
            ms, mi, mc = r["v%d_score" % case], r["v%d_id" % case], r["v%d_count" % case]
            for q in range(B):
                lists = [[(float(score[p, q, i]), int(idn[p, q, i])) for i in range(count[p, q])] for p in range(world)]
                assert [(float(ms[q, i]), int(mi[q, i])) for i in range(mc[q])] == orc.merge_vector(lists, limit), (case, q)
            os_, oa, orank, oc = r["b%d_score" % case], r["b%d_addr" % case], r["b%d_rank" % case], r["b%d_count" % case]
            for q in range(B):
                lists = [[(float(bs[p, q, i]), int(ba[p, q, i]), ids[p], p) for i in range(bc[p, q])] for p in range(world)]
                want = orc.merge_bm25(lists, limit)
                assert [(float(os_[q, i]), int(oa[q, i]), ids[int(orank[q, i])]) for i in range(oc[q])] == [(w[0], w[1], w[2]) for w in want], (case, q)
            for name, sign in (("d", 1), ("a", -1)):
                v = bv if sign == 1 else -bv
                vl, vc, vv = r["%s%d_rank" % (name, case)], r["%s%d_count" % (name, case)], r["%s%d_value" % (name, case)]
                for q in range(B):
                    heads = [[int(v[p, q, i]) for i in range(bc[p, q])] for p in range(world)]
                    assert [(int(vl[q, i]), int(vv[q, i])) for i in range(vc[q])] == _kmerge_strict(heads, limit, desc=(sign == 1)), (name, case, q)
        for r in res[1:]:   # every rank holds the same merged lists
            for key in res[0].files:
                assert np.array_equal(r[key], res[0][key]), key
    assert all(int(r["shape_error"][0]) == 1 for r in res)


@pytest.mark.gpu
def test_bench_launches_its_own_ranks_and_runs_the_library_exchange():
    """`python bench.py --gpus 2` without a launcher spawns its two ranks itself (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set per
    child); on this 1-GPU box NIDX_BENCH_SAME_DEVICE=1 lets them share GPU 0 with the DATA-PATH exchange running through the
    library (nidx_gpu_shard_exchange_merge_vector over its shared-memory transport: pack, gather layout, shard order and merge
    kernels are the RCCL path's own code).  The one JSON line must say n_gpus = 2, carry the recall of the MERGED hits against the
    merged exact scan, and `exchange_check: ok` (overlapped pipeline == plain search -> exchange; library exchange == the same
    exchange through torch.distributed).  Without the variable, asking for more GPUs than the node has is an error, not a 1-GPU run.
    Merge semantics: nidx/src/searcher/shard_merge.rs:332-348."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    args = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--n-vectors", "200000", "--corpus", "clustered", "--steps", "5", "--warmup", "2",
            "--min-timed-s", "0.2", "--parity-queries", "0", "--scan-check-queries", "0", "--segment-regime", "0", "--bf16-block-n", "0", "--ref-build-n", "0",
            "--single-query-calls", "0", "--cpu-queries", "0", "--bm25-block", "0", "--iso-recall", "0"]
    import torch

    if torch.cuda.device_count() < 2:
        r = subprocess.run(args, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode != 0 and "NIDX_BENCH_SAME_DEVICE" in r.stderr, (r.returncode, r.stderr[-2000:])
    r = subprocess.run(args, env=dict(env, NIDX_BENCH_SAME_DEVICE="1"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["shards"] == 2 and line["config"]["corpus_vectors"] == 400000
    assert line["config"]["exchange_check"] == "ok", line["config"]["exchange_check"]
    assert "shared-memory transport" in line["config"]["timed_region"]["entry"], line["config"]["timed_region"]["entry"]
    assert line["config"]["recall_at_10"] >= 0.9, line["config"]["recall_at_10"]
    assert line["value"] > 0 and not line.get("failures"), line.get("failures")
