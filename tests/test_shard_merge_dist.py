"""The N>1 path on CPU: world_size-2 (and -4) gloo processes exchange per-shard top-k lists with the same
all-gather + merge code bench.py runs over RCCL, and both ranks must end with the oracle's
merge_vector_responses / sort_documents_fn result (shard_merge.rs:211-234,332-348)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _shard_lists(rank, B, k, seed):
    """Deterministic per-shard hits: coarse scores (many cross-shard ties), ragged counts."""
    rng = np.random.default_rng(seed + rank)
    score = np.zeros((B, k), np.float32)
    ident = np.zeros((B, k), np.int64)
    count = rng.integers(0, k + 1, B).astype(np.int32)
    for q in range(B):
        s = np.sort(rng.integers(0, 7, count[q]).astype(np.float32) / 4)[::-1]
        score[q, : count[q]] = s
        ident[q, : count[q]] = (rank << 32) | np.arange(count[q])
    return score, ident, count


# compared as bytes, descending (shard_merge.rs:211-234): not in rank order, one a prefix of another
SHARD_IDS = [b"shard-b", b"shard-a", b"shard", b"shard-c"]


def _worker(rank, world, port, B, k, limit, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nucliadb_amd.shard_merge import exchange_and_merge_bm25, exchange_and_merge_vector

    score, ident, count = _shard_lists(rank, B, k, 100)
    ms, mi, mc = exchange_and_merge_vector(torch.from_numpy(score), torch.from_numpy(ident), torch.from_numpy(count), limit)
    # BM25 lists: docaddr ascending inside equal scores, shard ids compared as bytes
    bs, ba, bc = _shard_lists(rank, B, k, 200)
    ba = np.sort(np.abs(ba) & 0xFFFF, axis=1)
    shard_ids = SHARD_IDS[:world]
    os_, oa, osh, oc = exchange_and_merge_bm25(torch.from_numpy(bs), torch.from_numpy(ba), torch.from_numpy(bc), shard_ids, limit)
    out[rank] = (ms.numpy().copy(), mi.numpy().copy(), mc.numpy().copy(), os_, oa, osh, oc)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_world2_gloo_exchange_matches_oracle(orc, world):
    import __graft_entry__ as g

    g.build()
    B, k, limit = 33, 10, 10
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), B, k, limit, out), nprocs=world, join=True)
    assert set(out.keys()) == set(range(world))
    for r in range(1, world):
        for a, b in zip(out[0], out[r]):
            assert np.array_equal(a, b), "ranks disagree after the exchange"
    ms, mi, mc, bs, ba, bsh, bc = out[0]
    shards = [_shard_lists(r, B, k, 100) for r in range(world)]
    bshards = [_shard_lists(r, B, k, 200) for r in range(world)]
    shard_ids = SHARD_IDS[:world]
    for q in range(B):
        lists = [[(float(s[q, i]), int(d[q, i])) for i in range(c[q])] for s, d, c in shards]
        want = orc.merge_vector(lists, limit)
        got = [(float(ms[q, i]), int(mi[q, i])) for i in range(mc[q])]
        assert got == want, (q, got, want)
        blists = []
        for r, (s, d, c) in enumerate(bshards):
            addr = np.sort(np.abs(d) & 0xFFFF, axis=1)
            blists.append([(float(s[q, i]), int(addr[q, i]), shard_ids[r], r) for i in range(c[q])])
        bwant = orc.merge_bm25(blists, limit)
        bgot = [(float(bs[q, i]), int(ba[q, i]), shard_ids[int(bsh[q, i])]) for i in range(bc[q])]
        assert bgot == [(w[0], w[1], w[2]) for w in bwant], (q, bgot, bwant)
