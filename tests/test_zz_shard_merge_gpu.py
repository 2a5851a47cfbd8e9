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
