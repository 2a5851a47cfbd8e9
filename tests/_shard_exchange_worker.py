"""One rank of tests/test_zz_shard_merge_gpu.py::test_library_exchange_between_processes: python _shard_exchange_worker.py <id hex>
<rank> <world> <out.npz>.  Every rank builds ALL ranks' lists from the seeds (so the parent can too), hands its own to the library's
exchange over the shared-memory transport and saves what came back."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    from test_zz_shard_merge_gpu import SHARD_IDS, _bm25_lists, _lists

    from nucliadb_amd import _lib
    from nucliadb_amd.shard_merge import ShardComm

    ident, rank, world, out = bytes.fromhex(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    _lib.check(_lib.lib().nidx_gpu_set_device(0))
    comm = ShardComm(ident, rank, world, SHARD_IDS[rank])
    res = {}
    try:
        for case, (B, k, limit) in enumerate(((5, 10, 10), (257, 10, 4), (64, 7, 20))):
            score, idn, count = _lists(world, B, k, 500 + case)
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a[rank])).to(dev)
            ms, mi, mc = comm.exchange_merge_vector(t(score), t(idn), t(count), limit)
            res["v%d_score" % case], res["v%d_id" % case], res["v%d_count" % case] = ms.cpu().numpy(), mi.cpu().numpy(), mc.cpu().numpy()
            bs, ba, bv, bc = _bm25_lists(world, B, k, 600 + case)
            r = comm.exchange_merge_bm25(t(bs), t(ba), t(bc), limit)
            res["b%d_score" % case], res["b%d_addr" % case], res["b%d_rank" % case], res["b%d_count" % case] = [x.cpu().numpy() for x in r]
            for name, order, sign in (("d", _lib.MERGE_ORDER_VALUE_DESC, 1), ("a", _lib.MERGE_ORDER_VALUE_ASC, -1)):
                v = bv if sign == 1 else -bv
                r = comm.exchange_merge_bm25(t(bs), t(ba), t(bc), limit, value=t(v), order=order)
                res["%s%d_rank" % (name, case)], res["%s%d_count" % (name, case)], res["%s%d_value" % (name, case)] = r[2].cpu().numpy(), r[3].cpu().numpy(), r[4].cpu().numpy()
        # a rank that brings another shape is an error on every rank, not a hang
        B, k = (8, 5) if rank == 0 else (8, 6)
        z = torch.zeros((B, k), dtype=torch.float32, device=dev)
        try:
            comm.exchange_merge_vector(z, torch.zeros((B, k), dtype=torch.int64, device=dev), torch.zeros((B,), dtype=torch.int32, device=dev), 5)
            res["shape_error"] = np.array([0])
        except _lib.NidxGpuError as e:
            res["shape_error"] = np.array([1 if "brought" in str(e) else 2])
    finally:
        comm.close()
    np.savez(out, **res)


if __name__ == "__main__":
    main()
