"""Multi-shard result exchange + merge: the MI355X-native form of `nidx::searcher::shard_merge`.

The reference scatters a request to the searcher nodes over gRPC and merges the per-shard responses
with `merge_search` (nidx/src/searcher/shard_merge.rs:54-99).  Inside one 8-GPU node every GPU owns
one index shard; each produces `[B][k]` (score f32, id u64) hits plus a count per query, and the
exchange is ONE all-gather per tensor over RCCL/xGMI (120 KiB per GPU at B=1024, k=10: latency
bound, fully connected single hop) followed by the reference's k-way merge on every rank:

  vector    kmerge_by(a.score >= b.score).take(limit)                     shard_merge.rs:332-348
  bm25      bm25 desc (total_cmp), shard_id desc (bytes), docaddr asc     shard_merge.rs:211-234,289-312

With CUDA/HIP tensors the merge runs in the HIP kernel (nidx_gpu_merge_vector_device); with CPU
tensors (gloo, used by the multi-process tests) it runs in the library's host entry points — the
same comparator code either way, never a Python re-implementation.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

from . import _lib


def _world(group) -> Tuple[int, int]:
    if not dist.is_available() or not dist.is_initialized():
        return 0, 1
    return dist.get_rank(group), dist.get_world_size(group)


def all_gather_hits(score: torch.Tensor, ident: torch.Tensor, count: torch.Tensor, group=None):
    """All-gather of the per-shard top-k: returns ([P][B][k] score, [P][B][k] id, [P][B] count)."""
    _, world = _world(group)
    if world == 1:
        return score.unsqueeze(0), ident.unsqueeze(0), count.unsqueeze(0)
    def gather(t: torch.Tensor) -> torch.Tensor:
        # concatenated along dim 0 (the layout both RCCL and gloo accept), viewed as [P][...]
        out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, t.contiguous(), group=group)
        return out.view((world,) + tuple(t.shape))

    return gather(score), gather(ident), gather(count)


def merge_vector_lists(g_score: torch.Tensor, g_id: torch.Tensor, g_count: torch.Tensor, limit: int):
    """merge_vector_responses over P shard lists for every query of the batch.
    g_score [P][B][k] f32, g_id [P][B][k] i64, g_count [P][B] i32 -> ([B][limit], [B][limit], [B])."""
    L = _lib.lib()
    P, B, k = g_score.shape
    dev = g_score.device
    out_score = torch.zeros((B, limit), dtype=torch.float32, device=dev)
    out_id = torch.zeros((B, limit), dtype=torch.int64, device=dev)
    out_count = torch.zeros((B,), dtype=torch.int32, device=dev)
    if dev.type == "cuda":
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(L.nidx_gpu_merge_vector_device(g_score.data_ptr(), g_id.data_ptr(), g_count.data_ptr(), P, B, k, limit,
                                                  out_score.data_ptr(), out_id.data_ptr(), out_count.data_ptr(), stream))
        return out_score, out_id, out_count
    sc = g_score.contiguous().numpy()
    ids = g_id.contiguous().numpy().view(np.uint64)
    cnt = g_count.contiguous().numpy()
    os_, oi, oc = out_score.numpy(), out_id.numpy().view(np.uint64), out_count.numpy()
    lens = np.zeros(P, dtype=np.uint32)
    n = C.c_uint32()
    for q in range(B):
        rows_s = [np.ascontiguousarray(sc[p, q]) for p in range(P)]
        rows_i = [np.ascontiguousarray(ids[p, q]) for p in range(P)]
        lens[:] = cnt[:, q]
        ps = (C.c_void_p * P)(*[r.ctypes.data for r in rows_s])
        pi = (C.c_void_p * P)(*[r.ctypes.data for r in rows_i])
        row_s, row_i = np.zeros(limit, np.float32), np.zeros(limit, np.uint64)
        _lib.check(L.nidx_gpu_merge_vector(ps, pi, lens.ctypes.data, P, limit, row_s.ctypes.data, row_i.ctypes.data, None, C.byref(n)))
        os_[q], oi[q], oc[q] = row_s, row_i, n.value
    return out_score, out_id, out_count


def exchange_and_merge_vector(score: torch.Tensor, ident: torch.Tensor, count: torch.Tensor, limit: int, group=None):
    """K10 of SURVEY §2c: what every rank runs after its shard search."""
    return merge_vector_lists(*all_gather_hits(score, ident, count, group), limit)


def exchange_and_merge_bm25(score: torch.Tensor, docaddr: torch.Tensor, count: torch.Tensor, shard_ids: Sequence[bytes],
                            limit: int, group=None):
    """BM25 document/paragraph merge; shard_ids[p] = shard id of rank p (compared as bytes, descending).
    Host merge (the lists are k entries per shard; the reference budgets microseconds for it)."""
    L = _lib.lib()
    g_score, g_addr, g_count = all_gather_hits(score, docaddr, count, group)
    P, B, k = g_score.shape
    sc = g_score.cpu().contiguous().numpy()
    da = g_addr.cpu().contiguous().numpy().view(np.uint64)
    cnt = g_count.cpu().contiguous().numpy()
    sid = [np.frombuffer(bytes(s), np.uint8).copy() for s in shard_ids]
    sidl = np.array([len(s) for s in shard_ids], np.uint32)
    psid = (C.c_void_p * P)(*[a.ctypes.data for a in sid])
    out_score = np.zeros((B, limit), np.float32)
    out_addr = np.zeros((B, limit), np.uint64)
    out_shard = np.zeros((B, limit), np.uint32)
    out_count = np.zeros(B, np.uint32)
    lens = np.zeros(P, np.uint32)
    n = C.c_uint32()
    for q in range(B):
        rows_s = [np.ascontiguousarray(sc[p, q]) for p in range(P)]
        rows_a = [np.ascontiguousarray(da[p, q]) for p in range(P)]
        lens[:] = cnt[:, q]
        ps = (C.c_void_p * P)(*[r.ctypes.data for r in rows_s])
        pa = (C.c_void_p * P)(*[r.ctypes.data for r in rows_a])
        _lib.check(L.nidx_gpu_merge_bm25(ps, pa, lens.ctypes.data, psid, sidl.ctypes.data, P, limit, out_score[q].ctypes.data,
                                         out_addr[q].ctypes.data, out_shard[q].ctypes.data, C.byref(n)))
        out_count[q] = n.value
    return out_score, out_addr, out_shard, out_count
