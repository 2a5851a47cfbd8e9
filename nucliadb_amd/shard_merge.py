"""Multi-shard result exchange + merge: the MI355X-native form of `nidx::searcher::shard_merge`.

The reference scatters a request to the searcher nodes over gRPC and merges the per-shard responses
with `merge_search` (nidx/src/searcher/shard_merge.rs:54-99).  Inside one 8-GPU node every GPU owns
one index shard; each produces `[B][k]` (score f32, id u64) hits plus a count per query.

The product path is `ShardComm` below: the library's own RCCL communicator (csrc/shard_comm.cpp: ONE
ncclAllGather of a packed block per rank + the device merge kernel) — no torch on the data path.

The `exchange_and_merge_*` functions are the same exchange driven through torch.distributed; they exist for
the multi-process CPU tests (gloo: the library's batched HOST merges run on the gathered lists) and as the
cross-check bench.py runs beside the product path at N > 1.  Either way the merge itself is library code
with the reference's comparators, never a Python re-implementation:

  vector    kmerge_by(a.score >= b.score).take(limit)                     shard_merge.rs:332-348
  bm25      bm25 desc (total_cmp), shard_id desc (bytes), docaddr asc     shard_merge.rs:211-234,289-312
  date      sort value strictly greater / smaller first                   shard_merge.rs:236-250,314-329
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

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


def _shard_id_arrays(shard_ids: Sequence[bytes]):
    sid = [np.frombuffer(bytes(s) + b"\0", np.uint8).copy() for s in shard_ids]   # (+1 byte: an empty id still has an address)
    sidl = np.array([len(s) for s in shard_ids], np.uint32)
    psid = (C.c_void_p * max(1, len(sid)))(*[a.ctypes.data for a in sid])
    return sid, sidl, psid


def merge_vector_lists(g_score: torch.Tensor, g_id: torch.Tensor, g_count: torch.Tensor, limit: int):
    """merge_vector_responses over P shard lists for every query of the batch.
    g_score [P][B][k] f32, g_id [P][B][k] i64, g_count [P][B] i32 -> ([B][limit], [B][limit], [B])."""
    L = _lib.lib()
    P, B, k = g_score.shape
    dev = g_score.device
    g_score, g_id, g_count = g_score.contiguous(), g_id.contiguous(), g_count.contiguous()
    out_score = torch.zeros((B, limit), dtype=torch.float32, device=dev)
    out_id = torch.zeros((B, limit), dtype=torch.int64, device=dev)
    out_count = torch.zeros((B,), dtype=torch.int32, device=dev)
    if dev.type == "cuda":
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(L.nidx_gpu_merge_vector_device(g_score.data_ptr(), g_id.data_ptr(), g_count.data_ptr(), P, B, k, limit,
                                                  out_score.data_ptr(), out_id.data_ptr(), out_count.data_ptr(), stream))
    else:
        _lib.check(L.nidx_gpu_merge_vector_batch(g_score.data_ptr(), g_id.data_ptr(), g_count.data_ptr(), P, B, k, limit,
                                                 out_score.data_ptr(), out_id.data_ptr(), out_count.data_ptr()))
    return out_score, out_id, out_count


def merge_bm25_lists(g_score: torch.Tensor, g_addr: torch.Tensor, g_count: torch.Tensor, shard_ids: Sequence[bytes], limit: int,
                     g_value: Optional[torch.Tensor] = None, order: int = _lib.MERGE_ORDER_SCORE):
    """merge_document_responses / merge_paragraph_responses' kmerge over P shard lists for every query of the batch.
    -> (score [B][limit] f32, docaddr [B][limit] i64, list [B][limit] i32, count [B] i32[, value [B][limit] i64])"""
    L = _lib.lib()
    P, B, k = g_score.shape
    dev = g_score.device
    g_score, g_addr, g_count = g_score.contiguous(), g_addr.contiguous(), g_count.contiguous()
    if g_value is not None:
        g_value = g_value.contiguous()
    out_score = torch.zeros((B, limit), dtype=torch.float32, device=dev)
    out_addr = torch.zeros((B, limit), dtype=torch.int64, device=dev)
    out_list = torch.zeros((B, limit), dtype=torch.int32, device=dev)
    out_value = torch.zeros((B, limit), dtype=torch.int64, device=dev) if g_value is not None else None
    out_count = torch.zeros((B,), dtype=torch.int32, device=dev)
    _keep, sidl, psid = _shard_id_arrays(shard_ids)
    args = (g_score.data_ptr(), g_addr.data_ptr(), g_value.data_ptr() if g_value is not None else None, g_count.data_ptr(), psid,
            sidl.ctypes.data, P, B, k, limit, order, out_score.data_ptr(), out_addr.data_ptr(),
            out_value.data_ptr() if out_value is not None else None, out_list.data_ptr(), out_count.data_ptr())
    if dev.type == "cuda":
        _lib.check(L.nidx_gpu_merge_bm25_device(*args, torch.cuda.current_stream(dev).cuda_stream))
    else:
        _lib.check(L.nidx_gpu_merge_bm25_batch(*args))
    res = (out_score, out_addr, out_list, out_count)
    return res + (out_value,) if out_value is not None else res


def exchange_and_merge_vector(score: torch.Tensor, ident: torch.Tensor, count: torch.Tensor, limit: int, group=None):
    """K10 of SURVEY §2c through torch.distributed (tests / cross-check; the product path is ShardComm)."""
    return merge_vector_lists(*all_gather_hits(score, ident, count, group), limit)


def exchange_and_merge_bm25(score: torch.Tensor, docaddr: torch.Tensor, count: torch.Tensor, shard_ids: Sequence[bytes],
                            limit: int, group=None):
    """BM25 document/paragraph merge; shard_ids[p] = shard id of rank p (compared as bytes, descending).
    -> numpy (score, docaddr u64, shard u32, count u32), like the library's host merge."""
    g_score, g_addr, g_count = all_gather_hits(score, docaddr, count, group)
    os_, oa, ol, oc = merge_bm25_lists(g_score, g_addr, g_count, shard_ids, limit)
    return (os_.cpu().numpy(), oa.cpu().numpy().view(np.uint64), ol.cpu().numpy().view(np.uint32), oc.cpu().numpy().view(np.uint32))


def merge_facets(shards):
    """merge_facets (shard_merge.rs:380-414).  shards: per shard a list of (group, tag, total); -> [(group, tag, total)] sorted by
    (group, tag) bytes (the reference returns a HashMap: its order is unspecified)."""
    L = _lib.lib()
    keep, arrs, lens = [], [], []
    for facets in shards:
        a = (_lib.FacetCountC * max(1, len(facets)))()
        for i, (g, t, total) in enumerate(facets):
            gb, tb = C.create_string_buffer(bytes(g), len(g)), C.create_string_buffer(bytes(t), len(t))
            keep += [gb, tb]
            a[i].group, a[i].group_len, a[i].tag, a[i].tag_len, a[i].total = C.addressof(gb), len(g), C.addressof(tb), len(t), int(total)
        arrs.append(a)
        lens.append(len(facets))
    n = len(arrs)
    ptrs = (C.c_void_p * max(1, n))(*[C.addressof(a) for a in arrs])
    clens = (C.c_uint32 * max(1, n))(*lens)
    cap = max(1, sum(lens))
    out = (_lib.FacetCountC * cap)()
    n_out = C.c_uint32()
    _lib.check(L.nidx_gpu_merge_facets(ptrs, clens, n, out, cap, C.byref(n_out)))
    return [(C.string_at(out[i].group, out[i].group_len), C.string_at(out[i].tag, out[i].tag_len), out[i].total) for i in range(n_out.value)]


class ShardComm:
    """The library's RCCL communicator (include/nidx_gpu.h "the multi-GPU exchange").  Rank 0 creates the 128-byte id with
    `ShardComm.unique_id()`, the host ships it to the other ranks (here: any broadcast), every rank constructs `ShardComm(id, rank,
    world, shard_id)` with its GPU selected, then calls the exchange collectively, in the same order on every rank."""

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_uint8 * _lib.SHARD_COMM_ID_BYTES)()
        _lib.check(_lib.lib().nidx_gpu_shard_comm_unique_id(buf))
        return bytes(buf)

    @staticmethod
    def unique_id_shm() -> bytes:
        """An id of the shared-memory transport: ranks of one node that share a GPU (tests)."""
        buf = (C.c_uint8 * _lib.SHARD_COMM_ID_BYTES)()
        _lib.check(_lib.lib().nidx_gpu_shard_comm_unique_id_shm(buf))
        return bytes(buf)

    def __init__(self, unique_id: bytes, rank: int, world: int, shard_id: bytes = b""):
        assert len(unique_id) == _lib.SHARD_COMM_ID_BYTES
        self.rank, self.world = rank, world
        self._h = C.c_void_p()
        idbuf = (C.c_uint8 * _lib.SHARD_COMM_ID_BYTES).from_buffer_copy(unique_id)
        _lib.check(_lib.lib().nidx_gpu_shard_comm_init(idbuf, rank, world, bytes(shard_id), len(shard_id), C.byref(self._h)))

    def close(self):
        if self._h:
            _lib.lib().nidx_gpu_shard_comm_destroy(self._h)
            self._h = C.c_void_p()

    def exchange_merge_vector(self, score: torch.Tensor, ident: torch.Tensor, count: torch.Tensor, limit: int, out=None, stream=None):
        """score [B][k] f32, ident [B][k] i64, count [B] i32 on this rank's GPU -> merged (score [B][limit], id, count)."""
        B, k = score.shape
        dev = score.device
        if out is None:
            out = (torch.zeros((B, limit), dtype=torch.float32, device=dev), torch.zeros((B, limit), dtype=torch.int64, device=dev),
                   torch.zeros((B,), dtype=torch.int32, device=dev))
        st = stream if stream is not None else torch.cuda.current_stream(dev).cuda_stream
        _lib.check(_lib.lib().nidx_gpu_shard_exchange_merge_vector(self._h, score.data_ptr(), ident.data_ptr(), count.data_ptr(), B, k, limit,
                                                                   out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), st))
        return out

    def exchange_merge_bm25(self, score: torch.Tensor, docaddr: torch.Tensor, count: torch.Tensor, limit: int,
                            value: Optional[torch.Tensor] = None, order: int = _lib.MERGE_ORDER_SCORE, stream=None):
        """-> (score [B][limit], docaddr, rank of origin, count[, value])"""
        B, k = score.shape
        dev = score.device
        os_ = torch.zeros((B, limit), dtype=torch.float32, device=dev)
        oa = torch.zeros((B, limit), dtype=torch.int64, device=dev)
        orank = torch.zeros((B, limit), dtype=torch.int32, device=dev)
        ov = torch.zeros((B, limit), dtype=torch.int64, device=dev) if value is not None else None
        oc = torch.zeros((B,), dtype=torch.int32, device=dev)
        st = stream if stream is not None else torch.cuda.current_stream(dev).cuda_stream
        _lib.check(_lib.lib().nidx_gpu_shard_exchange_merge_bm25(
            self._h, score.data_ptr(), docaddr.data_ptr(), value.data_ptr() if value is not None else None, count.data_ptr(), B, k, limit,
            order, os_.data_ptr(), oa.data_ptr(), ov.data_ptr() if ov is not None else None, orank.data_ptr(), oc.data_ptr(), st))
        return (os_, oa, orank, oc) + ((ov,) if ov is not None else ())
