// shard_comm.cpp — the multi-GPU exchange behind the C ABI: one index shard per GPU, one process per GPU, RCCL over xGMI.
//
// The reference scatters a request to its searcher nodes over gRPC and merges the per-shard responses with merge_search
// (nidx/src/searcher/shard_merge.rs:54-99).  Inside an 8-GPU node each rank searches its own shard; what it then has to see of
// the others is their per-query top-k — k hits of 12 B (vector: score + id) or 20-28 B (BM25: score + docaddr [+ sort value]) —
// so the exchange is ONE ncclAllGather of a packed block per rank (scores | ids | [values] | counts; 120 KiB per rank at 1 024
// queries x 10 hits: latency-bound, a single fully connected hop over xGMI, no ring of large buffers) followed by the
// reference's k-way merge on every rank (shard_merge_device.hip).  No torch, no host staging: a Rust host drives it with
//   rank 0:  nidx_gpu_shard_comm_unique_id(id)  ->  (the host language ships the 128 bytes to the other ranks)
//   all:     nidx_gpu_shard_comm_init(id, rank, world, shard_id, ..)  ->  nidx_gpu_shard_exchange_merge_{vector,bm25}(..)
// librccl is bound at the first comm call (dlopen): a single-GPU host never maps it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <memory>
#include <mutex>

#include "host_common.h"
#include "shard_merge_device.h"

namespace nidx {
namespace {

struct Rccl {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
};

Rccl *rccl() {
    static std::mutex mu;
    static Rccl *lib = nullptr;   // never unloaded: communicators may outlive any static destructor order
    std::lock_guard<std::mutex> g(mu);
    if (lib && lib->handle) return lib;
    if (!lib) lib = new Rccl;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
        lib->handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (lib->handle) break;
    }
    if (!lib->handle) {
        const char *e = dlerror();
        lib->error = e ? e : "librccl.so.1 not found";
        return lib;
    }
    bool ok = true;
    auto sym = [&](const char *name) {
        void *p = dlsym(lib->handle, name);
        if (!p) { ok = false; lib->error = std::string("librccl lacks ") + name; }
        return p;
    };
    lib->GetUniqueId = reinterpret_cast<decltype(lib->GetUniqueId)>(sym("ncclGetUniqueId"));
    lib->CommInitRank = reinterpret_cast<decltype(lib->CommInitRank)>(sym("ncclCommInitRank"));
    lib->CommDestroy = reinterpret_cast<decltype(lib->CommDestroy)>(sym("ncclCommDestroy"));
    lib->AllGather = reinterpret_cast<decltype(lib->AllGather)>(sym("ncclAllGather"));
    lib->GetErrorString = reinterpret_cast<decltype(lib->GetErrorString)>(sym("ncclGetErrorString"));
    if (!ok) {
        dlclose(lib->handle);
        lib->handle = nullptr;
    }
    return lib;
}

int32_t rccl_fail(Rccl *r, ncclResult_t e, const char *what) {
    return fail(NIDX_ERR_DEVICE, "RCCL: %s failed: %s", what, r->GetErrorString ? r->GetErrorString(e) : "?");
}
#define NIDX_RCCL(r, expr)                                          \
    do {                                                            \
        ncclResult_t _e = (expr);                                   \
        if (_e != ncclSuccess) return rccl_fail((r), _e, #expr);    \
    } while (0)

constexpr uint32_t SHARD_ID_MAX = 120;   // a shard id is a uuid string (36 bytes) in the reference

}  // namespace

struct ShardComm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    uint32_t shard_order[MERGE_MAX_LISTS] = {0};   // rank of every rank's shard id in bytewise order
    std::mutex mu;          // one exchange at a time per communicator (collectives must be issued in the same order on every rank)
    DevBuf gather;          // [world] packed blocks, grow-only
    ~ShardComm() {
        if (comm) {
            Rccl *r = rccl();
            if (r->handle) (void)r->CommDestroy(comm);
        }
    }
};

namespace {

// block of one rank: scores [nq*k] f32 | pad to 8 | ids [nq*k] u64 | values [nq*k] i64 (optional) | counts [nq] u32 | pad to 16
struct BlockLayout {
    size_t off_ids, off_values, off_counts, bytes;
};
BlockLayout block_layout(uint32_t nq, uint32_t k, bool with_values) {
    BlockLayout b;
    size_t at = ((size_t)nq * k * 4 + 7) & ~(size_t)7;
    b.off_ids = at;
    at += (size_t)nq * k * 8;
    b.off_values = at;
    if (with_values) at += (size_t)nq * k * 8;
    b.off_counts = at;
    at += (size_t)nq * 4;
    b.bytes = (at + 15) & ~(size_t)15;
    return b;
}

int32_t exchange_and_merge(ShardComm *c, const float *d_scores, const uint64_t *d_ids, const int64_t *d_values, const uint32_t *d_counts,
                           uint32_t nq, uint32_t k, uint32_t limit, int mode, float *d_out_score, uint64_t *d_out_id,
                           int64_t *d_out_value, uint32_t *d_out_list, uint32_t *d_out_count, hipStream_t st) {
    if (!c || !d_scores || !d_ids || !d_counts || !d_out_count) return fail(NIDX_ERR_INVALID_ARGUMENT, "NULL argument");
    if (nq == 0) return NIDX_OK;
    if (k == 0 || limit == 0) return fail(NIDX_ERR_INVALID_ARGUMENT, "k and limit must be positive");
    Rccl *r = rccl();
    if (!r->handle) return fail(NIDX_ERR_DEVICE, "RCCL is not available: %s", r->error.c_str());
    std::lock_guard<std::mutex> g(c->mu);
    NIDX_HIP(hipSetDevice(c->device));
    const BlockLayout L = block_layout(nq, k, d_values != nullptr);
    if ((size_t)c->world * L.bytes > c->gather.bytes) {
        NIDX_HIP(hipStreamSynchronize(st));   // a previous exchange on this stream may still read the old buffer
        NIDX_HIP(c->gather.reserve((size_t)c->world * L.bytes));
    }
    uint8_t *base = c->gather.as<uint8_t>();
    uint8_t *mine = base + (size_t)c->rank * L.bytes;
    // pack this rank's lists into its own slot of the gather buffer (in-place all-gather: no send buffer)
    NIDX_HIP(hipMemcpyAsync(mine, d_scores, (size_t)nq * k * 4, hipMemcpyDeviceToDevice, st));
    NIDX_HIP(hipMemcpyAsync(mine + L.off_ids, d_ids, (size_t)nq * k * 8, hipMemcpyDeviceToDevice, st));
    if (d_values) NIDX_HIP(hipMemcpyAsync(mine + L.off_values, d_values, (size_t)nq * k * 8, hipMemcpyDeviceToDevice, st));
    NIDX_HIP(hipMemcpyAsync(mine + L.off_counts, d_counts, (size_t)nq * 4, hipMemcpyDeviceToDevice, st));
    NIDX_RCCL(r, r->AllGather(mine, base, L.bytes, ncclUint8, c->comm, st));
    MergeListsArgs a{};
    a.scores = base, a.scores_stride = L.bytes;
    a.ids = base + L.off_ids, a.ids_stride = L.bytes;
    a.values = d_values ? base + L.off_values : nullptr, a.values_stride = L.bytes;
    a.counts = base + L.off_counts, a.counts_stride = L.bytes;
    a.n_lists = (uint32_t)c->world, a.n_queries = nq, a.k = k, a.limit = limit;
    memcpy(a.shard_order, c->shard_order, sizeof(a.shard_order));
    a.out_score = d_out_score, a.out_id = d_out_id, a.out_value = d_values ? d_out_value : nullptr, a.out_list = d_out_list, a.out_count = d_out_count;
    return launch_merge_lists(a, mode, st);
}

}  // namespace
}  // namespace nidx

using namespace nidx;

extern "C" {

int32_t nidx_gpu_shard_comm_unique_id(uint8_t *id_out) try {
    if (!id_out) return fail(NIDX_ERR_INVALID_ARGUMENT, "NULL argument");
    Rccl *r = rccl();
    if (!r->handle) return fail(NIDX_ERR_DEVICE, "RCCL is not available: %s", r->error.c_str());
    static_assert(sizeof(ncclUniqueId) == NIDX_SHARD_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    NIDX_RCCL(r, r->GetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return NIDX_OK;
} NIDX_ABI_CATCH

int32_t nidx_gpu_shard_comm_init(const uint8_t *unique_id, int32_t rank, int32_t world, const uint8_t *shard_id, uint32_t shard_id_len,
                                 nidx_gpu_shard_comm_t **comm_out) try {
    if (!unique_id || !comm_out || (shard_id_len && !shard_id)) return fail(NIDX_ERR_INVALID_ARGUMENT, "NULL argument");
    *comm_out = nullptr;
    if (world < 1 || world > MERGE_MAX_LISTS || rank < 0 || rank >= world)
        return fail(NIDX_ERR_INVALID_ARGUMENT, "rank %d of %d (at most %d shards)", rank, world, MERGE_MAX_LISTS);
    if (shard_id_len > SHARD_ID_MAX) return fail(NIDX_ERR_UNSUPPORTED, "shard ids longer than %u bytes", SHARD_ID_MAX);
    Rccl *r = rccl();
    if (!r->handle) return fail(NIDX_ERR_DEVICE, "RCCL is not available: %s", r->error.c_str());
    std::unique_ptr<ShardComm> c(new ShardComm());
    c->rank = rank;
    c->world = world;
    NIDX_HIP(hipGetDevice(&c->device));
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    NIDX_RCCL(r, r->CommInitRank(&c->comm, world, id, rank));
    // every rank learns every shard id once (128-byte records: length + bytes), and with them the byte order the BM25 comparator
    // needs (`a.shard_id.cmp(&b.shard_id)`, shard_merge.rs:224,302)
    DevBuf ids;
    NIDX_HIP(ids.alloc((size_t)world * 128));
    uint8_t rec[128] = {0};
    memcpy(rec, &shard_id_len, 4);
    if (shard_id_len) memcpy(rec + 4, shard_id, shard_id_len);
    NIDX_HIP(hipMemcpy(ids.as<uint8_t>() + (size_t)rank * 128, rec, 128, hipMemcpyHostToDevice));
    NIDX_RCCL(r, r->AllGather(ids.as<uint8_t>() + (size_t)rank * 128, ids.p, 128, ncclUint8, c->comm, nullptr));
    NIDX_HIP(hipStreamSynchronize(nullptr));
    std::vector<uint8_t> all((size_t)world * 128);
    NIDX_HIP(hipMemcpy(all.data(), ids.p, all.size(), hipMemcpyDeviceToHost));
    std::vector<const uint8_t *> ptrs(world);
    std::vector<uint32_t> lens(world);
    for (int i = 0; i < world; i++) {
        memcpy(&lens[i], &all[(size_t)i * 128], 4);
        if (lens[i] > SHARD_ID_MAX) return fail(NIDX_ERR_DEVICE, "RCCL: the shard-id exchange returned garbage (rank %d)", i);
        ptrs[i] = &all[(size_t)i * 128 + 4];
    }
    shard_order_from_ids(ptrs.data(), lens.data(), (uint32_t)world, c->shard_order);
    *comm_out = reinterpret_cast<nidx_gpu_shard_comm_t *>(c.release());
    return NIDX_OK;
} NIDX_ABI_CATCH

void nidx_gpu_shard_comm_destroy(nidx_gpu_shard_comm_t *comm) {
    ShardComm *c = reinterpret_cast<ShardComm *>(comm);
    if (!c) return;
    (void)hipSetDevice(c->device);
    delete c;
}

int32_t nidx_gpu_shard_exchange_merge_vector(nidx_gpu_shard_comm_t *comm, const float *d_scores, const uint64_t *d_ids, const uint32_t *d_counts,
                                             uint32_t n_queries, uint32_t k, uint32_t limit, float *d_out_score, uint64_t *d_out_id,
                                             uint32_t *d_out_count, void *stream) try {
    return exchange_and_merge(reinterpret_cast<ShardComm *>(comm), d_scores, d_ids, nullptr, d_counts, n_queries, k, limit, 0, d_out_score, d_out_id,
                              nullptr, nullptr, d_out_count, (hipStream_t)stream);
} NIDX_ABI_CATCH

int32_t nidx_gpu_shard_exchange_merge_bm25(nidx_gpu_shard_comm_t *comm, const float *d_scores, const uint64_t *d_docaddrs,
                                           const int64_t *d_order_values, const uint32_t *d_counts, uint32_t n_queries, uint32_t k, uint32_t limit,
                                           int32_t order, float *d_out_score, uint64_t *d_out_docaddr, int64_t *d_out_order_value,
                                           uint32_t *d_out_rank, uint32_t *d_out_count, void *stream) try {
    if (order < 0 || order > NIDX_MERGE_ORDER_VALUE_ASC) return fail(NIDX_ERR_INVALID_ARGUMENT, "unknown merge order %d", order);
    if (order != NIDX_MERGE_ORDER_SCORE && !d_order_values) return fail(NIDX_ERR_INVALID_ARGUMENT, "ordering by value needs d_order_values");
    const int mode = order == NIDX_MERGE_ORDER_SCORE ? 1 : (order == NIDX_MERGE_ORDER_VALUE_DESC ? 2 : 3);
    return exchange_and_merge(reinterpret_cast<ShardComm *>(comm), d_scores, d_docaddrs, d_order_values, d_counts, n_queries, k, limit, mode,
                              d_out_score, d_out_docaddr, d_out_order_value, d_out_rank, d_out_count, (hipStream_t)stream);
} NIDX_ABI_CATCH

}  // extern "C"
