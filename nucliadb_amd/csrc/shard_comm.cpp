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
//
// The all-gather sits behind a two-entry transport table: RCCL (the product path), and a POSIX shared-memory gather for processes of
// one node (nidx_gpu_shard_comm_unique_id_shm: every rank copies its block to a shared segment, a barrier, every rank copies all
// blocks back) — slow, but it runs N ranks on ONE GPU, which is what a test box has: the pack / gather offsets / shard order /
// merge code around the transport is the same for both, so a world-of-2 run exercises everything but ncclAllGather itself.
#include <dlfcn.h>
#include <fcntl.h>
#include <rccl/rccl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cerrno>
#include <chrono>
#include <cstdio>
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

// ---- the shared-memory transport: a segment of [header | world x slot]; every collective is copy-in, barrier, copy-out, barrier ----
struct ShmHeader {
    std::atomic<uint32_t> magic;       // set by the creator once the segment is sized
    std::atomic<uint32_t> arrived;     // ranks inside the current barrier
    std::atomic<uint32_t> generation;  // barriers completed
    uint32_t world;
    uint64_t slot_bytes;
};
constexpr uint32_t SHM_MAGIC = 0x4e494458u;   // "NIDX"
constexpr size_t SHM_HEADER_BYTES = 4096;
constexpr uint64_t SHM_SLOT_BYTES = 32ull << 20;   // per rank: 1 024 queries x 512 hits x 28 B fit with room to spare
constexpr const char *SHM_ID_PREFIX = "nidx-shm:";

struct ShmTransport {
    std::string name;
    int fd = -1;
    uint8_t *map = nullptr;
    size_t map_bytes = 0;
    bool creator = false;
    ShmHeader *hdr() const { return reinterpret_cast<ShmHeader *>(map); }
    uint8_t *slot(int rank) const { return map + SHM_HEADER_BYTES + (size_t)rank * SHM_SLOT_BYTES; }
    ~ShmTransport() {
        if (map) munmap(map, map_bytes);
        if (fd >= 0) close(fd);
        if (creator) shm_unlink(name.c_str());
    }
    int32_t open(const char *nm, int rank, int world) {
        name = nm;
        map_bytes = SHM_HEADER_BYTES + (size_t)world * SHM_SLOT_BYTES;   // sparse: pages exist once they are written
        const auto t0 = std::chrono::steady_clock::now();
        if (rank == 0) {
            fd = shm_open(nm, O_CREAT | O_RDWR, 0600);
            if (fd < 0) return fail(NIDX_ERR_IO, "shm_open(%s): %s", nm, strerror(errno));
            creator = true;
            if (ftruncate(fd, (off_t)map_bytes) != 0) return fail(NIDX_ERR_IO, "ftruncate(%s): %s", nm, strerror(errno));
        } else {
            for (;;) {   // rank 0 may not have created it yet
                fd = shm_open(nm, O_RDWR, 0600);
                struct stat sb;
                if (fd >= 0 && fstat(fd, &sb) == 0 && (size_t)sb.st_size >= map_bytes) break;
                if (fd >= 0) close(fd), fd = -1;
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) return fail(NIDX_ERR_IO, "shared segment %s did not appear", nm);
                usleep(1000);
            }
        }
        void *m = mmap(nullptr, map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        if (m == MAP_FAILED) return fail(NIDX_ERR_IO, "mmap(%s): %s", nm, strerror(errno));
        map = static_cast<uint8_t *>(m);
        if (rank == 0) {
            hdr()->world = (uint32_t)world;
            hdr()->slot_bytes = SHM_SLOT_BYTES;
            hdr()->arrived.store(0);
            hdr()->generation.store(0);
            hdr()->magic.store(SHM_MAGIC, std::memory_order_release);
        } else {
            while (hdr()->magic.load(std::memory_order_acquire) != SHM_MAGIC) {
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) return fail(NIDX_ERR_IO, "shared segment %s was never initialised", nm);
                usleep(1000);
            }
            if (hdr()->world != (uint32_t)world) return fail(NIDX_ERR_INVALID_ARGUMENT, "shared segment %s belongs to a world of %u", nm, hdr()->world);
        }
        return NIDX_OK;
    }
    int32_t barrier(int world) {
        ShmHeader *h = hdr();
        const uint32_t gen = h->generation.load(std::memory_order_acquire);
        if (h->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)world) {
            h->arrived.store(0, std::memory_order_relaxed);
            h->generation.store(gen + 1, std::memory_order_release);
            return NIDX_OK;
        }
        const auto t0 = std::chrono::steady_clock::now();
        while (h->generation.load(std::memory_order_acquire) == gen) {
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) return fail(NIDX_ERR_DEVICE, "shared-memory exchange: a rank did not arrive");
            sched_yield();
        }
        return NIDX_OK;
    }
};

struct ShardComm {
    ncclComm_t comm = nullptr;            // transport 1: RCCL
    std::unique_ptr<ShmTransport> shm;    // transport 2: shared memory (same-node test processes)
    int rank = 0, world = 1, device = 0;
    uint32_t shard_order[MERGE_MAX_LISTS] = {0};   // rank of every rank's shard id in bytewise order
    std::mutex mu;          // one exchange at a time per communicator (collectives must be issued in the same order on every rank)
    DevBuf gather;          // [world] packed blocks, grow-only
    hipEvent_t merged = nullptr;   // the last merge kernel's end: the next exchange (possibly on another stream) rewrites `gather`
    bool merged_recorded = false;
    bool check_shapes = false;     // all-gather (n_queries, k, values?) first and compare (always on for the shared-memory transport)
    ~ShardComm() {
        if (merged) (void)hipEventDestroy(merged);
        if (comm) {
            Rccl *r = rccl();
            if (r->handle) (void)r->CommDestroy(comm);
        }
    }
    // In-place all-gather of `bytes` per rank inside `base` (rank i's block at base + i * bytes), device memory, ordered on `st`.
    int32_t all_gather(uint8_t *base, size_t bytes, hipStream_t st) {
        if (shm) {
            if (bytes > SHM_SLOT_BYTES) return fail(NIDX_ERR_UNSUPPORTED, "shared-memory transport: a block of %zu bytes exceeds the %llu-byte slot", bytes, (unsigned long long)SHM_SLOT_BYTES);
            NIDX_HIP(hipMemcpyAsync(shm->slot(rank), base + (size_t)rank * bytes, bytes, hipMemcpyDeviceToHost, st));
            NIDX_HIP(hipStreamSynchronize(st));
            int32_t rc = shm->barrier(world);
            if (rc != NIDX_OK) return rc;
            for (int i = 0; i < world; i++)
                if (i != rank) NIDX_HIP(hipMemcpyAsync(base + (size_t)i * bytes, shm->slot(i), bytes, hipMemcpyHostToDevice, st));
            NIDX_HIP(hipStreamSynchronize(st));
            return shm->barrier(world);   // nobody rewrites its slot before everybody has read it
        }
        Rccl *r = rccl();
        if (!r->handle) return fail(NIDX_ERR_DEVICE, "RCCL is not available: %s", r->error.c_str());
        NIDX_RCCL(r, r->AllGather(base + (size_t)rank * bytes, base, bytes, ncclUint8, comm, st));
        return NIDX_OK;
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
    if (nq == 0) return NIDX_OK;   // (a collective like every other call: every rank passes the same n_queries — also when it is 0)
    if (k == 0 || limit == 0) return fail(NIDX_ERR_INVALID_ARGUMENT, "k and limit must be positive");
    std::lock_guard<std::mutex> g(c->mu);
    NIDX_HIP(hipSetDevice(c->device));
    const BlockLayout L = block_layout(nq, k, d_values != nullptr);
    // the gather buffer is shared by every exchange of this communicator: the previous merge (whatever stream it ran on) has to be
    // through with it before it is rewritten
    if (c->merged_recorded) NIDX_HIP(hipStreamWaitEvent(st, c->merged, 0));
    const size_t need = std::max<size_t>((size_t)c->world * L.bytes, (size_t)c->world * 16);
    if (need > c->gather.bytes) {
        NIDX_HIP(hipStreamSynchronize(st));   // a previous exchange may still read the old buffer
        NIDX_HIP(c->gather.reserve(need));
    }
    uint8_t *base = c->gather.as<uint8_t>();
    if (c->check_shapes) {
        // every rank must bring the same block shape, or the gather offsets of the ranks disagree (a hang or garbage with RCCL)
        uint32_t mine_shape[4] = {nq, k, d_values ? 1u : 0u, (uint32_t)mode};
        NIDX_HIP(hipMemcpyAsync(base + (size_t)c->rank * 16, mine_shape, 16, hipMemcpyHostToDevice, st));
        NIDX_HIP(hipStreamSynchronize(st));
        int32_t rc = c->all_gather(base, 16, st);
        if (rc != NIDX_OK) return rc;
        std::vector<uint32_t> all((size_t)c->world * 4);
        NIDX_HIP(hipMemcpyAsync(all.data(), base, all.size() * 4, hipMemcpyDeviceToHost, st));
        NIDX_HIP(hipStreamSynchronize(st));
        for (int i = 0; i < c->world; i++)
            if (memcmp(&all[(size_t)i * 4], mine_shape, 16) != 0)
                return fail(NIDX_ERR_INVALID_ARGUMENT, "shard exchange: rank %d brought n_queries=%u k=%u values=%u order=%u, this rank %u %u %u %u", i, all[i * 4],
                            all[i * 4 + 1], all[i * 4 + 2], all[i * 4 + 3], mine_shape[0], mine_shape[1], mine_shape[2], mine_shape[3]);
    }
    uint8_t *mine = base + (size_t)c->rank * L.bytes;
    // pack this rank's lists into its own slot of the gather buffer (in-place all-gather: no send buffer)
    NIDX_HIP(hipMemcpyAsync(mine, d_scores, (size_t)nq * k * 4, hipMemcpyDeviceToDevice, st));
    NIDX_HIP(hipMemcpyAsync(mine + L.off_ids, d_ids, (size_t)nq * k * 8, hipMemcpyDeviceToDevice, st));
    if (d_values) NIDX_HIP(hipMemcpyAsync(mine + L.off_values, d_values, (size_t)nq * k * 8, hipMemcpyDeviceToDevice, st));
    NIDX_HIP(hipMemcpyAsync(mine + L.off_counts, d_counts, (size_t)nq * 4, hipMemcpyDeviceToDevice, st));
    {
        const int32_t rc = c->all_gather(base, L.bytes, st);
        if (rc != NIDX_OK) return rc;
    }
    MergeListsArgs a{};
    a.scores = base, a.scores_stride = L.bytes;
    a.ids = base + L.off_ids, a.ids_stride = L.bytes;
    a.values = d_values ? base + L.off_values : nullptr, a.values_stride = L.bytes;
    a.counts = base + L.off_counts, a.counts_stride = L.bytes;
    a.n_lists = (uint32_t)c->world, a.n_queries = nq, a.k = k, a.limit = limit;
    memcpy(a.shard_order, c->shard_order, sizeof(a.shard_order));
    a.out_score = d_out_score, a.out_id = d_out_id, a.out_value = d_values ? d_out_value : nullptr, a.out_list = d_out_list, a.out_count = d_out_count;
    const int32_t rc = launch_merge_lists(a, mode, st);
    if (rc != NIDX_OK) return rc;
    if (!c->merged) NIDX_HIP(hipEventCreateWithFlags(&c->merged, hipEventDisableTiming));
    NIDX_HIP(hipEventRecord(c->merged, st));
    c->merged_recorded = true;
    return NIDX_OK;
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

int32_t nidx_gpu_shard_comm_unique_id_shm(uint8_t *id_out) try {
    if (!id_out) return fail(NIDX_ERR_INVALID_ARGUMENT, "NULL argument");
    static std::atomic<uint32_t> serial{0};
    memset(id_out, 0, NIDX_SHARD_COMM_ID_BYTES);
    const auto now = std::chrono::steady_clock::now().time_since_epoch().count();
    snprintf(reinterpret_cast<char *>(id_out), NIDX_SHARD_COMM_ID_BYTES, "%s/nidx_gpu_%d_%u_%llx", SHM_ID_PREFIX, (int)getpid(), serial.fetch_add(1),
             (unsigned long long)now);
    return NIDX_OK;
} NIDX_ABI_CATCH

int32_t nidx_gpu_shard_comm_init(const uint8_t *unique_id, int32_t rank, int32_t world, const uint8_t *shard_id, uint32_t shard_id_len,
                                 nidx_gpu_shard_comm_t **comm_out) try {
    if (!unique_id || !comm_out || (shard_id_len && !shard_id)) return fail(NIDX_ERR_INVALID_ARGUMENT, "NULL argument");
    *comm_out = nullptr;
    if (world < 1 || world > MERGE_MAX_LISTS || rank < 0 || rank >= world)
        return fail(NIDX_ERR_INVALID_ARGUMENT, "rank %d of %d (at most %d shards)", rank, world, MERGE_MAX_LISTS);
    if (shard_id_len > SHARD_ID_MAX) return fail(NIDX_ERR_UNSUPPORTED, "shard ids longer than %u bytes", SHARD_ID_MAX);
    std::unique_ptr<ShardComm> c(new ShardComm());
    c->rank = rank;
    c->world = world;
    NIDX_HIP(hipGetDevice(&c->device));
    const bool use_shm = memcmp(unique_id, SHM_ID_PREFIX, strlen(SHM_ID_PREFIX)) == 0;
    if (use_shm) {
        char name[NIDX_SHARD_COMM_ID_BYTES + 1];
        memcpy(name, unique_id + strlen(SHM_ID_PREFIX), NIDX_SHARD_COMM_ID_BYTES - strlen(SHM_ID_PREFIX));
        name[NIDX_SHARD_COMM_ID_BYTES - strlen(SHM_ID_PREFIX)] = 0;
        c->shm.reset(new ShmTransport());
        const int32_t rc = c->shm->open(name, rank, world);
        if (rc != NIDX_OK) return rc;
        c->check_shapes = true;
    } else {
        Rccl *r = rccl();
        if (!r->handle) return fail(NIDX_ERR_DEVICE, "RCCL is not available: %s", r->error.c_str());
        ncclUniqueId id;
        memcpy(&id, unique_id, sizeof(id));
        NIDX_RCCL(r, r->CommInitRank(&c->comm, world, id, rank));
        c->check_shapes = getenv("NIDX_GPU_SHARD_COMM_CHECK") != nullptr;
    }
    // every rank learns every shard id once (128-byte records: length + bytes), and with them the byte order the BM25 comparator
    // needs (`a.shard_id.cmp(&b.shard_id)`, shard_merge.rs:224,302)
    DevBuf ids;
    NIDX_HIP(ids.alloc((size_t)world * 128));
    uint8_t rec[128] = {0};
    memcpy(rec, &shard_id_len, 4);
    if (shard_id_len) memcpy(rec + 4, shard_id, shard_id_len);
    NIDX_HIP(hipMemcpy(ids.as<uint8_t>() + (size_t)rank * 128, rec, 128, hipMemcpyHostToDevice));
    {
        const int32_t rc = c->all_gather(ids.as<uint8_t>(), 128, nullptr);
        if (rc != NIDX_OK) return rc;
    }
    NIDX_HIP(hipStreamSynchronize(nullptr));
    std::vector<uint8_t> all((size_t)world * 128);
    NIDX_HIP(hipMemcpy(all.data(), ids.p, all.size(), hipMemcpyDeviceToHost));
    std::vector<const uint8_t *> ptrs(world);
    std::vector<uint32_t> lens(world);
    for (int i = 0; i < world; i++) {
        memcpy(&lens[i], &all[(size_t)i * 128], 4);
        if (lens[i] > SHARD_ID_MAX) return fail(NIDX_ERR_DEVICE, "the shard-id exchange returned garbage (rank %d)", i);
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
