// exchange.cpp -- the all-gatherv of hit records between the GPUs of a job, inside the library (RCCL over xGMI).
//
// The reference has no distributed mode: its unit of parallelism is the independent record / chunk
// (internal/pipeline/pipeline.go:60-125).  One process per GPU scans its own records with the whole panel -- no
// data-path collective -- and only the verified hit records (32 bytes each, tens of KB per genome) are exchanged, after
// which the amplicon join (core/engine/engine.go:108-404) runs on the gathered list (ipcr_join_hits).  This file is that
// exchange for a Go or C++ host: ncclAllGather straight out of the scratch's DEVICE hit buffer (64-byte counter header +
// hit slots), so the records never visit the host on the sending side.  ipcr_amd/dist.py is a thin caller of it.
//
// Lock step: no rank ever decides alone.  A rank whose hits exceed the exchange capacity still enters the collective
// (its header carries the true count); after the gather EVERY rank reads the same headers, sees the same overflow, and
// all of them double the capacity and repeat that exchange together (ipcr_exchange_end).  A rank whose scratch buffer is
// smaller than the agreed shape sends through a device staging buffer of that shape: the collective's shape never
// depends on a local condition.
//
// librccl is opened at the first use (dlopen by SONAME: a process that already holds a copy -- PyTorch bundles one --
// keeps using that one), so single-GPU users of libipcr_hip.so never load it.
#include "ipcr_hip.h"

#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

ipcr_status ipcr_internal_fail(ipcr_status st, const char *fmt, ...);
// host.cpp: what a scratch sends -- its device hit block (header + slots), or, when that block does not hold the scan's
// results (*authoritative == 0), the host list that does
ipcr_status ipcr_internal_scratch_send_block(const ipcr_scratch *s, const void **dev_block, uint64_t *hcap, const ipcr_hit **host_hits,
                                             uint64_t *n_host, int *authoritative);
int ipcr_internal_slot_phys(int slot);
int ipcr_internal_slot_count();

namespace {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

Rccl *rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
        }
        if (!r.handle) { r.error = std::string("librccl not found: ") + (dlerror() ? dlerror() : "?"); return; }
        auto sym = [&](const char *n) { void *p = dlsym(r.handle, n); if (!p && r.error.empty()) r.error = std::string("librccl lacks ") + n; return p; };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    });
    return r.error.empty() ? &r : nullptr;
}

#define XHIP(expr)                                                                                         \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess) return ipcr_internal_fail(IPCR_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)
#define XNCCL(x, expr)                                                                                     \
    do {                                                                                                   \
        ncclResult_t r_ = (expr);                                                                          \
        if (r_ != ncclSuccess) return ipcr_internal_fail(IPCR_ERR_DEVICE, "%s: %s", #expr, (x)->GetErrorString ? (x)->GetErrorString(r_) : "rccl error"); \
    } while (0)

struct OnDevice { // the exchange's device for the calling thread, the previous one put back
    int prev = -1, want;
    explicit OnDevice(int phys) : want(phys) { if (hipGetDevice(&prev) == hipSuccess && prev != want) (void)hipSetDevice(want); else prev = want; }
    ~OnDevice() { if (prev != want) (void)hipSetDevice(prev); }
};

constexpr int SLOTS = 2; // exchanges in flight

} // namespace

struct ipcr_exchange {
    Rccl *lib = nullptr;
    ncclComm_t comm = nullptr;
    // IPCR_TEST_EXCHANGE_FAKE=1 (tests on a one-GPU box): no communicator -- the "all-gather" copies this rank's block into
    // every rank's place of the receive buffer, rank r's header rewritten to hold (hits >> r) records: everything around the
    // collective (shape agreement, strided header read-back, per-rank prefix copies, lock-step redo, rebasing) runs with world > 1
    bool fake = false;
    void *h_fake = nullptr; // pinned: world x 16 bytes, the rewritten header words
    int world = 1, rank = 0, device = 0, phys = 0;
    bool same_records = false;
    hipStream_t stream = nullptr;
    uint64_t cap = 0; // hit slots every rank sends
    void *d_stage = nullptr;           // 64 + cap * 32 bytes: for a scratch buffer smaller than the agreed shape
    void *d_recv[SLOTS] = {nullptr, nullptr}; // world x (64 + cap * 32)
    uint64_t recv_cap[SLOTS] = {0, 0};        // capacity each receive slot is allocated for (grown when the slot is next used)
    uint64_t stage_cap = 0, host_cap = 0;
    void *h_recv = nullptr;            // pinned, same size: where ipcr_exchange_end unpacks from
    void *d_meta = nullptr, *h_meta = nullptr; // record counts (world x 8 bytes)
    hipEvent_t done[SLOTS] = {nullptr, nullptr};
    struct Pending { bool active = false; const ipcr_scratch *scratch = nullptr; uint64_t cap = 0; } pend[SLOTS];
    int next_slot = 0;
    std::vector<uint64_t> rec_counts;  // records per rank (ipcr_exchange_set_records)
    std::vector<ipcr_hit> hits;        // result of the last ipcr_exchange_end
    std::vector<uint64_t> hit_start;   // world + 1: rank r's hits are [hit_start[r], hit_start[r + 1])
    std::vector<uint32_t> rec_offset;  // world + 1
    uint64_t redone = 0;
};

namespace {

size_t block_bytes(uint64_t cap) { return 64u + (size_t)cap * sizeof(ipcr_hit); }

// The capacity every rank sends from now on.  Buffers follow lazily: a receive slot (and the staging / host buffers) is
// regrown when it is next used, so an exchange still in flight in the other slot keeps the buffers it was enqueued with.
ipcr_status set_capacity(ipcr_exchange *x, uint64_t cap) {
    x->cap = cap;
    return IPCR_OK;
}

ipcr_status ensure_slot(ipcr_exchange *x, int slot) {
    const size_t nb = block_bytes(x->cap);
    if (x->recv_cap[slot] < x->cap) {
        if (x->d_recv[slot]) (void)hipFree(x->d_recv[slot]); // (hipFree waits for the device: nothing still reads it)
        x->d_recv[slot] = nullptr;
        XHIP(hipMalloc(&x->d_recv[slot], nb * (size_t)x->world));
        x->recv_cap[slot] = x->cap;
    }
    if (x->stage_cap < x->cap) {
        if (x->d_stage) (void)hipFree(x->d_stage);
        x->d_stage = nullptr;
        XHIP(hipMalloc(&x->d_stage, nb));
        // on the exchange's own stream, in front of the copies that fill the block: hipMemset runs on the null stream and does not
        // wait for the device, x->stream is non-blocking -- the fill could land AFTER the header and records staged below and the
        // rank would report zero hits (seen once in test_hit_cap_bounds_device_memory after a redo had regrown the buffer)
        XHIP(hipMemsetAsync(x->d_stage, 0, nb, x->stream));
        x->stage_cap = x->cap;
    }
    return IPCR_OK;
}

ipcr_status ensure_host(ipcr_exchange *x, uint64_t cap) {
    if (x->host_cap < cap) {
        if (x->h_recv) (void)hipHostFree(x->h_recv);
        x->h_recv = nullptr;
        XHIP(hipHostMalloc(&x->h_recv, block_bytes(cap) * (size_t)x->world, hipHostMallocDefault));
        x->host_cap = cap;
    }
    return IPCR_OK;
}

// enqueue one all-gather of the scratch's device hit block into receive slot `slot`
ipcr_status enqueue(ipcr_exchange *x, const ipcr_scratch *s, int slot) {
    const void *dev_block = nullptr;
    const ipcr_hit *host_hits = nullptr;
    uint64_t n_host = 0, hcap = 0;
    int authoritative = 1;
    ipcr_status st = ipcr_internal_scratch_send_block(s, &dev_block, &hcap, &host_hits, &n_host, &authoritative);
    if (st == IPCR_OK) st = ensure_slot(x, slot);
    if (st != IPCR_OK) return st;
    const size_t nb = block_bytes(x->cap);
    const void *send = dev_block;
    if (!authoritative) {
        // The scan's results live in the HOST list only: a capped scan with far more raw matches than the hit buffer may
        // take ran range of blocks by range of blocks (host.cpp: scan_segmented) and the device buffer holds the last
        // range's records.  The block is rebuilt in the staging buffer -- header with the true count, then the first `cap`
        // records -- and sent from there: same shape, same lock-step overflow handling as any other rank's block.
        uint64_t hdr[8] = {0, n_host, 0, 0, 0, 0, 0, 0};
        const uint64_t n_send = std::min<uint64_t>(n_host, x->cap);
        XHIP(hipMemcpyAsync(x->d_stage, hdr, sizeof hdr, hipMemcpyHostToDevice, x->stream));
        if (n_send) XHIP(hipMemcpyAsync(static_cast<uint8_t *>(x->d_stage) + 64, host_hits, (size_t)n_send * sizeof(ipcr_hit), hipMemcpyHostToDevice, x->stream));
        XHIP(hipStreamSynchronize(x->stream)); // (pageable sources: the list may change once this call has returned)
        send = x->d_stage;
    } else if (hcap < x->cap) { // this rank's buffer is smaller than the agreed shape: the same bytes through the staging buffer
        XHIP(hipMemcpyAsync(x->d_stage, dev_block, block_bytes(hcap), hipMemcpyDeviceToDevice, x->stream));
        send = x->d_stage;
    }
    if (!x->fake) {
        XNCCL(x->lib, x->lib->AllGather(send, x->d_recv[slot], nb, ncclUint8, x->comm, x->stream));
    } else { // tests: see ipcr_exchange::fake
        uint64_t hdr[8];
        XHIP(hipMemcpyAsync(hdr, send, sizeof hdr, hipMemcpyDeviceToHost, x->stream));
        XHIP(hipStreamSynchronize(x->stream));
        const uint64_t n = std::max(hdr[1], hdr[5]);
        uint64_t *hf = static_cast<uint64_t *>(x->h_fake);
        for (int r = 0; r < x->world; ++r) {
            uint8_t *dst = static_cast<uint8_t *>(x->d_recv[slot]) + nb * (size_t)r;
            XHIP(hipMemcpyAsync(dst, send, nb, hipMemcpyDeviceToDevice, x->stream));
            hf[2 * r] = n >> r; // "rank r found half of what rank r - 1 found": uneven counts, zero-hit ranks for small n
            hf[2 * r + 1] = 0;
            XHIP(hipMemcpyAsync(dst + 8, hf + 2 * r, 8, hipMemcpyHostToDevice, x->stream));      // counter set 0: hits
            XHIP(hipMemcpyAsync(dst + 40, hf + 2 * r + 1, 8, hipMemcpyHostToDevice, x->stream)); // counter set 1: zero
        }
        XHIP(hipStreamSynchronize(x->stream)); // (h_fake is rewritten by the next enqueue)
    }
    XHIP(hipEventRecord(x->done[slot], x->stream));
    x->pend[slot].active = true;
    x->pend[slot].scratch = s;
    x->pend[slot].cap = x->cap;
    return IPCR_OK;
}

// the counter set of the last scan is the non-zero one of the header's two
uint64_t header_hits(const uint8_t *block) {
    uint64_t h[8];
    memcpy(h, block, sizeof h);
    return std::max(h[1], h[5]);
}

// ---- the host side of ipcr_exchange_end, free of HIP and RCCL (ipcr_exchange_unpack runs it on the CPU for world > 1) ----
// What every rank derives from the gathered HEADERS alone -- all ranks read the same bytes, so all take the same decision.
struct GatherPlan {
    std::vector<uint64_t> counts;     // hits rank r reported (its true count, also when it exceeds the capacity)
    std::vector<uint64_t> hit_start;  // world + 1: where rank r's records go in the gathered list
    std::vector<uint32_t> rec_offset; // world + 1: what is added to rank r's record indices
    uint64_t need = 0;                // largest count: > cap means every rank regrows and repeats the exchange
};
// headers: rank r's 64-byte header at headers + r * stride
void plan_gather(const uint8_t *headers, size_t stride, int world, const std::vector<uint64_t> &rec_counts, bool same_records, GatherPlan &pl) {
    pl.counts.assign((size_t)world, 0);
    pl.hit_start.assign((size_t)world + 1, 0);
    pl.rec_offset.assign((size_t)world + 1, 0);
    pl.need = 0;
    for (int r = 0; r < world; ++r) {
        pl.counts[(size_t)r] = header_hits(headers + stride * (size_t)r);
        pl.need = std::max(pl.need, pl.counts[(size_t)r]);
    }
    for (int r = 0; r < world; ++r) {
        pl.hit_start[(size_t)r + 1] = pl.hit_start[(size_t)r] + pl.counts[(size_t)r];
        pl.rec_offset[(size_t)r + 1] = pl.rec_offset[(size_t)r] + (same_records ? 0u : (uint32_t)rec_counts[(size_t)r]);
    }
}
// byte offset of rank r's first record inside a gathered buffer of `world` blocks of block_bytes(cap)
size_t records_offset(uint64_t cap, int r) { return block_bytes(cap) * (size_t)r + 64u; }
// record indices -> job-global (rank r's records start at rec_offset[r])
void rebase_records(ipcr_hit *hits, const GatherPlan &pl, int world) {
    for (int r = 0; r < world; ++r)
        for (uint64_t i = pl.hit_start[(size_t)r]; i < pl.hit_start[(size_t)r + 1]; ++i) hits[(size_t)i].record += pl.rec_offset[(size_t)r];
}

} // namespace

extern "C" {

static bool fake_transport() {
    const char *v = getenv("IPCR_TEST_EXCHANGE_FAKE");
    return v && *v && atoi(v) != 0;
}

// Everything ipcr_exchange_create can fail on BEFORE it enters ncclCommInitRank, checked locally and without side effects
// on the job: librccl opens and has the entry points, the device exists.  A host takes the minimum of this over all ranks
// first (its own channel: an all-reduce, a barrier with a flag) and calls ipcr_exchange_create only when it is 1 on every
// rank -- a rank that failed here while the others were already inside ncclCommInitRank would leave them waiting for ever.
int32_t ipcr_exchange_available(int32_t device) {
    if (device < 0 || device >= ipcr_internal_slot_count()) {
        (void)ipcr_internal_fail(IPCR_ERR_INVALID, "ipcr_exchange_available: device %d of %d", device, ipcr_internal_slot_count());
        return 0;
    }
    if (fake_transport()) return 1;
    if (!rccl()) { (void)ipcr_internal_fail(IPCR_ERR_DEVICE, "RCCL is not available"); return 0; }
    return 1;
}

ipcr_status ipcr_exchange_unique_id(uint8_t *id_out) {
    if (!id_out) return ipcr_internal_fail(IPCR_ERR_INVALID, "ipcr_exchange_unique_id: null argument");
    if (fake_transport()) { memset(id_out, 0, IPCR_EXCHANGE_ID_BYTES); return IPCR_OK; }
    Rccl *r = rccl();
    if (!r) return ipcr_internal_fail(IPCR_ERR_DEVICE, "RCCL is not available");
    ncclUniqueId id;
    XNCCL(r, r->GetUniqueId(&id));
    static_assert(sizeof id == IPCR_EXCHANGE_ID_BYTES, "id size");
    memcpy(id_out, &id, sizeof id);
    return IPCR_OK;
}

ipcr_status ipcr_exchange_create(const uint8_t *id, int32_t world, int32_t rank, int32_t device, uint64_t cap_hits,
                                 int32_t same_records, ipcr_exchange **out) {
    if (!id || !out || world < 1 || rank < 0 || rank >= world) return ipcr_internal_fail(IPCR_ERR_INVALID, "ipcr_exchange_create: bad argument");
    *out = nullptr;
    if (device < 0 || device >= ipcr_internal_slot_count()) return ipcr_internal_fail(IPCR_ERR_INVALID, "ipcr_exchange_create: device %d of %d", device, ipcr_internal_slot_count());
    const bool fake = fake_transport();
    Rccl *r = fake ? nullptr : rccl();
    if (!r && !fake) return ipcr_internal_fail(IPCR_ERR_DEVICE, "RCCL is not available");
    // everything that can fail locally comes BEFORE the communicator: once a rank is inside ncclCommInitRank the others must follow
    ipcr_exchange *x = new (std::nothrow) ipcr_exchange;
    if (!x) return ipcr_internal_fail(IPCR_ERR_DEVICE, "out of memory");
    x->lib = r;
    x->fake = fake;
    x->world = world;
    x->rank = rank;
    x->device = device;
    x->phys = ipcr_internal_slot_phys(device);
    x->same_records = same_records != 0;
    x->rec_counts.assign((size_t)world, 0);
    OnDevice on(x->phys);
    auto build = [&]() -> ipcr_status {
        XHIP(hipStreamCreateWithFlags(&x->stream, hipStreamNonBlocking));
        for (int i = 0; i < SLOTS; ++i) XHIP(hipEventCreateWithFlags(&x->done[i], hipEventDisableTiming));
        XHIP(hipMalloc(&x->d_meta, 8u * (size_t)(world + 1)));
        XHIP(hipHostMalloc(&x->h_meta, 8u * (size_t)(world + 1), hipHostMallocDefault));
        if (fake) XHIP(hipHostMalloc(&x->h_fake, 16u * (size_t)world, hipHostMallocDefault));
        const ipcr_status cs = set_capacity(x, std::max<uint64_t>(cap_hits, 1));
        if (cs != IPCR_OK || fake) return cs;
        ncclUniqueId uid;
        memcpy(&uid, id, sizeof uid);
        XNCCL(r, r->CommInitRank(&x->comm, world, uid, rank)); // synchronises with the other ranks: the last thing that can fail
        return IPCR_OK;
    };
    const ipcr_status st = build();
    if (st != IPCR_OK) { ipcr_exchange_destroy(x); return st; }
    *out = x;
    return IPCR_OK;
}

void ipcr_exchange_destroy(ipcr_exchange *x) {
    if (!x) return;
    OnDevice on(x->phys);
    if (x->stream) (void)hipStreamSynchronize(x->stream);
    if (x->comm && x->lib) (void)x->lib->CommDestroy(x->comm);
    for (int i = 0; i < SLOTS; ++i) {
        if (x->d_recv[i]) (void)hipFree(x->d_recv[i]);
        if (x->done[i]) (void)hipEventDestroy(x->done[i]);
    }
    if (x->d_stage) (void)hipFree(x->d_stage);
    if (x->h_recv) (void)hipHostFree(x->h_recv);
    if (x->d_meta) (void)hipFree(x->d_meta);
    if (x->h_meta) (void)hipHostFree(x->h_meta);
    if (x->h_fake) (void)hipHostFree(x->h_fake);
    if (x->stream) (void)hipStreamDestroy(x->stream);
    delete x;
}

// records per rank (static for a job): one small synchronous all-gather, called by every rank together
ipcr_status ipcr_exchange_set_records(ipcr_exchange *x, uint32_t n_local_records) {
    if (!x) return ipcr_internal_fail(IPCR_ERR_INVALID, "null exchange");
    OnDevice on(x->phys);
    uint64_t *hm = static_cast<uint64_t *>(x->h_meta);
    hm[x->world] = n_local_records;
    uint64_t *dm = static_cast<uint64_t *>(x->d_meta);
    XHIP(hipMemcpyAsync(dm + x->world, hm + x->world, 8, hipMemcpyHostToDevice, x->stream));
    if (x->fake) { for (int r = 0; r < x->world; ++r) XHIP(hipMemcpyAsync(dm + r, dm + x->world, 8, hipMemcpyDeviceToDevice, x->stream)); }
    else XNCCL(x->lib, x->lib->AllGather(dm + x->world, dm, 8, ncclUint8, x->comm, x->stream));
    XHIP(hipMemcpyAsync(hm, dm, 8u * (size_t)x->world, hipMemcpyDeviceToHost, x->stream));
    XHIP(hipStreamSynchronize(x->stream));
    for (int r = 0; r < x->world; ++r) x->rec_counts[(size_t)r] = hm[r];
    return IPCR_OK;
}

// the same, when the host already knows every rank's count (no collective)
ipcr_status ipcr_exchange_set_record_counts(ipcr_exchange *x, const uint32_t *counts) {
    if (!x || !counts) return ipcr_internal_fail(IPCR_ERR_INVALID, "ipcr_exchange_set_record_counts: null argument");
    for (int r = 0; r < x->world; ++r) x->rec_counts[(size_t)r] = counts[r];
    return IPCR_OK;
}

ipcr_status ipcr_exchange_begin(ipcr_exchange *x, const ipcr_scratch *s, int32_t *ticket) {
    if (!x || !s || !ticket) return ipcr_internal_fail(IPCR_ERR_INVALID, "ipcr_exchange_begin: null argument");
    if (ipcr_scratch_device(s) != x->device) return ipcr_internal_fail(IPCR_ERR_INVALID, "ipcr_exchange_begin: the scratch lives on device %d, the exchange on %d", ipcr_scratch_device(s), x->device);
    const int slot = x->next_slot;
    if (x->pend[slot].active) return ipcr_internal_fail(IPCR_ERR_INVALID, "ipcr_exchange_begin: %d exchanges are in flight already (ipcr_exchange_end first)", SLOTS);
    OnDevice on(x->phys);
    const ipcr_status st = enqueue(x, s, slot);
    if (st != IPCR_OK) return st;
    x->next_slot = (slot + 1) % SLOTS;
    *ticket = slot;
    return IPCR_OK;
}

ipcr_status ipcr_exchange_end(ipcr_exchange *x, int32_t ticket, const ipcr_hit **hits, int64_t *n_hits,
                              const uint64_t **rank_hit_start, const uint32_t **rank_record_offset) {
    if (!x || ticket < 0 || ticket >= SLOTS || !x->pend[ticket].active) return ipcr_internal_fail(IPCR_ERR_INVALID, "ipcr_exchange_end: no such exchange in flight");
    OnDevice on(x->phys);
    ipcr_exchange::Pending pd = x->pend[ticket];
    x->pend[ticket].active = false;
    for (int attempt = 0; attempt < 24; ++attempt) {
        XHIP(hipEventSynchronize(x->done[ticket]));
        const size_t nb = block_bytes(pd.cap);
        { const ipcr_status hs = ensure_host(x, pd.cap); if (hs != IPCR_OK) return hs; }
        // every rank's header first: all ranks read the same counts and take the same decision
        XHIP(hipMemcpy2DAsync(x->h_recv, 64, x->d_recv[ticket], nb, 64, (size_t)x->world, hipMemcpyDeviceToHost, x->stream));
        XHIP(hipStreamSynchronize(x->stream));
        GatherPlan pl;
        plan_gather(static_cast<const uint8_t *>(x->h_recv), 64, x->world, x->rec_counts, x->same_records, pl);
        if (pl.need > pd.cap) { // some rank overflowed: every rank regrows and repeats the exchange (the scratch still holds the scan)
            ++x->redone;
            uint64_t cap = std::max(pd.cap, x->cap);
            while (cap < pl.need) cap *= 2;
            ipcr_status st = set_capacity(x, cap); // (an exchange in flight in the other slot keeps its own buffers and shape)
            if (st != IPCR_OK) return st;
            st = enqueue(x, pd.scratch, ticket);
            if (st != IPCR_OK) return st;
            pd = x->pend[ticket];
            x->pend[ticket].active = false;
            continue;
        }
        // the records: only the valid prefix of every rank's block
        x->hit_start = pl.hit_start;
        x->rec_offset = pl.rec_offset;
        x->hits.resize((size_t)pl.hit_start[(size_t)x->world]);
        for (int r = 0; r < x->world; ++r)
            if (pl.counts[(size_t)r])
                XHIP(hipMemcpyAsync(x->hits.data() + pl.hit_start[(size_t)r], static_cast<const uint8_t *>(x->d_recv[ticket]) + records_offset(pd.cap, r),
                                    (size_t)pl.counts[(size_t)r] * sizeof(ipcr_hit), hipMemcpyDeviceToHost, x->stream));
        XHIP(hipStreamSynchronize(x->stream));
        rebase_records(x->hits.data(), pl, x->world); // job-global record index
        if (hits) *hits = x->hits.data();
        if (n_hits) *n_hits = (int64_t)x->hits.size();
        if (rank_hit_start) *rank_hit_start = x->hit_start.data();
        if (rank_record_offset) *rank_record_offset = x->rec_offset.data();
        return IPCR_OK;
    }
    return ipcr_internal_fail(IPCR_ERR_CAPACITY, "ipcr_exchange_end: the exchange kept overflowing");
}

// The host side of ipcr_exchange_end over a gathered buffer in HOST memory: `world` blocks of 64 + cap * 32 bytes, as
// ncclAllGather leaves them.  Pure function (no device, no communicator): what the tests drive with synthetic buffers for
// world = 2, 3, 8, and what a host that runs the collective itself (MPI, its own RCCL communicator) can use.  *need = the
// largest count any rank reported; IPCR_ERR_CAPACITY when it exceeds cap (every rank sees the same headers, so every rank
// gets this status and repeats the exchange with a capacity >= *need) or when the records do not fit out_cap.
ipcr_status ipcr_exchange_unpack(const void *gathered, int32_t world, uint64_t cap, const uint32_t *rec_counts, int32_t same_records,
                                 ipcr_hit *out, uint64_t out_cap, uint64_t *rank_hit_start, uint32_t *rank_record_offset, uint64_t *need) {
    if (!gathered || world < 1 || (!rec_counts && !same_records)) return ipcr_internal_fail(IPCR_ERR_INVALID, "ipcr_exchange_unpack: bad argument");
    std::vector<uint64_t> rc((size_t)world, 0);
    for (int r = 0; r < world && rec_counts; ++r) rc[(size_t)r] = rec_counts[r];
    GatherPlan pl;
    plan_gather(static_cast<const uint8_t *>(gathered), block_bytes(cap), world, rc, same_records != 0, pl);
    if (need) *need = pl.need;
    if (rank_hit_start) memcpy(rank_hit_start, pl.hit_start.data(), ((size_t)world + 1) * sizeof(uint64_t));
    if (rank_record_offset) memcpy(rank_record_offset, pl.rec_offset.data(), ((size_t)world + 1) * sizeof(uint32_t));
    if (pl.need > cap) return ipcr_internal_fail(IPCR_ERR_CAPACITY, "a rank reported %llu hits, the exchange carries %llu: repeat it with a larger capacity on every rank",
                                                 (unsigned long long)pl.need, (unsigned long long)cap);
    const uint64_t total = pl.hit_start[(size_t)world];
    if (total > out_cap || (total && !out)) return ipcr_internal_fail(IPCR_ERR_CAPACITY, "%llu gathered hits do not fit the output (%llu)", (unsigned long long)total, (unsigned long long)out_cap);
    for (int r = 0; r < world; ++r)
        if (pl.counts[(size_t)r])
            memcpy(out + pl.hit_start[(size_t)r], static_cast<const uint8_t *>(gathered) + records_offset(cap, r), (size_t)pl.counts[(size_t)r] * sizeof(ipcr_hit));
    rebase_records(out, pl, world);
    return IPCR_OK;
}

// a capacity every rank already knows it needs (e.g. from a first, synchronous exchange): the same value on every rank
ipcr_status ipcr_exchange_reserve(ipcr_exchange *x, uint64_t cap_hits) {
    if (!x) return ipcr_internal_fail(IPCR_ERR_INVALID, "null exchange");
    if (cap_hits > x->cap) x->cap = cap_hits;
    return IPCR_OK;
}

uint64_t ipcr_exchange_capacity(const ipcr_exchange *x) { return x ? x->cap : 0; }
uint64_t ipcr_exchange_redone(const ipcr_exchange *x) { return x ? x->redone : 0; }

} // extern "C"
