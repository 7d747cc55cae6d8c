// host.cpp -- host runtime behind the C ABI (include/ipcr_hip.h).
//
// Mirrors, for the scan path only, the reference's engine package:
//   ipcr_panel_create   = engine.New + Engine.CompilePanel     (core/engine/compiled.go:96-136)
//   ipcr_scratch_create = Engine.NewSimulationScratch          (core/engine/hit_collect.go:21-34)
//   ipcr_scan_chunk     = Engine.ForEachCompiledProduct        (core/engine/compiled.go:162-267)
//   join                = Engine.forEachJoinedProduct          (core/engine/engine.go:108-404)
// The scan itself (reference rows: AC seed scan, halo rescue, verifyAt, FindMatches fallback)
// is replaced by the bit-sliced device filter + per-candidate verifier; this file turns the
// verified hits back into exactly the per-orientation match lists the reference would hold
// (ordering and HitCap rules included) and joins them.
#include "ipcr_hip.h"

#include <emmintrin.h>
#include <immintrin.h>
#include <sched.h>
#include <sys/prctl.h>
#include <time.h>
#include <unistd.h>
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "cpu_pool.h"
#include "device_types.h"
#include "fasta_hostpack.h"
#include "jit.h"
#include "launch.h"
#include "tile_layout.h"

using ipcr::PackPool;          // cpu_pool.h: the process's pool of pack threads
using ipcr::bind_this_thread;  // ... and the CPUs next to a device

static_assert(sizeof(ipcr_hit) == sizeof(ipcr_hit_rec), "hit layouts must agree");
static_assert(sizeof(ipcr_probe_hit) == sizeof(ipcr_probe_rec), "probe layouts must agree");

namespace {

thread_local std::string g_err;

ipcr_status fail(ipcr_status st, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return st;
}

} // namespace
ipcr_status ipcr_internal_fail(ipcr_status st, const char *fmt, ...) { // for the other translation units
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return st;
}
namespace {

// ------------------------------------------------------------ devices
// One host process may drive every GPU of a node: each scratch / genome belongs to a DEVICE SLOT, the panel keeps one set of
// device tables and code objects per slot, and every entry point selects its object's device itself -- HIP's current
// device is a per-thread setting, and a Go runtime moves goroutines between threads (internal/pipeline/pipeline.go:60-125:
// worker i -> slot i mod N, no collective).  Slot d is physical device d; IPCR_DEVICE_SLOTS=N (tests, rehearsals on a
// one-GPU box) adds slots beyond the physical devices, mapped onto them round-robin, each with tables of its own.
std::atomic<int> g_default_slot{-1}; // ipcr_set_device; -1: the calling thread's current HIP device

int phys_count() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
int slot_count() {
    const int n = phys_count();
    if (n <= 0) return 0;
    const char *v = getenv("IPCR_DEVICE_SLOTS");
    const int extra = (v && *v) ? atoi(v) : 0;
    return std::max(n, std::min(extra, 64));
}
int slot_phys(int slot) {
    const int n = phys_count();
    return n > 0 ? slot % n : 0;
}
int default_slot() {
    const int d = g_default_slot.load(std::memory_order_relaxed);
    if (d >= 0) return d;
    int cur = 0;
    return hipGetDevice(&cur) == hipSuccess ? cur : 0;
}
// selects a slot's device for the calling thread, and puts back what was there
struct DeviceGuard {
    int prev = -1;
    bool changed = false;
    explicit DeviceGuard(int slot) {
        const int phys = slot_phys(slot);
        if (hipGetDevice(&prev) == hipSuccess && prev != phys) changed = hipSetDevice(phys) == hipSuccess;
    }
    ~DeviceGuard() { if (changed) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

#define HIPCHK(expr)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) return fail(IPCR_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

bool env_flag(const char *name, bool dflt);

// ------------------------------------------------------------ the host's way into device memory
// Large BAR: the whole of the device's memory is mapped into the host's address space, and the CPU's write-combining
// stores reach it at ~45 GB/s (tools/exp/bar_write.cpp: 80 % of the link's DMA rate, from one thread or sixteen) -- the
// host's packer then writes its planes where the device reads them and no copy operation is queued at all.
// IPCR_CHUNK_BAR=0: pinned slabs + DMA (round 3's path; also what runs without a large BAR).
// What the CPU writes through the BAR passes the device's HDP block (host data path), which may hold it back: the runtime
// publishes the register that flushes it (HSA_AMD_AGENT_INFO_HDP_FLUSH) and uses it itself for the kernel arguments it keeps in
// device memory.  The library writes that register -- and reads it back, so that the write has arrived -- after the packer's
// stores and before the launch that reads them; a device whose register cannot be found does not take the BAR path.
// (The HSA runtime is the one the process has loaded already -- HIP sits on it -- found by dlopen(RTLD_NOLOAD): no link dependency.)
volatile uint32_t *find_hdp_flush_register(int phys) {
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, phys) != hipSuccess) return nullptr;
    unsigned dom = 0, b = 0, d = 0, f = 0;
    if (sscanf(bus, "%x:%x:%x.%x", &dom, &b, &d, &f) != 4) return nullptr;
    void *h = dlopen("libhsa-runtime64.so.1", RTLD_NOW | RTLD_NOLOAD);
    auto sym = [&](const char *n) { void *p = h ? dlsym(h, n) : nullptr; return p ? p : dlsym(RTLD_DEFAULT, n); };
    using iterate_t = hsa_status_t (*)(hsa_status_t (*)(hsa_agent_t, void *), void *);
    using info_t = hsa_status_t (*)(hsa_agent_t, hsa_agent_info_t, void *);
    struct Ctx { info_t info; uint32_t dom, bdf; volatile uint32_t *reg; } c{reinterpret_cast<info_t>(sym("hsa_agent_get_info")), dom, (b << 8) | (d << 3) | f, nullptr};
    const iterate_t iterate = reinterpret_cast<iterate_t>(sym("hsa_iterate_agents"));
    if (!iterate || !c.info) return nullptr;
    (void)iterate([](hsa_agent_t a, void *vp) -> hsa_status_t {
        Ctx &c = *static_cast<Ctx *>(vp);
        hsa_device_type_t type;
        uint32_t bdf = 0, dom = 0;
        if (c.info(a, HSA_AGENT_INFO_DEVICE, &type) != HSA_STATUS_SUCCESS || type != HSA_DEVICE_TYPE_GPU) return HSA_STATUS_SUCCESS;
        if (c.info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_BDFID, &bdf) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
        if (c.info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_DOMAIN, &dom) != HSA_STATUS_SUCCESS) dom = c.dom;
        if ((bdf & 0xFFFFu) != c.bdf || dom != c.dom) return HSA_STATUS_SUCCESS;
        hsa_amd_hdp_flush_t hdp{nullptr, nullptr};
        if (c.info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_HDP_FLUSH, &hdp) == HSA_STATUS_SUCCESS) c.reg = hdp.HDP_MEM_FLUSH_CNTL;
        return HSA_STATUS_INFO_BREAK;
    }, &c);
    return c.reg;
}

struct BarInfo { bool writable = false; volatile uint32_t *hdp_flush = nullptr; };
std::mutex g_bar_mu;
std::map<int, BarInfo> g_bar_cache;
BarInfo device_bar(int phys) {
    std::lock_guard<std::mutex> lk(g_bar_mu);
    auto it = g_bar_cache.find(phys);
    if (it != g_bar_cache.end()) return it->second;
    BarInfo bi;
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeIsLargeBar, phys) == hipSuccess && v != 0) {
        bi.hdp_flush = find_hdp_flush_register(phys);
        // IPCR_CHUNK_BAR=2: take the BAR path even where the flush register is not known (the runtime's own flush in front of every
        // dispatch then stands in for it: measurements only)
        bi.writable = bi.hdp_flush != nullptr || (getenv("IPCR_CHUNK_BAR") && atoi(getenv("IPCR_CHUNK_BAR")) == 2);
    }
    g_bar_cache[phys] = bi;
    return bi;
}
// host memory -> device memory through the BAR, in the widest non-temporal stores the CPU has (64-byte write-combining bursts reach
// ~45 GB/s from one thread, 8-byte ones 13: tools/exp/bar_write.cpp); both 64-byte aligned, bytes a multiple of 64
__attribute__((target("avx512f"))) static void bar_copy_512(uint8_t *dst, const uint8_t *src, uint64_t bytes) {
    for (uint64_t i = 0; i < bytes; i += 64) _mm512_stream_si512(reinterpret_cast<__m512i *>(dst + i), _mm512_load_si512(reinterpret_cast<const void *>(src + i)));
    _mm_sfence();
}
static void bar_copy(uint8_t *dst, const uint8_t *src, uint64_t bytes) {
    static const bool wide = __builtin_cpu_supports("avx512f");
    if (wide && ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src) | bytes) & 63u) == 0) { bar_copy_512(dst, src, bytes); return; }
    for (uint64_t i = 0; i + 16 <= bytes; i += 16) _mm_stream_si128(reinterpret_cast<__m128i *>(dst + i), _mm_loadu_si128(reinterpret_cast<const __m128i *>(src + i)));
    _mm_sfence();
}
// everything the packer's threads have stored through the BAR (each of them has fenced) is in device memory when this returns
inline void bar_flush(const BarInfo &bi) {
    if (!bi.hdp_flush) return;
    _mm_sfence();
    *bi.hdp_flush = 1u;
    (void)*bi.hdp_flush; // a read does not pass the writes in front of it: the flush has been taken
}
void host_writable_off(int phys) {
    std::lock_guard<std::mutex> lk(g_bar_mu);
    g_bar_cache[phys].writable = false;
}


// ------------------------------------------------------------ core/primer tables
struct Tables {
    uint8_t mask[256];
    uint8_t comp[256];
    Tables() {
        memset(mask, 0, sizeof mask);
        memset(comp, 0, sizeof comp);
        const char *codes = "ACGTURYSWKMBDHVN"; // core/primer/iupac.go:19-39
        const uint8_t bits[] = {1, 2, 4, 8, 8, 5, 10, 6, 9, 12, 3, 14, 13, 11, 7, 15};
        for (int i = 0; codes[i]; ++i) {
            mask[(uint8_t)codes[i]] = bits[i];
            mask[(uint8_t)(codes[i] | 0x20)] = bits[i]; // :41-57
        }
        const char *from = "ACGTRYSWKMBVDHN", *to = "TGCAYRSWMKVBHDN"; // core/primer/rc.go:8-24
        for (int i = 0; from[i]; ++i) comp[(uint8_t)from[i]] = (uint8_t)to[i];
    }
};
const Tables T;

bool revcomp(const std::string &in, std::string &out, size_t *bad) {
    out.resize(in.size());
    for (size_t i = 0; i < in.size(); ++i) {
        size_t src = in.size() - 1 - i;
        uint8_t c = T.comp[(uint8_t)in[src]];
        if (!c) { if (bad) *bad = src + 1; return false; }
        out[i] = (char)c;
    }
    return true;
}

int popcount4(uint8_t m) { return __builtin_popcount(m & 15u); }

// --------------------------------------------------- seeded-or-not (core/engine/seed.go)
// The device scan needs no seeds, but whether the reference WOULD have seeded an orientation
// decides which of its two code paths (collector vs FindMatches fallback) produced the match
// list, and those differ in hit-cap and ordering details (compiled.go:185-258).
int configured_seed_len(int plen, int cfg) { // seed.go:104-115
    if (cfg <= 0) return plen < 12 ? plen : 12;
    return cfg > plen ? plen : cfg;
}

struct SeedSpan { bool seeded; int off, len; };

SeedSpan reference_seed_span(const std::string &pat, int seed_len_cfg, bool prefer_right, int left_tw,
                             int right_tw, int max_mm) {
    SeedSpan r{false, 0, 0};
    const int plen = (int)pat.size();
    if (seed_len_cfg < 0 || plen == 0) return r; // seed.go:156-159,175-177
    const int len = configured_seed_len(plen, seed_len_cfg);
    if (len <= 0 || len > plen) return r;
    int best_off = 0;
    uint64_t best_score = 0;
    for (int off = 0; off + len <= plen; ++off) { // chooseSeedSpan seed.go:260-283
        uint64_t score = 1;
        for (int i = 0; i < len; ++i) {
            uint64_t n = (uint64_t)popcount4(T.mask[(uint8_t)pat[off + i]]);
            if (n == 0) n = 5;
            score *= n;
        }
        const bool tie = prefer_right ? off > best_off : off < best_off;
        if (off == 0 || score < best_score || (score == best_score && tie)) { best_off = off; best_score = score; }
    }
    r.off = best_off;
    r.len = len;
    // count enumerateSeedVariants' output (seed.go:304-367) without enumerating: strings within
    // <= max_mm mismatches, none at protected positions
    if (max_mm < 0) max_mm = 0;
    if (left_tw < 0) left_tw = 0;
    if (right_tw < 0) right_tw = 0;
    const uint64_t CAP = 1000000000ull;
    std::vector<uint64_t> f((size_t)max_mm + 1, 0), g((size_t)max_mm + 1, 0);
    f[0] = 1;
    for (int p = 0; p < len; ++p) {
        const int full = best_off + p;
        const bool prot = (left_tw > 0 && full < left_tw) || (right_tw > 0 && full >= plen - right_tw);
        const uint64_t nm = (uint64_t)popcount4(T.mask[(uint8_t)pat[full]]);
        const uint64_t nx = 4 - nm;
        for (int m = 0; m <= max_mm; ++m) {
            uint64_t v = f[(size_t)m] * nm;
            if (!prot && m > 0) v += f[(size_t)m - 1] * nx;
            g[(size_t)m] = v > CAP ? CAP : v;
        }
        f.swap(g);
    }
    uint64_t total = 0;
    for (int m = 0; m <= max_mm; ++m) total += f[(size_t)m];
    // > 50000 variants, none at all, or a seed the 2-bit key cannot hold (> 32 nt) -> unseeded
    r.seeded = total > 0 && total <= 50000 && len <= 32; // seed.go:97,188-207
    return r;
}

} // namespace

// ----------------------------------------------------------------------------- panel

struct PatternDef {
    std::string seq;
    bool left;   // protected window on the left (rc orientations 'a','b'), else right ('A','B')
    int tw_dev;  // protected length enforced on the device (0 = host filters)
    int seed_off, seed_len;
    bool seeded;
};

// seed index over the keyable patterns of a set (large panels; jit.cpp: jit_index_source)
struct IndexPlan {
    bool built = false, usable = false;
    bool all_acgt = true; // no keyed pattern holds an IUPAC code: entries are checked from their first 32 bytes
    int dl = 0;        // left-anchored windows are tested dl bases after their start
    int max_right = 0; // longest right-anchored pattern served
    int uniform_len = 0; // length shared by every served pattern (0: mixed)
    ipcr::IndexGeom geom() const {
        ipcr::IndexGeom g;
        g.tail_rows = std::max(max_right - 1, dl);
        g.all_acgt = all_acgt;
        g.uniform_len = uniform_len;
        g.dl = dl;
        g.table_entries = (uint32_t)table.size();
        return g;
    }
    std::vector<ipcr_index_shape> shapes;
    // the shapes' bitmaps (one bit per key, 2^(key bits) bits each), then for every 64-bit bitmap word the rank of its
    // first key (uint16, relative to the shape's first entry), then the shapes' first entries and constants
    std::vector<uint32_t> lds_image;
    std::vector<ipcr_index_entry> table; // entry r = first pattern of the r-th distinct (shape, key); more patterns of a key are chained
    std::vector<uint32_t> leftover;      // set-local patterns the index cannot serve (> 32 nt, too degenerate, ...)
};

// what a pattern set holds on ONE device slot: tables and the code objects loaded there (hiprtc output is cached per
// process by source, so a second slot pays the module load only)
struct SetDev {
    ipcr_dev_pattern *dev = nullptr;
    std::vector<ipcr::JitFilter *> jit; // one kernel per pattern group; written once, then published through jit_pub
    // what a scan uses: null until the kernels exist (a scan of a small panel does not wait for hiprtc: panel_upload)
    std::atomic<const std::vector<ipcr::JitFilter *> *> jit_pub{nullptr};
    std::thread jit_thread;     // builds `jit` in the background
    std::atomic<int> jit_state{0}; // 0 nothing in flight, 1 building, 2 built (to be published by the next panel_upload)
    bool jit_tried = false;
    bool index_tried = false;
    uint32_t *d_lds_image = nullptr;
    ipcr_index_entry *d_table = nullptr;
    uint32_t *d_leftover = nullptr;
    ipcr::JitFilter *index_jit = nullptr; // the seed-index kernel, with the key shapes baked in (hiprtc)
    std::vector<ipcr::JitFilter *> leftover_jit; // specialised spill-only filters for the index's leftovers (else the table-driven kernel takes them)
};
struct PanelDev {
    int slot = 0;
    SetDev set[2];
};

struct PatternSet {
    std::vector<uint32_t> ids; // global pattern ids scanned in this mode
    IndexPlan index;
    std::vector<ipcr_dev_pattern> host;
    std::string jit_error;
};

struct ipcr_panel {
    ipcr_config cfg{};
    int tw = 0; // effective (negative -> 0)
    std::vector<std::string> id, fwd, rev;
    std::vector<int32_t> minp, maxp;
    std::vector<std::array<std::string, 4>> ori; // A, B, rc(A), rc(B)  compiled.go:108-117
    std::vector<uint8_t> have;                   // orientationMask compiled.go:6-37
    std::vector<PatternDef> defs;                // global pattern table
    std::vector<std::array<std::array<uint32_t, 2>, 4>> slot; // [pair][which][mode] -> global id
    std::vector<std::vector<uint32_t>> users[2]; // [mode][global pattern] -> pairs scanning it (ascending)
    PatternSet set[2];                           // mode 0: no record holds a reset byte; mode 1: some do
    bool modes_equal = true;
    int max_len = 0;
    bool specialize = true;
    int32_t shard_index = 0, shard_count = 1; // ipcr_panel_set_shard: this panel object scans every count-th pattern
    mutable std::mutex mu;
    mutable std::map<int, std::unique_ptr<PanelDev>> devs; // device slot -> tables and kernels there (created at the first scan on the slot)
    // device scratches alive = workers scanning with this panel (shared: a scratch may outlive its panel object)
    std::shared_ptr<std::atomic<int>> live_scratches = std::make_shared<std::atomic<int>>(0);
};

namespace {

bool env_flag(const char *name, bool dflt);

// Pigeonhole keys for the seed-index filter.  Patterns are grouped by (anchored end, protected
// length t); a group with shortest pattern Lmin uses blocks of bf = (Lmin - t) / (k+1) bases laid
// next to the protected bases, so any window with <= k mismatches (none protected) has the
// protected bases and at least one whole block exact.  A key = protected part + (a prefix of) one
// block, at most 8 bases = 16 bits -- 17 where a spare base next to a five-base block can lend one bit
// (device_types.h) and the bitmaps still leave the waves their hit queues.
void build_index(const ipcr_panel &p, PatternSet &set) {
    IndexPlan &ix = set.index;
    ix.built = true;
    const int k = p.cfg.max_mm;
    const size_t P = set.host.size();
    // nblk: blocks the bases behind the first t are cut into (k + 1, or k: see "split" below); noprot: the keys hold no protected base
    struct Group { bool left; int t; int lmin; std::vector<uint32_t> members; int nblk; bool noprot; };
    std::vector<Group> groups;
    struct Pat { uint64_t ok[4]; uint64_t prot2; int len; bool left; };
    std::vector<Pat> pats(P);
    for (uint32_t q = 0; q < P; ++q) {
        const ipcr_dev_pattern &dp = set.host[q];
        const PatternDef &d = p.defs[set.ids[q]];
        const int L = dp.len;
        bool usable = L >= 1 && L <= 32;
        Pat &pt = pats[q];
        memset(&pt, 0, sizeof pt);
        for (int j = 0; j < L && usable; ++j) {
            const uint8_t m = dp.mask[j] & 15u;
            if (m == 0) { usable = false; break; } // matches nothing: leave it to the table-driven kernel
            for (int b = 0; b < 4; ++b)
                if (m & (1u << b)) pt.ok[b] |= 1ull << (2 * (L - 1 - j));
            if (dp.mask[j] & 16u) pt.prot2 |= 1ull << (2 * (L - 1 - j));
        }
        if (!usable) { ix.leftover.push_back(q); continue; }
        pt.len = L;
        pt.left = d.left;
        int t = k == 0 ? L : std::min(d.tw_dev, L);
        if (t < 0) t = 0;
        auto join = [&](int gt, int nblk, bool noprot) {
            for (Group &g : groups)
                if (g.left == d.left && g.t == gt && g.nblk == nblk && g.noprot == noprot) { g.members.push_back(q); g.lmin = std::min(g.lmin, L); return; }
            groups.push_back(Group{d.left, gt, L, {q}, nblk, noprot});
        };
        // SPLIT.  An rc orientation that the device scans WITHOUT its 5' window (the reference caps it before the window
        // filter: core/engine/compiled.go:249-256 -- every chunk, and every record of a genome that holds an N) has nothing
        // protected to key on: 20-mers at k = 2 would go under three keys of 6-7 bases, 2048 patterns in 2^12..2^14 slots, and
        // a third of all lane steps would hit (measured: 33 ms per 3 Gb against 4.1).  But the window is still there in the
        // panel, and a match either has its first tw bases exact -- then <= k mismatches lie behind them: "3 window bases +
        // one of k + 1 blocks", the protected orientation's own 16-17-bit keys -- or it has a mismatch among them -- then
        // <= k - 1 lie behind: one of only k blocks, twice as long (8 bases at k = 2), is exact.  The pattern is filed under
        // both families; the exact check accepts any window with <= k mismatches, so the union is exactly the raw matches.
        const bool split_on = env_flag("IPCR_INDEX_SPLIT", true); // (read per panel: the tests compare the two in one process)
        const int twp = std::min(p.tw, L);
        if (split_on && d.left && d.tw_dev == 0 && k >= 1 && twp >= 1 && L - twp >= k + 1) {
            join(std::min(twp, 32), k + 1, false);
            join(std::min(twp, 32), k, true);
        } else
            join(std::min(t, 32), k + 1, false);
    }
    std::vector<std::pair<uint32_t, uint32_t>> ents; // (tag, pattern)
    std::vector<char> dropped(P, 0);
    // left-anchored windows are tested when their start lies DL bases behind the newest base: the longest left
    // pattern then ends exactly at the newest base (31 for a 32-nt pattern)
    int DL = 0;
    for (const Group &g : groups)
        if (g.left)
            for (uint32_t q : g.members) DL = std::max(DL, pats[q].len - 1);
    ix.dl = DL;
    uint8_t next_group = 0;
    // LDS the bitmaps may take (8 B per 64 keys + 2 B of rank prefix): the 16 waves of a CU keep >= 192 queue entries each
    const size_t lds_budget = 160u * 1024u - 16u * 192u * 16u - 256u;
    size_t lds_used = 0, lds_reserved = 0, lds_upgrades = 0;
    auto shape_bytes = [](unsigned key_bits) { return (size_t)(key_bits <= 10 ? 16u : (1u << (key_bits - 6))) * 10u; };
    // blocks that may take the 17th key bit, panel-wide (IPCR_INDEX_HALF_MAX; every one doubles a 10 KB bitmap and the waves'
    // hit queues shrink by as much: with the eight shapes of a split panel four of them cost more than they save, 6.84 ms
    // per 3 Gb against 6.29 with none)
    int half_left = env_flag("IPCR_INDEX_HALF_BASES", true) ? (getenv("IPCR_INDEX_HALF_MAX") ? atoi(getenv("IPCR_INDEX_HALF_MAX")) : -1) : 0;
    // Two steps per lookup (jit.cpp): possible when EVERY group keys on "3 protected bases + 5 block bases" -- the panel's
    // tables are then 2^12 32-bit words per shape (18 KiB with the rank prefixes), one ds_read_b32 serves two base steps.
    bool paired = !groups.empty();
    {
        size_t ns_all = 0;
        int lmax = 0;
        for (const Group &g : groups) {
            const int t = std::min(g.t, g.lmin);
            const int bf = (t >= g.lmin) ? 0 : (g.lmin - t) / g.nblk;
            if (!(k >= 1 && t >= 3 && bf >= 5) || g.noprot) paired = false;
            ns_all += (size_t)g.nblk;
            for (uint32_t q : g.members) lmax = std::max(lmax, pats[q].len);
        }
        if (ns_all > IPCR_INDEX_MAX_SHAPES || ns_all * (2048u * 9u) > lds_budget) paired = false;
        if (paired) paired = ipcr::jit_index_pairable(ns_all, lmax - 1);
    }
    {   // what the groups' bitmaps take before any block is given a 17th bit: an upgrade of an early group must not eat the
        // queue room that a later group's plain bitmaps need
        size_t base_all = 0;
        for (const Group &g : groups) {
            const int t = std::min(g.t, g.lmin);
            const int bf = (t >= g.lmin) ? 0 : (g.lmin - t) / g.nblk;
            const bool tri = !g.noprot && k >= 1 && t >= 3 && bf >= 1;
            const int tu = g.noprot ? 0 : (tri ? 3 : std::min(t, 8));
            const int b = tri ? std::min(5, bf) : ((tu >= 8) ? 0 : std::min(8 - tu, bf));
            base_all += (size_t)(b > 0 ? g.nblk : 1) * shape_bytes((unsigned)(2 * tu + 2 * b));
        }
        lds_reserved = base_all;
    }
    for (Group &g : groups) {
        const int t = std::min(g.t, g.lmin);
        const int bf = (t >= g.lmin) ? 0 : (g.lmin - t) / g.nblk;
        // With k >= 1 and at least three protected bases the key is "3 protected bases next to the anchor + 5 block
        // bases" for every block: the protected part is then the low six key bits of all the group's shapes, which
        // the kernel extracts once per step (device_types.h).  Otherwise: as many protected bases as fit.
        // (A split group of the second family, noprot, keys on its blocks alone: the first t bases hold a mismatch.)
        const bool tri = !g.noprot && k >= 1 && t >= 3 && bf >= 1;
        const int tu = g.noprot ? 0 : (tri ? 3 : std::min(t, 8));
        const int b = tri ? std::min(5, bf) : ((tu >= 8) ? 0 : std::min(8 - tu, bf));
        const int ns = b > 0 ? g.nblk : 1;
        if ((tu == 0 && b == 0) || ix.shapes.size() + (size_t)ns > IPCR_INDEX_MAX_SHAPES) { // nothing exact to key on
            for (uint32_t q : g.members) dropped[q] = 1;
            continue;
        }
        const bool fast = (ns > 1 && tu == 3) || (ns == 1 && b == 0 && tu >= 3);
        // Where the k+1 blocks lie among the A = lmin - t bases behind the protected ones is free, as long as no base
        // feeds two of them.  Candidates: blocks of b bases packed towards the anchor, spare bases at the far end --
        // except that every block of a chosen subset owns a (b+1)-th base and reads one bit of it (11 block bits: half
        // the false hits, twice the bitmap).  The layout is picked on the panel itself: its cost is the drain work one
        // genome base causes, sum over shapes of (keys + 2 x further patterns chained under a key) / 2^(key bits) --
        // e.g. the first bases of the reference's benchmark primers take only four values
        // (performance_benchmark_test.go:78-93), and a block that ends on them files 2048 primers under 891 keys.
        const int A = g.lmin - t;
        std::vector<int> bits((size_t)ns, 2 * b), pos((size_t)ns, 0);
        if ((!tri || b < 5) && b > 0 && ns > 1) {
            // fewer than three protected bases: a key is (the protected bases +) one block of up to 8 - tu bases, and
            // short keys are dense (1024 rows, k = 2, no window: 2048 patterns per side under 12-bit keys).  The blocks
            // tile ALL the A bases -- the first A mod (k+1) blocks one base longer -- and read as much of their span
            // as a key can hold (20-mers, k = 2: 7 + 7 + 6 bases instead of 6 + 6 + 6: half the hits).  The same
            // for three protected bases + blocks shorter than five (k = 3 on 20-mers: 5 + 4 + 4 + 4)
            const int rem = A - bf * ns;
            for (int j = 0; j < ns; ++j) {
                const int span = bf + (j < rem ? 1 : 0);
                bits[(size_t)j] = 2 * std::min(tri ? 5 : 8 - tu, span);
                if (j) pos[(size_t)j] = pos[(size_t)j - 1] + bf + (j - 1 < rem ? 1 : 0);
            }
        } else
            for (int j = 1; j < ns; ++j) pos[(size_t)j] = pos[(size_t)j - 1] + (b > 0 ? b : 0);
        for (int j = 0; j < ns; ++j) lds_used += paired ? (size_t)2048u * 9u : shape_bytes((unsigned)(2 * tu + bits[(size_t)j]));
        // files the group's shapes and keys for one layout; returns its cost (see above)
        auto emit = [&](const std::vector<int> &pos, const std::vector<int> &bits) -> double {
            const size_t ents_before = ents.size();
            const size_t shapes_before = ix.shapes.size();
            for (int j = 0; j < ns; ++j) {
                ipcr_index_shape sh{};
                sh.left = g.left ? 1 : 0;
                sh.group = next_group;
                sh.fast = fast ? 1 : 0;
                sh.paired = paired ? 1 : 0;
                sh.dl = (uint8_t)(g.left ? DL : 0);
                sh.tw_bits = (uint8_t)(2 * tu);
                sh.tw_mask = tu ? (uint32_t)((1ull << (2 * tu)) - 1ull) : 0u;
                const int nb = bits[(size_t)j], pj = pos[(size_t)j];
                const int bases = (nb + 1) / 2; // bases the block field touches (the last one with one bit only when nb is odd)
                sh.blk_mask = nb ? (uint32_t)((1ull << nb) - 1ull) : 0u;
                uint64_t vm = 0;
                if (!g.left) { // anchored at the window end = lowest bits of the k-mer; an odd field ends with the LOW bit of its last base
                    sh.tw_shift = 0;
                    sh.blk_shift = (uint8_t)(2 * (t + pj));
                    for (int u = 0; u < tu; ++u) vm |= 1ull << (2 * u);
                    for (int u = 0; u < bases; ++u) vm |= 1ull << (2 * (t + pj + u));
                } else {       // anchored at the window start = base DL of the k-mer, the pattern runs towards base 0; an odd field starts with the HIGH bit of its last base
                    sh.tw_shift = (uint8_t)(tu ? 2 * (DL + 1 - tu) : 0);
                    sh.blk_shift = (uint8_t)(nb ? 2 * (DL + 1 - (t + pj)) - nb : 0);
                    for (int u = 0; u < tu; ++u) vm |= 1ull << (2 * (DL - u));
                    for (int u = 0; u < bases; ++u) vm |= 1ull << (2 * (DL - (t + pj + u)));
                }
                sh.valid_mask = vm;
                const uint32_t s = (uint32_t)ix.shapes.size();
                ix.shapes.push_back(sh);
                for (uint32_t q : g.members) {
                    if (dropped[q]) continue;
                    const Pat &pt = pats[q];
                    const int L = pt.len;
                    const int up = g.left ? 2 * (DL + 1 - L) : 0; // pattern bit -> k-mer bit
                    // key positions of this shape, in pattern (right-aligned) bit coordinates, and the
                    // bases each may take: IUPAC codes expand into every concrete key (capped)
                    std::vector<int> kb;
                    for (int bit = 0; bit < 64; bit += 2)
                        if (vm & (1ull << bit)) kb.push_back(bit - up);
                    uint64_t combos = 1;
                    for (int pb : kb) {
                        int n = 0;
                        for (int bb = 0; bb < 4; ++bb) n += (pt.ok[bb] >> pb) & 1;
                        combos *= (uint64_t)n;
                        if (combos > 64) break;
                    }
                    if (combos == 0 || combos > 64) { dropped[q] = 1; continue; } // too degenerate to key
                    for (uint64_t it = 0; it < combos; ++it) {
                        uint64_t km = 0, rem = it;
                        for (size_t z = 0; z < kb.size(); ++z) {
                            int opts[4], n = 0;
                            for (int bb = 0; bb < 4; ++bb)
                                if ((pt.ok[bb] >> kb[z]) & 1) opts[n++] = bb;
                            const int pick = opts[rem % (uint64_t)n];
                            rem /= (uint64_t)n;
                            km |= (uint64_t)pick << (kb[z] + up);
                        }
                        const uint32_t key = ((uint32_t)(km >> sh.tw_shift) & sh.tw_mask) |
                                             (((uint32_t)(km >> sh.blk_shift) & sh.blk_mask) << sh.tw_bits);
                        ents.emplace_back((s << 20) | key, q);
                    }
                }
            }
            // a pattern dropped in a later shape must lose the keys of the earlier ones too
            size_t w = ents_before;
            for (size_t i = ents_before; i < ents.size(); ++i)
                if (!dropped[ents[i].second]) ents[w++] = ents[i];
            ents.resize(w);
            bool any = false;
            for (uint32_t q : g.members) any |= !dropped[q];
            if (!any) ix.shapes.resize(shapes_before);
            else ++next_group;
            double cost = 0;
            {   // keys and chained patterns per shape (the group's entries are the tail of `ents`)
                std::vector<std::pair<uint32_t, uint32_t>> mine(ents.begin() + (long)ents_before, ents.end());
                std::sort(mine.begin(), mine.end());
                mine.erase(std::unique(mine.begin(), mine.end()), mine.end());
                for (size_t i = 0; i < mine.size(); ++i) {
                    const bool first = i == 0 || mine[i].first != mine[i - 1].first;
                    const uint32_t sidx = mine[i].first >> 20;
                    cost += (first ? 1.0 : 2.0) / (double)(1ull << ipcr_index_key_bits(ix.shapes[sidx]));
                }
            }
            return cost;
        };
        if (tri && b == 5 && ns > 1 && !paired) {
            // Measured on the 4096-pattern panel (C4, 3 Gb): blocks of 11, 10, 11 bits are picked (cost 0.13 against 0.18
            // for three 10-bit blocks) and a third fewer hits reach the drain.  The larger bitmaps leave the waves smaller
            // hit queues (round 2, two steps per entry: 1 % slower); with one queue entry per four steps the queues fill
            // half as fast and the layout wins: 5.13 -> 4.97 ms.  IPCR_INDEX_HALF_BASES=0 turns it off.
            const bool half_bases = half_left != 0;
            const int spare = A - b * ns;
            double best = -1;
            unsigned best_mask = 0;
            const std::vector<char> dropped_keep = dropped;
            const size_t ents_keep = ents.size(), shapes_keep = ix.shapes.size();
            const uint8_t group_keep = next_group;
            for (unsigned mask = 0; mask < (half_bases ? (1u << ns) : 1u); ++mask) {
                if (__builtin_popcount(mask) > spare) continue;
                if (half_left >= 0 && __builtin_popcount(mask) > half_left) continue;
                if (std::max(lds_used, lds_reserved + lds_upgrades) + (size_t)__builtin_popcount(mask) * (shape_bytes(17) - shape_bytes(16)) > lds_budget) continue;
                std::vector<int> cb(bits), cp(pos);
                for (int j = 0; j < ns; ++j) {
                    if (mask & (1u << j)) cb[(size_t)j] = 11;
                    if (j) cp[(size_t)j] = cp[(size_t)j - 1] + ((mask & (1u << (j - 1))) ? b + 1 : b);
                }
                const double c = emit(cp, cb);
                bool lost = false; // a layout that makes a pattern unkeyable (too many IUPAC expansions) only as a last resort
                for (uint32_t q : g.members) lost |= dropped[q] && !dropped_keep[q];
                const double score = c + (lost ? 1e3 : 0.0);
                if (best < 0 || score < best) { best = score; best_mask = mask; }
                dropped = dropped_keep; // roll back
                ents.resize(ents_keep);
                ix.shapes.resize(shapes_keep);
                next_group = group_keep;
            }
            for (int j = 0; j < ns; ++j) {
                if (best_mask & (1u << j)) { bits[(size_t)j] = 11; lds_used += shape_bytes(17) - shape_bytes(16); lds_upgrades += shape_bytes(17) - shape_bytes(16); if (half_left > 0) --half_left; }
                if (j) pos[(size_t)j] = pos[(size_t)j - 1] + ((best_mask & (1u << (j - 1))) ? b + 1 : b);
            }
        }
        (void)emit(pos, bits);
    }
    ix.max_right = 0; // longest right-anchored pattern the index serves
    ix.uniform_len = -1;
    for (uint32_t q = 0; q < P; ++q) {
        if (dropped[q] || pats[q].len <= 0) continue;
        if (!pats[q].left) ix.max_right = std::max(ix.max_right, pats[q].len);
        ix.uniform_len = (ix.uniform_len == -1 || ix.uniform_len == pats[q].len) ? pats[q].len : 0;
    }
    if (ix.uniform_len < 0) ix.uniform_len = 0;
    for (uint32_t q = 0; q < P; ++q)
        if (dropped[q]) ix.leftover.push_back(q);
    if (paired && !ix.shapes.empty() && !ix.shapes[0].paired) paired = false; // (cannot happen: every shape was filed under the same decision)
    if (paired) // entries are ranked in the order of the tables' low halves: word = the key's low 4 + 8 bits, bit = its high 2 + 2
        for (auto &e : ents) {
            const uint32_t key = e.first & 0xFFFFu, prot = key & 63u, blk = key >> 6;
            const uint32_t wo = ((blk & 0xFFu) << 4) | (prot & 15u), bo = ((blk >> 8) << 2) | (prot >> 4);
            e.first = (e.first & ~0xFFFFFu) | (wo << 4) | bo;
        }
    std::sort(ents.begin(), ents.end());
    ents.erase(std::unique(ents.begin(), ents.end()), ents.end());
    // Direct index instead of a hash table: a set bitmap bit means the key is in the panel, and its rank among
    // the set bits IS the index of its entry -- one entry load per hit, no tag compare, no probing.  The rank
    // comes from a per-256-bit-group prefix (uint16, LDS) plus popcounts inside the group.
    const size_t NS = ix.shapes.size();
    // LDS image: the shapes' bitmaps one after the other (64-bit words) | a uint16 rank prefix per bitmap word |
    // NS x 4 words of shape constants (shifts, masks, first bitmap word, first entry)
    std::vector<uint32_t> off64(NS + 1, 0);
    for (size_t i = 0; i < NS; ++i) off64[i + 1] = off64[i] + ipcr_index_words64(ix.shapes[i]);
    const size_t T64 = off64[NS];
    const size_t pfx_words = paired ? T64 / 4 : T64 / 2; // one uint16 per 64 keys: a 64-bit bitmap word, or the low halves of four table words
    const size_t img_words = T64 * 2 + pfx_words + 4 * NS; // (T64 is a multiple of 16: the constants are 16-byte aligned)
    ix.lds_image.assign(std::max<size_t>(1, img_words), 0u);
    uint16_t *prefix = reinterpret_cast<uint16_t *>(ix.lds_image.data() + T64 * 2);
    uint32_t *base = ix.lds_image.data() + T64 * 2 + pfx_words;
    size_t ndistinct = 0;
    for (size_t i = 0; i < ents.size(); ++i)
        if (i == 0 || ents[i].first != ents[i - 1].first) ++ndistinct;
    ipcr_index_entry blank{};
    blank.next = 0xFFFFFFFFu;
    ix.table.assign(ndistinct, blank);
    auto fill = [&](ipcr_index_entry &en, uint32_t q) {
        en.pattern = q;
        for (int bb = 0; bb < 4; ++bb) en.ok[bb] = pats[q].ok[bb];
        en.prot2 = pats[q].prot2;
        en.len = (uint32_t)pats[q].len;
        const uint64_t E = 0x5555555555555555ull;
        const uint64_t wm = pats[q].len >= 32 ? E : (((1ull << (2 * pats[q].len)) - 1ull) & E);
        const uint64_t a = pats[q].ok[0], c = pats[q].ok[1], g = pats[q].ok[2], t = pats[q].ok[3];
        // exactly one base allowed at every position?
        const bool acgt = ((a ^ c ^ g ^ t) == wm) && ((a & c) | (a & g) | (a & t) | (c & g) | (c & t) | (g & t)) == 0;
        en.seq2 = acgt ? ((c | t) | ((g | t) << 1)) : 0ull;
        en.flags = (pats[q].left ? 1u : 0u) | (acgt ? 2u : 0u);
        if (!acgt) ix.all_acgt = false;
    };
    std::vector<uint32_t> shape_count(NS + 1, 0);
    size_t r = 0;
    for (size_t i = 0; i < ents.size();) {
        const uint32_t tag = ents[i].first, sidx = tag >> 20, key = tag & 0xFFFFFu;
        if (paired) {
            // low half: the key as the older step of a pair looks it up; high half: as the newer one does (word = the key's
            // high 4 + 8 bits, bit = its low 2 + 2)
            const uint32_t wo = key >> 4, bo = key & 15u;
            const uint32_t prot = ((bo & 3u) << 4) | (wo & 15u), blk = ((bo >> 2) << 8) | (wo >> 4);
            ix.lds_image[off64[sidx] * 2u + wo] |= 1u << bo;
            ix.lds_image[off64[sidx] * 2u + (((blk >> 2) << 4) | (prot >> 2))] |= 1u << (16u + (((blk & 3u) << 2) | (prot & 3u)));
        } else
            ix.lds_image[off64[sidx] * 2u + (key >> 5)] |= 1u << (key & 31u);
        ++shape_count[sidx];
        fill(ix.table[r], ents[i].second);
        uint32_t tail = (uint32_t)r;
        for (++i; i < ents.size() && ents[i].first == tag; ++i) { // further patterns with the same key
            ipcr_index_entry more = blank;
            fill(more, ents[i].second);
            ix.table[tail].next = (uint32_t)ix.table.size();
            tail = (uint32_t)ix.table.size();
            ix.table.push_back(more);
        }
        ++r;
    }
    uint32_t run = 0;
    bool prefix_overflow = false;
    std::vector<uint32_t> first_entry(NS, 0);
    for (size_t sidx = 0; sidx < NS; ++sidx) {
        first_entry[sidx] = run;
        uint32_t within = 0;
        if (paired)
            for (uint32_t g = off64[sidx] / 2u; g < off64[sidx + 1] / 2u; ++g) { // one prefix per four table words (their low halves: 64 keys)
                prefix[g] = (uint16_t)within;
                for (uint32_t w = 0; w < 4; ++w) within += (uint32_t)__builtin_popcount(ix.lds_image[g * 4u + w] & 0xFFFFu);
            }
        else
        for (uint32_t g = off64[sidx]; g < off64[sidx + 1]; ++g) { // one prefix per 64-bit word of the bitmap
            prefix[g] = (uint16_t)within; // checked below: a shape files fewer than 65536 keys
            for (uint32_t w = 0; w < 2; ++w) within += (uint32_t)__builtin_popcount(ix.lds_image[g * 2u + w]);
        }
        if (within >= 65536u) prefix_overflow = true;
        run += shape_count[sidx];
    }
    // the shapes' constants for the drain, which handles a hit's shape as a run-time value:
    // {tw_shift | blk_shift << 8 | tw_bits << 16, tw_mask | blk_mask << 16, first 64-bit bitmap word, first entry}
    for (size_t sidx = 0; sidx < NS; ++sidx) {
        const ipcr_index_shape &sh = ix.shapes[sidx];
        base[4 * sidx] = (uint32_t)sh.tw_shift | ((uint32_t)sh.blk_shift << 8) | ((uint32_t)sh.tw_bits << 16);
        base[4 * sidx + 1] = (sh.tw_mask & 0xFFFFu) | ((sh.blk_mask & 0xFFFFu) << 16);
        base[4 * sidx + 2] = off64[sidx];
        base[4 * sidx + 3] = first_entry[sidx];
    }
    static const bool debug = env_flag("IPCR_INDEX_DEBUG", false);
    if (debug) { // per shape: key bits, distinct keys, filed patterns, longest chain
        std::vector<uint32_t> ent_count(NS, 0), longest(NS, 0);
        for (size_t i = 0; i < ents.size();) {
            size_t j = i;
            while (j < ents.size() && ents[j].first == ents[i].first) ++j;
            const uint32_t sidx = ents[i].first >> 20;
            ent_count[sidx] += (uint32_t)(j - i);
            longest[sidx] = std::max(longest[sidx], (uint32_t)(j - i));
            i = j;
        }
        for (size_t sidx = 0; sidx < NS; ++sidx)
            fprintf(stderr, "ipcr index shape %zu: %s group %u, %u key bits (block shift %u, mask %#x), %u keys, %u patterns filed, longest chain %u\n",
                    sidx, ix.shapes[sidx].left ? "left" : "right", ix.shapes[sidx].group, ipcr_index_key_bits(ix.shapes[sidx]),
                    ix.shapes[sidx].blk_shift, ix.shapes[sidx].blk_mask, shape_count[sidx], ent_count[sidx], longest[sidx]);
    }
    std::sort(ix.leftover.begin(), ix.leftover.end());
    ix.usable = !ix.shapes.empty() && !ents.empty() && !prefix_overflow && (size_t)ipcr::jit_index_image_bytes(ix.shapes) + 16u * 128u * 16u <= 160u * 1024u;
}

void build_dev_pattern(const ipcr_panel &p, const PatternDef &d, uint32_t gid, ipcr_dev_pattern &o) {
    memset(&o, 0, sizeof o);
    const int L = (int)d.seq.size();
    o.len = (uint16_t)L;
    o.seed_off = (uint16_t)d.seed_off;
    o.seed_len = (uint16_t)(d.seeded ? d.seed_len : 0);
    o.global_id = gid;
    int tw = d.tw_dev;
    if (tw > L) tw = L;
    for (int j = 0; j < L; ++j) {
        uint8_t m = T.mask[(uint8_t)d.seq[(size_t)j]] & 15u;
        const bool prot = p.cfg.max_mm == 0 || (tw > 0 && (d.left ? j < tw : j >= L - tw));
        o.mask[j] = (uint8_t)(m | (prot ? 16u : 0u));
    }
}

} // namespace

extern "C" {

const char *ipcr_version(void) { return "ipcr-hip 0.1.0 (bitsliced-pigeonhole-gfx950)"; }
const char *ipcr_last_error(void) { return g_err.c_str(); }

int ipcr_device_count(void) { return slot_count(); }

ipcr_status ipcr_set_device(int device) {
    if (device < 0 || device >= slot_count()) return fail(IPCR_ERR_INVALID, "ipcr_set_device: device %d of %d", device, slot_count());
    g_default_slot.store(device, std::memory_order_relaxed); // what ipcr_scratch_create / ipcr_genome_create use from now on, on any thread
    HIPCHK(hipSetDevice(slot_phys(device)));                 // and the calling thread's own HIP device (a Python host shares it with torch)
    return IPCR_OK;
}

int ipcr_bind_thread_to_device(int device) {
    if (device < 0) device = default_slot();
    if (device >= slot_count()) return 0;
    return bind_this_thread(slot_phys(device)) ? 1 : 0;
}

uint8_t ipcr_iupac_mask(uint8_t c) { return T.mask[c]; }

int ipcr_base_match(uint8_t g, uint8_t p) { // core/primer/iupac.go:62-67
    if (g != 'A' && g != 'C' && g != 'G' && g != 'T') return 0;
    return (T.mask[p] & T.mask[g]) != 0;
}

ipcr_status ipcr_revcomp(const char *seq, size_t n, char *out) {
    if (!seq || !out) return fail(IPCR_ERR_INVALID, "ipcr_revcomp: null argument");
    std::string o;
    size_t bad = 0;
    if (!revcomp(std::string(seq, n), o, &bad))
        return fail(IPCR_ERR_PRIMER,
                    "invalid reverse-complement base %c at position %zu; expected normalized uppercase IUPAC DNA",
                    seq[bad - 1], bad);
    memcpy(out, o.data(), n);
    return IPCR_OK;
}

ipcr_status ipcr_panel_create(const ipcr_config *cfg, const ipcr_pair *pairs, int32_t n_pairs, ipcr_panel **out) {
    if (!cfg || !out || n_pairs < 0 || (n_pairs > 0 && !pairs)) return fail(IPCR_ERR_INVALID, "ipcr_panel_create: null argument");
    *out = nullptr;
    // The reference never validates --mismatches (internal/clibase/common.go:124-174) and its two
    // matchers disagree on negative values (ac.go:207 vs match.go:79); the boundary refuses them.
    if (cfg->max_mm < 0) return fail(IPCR_ERR_INVALID, "max_mm must be >= 0 (got %d)", cfg->max_mm);
    if (cfg->max_mm > IPCR_MAX_MM) return fail(IPCR_ERR_UNSUPPORTED, "max_mm %d exceeds IPCR_MAX_MM (%d)", cfg->max_mm, IPCR_MAX_MM);
    std::unique_ptr<ipcr_panel> p(new ipcr_panel);
    p->cfg = *cfg;
    p->tw = cfg->terminal_window > 0 ? cfg->terminal_window : 0;
    p->specialize = env_flag("IPCR_SPECIALIZE", true); // 0: the table-driven kernel for every panel (profiles; ipcr_panel_set_specialize per panel)
    const int k = cfg->max_mm, tw = p->tw;
    std::map<std::string, uint32_t> index; // key: seq | side | tw_dev
    auto intern = [&](const std::string &seq, bool left, int tw_dev, const SeedSpan &sp) -> uint32_t {
        std::string key = seq + (left ? "|L" : "|R") + std::to_string(tw_dev);
        auto it = index.find(key);
        if (it != index.end()) return it->second;
        PatternDef d{seq, left, tw_dev, sp.off, sp.len, sp.seeded};
        p->defs.push_back(d);
        index.emplace(key, (uint32_t)p->defs.size() - 1);
        return (uint32_t)p->defs.size() - 1;
    };
    for (int i = 0; i < n_pairs; ++i) {
        if (!pairs[i].forward || !pairs[i].reverse) return fail(IPCR_ERR_INVALID, "pair %d: null primer", i);
        std::array<std::string, 4> o;
        o[0] = pairs[i].forward;
        o[1] = pairs[i].reverse;
        size_t bad = 0;
        for (int w = 0; w < 2; ++w) {
            if (o[(size_t)w].size() > IPCR_MAX_PRIMER_LEN)
                return fail(IPCR_ERR_UNSUPPORTED, "pair %d: primer of %zu nt exceeds IPCR_MAX_PRIMER_LEN (%d)", i,
                            o[(size_t)w].size(), IPCR_MAX_PRIMER_LEN);
            if (!revcomp(o[(size_t)w], o[(size_t)w + 2], &bad)) // core/primer/rc.go:27-34 panics here
                return fail(IPCR_ERR_PRIMER,
                            "pair %d: invalid reverse-complement base %c at position %zu; expected normalized uppercase IUPAC DNA",
                            i, o[(size_t)w][bad - 1], bad);
        }
        p->id.push_back(pairs[i].id ? pairs[i].id : "");
        p->fwd.push_back(o[0]);
        p->rev.push_back(o[1]);
        p->minp.push_back(pairs[i].min_product);
        p->maxp.push_back(pairs[i].max_product);
        uint8_t have = 0;
        std::array<std::array<uint32_t, 2>, 4> sl{};
        for (int w = 0; w < 4; ++w) {
            const bool left = w >= 2;
            const std::string &s = o[(size_t)w];
            if ((int)s.size() > p->max_len) p->max_len = (int)s.size();
            const SeedSpan sp = reference_seed_span(s, cfg->seed_len, !left, left ? tw : 0, left ? 0 : tw, k);
            if (sp.seeded) have |= (uint8_t)(1u << w);
            for (int mode = 0; mode < 2; ++mode) {
                int tw_dev = tw;
                // rc orientations on the reference's FindMatches path are capped BEFORE the 5' window
                // filter (compiled.go:249-256): scan them unprotected and filter on the host.
                if (left && tw > 0 && k > 0 && cfg->hit_cap > 0 && (!sp.seeded || mode == 1)) tw_dev = 0;
                sl[(size_t)w][(size_t)mode] = intern(s, left, tw_dev, sp);
            }
        }
        p->have.push_back(have);
        p->slot.push_back(sl);
        p->ori.push_back(std::move(o));
    }
    for (int mode = 0; mode < 2; ++mode) {
        std::vector<char> used(p->defs.size(), 0);
        for (auto &sl : p->slot)
            for (int w = 0; w < 4; ++w) used[sl[(size_t)w][(size_t)mode]] = 1;
        for (uint32_t g = 0; g < p->defs.size(); ++g)
            if (used[g] && !p->defs[g].seq.empty()) {
                p->set[mode].ids.push_back(g);
                ipcr_dev_pattern dp;
                build_dev_pattern(*p, p->defs[g], g, dp);
                p->set[mode].host.push_back(dp);
            }
    }
    p->modes_equal = p->set[0].ids == p->set[1].ids;
    for (int mode = 0; mode < 2; ++mode) {
        p->users[mode].assign(p->defs.size(), {});
        for (uint32_t i = 0; i < p->slot.size(); ++i)
            for (int w = 0; w < 4; ++w) {
                auto &u = p->users[mode][p->slot[i][(size_t)w][(size_t)mode]];
                if (u.empty() || u.back() != i) u.push_back(i);
            }
    }
    *out = p.release();
    return IPCR_OK;
}

void ipcr_panel_destroy(ipcr_panel *p) {
    if (!p) return;
    for (auto &kv : p->devs) {
        DeviceGuard dg(kv.first);
        for (SetDev &s : kv.second->set) {
            if (s.jit_thread.joinable()) s.jit_thread.join();
            if (s.index_jit) ipcr::jit_destroy(s.index_jit);
            for (ipcr::JitFilter *f : s.leftover_jit) ipcr::jit_destroy(f);
            if (s.d_lds_image) (void)hipFree(s.d_lds_image);
            if (s.d_table) (void)hipFree(s.d_table);
            if (s.d_leftover) (void)hipFree(s.d_leftover);
            if (s.dev) (void)hipFree(s.dev);
            for (ipcr::JitFilter *f : s.jit) ipcr::jit_destroy(f);
        }
    }
    delete p;
}

ipcr_status ipcr_panel_wait_ready(const ipcr_panel *cp) {
    if (!cp) return fail(IPCR_ERR_INVALID, "null panel");
    ipcr_panel *p = const_cast<ipcr_panel *>(cp);
    std::lock_guard<std::mutex> lock(p->mu);
    for (auto &kv : p->devs)
        for (SetDev &d : kv.second->set) {
            if (d.jit_thread.joinable()) d.jit_thread.join();
            if (d.jit_state.load(std::memory_order_acquire) == 2) {
                d.jit_pub.store(&d.jit, std::memory_order_release);
                d.jit_state.store(0);
            }
        }
    return IPCR_OK;
}

int32_t ipcr_panel_device_slots(const ipcr_panel *p) {
    if (!p) return 0;
    std::lock_guard<std::mutex> lock(p->mu);
    return (int32_t)p->devs.size();
}

int32_t ipcr_panel_num_pairs(const ipcr_panel *p) { return p ? (int32_t)p->id.size() : 0; }
int32_t ipcr_panel_num_patterns(const ipcr_panel *p) { return p ? (int32_t)p->set[0].ids.size() : 0; }
int32_t ipcr_panel_max_primer_len(const ipcr_panel *p) { return p ? p->max_len : 0; }

int32_t ipcr_panel_have(const ipcr_panel *p, int32_t pair, char which) {
    if (!p || pair < 0 || pair >= (int32_t)p->have.size()) return 0;
    int w = which == 'A' ? 0 : which == 'B' ? 1 : which == 'a' ? 2 : which == 'b' ? 3 : -1;
    if (w < 0) return 0;
    return (p->have[(size_t)pair] >> w) & 1;
}

int32_t ipcr_panel_num_patterns_total(const ipcr_panel *p) { return p ? (int32_t)p->defs.size() : 0; }

ipcr_status ipcr_panel_pattern_info(const ipcr_panel *p, int32_t pattern, char *seq_out, size_t cap,
                                    int32_t *left_window, int32_t *tw_dev, int32_t *seed_off,
                                    int32_t *seed_len) {
    if (!p || pattern < 0 || pattern >= (int32_t)p->defs.size()) return fail(IPCR_ERR_INVALID, "pattern index out of range");
    const PatternDef &d = p->defs[(size_t)pattern];
    if (seq_out && cap) {
        const size_t n = std::min(cap - 1, d.seq.size());
        memcpy(seq_out, d.seq.data(), n);
        seq_out[n] = 0;
    }
    if (left_window) *left_window = d.left ? 1 : 0;
    if (tw_dev) *tw_dev = p->cfg.max_mm == 0 ? (int32_t)d.seq.size() : d.tw_dev;
    if (seed_off) *seed_off = d.seed_off;
    if (seed_len) *seed_len = d.seeded ? d.seed_len : 0;
    return IPCR_OK;
}

int32_t ipcr_panel_slot_pattern(const ipcr_panel *p, int32_t pair, char which, int32_t mode) {
    if (!p || pair < 0 || pair >= (int32_t)p->slot.size() || mode < 0 || mode > 1) return -1;
    const int w = which == 'A' ? 0 : which == 'B' ? 1 : which == 'a' ? 2 : which == 'b' ? 3 : -1;
    if (w < 0) return -1;
    return (int32_t)p->slot[(size_t)pair][(size_t)w][(size_t)mode];
}

ipcr_status ipcr_panel_filter_source(const ipcr_panel *p, int32_t mode, char *out, size_t cap, size_t *needed) {
    // mode 0 / 1: the specialised filter of that pattern set; 2 / 3: the seed-index filter's; 4 / 5: the specialised filter in
    // its form for small launches (a block shared by four waves)
    int segs = 1;
    if (mode == 4 || mode == 5) { segs = 4; mode -= 4; }
    if (!p || mode < 0 || mode > 3) return fail(IPCR_ERR_INVALID, "ipcr_panel_filter_source: bad argument");
    if (mode >= 2) { // the seed-index filter's source for mode - 2
        ipcr_panel *mp = const_cast<ipcr_panel *>(p);
        std::lock_guard<std::mutex> lock(mp->mu);
        PatternSet &ps = mp->set[mode - 2];
        if (!ps.index.built) build_index(*mp, ps);
        const std::string isrc = ps.index.usable ? ipcr::jit_index_source(ps.index.shapes, ps.index.geom()) : std::string();
        if (needed) *needed = isrc.size() + 1;
        if (out && cap) {
            const size_t n = std::min(cap - 1, isrc.size());
            memcpy(out, isrc.data(), n);
            out[n] = 0;
        }
        return IPCR_OK;
    }
    const std::vector<ipcr_dev_pattern> &all = p->set[mode].host;
    const size_t G = ipcr::jit_group_size(all, p->cfg.max_mm); // source of the first pattern group
    const std::string src = G ? ipcr::jit_source(std::vector<ipcr_dev_pattern>(all.begin(), all.begin() + (long)std::min(G, all.size())), p->cfg.max_mm, 0, nullptr, false, segs) : std::string();
    if (needed) *needed = src.size() + 1;
    if (out && cap) {
        const size_t n = std::min(cap - 1, src.size());
        memcpy(out, src.data(), n);
        out[n] = 0;
    }
    return IPCR_OK;
}

ipcr_status ipcr_panel_set_shard(ipcr_panel *p, int32_t index, int32_t count) {
    if (!p || count < 1 || index < 0 || index >= count) return fail(IPCR_ERR_INVALID, "ipcr_panel_set_shard: need 0 <= index < count");
    std::lock_guard<std::mutex> lock(p->mu);
    for (int mode = 0; mode < 2; ++mode) {
        PatternSet &s = p->set[mode];
        if (!p->devs.empty() || s.index.built) return fail(IPCR_ERR_INVALID, "ipcr_panel_set_shard: the panel has already been scanned with");
        if (p->shard_count != 1) return fail(IPCR_ERR_INVALID, "ipcr_panel_set_shard: the panel is already a shard");
    }
    for (int mode = 0; mode < 2; ++mode) { // every count-th distinct pattern of the scanned list, starting at index
        PatternSet &s = p->set[mode];
        std::vector<uint32_t> ids;
        std::vector<ipcr_dev_pattern> host;
        for (size_t i = 0; i < s.ids.size(); ++i)
            if ((int32_t)(i % (size_t)count) == index) { ids.push_back(s.ids[i]); host.push_back(s.host[i]); }
        s.ids.swap(ids);
        s.host.swap(host);
    }
    p->shard_index = index;
    p->shard_count = count;
    return IPCR_OK;
}

int32_t ipcr_panel_scanned_patterns(const ipcr_panel *p, int32_t mode, int32_t *out, int32_t cap) {
    if (!p || mode < 0 || mode > 1) return 0;
    const std::vector<uint32_t> &ids = p->set[mode].ids;
    for (size_t i = 0; i < ids.size() && out && (int32_t)i < cap; ++i) out[i] = (int32_t)ids[i];
    return (int32_t)ids.size();
}

ipcr_status ipcr_panel_set_specialize(ipcr_panel *p, int32_t enable) {
    if (!p) return fail(IPCR_ERR_INVALID, "null panel");
    p->specialize = enable != 0;
    return IPCR_OK;
}

} // extern "C"

// ---------------------------------------------------------------------------- genome

struct ipcr_genome {
    int device = 0; // device slot
    uint64_t cap_cols = 0; // columns available for records (buffer holds one extra pad block beyond)
    uint32_t max_records = 0;
    uint32_t *planes = nullptr;
    uint32_t *rst = nullptr;
    uint32_t *d_flags = nullptr; // per record, bit0 = holds a non-ACGTacgt byte
    uint64_t *d_rec_start = nullptr, *d_rec_len = nullptr;
    uint32_t *d_block_rec = nullptr; // per block: last record that starts at or before the block (verify's first guess)
    std::vector<uint32_t> block_rec;
    std::vector<uint64_t> rec_start, rec_len; // padded start, length
    std::vector<std::string> ids;             // record IDs (FASTA loader; empty for records added as bytes)
    mutable std::vector<uint8_t> flags;
    mutable bool flags_valid = false;
    bool tables_dirty = true;
    uint64_t next_col = 0;
    uint64_t padded_until = 0; // columns [max(next_col, stale_until), padded_until) are known to be padding
    uint64_t stale_until = 0;  // columns below hold the bases of records forgotten by genome_clear (chunk genome reuse)
    uint64_t total_bases = 0;
    uint8_t *staging = nullptr;
    uint64_t staging_cap = 0;
    bool staging_fine = false; // allocated fine-grained (the host writes it through the BAR: ipcr_scan_chunk)
    uint8_t *h_planes = nullptr; // pinned: the invalid / reset planes of one group of columns (ipcr_genome_add_record through the BAR)
    hipStream_t stream = nullptr;
    bool shared_stream = false; // stream belongs to a scratch (its private chunk genome): never destroyed here
    hipEvent_t e0 = nullptr, e1 = nullptr;
    double pack_ms = 0;
};

namespace {

uint64_t record_cols(uint64_t len) { // whole column pairs, with >= IPCR_PAD_BASES of padding after the record
    const uint64_t two = 2ull * IPCR_COLUMN_BASES;
    return ((len + IPCR_PAD_BASES + two - 1) / two) * 2ull;
}

ipcr_status genome_alloc(ipcr_genome *g, uint64_t cap_cols, uint32_t max_records) {
    g->cap_cols = ((cap_cols + 63) / 64) * 64;
    g->max_records = max_records;
    const uint64_t blocks = g->cap_cols / 64 + 1;
    HIPCHK(hipMalloc((void **)&g->planes, blocks * IPCR_BLOCK_PLANE_WORDS * 4ull));
    HIPCHK(hipMalloc((void **)&g->rst, blocks * IPCR_BLOCK_RST_WORDS * 4ull));
    HIPCHK(hipMalloc((void **)&g->d_flags, (uint64_t)max_records * 4ull));
    HIPCHK(hipMalloc((void **)&g->d_rec_start, (uint64_t)max_records * 8ull));
    HIPCHK(hipMalloc((void **)&g->d_block_rec, (blocks + 1) * 4ull));
    HIPCHK(hipMalloc((void **)&g->d_rec_len, (uint64_t)max_records * 8ull));
    HIPCHK(hipMemset(g->d_flags, 0, (uint64_t)max_records * 4ull));
    HIPCHK(hipStreamSynchronize(nullptr)); // (the fill runs on the null stream; the genome's stream is non-blocking and must not overtake it)
    return IPCR_OK;
}

void genome_free_buffers(ipcr_genome *g) {
    if (g->planes) (void)hipFree(g->planes);
    if (g->rst) (void)hipFree(g->rst);
    if (g->d_flags) (void)hipFree(g->d_flags);
    if (g->d_rec_start) (void)hipFree(g->d_rec_start);
    if (g->d_block_rec) (void)hipFree(g->d_block_rec);
    g->d_block_rec = nullptr;
    if (g->d_rec_len) (void)hipFree(g->d_rec_len);
    g->planes = g->rst = g->d_flags = nullptr;
    g->d_rec_start = g->d_rec_len = nullptr;
}

void genome_clear(ipcr_genome *g, bool flags_elsewhere = false) { // forget the records, keep the buffers (and what is known about padding)
    if (!flags_elsewhere && !g->rec_start.empty() && g->d_flags) (void)hipMemsetAsync(g->d_flags, 0, g->rec_start.size() * 4ull, g->stream);
    g->stale_until = std::max(g->stale_until, g->next_col);
    g->rec_start.clear();
    g->rec_len.clear();
    g->ids.clear();
    g->flags.clear();
    g->flags_valid = false;
    g->tables_dirty = true;
    g->next_col = 0;
    g->total_bases = 0;
}

void genome_account_record(ipcr_genome *g, uint64_t len, uint64_t cols) { // a record's tiles are (being) written at next_col
    g->rec_start.push_back(g->next_col * IPCR_COLUMN_BASES);
    g->rec_len.push_back(len);
    g->ids.emplace_back();
    g->next_col += cols;
    if (g->padded_until < g->next_col) g->padded_until = g->next_col;
    g->total_bases += len;
    g->tables_dirty = true;
    g->flags_valid = false;
}

// ext_flag: where the record's reset-byte flag goes instead of d_flags (the chunk path keeps it in pinned host memory);
// with it the pack kernel also writes the record's start / length into the device tables (no copy operations)
ipcr_status genome_add_device(ipcr_genome *g, const uint8_t *dseq, uint64_t len, bool wait = true, uint32_t *ext_flag = nullptr) {
    const uint64_t cols = record_cols(len);
    if (g->rec_start.size() >= g->max_records) return fail(IPCR_ERR_CAPACITY, "genome holds its maximum of %u records", g->max_records);
    if (g->next_col + cols > g->cap_cols)
        return fail(IPCR_ERR_CAPACITY, "genome capacity exceeded (%llu + %llu columns > %llu)",
                    (unsigned long long)g->next_col, (unsigned long long)cols, (unsigned long long)g->cap_cols);
    if ((reinterpret_cast<uintptr_t>(dseq) & 15u) != 0) return fail(IPCR_ERR_INVALID, "device sequence pointer must be 16-byte aligned");
    const uint32_t rec = (uint32_t)g->rec_start.size();
    if (ext_flag) HIPCHK(ipcr::launch_pack(g->stream, dseq, len, g->next_col, cols, g->planes, g->rst, ext_flag, g->d_rec_start + rec, g->d_rec_len + rec, g->e0, g->e1));
    else HIPCHK(ipcr::launch_pack(g->stream, dseq, len, g->next_col, cols, g->planes, g->rst, g->d_flags + rec, nullptr, nullptr, g->e0, g->e1));
    if (wait) {
        HIPCHK(hipEventSynchronize(g->e1));
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, g->e0, g->e1));
        g->pack_ms += ms;
    }
    genome_account_record(g, len, cols);
    return IPCR_OK;
}

// many records whose bytes lie in one device buffer (every one 16-byte aligned): one pack launch for all of them
ipcr_status genome_add_device_batch(ipcr_genome *g, const uint8_t *dbase, const uint64_t *offs, const uint64_t *lens, size_t n,
                                    void *d_tmp, size_t tmp_bytes, const std::string *ids = nullptr) {
    if (n == 0) return IPCR_OK;
    if (g->rec_start.size() + n > g->max_records) return fail(IPCR_ERR_CAPACITY, "genome holds its maximum of %u records", g->max_records);
    std::vector<ipcr_pack_rec> recs(n);
    std::vector<uint32_t> prefix(n + 1, 0);
    uint64_t col = g->next_col;
    const uint32_t first = (uint32_t)g->rec_start.size();
    for (size_t i = 0; i < n; ++i) {
        const uint64_t cols = record_cols(lens[i]);
        recs[i] = ipcr_pack_rec{offs[i], lens[i], col, (uint32_t)cols, first + (uint32_t)i};
        prefix[i + 1] = prefix[i] + (uint32_t)(cols / 2u);
        col += cols;
    }
    if (col > g->cap_cols)
        return fail(IPCR_ERR_CAPACITY, "genome capacity exceeded (%llu columns > %llu)", (unsigned long long)col, (unsigned long long)g->cap_cols);
    const size_t need = n * sizeof(ipcr_pack_rec) + (n + 1) * 4 + 16;
    if (need > tmp_bytes) return fail(IPCR_ERR_CAPACITY, "batch table does not fit its device buffer");
    ipcr_pack_rec *d_recs = static_cast<ipcr_pack_rec *>(d_tmp);
    uint32_t *d_prefix = reinterpret_cast<uint32_t *>(static_cast<uint8_t *>(d_tmp) + ((n * sizeof(ipcr_pack_rec) + 15) & ~(size_t)15));
    HIPCHK(hipMemcpyAsync(d_recs, recs.data(), n * sizeof(ipcr_pack_rec), hipMemcpyHostToDevice, g->stream));
    HIPCHK(hipMemcpyAsync(d_prefix, prefix.data(), (n + 1) * 4, hipMemcpyHostToDevice, g->stream));
    HIPCHK(ipcr::launch_pack_batch(g->stream, dbase, d_recs, d_prefix, (uint32_t)n, prefix[n], g->planes, g->rst, g->d_flags));
    HIPCHK(hipStreamSynchronize(g->stream)); // recs / prefix are host vectors of this call
    for (size_t i = 0; i < n; ++i) {
        g->rec_start.push_back(recs[i].col0 * IPCR_COLUMN_BASES);
        g->rec_len.push_back(lens[i]);
        g->ids.emplace_back(ids ? ids[i] : std::string());
        g->total_bases += lens[i];
    }
    g->next_col = col;
    if (g->padded_until < g->next_col) g->padded_until = g->next_col;
    g->tables_dirty = true;
    g->flags_valid = false;
    return IPCR_OK;
}

// before a scan: the rest of the last block and one block beyond must be padding, and the
// record tables must be on the device
// the rest of the last block and one block beyond must be padding: fill what is not known to be
ipcr_status genome_pad(ipcr_genome *g) {
    const uint64_t need = (g->next_col + 63) / 64 * 64 + 64;
    const uint64_t lo = g->next_col; // first column that must be padding; known padding: [max(lo, stale_until), padded_until)
    if (g->stale_until > lo) {       // bases of a forgotten, longer record follow the last one (a reused chunk genome)
        HIPCHK(ipcr::launch_fill_pad(g->stream, g->planes, g->rst, lo, std::min(g->stale_until, need)));
        if (g->stale_until > need) return IPCR_OK; // [lo, need) is padding now; what lies beyond stays marked stale
        g->padded_until = std::max(g->padded_until, g->stale_until);
        g->stale_until = lo;
    }
    if (g->padded_until < need) {
        HIPCHK(ipcr::launch_fill_pad(g->stream, g->planes, g->rst, std::max(lo, g->padded_until), need));
        g->padded_until = need;
    }
    return IPCR_OK;
}

ipcr_status genome_finalize(ipcr_genome *g) {
    { const ipcr_status ps = genome_pad(g); if (ps != IPCR_OK) return ps; }
    if (g->tables_dirty && !g->rec_start.empty()) {
        HIPCHK(hipMemcpyAsync(g->d_rec_start, g->rec_start.data(), g->rec_start.size() * 8ull, hipMemcpyHostToDevice, g->stream));
        {
            const uint64_t nb = (g->next_col + 63) / 64 + 1;
            g->block_rec.resize(nb);
            uint32_t r = 0;
            for (uint64_t b = 0; b < nb; ++b) {
                while (r + 1 < g->rec_start.size() && g->rec_start[r + 1] <= b * (uint64_t)IPCR_BLOCK_BASES) ++r;
                g->block_rec[b] = r;
            }
            HIPCHK(hipMemcpyAsync(g->d_block_rec, g->block_rec.data(), nb * 4ull, hipMemcpyHostToDevice, g->stream));
        }
        HIPCHK(hipMemcpyAsync(g->d_rec_len, g->rec_len.data(), g->rec_len.size() * 8ull, hipMemcpyHostToDevice, g->stream));
        g->tables_dirty = false;
    }
    if (!g->flags_valid) {
        std::vector<uint32_t> f(g->rec_start.size());
        if (!f.empty()) HIPCHK(hipMemcpyAsync(f.data(), g->d_flags, f.size() * 4ull, hipMemcpyDeviceToHost, g->stream));
        HIPCHK(hipStreamSynchronize(g->stream));
        g->flags.assign(f.size(), 0);
        for (size_t i = 0; i < f.size(); ++i) g->flags[i] = (uint8_t)(f[i] & 1u);
        g->flags_valid = true;
    } else {
        HIPCHK(hipStreamSynchronize(g->stream));
    }
    return IPCR_OK;
}

// chunk path (one record, the scratch's own stream): same preparation without waiting for anything; the
// record's reset-byte flag arrives in *pinned_flag when the stream has passed this point
// chunk path (one record, the scratch's own stream): nothing is copied and nothing waited for -- the pack kernel has
// written the record's table entries and writes its reset-byte flag into pinned host memory; every block belongs to
// record 0 (d_block_rec is zeroed once, when the chunk genome is created)
ipcr_status genome_finalize_async(ipcr_genome *g) {
    const ipcr_status ps = genome_pad(g);
    if (ps != IPCR_OK) return ps;
    g->tables_dirty = false;
    return IPCR_OK;
}

bool genome_any_reset(const ipcr_genome *g) {
    for (uint8_t f : g->flags)
        if (f & 1u) return true;
    return false;
}

} // namespace

extern "C" {

ipcr_status ipcr_genome_create(uint64_t capacity_bases, uint32_t max_records, ipcr_genome **out) {
    return ipcr_genome_create_on(capacity_bases, max_records, default_slot(), out);
}

int32_t ipcr_genome_device(const ipcr_genome *g) { return g ? g->device : -1; }

ipcr_status ipcr_genome_create_on(uint64_t capacity_bases, uint32_t max_records, int32_t device, ipcr_genome **out) {
    if (!out) return fail(IPCR_ERR_INVALID, "null out");
    *out = nullptr;
    if (slot_count() == 0) return fail(IPCR_ERR_DEVICE, "no HIP device visible: the ipcr scan path has no CPU fallback");
    if (device < 0 || device >= slot_count()) return fail(IPCR_ERR_INVALID, "ipcr_genome_create_on: device %d of %d", device, slot_count());
    if (max_records == 0) max_records = 1;
    DeviceGuard dg(device);
    std::unique_ptr<ipcr_genome> g(new ipcr_genome);
    g->device = device;
    HIPCHK(hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreate(&g->e0));
    HIPCHK(hipEventCreate(&g->e1));
    const uint64_t cols = (capacity_bases + IPCR_COLUMN_BASES - 1) / IPCR_COLUMN_BASES + 3ull * max_records;
    ipcr_status st = genome_alloc(g.get(), cols, max_records);
    if (st != IPCR_OK) { ipcr_genome_destroy(g.release()); return st; }
    *out = g.release();
    return IPCR_OK;
}

void ipcr_genome_destroy(ipcr_genome *g) {
    if (!g) return;
    DeviceGuard dg(g->device);
    genome_free_buffers(g);
    if (g->staging) (void)hipFree(g->staging);
    if (g->h_planes) (void)hipHostFree(g->h_planes);
    if (g->e0) (void)hipEventDestroy(g->e0);
    if (g->e1) (void)hipEventDestroy(g->e1);
    if (g->stream && !g->shared_stream) (void)hipStreamDestroy(g->stream);
    delete g;
}

// A record of host bytes into a resident genome, packed by the process's pool straight into device memory through the BAR
// (the resident form of what ipcr_scan_chunk does for a chunk: DESIGN 5): the code planes are written where the conversion
// kernel reads them, the invalid / reset planes of a group of columns follow by DMA only if the group holds such a byte.
// false: not here (no large BAR, no SIMD packer, a record too small to be worth it) -- the caller sends ASCII.
static bool genome_add_host_packed(ipcr_genome *g, const uint8_t *seq, uint64_t len, ipcr_status *st) {
    *st = IPCR_OK;
    const BarInfo bi = device_bar(slot_phys(g->device));
    if (!bi.writable || !env_flag("IPCR_CHUNK_BAR", true) || !ipcr::pack_linear_is_simd() || len < 4096) return false;
    const uint64_t cols = record_cols(len), col0 = g->next_col;
    if (g->rec_start.size() >= g->max_records) { *st = fail(IPCR_ERR_CAPACITY, "genome holds its maximum of %u records", g->max_records); return true; }
    if (col0 + cols > g->cap_cols) {
        *st = fail(IPCR_ERR_CAPACITY, "genome capacity exceeded (%llu + %llu columns > %llu)", (unsigned long long)col0, (unsigned long long)cols, (unsigned long long)g->cap_cols);
        return true;
    }
    auto hip = [&](hipError_t e, const char *what) { if (e != hipSuccess && *st == IPCR_OK) *st = fail(IPCR_ERR_DEVICE, "%s: %s", what, hipGetErrorString(e)); return e == hipSuccess; };
    constexpr uint64_t GROUP = 2048; // columns per group: 8 Mb
    const uint64_t gcols = std::min(cols, GROUP), dev_bytes = gcols * 2048ull;
    if (dev_bytes > g->staging_cap || !g->staging_fine) {
        if (g->staging) (void)hipFree(g->staging);
        g->staging = nullptr;
        g->staging_fine = false;
        g->staging_cap = std::max(g->staging_cap, dev_bytes);
        if (hipExtMallocWithFlags((void **)&g->staging, g->staging_cap, hipDeviceMallocFinegrained) != hipSuccess) {
            (void)hipGetLastError();
            g->staging = nullptr;
            g->staging_cap = 0;
            return false;
        }
        g->staging_fine = true;
    }
    if (!g->h_planes) { // the invalid / reset planes of one group (pinned: a dirty group's DMA reads them)
        if (!hip(hipHostMalloc((void **)&g->h_planes, GROUP * 1024ull, hipHostMallocDefault), "hipHostMalloc")) return true;
    }
    const uint64_t nthreads = std::max<uint64_t>(1, PackPool::get().size());
    uint32_t flags_all = 0;
    for (uint64_t c0 = 0; c0 < cols && *st == IPCR_OK; c0 += GROUP) {
        const uint64_t nc = std::min(GROUP, cols - c0), W = nc * 128u;
        const uint64_t per = std::max<uint64_t>(8, ((nc + nthreads - 1) / nthreads + 7) / 8 * 8);
        const size_t nitems = (size_t)((nc + per - 1) / per);
        std::vector<uint32_t> iflags(nitems, 0);
        uint32_t *dlo = reinterpret_cast<uint32_t *>(g->staging), *hiv = reinterpret_cast<uint32_t *>(g->h_planes);
        // (the previous group's kernel has read the staging buffer and its DMA the pinned planes: in order on the stream, waited for below)
        PackPool::get().run(nitems, [&](size_t k) {
            const uint64_t a = (uint64_t)k * per, n = std::min(per, nc - a), b0 = (c0 + a) * IPCR_COLUMN_BASES;
            const uint64_t nb = b0 < len ? std::min<uint64_t>(len - b0, n * IPCR_COLUMN_BASES) : 0;
            iflags[k] = ipcr::pack_linear(seq + (nb ? b0 : 0), nb, n * IPCR_COLUMN_BASES, dlo + a * 128u, dlo + W + a * 128u, hiv + a * 128u, hiv + W + a * 128u);
        }, slot_phys(g->device));
        uint32_t fl = 0;
        for (uint32_t f : iflags) fl |= f;
        flags_all |= fl;
        bar_flush(bi);
        const bool lower = (fl & 2u) != 0, need_inv = lower || (fl & 1u) != 0;
        if (need_inv) bar_copy(g->staging + W * 8u, g->h_planes, W * 4u * (lower ? 2u : 1u)); // (through the BAR as well: no copy operation)
        bar_flush(bi);
        if (!hip(ipcr::launch_tiles_from_linear(g->stream, dlo, dlo + W, need_inv ? dlo + 2 * W : nullptr, lower ? dlo + 3 * W : nullptr, col0, col0 + c0, nc, len,
                                                g->planes, g->rst, nullptr, nullptr, c0 == 0 ? g->e0 : nullptr, c0 + nc >= cols ? g->e1 : nullptr), "tiles_from_linear")) break;
        if (c0 + nc < cols && !hip(hipStreamSynchronize(g->stream), "hipStreamSynchronize")) break; // the next group reuses both buffers
    }
    if (*st != IPCR_OK) { (void)hipStreamSynchronize(g->stream); return true; }
    const uint32_t rec = (uint32_t)g->rec_start.size();
    if (!hip(hipMemsetD32Async((hipDeviceptr_t)(g->d_flags + rec), (int)(flags_all & 1u), 1, g->stream), "hipMemsetD32Async")) return true;
    genome_account_record(g, len, cols);
    if (!hip(hipStreamSynchronize(g->stream), "hipStreamSynchronize")) return true; // the caller may free its bytes; the pinned planes are free again
    float ms = 0;
    if (hipEventElapsedTime(&ms, g->e0, g->e1) == hipSuccess) g->pack_ms += ms;
    return true;
}

// A FASTA file into a resident genome with the text packed on the host (fasta_hostpack.cpp) and the code planes written straight
// into device memory through the BAR: 0.25 bytes per base on the link instead of the device loader's 1.0125.  *taken = 0: not a
// file for this way in (no large BAR, no AVX-512 + BMI2, gzip / stdin, lines of several widths, blanks at line ends, thousands of
// records ...): nothing has been changed and the caller takes the device loader.
namespace {
struct FastaStage { // kept for the process's next load (allocating and mapping them is a few milliseconds)
    int phys = -1;
    uint64_t words = 0; // 64-bit words per plane
    uint64_t *d_lo = nullptr, *d_hi = nullptr, *d_iv = nullptr, *h_iv = nullptr;
    uint32_t *d_bits = nullptr; // one bit per column: its invalid plane has crossed the link (behind the three planes)
};
std::mutex g_fasta_stage_mu;
FastaStage g_fasta_stage;
std::atomic<uint64_t> g_fasta_hostpacked_loads{0};
} // namespace

ipcr_status ipcr_internal_genome_add_fasta_hostpacked(ipcr_genome *g, const char *path, uint32_t *n_added, std::string *ids, int *taken) {
    *taken = 0;
    if (!env_flag("IPCR_FASTA_HOSTPACK", true) || !env_flag("IPCR_CHUNK_BAR", true) || !ipcr::fasta_blocks_supported()) return IPCR_OK;
    if (getenv("IPCR_FASTA_SLAB")) return IPCR_OK; // (somebody is tuning or testing the device loader's slabs: that loader it is)
    const int phys = slot_phys(g->device);
    const BarInfo bi = device_bar(phys);
    if (!bi.writable) return IPCR_OK;
    const bool times = getenv("IPCR_DEBUG_TIMES") != nullptr;
    const auto tt0 = std::chrono::steady_clock::now();
    auto since0 = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tt0).count(); };
    ipcr::FastaText t;
    if (!t.open(path) || t.records.empty() || t.records.size() > 2048) return IPCR_OK;
    const double ms_open = since0();
    if (g->rec_start.size() + t.records.size() > g->max_records) return IPCR_OK; // (the device loader reports it)
    constexpr uint64_t GROUP = 256; // columns per group (1 Mb): the groups that hold an invalid base are looked over column by column
    std::vector<uint64_t> word0(t.records.size() + 1, 0), cols(t.records.size()), group0(t.records.size() + 1, 0);
    uint64_t total_cols = 0;
    for (size_t r = 0; r < t.records.size(); ++r) {
        cols[r] = record_cols(t.records[r].len);
        word0[r + 1] = word0[r] + cols[r] * 64u;
        group0[r + 1] = group0[r] + (cols[r] + GROUP - 1) / GROUP;
        total_cols += cols[r];
    }
    if (g->next_col + total_cols > g->cap_cols) return IPCR_OK;
    DeviceGuard dg(g->device);
    std::lock_guard<std::mutex> stage_lock(g_fasta_stage_mu);
    FastaStage &fs = g_fasta_stage;
    const uint64_t words = word0.back();
    if (fs.phys != phys || fs.words < words) {
        if (fs.d_lo) (void)hipFree(fs.d_lo);
        if (fs.h_iv) free(fs.h_iv);
        fs = FastaStage();
        const uint64_t want = words + (words >> 3);
        uint8_t *base = nullptr;
        // (+ the column bitmaps: a column is 64 words, a record's bitmap is padded to 64 bytes)
        if (hipExtMallocWithFlags((void **)&base, want * 24u + want / 8u + 2048u * 64u + 64u, hipDeviceMallocFinegrained) != hipSuccess) { (void)hipGetLastError(); return IPCR_OK; }
        fs.d_lo = reinterpret_cast<uint64_t *>(base);
        fs.d_hi = fs.d_lo + want;
        fs.d_iv = fs.d_hi + want;
        fs.d_bits = reinterpret_cast<uint32_t *>(fs.d_iv + want);
        fs.h_iv = static_cast<uint64_t *>(malloc(want * 8u));
        if (!fs.h_iv) { (void)hipFree(base); fs = FastaStage(); return IPCR_OK; }
        fs.phys = phys;
        fs.words = want;
    }
    const double ms_stage = since0();
    // ---- every record's text into its planes (the pool shares a record's pieces); nothing is registered before all of them are in
    std::vector<uint8_t> dirty((size_t)group0.back(), 0);
    for (size_t r = 0; r < t.records.size(); ++r)
        if (!t.pack(t.records[r], fs.d_lo + word0[r], fs.d_hi + word0[r], fs.h_iv + word0[r], cols[r] * 64u, dirty.data() + group0[r], GROUP))
            return IPCR_OK; // irregular text after all: the device loader takes the file
    const double ms_pack = since0();
    // ---- invalid bases: runs of N are short and far between, so of the groups that hold one only the COLUMNS whose invalid plane holds
    // a bit bring it along (512 bytes each, host memory -> device through the BAR as well); a bitmap per record tells the
    // conversion kernel which columns did -- it makes the others' bits itself
    std::vector<uint64_t> bit0(t.records.size() + 1, 0); // 32-bit words of the records' bitmaps, each padded to 64 bytes
    for (size_t r = 0; r < t.records.size(); ++r) bit0[r + 1] = bit0[r] + ((cols[r] + 31) / 32 + 15) / 16 * 16;
    std::vector<uint32_t> bits((size_t)bit0.back(), 0);
    std::vector<uint8_t> rec_dirty(t.records.size(), 0);
    {
        struct Span { size_t r; uint64_t c0, nc; };
        std::vector<Span> spans;
        for (size_t r = 0; r < t.records.size(); ++r)
            for (uint64_t q = group0[r]; q < group0[r + 1]; ++q)
                if (dirty[(size_t)q]) {
                    const uint64_t c0 = (q - group0[r]) * GROUP;
                    spans.push_back({r, c0, std::min(GROUP, cols[r] - c0)});
                    rec_dirty[r] = 1;
                }
        if (!spans.empty())
            PackPool::get().run(spans.size(), [&](size_t i) {
                const Span &sp = spans[i];
                const uint64_t *src = fs.h_iv + word0[sp.r];
                uint64_t *dst = fs.d_iv + word0[sp.r];
                for (uint64_t c = sp.c0; c < sp.c0 + sp.nc; ++c) { // (a group is 256 columns = 8 whole bitmap words: no word is shared)
                    uint64_t any = 0;
                    for (uint32_t k = 0; k < 64u; ++k) any |= src[c * 64u + k];
                    if (any) {
                        bits[(size_t)bit0[sp.r] + (size_t)(c >> 5)] |= 1u << (c & 31u);
                        bar_copy(reinterpret_cast<uint8_t *>(dst + c * 64u), reinterpret_cast<const uint8_t *>(src + c * 64u), 512u);
                    }
                }
            }, phys);
        for (size_t r = 0; r < t.records.size(); ++r)
            if (rec_dirty[r])
                bar_copy(reinterpret_cast<uint8_t *>(fs.d_bits + bit0[r]), reinterpret_cast<const uint8_t *>(bits.data() + bit0[r]), (bit0[r + 1] - bit0[r]) * 4u);
    }
    bar_flush(bi);
    // ---- tiles: one launch per record
    for (size_t r = 0; r < t.records.size(); ++r) {
        const ipcr::FastaRecord &fr = t.records[r];
        const uint64_t col0 = g->next_col;
        const uint32_t *lo32 = reinterpret_cast<const uint32_t *>(fs.d_lo + word0[r]), *hi32 = reinterpret_cast<const uint32_t *>(fs.d_hi + word0[r]),
                       *iv32 = reinterpret_cast<const uint32_t *>(fs.d_iv + word0[r]);
        const uint32_t any = rec_dirty[r];
        HIPCHK(ipcr::launch_tiles_from_linear(g->stream, lo32, hi32, any ? iv32 : nullptr, nullptr, col0, col0, cols[r], fr.len,
                                              g->planes, g->rst, nullptr, nullptr, nullptr, nullptr, any ? fs.d_bits + bit0[r] : nullptr));
        const uint32_t rec = (uint32_t)g->rec_start.size();
        HIPCHK(hipMemsetD32Async((hipDeviceptr_t)(g->d_flags + rec), (int)any, 1, g->stream));
        genome_account_record(g, fr.len, cols[r]);
        g->ids.back() = fr.id;
        if (n_added) ++*n_added;
        if (ids) { if (r) ids->push_back('\n'); *ids += fr.id; }
    }
    HIPCHK(hipStreamSynchronize(g->stream)); // the staging planes are free for the next load
    if (!env_flag("IPCR_FASTA_CACHE", true)) { // (kept for the process's next load otherwise: 0.5 bytes per base, a quarter of them host memory)
        (void)hipFree(fs.d_lo);
        free(fs.h_iv);
        fs = FastaStage();
    }
    if (times)
        fprintf(stderr, "fasta host packer: map + headers + plan %.2f ms, staging %.2f, pack %.2f, invalid planes + tiles %.2f (total %.2f ms, %zu records)\n",
                ms_open, ms_stage - ms_open, ms_pack - ms_stage, since0() - ms_pack, since0(), t.records.size());
    *taken = 1;
    g_fasta_hostpacked_loads.fetch_add(1, std::memory_order_relaxed);
    return IPCR_OK;
}
// tests: files loaded that way so far, in this process
uint64_t ipcr_internal_fasta_hostpacked_loads(void) { return g_fasta_hostpacked_loads.load(std::memory_order_relaxed); }

ipcr_status ipcr_genome_add_record(ipcr_genome *g, const uint8_t *seq, uint64_t len) {
    if (!g || (!seq && len)) return fail(IPCR_ERR_INVALID, "ipcr_genome_add_record: null argument");
    DeviceGuard dg(g->device);
    {
        ipcr_status hst = IPCR_OK;
        if (genome_add_host_packed(g, seq, len, &hst)) return hst;
    }
    if (len + 16 > g->staging_cap) {
        if (g->staging) (void)hipFree(g->staging);
        g->staging = nullptr;
        g->staging_cap = len + 16 + (len >> 3);
        HIPCHK(hipMalloc((void **)&g->staging, g->staging_cap));
    }
    if (len) HIPCHK(hipMemcpyAsync(g->staging, seq, len, hipMemcpyHostToDevice, g->stream));
    return genome_add_device(g, g->staging, len);
}

ipcr_status ipcr_genome_add_record_device(ipcr_genome *g, const void *dev_seq, uint64_t len) {
    if (!g || (!dev_seq && len)) return fail(IPCR_ERR_INVALID, "ipcr_genome_add_record_device: null argument");
    DeviceGuard dg(g->device);
    HIPCHK(hipDeviceSynchronize()); // the caller's producer may live on another stream
    return genome_add_device(g, static_cast<const uint8_t *>(dev_seq), len);
}

} // extern "C"

// hooks for fasta.cpp (the FASTA loader packs records it has normalised on the device)
ipcr_status ipcr_internal_genome_add_device(ipcr_genome *g, const uint8_t *dseq, uint64_t len, const char *id) {
    // no wait per record (a fragmented assembly has tens of thousands): the loader's copies and packs are
    // in order on the genome's stream, and every scan starts with genome_finalize, which waits for it
    const ipcr_status st = genome_add_device(g, dseq, len, false);
    if (st == IPCR_OK && id) g->ids.back() = id;
    return st;
}
hipStream_t ipcr_internal_genome_stream(ipcr_genome *g) { return g->stream; }
int ipcr_internal_genome_phys_device(const ipcr_genome *g) { return slot_phys(g->device); }
int ipcr_internal_slot_phys(int slot) { return slot_phys(slot); }
int ipcr_internal_slot_count() { return slot_count(); }
ipcr_status ipcr_internal_genome_add_batch(ipcr_genome *g, const uint8_t *dbase, const uint64_t *offs, const uint64_t *lens,
                                           const std::string *ids, size_t n, void *d_tmp, size_t tmp_bytes) {
    return genome_add_device_batch(g, dbase, offs, lens, n, d_tmp, tmp_bytes, ids);
}
size_t ipcr_internal_batch_table_bytes(size_t n) { return n * sizeof(ipcr_pack_rec) + (n + 1) * 4 + 64; }

extern "C" {

ipcr_status ipcr_lcg_fill_device(void *dev_out, uint64_t len, uint32_t seed, uint64_t stream_offset) {
    if (!dev_out && len) return fail(IPCR_ERR_INVALID, "ipcr_lcg_fill_device: null argument");
    if (len == 0) return IPCR_OK;
    hipPointerAttribute_t attr; // the buffer says which device it lives on
    HIPCHK(hipPointerGetAttributes(&attr, dev_out));
    int prev = -1;
    HIPCHK(hipGetDevice(&prev));
    if (prev != attr.device) HIPCHK(hipSetDevice(attr.device));
    struct Back { int d, cur; ~Back() { if (d != cur) (void)hipSetDevice(d); } } back{prev, attr.device};
    HIPCHK(ipcr::launch_lcg(nullptr, static_cast<uint8_t *>(dev_out), len, seed, stream_offset));
    HIPCHK(hipDeviceSynchronize());
    return IPCR_OK;
}

ipcr_status ipcr_genome_read(const ipcr_genome *g, uint32_t record, uint64_t pos, uint8_t *out, uint64_t len) {
    if (!g || !out) return fail(IPCR_ERR_INVALID, "ipcr_genome_read: null argument");
    if (record >= g->rec_start.size() || pos + len > g->rec_len[record]) return fail(IPCR_ERR_INVALID, "ipcr_genome_read: range outside record");
    if (len == 0) return IPCR_OK;
    DeviceGuard dg(g->device);
    uint8_t *d = nullptr;
    HIPCHK(hipMalloc((void **)&d, len));
    hipError_t e = ipcr::launch_unpack(g->stream, g->planes, g->rst, g->rec_start[record] + pos, len, d);
    if (e == hipSuccess) e = hipMemcpyAsync(out, d, len, hipMemcpyDeviceToHost, g->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(g->stream);
    (void)hipFree(d);
    HIPCHK(e);
    return IPCR_OK;
}

uint32_t ipcr_genome_num_records(const ipcr_genome *g) { return g ? (uint32_t)g->rec_start.size() : 0; }
uint64_t ipcr_genome_record_len(const ipcr_genome *g, uint32_t r) { return (g && r < g->rec_len.size()) ? g->rec_len[r] : 0; }
const char *ipcr_genome_record_id(const ipcr_genome *g, uint32_t r) { return (g && r < g->ids.size()) ? g->ids[r].c_str() : ""; }
uint64_t ipcr_genome_total_bases(const ipcr_genome *g) { return g ? g->total_bases : 0; }
uint64_t ipcr_genome_tile_bytes(const ipcr_genome *g) { // bytes the filter kernel streams: whole blocks of lo/hi/inv
    return g ? ((g->next_col + 63) / 64) * IPCR_BLOCK_PLANE_WORDS * 4ull : 0;
}
double ipcr_genome_pack_ms(const ipcr_genome *g) { return g ? g->pack_ms : 0; }

uint8_t ipcr_genome_record_flags(const ipcr_genome *g, uint32_t record) {
    if (!g || record >= g->rec_start.size()) return 0;
    if (!g->flags_valid) {
        DeviceGuard dg(g->device);
        if (genome_finalize(const_cast<ipcr_genome *>(g)) != IPCR_OK) return 0;
    }
    return (uint8_t)((g->flags[record] & 1u) | (genome_any_reset(g) ? 2u : 0u));
}

} // extern "C"

// --------------------------------------------------------------------------- scratch

struct ipcr_scratch {
    const ipcr_panel *panel = nullptr;
    int device = 0; // device slot
    SetDev *sdev[2] = {nullptr, nullptr}; // the panel's tables and kernels on this slot, per scan mode (panel_upload)
    hipStream_t stream = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr}; // filter start/stop, verify start/stop
    ipcr_queue_entry *d_queue = nullptr; // IPCR_QUEUE_SHARDS segments of qcap entries
    uint64_t qcap = 0;                   // capacity of ONE segment
    unsigned long long *d_qcounts = nullptr; // 2 sets x IPCR_QUEUE_SHARDS counters, 128 B apart
    uint64_t prefix_hint = 256; // hits copied back together with the counters
    ipcr_hit_rec *d_hits = nullptr;
    uint64_t hcap = 0;
    // two alternating counter sets ([0] queue entries, [1] hits, [2] candidate windows) live in the
    // 64 bytes in FRONT of the hit records (d_hitbuf): one D2H copy brings counters + first hits
    // back, and each scan's verify kernel clears the set of the next scan (no memset launch)
    unsigned long long *d_counts = nullptr; // set 0; set 1 = d_counts + 4
    void *d_hitbuf = nullptr;
    uint32_t cset = 0;
    // pipelined scans (ipcr_scratch_chain_after): the scans of a chain of scratches are enqueued on ONE
    // in-order stream (the "lane" = the stream of the chain's first scratch), so scan i+1's sweep starts
    // the moment scan i's kernels and read-back are done, with no cross-stream dependency (tens of
    // microseconds each on this runtime) in front of the bandwidth-bound kernel
    struct Lane {
        hipStream_t s = nullptr;
        ~Lane() { if (s) (void)hipStreamDestroy(s); }
    };
    std::shared_ptr<Lane> own_lane;   // holds `stream`
    std::shared_ptr<Lane> lane_next;  // lane of the next scan (null: own)
    std::shared_ptr<Lane> lane_used;  // lane the last scan ran on
    hipEvent_t ev_done = nullptr;     // recorded behind a scan's read-back when it runs on a shared lane
    hipStream_t cstream = nullptr;    // fetches the hit records of a published scan (never waits behind a sweep)
    struct Pending { // a scan enqueued by scan_enqueue and not yet collected
        bool active = false, empty = false;
        int mode = 0;
        bool fused = false;    // the filter verified its own survivors (specialised kernel)
        bool verified = false; // the stand-alone verifier has run over the queue
        bool on_lane = false;  // wait for ev_done (false: a follow-up on the private stream, wait for the stream)
        bool published = false; // the filter's last wave writes counters + sequence word to pinned memory: poll
        bool times_pending = false;
        uint32_t nrec = 0, check_rst = 0, cset_used = 0;
        uint64_t nblocks = 0, pre = 0;
        uint64_t block0 = 0;   // the launch sweeps blocks [block0, block0 + nblocks): the whole genome unless ...
        bool segment = false;  // ... the scan runs in segments (scan_segmented)
        std::chrono::steady_clock::time_point t0;
    } pend;
    void *pinned = nullptr;                 // two counter sets (64 B) + first PREFIX hits + the sequence word
    uint32_t *d_tickets = nullptr;          // 65 counters, 128 B apart (specialised filter: last wave publishes)
    uint32_t seq = 0;                       // id of the last scan launched on this scratch
    uint64_t seg_overflow = 0;       // scan_segmented: raw hits of a segment that did not fit
    std::vector<ipcr_hit> hits;      // sorted by (record, pattern, pos)
    std::vector<ipcr_hit> hits_raw;  // in device append order
    std::vector<uint32_t> bucket;    // scratch of sort_hits
    std::vector<ipcr_product> products;
    std::vector<uint64_t> last_rec_len; // of the last scanned genome (for probe)
    std::vector<uint64_t> last_rec_start;
    ipcr_scan_stats stats{};
    std::vector<ipcr_chunk_window> windows; // of the last ipcr_scan_genome_chunked
    bool products_in_windows = false;       // ... whose products are the current ones: `record` = window, coordinates window-local
    ipcr_genome *chunk = nullptr; // private genome of ipcr_scan_chunk
    bool dev_hits_stale = false;  // the device hit buffer does NOT hold the last scan's hits (scan_segmented: only its last range): hits_raw does
    bool last_was_chunk = false;  // the products in `products` are those of an ipcr_scan_chunk: their amplicons lie in `chunk`
    // ipcr_scan_chunk with several workers: the caller's (pageable) bytes go through two pinned slices of this
    // scratch -- the CPU copy of slice i+1 runs under the DMA of slice i, and workers do not meet in the runtime's
    // own pageable-copy path
    uint8_t *h_stage[2] = {nullptr, nullptr};
    hipEvent_t ev_stage[2] = {nullptr, nullptr};
    uint8_t *h_planes = nullptr; // a whole record's host-packed planes (one worker, large record: packed by the process's pool)
    uint64_t h_planes_cap = 0;
    std::shared_ptr<std::atomic<int>> counted_in; // the panel's worker count this scratch is part of
    // probe buffers
    uint8_t *d_amps = nullptr; uint64_t amps_cap = 0;
    ipcr_genome *nest = nullptr; // amplicons of a nested-PCR batch, one record each (ipcr_nested_windows)
    void *d_probe_misc = nullptr; uint64_t probe_misc_cap = 0;
    // ipcr_probe_products: its own lane, and ONE pinned block the two kernels read their arguments from and write the
    // results to (no copy operation in either direction)
    hipStream_t probe_stream = nullptr;
    uint8_t *h_probe = nullptr; uint64_t h_probe_cap = 0;
    uint32_t probe_tag = 0;         // the rescan in flight marks its records with it (the host spins on the tags in h_probe)
    hipStream_t probe_on = nullptr; // the stream it was queued on
    int64_t probe_pending = -1;     // ipcr_probe_products_begin: products whose rescan is in flight (-1: none)
    uint64_t probe_res_off = 0;     // ... and where in h_probe its results arrive
};

namespace {

constexpr uint64_t PREFIX_HITS = 65536;
constexpr uint64_t QCAP_INIT = 1ull << 13;  // per segment: 256 x 8 Ki surviving words (32 MiB)
constexpr uint64_t HCAP_INIT = 1ull << 20;  // 1 Mi hits (32 MiB)
constexpr uint64_t QCAP_MAX = 1ull << 23;   // per segment (2 Gi words in all)
constexpr uint64_t HCAP_MAX = 1ull << 28;
// With a hit cap, only the first HitCap matches per orientation and record can matter (core/primer/match.go:86-88,
// core/engine/hit_collect.go:80-82): a scan whose raw matches exceed this many records is not regrown any further but
// repeated in position order, segment by segment (scan_segmented).  IPCR_TEST_HCAP_SOFT: tests.
uint64_t hcap_soft() {
    const char *e = getenv("IPCR_TEST_HCAP_SOFT"); // (read at every overflow: a rare path)
    return e && *e ? std::max<uint64_t>(64, strtoull(e, nullptr, 10)) : (1ull << 24);
}

ipcr_status panel_upload(const ipcr_panel *cp, int mode, int slot, SetDev **out) {
    ipcr_panel *p = const_cast<ipcr_panel *>(cp);
    std::lock_guard<std::mutex> lock(p->mu);
    std::unique_ptr<PanelDev> &pd = p->devs[slot];
    if (!pd) { pd.reset(new PanelDev); pd->slot = slot; }
    PatternSet &s = p->set[mode];
    SetDev &d = pd->set[mode];
    *out = &d;
    if (!d.dev && !s.host.empty()) {
        HIPCHK(hipMalloc((void **)&d.dev, s.host.size() * sizeof(ipcr_dev_pattern)));
        HIPCHK(hipMemcpy(d.dev, s.host.data(), s.host.size() * sizeof(ipcr_dev_pattern), hipMemcpyHostToDevice));
    }
    const bool force_index = getenv("IPCR_FORCE_INDEX") && atoi(getenv("IPCR_FORCE_INDEX")) != 0;
    if (p->specialize && !d.jit_tried && !s.host.empty()) {
        d.jit_tried = true;
        // hiprtc takes ~0.8 s for a C2-sized panel; the table-driven kernel scans such a panel at ~1.6 ms per pattern and 3 Gb (four patterns per walk).
        // So a SMALL panel's first scans do not wait: the kernels are built on a thread of their own, the scans that come
        // before they are ready take the table-driven kernel (same results: both are parity-tested against the oracle), and
        // the first panel_upload after the build publishes them.  A cold `ipcr` run on a 3 Gb genome has its products after
        // ~20 ms instead of ~0.8 s.  Larger panels wait (their table-driven scan would cost more than the build).
        // IPCR_JIT_ASYNC=0: always wait (tests, measurements); ipcr_panel_wait_ready: wait now.
        static const bool async_on = env_flag("IPCR_JIT_ASYNC", true);
        const bool async = async_on && !force_index && s.host.size() <= 16;
        if (!force_index && !async) {
            d.jit = ipcr::jit_build(s.host, p->cfg.max_mm, s.jit_error);
            d.jit_pub.store(&d.jit, std::memory_order_release);
        } else if (async) {
            d.jit_state.store(1);
            const int max_mm = p->cfg.max_mm;
            SetDev *dp = &d;
            PatternSet *sp = &s;
            d.jit_thread = std::thread([dp, sp, max_mm, slot] {
                DeviceGuard dgt(slot); // the code objects are loaded onto the slot's device
                std::string err;
                dp->jit = ipcr::jit_build(sp->host, max_mm, err);
                if (dp->jit.empty()) sp->jit_error = err; // (read only after the join / state 2)
                dp->jit_state.store(2, std::memory_order_release);
            });
        }
    }
    if (d.jit_state.load(std::memory_order_acquire) == 2) { // the background build has finished: scans from now on use its kernels
        if (d.jit_thread.joinable()) d.jit_thread.join();
        d.jit_pub.store(&d.jit, std::memory_order_release);
        d.jit_state.store(0);
    }
    const bool building = d.jit_state.load() == 1;
    // panels too large to specialise: seed-index filter (+ table-driven kernel for what it cannot key)
    if (p->specialize && !building && d.jit.empty() && !d.index_tried && !s.host.empty() && p->cfg.max_mm <= 3) {
        d.index_tried = true;
        if (!s.index.built) build_index(*p, s);
        IndexPlan &ix = s.index;
        if (ix.usable) {
            std::string jerr;
            d.index_jit = ipcr::jit_build_index(ix.shapes, ix.geom(), jerr);
            if (!d.index_jit) { // no hiprtc: the table-driven kernel serves
                if (env_flag("IPCR_INDEX_DEBUG", false)) fprintf(stderr, "ipcr index kernel: %s\n", jerr.c_str());
                return IPCR_OK;
            }
            HIPCHK(hipMalloc((void **)&d.d_lds_image, ix.lds_image.size() * 4));
            HIPCHK(hipMemcpy(d.d_lds_image, ix.lds_image.data(), ix.lds_image.size() * 4, hipMemcpyHostToDevice));
            HIPCHK(hipMalloc((void **)&d.d_table, ix.table.size() * sizeof(ipcr_index_entry)));
            HIPCHK(hipMemcpy(d.d_table, ix.table.data(), ix.table.size() * sizeof(ipcr_index_entry), hipMemcpyHostToDevice));
            if (!ix.leftover.empty()) {
                // What the index cannot key (primers > 32 nt, too many IUPAC expansions in a key): a handful of patterns
                // in practice.  The table-driven kernel costs ~1.6 ms per pattern and 3 Gb; specialised spill-only filters
                // (their survivors join the index's in the candidate queue) cost one ~0.25 ms sweep per 12 patterns.
                std::string lerr;
                d.leftover_jit = ipcr::jit_build(s.host, p->cfg.max_mm, lerr, &ix.leftover);
                if (d.leftover_jit.empty()) {
                    HIPCHK(hipMalloc((void **)&d.d_leftover, ix.leftover.size() * 4));
                    HIPCHK(hipMemcpy(d.d_leftover, ix.leftover.data(), ix.leftover.size() * 4, hipMemcpyHostToDevice));
                }
            }
        }
    }
    return IPCR_OK;
}

} // namespace

// tests: launches of the specialised filter that took its form for small launches, so far in this process
extern "C" uint64_t ipcr_internal_small_launches(void) { return ipcr::jit_small_launches(); }

// tests, chunk_workers: how ipcr_scan_chunk's packer reaches device slot `slot` -- 0: pinned slabs + DMA; 1: through the BAR,
// flush register unknown (IPCR_CHUNK_BAR=2 only); 2: through the BAR, with the HDP flush in front of every launch
extern "C" int32_t ipcr_internal_device_bar(int32_t slot) {
    if (slot < 0 || slot >= slot_count()) return 0;
    const BarInfo bi = device_bar(slot_phys(slot));
    return !bi.writable || !env_flag("IPCR_CHUNK_BAR", true) ? 0 : (bi.hdp_flush ? 2 : 1);
}

// fasta.cpp: its file reads run on the same threads (threads started per slab slowed the slab copies, see there)
void ipcr_internal_pool_run(size_t n, const std::function<void(size_t)> &fn, int phys) { PackPool::get().run(n, [&](size_t i) { fn(i); }, phys); }
bool ipcr_internal_bind_thread(int phys) { return bind_this_thread(phys); }
unsigned ipcr_internal_pool_size() { return PackPool::get().size(); }

namespace {

struct HitLess {
    bool operator()(const ipcr_hit &a, const ipcr_hit &b) const {
        if (a.record != b.record) return a.record < b.record;
        const uint32_t pa = a.pattern & 0x7FFFFFFFu, pb = b.pattern & 0x7FFFFFFFu;
        if (pa != pb) return pa < pb;
        return a.pos < b.pos;
    }
};

// The seed-index filter can report one window through several of its keys: keep one hit per
// (record, pattern, position).
void dedup_sorted_hits(std::vector<ipcr_hit> &v) {
    size_t w = 0;
    for (size_t i = 0; i < v.size(); ++i) {
        if (w && v[w - 1].record == v[i].record && v[w - 1].pos == v[i].pos &&
            (v[w - 1].pattern & 0x7FFFFFFFu) == (v[i].pattern & 0x7FFFFFFFu))
            continue;
        v[w++] = v[i];
    }
    v.resize(w);
}

// order inside one (record, pattern): by position; records of one window (the seed index may report it through several keys)
// are identical, but the order is total all the same, so that every form of the sort keeps the same one of them
inline bool hit_pos_less(const ipcr_hit &x, const ipcr_hit &y) {
    if (x.pos != y.pos) return x.pos < y.pos;
    if (x.mm_mask[0] != y.mm_mask[0]) return x.mm_mask[0] < y.mm_mask[0];
    if (x.mm_mask[1] != y.mm_mask[1]) return x.mm_mask[1] < y.mm_mask[1];
    return x.pattern < y.pattern;
}

// A large list (a 1024-row panel at k = 3: 365 000 hits, 6 ms on one thread) is sorted by the process's pool: the input is
// cut into slices, every slice is counted by record, the slices are scattered into the records' ranges (each slice has its
// own place in every range), and the records are sorted -- (pattern, position) -- side by side.  Same result as sort_hits.
bool sort_hits_parallel(const std::vector<ipcr_hit> &in, std::vector<ipcr_hit> &out, uint32_t nrec, uint32_t npat) {
    const size_t n = in.size();
    const size_t T = std::min<size_t>(PackPool::get().size(), 16);
    if (T < 2 || nrec < 2 || nrec > 65536) return false;
    for (const ipcr_hit &h : in)
        if (h.record >= nrec || (h.pattern & 0x7FFFFFFFu) >= npat) return false; // (the one-thread form handles stray records)
    std::vector<uint32_t> cnt(T * (size_t)nrec, 0);
    const size_t per = (n + T - 1) / T;
    PackPool::get().run(T, [&](size_t t) {
        uint32_t *c = cnt.data() + t * nrec;
        for (size_t i = t * per, e = std::min(n, (t + 1) * per); i < e; ++i) ++c[in[i].record];
    });
    std::vector<uint64_t> rec_begin((size_t)nrec + 1, 0);
    std::vector<uint64_t> place(T * (size_t)nrec, 0); // where slice t's hits of record r go
    uint64_t run = 0;
    for (uint32_t r = 0; r < nrec; ++r) {
        rec_begin[r] = run;
        for (size_t t = 0; t < T; ++t) { place[t * nrec + r] = run; run += cnt[t * nrec + r]; }
    }
    rec_begin[nrec] = run;
    out.resize(n);
    PackPool::get().run(T, [&](size_t t) {
        uint64_t *pl = place.data() + t * nrec;
        for (size_t i = t * per, e = std::min(n, (t + 1) * per); i < e; ++i) out[pl[in[i].record]++] = in[i];
    });
    PackPool::get().run(nrec, [&](size_t r) { // one record: counting sort by pattern, then every small bucket by position
        static thread_local std::vector<uint32_t> pc;
        static thread_local std::vector<ipcr_hit> tmp;
        ipcr_hit *base = out.data() + rec_begin[r];
        const size_t m = (size_t)(rec_begin[r + 1] - rec_begin[r]);
        if (m < 2) return;
        pc.assign((size_t)npat + 1, 0);
        for (size_t i = 0; i < m; ++i) ++pc[(base[i].pattern & 0x7FFFFFFFu) + 1];
        for (uint32_t q = 0; q < npat; ++q) pc[q + 1] += pc[q];
        tmp.resize(m);
        {
            std::vector<uint32_t> &cur = pc; // (consumed as the write cursor; the bucket ends are the next bucket's start afterwards)
            for (size_t i = 0; i < m; ++i) tmp[cur[base[i].pattern & 0x7FFFFFFFu]++] = base[i];
        }
        // pc[q] is now the END of bucket q = the start of bucket q + 1
        size_t b = 0;
        for (uint32_t q = 0; q < npat; ++q) {
            const size_t e = pc[q];
            if (e - b > 1) std::sort(tmp.begin() + (long)b, tmp.begin() + (long)e, hit_pos_less);
            b = e;
        }
        memcpy(base, tmp.data(), m * sizeof(ipcr_hit));
    });
    return true;
}

// Hits come back in atomic-append order; the join wants them grouped by (record, pattern) and
// ascending in position.  Counting sort over the (record, pattern) buckets, then each small
// bucket by position -- O(n) instead of a comparison sort of 32-byte records.
void sort_hits(const std::vector<ipcr_hit> &in, std::vector<ipcr_hit> &out, uint32_t nrec, uint32_t npat) {
    if (in.size() >= 65536 && env_flag("IPCR_JOIN_PARALLEL", true) && sort_hits_parallel(in, out, nrec, npat)) {
        dedup_sorted_hits(out);
        return;
    }
    static thread_local std::vector<uint32_t> cnt, cur;
    const size_t n = in.size();
    out.resize(n);
    if (n == 0) return;
    const uint64_t nb = (uint64_t)nrec * npat;
    bool ok = nb > 0 && nb <= (1u << 22);
    if (ok)
        for (const ipcr_hit &h : in)
            if (h.record >= nrec || (h.pattern & 0x7FFFFFFFu) >= npat) { ok = false; break; }
    if (!ok) {
        out = in;
        std::sort(out.begin(), out.end(), HitLess());
        dedup_sorted_hits(out);
        return;
    }
    cnt.assign(nb + 1, 0);
    for (const ipcr_hit &h : in) ++cnt[(uint64_t)h.record * npat + (h.pattern & 0x7FFFFFFFu) + 1];
    for (uint64_t i = 0; i < nb; ++i) cnt[i + 1] += cnt[i];
    cur.assign(cnt.begin(), cnt.end() - 1);
    for (const ipcr_hit &h : in) out[cur[(uint64_t)h.record * npat + (h.pattern & 0x7FFFFFFFu)]++] = h;
    for (uint64_t i = 0; i < nb; ++i) {
        const uint32_t b = cnt[i], e = cnt[i + 1];
        if (e - b > 1) std::sort(out.begin() + b, out.begin() + e, hit_pos_less);
    }
    dedup_sorted_hits(out);
}

// IPCR_DEBUG_TIMES=1: host-side timeline of the scan calls on stderr
void trace(const char *what, const void *s) {
    static const bool on = getenv("IPCR_DEBUG_TIMES") != nullptr;
    static const auto t0 = std::chrono::steady_clock::now();
    if (on) fprintf(stderr, "[t %10.1f us] %p %s\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(), s, what);
}

double ms_since(std::chrono::steady_clock::time_point a) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count();
}

bool env_flag(const char *name, bool dflt) {
    const char *v = getenv(name);
    return (v && *v) ? atoi(v) != 0 : dflt;
}

uint32_t *pinned_seq(const ipcr_scratch *s) {
    return reinterpret_cast<uint32_t *>(static_cast<char *>(s->pinned) + 64 + PREFIX_HITS * sizeof(ipcr_hit));
}

std::atomic<bool> g_publish_broken{false}; // a sweep ended without its results reaching pinned memory: copy path from then on

bool publish_enabled() { // IPCR_PUBLISH=0: read-back by a copy operation behind the kernel instead
    static const bool on = env_flag("IPCR_PUBLISH", true);
    return on && !g_publish_broken.load(std::memory_order_relaxed);
}

// the specialised filter's last wave has written counters, first hits and finally the scan's
// sequence word into pinned memory: spin on that word (microseconds after the kernel's last wave,
// no marker packet between two chained sweeps)
ipcr_status wait_published(ipcr_scratch *s) {
    volatile uint32_t *w = pinned_seq(s);
    const uint32_t want = s->seq;
    const hipStream_t lane = s->lane_used ? s->lane_used->s : s->stream;
    const auto t0 = std::chrono::steady_clock::now();
    // IPCR_WAIT_SLEEP_US=n: after IPCR_WAIT_SPIN_US of spinning, sleep n us between two looks.  For a pool of MORE workers than
    // the process has CPUs (a cgroup quota counts spinning as work): 24 / 48 workers on 16 CPUs 78 / 47 -> 122 / 83-95 Gbases/s
    // with n = 20 (chunks over DMA; 16 workers: 137-144 either way).  Off by default: with as many workers as CPUs it changes nothing
    static const int sleep_us = getenv("IPCR_WAIT_SLEEP_US") ? atoi(getenv("IPCR_WAIT_SLEEP_US")) : 0;
    static const int spin_us = getenv("IPCR_WAIT_SPIN_US") ? atoi(getenv("IPCR_WAIT_SPIN_US")) : 30;
    if (sleep_us > 0) {
        static thread_local bool slack_set = false;
        if (!slack_set) { (void)prctl(PR_SET_TIMERSLACK, 1000UL, 0, 0, 0); slack_set = true; } // (the default slack of 50 us would round every sleep up)
    }
    for (uint64_t spin = 1;; ++spin) {
        if (__atomic_load_n(w, __ATOMIC_ACQUIRE) == want) return IPCR_OK;
        if (sleep_us > 0 && (spin & 63u) == 0 && ms_since(t0) * 1000.0 > (double)spin_us) {
            struct timespec ts = {0, sleep_us * 1000L};
            (void)nanosleep(&ts, nullptr);
        }
        if (spin < 0x10000u) __builtin_ia32_pause(); // the first millisecond (a sweep takes 0.2 ms): pure spin
        else std::this_thread::yield();              // long scans (large panels, huge genomes): let other workers run
        if ((spin & 0xFFFFu) == 0) { // a fault on the stream would otherwise spin for ever
            const hipError_t q = hipStreamQuery(lane);
            if (q == hipSuccess) { // stream drained: the word is there, or device stores do not reach this host's pinned memory
                if (__atomic_load_n(w, __ATOMIC_ACQUIRE) == want) return IPCR_OK;
                return IPCR_ERR_UNSUPPORTED; // caller: switch to the copy path and rescan
            }
            if (q != hipErrorNotReady) return fail(IPCR_ERR_DEVICE, "HIP: %s", hipGetErrorString(q));
            if (ms_since(t0) > 120000.0) return fail(IPCR_ERR_DEVICE, "scan did not finish within 120 s");
        }
    }
}

ipcr_status scan_segmented(const ipcr_panel *p, ipcr_scratch *s, ipcr_genome *g, uint64_t seen_hits);
inline bool hit_has_idx_below(const ipcr_hit *h, int tw);

// enqueue one attempt: the specialised filter alone (it verifies and publishes itself), or the seed-index /
// table-driven filter + verify kernel + read-back of counters and first hits behind a marker event
ipcr_status scan_launch(const ipcr_panel *p, ipcr_scratch *s, ipcr_genome *g) {
    ipcr_scratch::Pending &pd = s->pend;
    const PatternSet &set = p->set[pd.mode];
    const SetDev &sd = *s->sdev[pd.mode]; // this slot's tables and kernels
    static const std::vector<ipcr::JitFilter *> none;
    const std::vector<ipcr::JitFilter *> *pub = sd.jit_pub.load(std::memory_order_acquire);
    const std::vector<ipcr::JitFilter *> &jit = pub ? *pub : none; // (empty while a small panel's kernels are still being built)
    const auto te = std::chrono::steady_clock::now();
    unsigned long long *cnt = s->d_counts + 4u * s->cset, *cnt_next = s->d_counts + 4u * (s->cset ^ 1u);
    const uint64_t qset = (uint64_t)IPCR_QUEUE_SHARDS * IPCR_QUEUE_COUNTER_STRIDE;
    unsigned long long *qc = s->d_qcounts + qset * s->cset, *qc_next = s->d_qcounts + qset * (s->cset ^ 1u);
    pd.cset_used = s->cset;
    s->cset ^= 1u;
    const uint64_t nblocks = pd.nblocks, block0 = pd.block0;
    trace("launch>", s);
    const hipStream_t own = s->stream;
    const hipStream_t lane = s->lane_next ? s->lane_next->s : own;
    s->lane_used = s->lane_next ? std::move(s->lane_next) : s->own_lane;
    s->lane_next.reset();
    pd.fused = false;
    pd.verified = false;
    pd.published = false;
    if (!jit.empty()) {
        ipcr::JitVerify v;
        v.rst = g->rst;
        v.pats = sd.dev;
        v.rec_start = g->d_rec_start;
        v.block_rec = g->d_block_rec;
        v.rec_len = g->d_rec_len;
        v.nrec = pd.nrec;
        v.max_mm = (uint32_t)p->cfg.max_mm;
        v.check_rst = pd.check_rst;
        v.hits = s->d_hits;
        v.hcap = s->hcap;
        v.counts = cnt;
        pd.pre = std::min<uint64_t>(PREFIX_HITS, s->hcap);
        ++s->seq;
        for (size_t gi = 0; gi < jit.size(); ++gi) { // every group streams the tiles once
            v.next_counts = gi == 0 ? cnt_next : nullptr;
            v.next_qcount = gi == 0 ? qc_next : nullptr;
            if (publish_enabled()) { // every kernel of the scan writes its first hits to the pinned buffer too
                v.seq = s->seq;
                v.tickets = s->d_tickets;
                v.pub_hits = reinterpret_cast<ipcr_hit_rec *>(static_cast<unsigned long long *>(s->pinned) + 8);
                v.pre = (uint32_t)pd.pre;
            }
            if (gi + 1 == jit.size() && publish_enabled()) { // the scan's last kernel hands the results over itself
                v.tickets = s->d_tickets;
                v.pub = static_cast<unsigned long long *>(s->pinned) + 4u * pd.cset_used;
                v.pub_seq = pinned_seq(s);
                static const bool break_it = env_flag("IPCR_TEST_BREAK_PUBLISH", false); // tests: the word never reaches the host
                if (break_it) v.pub_seq = s->d_tickets + 2064;
                v.seq = s->seq;
                pd.published = true;
            }
            {   // tests: IPCR_TEST_WITHHOLD_TAG=<slot+1>: that record's first half carries a stale tag (a torn record)
                static const int withhold = [] {
                    const int w = getenv("IPCR_TEST_WITHHOLD_TAG") ? atoi(getenv("IPCR_TEST_WITHHOLD_TAG")) : 0;
                    if (w > 0) fprintf(stderr, "ipcr_hip: IPCR_TEST_WITHHOLD_TAG=%d is set: hit record %d of every scan is published torn on purpose "
                                               "(a test hook; every scan then waits out the hand-over deadline and refetches)\n", w, w - 1);
                    return w;
                }();
                v.withhold = withhold > 0 ? (uint32_t)withhold : 0u;
            }
            HIPCHK(ipcr::jit_launch(jit[gi], lane, g->planes, block0, nblocks, s->d_queue, s->qcap, qc, v,
                                    gi == 0 ? s->ev[0] : nullptr, gi + 1 == jit.size() ? s->ev[1] : nullptr));
        }
        s->stats.kernel_kind = 1;
        pd.fused = true;
    } else if (sd.index_jit) {
        const IndexPlan &ix = set.index;
        const bool more = !ix.leftover.empty();
        // The index's exact check serves every pattern of this panel: its lanes write the hit records themselves and the last
        // wave to leave publishes the counters -- ONE kernel per scan, as with the specialised filter (no candidate queue, no
        // verify kernel, no copy operation: a chunk's scan under a large panel is a conversion launch and a sweep).
        const bool fuse = !more && publish_enabled() && ipcr::jit_index_fusable();
        ipcr::JitVerify fv;
        if (fuse) {
            fv.rst = g->rst;
            fv.pats = sd.dev;
            fv.rec_start = g->d_rec_start;
            fv.block_rec = g->d_block_rec;
            fv.rec_len = g->d_rec_len;
            fv.nrec = pd.nrec;
            fv.max_mm = (uint32_t)p->cfg.max_mm;
            fv.check_rst = pd.check_rst;
            fv.hits = s->d_hits;
            fv.hcap = s->hcap;
            fv.counts = cnt;
            fv.next_counts = cnt_next;
            fv.next_qcount = qc_next;
            pd.pre = std::min<uint64_t>(PREFIX_HITS, s->hcap);
            ++s->seq;
            fv.seq = s->seq;
            fv.pub = static_cast<unsigned long long *>(s->pinned) + 4u * pd.cset_used;
            fv.pub_hits = reinterpret_cast<ipcr_hit_rec *>(static_cast<unsigned long long *>(s->pinned) + 8);
            fv.pre = (uint32_t)pd.pre;
            fv.pub_seq = pinned_seq(s);
            static const bool break_it = env_flag("IPCR_TEST_BREAK_PUBLISH", false); // tests: the word never reaches the host
            if (break_it) fv.pub_seq = s->d_tickets + 2064;
        }
        HIPCHK(ipcr::jit_launch_index(sd.index_jit, lane, g->planes, block0, nblocks, (uint32_t)ix.shapes.size(), sd.d_lds_image, sd.d_table,
                                      (uint32_t)p->cfg.max_mm, s->d_queue, s->qcap, qc, s->d_tickets, s->ev[0], more ? nullptr : s->ev[1],
                                      fuse ? &fv : nullptr));
        if (fuse) { pd.fused = true; pd.published = true; }
        if (more && !sd.leftover_jit.empty()) { // patterns the index cannot key: specialised filters that only fill the queue
            ipcr::JitVerify v;
            v.rst = g->rst;
            v.pats = sd.dev;
            v.rec_start = g->d_rec_start;
            v.block_rec = g->d_block_rec;
            v.rec_len = g->d_rec_len;
            v.nrec = pd.nrec;
            v.max_mm = (uint32_t)p->cfg.max_mm;
            v.check_rst = pd.check_rst;
            v.hits = s->d_hits;
            v.hcap = s->hcap;
            v.counts = cnt;
            for (size_t gi = 0; gi < sd.leftover_jit.size(); ++gi)
                HIPCHK(ipcr::jit_launch(sd.leftover_jit[gi], lane, g->planes, block0, nblocks, s->d_queue, s->qcap, qc, v, nullptr,
                                        gi + 1 == sd.leftover_jit.size() ? s->ev[1] : nullptr));
        } else if (more)
            HIPCHK(ipcr::launch_filter_generic(lane, g->planes, block0, nblocks, sd.dev, (uint32_t)ix.leftover.size(),
                                               (uint32_t)p->cfg.max_mm, sd.d_leftover, s->d_queue, s->qcap, qc,
                                               nullptr, s->ev[1]));
        s->stats.kernel_kind = 3;
        s->stats.leftover_patterns = (uint32_t)ix.leftover.size();
        s->stats.leftover_kernels = (uint32_t)sd.leftover_jit.size();
    } else {
        HIPCHK(ipcr::launch_filter_generic(lane, g->planes, block0, nblocks, sd.dev, (uint32_t)set.ids.size(),
                                           (uint32_t)p->cfg.max_mm, nullptr, s->d_queue, s->qcap, qc, s->ev[0], s->ev[1]));
        s->stats.kernel_kind = 2;
    }
    if (!pd.fused) {
        // (the seed index files one window per queue entry; its leftover filters and the table-driven kernel 32 strands)
        const uint32_t single = s->stats.kernel_kind == 3 && s->stats.leftover_kernels == 0 && s->stats.leftover_patterns == 0 ? 1u : 0u;
        HIPCHK(ipcr::launch_verify(lane, g->planes, g->rst, sd.dev, (uint32_t)p->cfg.max_mm, g->d_rec_start,
                                   g->d_rec_len, pd.nrec, pd.check_rst, s->d_queue, s->qcap, qc, s->d_hits,
                                   s->hcap, cnt + 1, cnt + 2, cnt_next, qc_next, s->ev[2], s->ev[3], single));
        pd.verified = true;
    }
    if (!pd.published) {
        pd.pre = std::min<uint64_t>(std::min<uint64_t>(s->prefix_hint, PREFIX_HITS), s->hcap);
        HIPCHK(hipMemcpyAsync(s->pinned, s->d_hitbuf, 64 + pd.pre * sizeof(ipcr_hit), hipMemcpyDeviceToHost, lane)); // counters + hits
        // the host waits for this marker, not for the stream: a scan chained after this one may already
        // be queued behind it on the same stream (ipcr_scratch_chain_after)
        if (!s->ev_done) HIPCHK(hipEventCreateWithFlags(&s->ev_done, hipEventDisableTiming));
        HIPCHK(hipEventRecord(s->ev_done, lane));
    }
    pd.on_lane = true;
    s->stats.enqueue_ms = ms_since(te);
    trace("launch<", s);
    return IPCR_OK;
}

// first half of a scan: everything up to (and including) the enqueue; returns without waiting
// chunk_reset (chunk path): 1 = the chunk holds a reset byte, 0 = it holds none (the host's packer has seen every byte),
// -1 = not known yet (the pack kernel finds out on the device)
ipcr_status scan_enqueue(const ipcr_panel *p, ipcr_scratch *s, ipcr_genome *g, bool chunk = false, int chunk_reset = -1) {
    ipcr_scratch::Pending &pd = s->pend;
    if (pd.active) return fail(IPCR_ERR_INVALID, "a scan is already in flight on this scratch (ipcr_scan_genome_end not called)");
    pd.t0 = std::chrono::steady_clock::now();
    s->hits.clear();
    s->products.clear();
    s->products_in_windows = false;
    s->last_was_chunk = false;
    s->dev_hits_stale = false;
    const double pack_ms_keep = s->stats.pack_ms;
    memset(&s->stats, 0, sizeof s->stats);
    s->stats.pack_ms = pack_ms_keep;
    trace("enqueue>", s);
    // chunk path: nothing is waited for before the sweep is enqueued.  Where the device packs the bases, whether the
    // record holds a reset byte is only known afterwards, so it is scanned the way a genome with such records is (rc
    // patterns unprotected, the host applies the 5' window): right for both kinds of record, as in a resident genome that
    // mixes them.  Where the HOST has packed them it knows, and a chunk without one is scanned like a genome without
    // one -- the pattern set in which every orientation keeps its window (a 1024-row panel: 4.2 ms per 3 Gb against 6.0)
    ipcr_status st = chunk ? genome_finalize_async(g) : genome_finalize(g);
    if (st != IPCR_OK) return st;
    trace("finalized", s);
    s->last_rec_len = g->rec_len;
    s->last_rec_start = g->rec_start;
    const bool clean_chunks = !chunk || env_flag("IPCR_CHUNK_CLEAN_MODE", true); // (per call, for the same reason as IPCR_CHUNK_HOSTPACK)
    const bool any_reset = chunk ? (chunk_reset != 0 || !clean_chunks) : genome_any_reset(g);
    pd.mode = (!p->modes_equal && any_reset) ? 1 : 0;
    st = panel_upload(p, pd.mode, s->device, &s->sdev[pd.mode]);
    if (st != IPCR_OK) return st;
    const PatternSet &set = p->set[pd.mode];
    pd.nrec = (uint32_t)g->rec_start.size();
    pd.nblocks = (g->next_col + 63) / 64;
    pd.block0 = 0;
    pd.segment = false;
    pd.check_rst = any_reset ? 1u : 0u;
    s->stats.pattern_set = (uint32_t)pd.mode;
    s->stats.bases = g->total_bases;
    s->stats.tile_bytes = pd.nblocks * IPCR_BLOCK_PLANE_WORDS * 4ull;
    s->stats.n_patterns = (int32_t)set.ids.size();
    pd.empty = pd.nrec == 0 || set.ids.empty();
    pd.active = true;
    if (pd.empty) return IPCR_OK;
    st = scan_launch(p, s, g);
    if (st != IPCR_OK) pd.active = false;
    return st;
}

// second half: wait, regrow + rescan if a buffer overflowed, bring the hits into join order
ipcr_status scan_collect(const ipcr_panel *p, ipcr_scratch *s, ipcr_genome *g) {
    ipcr_scratch::Pending &pd = s->pend;
    if (!pd.active) return fail(IPCR_ERR_INVALID, "no scan in flight on this scratch");
    pd.active = false;
    if (pd.empty) return IPCR_OK;
    for (int attempt = 0; attempt < 12; ++attempt) {
        const auto tw = std::chrono::steady_clock::now();
        trace("wait>", s);
        if (pd.published) {
            const ipcr_status ws = wait_published(s);
            if (ws == IPCR_ERR_UNSUPPORTED) {
                // never seen on the boxes this was developed on; a platform where the kernel's stores to pinned host
                // memory are not visible would otherwise fail every scan.  Results are still complete in device memory.
                g_publish_broken.store(true);
                static std::atomic<bool> told{false};
                if (!told.exchange(true)) fprintf(stderr, "ipcr_hip: in-kernel hand-over did not reach pinned memory; using the copy path\n");
                const ipcr_status rs = scan_launch(p, s, g);
                if (rs != IPCR_OK) return rs;
                continue;
            }
            if (ws != IPCR_OK) return ws;
        } else if (pd.on_lane) HIPCHK(hipEventSynchronize(s->ev_done)); // the lane already carries the next scan
        else HIPCHK(hipStreamSynchronize(s->stream));
        s->stats.wait_ms += ms_since(tw);
        trace("wait<", s);
        const unsigned long long *pc = static_cast<unsigned long long *>(s->pinned) + 4u * pd.cset_used;
        const ipcr_hit *ph = reinterpret_cast<ipcr_hit *>(static_cast<unsigned long long *>(s->pinned) + 8);
        if (!pd.verified && pc[0] > 0) {
            // some wave's survivor list was full and spilled to the queue (dense matches): run the
            // stand-alone verifier over the spilled words; it appends to the same hit buffer
            const SetDev &sd = *s->sdev[pd.mode];
            // the sweep may have run on another scratch's stream (chained scans) and has only published, not retired:
            // its queue entries become visible to a later kernel on OUR stream when it has ended
            if (pd.published) HIPCHK(hipEventSynchronize(s->ev[1]));
            const uint64_t qset = (uint64_t)IPCR_QUEUE_SHARDS * IPCR_QUEUE_COUNTER_STRIDE;
            unsigned long long *cnt = s->d_counts + 4u * pd.cset_used, *cnt_next = s->d_counts + 4u * (pd.cset_used ^ 1u);
            unsigned long long *qc = s->d_qcounts + qset * pd.cset_used, *qc_next = s->d_qcounts + qset * (pd.cset_used ^ 1u);
            HIPCHK(ipcr::launch_verify(s->stream, g->planes, g->rst, sd.dev, (uint32_t)p->cfg.max_mm, g->d_rec_start,
                                       g->d_rec_len, pd.nrec, pd.check_rst, s->d_queue, s->qcap, qc, s->d_hits,
                                       s->hcap, cnt + 1, cnt + 2, cnt_next, qc_next, s->ev[2], s->ev[3]));
            HIPCHK(hipMemcpyAsync(s->pinned, s->d_hitbuf, 64 + pd.pre * sizeof(ipcr_hit), hipMemcpyDeviceToHost, s->stream));
            pd.verified = true;
            pd.on_lane = false;
            pd.published = false;
            continue;
        }
        const uint64_t nhit = pc[1], ncand = pc[2], fullest = pc[3];
        if (fullest > s->qcap) { // a queue segment overflowed: regrow all segments and rescan
            uint64_t want = s->qcap;
            while (want < fullest) want *= 2;
            if (want > QCAP_MAX) {
                if (p->cfg.hit_cap > 0) { // as below: a capped scan goes on in segments
                    if (pd.segment) { s->seg_overflow = ~0ull; return IPCR_ERR_CAPACITY; }
                    return scan_segmented(p, s, g, hcap_soft() * 16u);
                }
                return fail(IPCR_ERR_CAPACITY, "%llu filter survivors in one queue segment exceed the device queue limit", (unsigned long long)fullest);
            }
            HIPCHK(hipFree(s->d_queue));
            s->d_queue = nullptr;
            HIPCHK(hipMalloc((void **)&s->d_queue, want * IPCR_QUEUE_SHARDS * sizeof(ipcr_queue_entry)));
            s->qcap = want;
            ipcr_status st = scan_launch(p, s, g);
            if (st != IPCR_OK) return st;
            continue;
        }
        if (nhit > s->hcap) {
            uint64_t want = s->hcap;
            while (want < nhit) want *= 2;
            if (p->cfg.hit_cap > 0 && nhit > hcap_soft()) {
                // far more raw matches than a capped scan can use (a low-complexity primer on low-complexity sequence): the
                // reference stops every orientation at HitCap; here the scan is repeated in position order, in segments
                if (pd.segment) { s->seg_overflow = nhit; return IPCR_ERR_CAPACITY; } // (a segment of scan_segmented: it halves and retries)
                return scan_segmented(p, s, g, nhit);
            }
            if (want > HCAP_MAX) return fail(IPCR_ERR_CAPACITY, "%llu hits exceed the device hit-buffer limit", (unsigned long long)nhit);
            HIPCHK(hipFree(s->d_hitbuf));
            s->d_hitbuf = nullptr;
            s->d_hits = nullptr;
            s->d_counts = nullptr;
            HIPCHK(hipMalloc(&s->d_hitbuf, (want + 2) * sizeof(ipcr_hit_rec)));
            HIPCHK(hipMemset(s->d_hitbuf, 0, 64));
            HIPCHK(hipStreamSynchronize(nullptr)); // (null-stream fill: done before the sweep on the scratch's non-blocking stream counts into it)
            s->d_counts = static_cast<unsigned long long *>(s->d_hitbuf);
            s->d_hits = static_cast<ipcr_hit_rec *>(s->d_hitbuf) + 2;
            s->hcap = want;
            ipcr_status st = scan_launch(p, s, g);
            if (st != IPCR_OK) return st;
            continue;
        }
        std::vector<ipcr_hit> &raw = s->hits_raw;
        raw.resize(nhit);
        uint64_t got = std::min<uint64_t>(nhit, pd.pre);
        if (pd.published && got) {
            // The records were written to pinned memory by whichever wave found them, each as two 16-byte stores that
            // nothing orders on their way to the host: each half carries this scan's tag (jit.cpp: publish) and a record
            // is taken only when BOTH are there.  Wait for the ones still on their way (normally none); after 2 ms fetch
            // the whole prefix from device memory, where the kernel has left the same records.
            const volatile ipcr_hit *vh = ph;
            const uint64_t tag_pos = (uint64_t)(s->seq & 0xFFFFFFu);
            const uint64_t tag_hi = (uint64_t)s->seq << 32;
            bool complete = true;
            const auto tw0 = std::chrono::steady_clock::now();
            for (uint64_t i = 0; i < got && complete; ++i) {
                for (uint64_t spin = 1; (vh[i].pos >> 40) != tag_pos || vh[i].mm_mask[1] != (tag_hi | (uint32_t)i); ++spin) {
                    __builtin_ia32_pause();
                    if ((spin & 0xFFFu) == 0 && ms_since(tw0) > 2.0) { complete = false; break; }
                }
            }
            std::atomic_thread_fence(std::memory_order_acquire);
            if (!complete) {
                ++s->stats.handover_refetched;
                HIPCHK(hipEventSynchronize(s->ev[1]));
                // a stream of its own (never waits behind a sweep), created on first use: every stream takes a turn in the
                // runtime's stream -> hardware-queue mapping, and a worker pool should not pay for streams it never uses
                if (!s->cstream) HIPCHK(hipStreamCreateWithFlags(&s->cstream, hipStreamNonBlocking));
                HIPCHK(hipMemcpyAsync(raw.data(), s->d_hits, got * sizeof(ipcr_hit), hipMemcpyDeviceToHost, s->cstream));
                HIPCHK(hipStreamSynchronize(s->cstream));
            } else {
                memcpy(raw.data(), ph, got * sizeof(ipcr_hit));
                for (uint64_t i = 0; i < got; ++i) { // strip the tags (patterns of the specialised filter are <= 32 nt: m1 is zero)
                    raw[i].pos &= (1ull << 40) - 1ull;
                    raw[i].mm_mask[1] = 0;
                }
            }
        } else if (got) {
            memcpy(raw.data(), ph, got * sizeof(ipcr_hit));
        }
        // the kernel retires a moment after its last wave; only then is device memory (records beyond the pinned
        // prefix, the buffer an all-gather may read next) guaranteed to hold what its waves wrote
        if (pd.published) HIPCHK(hipEventSynchronize(s->ev[1]));
        if (nhit > got) HIPCHK(hipMemcpy(raw.data() + got, s->d_hits + got, (nhit - got) * sizeof(ipcr_hit), hipMemcpyDeviceToHost));
        s->prefix_hint = std::max<uint64_t>(256, nhit + nhit / 4 + 16);
        float fms = 0, vms = 0;
        static const bool check_pub = env_flag("IPCR_DEBUG_PUBLISH_CHECK", false);
        if (check_pub && pd.published && got) { // what the host took from pinned memory vs what the kernel left in device memory
            std::vector<ipcr_hit> dev(got);
            HIPCHK(hipMemcpy(dev.data(), s->d_hits, got * sizeof(ipcr_hit), hipMemcpyDeviceToHost));
            uint64_t diff = 0;
            for (uint64_t i = 0; i < got; ++i)
                if (memcmp(&dev[i], &raw[i], sizeof(ipcr_hit)) != 0) ++diff;
            s->stats.handover_check_diffs += diff;
            ++s->stats.handover_checked;
            if (diff) fprintf(stderr, "publish check: %llu of %llu hit records taken from pinned memory differ from device memory\n",
                              (unsigned long long)diff, (unsigned long long)got);
        }
        trace("retired", s);
        HIPCHK(hipEventElapsedTime(&fms, s->ev[0], s->ev[1]));
        if (pd.verified) HIPCHK(hipEventElapsedTime(&vms, s->ev[2], s->ev[3]));
        s->stats.filter_ms = fms;
        s->stats.verify_ms = vms;
        s->stats.candidates = ncand;
        s->stats.hits = nhit;
        const auto ts = std::chrono::steady_clock::now();
        sort_hits(raw, s->hits, pd.nrec, (uint32_t)p->defs.size());
        s->stats.sort_ms = ms_since(ts);
        s->stats.total_ms = ms_since(pd.t0);
        trace("sorted", s);
        return IPCR_OK;
    }
    return fail(IPCR_ERR_CAPACITY, "scan buffers kept overflowing");
}

// A capped scan with far more raw matches than anybody can use.  The reference's collectors stop at HitCap matches
// per orientation and record, in ascending position (core/engine/hit_collect.go:80-82; FindMatches: core/primer/match.go:86-88),
// so it never holds more; the device appends matches in no particular order, so it cannot stop early in one sweep.  Here the
// sweep is repeated over consecutive RANGES OF BLOCKS (positions ascend from range to range; every window belongs to the
// block it starts in, so the ranges partition the windows exactly), the host keeps a (record, pattern)'s matches until
// HitCap of them have passed the 5' window filter the host applies for patterns the device scans unprotected (that is
// at least HitCap raw ones: enough for the reference's cap-before-filter quirk too, orientation_matches), and ignores the
// rest.  Memory stays bounded by the soft limit per range; a range that still overflows is halved.
ipcr_status scan_segmented(const ipcr_panel *p, ipcr_scratch *s, ipcr_genome *g, uint64_t seen_hits) {
    ipcr_scratch::Pending &pd = s->pend;
    const uint64_t total = (g->next_col + 63) / 64;
    const uint64_t cap = (uint64_t)p->cfg.hit_cap;
    const size_t ndefs = p->defs.size();
    std::vector<ipcr_hit> kept;
    std::map<uint64_t, uint64_t> passed; // (record << 32 | pattern) -> matches that passed the host's window filter so far
    uint64_t seg = std::max<uint64_t>(1, total * (hcap_soft() / 2) / std::max<uint64_t>(seen_hits, 1)); // expect ~half the limit per range
    double filter_ms = 0, verify_ms = 0;
    uint64_t cand = 0;
    for (uint64_t b0 = 0; b0 < total;) {
        const uint64_t nb = std::min(seg, total - b0);
        pd.active = true;
        pd.segment = true;
        pd.block0 = b0;
        pd.nblocks = nb;
        ipcr_status st = scan_launch(p, s, g);
        if (st == IPCR_OK) st = scan_collect(p, s, g);
        if (st == IPCR_ERR_CAPACITY && s->seg_overflow) { // this range alone exceeds the limit: smaller ranges
            s->seg_overflow = 0;
            if (seg == 1) return fail(IPCR_ERR_CAPACITY, "one block of the genome holds more raw matches than the device hit buffer may take");
            seg = std::max<uint64_t>(1, nb / 2);
            continue;
        }
        if (st != IPCR_OK) return st;
        filter_ms += s->stats.filter_ms;
        verify_ms += s->stats.verify_ms;
        cand += s->stats.candidates;
        // s->hits: this range's matches, sorted by (record, pattern, position) and free of duplicates
        for (const ipcr_hit &h : s->hits) {
            const uint32_t pat = h.pattern & 0x7FFFFFFFu;
            uint64_t &n = passed[((uint64_t)h.record << 32) | pat];
            if (n >= cap) continue; // this orientation of this record has all it can use
            kept.push_back(h);
            const bool dropped_by_host = pat < ndefs && p->defs[pat].left && hit_has_idx_below(&h, p->tw);
            if (!dropped_by_host) ++n;
        }
        b0 += nb;
    }
    pd.segment = false;
    pd.block0 = 0;
    pd.nblocks = total;
    s->hits_raw = kept;
    sort_hits(s->hits_raw, s->hits, pd.nrec, (uint32_t)ndefs);
    s->stats.filter_ms = filter_ms;
    s->stats.verify_ms = verify_ms;
    s->stats.candidates = cand;
    s->stats.hits = s->hits.size();
    s->stats.segmented = 1;
    s->dev_hits_stale = true; // the device buffer and its header hold the last range only (ipcr_scratch_device_hits, the exchange)
    s->stats.total_ms = ms_since(pd.t0);
    return IPCR_OK;
}

ipcr_status scan_hits(const ipcr_panel *p, ipcr_scratch *s, ipcr_genome *g) {
    ipcr_status st = scan_enqueue(p, s, g);
    if (st != IPCR_OK) return st;
    return scan_collect(p, s, g);
}

// ---------------------------------------------------------------------------- join
struct MatchRef {
    int64_t pos;
    const ipcr_hit *h;
};

inline int hit_mm(const ipcr_hit *h) { return __builtin_popcountll(h->mm_mask[0]) + __builtin_popcountll(h->mm_mask[1]); }

inline bool hit_has_idx_below(const ipcr_hit *h, int tw) { // filterLeftTW engine.go:53-68
    if (tw <= 0) return false;
    if (tw >= 64) {
        if (h->mm_mask[0]) return true;
        if (tw >= 128) return h->mm_mask[1] != 0;
        return (h->mm_mask[1] & ((1ull << (tw - 64)) - 1ull)) != 0;
    }
    return (h->mm_mask[0] & ((1ull << tw) - 1ull)) != 0;
}

int fill_idx(const ipcr_hit *h, uint8_t *out, bool flip, int plen) {
    int n = 0;
    for (int w = 0; w < 2; ++w) {
        uint64_t m = h->mm_mask[w];
        while (m && n < IPCR_MAX_MM) {
            const int j = __builtin_ctzll(m) + 64 * w;
            m &= m - 1;
            out[n++] = (uint8_t)(flip ? plen - 1 - j : j); // engine.go:127-136
        }
    }
    return n;
}

struct JoinCtx {
    const ipcr_panel *p;
    std::vector<ipcr_product> *out;
    ipcr_emit_fn emit;
    void *user;
    bool aborted = false;
};

bool push_product(JoinCtx &c, int pair, uint32_t rec, int64_t start, int64_t end, int64_t length, int type,
                  const MatchRef &mf, const MatchRef &mr, int rlen) {
    c.out->emplace_back();
    ipcr_product &pr = c.out->back();
    pr.start = start; pr.end = end; pr.length = length;
    pr.pair = pair; pr.record = (int32_t)rec; pr.type = type;
    memset(pr.fwd_idx, 0, sizeof pr.fwd_idx + sizeof pr.rev_idx);
    pr.fwd_mm = hit_mm(mf.h);
    pr.rev_mm = hit_mm(mr.h);
    pr.n_fwd_idx = pr.fwd_mm ? fill_idx(mf.h, pr.fwd_idx, false, 0) : 0;
    pr.n_rev_idx = pr.rev_mm ? fill_idx(mr.h, pr.rev_idx, true, rlen) : 0;
    if (c.emit && c.emit(&pr, c.user) != 0) { c.aborted = true; return false; }
    return true;
}

// one direction of forEachJoinedProduct (engine.go:143-272 / :274-401)
bool join_direction(JoinCtx &c, int pair, uint32_t rec, int64_t seqlen, int64_t minL, int64_t maxL,
                    const std::vector<MatchRef> &left, const std::vector<MatchRef> &right, int rlen, int type) {
    auto lower = [&](int64_t pos) { return std::lower_bound(right.begin(), right.end(), pos, [](const MatchRef &m, int64_t v) { return m.pos < v; }) - right.begin(); };
    auto upper = [&](int64_t pos) { return std::upper_bound(right.begin(), right.end(), pos, [](int64_t v, const MatchRef &m) { return v < m.pos; }) - right.begin(); };
    for (const MatchRef &ma : left) {
        const int64_t last = seqlen - rlen;
        int64_t lo = ma.pos + 1;
        if (minL > 0) {
            lo = ma.pos + minL - rlen;
            if (lo <= ma.pos) lo = ma.pos + 1;
        }
        if (lo < 0) lo = 0;
        int64_t hi = last;
        if (maxL > 0) {
            hi = ma.pos + maxL - rlen;
            if (hi > last) hi = last;
        }
        if (hi >= lo) {
            const int64_t iMin = lower(lo), iMax = upper(hi) - 1;
            for (int64_t j = iMax; j >= iMin; --j) {
                const MatchRef &mb = right[(size_t)j];
                const int64_t end = mb.pos + rlen, length = end - ma.pos;
                if ((minL != 0 && length < minL) || (maxL != 0 && length > maxL)) continue;
                if (!push_product(c, pair, rec, ma.pos, end, length, type, ma, mb, rlen)) return false;
            }
        }
        if (c.p->cfg.circular) {
            const int64_t X = seqlen - ma.pos;
            int64_t loWrap = 0;
            if (minL > 0) {
                int64_t needed = minL - X - rlen;
                if (needed < 0) needed = 0;
                loWrap = needed;
            }
            int64_t hiWrap = ma.pos - 1;
            if (maxL > 0) {
                const int64_t allowed = maxL - X - rlen;
                if (allowed < hiWrap) hiWrap = allowed;
            }
            if (hiWrap >= loWrap) {
                const int64_t iMinW = lower(loWrap), iMaxW = upper(hiWrap) - 1;
                for (int64_t j = iMaxW; j >= iMinW; --j) {
                    const MatchRef &mb = right[(size_t)j];
                    if (mb.pos >= ma.pos) continue;
                    const int64_t end = mb.pos + rlen, length = (seqlen - ma.pos) + end;
                    if ((minL != 0 && length < minL) || (maxL != 0 && length > maxL)) continue;
                    if (!push_product(c, pair, rec, ma.pos, end, length, type, ma, mb, rlen)) return false;
                }
            }
        }
    }
    return true;
}

// Rebuild the match list the reference would hold for one orientation of one record
// (compiled.go:185-258) from the device hits of its pattern (ascending position).
void orientation_matches(const ipcr_panel *p, int pair, int w, bool rec_reset, const ipcr_hit *begin,
                         const ipcr_hit *end, std::vector<MatchRef> &out) {
    out.clear();
    const bool left = w >= 2;
    const bool seeded = (p->have[(size_t)pair] >> w) & 1;
    const int tw = p->tw, k = p->cfg.max_mm, cap = p->cfg.hit_cap;
    const bool fallback = !seeded || (rec_reset && k > 0 && cap > 0); // compiled.go:189-190,238-258
    const bool cap_before_filter = left && tw > 0 && k > 0 && cap > 0 && fallback;
    if (cap_before_filter) {
        int64_t n = 0;
        for (const ipcr_hit *h = begin; h != end; ++h) {
            if (n >= cap) break; // FindMatches stops at capHits raw matches (match.go:86-88)
            ++n;
            if (!hit_has_idx_below(h, tw)) out.push_back({(int64_t)h->pos, h});
        }
        return;
    }
    for (const ipcr_hit *h = begin; h != end; ++h) {
        if (left && hit_has_idx_below(h, tw)) continue; // device scanned this pattern unprotected
        if (cap > 0 && (int64_t)out.size() >= cap) break;
        out.push_back({(int64_t)h->pos, h});
    }
    // collector order when HitCap == 0: automaton hits first, then non-ACGT halo hits
    // (compiled.go:211-232; a start is halo-only iff its seed span holds a reset byte)
    if (seeded && cap <= 0 && k > 0 && rec_reset)
        std::stable_partition(out.begin(), out.end(), [](const MatchRef &m) { return (m.h->pattern >> 31) == 0; });
}

// the products of ONE record: hits [i, j) of H (sorted by pattern, position), all of record `rec`; false when emit aborted
bool join_one_record(const ipcr_panel *p, const std::vector<ipcr_hit> &H, size_t i, size_t j, uint32_t rec, int64_t seqlen, uint8_t fl, JoinCtx &c) {
    const size_t npairs = p->id.size();
    static thread_local std::vector<MatchRef> m[4];
    static thread_local std::vector<MatchRef> sorted_right;
    static thread_local std::vector<uint32_t> touched;
    static thread_local std::vector<uint32_t> pat_begin;
    const size_t ndefs = p->defs.size();
    const bool rec_reset = fl & 1u;
    const int mode = (!p->modes_equal && (fl & 2u)) ? 1 : 0;
    // hits of this record are sorted by pattern: where each pattern's run begins (one pass instead of four
    // binary searches per touched pair -- a 1024-pair panel touches ~1000 pairs per record)
    pat_begin.resize(ndefs + 1);
    {
        size_t q = i;
        for (size_t gdx = 0; gdx <= ndefs; ++gdx) {
            while (q < j && (H[q].pattern & 0x7FFFFFFFu) < gdx) ++q;
            pat_begin[gdx] = (uint32_t)q;
        }
    }
    // only pairs that scan a pattern with hits in this record can yield products
    touched.clear();
    for (size_t h = i; h < j;) {
        const uint32_t gid = H[h].pattern & 0x7FFFFFFFu;
        if (gid < p->users[mode].size())
            touched.insert(touched.end(), p->users[mode][gid].begin(), p->users[mode][gid].end());
        while (h < j && (H[h].pattern & 0x7FFFFFFFu) == gid) ++h;
    }
    std::sort(touched.begin(), touched.end());
    touched.erase(std::unique(touched.begin(), touched.end()), touched.end());
    for (const uint32_t pi32 : touched) {
        const size_t pi = pi32;
        if (pi >= npairs) continue;
        bool any = false;
        for (int w = 0; w < 4; ++w) {
            const uint32_t gid = p->slot[pi][(size_t)w][(size_t)mode];
            const ipcr_hit *b = H.data() + pat_begin[gid], *e = H.data() + pat_begin[gid + 1];
            orientation_matches(p, (int)pi, w, rec_reset, b, e, m[w]);
            any |= !m[w].empty();
        }
        if (!any) continue;
        int64_t minL = p->minp[pi], maxL = p->maxp[pi]; // engine.go:113-120
        if (minL == 0) minL = p->cfg.min_len;
        if (maxL == 0) maxL = p->cfg.max_len;
        const int alen = (int)p->fwd[pi].size(), blen = (int)p->rev[pi].size();
        auto by_pos = [](const MatchRef &a, const MatchRef &b) { return a.pos < b.pos; };
        // "forward": A x rc(B)   (rc list sorted by position unless it already is, engine.go:70-85,144)
        const std::vector<MatchRef> *right = &m[3];
        if (!std::is_sorted(m[3].begin(), m[3].end(), by_pos)) {
            sorted_right = m[3];
            std::stable_sort(sorted_right.begin(), sorted_right.end(), by_pos);
            right = &sorted_right;
        }
        if (!m[0].empty() && !right->empty() &&
            !join_direction(c, (int)pi, rec, seqlen, minL, maxL, m[0], *right, blen, 0)) return false;
        // "revcomp": B x rc(A)   (engine.go:275)
        right = &m[2];
        if (!std::is_sorted(m[2].begin(), m[2].end(), by_pos)) {
            sorted_right = m[2];
            std::stable_sort(sorted_right.begin(), sorted_right.end(), by_pos);
            right = &sorted_right;
        }
        if (!m[1].empty() && !right->empty() &&
            !join_direction(c, (int)pi, rec, seqlen, minL, maxL, m[1], *right, alen, 1)) return false;
    }
    return true;
}

ipcr_status join_sorted_hits(const ipcr_panel *p, ipcr_scratch *s, const uint64_t *rec_len, const uint8_t *rec_flags,
                             uint32_t nrec, ipcr_emit_fn emit, void *user) {
    const std::vector<ipcr_hit> &H = s->hits;
    struct Range { size_t i, j; uint32_t rec; };
    std::vector<Range> ranges;
    for (size_t i = 0; i < H.size();) {
        const uint32_t rec = H[i].record;
        size_t j = i;
        while (j < H.size() && H[j].record == rec) ++j;
        if (rec >= nrec) return fail(IPCR_ERR_INVALID, "hit refers to record %u of %u", rec, nrec);
        ranges.push_back({i, j, rec});
        i = j;
    }
    // Records are independent (core/engine/compiled.go:162-267 is called per record), products come out record by record: a
    // large hit list (a 1024-row panel at k = 3: 365 000 hits, 15 ms of match lists and binary searches on one thread) is joined
    // by the process's pool, every record into a list of its own, and the lists are handed out in record order.
    const bool par_env = env_flag("IPCR_JOIN_PARALLEL", true); // (read per call: the tests compare the two forms in one process)
    if (par_env && ranges.size() > 1 && H.size() >= 32768 && PackPool::get().size() > 1) {
        std::vector<std::vector<ipcr_product>> outs(ranges.size());
        PackPool::get().run(ranges.size(), [&](size_t r) {
            JoinCtx c{p, &outs[r], nullptr, nullptr};
            const Range &g = ranges[r];
            (void)join_one_record(p, H, g.i, g.j, g.rec, (int64_t)rec_len[g.rec], rec_flags ? rec_flags[g.rec] : 0, c);
        });
        for (const auto &o : outs)
            for (const ipcr_product &pr : o) {
                s->products.push_back(pr);
                if (emit && emit(&s->products.back(), user) != 0) return fail(IPCR_ERR_ABORTED, "emit callback aborted the scan");
            }
    } else {
        JoinCtx c{p, &s->products, emit, user};
        for (const Range &g : ranges)
            if (!join_one_record(p, H, g.i, g.j, g.rec, (int64_t)rec_len[g.rec], rec_flags ? rec_flags[g.rec] : 0, c))
                return fail(IPCR_ERR_ABORTED, "emit callback aborted the scan");
    }
    s->stats.products = s->products.size();
    return IPCR_OK;
}

ipcr_status scratch_ready(const ipcr_panel *p, ipcr_scratch *s, bool need_device = true) {
    if (!p || !s) return fail(IPCR_ERR_INVALID, "null panel or scratch");
    if (s->panel != p) return fail(IPCR_ERR_INVALID, "scratch was created for a different panel");
    if (need_device && !s->stream) return fail(IPCR_ERR_DEVICE, "host-only scratch cannot scan: the ipcr scan path has no CPU fallback");
    return IPCR_OK;
}

ipcr_status same_device(const ipcr_scratch *s, const ipcr_genome *g) {
    if (!g) return fail(IPCR_ERR_INVALID, "null genome");
    if (s->stream && s->device != g->device)
        return fail(IPCR_ERR_INVALID, "scratch lives on device %d, the genome on device %d: scan a genome with a scratch of its own device (ipcr_scratch_create_on)", s->device, g->device);
    return IPCR_OK;
}

} // namespace

extern "C" {

ipcr_status ipcr_scratch_create(const ipcr_panel *p, ipcr_scratch **out) {
    return ipcr_scratch_create_on(p, default_slot(), out);
}

int32_t ipcr_scratch_device(const ipcr_scratch *s) { return (s && s->stream) ? s->device : -1; }

ipcr_status ipcr_scratch_create_on(const ipcr_panel *p, int32_t device, ipcr_scratch **out) {
    if (!p || !out) return fail(IPCR_ERR_INVALID, "ipcr_scratch_create: null argument");
    *out = nullptr;
    if (slot_count() == 0) return fail(IPCR_ERR_DEVICE, "no HIP device visible: the ipcr scan path has no CPU fallback");
    if (device < 0 || device >= slot_count()) return fail(IPCR_ERR_INVALID, "ipcr_scratch_create_on: device %d of %d", device, slot_count());
    DeviceGuard dg(device);
    std::unique_ptr<ipcr_scratch> s(new ipcr_scratch);
    s->panel = p;
    ipcr_scratch *raw = s.get();
    auto build = [&]() -> ipcr_status {
        raw->device = device;
        raw->own_lane = std::make_shared<ipcr_scratch::Lane>();
        {
            // Streams share a few hardware queues (creation order decides which); a sweep lane that lands on the queue
            // of, say, the RCCL stream has every collective queued between two sweeps (+25 us per step measured).
            // IPCR_LANE_PRIORITY=1 asks for a queue of the high-priority class instead.
            static const int lane_prio = env_flag("IPCR_LANE_PRIORITY", false) ? 1 : 0;
            if (lane_prio) {
                int lo = 0, hi = 0; // numerically lower = higher priority
                HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
                HIPCHK(hipStreamCreateWithPriority(&raw->own_lane->s, hipStreamNonBlocking, hi));
            } else {
                HIPCHK(hipStreamCreateWithFlags(&raw->own_lane->s, hipStreamNonBlocking));
            }
        }
        raw->stream = raw->own_lane->s;
        for (auto &e : raw->ev) HIPCHK(hipEventCreate(&e));
        raw->qcap = QCAP_INIT;
        raw->hcap = HCAP_INIT;
        HIPCHK(hipMalloc((void **)&raw->d_queue, raw->qcap * IPCR_QUEUE_SHARDS * sizeof(ipcr_queue_entry)));
        HIPCHK(hipMalloc((void **)&raw->d_qcounts, 2ull * IPCR_QUEUE_SHARDS * IPCR_QUEUE_COUNTER_STRIDE * 8ull));
        HIPCHK(hipMemset(raw->d_qcounts, 0, 2ull * IPCR_QUEUE_SHARDS * IPCR_QUEUE_COUNTER_STRIDE * 8ull));
        HIPCHK(hipMalloc(&raw->d_hitbuf, (raw->hcap + 2) * sizeof(ipcr_hit_rec)));
        HIPCHK(hipMemset(raw->d_hitbuf, 0, 64));
        raw->d_counts = static_cast<unsigned long long *>(raw->d_hitbuf);
        raw->d_hits = static_cast<ipcr_hit_rec *>(raw->d_hitbuf) + 2;
        HIPCHK(hipHostMalloc(&raw->pinned, 64 + PREFIX_HITS * sizeof(ipcr_hit) + 64, hipHostMallocDefault));
        memset(raw->pinned, 0, 64 + PREFIX_HITS * sizeof(ipcr_hit) + 64);
        HIPCHK(hipMalloc((void **)&raw->d_tickets, 65 * 128));
        HIPCHK(hipMemset(raw->d_tickets, 0, 65 * 128));
        HIPCHK(hipStreamSynchronize(nullptr)); // (the three fills above run on the null stream: done before the first sweep on the scratch's own, non-blocking stream)
        raw->counted_in = p->live_scratches;
        raw->counted_in->fetch_add(1);
        return IPCR_OK;
    };
    ipcr_status st = build();
    if (st != IPCR_OK) { ipcr_scratch_destroy(s.release()); return st; }
    *out = s.release();
    return IPCR_OK;
}

ipcr_status ipcr_scratch_create_host(const ipcr_panel *p, ipcr_scratch **out) {
    if (!p || !out) return fail(IPCR_ERR_INVALID, "ipcr_scratch_create_host: null argument");
    ipcr_scratch *s = new ipcr_scratch;
    s->panel = p;
    *out = s;
    return IPCR_OK;
}

void ipcr_scratch_destroy(ipcr_scratch *s) {
    if (!s) return;
    DeviceGuard dg(s->device);
    if (s->stream) { // a scan left in flight still reads and writes the buffers freed below
        if (s->pend.active && s->lane_used) (void)hipStreamSynchronize(s->lane_used->s);
        (void)hipStreamSynchronize(s->stream);
    }
    if (s->counted_in) s->counted_in->fetch_sub(1);
    for (int h = 0; h < 2; ++h) {
        if (s->h_stage[h]) (void)hipHostFree(s->h_stage[h]);
        if (s->ev_stage[h]) (void)hipEventDestroy(s->ev_stage[h]);
        if (h == 0 && s->h_planes) (void)hipHostFree(s->h_planes);
    }
    if (s->chunk) ipcr_genome_destroy(s->chunk);
    if (s->nest) ipcr_genome_destroy(s->nest);
    if (s->d_queue) (void)hipFree(s->d_queue);
    if (s->d_qcounts) (void)hipFree(s->d_qcounts);
    if (s->d_hitbuf) (void)hipFree(s->d_hitbuf);
    if (s->d_amps) (void)hipFree(s->d_amps);
    if (s->d_probe_misc) (void)hipFree(s->d_probe_misc);
    if (s->pinned) (void)hipHostFree(s->pinned);
    if (s->d_tickets) (void)hipFree(s->d_tickets);
    for (auto &e : s->ev)
        if (e) (void)hipEventDestroy(e);
    if (s->ev_done) (void)hipEventDestroy(s->ev_done);
    if (s->cstream) (void)hipStreamDestroy(s->cstream);
    if (s->probe_stream) (void)hipStreamDestroy(s->probe_stream);
    if (s->h_probe) (void)hipHostFree(s->h_probe);
    // the stream goes with the last scratch that shares it (own_lane / lane_used references)
    delete s;
}

ipcr_status ipcr_scratch_stats(const ipcr_scratch *s, ipcr_scan_stats *out) {
    if (!s || !out) return fail(IPCR_ERR_INVALID, "null argument");
    *out = s->stats;
    return IPCR_OK;
}

ipcr_status ipcr_scratch_products(const ipcr_scratch *s, const ipcr_product **out, int64_t *n) {
    if (!s || !out || !n) return fail(IPCR_ERR_INVALID, "null argument");
    *out = s->products.data();
    *n = (int64_t)s->products.size();
    return IPCR_OK;
}

ipcr_status ipcr_scratch_hits(const ipcr_scratch *s, const ipcr_hit **out, int64_t *n) {
    if (!s || !out || !n) return fail(IPCR_ERR_INVALID, "null argument");
    *out = s->hits.data();
    *n = (int64_t)s->hits.size();
    return IPCR_OK;
}

ipcr_status ipcr_scratch_device_hits(const ipcr_scratch *s, const void **dev_block, uint64_t *n_hits, uint64_t *capacity) {
    if (!s || !dev_block || !n_hits || !capacity) return fail(IPCR_ERR_INVALID, "null argument");
    if (!s->d_hitbuf) return fail(IPCR_ERR_DEVICE, "host-only scratch has no device hit buffer");
    *dev_block = s->d_hitbuf;
    *n_hits = s->hits_raw.size();
    *capacity = s->hcap;
    if (s->dev_hits_stale) // (the outputs are filled all the same: the buffer exists, it just is not this scan's result)
        return fail(IPCR_ERR_UNSUPPORTED, "the device hit buffer does not hold this scan's hits: a capped scan that ran in segments keeps them on the host "
                                          "(ipcr_scratch_hits; ipcr_exchange_begin sends them from there)");
    return IPCR_OK;
}

} // extern "C"
ipcr_status ipcr_internal_scratch_send_block(const ipcr_scratch *s, const void **dev_block, uint64_t *hcap, const ipcr_hit **host_hits,
                                             uint64_t *n_host, int *authoritative) {
    if (!s || !s->d_hitbuf) return fail(IPCR_ERR_DEVICE, "host-only scratch has no device hit buffer");
    *dev_block = s->d_hitbuf;
    *hcap = s->hcap;
    *host_hits = s->hits_raw.data();
    *n_host = s->hits_raw.size();
    *authoritative = s->dev_hits_stale ? 0 : 1;
    return IPCR_OK;
}
extern "C" {

ipcr_status ipcr_scan_genome_hits(const ipcr_panel *p, ipcr_scratch *s, const ipcr_genome *g) {
    ipcr_status st = scratch_ready(p, s);
    if (st == IPCR_OK) st = same_device(s, g);
    if (st != IPCR_OK) return st;
    DeviceGuard dg(s->device);
    s->stats.pack_ms = 0;
    return scan_hits(p, s, const_cast<ipcr_genome *>(g));
}

ipcr_status ipcr_scan_genome(const ipcr_panel *p, ipcr_scratch *s, const ipcr_genome *g, ipcr_emit_fn emit, void *user) {
    const auto t0 = std::chrono::steady_clock::now();
    ipcr_status st = ipcr_scan_genome_hits(p, s, g);
    if (st != IPCR_OK) return st;
    std::vector<uint8_t> fl(g->rec_start.size());
    const bool any = genome_any_reset(g);
    for (size_t r = 0; r < fl.size(); ++r) fl[r] = (uint8_t)((g->flags[r] & 1u) | (any ? 2u : 0u));
    const auto tj = std::chrono::steady_clock::now();
    st = join_sorted_hits(p, s, g->rec_len.data(), fl.data(), (uint32_t)fl.size(), emit, user);
    s->stats.join_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tj).count();
    s->stats.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return st;
}

// The rolling windows of one record, as the streaming reader emits them (fasta.cpp: ipcr_fasta_next; core/fasta/path_ctx.go:
// 148-160 a window leaves as soon as more than chunk_size bases are waiting, the reader then moves on by `step`; :126-138 at
// the record's end what is left goes out -- the whole record under its own ID if no window ever left).
static void record_windows(uint32_t rec, uint64_t len, int64_t chunk, int64_t step, std::vector<ipcr_chunk_window> &out) {
    uint64_t ws = 0, last_end = 0;
    bool emitted = false;
    if (step > 0)
        while (len - ws > (uint64_t)chunk) {
            out.push_back({rec, 0u, ws, ws + (uint64_t)chunk, 0u, 0u});
            last_end = ws + (uint64_t)chunk;
            emitted = true;
            ws += (uint64_t)step;
        }
    if (!emitted) out.push_back({rec, 1u, 0, len, 0u, 0u});
    else if (last_end < len) out.push_back({rec, 0u, ws, len, 0u, 0u});
}

ipcr_status ipcr_chunk_windows(uint64_t len, int64_t chunk_size, int64_t overlap, ipcr_chunk_window *out, int64_t cap, int64_t *n) {
    if (!n) return fail(IPCR_ERR_INVALID, "ipcr_chunk_windows: null argument");
    int64_t step = chunk_size - overlap;
    if (chunk_size <= 0 || step <= 0) step = 0; // whole records (path_ctx.go:87-90)
    std::vector<ipcr_chunk_window> w;
    record_windows(0, len, chunk_size, step, w);
    *n = (int64_t)w.size();
    for (int64_t i = 0; out && i < cap && i < *n; ++i) out[i] = w[(size_t)i];
    return IPCR_OK;
}

ipcr_status ipcr_scratch_chunk_windows(const ipcr_scratch *s, const ipcr_chunk_window **out, int64_t *n) {
    if (!s || !out || !n) return fail(IPCR_ERR_INVALID, "ipcr_scratch_chunk_windows: null argument");
    *out = s->windows.data();
    *n = (int64_t)s->windows.size();
    return IPCR_OK;
}

ipcr_status ipcr_scan_genome_chunked(const ipcr_panel *p, ipcr_scratch *s, const ipcr_genome *g, int64_t chunk_size, int64_t overlap,
                                     ipcr_emit_fn emit, void *user) {
    const auto t0 = std::chrono::steady_clock::now();
    if (p && p->cfg.circular) return fail(IPCR_ERR_INVALID, "ipcr_scan_genome_chunked: chunking is disabled for circular templates (runutil.go:46-49)");
    ipcr_status st = ipcr_scan_genome_hits(p, s, g); // ONE sweep of the tiles: every window's hits are in the record's list
    if (st != IPCR_OK) return st;
    if (s->stats.segmented)
        return fail(IPCR_ERR_UNSUPPORTED, "ipcr_scan_genome_chunked: the capped scan ran in segments (the device kept per record what HitCap can use, not per window)");
    DeviceGuard dg(s->device);
    int64_t step = chunk_size - overlap;
    if (chunk_size <= 0 || step <= 0) step = 0;
    const uint32_t nrec = (uint32_t)g->rec_start.size();
    s->windows.clear();
    std::vector<size_t> first_window(nrec + 1, 0);
    for (uint32_t r = 0; r < nrec; ++r) {
        first_window[r] = s->windows.size();
        record_windows(r, g->rec_len[r], chunk_size, step, s->windows);
    }
    first_window[nrec] = s->windows.size();
    // which windows hold a reset byte: asked on the device for the windows of records that hold one at all
    {
        std::vector<uint64_t> ab;
        std::vector<size_t> which;
        for (size_t w = 0; w < s->windows.size(); ++w) {
            const ipcr_chunk_window &cw = s->windows[w];
            if (!(g->flags[cw.record] & 1u) || cw.end <= cw.start) continue;
            ab.push_back(g->rec_start[cw.record] + cw.start);
            ab.push_back(g->rec_start[cw.record] + cw.end);
            which.push_back(w);
        }
        if (!which.empty()) {
            uint64_t *d_ab = nullptr;
            uint32_t *d_fl = nullptr;
            std::vector<uint32_t> fl(which.size(), 0);
            HIPCHK(hipMalloc((void **)&d_ab, ab.size() * 8u + which.size() * 4u));
            d_fl = reinterpret_cast<uint32_t *>(d_ab + ab.size());
            hipError_t e = hipMemcpyAsync(d_ab, ab.data(), ab.size() * 8u, hipMemcpyHostToDevice, s->stream);
            if (e == hipSuccess) e = ipcr::launch_window_reset(s->stream, g->rst, d_ab, (uint32_t)which.size(), d_fl);
            if (e == hipSuccess) e = hipMemcpyAsync(fl.data(), d_fl, which.size() * 4u, hipMemcpyDeviceToHost, s->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
            (void)hipFree(d_ab);
            if (e != hipSuccess) return fail(IPCR_ERR_DEVICE, "HIP: %s (window reset flags)", hipGetErrorString(e));
            for (size_t i = 0; i < which.size(); ++i) s->windows[which[i]].reset = fl[i] ? 1u : 0u;
        }
    }
    // every window is one ForEachCompiledProduct call over its own hits (those that lie wholly inside it), window-local
    const auto tj = std::chrono::steady_clock::now();
    const bool mode1 = s->pend.mode == 1;
    const std::vector<ipcr_hit> &H = s->hits; // sorted by (record, pattern, position)
    std::vector<ipcr_hit> wh;
    std::vector<uint32_t> by_pos; // the record's hits in position order: the windows move along it (a record of hundreds of windows
                                  // and hundreds of thousands of hits is then cut in one pass, not searched once per window)
    JoinCtx c{p, &s->products, emit, user};
    size_t i = 0;
    for (uint32_t r = 0; r < nrec && !c.aborted; ++r) {
        while (i < H.size() && H[i].record < r) ++i;
        size_t j = i;
        while (j < H.size() && H[j].record == r) ++j;
        by_pos.resize(j - i);
        for (size_t h = i; h < j; ++h) by_pos[h - i] = (uint32_t)(h - i);
        std::sort(by_pos.begin(), by_pos.end(), [&](uint32_t a, uint32_t b) { return H[i + a].pos != H[i + b].pos ? H[i + a].pos < H[i + b].pos : a < b; });
        size_t lo = 0; // first hit (in position order) at or behind the window's start: window starts only grow
        for (size_t w = first_window[r]; w < first_window[r + 1] && !c.aborted; ++w) {
            const ipcr_chunk_window &cw = s->windows[w];
            wh.clear();
            while (lo < by_pos.size() && H[i + by_pos[lo]].pos < cw.start) ++lo;
            for (size_t q = lo; q < by_pos.size() && H[i + by_pos[q]].pos < cw.end; ++q) {
                const ipcr_hit &hh = H[i + by_pos[q]];
                const uint32_t gid = hh.pattern & 0x7FFFFFFFu;
                const uint64_t L = gid < p->defs.size() ? p->defs[gid].seq.size() : 0;
                if (hh.pos + L > cw.end) continue; // the window must hold the whole site (ac.go:188-190)
                wh.push_back(hh);
                wh.back().pos -= cw.start;
                wh.back().record = (uint32_t)w;
            }
            if (wh.empty()) continue;
            std::sort(wh.begin(), wh.end(), HitLess()); // back into (pattern, position) order: what the join reads
            // a hit's "seed span touched a reset byte" bit was taken in the record: it lies inside the hit's window, so it holds in
            // every window that holds the hit
            const uint8_t fl = (uint8_t)((cw.reset ? 1u : 0u) | (mode1 ? 2u : 0u));
            (void)join_one_record(p, wh, 0, wh.size(), (uint32_t)w, (int64_t)(cw.end - cw.start), fl, c);
        }
        i = j;
    }
    s->stats.products = s->products.size();
    s->products_in_windows = true;
    s->stats.join_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tj).count();
    s->stats.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (c.aborted) return fail(IPCR_ERR_ABORTED, "emit callback aborted the scan");
    return IPCR_OK;
}

ipcr_status ipcr_scratch_chain_after(ipcr_scratch *s, const ipcr_scratch *prev) {
    if (!s || !prev) return fail(IPCR_ERR_INVALID, "ipcr_scratch_chain_after: null argument");
    if (!s->stream || !prev->stream) return fail(IPCR_ERR_DEVICE, "host-only scratch cannot scan");
    if (s == prev) return fail(IPCR_ERR_INVALID, "ipcr_scratch_chain_after: a scratch cannot follow itself");
    if (s->pend.active) return fail(IPCR_ERR_INVALID, "ipcr_scratch_chain_after: a scan is in flight on this scratch");
    if (s->device != prev->device) return fail(IPCR_ERR_INVALID, "ipcr_scratch_chain_after: scratches of different devices");
    s->lane_next = prev->lane_used ? prev->lane_used : prev->own_lane; // stream order is the dependency
    return IPCR_OK;
}

ipcr_status ipcr_scan_genome_begin(const ipcr_panel *p, ipcr_scratch *s, const ipcr_genome *g) {
    ipcr_status st = scratch_ready(p, s);
    if (st == IPCR_OK) st = same_device(s, g);
    if (st != IPCR_OK) return st;
    DeviceGuard dg(s->device);
    s->stats.pack_ms = 0;
    return scan_enqueue(p, s, const_cast<ipcr_genome *>(g));
}

ipcr_status ipcr_scan_genome_end(const ipcr_panel *p, ipcr_scratch *s, const ipcr_genome *g, ipcr_emit_fn emit, void *user) {
    ipcr_status st = scratch_ready(p, s);
    if (st == IPCR_OK) st = same_device(s, g);
    if (st != IPCR_OK) return st;
    DeviceGuard dg(s->device);
    st = scan_collect(p, s, const_cast<ipcr_genome *>(g));
    if (st != IPCR_OK) return st;
    std::vector<uint8_t> fl(g->rec_start.size());
    const bool any = genome_any_reset(g);
    for (size_t r = 0; r < fl.size(); ++r) fl[r] = (uint8_t)((g->flags[r] & 1u) | (any ? 2u : 0u));
    const auto tj = std::chrono::steady_clock::now();
    st = join_sorted_hits(p, s, g->rec_len.data(), fl.data(), (uint32_t)fl.size(), emit, user);
    s->stats.join_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tj).count();
    s->stats.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - s->pend.t0).count();
    trace("joined", s);
    return st;
}

ipcr_status ipcr_join_hits(const ipcr_panel *p, ipcr_scratch *s, const ipcr_hit *hits, int64_t n_hits,
                           const uint64_t *record_len, const uint8_t *record_flags, uint32_t n_records,
                           ipcr_emit_fn emit, void *user) {
    ipcr_status st = scratch_ready(p, s, false);
    if (st != IPCR_OK) return st;
    if ((n_hits > 0 && !hits) || (n_records > 0 && !record_len) || n_hits < 0) return fail(IPCR_ERR_INVALID, "ipcr_join_hits: null argument");
    s->hits_raw.assign(hits, hits + n_hits);
    sort_hits(s->hits_raw, s->hits, n_records, (uint32_t)p->defs.size());
    s->products.clear();
    s->products_in_windows = false;
    s->last_was_chunk = false;
    return join_sorted_hits(p, s, record_len, record_flags, n_records, emit, user);
}

// Caller's bytes -> a pinned slice (16-byte aligned) with non-temporal stores: the slice is read next by the DMA engine,
// not by this core, and a plain memcpy's read-for-ownership of the destination lines costs a third of the memory
// traffic (measured per 4 MB chunk with 8 threads copying at once: memcpy 0.27 ms, this 0.17 ms).
static void stream_copy(uint8_t *dst, const uint8_t *src, uint64_t n) {
    static const bool plain = getenv("IPCR_CHUNK_NTCOPY") && atoi(getenv("IPCR_CHUNK_NTCOPY")) == 0;
    uint64_t i = 0;
    if (!plain && (reinterpret_cast<uintptr_t>(dst) & 15u) == 0) {
        for (; i + 64 <= n; i += 64) {
            const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src + i));
            const __m128i b = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src + i + 16));
            const __m128i c = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src + i + 32));
            const __m128i d = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src + i + 48));
            _mm_stream_si128(reinterpret_cast<__m128i *>(dst + i), a);
            _mm_stream_si128(reinterpret_cast<__m128i *>(dst + i + 16), b);
            _mm_stream_si128(reinterpret_cast<__m128i *>(dst + i + 32), c);
            _mm_stream_si128(reinterpret_cast<__m128i *>(dst + i + 48), d);
        }
        _mm_sfence(); // the DMA descriptor written next must not overtake the streamed lines
    }
    if (i < n) memcpy(dst + i, src + i, n - i);
}

ipcr_status ipcr_scan_chunk(const ipcr_panel *p, ipcr_scratch *s, const uint8_t *seq, uint64_t len,
                            ipcr_emit_fn emit, void *user) {
    const auto t0 = std::chrono::steady_clock::now();
    ipcr_status st = scratch_ready(p, s);
    if (st != IPCR_OK) return st;
    if (!seq && len) return fail(IPCR_ERR_INVALID, "null sequence");
    s->hits.clear();
    s->products.clear();
    s->products_in_windows = false;
    memset(&s->stats, 0, sizeof s->stats);
    s->last_was_chunk = p->id.empty();
    if (p->id.empty()) return IPCR_OK; // compiled.go:163-165
    DeviceGuard dg(s->device); // the worker's thread may never have selected a device (a goroutine on any thread)
    const uint64_t need_cols = record_cols(len) + 64;
    if (!s->chunk || s->chunk->cap_cols < need_cols) {
        if (s->chunk) ipcr_genome_destroy(s->chunk);
        s->chunk = nullptr;
        st = ipcr_genome_create_on((need_cols + (need_cols >> 2)) * IPCR_COLUMN_BASES, 1, s->device, &s->chunk);
        if (st != IPCR_OK) return st;
        // the private chunk genome lives on the scratch's stream: copy, pack, sweep and hand-over are one
        // in-order sequence and the host waits once, at the end
        (void)hipStreamDestroy(s->chunk->stream);
        s->chunk->stream = s->stream;
        s->chunk->shared_stream = true;
        HIPCHK(hipMemsetAsync(s->chunk->d_block_rec, 0, (s->chunk->cap_cols / 64 + 2) * 4ull, s->stream)); // one record: every block is record 0's
    }
    ipcr_genome *g = s->chunk;
    genome_clear(g, true);
    // Whatever goes wrong between here and the hand-over -- any early return below -- no DMA and no kernel may still read the
    // caller's bytes or a pinned slab when the call returns: the caller may free the one, the next call rewrites the other.
    struct Drain {
        hipStream_t st;
        bool armed = true;
        ~Drain() { if (armed) (void)hipStreamSynchronize(st); }
    } drain{g->stream};
    uint32_t *pinned_flag = pinned_seq(s) + 4; // set when the record holds a byte outside ACGTacgt (by the pack kernel, or by the host's packer)
    *pinned_flag = 0u;
    const int live = p->live_scratches->load(std::memory_order_relaxed);
    // How the bases reach the device.  (a) Packed on the host into 2-bit + invalid-bit planes while they are staged in
    // pinned memory -- 0.375 bytes per base on the link instead of 1 (hostpack.cpp), tiles made from them on the device:
    // what a pool of workers does (every worker packs its own chunk), and a single worker with a record of 16 Mb or more
    // (the process's pack pool shares the slices).  (b) As ASCII: a single worker with a small chunk -- the runtime's
    // pageable copy pins the caller's pages in place and runs at the link rate, faster than one core packs.
    // IPCR_CHUNK_HOSTPACK=0/1 forces.
    const char *hp_str = getenv("IPCR_CHUNK_HOSTPACK"); // (read per call: the tests run both forms in one process)
    const int hp_env = hp_str ? atoi(hp_str) : -1;
    // (c) Where the packer writes through the PCIe BAR (round 4) it is the faster way at EVERY size -- a lone worker's 50 kb / 150 kb /
    // 400 kb / 900 kb chunks: 45 / 47 / 56 / 58 us per call against 59 / 57 / 72 / 93 as ASCII -- and the chunk's reset flag is known
    // before the sweep is launched.
    const BarInfo bar_info = device_bar(slot_phys(g->device));
    const bool bar_on = bar_info.writable && env_flag("IPCR_CHUNK_BAR", true);
    const bool hostpack = hp_env >= 0 ? hp_env != 0 : (ipcr::pack_linear_is_simd() && (bar_on || live > 1 || (len >= (1ull << 20) && PackPool::get().size() > 1) || len >= (16ull << 20)));
    if (hostpack) {
        const auto th0 = std::chrono::steady_clock::now();
        const uint64_t cols = record_cols(len), col0 = g->next_col;
        if (g->rec_start.size() >= g->max_records || col0 + cols > g->cap_cols) return fail(IPCR_ERR_CAPACITY, "chunk genome capacity exceeded");
        const uint64_t dev_bytes = cols * 2048ull; // four planes x 128 words per column
        bool bar = bar_on;
        if (dev_bytes > g->staging_cap || (bar && !g->staging_fine)) {
            if (g->staging) (void)hipFree(g->staging);
            g->staging = nullptr;
            g->staging_fine = false;
            g->staging_cap = std::max(g->staging_cap, dev_bytes + (dev_bytes >> 3));
            // fine-grained: the device reads what the host has just written through the BAR past its L2, never a stale line
            if (bar && hipExtMallocWithFlags((void **)&g->staging, g->staging_cap, hipDeviceMallocFinegrained) == hipSuccess) g->staging_fine = true;
            else {
                if (bar) { (void)hipGetLastError(); host_writable_off(slot_phys(g->device)); } // no such memory here: pinned slabs + DMA from now on
                g->staging = nullptr;
                HIPCHK(hipMalloc((void **)&g->staging, g->staging_cap));
            }
        }
        bar = bar && g->staging_fine;
        const bool pooled = live <= 1 && PackPool::get().size() > 1 && cols >= 64;
        // columns per slice: up to 1024 = 4 Mbases, i.e. a worker's 4 Mb chunk is ONE copy + ONE conversion launch.  Cutting it
        // in two or four (IPCR_CHUNK_SPLIT: the first part crosses the link while the next is packed) was slower under a
        // pool -- 16 workers: 94 / 72 / 49 Gbases/s for 1 / 2 / 4 parts -- every operation on a stream costs its dispatch latency
        static const uint64_t split_env = getenv("IPCR_CHUNK_SPLIT") ? strtoull(getenv("IPCR_CHUNK_SPLIT"), nullptr, 10) : 1;
        // A LONE worker (internal/pipeline/pipeline.go:60-125 with Threads = 1; a genome of few large records): a GROUP of
        // columns is packed by the process's pool, every thread a run of its columns into the group's planes, and sent as
        // ONE conversion launch that reads the pinned planes over the link itself; the next group is packed under it.  Two
        // groups for a 4 Mb chunk (pack 0 | pack 1 under transfer 0 | transfer 1, sweep), 8 Mb groups for a chromosome.
        const uint64_t SLC = pooled ? 2048 : std::min<uint64_t>(1024, std::max<uint64_t>(128, (cols + split_env - 1) / std::max<uint64_t>(split_env, 1)));
        // first columns of the slices (+ the end).  The lone worker's chunk of up to 8 Mb goes as TWO groups, three quarters and
        // a quarter: what follows the packing on the device -- the last group's way over the link, its conversion -- is then
        // short, and the first group's transfer hides under the packing of the second
        std::vector<uint64_t> gs;
        if (pooled && cols <= 2048 && cols >= 256) { // (one group instead, now that the packer writes through the BAR: no difference, 47-52 Gbases/s either way)
            const uint64_t first = std::min<uint64_t>(cols - 64, (cols * 3 / 4 + 7) / 8 * 8);
            gs = {0, first, cols};
        } else {
            for (uint64_t c = 0; c < cols; c += SLC) gs.push_back(c);
            gs.push_back(cols);
        }
        const uint64_t nsl = gs.size() - 1;
        std::vector<uint32_t> sflags((size_t)nsl, 0);
        // columns [c0, c0 + nc) of slice i (whose first column is gs[i] and which holds snc columns) into the slice's planes
        // at `slab`: [lo | hi | inv | rst], snc x 128 words each
        // (bar: the two code planes go straight into the slice's place in device memory, write-only, through the BAR; the
        // invalid and reset planes stay in the pinned slab and follow by DMA only if the slice holds such a byte)
        auto pack_cols = [&](uint64_t i, uint8_t *slab, uint64_t c0, uint64_t nc) -> uint32_t {
            const uint64_t s0 = gs[(size_t)i], snc = gs[(size_t)i + 1] - s0, W = snc * 128u, b0 = c0 * IPCR_COLUMN_BASES;
            const uint64_t nb = b0 < len ? std::min<uint64_t>(len - b0, nc * IPCR_COLUMN_BASES) : 0;
            uint32_t *w = reinterpret_cast<uint32_t *>(slab) + (c0 - s0) * 128u;
            uint32_t *wd = bar ? reinterpret_cast<uint32_t *>(g->staging + s0 * 2048ull) + (c0 - s0) * 128u : w;
            return ipcr::pack_linear(seq + (nb ? b0 : 0), nb, nc * IPCR_COLUMN_BASES, wd, wd + W, w + 2 * W, w + 3 * W);
        };
        auto pack_slice = [&](uint64_t i, uint8_t *slab) { // planes of slice i: [lo | hi | inv | rst], nc x 128 words each
            const uint64_t c0 = gs[(size_t)i], nc = gs[(size_t)i + 1] - c0;
            sflags[(size_t)i] = pack_cols(i, slab, c0, nc);
        };
        // Runs of N are short and far between (a genome's 0.1 %: one column of 4096 bases in a hundred holds one): of a slice that
        // holds an invalid base only the columns whose invalid plane holds a bit cross the link -- 512 bytes each, through the BAR --
        // and a bitmap in the (unused) reset plane's place tells the conversion kernel which; it makes the others' bits itself.
        // Whoever packed the columns looks them over (the pool's threads for a lone worker's groups), the sender writes the bitmap.
        static const bool inv_cols = env_flag("IPCR_CHUNK_INV_COLUMNS", true);
        std::vector<uint32_t> colbits((size_t)nsl * 64u, 0);
        auto mark_columns = [&](uint64_t i, const uint8_t *slab, uint64_t c0, uint64_t nc) { // columns [c0, c0 + nc) of slice i
            const uint64_t s0 = gs[(size_t)i], snc = gs[(size_t)i + 1] - s0, W = snc * 128u;
            if (snc > 2048) return;
            uint8_t *d = g->staging + s0 * 2048ull;
            const uint32_t *hiv = reinterpret_cast<const uint32_t *>(slab) + 2 * W;
            for (uint64_t c = c0 - s0; c < c0 - s0 + nc; ++c) {
                uint32_t any = 0;
                for (uint32_t k = 0; k < 128u; ++k) any |= hiv[c * 128u + k];
                if (any) {
                    __atomic_fetch_or(&colbits[(size_t)i * 64u + (size_t)(c >> 5)], 1u << (c & 31u), __ATOMIC_RELAXED);
                    bar_copy(d + W * 8u + c * 512u, slab + W * 8u + c * 512u, 512u);
                }
            }
        };
        std::vector<uint8_t> marked((size_t)nsl, 0); // the slice's columns have been looked over by its packers
        auto send_slice = [&](uint64_t i, const uint8_t *slab) -> ipcr_status { // the rst plane crosses the link only if the slice holds lower case
            const uint64_t c0 = gs[(size_t)i], nc = gs[(size_t)i + 1] - c0, W = nc * 128u;
            const bool lower = (sflags[(size_t)i] & 2u) != 0;
            const uint32_t *iv_cols = nullptr; // (device) one bit per column: its invalid plane has crossed the link
            // ... and the invalid-bit plane only if it holds a byte outside ACGT at all: the conversion kernel knows where the
            // record ends and makes the padding's bits itself (0.25 B/base on the link: IPCR_CHUNK_SKIP_INV=0 sends it always)
            const bool need_inv = lower || (sflags[(size_t)i] & 1u) != 0 || !env_flag("IPCR_CHUNK_SKIP_INV", true);
            uint8_t *d = g->staging + c0 * 2048ull;
            // A lone caller: the conversion kernel reads the pinned slab over the link itself, no copy operation in between
            // (whole 150 Mb record: 60 Gbases/s against 51).  A pool of workers keeps the copy engine: their kernels would
            // otherwise wait on the link with the compute units held (16 workers: 78 Gbases/s against 99).
            // IPCR_CHUNK_ZEROCOPY=0/1 forces.
            static const int zc_env = getenv("IPCR_CHUNK_ZEROCOPY") ? atoi(getenv("IPCR_CHUNK_ZEROCOPY")) : -1;
            const bool zerocopy = !bar && (zc_env >= 0 ? zc_env != 0 : live <= 1);
            if (bar) { // the code planes are there already; the other two follow only if the slice needs them -- through the BAR as well
                // (a copy operation per dirty chunk cost a stream of chunks with N a third of its rate: 16 workers 108 Gbases/s against 157)
                if (need_inv) {
                    static const bool inv_dma = env_flag("IPCR_CHUNK_INV_DMA", false);
                    if (inv_dma) HIPCHK(hipMemcpyAsync(d + W * 8u, slab + W * 8u, W * 4u * (lower ? 2u : 1u), hipMemcpyHostToDevice, g->stream));
                    else if (lower || !inv_cols || nc > 2048) bar_copy(d + W * 8u, slab + W * 8u, W * 4u * (lower ? 2u : 1u));
                    else {
                        if (!marked[(size_t)i]) mark_columns(i, slab, c0, nc); // (a worker's own slice: nobody has looked yet)
                        alignas(64) uint32_t bits[64];
                        memcpy(bits, &colbits[(size_t)i * 64u], sizeof bits);
                        bar_copy(d + W * 12u, reinterpret_cast<const uint8_t *>(bits), sizeof bits);
                        iv_cols = reinterpret_cast<const uint32_t *>(d + W * 12u);
                    }
                }
                bar_flush(bar_info);
            } else if (!zerocopy) HIPCHK(hipMemcpyAsync(d, slab, W * 4u * (lower ? 4u : need_inv ? 3u : 2u), hipMemcpyHostToDevice, g->stream));
            const uint32_t *dl = zerocopy ? reinterpret_cast<const uint32_t *>(slab) : reinterpret_cast<const uint32_t *>(d);
            HIPCHK(ipcr::launch_tiles_from_linear(g->stream, dl, dl + W, need_inv ? dl + 2 * W : nullptr, lower ? dl + 3 * W : nullptr, col0, col0 + c0, nc, len,
                                                  g->planes, g->rst, i == 0 ? g->d_rec_start : nullptr, i == 0 ? g->d_rec_len : nullptr,
                                                  i == 0 ? g->e0 : nullptr, i + 1 == nsl ? g->e1 : nullptr, iv_cols));
            return IPCR_OK;
        };
        if (!pooled) { // this worker's own two pinned slices: slice i + 1 is packed while slice i crosses the link
            constexpr uint64_t SLAB = 8ull << 20;
            for (int h = 0; h < 2; ++h) {
                if (!s->h_stage[h]) HIPCHK(hipHostMalloc((void **)&s->h_stage[h], SLAB, hipHostMallocDefault));
                if (!s->ev_stage[h]) HIPCHK(hipEventCreateWithFlags(&s->ev_stage[h], hipEventDisableTiming));
            }
            for (uint64_t i = 0; i < nsl; ++i) {
                const int h = (int)(i & 1u);
                if (i >= 2) HIPCHK(hipEventSynchronize(s->ev_stage[h])); // the DMA that read this slab has finished
                pack_slice(i, s->h_stage[h]);
                st = send_slice(i, s->h_stage[h]);
                if (st != IPCR_OK) return st;
                if (i + 2 < nsl) HIPCHK(hipEventRecord(s->ev_stage[h], g->stream));
            }
        } else { // one worker: the pool packs group i + 1 while group i crosses the link
            if (dev_bytes > s->h_planes_cap) {
                if (s->h_planes) (void)hipHostFree(s->h_planes);
                s->h_planes = nullptr;
                s->h_planes_cap = dev_bytes + (dev_bytes >> 3);
                HIPCHK(hipHostMalloc((void **)&s->h_planes, s->h_planes_cap, hipHostMallocDefault));
            }
            // ONE pool run over the items of all groups, group by group; the caller's thread does not pack: it sends group i
            // (one conversion launch that reads the pinned planes over the link) the moment the group's last item is done, while
            // the pool is already packing group i + 1
            const uint64_t nthreads = std::max<uint64_t>(1, PackPool::get().size() - 1);
            struct Item { uint64_t group, c0, nc; };
            std::vector<Item> items;
            std::vector<uint32_t> group_items((size_t)nsl, 0);
            for (uint64_t i = 0; i < nsl; ++i) {
                const uint64_t c0 = gs[(size_t)i], nc = gs[(size_t)i + 1] - c0;
                const uint64_t per = std::max<uint64_t>(32, ((nc + nthreads - 1) / nthreads + 7) / 8 * 8); // columns per item: 128 KB of bases at least
                for (uint64_t a = c0; a < c0 + nc; a += per) { items.push_back({i, a, std::min<uint64_t>(per, c0 + nc - a)}); ++group_items[(size_t)i]; }
            }
            std::vector<uint32_t> iflags(items.size(), 0);
            std::unique_ptr<std::atomic<uint32_t>[]> group_done(new std::atomic<uint32_t>[(size_t)nsl]);
            for (uint64_t i = 0; i < nsl; ++i) group_done[(size_t)i].store(0);
            uint64_t sent = 0;
            ipcr_status send_st = IPCR_OK;
            auto send_ready = [&]() {
                while (sent < nsl && send_st == IPCR_OK && group_done[(size_t)sent].load(std::memory_order_acquire) == group_items[(size_t)sent]) {
                    for (size_t k = 0; k < items.size(); ++k)
                        if (items[k].group == sent) sflags[(size_t)sent] |= iflags[k];
                    trace("group packed", s);
                    send_st = send_slice(sent, s->h_planes + gs[(size_t)sent] * 2048ull);
                    ++sent;
                }
            };
            const std::function<void()> idle = send_ready;
            trace("pack>", s);
            static const bool item_times = getenv("IPCR_DEBUG_TIMES") != nullptr;
            std::vector<double> it0(item_times ? items.size() : 0), it1(item_times ? items.size() : 0);
            std::vector<int> itcpu(item_times ? items.size() : 0);
            if (bar && inv_cols) std::fill(marked.begin(), marked.end(), (uint8_t)1); // (every item looks its own columns over)
            const auto tp0 = std::chrono::steady_clock::now();
            PackPool::get().run(items.size(), [&](size_t k) {
                const Item &it = items[k];
                if (item_times) { it0[k] = ms_since(tp0) * 1000.0; itcpu[k] = sched_getcpu(); }
                iflags[k] = pack_cols(it.group, s->h_planes + gs[(size_t)it.group] * 2048ull, it.c0, it.nc);
                if (bar && inv_cols && (iflags[k] & 1u)) mark_columns(it.group, s->h_planes + gs[(size_t)it.group] * 2048ull, it.c0, it.nc);
                if (item_times) it1[k] = ms_since(tp0) * 1000.0;
                group_done[(size_t)it.group].fetch_add(1, std::memory_order_release);
            }, slot_phys(g->device), &idle);
            send_ready();
            if (item_times)
                for (size_t k = 0; k < items.size(); ++k)
                    fprintf(stderr, "    item %2zu group %llu cols %4llu cpu %3d: %6.1f .. %6.1f us\n", k, (unsigned long long)items[k].group,
                            (unsigned long long)items[k].nc, itcpu[k], it0[k], it1[k]);
            if (send_st != IPCR_OK) return send_st;
        }
        uint32_t fl = 0;
        for (uint32_t f : sflags) fl |= f;
        *pinned_flag = fl & 1u;
        genome_account_record(g, len, cols);
        s->stats.hostpack_ms = ms_since(th0);
    } else {
    constexpr uint64_t SLICE_MAX = 8ull << 20;
    // One worker: the runtime's pageable copy (it pins the caller's pages in place) runs at the link rate.  Several
    // workers calling it at once serialise inside the runtime (8 workers x 4 Mb chunks: 12 Gbases/s in all, a single
    // worker 22): then every worker stages through its own pinned slices instead.  IPCR_CHUNK_STAGING=0/1 forces.
    static const int force = getenv("IPCR_CHUNK_STAGING") ? atoi(getenv("IPCR_CHUNK_STAGING")) : -1;
    const bool staged = force >= 0 ? force != 0 : live > 1;
    const uint8_t *pack_src = nullptr;
    if (len + 16 > g->staging_cap) {
        if (g->staging) (void)hipFree(g->staging);
        g->staging = nullptr;
        g->staging_cap = len + 16 + (len >> 3);
        HIPCHK(hipMalloc((void **)&g->staging, g->staging_cap));
    }
    if (len) {
        pack_src = g->staging;
        if (!staged) {
            HIPCHK(hipMemcpyAsync(g->staging, seq, len, hipMemcpyHostToDevice, g->stream));
        } else {
            static const uint64_t slice_env = getenv("IPCR_CHUNK_SLICE") ? strtoull(getenv("IPCR_CHUNK_SLICE"), nullptr, 10) : 0;
            // A chunk of up to 8 MiB goes in two halves: the DMA of the first runs under the CPU copy of the second, and
            // two slices need no marker between them (8 workers x 4 Mb chunks: one slice 49.0, two 50.6, four 49.3
            // Gbases/s; every further operation on the stream costs its own dispatch latency).  Anything larger -- a
            // whole chromosome -- is double-buffered in 8 MiB slices.
            const uint64_t halves = ((len + 1) / 2 + 4095) & ~4095ull;
            const uint64_t SLICE = slice_env ? std::min<uint64_t>(std::max<uint64_t>(slice_env, 65536), SLICE_MAX)
                                             : (len <= SLICE_MAX ? std::max<uint64_t>(halves, 65536) : SLICE_MAX);
            for (int h = 0; h < 2; ++h) {
                if (!s->h_stage[h]) HIPCHK(hipHostMalloc((void **)&s->h_stage[h], SLICE_MAX, hipHostMallocDefault));
                if (!s->ev_stage[h]) HIPCHK(hipEventCreateWithFlags(&s->ev_stage[h], hipEventDisableTiming));
            }
            uint64_t i = 0;
            for (uint64_t off = 0; off < len; off += SLICE, ++i) {
                const int h = (int)(i & 1u);
                const uint64_t n = std::min<uint64_t>(SLICE, len - off);
                if (i >= 2) HIPCHK(hipEventSynchronize(s->ev_stage[h])); // the DMA that read this slice has finished
                stream_copy(s->h_stage[h], seq + off, n);
                HIPCHK(hipMemcpyAsync(g->staging + off, s->h_stage[h], n, hipMemcpyHostToDevice, g->stream));
                if (off + 2 * SLICE < len) HIPCHK(hipEventRecord(s->ev_stage[h], g->stream)); // slice i + 2 will wait for it; a chunk of one or two slices needs no marker
            }
        }
    }
    st = genome_add_device(g, pack_src, len, false, pinned_flag);
    if (st != IPCR_OK) return st;
    }
    const double hostpack_keep = s->stats.hostpack_ms; // (scan_enqueue starts the statistics afresh)
    st = scan_enqueue(p, s, g, true, hostpack ? (int)(*pinned_flag & 1u) : -1);
    if (st == IPCR_OK) st = scan_collect(p, s, g);
    if (st != IPCR_OK) return st;
    drain.armed = false; // the scan has been collected: the stream has passed everything this call queued
    s->stats.hostpack_ms = hostpack_keep;
    {
        float ms = 0; // the pack kernel's events lie in front of the sweep on the same stream
        HIPCHK(hipEventElapsedTime(&ms, g->e0, g->e1));
        g->pack_ms += ms;
        s->stats.pack_ms = ms;
    }
    g->flags.assign(1, (uint8_t)(pinned_seq(s)[4] & 1u));
    g->flags_valid = true;
    const uint8_t fl = (uint8_t)((g->flags[0] & 1u) | (s->pend.mode == 1 ? 2u : 0u));
    st = join_sorted_hits(p, s, g->rec_len.data(), &fl, 1, emit, user);
    s->last_was_chunk = st == IPCR_OK; // the products' amplicons lie in s->chunk until the next scan (ipcr_probe_scratch_products)
    s->stats.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return st;
}

// ------------------------------------------------------------------------------ probe

static ipcr_status normalize_probe(const char *probe, std::string &prb) {
    prb.clear();
    for (const char *q = probe; *q; ++q) { // core/primer/validate.go:12-22 Normalize
        char ch = *q;
        if (ch == ' ' || ch == '\t' || ch == '\n' || ch == '\r' || ch == '\v' || ch == '\f' || ch == '\'' || ch == '"') continue;
        if (ch >= 'a' && ch <= 'z') ch = (char)(ch - 32);
        prb.push_back(ch);
    }
    for (size_t i = 0; i < prb.size(); ++i) // validate.go:27-36
        if (!strchr("ACGTRYSWKMBDHVN", prb[i]))
            return fail(IPCR_ERR_PRIMER, "invalid probe base %c at position %zu; allowed: A C G T R Y S W K M B D H V N", prb[i], i + 1);
    if (prb.size() > IPCR_MAX_PRIMER_LEN) return fail(IPCR_ERR_UNSUPPORTED, "probe of %zu nt exceeds IPCR_MAX_PRIMER_LEN", prb.size());
    return IPCR_OK;
}

// probe and rc(probe) as the two rows of IUPAC masks the probe kernel reads; returns the kernel's fast-path flag
static uint32_t probe_masks(const std::string &prb, int32_t max_mm, uint8_t *masks /* 256 bytes */) {
    std::string rc;
    revcomp(prb, rc, nullptr);
    memset(masks, 0, 256);
    bool strict = !prb.empty();
    for (size_t i = 0; i < prb.size(); ++i) {
        masks[i] = T.mask[(uint8_t)prb[i]];
        masks[128 + i] = T.mask[(uint8_t)rc[i]];
        if (prb[i] != 'A' && prb[i] != 'C' && prb[i] != 'G' && prb[i] != 'T') strict = false;
    }
    return (max_mm == 0 && strict) ? 1u : 0u; // oligo.go:33-42
}

// ipcr_probe_best_hit is what a host without the batched form calls once per product (visitors.Probe.Visit runs on the
// collector goroutine, internal/visitors/probe.go:18-33), next to a pool of workers that sweep on the same device.  So it
// must not allocate or free device memory (hipFree waits for the whole device: every worker's sweep), use the null stream
// or a copy operation.  A call borrows one of these from a free list (one per concurrent caller ever seen, per device
// slot; they live as long as the process): a non-blocking stream and ONE pinned block the kernel reads its arguments
// and the amplicon from (it stages them in LDS) and writes its 16-byte result to, tagged, where the caller spins.
struct ProbeCtx {
    int slot = 0;
    hipStream_t st = nullptr;
    uint8_t *h = nullptr; // [0, 256) masks | [256, 272) result | [272, 288) the two offsets | [320, ...) amplicon + 16 spare bytes
    uint64_t hcap = 0;
    uint32_t tag = 0;
};
static std::mutex g_probe_mu;
static std::vector<ProbeCtx *> g_probe_free;
static constexpr uint64_t PROBE_CTX_AMP_OFF = 320;

static ipcr_status probe_ctx_acquire(int slot, uint64_t amp_len, ProbeCtx **out) {
    ProbeCtx *c = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_probe_mu);
        for (size_t i = 0; i < g_probe_free.size(); ++i)
            if (g_probe_free[i]->slot == slot) { c = g_probe_free[i]; g_probe_free.erase(g_probe_free.begin() + (long)i); break; }
    }
    if (!c) {
        c = new ProbeCtx;
        c->slot = slot;
        const hipError_t e = hipStreamCreateWithFlags(&c->st, hipStreamNonBlocking);
        if (e != hipSuccess) { delete c; return fail(IPCR_ERR_DEVICE, "hipStreamCreateWithFlags: %s", hipGetErrorString(e)); }
    }
    const uint64_t need = PROBE_CTX_AMP_OFF + amp_len + 64;
    if (need > c->hcap) { // (first use, or an amplicon beyond 64 KB: --max-length is 2000 by default)
        if (c->h) (void)hipHostFree(c->h);
        c->h = nullptr;
        c->hcap = std::max<uint64_t>(need + (need >> 2), 64u << 10);
        const hipError_t e = hipHostMalloc((void **)&c->h, c->hcap, hipHostMallocDefault);
        if (e != hipSuccess) {
            c->hcap = 0;
            std::lock_guard<std::mutex> lk(g_probe_mu);
            g_probe_free.push_back(c);
            return fail(IPCR_ERR_DEVICE, "hipHostMalloc: %s", hipGetErrorString(e));
        }
        memset(c->h, 0, PROBE_CTX_AMP_OFF);
    }
    *out = c;
    return IPCR_OK;
}
static void probe_ctx_release(ProbeCtx *c) {
    std::lock_guard<std::mutex> lk(g_probe_mu);
    g_probe_free.push_back(c);
}

ipcr_status ipcr_probe_best_hit(const uint8_t *amplicon, uint64_t len, const char *probe, int32_t max_mm,
                                ipcr_probe_hit *out) {
    if (!out || !probe || (!amplicon && len)) return fail(IPCR_ERR_INVALID, "ipcr_probe_best_hit: null argument");
    memset(out, 0, sizeof *out);
    std::string prb;
    ipcr_status st = normalize_probe(probe, prb);
    if (st != IPCR_OK) return st;
    if (prb.empty()) return IPCR_OK; // oligo.go:21-23
    if (slot_count() == 0) return fail(IPCR_ERR_DEVICE, "no HIP device visible: the probe rescan has no CPU fallback");
    if (len < prb.size()) return IPCR_OK; // no window fits (core/primer/match.go:31-34)
    if (len > 0x7FFFFFFFull) return fail(IPCR_ERR_UNSUPPORTED, "amplicon of %llu bases: ipcr_probe_hit positions are 32-bit", (unsigned long long)len);
    const int slot = default_slot();
    DeviceGuard dg(slot);
    ProbeCtx *c = nullptr;
    st = probe_ctx_acquire(slot, len, &c);
    if (st != IPCR_OK) return st;
    const uint32_t fast = probe_masks(prb, max_mm, c->h);
    uint64_t *offs = reinterpret_cast<uint64_t *>(c->h + 272);
    offs[0] = 0;
    offs[1] = len;
    memcpy(c->h + PROBE_CTX_AMP_OFF, amplicon, len);
    c->tag = (c->tag % 0x3FFFFFFFu) + 1u; // never 0, never the tag the result slot still holds
    const uint32_t tag = c->tag;
    volatile int32_t *res = reinterpret_cast<volatile int32_t *>(c->h + 256);
    const hipError_t le = ipcr::launch_probe(c->st, c->h + PROBE_CTX_AMP_OFF, offs, 1, c->h, c->h + 128, (uint32_t)prb.size(),
                                             (uint32_t)(max_mm < 0 ? 0 : max_mm), fast, reinterpret_cast<ipcr_probe_rec *>(c->h + 256), tag);
    if (le != hipSuccess) { probe_ctx_release(c); return fail(IPCR_ERR_DEVICE, "probe kernel: %s", hipGetErrorString(le)); }
    // the record is one 16-byte store into pinned memory: spin on its tag (a kernel of one wave: microseconds)
    bool got = false;
    for (uint64_t spin = 1;; ++spin) {
        if (((uint32_t)__atomic_load_n(res, __ATOMIC_ACQUIRE) >> 1) == tag) { got = true; break; }
        if (spin < 0x4000u) __builtin_ia32_pause();
        else std::this_thread::yield();
        if ((spin & 0x3FFFu) == 0) {
            const hipError_t q = hipStreamQuery(c->st);
            if (q == hipErrorNotReady) continue;
            got = ((uint32_t)__atomic_load_n(res, __ATOMIC_ACQUIRE) >> 1) == tag; // the stream has drained
            if (!got) {
                probe_ctx_release(c);
                return fail(IPCR_ERR_DEVICE, "probe kernel: %s", q == hipSuccess ? "its result did not reach pinned memory" : hipGetErrorString(q));
            }
            break;
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    out->found = res[0] & 1;
    out->strand = res[1];
    out->pos = res[2];
    out->mm = res[3];
    if (!out->found) memset(out, 0, sizeof *out);
    // (the kernel reads nothing of the block after lane 0's store: the next call may rewrite it.  A wait for the stream
    // now and then lets the runtime retire its bookkeeping of the launches.)
    if ((tag & 0xFFu) == 0u) (void)hipStreamSynchronize(c->st);
    probe_ctx_release(c);
    return IPCR_OK;
}

ipcr_status ipcr_probe_products(ipcr_scratch *s, const ipcr_genome *g, const char *probe, int32_t max_mm,
                                ipcr_probe_hit *out, int64_t n_out) {
    const ipcr_status st = ipcr_probe_products_begin(s, g, probe, max_mm);
    return st != IPCR_OK ? st : ipcr_probe_products_end(s, out, n_out);
}

ipcr_status ipcr_probe_products_end(ipcr_scratch *s, ipcr_probe_hit *out, int64_t n_out) {
    if (!s || (!out && n_out)) return fail(IPCR_ERR_INVALID, "ipcr_probe_products_end: null argument");
    if (s->probe_pending < 0) return fail(IPCR_ERR_INVALID, "ipcr_probe_products_end without ipcr_probe_products_begin");
    const int64_t n = s->probe_pending;
    s->probe_pending = -1;
    if (n != n_out) return fail(IPCR_ERR_INVALID, "n_out (%lld) != products the rescan was begun for (%lld)", (long long)n_out, (long long)n);
    if (n == 0) return IPCR_OK;
    if (s->probe_res_off == 0) { memset(out, 0, (size_t)n * sizeof *out); return IPCR_OK; } // empty probe: nothing found (oligo.go:21-23)
    DeviceGuard dg(s->device);
    // every record is ONE tagged 16-byte store into pinned memory: spin on the tags (a handful of products per chunk)
    const volatile int32_t *res = reinterpret_cast<const volatile int32_t *>(s->h_probe + s->probe_res_off);
    const uint32_t tag = s->probe_tag;
    for (int64_t i = 0; i < n; ++i) {
        for (uint64_t spin = 1; ((uint32_t)__atomic_load_n(res + 4 * i, __ATOMIC_ACQUIRE) >> 1) != tag; ++spin) {
            if (spin < 0x4000u) __builtin_ia32_pause();
            else std::this_thread::yield();
            if ((spin & 0x3FFFu) == 0) {
                const hipError_t q = hipStreamQuery(s->probe_on);
                if (q == hipErrorNotReady) continue;
                if (((uint32_t)__atomic_load_n(res + 4 * i, __ATOMIC_ACQUIRE) >> 1) == tag) break; // the stream has drained
                return fail(IPCR_ERR_DEVICE, "probe rescan: %s", q == hipSuccess ? "its results did not reach pinned memory" : hipGetErrorString(q));
            }
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    for (int64_t i = 0; i < n; ++i) { // (ipcr_probe_rec is layout-identical; the tag rides above bit 0 of `found`)
        out[i].found = res[4 * i] & 1;
        out[i].strand = res[4 * i + 1];
        out[i].pos = res[4 * i + 2];
        out[i].mm = res[4 * i + 3];
    }
    return IPCR_OK;
}

// the batched rescan of the scratch's current products against the tiles of `g`: the resident genome the scan ran over, or
// the scratch's own chunk genome (ipcr_probe_scratch_products)
static ipcr_status probe_begin(ipcr_scratch *s, const ipcr_genome *g, const char *probe, int32_t max_mm) {
    if (s->probe_pending >= 0) return fail(IPCR_ERR_INVALID, "ipcr_probe_products_begin: the rescan begun before has not been ended");
    DeviceGuard dg(s->device);
    const size_t n = s->products.size();
    std::string prb;
    ipcr_status st = normalize_probe(probe, prb);
    if (st != IPCR_OK) return st;
    s->probe_res_off = 0;
    if (n == 0 || prb.empty()) { s->probe_pending = (int64_t)n; return IPCR_OK; }
    // The rescan runs on a lane of its own and its kernels take their arguments straight out of pinned host memory and
    // put the results there: chained scratches share ONE in-order stream on which the next pass's sweep is already
    // queued when the host gets here, and three small pageable copies + one back cost more than the two kernels.
    // pinned block: [0, 256) probe / rc masks | results | segments | offsets
    const uint64_t res_off = 256, seg_off = (res_off + n * sizeof(ipcr_probe_rec) + 15) & ~15ull;
    const uint64_t off_off = seg_off + n * sizeof(ipcr_amp_seg), hbytes = off_off + (n + 1) * 8;
    if (hbytes > s->h_probe_cap) {
        if (s->h_probe) (void)hipHostFree(s->h_probe);
        s->h_probe = nullptr;
        s->h_probe_cap = hbytes * 2;
        HIPCHK(hipHostMalloc((void **)&s->h_probe, s->h_probe_cap, hipHostMallocDefault));
    }
    if (!s->probe_stream) HIPCHK(hipStreamCreateWithFlags(&s->probe_stream, hipStreamNonBlocking));
    // amplicon = record[start:end], or record[start:] ++ record[:end] for wrap-around products
    // (internal/pipeline/pipeline.go:80-89)
    ipcr_amp_seg *segs = reinterpret_cast<ipcr_amp_seg *>(s->h_probe + seg_off);
    uint64_t *offs = reinterpret_cast<uint64_t *>(s->h_probe + off_off);
    offs[0] = 0;
    for (size_t i = 0; i < n; ++i) {
        ipcr_product pr = s->products[i];
        if (s->products_in_windows && g != s->chunk) { // products of ipcr_scan_genome_chunked: `record` is a window, coordinates are window-local
            if (pr.record < 0 || (size_t)pr.record >= s->windows.size()) return fail(IPCR_ERR_INVALID, "product window outside the window list");
            const ipcr_chunk_window &cw = s->windows[(size_t)pr.record];
            pr.record = (int32_t)cw.record;
            pr.start += (int64_t)cw.start;
            pr.end += (int64_t)cw.start;
        }
        if ((size_t)pr.record >= g->rec_start.size()) return fail(IPCR_ERR_INVALID, "product record outside genome");
        const uint64_t rs = g->rec_start[(size_t)pr.record], rl = g->rec_len[(size_t)pr.record];
        if (pr.start < 0 || pr.end < 0 || (uint64_t)pr.start > rl || (uint64_t)pr.end > rl) return fail(IPCR_ERR_INVALID, "product outside its record");
        ipcr_amp_seg sg{};
        if (pr.start <= pr.end) { sg.pa = rs + (uint64_t)pr.start; sg.len_a = (uint64_t)(pr.end - pr.start); sg.pb = rs; sg.len_b = 0; }
        else { sg.pa = rs + (uint64_t)pr.start; sg.len_a = rl - (uint64_t)pr.start; sg.pb = rs; sg.len_b = (uint64_t)pr.end; }
        sg.out_off = offs[i];
        offs[i + 1] = offs[i] + sg.len_a + sg.len_b;
        segs[i] = sg;
    }
    const uint64_t amp_bytes = offs[n] + 16;
    uint64_t longest = 0;
    for (size_t i = 0; i < n; ++i) longest = std::max(longest, offs[i + 1] - offs[i]);
    const uint32_t fast = probe_masks(prb, max_mm, s->h_probe);
    ipcr_probe_rec *res = reinterpret_cast<ipcr_probe_rec *>(s->h_probe + res_off);
    s->probe_tag = (s->probe_tag % 0x3FFFFFFFu) + 1u; // never 0, never what the result slots still hold
    // A chunk's products (the scratch's own tiles): the scratch's stream is idle -- its scan has been collected -- and a
    // second stream per worker only thins out the hardware queues a pool shares.  A resident genome: the lane of its own.
    const hipStream_t pst = g == s->chunk ? s->stream : s->probe_stream;
    s->probe_on = pst;
    if (longest <= ipcr::launch_probe_tiles_max()) {
        // ONE launch: every workgroup reads its amplicon straight from the tiles into LDS and rescans it there
        HIPCHK(ipcr::launch_probe_tiles(pst, g->planes, g->rst, segs, (uint32_t)n, s->h_probe, s->h_probe + 128, (uint32_t)prb.size(),
                                        (uint32_t)(max_mm < 0 ? 0 : max_mm), fast, res, s->probe_tag));
    } else { // an amplicon beyond the kernel's LDS stage (--max-length above 16 384): gathered to device memory first
        if (amp_bytes > s->amps_cap) {
            // (grows to the largest batch seen and stays: hipFree waits for the device, so a steady state must not come here)
            if (s->d_amps) (void)hipFree(s->d_amps);
            s->d_amps = nullptr;
            s->amps_cap = std::max<uint64_t>(amp_bytes + (amp_bytes >> 1), 1u << 20);
            HIPCHK(hipMalloc((void **)&s->d_amps, s->amps_cap));
        }
        HIPCHK(ipcr::launch_gather(pst, g->planes, g->rst, segs, (uint32_t)n, s->d_amps));
        HIPCHK(ipcr::launch_probe(pst, s->d_amps, offs, (uint32_t)n, s->h_probe, s->h_probe + 128, (uint32_t)prb.size(),
                                  (uint32_t)(max_mm < 0 ? 0 : max_mm), fast, res, s->probe_tag));
    }
    s->probe_res_off = res_off;
    s->probe_pending = (int64_t)n; // (set last: a begin that failed leaves nothing to end)
    return IPCR_OK;
}

ipcr_status ipcr_probe_products_begin(ipcr_scratch *s, const ipcr_genome *g, const char *probe, int32_t max_mm) {
    if (!s || !g || !probe) return fail(IPCR_ERR_INVALID, "ipcr_probe_products: null argument");
    if (!s->stream) return fail(IPCR_ERR_DEVICE, "host-only scratch: the probe rescan has no CPU fallback");
    { const ipcr_status ds = same_device(s, g); if (ds != IPCR_OK) return ds; }
    return probe_begin(s, g, probe, max_mm);
}

// ipcr-probe behind the drop-in call: the products of the scratch's last ipcr_scan_chunk, rescanned from the tiles that
// call has just packed (they stay in the scratch's private genome until its next scan) -- what visitors.Probe.Visit
// computes from p.Seq (internal/visitors/probe.go:18-33), which the pipeline slices chunk-locally on the worker
// (internal/pipeline/pipeline.go:80-89; wrap-around products of a circular record included)
ipcr_status ipcr_probe_scratch_products_begin(ipcr_scratch *s, const char *probe, int32_t max_mm) {
    if (!s || !probe) return fail(IPCR_ERR_INVALID, "ipcr_probe_scratch_products: null argument");
    if (!s->stream) return fail(IPCR_ERR_DEVICE, "host-only scratch: the probe rescan has no CPU fallback");
    if (!s->last_was_chunk || !s->chunk) {
        if (s->products.empty() && s->last_was_chunk) { s->probe_pending = 0; s->probe_res_off = 0; return IPCR_OK; } // (an empty panel's chunk scan packs nothing)
        return fail(IPCR_ERR_INVALID, "ipcr_probe_scratch_products: the scratch's last scan was not an ipcr_scan_chunk");
    }
    return probe_begin(s, s->chunk, probe, max_mm);
}

ipcr_status ipcr_probe_scratch_products(ipcr_scratch *s, const char *probe, int32_t max_mm, ipcr_probe_hit *out, int64_t n_out) {
    const ipcr_status st = ipcr_probe_scratch_products_begin(s, probe, max_mm);
    return st != IPCR_OK ? st : ipcr_probe_products_end(s, out, n_out);
}

// ------------------------------------------------------------------------------ nested PCR

ipcr_status ipcr_nested_windows(const ipcr_genome *g, const ipcr_window *windows, int64_t n64, const ipcr_panel *inner,
                                ipcr_scratch *s, ipcr_nested_hit *out) {
    ipcr_status st = scratch_ready(inner, s);
    if (st == IPCR_OK) st = same_device(s, g);
    if (st != IPCR_OK) return st;
    if (n64 < 0 || (n64 && (!windows || !out))) return fail(IPCR_ERR_INVALID, "ipcr_nested_windows: null argument");
    DeviceGuard dg(s->device);
    const size_t n = (size_t)n64;
    if (n == 0) return IPCR_OK;
    memset(out, 0, n * sizeof *out);
    st = genome_finalize(const_cast<ipcr_genome *>(g)); // the gather below reads the tiles: padding and packing must be complete
    if (st != IPCR_OK) return st;
    // amplicon = record[start:end], or record[start:] ++ record[:end] for wrap-around products
    // (internal/pipeline/pipeline.go:80-89); every amplicon starts 16-byte aligned (pack kernel input)
    std::vector<ipcr_amp_seg> segs(n);
    std::vector<uint64_t> offs(n), lens(n);
    uint64_t off = 0, cols = 0;
    for (size_t i = 0; i < n; ++i) {
        const ipcr_window &w = windows[i];
        if (w.record < 0 || (size_t)w.record >= g->rec_start.size()) return fail(IPCR_ERR_INVALID, "window %zu: record outside the genome", i);
        const uint64_t rs = g->rec_start[(size_t)w.record], rl = g->rec_len[(size_t)w.record];
        if (w.start < 0 || w.end < 0 || (uint64_t)w.start > rl || (uint64_t)w.end > rl) return fail(IPCR_ERR_INVALID, "window %zu outside its record", i);
        ipcr_amp_seg sg{};
        if (w.start <= w.end) { sg.pa = rs + (uint64_t)w.start; sg.len_a = (uint64_t)(w.end - w.start); sg.pb = rs; sg.len_b = 0; }
        else { sg.pa = rs + (uint64_t)w.start; sg.len_a = rl - (uint64_t)w.start; sg.pb = rs; sg.len_b = (uint64_t)w.end; }
        sg.out_off = off;
        offs[i] = off;
        lens[i] = sg.len_a + sg.len_b;
        off = (off + lens[i] + 15) & ~15ull;
        cols += record_cols(lens[i]);
        segs[i] = sg;
    }
    if (off + 16 > s->amps_cap) {
        if (s->d_amps) (void)hipFree(s->d_amps);
        s->d_amps = nullptr;
        s->amps_cap = off + 16 + (off >> 2);
        HIPCHK(hipMalloc((void **)&s->d_amps, s->amps_cap));
    }
    const uint64_t misc_bytes = n * (sizeof(ipcr_amp_seg) + sizeof(ipcr_pack_rec) + 4) + 256;
    if (misc_bytes > s->probe_misc_cap) {
        if (s->d_probe_misc) (void)hipFree(s->d_probe_misc);
        s->d_probe_misc = nullptr;
        s->probe_misc_cap = misc_bytes * 2;
        HIPCHK(hipMalloc(&s->d_probe_misc, s->probe_misc_cap));
    }
    ipcr_amp_seg *dsegs = static_cast<ipcr_amp_seg *>(s->d_probe_misc);
    HIPCHK(hipMemcpyAsync(dsegs, segs.data(), n * sizeof(ipcr_amp_seg), hipMemcpyHostToDevice, s->stream));
    HIPCHK(ipcr::launch_gather(s->stream, g->planes, g->rst, dsegs, (uint32_t)n, s->d_amps));
    // the amplicons become the records of a private genome on this scratch's stream: no waits between the packs
    if (!s->nest || s->nest->cap_cols < cols + 64 || s->nest->max_records < n) {
        if (s->nest) ipcr_genome_destroy(s->nest);
        s->nest = nullptr;
        st = ipcr_genome_create_on((cols + (cols >> 2) + 64) * IPCR_COLUMN_BASES, (uint32_t)std::max<size_t>(n + (n >> 2), 16), s->device, &s->nest);
        if (st != IPCR_OK) return st;
        (void)hipStreamDestroy(s->nest->stream);
        s->nest->stream = s->stream;
        s->nest->shared_stream = true;
    }
    genome_clear(s->nest);
    {   // one pack launch for all amplicons; its record table goes behind the segment table in the same device buffer
        const uint64_t seg_bytes = (n * sizeof(ipcr_amp_seg) + 63) & ~63ull;
        st = genome_add_device_batch(s->nest, s->d_amps, offs.data(), lens.data(), n, static_cast<uint8_t *>(s->d_probe_misc) + seg_bytes,
                                     (size_t)(s->probe_misc_cap - seg_bytes));
        if (st != IPCR_OK) return st;
    }
    st = scan_hits(inner, s, s->nest);
    if (st != IPCR_OK) return st;
    std::vector<uint8_t> fl(n);
    const bool any = genome_any_reset(s->nest);
    for (size_t r = 0; r < n; ++r) fl[r] = (uint8_t)((s->nest->flags[r] & 1u) | (any ? 2u : 0u));
    st = join_sorted_hits(inner, s, s->nest->rec_len.data(), fl.data(), (uint32_t)n, nullptr, nullptr);
    if (st != IPCR_OK) return st;
    // best inner product per amplicon: fewest total mismatches, longest, leftmost, end, pair ID (nested.go:35-51);
    // ties keep the engine's emission order (sort.SliceStable)
    for (const ipcr_product &pr : s->products) {
        ipcr_nested_hit &b = out[(size_t)pr.record];
        bool better = !b.found;
        if (!better) {
            const int mi = pr.fwd_mm + pr.rev_mm, mj = b.fwd_mm + b.rev_mm;
            if (mi != mj) better = mi < mj;
            else if (pr.length != b.length) better = pr.length > b.length;
            else if (pr.start != b.start) better = pr.start < b.start;
            else if (pr.end != b.end) better = pr.end < b.end;
            else better = inner->id[(size_t)pr.pair] < inner->id[(size_t)b.pair];
        }
        if (better) {
            b.found = 1; b.pair = pr.pair; b.type = pr.type; b.fwd_mm = pr.fwd_mm; b.rev_mm = pr.rev_mm;
            b.start = pr.start; b.end = pr.end; b.length = pr.length;
        }
    }
    return IPCR_OK;
}

ipcr_status ipcr_nested_products(const ipcr_scratch *outer, const ipcr_genome *g, const ipcr_panel *inner,
                                 ipcr_scratch *s, ipcr_nested_hit *out, int64_t n_out) {
    if (!outer) return fail(IPCR_ERR_INVALID, "ipcr_nested_products: null argument");
    if (outer == s) return fail(IPCR_ERR_INVALID, "ipcr_nested_products: the inner scan needs a scratch of its own");
    const size_t n = outer->products.size();
    if ((int64_t)n != n_out) return fail(IPCR_ERR_INVALID, "n_out (%lld) != products of the last scan (%zu)", (long long)n_out, n);
    std::vector<ipcr_window> w(n);
    for (size_t i = 0; i < n; ++i) {
        w[i].start = outer->products[i].start;
        w[i].end = outer->products[i].end;
        w[i].record = outer->products[i].record;
        w[i].reserved = 0;
    }
    return ipcr_nested_windows(g, w.data(), (int64_t)n, inner, s, out);
}

} // extern "C"
