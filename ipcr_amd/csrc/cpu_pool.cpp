// cpu_pool.cpp -- see cpu_pool.h
#include "cpu_pool.h"

#include <hip/hip_runtime_api.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>

namespace ipcr {
namespace {

// ------------------------------------------------------------ the CPUs next to a device
// A slab of pinned memory that cores of the OTHER socket have just written crosses the link at 31 GB/s instead of 54: the
// DMA engine's reads find the lines dirty in caches two hops away (measured round 3 -- the FASTA
// loader's threads bound to the far socket: 40 ms per GB, to the device's own: 26 ms; unbound it was the scheduler's luck).
// So the threads of this library that fill pinned memory run on the CPUs the kernel lists as local to the device
// (/sys/bus/pci/devices/<bus id>/local_cpulist), as far as the process is allowed on them.  IPCR_BIND_THREADS=0: never.
struct CpuSet {
    cpu_set_t set;
    bool known = false;
};
// What the process may run on, captured ONCE when the library is loaded: sched_getaffinity later would return the mask of
// whichever thread asks -- after ipcr_bind_thread_to_device has narrowed the main thread to one device's CPUs, every other
// device's set (allowed AND local) would come out empty and stay cached so.
const CpuSet g_initial_cpus = [] {
    CpuSet c;
    CPU_ZERO(&c.set);
    c.known = sched_getaffinity(0, sizeof c.set, &c.set) == 0;
    return c;
}();

const CpuSet &device_cpus(int phys) {
    static std::mutex mu;
    static std::map<int, CpuSet> cache;
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(phys);
    if (it != cache.end()) return it->second;
    CpuSet &c = cache[phys];
    CPU_ZERO(&c.set);
    if (const char *v = getenv("IPCR_BIND_THREADS")) if (*v && atoi(v) == 0) return c;
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, phys) != hipSuccess) return c;
    for (char *q = bus; *q; ++q) *q = (char)tolower((unsigned char)*q);
    const std::string path = std::string("/sys/bus/pci/devices/") + bus + "/local_cpulist";
    FILE *fh = fopen(path.c_str(), "r");
    if (!fh) return c;
    char line[4096] = {0};
    const bool got = fgets(line, sizeof line, fh) != nullptr;
    fclose(fh);
    if (!got) return c;
    cpu_set_t allowed = g_initial_cpus.set, local; // the process's mask as it was when the library was loaded, before anybody was bound
    CPU_ZERO(&local);
    if (!g_initial_cpus.known) return c;
    for (const char *q = line; *q;) { // "0-63,128-191"
        char *e = nullptr;
        const long a = strtol(q, &e, 10);
        if (e == q) break;
        long b = a;
        if (*e == '-') { q = e + 1; b = strtol(q, &e, 10); }
        for (long k = a; k <= b && k < CPU_SETSIZE; ++k) if (k >= 0) CPU_SET((int)k, &local);
        q = (*e == ',') ? e + 1 : e;
        if (*e != ',') break;
    }
    CPU_AND(&c.set, &allowed, &local);
    c.known = CPU_COUNT(&c.set) > 0 && CPU_COUNT(&c.set) < CPU_COUNT(&allowed); // nothing to choose on a one-socket host
    return c;
}
} // namespace

// the calling thread onto the device's CPUs; false when they are not known (or binding is off)
bool bind_this_thread(int phys) {
    const CpuSet &c = device_cpus(phys);
    return c.known && sched_setaffinity(0, sizeof c.set, &c.set) == 0;
}

namespace {

// The pack pool's threads, one physical core each, dealt round-robin over the L3 domains (CCDs) of the device's CPUs.  Left to
// the scheduler, ten of fifteen polling threads ended up on ONE CCD (woken next to their waker), and a CCD's path to the I/O
// die carries only so many write-combined stores: the same 229 KB item took 10-13 us on a thread alone on its CCD and 45-60 us
// on that one (round 4, IPCR_DEBUG_TIMES).  -> the CPU sets (a core's hardware threads) in dealing order; empty when the
// topology cannot be read.
const std::vector<cpu_set_t> &spread_core_sets(int phys) {
    static std::mutex mu;
    static std::map<int, std::vector<cpu_set_t>> cache;
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(phys);
    if (it != cache.end()) return it->second;
    std::vector<cpu_set_t> &out = cache[phys];
    const CpuSet &dc = device_cpus(phys);
    const cpu_set_t &allowed = dc.known ? dc.set : g_initial_cpus.set;
    if (!dc.known && !g_initial_cpus.known) return out;
    auto read_int = [](const char *fmt, int cpu, long &v) {
        char path[160];
        snprintf(path, sizeof path, fmt, cpu);
        FILE *fh = fopen(path, "r");
        if (!fh) return false;
        const bool ok = fscanf(fh, "%ld", &v) == 1;
        fclose(fh);
        return ok;
    };
    std::map<long, std::map<long, cpu_set_t>> l3; // L3 domain -> (package, core) -> its hardware threads
    for (int cpu = 0; cpu < CPU_SETSIZE; ++cpu) {
        if (!CPU_ISSET(cpu, &allowed)) continue;
        long core = 0, pkg = 0, dom = 0;
        if (!read_int("/sys/devices/system/cpu/cpu%d/topology/core_id", cpu, core)) { out.clear(); return out; }
        (void)read_int("/sys/devices/system/cpu/cpu%d/topology/physical_package_id", cpu, pkg);
        if (!read_int("/sys/devices/system/cpu/cpu%d/cache/index3/id", cpu, dom)) dom = pkg;
        auto &cs = l3[dom];
        auto f = cs.find(pkg * 100000 + core);
        if (f == cs.end()) { cpu_set_t z; CPU_ZERO(&z); f = cs.emplace(pkg * 100000 + core, z).first; }
        CPU_SET(cpu, &f->second);
    }
    for (bool any = true; any;) {
        any = false;
        for (auto &d : l3)
            if (!d.second.empty()) {
                out.push_back(d.second.begin()->second);
                d.second.erase(d.second.begin());
                any = true;
            }
    }
    return out;
}
} // namespace

// pool thread `index` onto its core; false: the topology is unknown (the caller falls back to the device's whole set)
bool bind_pool_thread(int phys, unsigned index) {
    static const bool on = !(getenv("IPCR_POOL_SPREAD") && atoi(getenv("IPCR_POOL_SPREAD")) == 0) &&
                           !(getenv("IPCR_BIND_THREADS") && *getenv("IPCR_BIND_THREADS") && atoi(getenv("IPCR_BIND_THREADS")) == 0);
    if (!on) return false;
    const std::vector<cpu_set_t> &cores = spread_core_sets(phys);
    if (cores.empty()) return false;
    // (from the far end of the list: the caller's own thread and the runtime's helpers tend to sit on the first CPUs)
    const cpu_set_t &c = cores[cores.size() - 1 - index % cores.size()];
    return sched_setaffinity(0, sizeof c, &c) == 0;
}


} // namespace ipcr

using ipcr::PackPool;

// tests/test_host_logic.py: runs of every size in quick succession, from two callers at once, with and without an idle
// callback; every item of every run must have been called exactly once when its run returns.  -> the number of violations
extern "C" int32_t ipcr_internal_pool_selftest(uint32_t rounds, uint32_t max_items) {
    std::atomic<int32_t> bad{0};
    auto caller = [&](uint32_t seed) {
        uint32_t x = seed;
        for (uint32_t r = 0; r < rounds; ++r) {
            x = x * 1664525u + 1013904223u;
            const size_t n = 1u + (x >> 8) % std::max(1u, max_items);
            std::unique_ptr<std::atomic<uint32_t>[]> calls(new std::atomic<uint32_t>[n]);
            for (size_t i = 0; i < n; ++i) calls[i].store(0);
            std::atomic<uint64_t> idle_calls{0};
            const std::function<void()> idle = [&] { idle_calls.fetch_add(1, std::memory_order_relaxed); };
            const bool with_idle = (x >> 4) & 1u;
            PackPool::get().run(n, [&](size_t i) {
                volatile uint32_t sink = 0;
                for (uint32_t k = 0; k < ((uint32_t)i * 2654435761u >> 24); ++k) sink = sink + k; // items of uneven length
                calls[i].fetch_add(1, std::memory_order_relaxed);
            }, -1, with_idle ? &idle : nullptr);
            for (size_t i = 0; i < n; ++i)
                if (calls[i].load() != 1u) bad.fetch_add(1);
        }
    };
    std::thread other(caller, 0x1234567u);
    caller(0x7654321u);
    other.join();
    return bad.load();
}

