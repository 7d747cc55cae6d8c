// numa_probe.hip -- where does pinned memory live, and what does the socket of the core that wrote it cost the copy?
// (development tool)   numa_probe
// For the allocating thread on node 0 / node 1: hipHostMalloc 64 MiB, ask the kernel which node its pages are on
// (move_pages), then for a writer thread on node 0 / node 1 and ordinary / non-temporal stores: fill the buffer, copy it
// to the device, print the copy's duration.
#include <hip/hip_runtime.h>
#include <immintrin.h>
#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>

static bool parse_list(const char *line, cpu_set_t *set) {
    CPU_ZERO(set);
    for (const char *q = line; *q;) {
        char *e = nullptr;
        const long a = strtol(q, &e, 10);
        if (e == q) break;
        long b = a;
        if (*e == '-') { q = e + 1; b = strtol(q, &e, 10); }
        for (long k = a; k <= b && k < CPU_SETSIZE; ++k) CPU_SET((int)k, set);
        if (*e != ',') break;
        q = e + 1;
    }
    return CPU_COUNT(set) > 0;
}

static bool node_cpus(int node, cpu_set_t *set) {
    char path[128], line[4096] = {0};
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    FILE *f = fopen(path, "r");
    if (!f) return false;
    const bool ok = fgets(line, sizeof line, f) != nullptr;
    fclose(f);
    return ok && parse_list(line, set);
}

static int page_node(void *p) {
    int status = -1;
    void *pages[1] = {p};
    if (syscall(SYS_move_pages, 0, 1ul, pages, nullptr, &status, 0) != 0) return -2;
    return status;
}

int main() {
    char bus[64] = {0};
    (void)hipDeviceGetPCIBusId(bus, sizeof bus, 0);
    for (char *q = bus; *q; ++q) *q = (char)tolower((unsigned char)*q);
    {
        const std::string p = std::string("/sys/bus/pci/devices/") + bus + "/local_cpulist", p2 = std::string("/sys/bus/pci/devices/") + bus + "/numa_node";
        char line[4096] = {0}, l2[64] = {0};
        FILE *f = fopen(p.c_str(), "r");
        if (f) { (void)!fgets(line, sizeof line, f); fclose(f); }
        f = fopen(p2.c_str(), "r");
        if (f) { (void)!fgets(l2, sizeof l2, f); fclose(f); }
        printf("device 0 = %s, numa_node %s  local_cpulist %s", bus, l2, line);
    }
    const size_t n = (size_t)64 << 20;
    uint8_t *dev = nullptr;
    if (hipMalloc((void **)&dev, n) != hipSuccess) return 3;
    hipStream_t st;
    (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    cpu_set_t all;
    sched_getaffinity(0, sizeof all, &all);
    for (int an = 0; an < 2; ++an) {
        cpu_set_t cs;
        if (!node_cpus(an, &cs)) continue;
        sched_setaffinity(0, sizeof cs, &cs);
        uint8_t *pin = nullptr;
        if (hipHostMalloc((void **)&pin, n, hipHostMallocDefault) != hipSuccess) return 3;
        uint8_t *pin_user = nullptr;
        const bool have_user = hipHostMalloc((void **)&pin_user, n, hipHostMallocNumaUser) == hipSuccess;
        printf("allocating thread on node %d: pages of hipHostMallocDefault on node %d / %d (first, last)", an, page_node(pin), page_node(pin + n - 4096));
        if (have_user) printf("; hipHostMallocNumaUser on node %d", page_node(pin_user));
        printf("\n");
        sched_setaffinity(0, sizeof all, &all);
        for (int wn = 0; wn < 2; ++wn) {
            for (int nt = 0; nt < 2; ++nt) {
                float best = 1e9f;
                for (int rep = 0; rep < 3; ++rep) {
                    std::thread w([&] {
                        cpu_set_t ws;
                        if (node_cpus(wn, &ws)) sched_setaffinity(0, sizeof ws, &ws);
                        const __m256i v = _mm256_set1_epi8((char)(rep + 1));
                        for (size_t i = 0; i < n; i += 32) {
                            if (nt) _mm256_stream_si256((__m256i *)(pin + i), v); else _mm256_store_si256((__m256i *)(pin + i), v);
                        }
                        _mm_sfence();
                    });
                    w.join();
                    (void)hipEventRecord(e0, st);
                    (void)hipMemcpyAsync(dev, pin, n, hipMemcpyHostToDevice, st);
                    (void)hipEventRecord(e1, st);
                    (void)hipStreamSynchronize(st);
                    float ms = 0;
                    (void)hipEventElapsedTime(&ms, e0, e1);
                    if (ms < best) best = ms;
                }
                printf("  writer on node %d, %s stores: copy of 64 MiB %.2f ms = %.1f GB/s\n", wn, nt ? "non-temporal" : "ordinary    ", best, n / (best * 1e-3) / 1e9);
            }
        }
        (void)hipHostFree(pin);
        if (have_user) (void)hipHostFree(pin_user);
    }
    return 0;
}
