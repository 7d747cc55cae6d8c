// pread_rate.cpp -- how fast do file bytes get from the page cache into pinned memory on this host, and what does a
// concurrent host-to-device copy cost them?  (development tool for the FASTA loader; build: tools/ubench/build.sh)
//   pread_rate <file> [threads ...]
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <class F> static double par(unsigned nt, size_t n, F f) { // f(a, b) over [0, n) split into nt pieces
    const double t0 = now();
    std::vector<std::thread> th;
    const size_t per = ((n + nt - 1) / nt + 4095) & ~(size_t)4095;
    for (unsigned t = 0; t < nt; ++t) {
        const size_t a = std::min(n, (size_t)t * per), b = std::min(n, a + per);
        if (a < b) th.emplace_back(f, a, b);
    }
    for (auto &t : th) t.join();
    return now() - t0;
}

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    const int fd = open(argv[1], O_RDONLY);
    struct stat sb;
    if (fd < 0 || fstat(fd, &sb)) return 2;
    const size_t n = std::min<size_t>((size_t)sb.st_size, (size_t)1 << 30);
    uint8_t *pin = nullptr, *dev = nullptr, *pin2 = nullptr;
    uint8_t *mal = (uint8_t *)aligned_alloc(4096, n);
    memset(mal, 1, n);
    if (hipHostMalloc((void **)&pin, n, hipHostMallocDefault) != hipSuccess) return 3;
    if (hipHostMalloc((void **)&pin2, n, hipHostMallocDefault) != hipSuccess) return 3;
    if (hipMalloc((void **)&dev, n) != hipSuccess) return 3;
    memset(pin, 1, n);
    memset(pin2, 2, n);
    const uint8_t *map = (const uint8_t *)mmap(nullptr, n, PROT_READ, MAP_SHARED | MAP_POPULATE, fd, 0);
    hipStream_t st;
    (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    std::vector<unsigned> nts;
    for (int i = 2; i < argc; ++i) nts.push_back((unsigned)atoi(argv[i]));
    if (nts.empty()) nts = {1, 8, 16, 32};
    auto rd = [&](uint8_t *dst) { return [=](size_t a, size_t b) { while (a < b) { ssize_t r = pread(fd, dst + a, std::min<size_t>(b - a, 4u << 20), (off_t)a); if (r <= 0) return; a += (size_t)r; } }; };
    auto cp = [&](uint8_t *dst) { return [=](size_t a, size_t b) { memcpy(dst + a, map + a, b - a); }; };
    for (int rep = 0; rep < 2; ++rep) {
        const double t0 = now();
        (void)hipMemcpyAsync(dev, pin2, n, hipMemcpyHostToDevice, st);
        (void)hipStreamSynchronize(st);
        printf("h2d alone %.1f GB/s\n", n / (now() - t0) / 1e9);
    }
    for (unsigned nt : nts) {
        par(nt, n, rd(mal));
        const double a = par(nt, n, rd(mal)), b = par(nt, n, rd(pin)), c = map != MAP_FAILED ? par(nt, n, cp(pin)) : 0;
        (void)hipMemcpyAsync(dev, pin2, n, hipMemcpyHostToDevice, st);
        const double t0 = now();
        const double d = par(nt, n, rd(pin));
        (void)hipStreamSynchronize(st);
        const double e = now() - t0;
        printf("%2u threads: pread->malloc %.1f  pread->pinned %.1f  mmap memcpy->pinned %.1f GB/s; with a concurrent h2d: pread->pinned %.1f, both done at %.1f GB/s each\n",
               nt, n / a / 1e9, n / b / 1e9, c ? n / c / 1e9 : 0.0, n / d / 1e9, n / e / 1e9);
    }
    // the loader's pipeline: 64 MiB slabs through four pinned buffers, the copy of slab j queued behind its fill
    {
        const double t0 = now();
        void *m2 = mmap(nullptr, n, PROT_READ, MAP_SHARED | MAP_POPULATE, fd, 0);
        const double t1 = now();
        munmap(m2, n);
        printf("mmap + MAP_POPULATE of the file: %.2f ms, munmap %.2f ms\n", (t1 - t0) * 1e3, (now() - t1) * 1e3);
    }
    for (int mode = 0; mode < 4; ++mode) { // 0: pread, threads started per slab; 1: memcpy out of the mapping; 2: one pread thread per slab, 4 slabs in flight
        for (unsigned nt : {8u, 16u}) {
            const size_t slab = (size_t)64 << 20;
            hipEvent_t ev[4];
            for (auto &e : ev) (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
            const double t0 = now();
            uint64_t j = 0;
            if (mode == 2) continue;
            const uint8_t *map0 = map;
            if (mode == 3) { // a fresh mapping, nothing populated: the copy threads take the page faults
                map = (const uint8_t *)mmap(nullptr, n, PROT_READ, MAP_SHARED, fd, 0);
                madvise((void *)map, n, MADV_SEQUENTIAL);
            }
            if (mode != 2) {
                for (size_t off = 0; off < n; off += slab, ++j) {
                    uint8_t *b = pin + (j % 4) * slab;
                    const size_t len = std::min(slab, n - off);
                    if (j >= 4) (void)hipEventSynchronize(ev[j % 4]);
                    if (mode == 0) par(nt, len, [=](size_t a, size_t e) { while (a < e) { ssize_t r = pread(fd, b + a, e - a, (off_t)(off + a)); if (r <= 0) return; a += (size_t)r; } });
                    else par(nt, len, [=](size_t a, size_t e) { memcpy(b + a, map + off + a, e - a); });
                    (void)hipMemcpyAsync(dev + off, b, len, hipMemcpyHostToDevice, st);
                    (void)hipEventRecord(ev[j % 4], st);
                }
            } else {
                // nt threads, each owns every nt-th piece of 4 MiB of every slab; pieces are copied as they land
                const size_t piece = (size_t)4 << 20;
                std::atomic<uint64_t> next{0};
                std::vector<std::thread> th;
                std::vector<hipStream_t> sts(nt);
                for (auto &x : sts) (void)hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
                uint8_t *ring = pin; // n bytes of pinned memory: no reuse in this mode (an upper bound of the scheme)
                for (unsigned t = 0; t < nt; ++t) th.emplace_back([&, t] {
                    for (;;) {
                        const uint64_t k = next.fetch_add(1);
                        const size_t off = (size_t)k * piece;
                        if (off >= n) break;
                        const size_t len = std::min(piece, n - off);
                        size_t a = 0;
                        while (a < len) { ssize_t r = pread(fd, ring + off + a, len - a, (off_t)(off + a)); if (r <= 0) break; a += (size_t)r; }
                        (void)hipMemcpyAsync(dev + off, ring + off, len, hipMemcpyHostToDevice, sts[t]);
                    }
                    (void)hipStreamSynchronize(sts[t]);
                });
                for (auto &t : th) t.join();
            }
            (void)hipStreamSynchronize(st);
            if (mode == 3) { munmap((void *)map, n); map = map0; }
            printf("pipeline mode %d, %2u threads: %.1f GB/s (%.2f ms)\n", mode, nt, n / (now() - t0) / 1e9, (now() - t0) * 1e3);
        }
    }
    return 0;
}
