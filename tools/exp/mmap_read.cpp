// Experiment: how fast can 16 threads get a 1 GB file's bytes out of the page cache -- pread into thread-local buffers
// (what the FASTA loader does, into pinned slabs) against loads from a mapping of the file.
// build: g++ -O2 -mavx2 -pthread -o /tmp/mmap_read tools/exp/mmap_read.cpp
#include <fcntl.h>
#include <immintrin.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <atomic>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static uint64_t sum_bytes(const uint8_t *p, size_t n) {
    __m256i acc = _mm256_setzero_si256();
    for (size_t i = 0; i + 32 <= n; i += 32) acc = _mm256_add_epi64(acc, _mm256_sad_epu8(_mm256_loadu_si256((const __m256i *)(p + i)), _mm256_setzero_si256()));
    uint64_t v[4]; _mm256_storeu_si256((__m256i *)v, acc);
    return v[0] + v[1] + v[2] + v[3];
}
int main(int argc, char **argv) {
    const char *path = argc > 1 ? argv[1] : "/tmp/mmap_read.dat";
    const size_t bytes = 1ull << 30;
    {
        int fd = open(path, O_CREAT | O_TRUNC | O_WRONLY, 0600);
        std::vector<uint8_t> buf(1 << 22, 'A');
        for (size_t i = 0; i < buf.size(); i += 81) buf[i] = '\n';
        for (size_t off = 0; off < bytes; off += buf.size()) if (write(fd, buf.data(), buf.size()) < 0) return 1;
        close(fd);
    }
    int fd = open(path, O_RDONLY);
    for (int T : {1, 8, 16, 32}) {
        for (int rep = 0; rep < 2; ++rep) {
            std::atomic<uint64_t> total{0};
            std::atomic<size_t> next{0};
            const size_t piece = 2u << 20, np = bytes / piece;
            double t0 = now();
            {
                std::vector<std::thread> th;
                for (int t = 0; t < T; ++t) th.emplace_back([&] {
                    std::vector<uint8_t> buf(piece);
                    uint64_t s = 0;
                    for (;;) { size_t i = next.fetch_add(1); if (i >= np) break; if (pread(fd, buf.data(), piece, (off_t)(i * piece)) != (ssize_t)piece) break; s += sum_bytes(buf.data(), piece); }
                    total += s;
                });
                for (auto &x : th) x.join();
            }
            const double tp = now() - t0;
            next = 0;
            t0 = now();
            void *m = mmap(nullptr, bytes, PROT_READ, MAP_SHARED, fd, 0);
            uint64_t tm = 0;
            {
                std::vector<std::thread> th;
                std::atomic<uint64_t> tot2{0};
                for (int t = 0; t < T; ++t) th.emplace_back([&] {
                    uint64_t s = 0;
                    for (;;) { size_t i = next.fetch_add(1); if (i >= np) break; s += sum_bytes((const uint8_t *)m + i * piece, piece); }
                    tot2 += s;
                });
                for (auto &x : th) x.join();
                tm = tot2;
            }
            const double tmm = now() - t0;
            munmap(m, bytes);
            printf("%2d threads: pread %.1f ms (%.1f GB/s)   mmap+loads %.1f ms (%.1f GB/s)   %s\n", T, tp * 1e3, bytes / tp / 1e9, tmm * 1e3, bytes / tmm / 1e9, total.load() == tm ? "" : "SUM MISMATCH");
        }
    }
    close(fd);
    unlink(path);
    return 0;
}
