// slab_pipeline.hip -- what slows the host-to-device copies of the FASTA loader's slab pipeline?  (development tool)
//   slab_pipeline <file> <variant>...      variants: bits  1 = consumer thread (kernel over the slab + 4-byte D2H + sync per slab)
//                                                          2 = fill by pread (else memcpy out of a populated mapping)
//                                                          4 = persistent fill threads (else started per slab)
//                                                          8 = consumer's kernel waits on the copy's event on its own stream (else the consumer waits on the host)
//                                                         16 = the fill lands 1000 bytes into the pinned slab (the loader's carried line), odd copy length
//                                                         32 = pieces of 2 MiB handed out by a counter (else one sixteenth of the slab per thread)
//                                                         64 = consumer: a second kernel that writes 64 MiB, a 64 MiB device-to-device copy, a second sync
//                                                        128 = consumer: hipMalloc + hipFree of 128 MiB per slab
//                                                        256 = the loader's offsets: every slab carries 61 bytes over, so neither the file offsets nor the targets of the reads are page-aligned
// prints the total and the duration of every copy (events on the copy stream).
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void scribble(uint4 *p, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4((unsigned)i, 1, 2, 3);
}

__global__ void touch(const uint4 *p, size_t n16, unsigned *out) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) acc += p[i].x ^ p[i].w;
    if (acc == 0x12345678u) atomicAdd(out, 1u);
}

struct Pool { // persistent fill threads: run(f) calls f(t) on every thread and returns when all are done
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<void(unsigned)> job;
    uint64_t gen = 0;
    unsigned left = 0;
    bool stop = false;
    explicit Pool(unsigned n) {
        for (unsigned t = 0; t < n; ++t) th.emplace_back([this, t] {
            uint64_t seen = 0;
            for (;;) {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || gen != seen; });
                if (stop) return;
                seen = gen;
                auto j = job;
                lk.unlock();
                j(t);
                lk.lock();
                if (--left == 0) cv.notify_all();
            }
        });
    }
    void run(std::function<void(unsigned)> f) {
        std::unique_lock<std::mutex> lk(mu);
        job = std::move(f);
        left = (unsigned)th.size();
        ++gen;
        cv.notify_all();
        cv.wait(lk, [&] { return left == 0; });
    }
    ~Pool() { { std::lock_guard<std::mutex> lk(mu); stop = true; } cv.notify_all(); for (auto &t : th) t.join(); }
};

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    const int fd = open(argv[1], O_RDONLY);
    struct stat sb;
    if (fd < 0 || fstat(fd, &sb)) return 2;
    const size_t n = (size_t)sb.st_size, slab = (size_t)64 << 20;
    const unsigned nt = 16;
    const uint8_t *map = (const uint8_t *)mmap(nullptr, n, PROT_READ, MAP_SHARED | MAP_POPULATE, fd, 0);
    uint8_t *pin[4], *draw[2];
    unsigned *d_out, *h_out;
    for (auto &p : pin) { if (hipHostMalloc((void **)&p, slab, hipHostMallocDefault) != hipSuccess) return 3; memset(p, 1, slab); }
    for (auto &p : draw) if (hipMalloc((void **)&p, slab) != hipSuccess) return 3;
    uint8_t *d_a, *d_b;
    (void)hipMalloc((void **)&d_a, slab);
    (void)hipMalloc((void **)&d_b, slab);
    (void)hipMalloc((void **)&d_out, 4);
    (void)hipHostMalloc((void **)&h_out, 4, hipHostMallocDefault);
    hipStream_t cs, st;
    (void)hipStreamCreateWithFlags(&cs, hipStreamNonBlocking);
    (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    Pool pool(nt);
    for (int a = 2; a < argc; ++a) {
        const int v = atoi(argv[a]);
        for (int rep = 0; rep < 2; ++rep) {
            const size_t nsl = (n + slab - 1) / slab;
            std::vector<hipEvent_t> t0(nsl), t1(nsl), done(nsl), freed(nsl);
            for (size_t j = 0; j < nsl; ++j) { (void)hipEventCreate(&t0[j]); (void)hipEventCreate(&t1[j]); (void)hipEventCreateWithFlags(&done[j], hipEventDisableTiming); (void)hipEventCreateWithFlags(&freed[j], hipEventDisableTiming); }
            std::mutex mu;
            std::condition_variable cv;
            size_t queued = 0, consumed = 0, freed_n = 0;
            const double w0 = now();
            std::thread consumer;
            if (v & 1) consumer = std::thread([&] {
                for (size_t j = 0; j < nsl; ++j) {
                    { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return queued > j; }); }
                    const size_t len = std::min(slab, n - j * slab);
                    if (v & 8) (void)hipStreamWaitEvent(st, done[j], 0); else (void)hipEventSynchronize(done[j]);
                    touch<<<1024, 256, 0, st>>>((const uint4 *)draw[j & 1], len / 16, d_out);
                    (void)hipEventRecord(freed[j], st);
                    { std::lock_guard<std::mutex> lk(mu); freed_n = j + 1; }
                    cv.notify_all();
                    (void)hipMemcpyAsync(h_out, d_out, 4, hipMemcpyDeviceToHost, st);
                    (void)hipStreamSynchronize(st);
                    if (v & 64) {
                        scribble<<<1024, 256, 0, st>>>((uint4 *)d_a, slab / 16);
                        (void)hipMemcpyAsync(d_b, d_a, slab, hipMemcpyDeviceToDevice, st);
                        (void)hipMemcpyAsync(h_out, d_out, 4, hipMemcpyDeviceToHost, st);
                        (void)hipStreamSynchronize(st);
                    }
                    if (v & 128) { void *tmp = nullptr; (void)hipMalloc(&tmp, (size_t)128 << 20); (void)hipFree(tmp); }
                    { std::lock_guard<std::mutex> lk(mu); consumed = j + 1; }
                    cv.notify_all();
                }
            });
            for (size_t j = 0; j < nsl; ++j) {
                const size_t off = j * slab, len = std::min(slab, n - off);
                uint8_t *b = pin[j % 4];
                if (j >= 4) {
                    if (v & 1) { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return consumed + 4 > j; }); }
                    (void)hipEventSynchronize(done[j - 4]);
                }
                const size_t per = ((len + nt - 1) / nt + 4095) & ~(size_t)4095;
                const size_t skew = (v & 256) ? 61 : (v & 16) ? 1000 : 0, flen = len - skew;
                const size_t foff = (v & 256) ? j * (slab - 61) : off;
                std::atomic<size_t> nextp{0};
                auto span = [&](size_t a0, size_t e) {
                    if (v & 2) { while (a0 < e) { ssize_t r = pread(fd, b + skew + a0, e - a0, (off_t)(foff + a0)); if (r <= 0) return; a0 += (size_t)r; } }
                    else if (a0 < e) memcpy(b + skew + a0, map + foff + a0, e - a0);
                };
                auto piece = [&](unsigned t) {
                    if (v & 32) { for (;;) { const size_t k = nextp.fetch_add(1), a0 = k << 21; if (a0 >= flen) return; span(a0, std::min(flen, a0 + ((size_t)2 << 20))); } }
                    else span(std::min(flen, (size_t)t * per), std::min(flen, std::min(flen, (size_t)t * per) + per));
                };
                if (v & 4) pool.run(piece);
                else { std::vector<std::thread> th; for (unsigned t = 0; t < nt; ++t) th.emplace_back(piece, t); for (auto &t : th) t.join(); }
                if ((v & 1) && j >= 2) {
                    { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return freed_n + 2 > j; }); }
                    (void)hipStreamWaitEvent(cs, freed[j - 2], 0);
                }
                (void)hipEventRecord(t0[j], cs);
                (void)hipMemcpyAsync(draw[j & 1], b, (v & 16) ? len - 13 : len, hipMemcpyHostToDevice, cs);
                (void)hipEventRecord(t1[j], cs);
                (void)hipEventRecord(done[j], cs);
                { std::lock_guard<std::mutex> lk(mu); queued = j + 1; }
                cv.notify_all();
            }
            if (consumer.joinable()) consumer.join();
            (void)hipStreamSynchronize(cs);
            const double w = now() - w0;
            printf("variant %2d: %.2f ms = %.1f GB/s; copies (ms):", v, w * 1e3, n / w / 1e9);
            for (size_t j = 0; j < nsl; ++j) { float ms = 0; (void)hipEventElapsedTime(&ms, t0[j], t1[j]); printf(" %.2f", ms); }
            printf("\n");
            for (size_t j = 0; j < nsl; ++j) { (void)hipEventDestroy(t0[j]); (void)hipEventDestroy(t1[j]); (void)hipEventDestroy(done[j]); (void)hipEventDestroy(freed[j]); }
        }
    }
    return 0;
}
