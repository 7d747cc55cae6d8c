// Experiment: can the CPU write straight into HBM (fine-grained device memory through the PCIe BAR), and how fast?
// build on the GPU box: hipcc -O2 -mavx512f -mavx512bw -o /tmp/bar_write tools/exp/bar_write.cpp -lpthread
#include <hip/hip_runtime.h>
#include <immintrin.h>
#include <atomic>
#include <chrono>
#include <csetjmp>
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static sigjmp_buf jb;
static void on_segv(int) { siglongjmp(jb, 1); }

__global__ void sum_kernel(const uint32_t *p, size_t n, unsigned long long *out) {
    unsigned long long s = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i];
    atomicAdd(out, s);
}

static void nt_fill(uint8_t *dst, size_t bytes, uint32_t v) {
    const __m512i x = _mm512_set1_epi32((int)v);
    for (size_t i = 0; i + 64 <= bytes; i += 64) _mm512_stream_si512((__m512i *)(dst + i), x);
    _mm_sfence();
}

int main() {
    const size_t bytes = 256u << 20;
    struct { const char *name; unsigned flag; } kinds[] = {{"hipDeviceMallocFinegrained", hipDeviceMallocFinegrained},
                                                            {"hipDeviceMallocUncached", hipDeviceMallocUncached},
                                                            {"hipDeviceMallocDefault", hipDeviceMallocDefault}};
    unsigned long long *d_out = nullptr;
    hipMalloc((void **)&d_out, 8);
    for (auto &k : kinds) {
        void *p = nullptr;
        hipError_t e = hipExtMallocWithFlags(&p, bytes, k.flag);
        printf("%s: alloc %s\n", k.name, hipGetErrorString(e));
        if (e != hipSuccess) continue;
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, p) == hipSuccess) printf("  type %d device %d host %p dev %p managed %d\n", (int)at.type, at.device, at.hostPointer, at.devicePointer, at.isManaged);
        signal(SIGSEGV, on_segv);
        signal(SIGBUS, on_segv);
        if (sigsetjmp(jb, 1)) { printf("  CPU store faulted: not host-accessible\n"); hipFree(p); continue; }
        ((volatile uint32_t *)p)[0] = 7u; // first touch
        printf("  CPU store ok\n");
        for (int threads : {1, 4, 8, 16}) {
            double best = 0;
            for (int rep = 0; rep < 3; ++rep) {
                std::vector<std::thread> th;
                const double t0 = now();
                for (int t = 0; t < threads; ++t)
                    th.emplace_back([&, t] { nt_fill((uint8_t *)p + (bytes / threads) * t, bytes / threads, 3u); });
                for (auto &x : th) x.join();
                best = std::max(best, bytes / (now() - t0) / 1e9);
            }
            printf("  %2d threads NT stores: %.1f GB/s\n", threads, best);
        }
        // stale lines? the device reads (its L2 may keep the lines), the CPU overwrites, the device reads again
        for (uint32_t v : {5u, 9u, 3u}) {
            nt_fill((uint8_t *)p, bytes, v);
            hipMemset(d_out, 0, 8);
            sum_kernel<<<1024, 256>>>((const uint32_t *)p, bytes / 4, d_out);
            unsigned long long g2 = 0;
            hipMemcpy(&g2, d_out, 8, hipMemcpyDeviceToHost);
            printf("  after CPU fill with %u the device sums %llu (want %llu)%s\n", v, g2, (unsigned long long)v * (bytes / 4), g2 == (unsigned long long)v * (bytes / 4) ? "" : "  STALE");
        }
        // as the packer writes: every thread two streams of 8-byte NT stores (lo and hi planes of its columns)
        for (int threads : {1, 8, 16}) {
            std::vector<std::thread> th;
            const double t0 = now();
            for (int t = 0; t < threads; ++t)
                th.emplace_back([&, t] {
                    long long *a = (long long *)((uint8_t *)p + (bytes / threads) * t), *b = a + bytes / threads / 16;
                    for (size_t i = 0; i < bytes / threads / 16; ++i) { _mm_stream_si64(a + i, 3); _mm_stream_si64(b + i, 3); }
                    _mm_sfence();
                });
            for (auto &x : th) x.join();
            printf("  %2d threads, two 8-byte NT streams each: %.1f GB/s\n", threads, bytes / (now() - t0) / 1e9);
        }
        hipMemset(d_out, 0, 8);
        sum_kernel<<<1024, 256>>>((const uint32_t *)p, bytes / 4, d_out);
        unsigned long long got = 0;
        hipMemcpy(&got, d_out, 8, hipMemcpyDeviceToHost);
        printf("  device sees sum %llu (want %llu)\n", got, 3ull * (bytes / 4));
        hipFree(p);
    }
    // reference: pinned host memory written by the CPU, then DMA
    void *h = nullptr, *d = nullptr;
    hipHostMalloc(&h, bytes, hipHostMallocDefault);
    hipMalloc(&d, bytes);
    for (int threads : {1, 16}) {
        std::vector<std::thread> th;
        const double t0 = now();
        for (int t = 0; t < threads; ++t) th.emplace_back([&, t] { nt_fill((uint8_t *)h + (bytes / threads) * t, bytes / threads, 3u); });
        for (auto &x : th) x.join();
        printf("pinned host, %2d threads NT stores: %.1f GB/s\n", threads, bytes / (now() - t0) / 1e9);
    }
    hipDeviceSynchronize();
    const double t0 = now();
    hipMemcpy(d, h, bytes, hipMemcpyHostToDevice);
    printf("DMA pinned -> device: %.1f GB/s\n", bytes / (now() - t0) / 1e9);
    return 0;
}
