// chunk_workers -- the drop-in entry point under the reference's worker model, driven from native threads.
//
//   chunk_workers [--devices d0,d1,...] [--bind] [--probe] [record_bases] [chunk_bases] [workers ...]   (defaults: device 0, 125000000 4000000 1 8 16)
//
// --probe: ipcr-probe (BASELINE C5) as the Go pipeline would run it over the drop-in call -- every planted amplicon carries
// an internal probe, every worker annotates the products of the chunk it has just scanned with ipcr_probe_scratch_products
// (--probe-max-mm 2) and checks the planted ones; beside the pool one "collector" thread calls ipcr_probe_best_hit per
// amplicon the whole time (internal/visitors/probe.go:18-33: the form a host without the batched call uses) and its latency
// under the sweeping workers is reported next to the latency on an idle device.
//
// --devices: worker i creates its scratch on device d[i mod n] (ipcr_scratch_create_on) -- one host process, every GPU of
// the node, no collective: chunks are independent.  The worker threads never select a device themselves; every entry
// point of the library does.  A device may be listed twice, and with IPCR_DEVICE_SLOTS=N in the environment devices
// beyond the physical ones exist as slots with tables of their own (a one-GPU box rehearses the N-device flow).
// --bind: every worker thread first moves onto the CPUs next to its device (ipcr_bind_thread_to_device).  Not the default:
// a pool of 16 workers is not limited by the link, and on the shared hosts these numbers come from 16 threads held on
// one socket lost more to its other tenants (63 Gbases/s) than the scheduler's free choice of both sockets (96).
//
// internal/pipeline/pipeline.go:60-125: CompilePanel once, one scratch per worker, every worker pulls rolling chunks
// (core/fasta/path_ctx.go:83-179: chunk size, overlap = max product length) of a record from one queue and calls
// ForEachCompiledProduct = ipcr_scan_chunk on host ASCII.  Prints one JSON line: the pinned H2D rate of the link and
// the aggregate PCIe-inclusive Gbases/s per worker count.  Workload C2 (benchPrimer pair 0 + self pairs, k=2,
// 3'-window 5) over a benchDNA record with planted amplicons; the product count of every pass is checked.
// bench.py runs this binary for config.other_workloads.scan_chunk: a Python thread pool adds its own per-call
// interpreter work to what is a 0.1 ms call.
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "ipcr_hip.h"

static std::string bench_primer(unsigned idx, int n = 20) { // core/engine/performance_benchmark_test.go:78-93
    unsigned x = 0x9e3779b9u ^ (idx * 0x45d9f3bu);
    std::string s((size_t)n, 'A');
    for (int i = 0; i < n; ++i) {
        x = x * 1103515245u + 12345u + (unsigned)(i * 97);
        s[(size_t)i] = "ACGT"[(x >> 29) & 3u];
    }
    s[0] = "ACGT"[idx & 3u];
    s[1] = "ACGT"[(idx + 1) & 3u];
    s[2] = "ACGT"[(idx + 2) & 3u];
    s[(size_t)n - 1] = "ACGT"[(idx + 3) & 3u];
    return s;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

extern "C" int32_t ipcr_internal_device_bar(int32_t slot); // host.cpp: how the packer's planes reach the device

static int usage(const char *why) {
    fprintf(stderr, "chunk_workers: %s\nusage: chunk_workers [--devices d0,d1,...] [--bind] [--probe] [--with-n] [record_bases] [chunk_bases > 2000] [workers >= 1 ...]\n", why);
    return 2;
}

int main(int argc, char **argv) {
    std::vector<int> devices;
    std::vector<const char *> pos;
    bool bind = false, probe_on = false, with_n = false;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--bind")) bind = true;
        else if (!strcmp(argv[i], "--probe")) probe_on = true;
        else if (!strcmp(argv[i], "--with-n")) with_n = true;
        else if (!strcmp(argv[i], "--devices")) {
            if (i + 1 >= argc) return usage("--devices needs a list");
            for (const char *q = argv[++i]; *q;) {
                char *end = nullptr;
                const long d = strtol(q, &end, 10);
                if (end == q || d < 0) return usage("--devices: comma-separated device numbers");
                devices.push_back((int)d);
                q = *end == ',' ? end + 1 : end;
                if (*end && *end != ',') return usage("--devices: comma-separated device numbers");
            }
        } else pos.push_back(argv[i]);
    }
    const uint64_t n = pos.size() > 0 ? strtoull(pos[0], nullptr, 10) : 125000000ull;
    const uint64_t chunk = pos.size() > 1 ? strtoull(pos[1], nullptr, 10) : 4000000ull;
    std::vector<int> workers;
    for (size_t i = 2; i < pos.size(); ++i) workers.push_back(atoi(pos[i]));
    if (workers.empty()) workers = {1, 8, 16};
    const uint64_t overlap = 2000;
    if (n == 0) return usage("record_bases must be positive");
    if (chunk <= overlap) return usage("chunk_bases must exceed the overlap of 2000 (the rolling window would not advance)");
    for (int W : workers)
        if (W < 1 || W > 1024) return usage("a worker count must be in 1..1024");
    // HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues in creation order.  With the chunks packed on the host (round 3:
    // one 1.5 MB copy, one conversion launch and one sweep per call) the runtime's default of 4 is the best setting --
    // 8 / 16 / 24 workers, three runs each (tools/knobs.sh: chunk GPU_MAX_HW_QUEUES=q): 2 queues 76-79 / 72-74 / 85-87 Gbases/s; 3 queues
    // 95-102 / 93-98 / 95-97; 4 queues 94-105 / 110-116 / 106-110; 5 queues 93-112 / 103-114 / 101-107; 8 queues
    // 51-92 / 99-100 / 93-96; 12 and 16: 81-90 / 97-107.  (Round 2, ASCII over the link: 4 queues 44 / 27, 8 queues
    // 44 / 49 / 51 -- the setting this driver forced until now.)  An explicit GPU_MAX_HW_QUEUES in the environment stays.
    setenv("GPU_MAX_HW_QUEUES", "4", 0);
    if (ipcr_device_count() < 1) { fprintf(stderr, "chunk_workers: no HIP device (the scan path has no CPU fallback)\n"); return 2; }
    if (devices.empty()) devices.push_back(0);
    for (int d : devices)
        if (d >= ipcr_device_count()) { fprintf(stderr, "chunk_workers: device %d of %d\n", d, ipcr_device_count()); return 2; }

    // --bind: this thread too, before it makes the record -- the bytes the workers pack then live next to the device as well
    if (bind) (void)ipcr_bind_thread_to_device(devices[0]);
    // the record: benchDNA (performance_benchmark_test.go:67-76) + an amplicon of pair 0 every 1 Mb
    std::vector<uint8_t> seq(n);
    unsigned x = 0x5eed1234u;
    for (uint64_t i = 0; i < n; ++i) { x = x * 1664525u + 1013904223u; seq[i] = (uint8_t)"ACGT"[(x >> 30) & 3u]; }
    const std::string fwd = bench_primer(0), rev = bench_primer(1);
    std::string rc_rev(rev.size(), 'A');
    ipcr_revcomp(rev.data(), rev.size(), &rc_rev[0]);
    uint64_t planted = 0;
    static const char PROBE[] = "TGGACCTTAGCAGGTCATTCAG"; // bench.py: PROBE
    for (uint64_t a = 500000; a + 180 < n; a += 1000000, ++planted) {
        memcpy(&seq[a], fwd.data(), 20);
        memcpy(&seq[a + 160], rc_rev.data(), 20);
        if (probe_on) memcpy(&seq[a + 60], PROBE, sizeof PROBE - 1);
    }
    // --with-n: 0.1 % of the positions become 'N' in runs of 1..1000 (bench.py: n_runs; SURVEY 8d "+N") -- every chunk then
    // holds a reset byte and takes the pattern set in which the rc orientations are scanned without their window
    // (core/engine/compiled.go:185-190, 249-256).  Without it a chunk is clean, and the host's packer knows that.
    uint64_t n_positions = 0;
    if (with_n) {
        uint32_t r = 0x2545F491u;
        auto next = [&r] { r ^= r << 13; r ^= r >> 17; r ^= r << 5; return r; };
        while (n_positions < n / 1000 && n > 1002) {
            const uint64_t ln = 1 + next() % 1000, at = next() % (n - ln);
            const uint64_t rel = at % 1000000; // the planted amplicons lie at 500000 + 1 Mb * i, 180 bases each: keep clear of them
            if (rel + ln > 499000 && rel < 501000) continue;
            memset(&seq[at], 'N', ln);
            n_positions += ln;
        }
    }

    // the link: pinned host -> device, 256 MiB, best of 4
    double h2d = 0;
    {
        const size_t bytes = 256u << 20;
        void *h = nullptr, *d = nullptr;
        if (hipHostMalloc(&h, bytes, hipHostMallocDefault) != hipSuccess || hipMalloc(&d, bytes) != hipSuccess) return 3;
        memset(h, 1, bytes);
        for (int r = 0; r < 4; ++r) {
            (void)hipDeviceSynchronize();
            const double t0 = now();
            (void)hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, nullptr);
            (void)hipDeviceSynchronize();
            h2d = std::max(h2d, (double)bytes / (now() - t0) / 1e9);
        }
        (void)hipHostFree(h);
        (void)hipFree(d);
    }

    // the link as a worker pool sees it: W threads, each with its own stream and pinned buffer, copy chunk-sized pieces
    // and wait for each one (what the calls below do per chunk, without the scan)
    auto pool_h2d = [&](int W) -> double {
        std::vector<void *> h((size_t)W, nullptr), d((size_t)W, nullptr);
        std::vector<hipStream_t> st((size_t)W, nullptr);
        const size_t bytes = (size_t)std::min<uint64_t>(chunk, n);
        for (int w = 0; w < W; ++w) {
            if (hipHostMalloc(&h[(size_t)w], bytes, hipHostMallocDefault) != hipSuccess || hipMalloc(&d[(size_t)w], bytes) != hipSuccess ||
                hipStreamCreateWithFlags(&st[(size_t)w], hipStreamNonBlocking) != hipSuccess) return 0.0;
            memset(h[(size_t)w], 1, bytes);
        }
        const int per = 48;
        double best = 0;
        for (int pass = 0; pass < 3; ++pass) {
            std::vector<std::thread> th;
            const double t0 = now();
            for (int w = 0; w < W; ++w)
                th.emplace_back([&, w] {
                    for (int i = 0; i < per; ++i) {
                        (void)hipMemcpyAsync(d[(size_t)w], h[(size_t)w], bytes, hipMemcpyHostToDevice, st[(size_t)w]);
                        (void)hipStreamSynchronize(st[(size_t)w]);
                    }
                });
            for (auto &t : th) t.join();
            best = std::max(best, (double)bytes * per * W / (now() - t0) / 1e9);
        }
        for (int w = 0; w < W; ++w) { (void)hipStreamDestroy(st[(size_t)w]); (void)hipHostFree(h[(size_t)w]); (void)hipFree(d[(size_t)w]); }
        return best;
    };

    // C2 as `ipcr` scans it: pair 0 + its self pairs (internal/common/primers.go:11-37)
    ipcr_config cfg{};
    cfg.max_mm = 2; cfg.terminal_window = 5; cfg.max_len = 2000; cfg.hit_cap = 10000; cfg.seed_len = 12;
    const ipcr_pair pairs[3] = {{"bench_000", fwd.c_str(), rev.c_str(), 128, 212},
                                {"bench_000+A:self", fwd.c_str(), fwd.c_str(), 128, 212},
                                {"bench_000+B:self", rev.c_str(), rev.c_str(), 128, 212}};
    ipcr_panel *panel = nullptr;
    // CHUNK_PANEL_ROWS=n: an n-row multiplex panel instead (rows 0..n-1 of the benchmark primers + their self pairs,
    // internal/common/primers.go:41-74; k=2, window 3): what the seed-index kernel does under a worker pool
    const int rows = getenv("CHUNK_PANEL_ROWS") ? atoi(getenv("CHUNK_PANEL_ROWS")) : 0;
    std::vector<std::string> names, seqs;
    std::vector<ipcr_pair> many;
    if (rows > 0) {
        cfg.terminal_window = 3;
        for (int i = 0; i < rows; ++i) { seqs.push_back(bench_primer(2u * (unsigned)i)); seqs.push_back(bench_primer(2u * (unsigned)i + 1u)); }
        for (int i = 0; i < rows; ++i) {
            char nm[64];
            snprintf(nm, sizeof nm, "bench_%03d", i); names.push_back(nm);
            snprintf(nm, sizeof nm, "bench_%03d+A:self", i); names.push_back(nm);
            snprintf(nm, sizeof nm, "bench_%03d+B:self", i); names.push_back(nm);
        }
        for (int i = 0; i < rows; ++i) {
            const char *a = seqs[2 * (size_t)i].c_str(), *b = seqs[2 * (size_t)i + 1].c_str();
            many.push_back({names[3 * (size_t)i].c_str(), a, b, 128, 212});
            many.push_back({names[3 * (size_t)i + 1].c_str(), a, a, 128, 212});
            many.push_back({names[3 * (size_t)i + 2].c_str(), b, b, 128, 212});
        }
    }
    if (ipcr_panel_create(&cfg, rows > 0 ? many.data() : pairs, rows > 0 ? (int32_t)many.size() : 3, &panel) != IPCR_OK) { fprintf(stderr, "%s\n", ipcr_last_error()); return 4; }

    std::vector<uint64_t> starts;
    if (n > chunk) for (uint64_t s = 0; s < n; s += chunk - overlap) { starts.push_back(s); if (s + chunk >= n) break; }
    else starts.push_back(0);
    // every job owns a private copy of its bytes, as the FASTA layer hands them out (core/fasta/path_ctx.go:117)
    std::vector<std::vector<uint8_t>> jobs;
    for (uint64_t s : starts) jobs.emplace_back(seq.begin() + (long)s, seq.begin() + (long)std::min(n, s + chunk));

    printf("{\"pinned_h2d_GBps\": %.1f, \"record_bases\": %llu, \"chunk_bases\": %llu, \"chunks\": %zu, \"planted\": %llu, \"n_positions\": %llu, \"devices\": [",
           h2d, (unsigned long long)n, (unsigned long long)chunk, jobs.size(), (unsigned long long)planted, (unsigned long long)n_positions);
    for (size_t i = 0; i < devices.size(); ++i) printf("%s%d", i ? ", " : "", devices[i]);
    printf("], \"hw_queues\": %d, \"packer_path\": \"%s\"", atoi(getenv("GPU_MAX_HW_QUEUES")),
           ipcr_internal_device_bar(devices[0]) == 2 ? "PCIe BAR + HDP flush" : ipcr_internal_device_bar(devices[0]) == 1 ? "PCIe BAR" : "pinned slabs + DMA");
    int rc = 0;
    for (int W : workers) {
        std::vector<ipcr_scratch *> scs((size_t)W, nullptr);
        for (size_t w = 0; w < scs.size(); ++w) // worker w -> device w mod N
            if (ipcr_scratch_create_on(panel, devices[w % devices.size()], &scs[w]) != IPCR_OK) { fprintf(stderr, "%s\n", ipcr_last_error()); return 5; }
        for (auto &sc : scs) (void)ipcr_scan_chunk(panel, sc, jobs[0].data(), jobs[0].size(), nullptr, nullptr); // kernel build, buffers
        (void)ipcr_panel_wait_ready(panel); // (a small panel's kernels are built in the background: the timed passes run on them)
        const size_t reps = std::max<size_t>(1, ((size_t)32 * (size_t)W + jobs.size() - 1) / jobs.size()); // >= 32 chunks per worker
        const size_t total = reps * jobs.size();
        uint64_t bases = 0;
        for (const auto &j : jobs) bases += j.size();
        constexpr int PASSES = 3;
        // the pool lives as long as the run, as the reference's worker goroutines do: the threads start once
        // and meet at a barrier before every timed pass (best of three: a thread's first HIP call falls into the first)
        std::atomic<size_t> next{0};
        std::atomic<long long> nprod{0}, nannot{0}, probe_ns{0}, nset1{0}, ns[4] = {{0}, {0}, {0}, {0}};
        std::atomic<int> failed{0}, at_gate{0}, finished{0}, pass_no{-1};
        std::atomic<int> bound{0};
        auto work = [&](ipcr_scratch *sc) {
            std::vector<ipcr_probe_hit> ph;
            if (bind && ipcr_bind_thread_to_device(ipcr_scratch_device(sc))) bound.fetch_add(1);
            for (int pass = 0; pass < PASSES; ++pass) {
                at_gate.fetch_add(1);
                while (pass_no.load(std::memory_order_acquire) < pass) std::this_thread::yield();
                for (;;) {
                    const size_t j = next.fetch_add(1);
                    if (j >= total) break;
                    const auto &job = jobs[j % jobs.size()];
                    if (ipcr_scan_chunk(panel, sc, job.data(), job.size(), nullptr, nullptr) != IPCR_OK) { failed.store(1); break; }
                    const ipcr_product *pr = nullptr;
                    int64_t np = 0;
                    (void)ipcr_scratch_products(sc, &pr, &np);
                    nprod.fetch_add(np);
                    if (probe_on) { // the worker annotates its own chunk's products: the chunk's tiles are still in its scratch
                        const double tp = now();
                        ph.resize((size_t)np + 1);
                        if (ipcr_probe_scratch_products(sc, PROBE, 2, ph.data(), np) != IPCR_OK) { failed.store(1); break; }
                        probe_ns.fetch_add((long long)((now() - tp) * 1e9));
                        for (int64_t i = 0; i < np; ++i)
                            if (pr[i].pair == 0 && pr[i].type == 0 && pr[i].length == 180) {
                                const ipcr_probe_hit &h = ph[(size_t)i];
                                if (h.found && h.strand == '+' && h.pos == 60 && h.mm == 0) nannot.fetch_add(1);
                                else failed.store(2);
                            }
                    }
                    ipcr_scan_stats stt;
                    if (ipcr_scratch_stats(sc, &stt) == IPCR_OK) {
                        ns[0].fetch_add((long long)(stt.total_ms * 1e6));
                        ns[1].fetch_add((long long)(stt.enqueue_ms * 1e6));
                        ns[2].fetch_add((long long)(stt.wait_ms * 1e6));
                        ns[3].fetch_add((long long)((stt.sort_ms + stt.join_ms) * 1e6));
                        if (stt.pattern_set) nset1.fetch_add(1);
                    }
                }
                finished.fetch_add(1);
            }
        };
        std::vector<std::thread> th;
        for (int w = 0; w < W; ++w) th.emplace_back(work, scs[(size_t)w]);
        // --probe: the collector's form of the rescan, one ipcr_probe_best_hit per amplicon, the whole time the pool sweeps
        std::atomic<int> collector_stop{0};
        std::atomic<long long> bh_calls{0}, bh_ns{0}, bh_bad{0};
        std::thread collector;
        if (probe_on && W == workers.back())
            collector = std::thread([&] {
                while (pass_no.load(std::memory_order_acquire) < 0 && !collector_stop.load()) std::this_thread::yield();
                const uint8_t *amp = &seq[500000];
                while (!collector_stop.load(std::memory_order_relaxed)) {
                    ipcr_probe_hit h;
                    const double t0 = now();
                    if (ipcr_probe_best_hit(amp, 180, PROBE, 2, &h) != IPCR_OK || !h.found || h.pos != 60 || h.mm != 0 || h.strand != '+') bh_bad.fetch_add(1);
                    bh_ns.fetch_add((long long)((now() - t0) * 1e9));
                    bh_calls.fetch_add(1);
                }
            });
        double best = 0, call_ms[4] = {0, 0, 0, 0}, probe_call_ms = 0; // of the best pass: whole call, enqueue, wait, host sort + join
        long long products = -1, annotated = 0;
        for (int pass = 0; pass < PASSES; ++pass) {
            while (at_gate.load() < (pass + 1) * W) std::this_thread::yield(); // everyone is at the gate
            next.store(0); nprod.store(0); nannot.store(0); probe_ns.store(0); nset1.store(0);
            for (auto &x : ns) x.store(0);
            const double t0 = now();
            pass_no.store(pass, std::memory_order_release);
            while (finished.load() < (pass + 1) * W) std::this_thread::yield();
            const double dt = now() - t0;
            if (failed.load()) { fprintf(stderr, "scan failed: %s\n", ipcr_last_error()); rc = 6; }
            const double rate = (double)(bases * reps) / dt / 1e9;
            if (rate > best) {
                best = rate;
                for (int i = 0; i < 4; ++i) call_ms[i] = (double)ns[i].load() / 1e6 / (double)total;
                probe_call_ms = (double)probe_ns.load() / 1e6 / (double)total;
            }
            annotated = nannot.load() / (long long)reps;
            const long long per_pass = nprod.load() / (long long)reps;
            if (products >= 0 && per_pass != products) { fprintf(stderr, "product count changed between passes\n"); rc = 7; }
            products = per_pass;
        }
        for (auto &t : th) t.join();
        collector_stop.store(1);
        if (collector.joinable()) collector.join();
        if (probe_on && annotated < (long long)planted) { fprintf(stderr, "%lld annotated amplicons for %llu planted\n", annotated, (unsigned long long)planted); rc = 9; }
        if (bh_bad.load()) { fprintf(stderr, "ipcr_probe_best_hit: %lld wrong results\n", bh_bad.load()); rc = 10; }
        if (products < (long long)planted) { fprintf(stderr, "%lld products for %llu planted amplicons\n", products, (unsigned long long)planted); rc = 8; }
        printf(", \"gbases_per_s_%d_worker%s\": %.2f", W, W == 1 ? "" : "s", best);
        if (W > 1) printf(", \"pinned_h2d_GBps_%d_streams\": %.1f", W, pool_h2d(W));
        // per call: the rest of `call` is the copy into device memory (through pinned slices under a pool) and the pack enqueue
        printf(", \"call_ms_%d_worker%s\": {\"call\": %.3f, \"enqueue\": %.3f, \"wait\": %.3f, \"sort_join\": %.3f}", W, W == 1 ? "" : "s",
               call_ms[0], call_ms[1], call_ms[2], call_ms[3]);
        // calls that scanned the rc orientations without their window (a reset byte in the chunk, or the device packed it)
        printf(", \"calls_unwindowed_%d_worker%s\": %.3f", W, W == 1 ? "" : "s", (double)nset1.load() / (double)total);
        if (probe_on) printf(", \"probe_rescan_ms_per_call_%d_worker%s\": %.4f", W, W == 1 ? "" : "s", probe_call_ms);
        if (probe_on && W == workers.back())
            printf(", \"probe\": \"%s\", \"probe_annotated_per_pass\": %lld, \"probe_best_hit_us_under_%d_workers\": %.2f, \"probe_best_hit_calls\": %lld", PROBE, annotated, W,
                   bh_calls.load() ? (double)bh_ns.load() / 1e3 / (double)bh_calls.load() : 0.0, bh_calls.load());
        if (W == workers.back()) printf(", \"products_per_pass\": %lld, \"panel_device_slots\": %d, \"workers_on_device_cpus\": %d", products, ipcr_panel_device_slots(panel), bound.load());
        for (auto &sc : scs) ipcr_scratch_destroy(sc);
    }
    if (probe_on) { // ipcr_probe_best_hit on an idle device: amplicons of 180 and 2000 bases, 2000 calls each after 200 to warm
        for (uint64_t alen : {180ull, 2000ull}) {
            if (500000 + alen > n) continue;
            ipcr_probe_hit h;
            for (int i = 0; i < 200; ++i) (void)ipcr_probe_best_hit(&seq[500000], alen, PROBE, 2, &h);
            const double t0 = now();
            for (int i = 0; i < 2000; ++i)
                if (ipcr_probe_best_hit(&seq[500000], alen, PROBE, 2, &h) != IPCR_OK || !h.found || h.pos != 60) { rc = 10; break; }
            printf(", \"probe_best_hit_us_idle_%llu\": %.2f", (unsigned long long)alen, (now() - t0) * 1e6 / 2000.0);
        }
    } else
    {   // one worker, the whole record in one call
        ipcr_scratch *sc = nullptr;
        if (ipcr_scratch_create_on(panel, devices[0], &sc) != IPCR_OK) return 5;
        if (bind) (void)ipcr_bind_thread_to_device(devices[0]);
        double best = 0;
        for (int r = 0; r < 4; ++r) {
            const double t0 = now();
            if (ipcr_scan_chunk(panel, sc, seq.data(), n, nullptr, nullptr) != IPCR_OK) { rc = 6; break; }
            if (r) best = std::max(best, (double)n / (now() - t0) / 1e9);
        }
        printf(", \"gbases_per_s_whole_record\": %.2f", best);
        ipcr_scratch_destroy(sc);
    }
    printf("}\n");
    ipcr_panel_destroy(panel);
    return rc;
}
