// cpu_pool.h -- the host threads of the library: which CPUs lie next to a device, and the process's pool of pack threads
// (ipcr_scan_chunk's packer for a lone worker, the FASTA loader's file reads, the parallel hit sort and join).
#pragma once
#include <sched.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace ipcr {

// the calling thread onto the CPUs next to HIP device `phys`; false when they are not known (or IPCR_BIND_THREADS=0)
bool bind_this_thread(int phys);
// pool thread `index` onto a core of its own next to the device, cores dealt round-robin over the L3 domains; false: topology unknown
bool bind_pool_thread(int phys, unsigned index);

// A few threads that pack slices of ONE large record (a single worker scanning whole chromosomes: a lone core packs
// ~10 GB/s of ASCII, the link carries 57): created at the first use, they live as long as the process.  A pool of
// workers never comes here -- every worker packs its own chunk.
class PackPool {
public:
    static PackPool &get() { static PackPool *p = new PackPool; return *p; } // never destroyed: its threads sleep on the condition variable until the process ends
    unsigned size() const { return (unsigned)threads_.size() + 1u; }
    // fn(i) for i in [0, n), on the pool's threads and the caller's; returns when all are done
    // phys >= 0: what the items write is pinned memory read by that device next -- the pool's own threads move onto its CPUs
    // (device_cpus; the caller's thread stays where its owner put it)
    // on_idle (optional): called again and again by the CALLER's thread while the pool works -- the caller then takes no
    // items itself (ipcr_scan_chunk sends a group of columns to the device the moment the pool has packed it)
    template <class F> void run(size_t n, F fn, int phys = -1, const std::function<void()> *on_idle = nullptr) {
        if (n == 0) return;
        std::unique_lock<std::mutex> big(run_mu_); // one record at a time
        // Every run is an object of its own: a pool thread that wakes late still holds the run it woke for -- whose
        // items are all taken, so it does nothing -- and never reads the fields of the run that has begun since.
        auto job = std::make_shared<Job>();
        job->fn = [&fn](size_t i) { fn(i); };
        job->n = n;
        job->phys = phys;
        job->taken.reset(new std::atomic<uint8_t>[n]);
        for (size_t i = 0; i < n; ++i) job->taken[i].store(0, std::memory_order_relaxed);
        // Every pool thread has a mailbox of its own (one cache line): the run goes into all of them -- reference counts taken
        // here, by one thread -- and then the generation moves on.  A polling thread that sees it takes the run out of ITS box and
        // begins with the item of its own number: no lock, no counter and no reference count shared with the fourteen others on
        // its way to the first byte (through one mutex they began 25 us apart, through one spin lock + one shared counter 15 us:
        // cache lines crossing between CCDs; the items themselves take 10-20 us).
        for (Mailbox &m : boxes_) {
            SpinGuard sg(m.lock);
            m.job = job;
        }
        gen_.fetch_add(1, std::memory_order_release);
        { std::lock_guard<std::mutex> lk(mu_); } // (a thread on its way to sleep has either seen the new generation or is waiting by now)
        cv_.notify_all();
        if (on_idle && !threads_.empty()) {
            while (job->done.load(std::memory_order_acquire) < n) { (*on_idle)(); __builtin_ia32_pause(); }
            return;
        }
        work(*job, ~(size_t)0);
        // (an item that has been taken is finished before `done` reaches n: fn is not called once this returns)
        for (unsigned spin = 0; job->done.load(std::memory_order_acquire) < n; ++spin) {
            if (spin < 2000u) { __builtin_ia32_pause(); continue; }
            std::unique_lock<std::mutex> lk(mu_);
            cv_done_.wait_for(lk, std::chrono::microseconds(200), [&] { return job->done.load(std::memory_order_acquire) >= n; });
        }
    }
private:
    struct Job {
        std::function<void(size_t)> fn;
        size_t n = 0;
        int phys = -1;
        std::unique_ptr<std::atomic<uint8_t>[]> taken; // per item: somebody has it
        std::atomic<size_t> next{0}, done{0};
    };
    struct alignas(64) Mailbox {
        std::atomic_flag lock = ATOMIC_FLAG_INIT;
        std::shared_ptr<Job> job;
    };
    struct SpinGuard {
        std::atomic_flag &f;
        explicit SpinGuard(std::atomic_flag &x) : f(x) { while (f.test_and_set(std::memory_order_acquire)) __builtin_ia32_pause(); }
        ~SpinGuard() { f.clear(std::memory_order_release); }
    };
    PackPool() {
        unsigned t = std::min(std::thread::hardware_concurrency(), 16u); // IPCR_PACK_THREADS: up to 64
        if (const char *v = getenv("IPCR_PACK_THREADS")) t = (unsigned)std::max(1, atoi(v));
        t = std::min(std::max(t, 1u), 64u);
        boxes_ = std::vector<Mailbox>(t > 1 ? t - 1 : 0);
        for (unsigned i = 1; i < t; ++i) threads_.emplace_back([this, i] { loop(i - 1); });
        for (auto &th : threads_) th.detach(); // they sleep on the condition variable for the rest of the process's life
    }
    void one(Job &j, size_t i) {
        j.fn(i);
        if (j.done.fetch_add(1, std::memory_order_acq_rel) + 1 >= j.n) { std::lock_guard<std::mutex> lk(mu_); cv_done_.notify_all(); }
    }
    // first: the item this thread begins with if nobody has it yet (its own number), then whatever the counter hands out -- the
    // counter runs over every item, so the item of a thread that sleeps is taken by the others
    void work(Job &j, size_t first) {
        if (first < j.n && j.taken[first].exchange(1, std::memory_order_acq_rel) == 0) one(j, first);
        for (;;) {
            const size_t i = j.next.fetch_add(1);
            if (i >= j.n) break;
            if (j.taken[i].exchange(1, std::memory_order_acq_rel) == 0) one(j, i);
        }
    }
    void loop(unsigned index) {
        uint64_t seen = 0;
        int bound = -1;
        Mailbox &box = boxes_[index];
        for (;;) {
            // a lone worker that scans chunk after chunk comes back every ~100 us: poll for that long before sleeping (a
            // wake-up through the condition variable costs 20-50 us of the ~25 us a 4 Mb chunk takes to pack)
            const auto t0 = std::chrono::steady_clock::now();
            bool changed = false;
            while (!(changed = gen_.load(std::memory_order_acquire) != seen)) {
                __builtin_ia32_pause();
                if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(150)) break;
            }
            if (!changed) {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return gen_.load(std::memory_order_acquire) != seen; });
            }
            seen = gen_.load(std::memory_order_acquire);
            std::shared_ptr<Job> job;
            {
                SpinGuard sg(box.lock); // (uncontended but for the moment the next run is being posted)
                job = std::move(box.job);
            }
            if (!job) continue; // (posted and taken already: this thread saw two generations in one look)
            if (job->phys >= 0 && job->phys != bound) { // onto a core of its own next to the device (or, failing that, anywhere next to it)
                if (!bind_pool_thread(job->phys, index)) (void)bind_this_thread(job->phys);
                bound = job->phys;
            }
            work(*job, index);
        }
    }
    std::mutex mu_, run_mu_;
    std::condition_variable cv_, cv_done_;
    std::vector<Mailbox> boxes_; // one per pool thread
    std::atomic<uint64_t> gen_{0};
    std::vector<std::thread> threads_;
};

} // namespace ipcr
