// fasta.cpp -- FASTA record / rolling-chunk stream feeding the device tiles.
//
// Restates the reference's core/fasta package for the one job the scan path needs:
//   open (gzip by magic or .gz suffix, "-" = stdin)      core/fasta/open.go:29-50
//   line scan, '>' headers at line start                 core/fasta/scan.go:10-69
//   header ID = trimmed text up to the first blank/tab   core/fasta/stream.go:125-131
//   sequence lines: TrimSpace, a-z -> A-Z                core/fasta/normalize.go:5-14
//   sequence lines before the first header are ignored   core/fasta/path_ctx.go:142-144
//   rolling chunks "id:start-end", step = chunk-overlap  core/fasta/path_ctx.go:83-179
// Two consumers: the streaming reader (ipcr_fasta_next: host parse, what a worker of the drop-in
// pipeline or the CLI's chunked mode pulls from) and the resident-genome loader
// (ipcr_genome_add_fasta: raw slabs to the device, normalisation in fasta_kernels.hip).
#include <errno.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <hip/hip_runtime_api.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

#include "device_types.h"
#include "ipcr_hip.h"
#include "launch.h"

struct ipcr_fasta {
    gzFile fh = nullptr; // gzip input, or stdin
    int fd = -1;         // a plain file is read with read(2): zlib's transparent mode copies every byte through a buffer of its own
    std::vector<uint8_t> buf; // raw read buffer
    size_t bpos = 0, blen = 0;
    bool eof = false;
    bool at_line_start = true;
    // record state (path_ctx.go:91-107)
    std::string id;
    bool have_id = false;
    std::vector<uint8_t> window;
    uint64_t window_start = 0, total_len = 0, last_emitted_end = 0;
    bool emitted_chunk = false;
    int64_t chunk_size = 0, overlap = 0, step = 0;
    // pending output
    std::vector<uint8_t> out_seq;
    std::string out_id;
    std::string pending_header; // header seen while a record still had to be flushed
    bool header_pending = false;
    bool finished = false;
    std::string line; // current (possibly partial) line being assembled
};

extern ipcr_status ipcr_internal_fail(ipcr_status st, const char *fmt, ...);
// host.cpp: pack a record that already lies in device memory (16-byte aligned) and remember its ID
extern ipcr_status ipcr_internal_genome_add_device(ipcr_genome *g, const uint8_t *dseq, uint64_t len, const char *id);
extern hipStream_t ipcr_internal_genome_stream(ipcr_genome *g);
extern int ipcr_internal_genome_phys_device(const ipcr_genome *g);
// host.cpp: records lying anywhere in one device buffer, packed by one launch (d_tmp: room for the record table)
extern ipcr_status ipcr_internal_genome_add_batch(ipcr_genome *g, const uint8_t *dbase, const uint64_t *offs, const uint64_t *lens,
                                                  const std::string *ids, size_t n, void *d_tmp, size_t tmp_bytes);
extern size_t ipcr_internal_batch_table_bytes(size_t n);
// host.cpp: the process's pool of worker threads (created at first use, they live as long as the process)
extern void ipcr_internal_pool_run(size_t n, const std::function<void(size_t)> &fn, int phys); // phys >= 0: on that device's CPUs
extern bool ipcr_internal_bind_thread(int phys);
extern unsigned ipcr_internal_pool_size();
// host.cpp: the file loaded with its text packed on the host and written through the PCIe BAR (*taken = 1), or not a file for that
extern "C" ipcr_status ipcr_internal_genome_add_fasta_hostpacked(ipcr_genome *g, const char *path, uint32_t *n_added, std::string *ids, int *taken);

namespace {

// bytes.TrimSpace's ASCII set.  (Go also trims multi-byte Unicode spaces such as U+0085 / U+00A0 in
// their UTF-8 form; FASTA is ASCII, those stay sequence bytes here -- and are non-ACGT either way.)
bool is_space(uint8_t c) { return c == ' ' || (c >= '\t' && c <= '\r'); }

void trim(const std::string &s, size_t &a, size_t &b) {
    a = 0;
    b = s.size();
    while (a < b && is_space((uint8_t)s[a])) ++a;
    while (b > a && is_space((uint8_t)s[b - 1])) --b;
}

std::string parse_header_id(const std::string &hdr) { // stream.go:125-131
    size_t a, b;
    trim(hdr, a, b);
    size_t e = a;
    while (e < b && hdr[e] != ' ' && hdr[e] != '\t') ++e;
    return hdr.substr(a, e - a);
}

// read one line including its '\n' (if any); false at EOF with nothing read
bool read_line(ipcr_fasta *f, std::string &line) {
    line.clear();
    for (;;) {
        if (f->bpos == f->blen) {
            if (f->eof) return !line.empty();
            long n;
            if (f->fd >= 0) { do n = (long)::read(f->fd, f->buf.data(), f->buf.size()); while (n < 0 && errno == EINTR); }
            else n = gzread(f->fh, f->buf.data(), (unsigned)f->buf.size());
            if (n <= 0) { f->eof = true; f->bpos = f->blen = 0; return !line.empty(); }
            f->bpos = 0;
            f->blen = (size_t)n;
        }
        const uint8_t *p = f->buf.data() + f->bpos;
        const size_t avail = f->blen - f->bpos;
        const void *nl = memchr(p, '\n', avail);
        if (nl) {
            const size_t n = (size_t)((const uint8_t *)nl - p) + 1;
            line.append((const char *)p, n);
            f->bpos += n;
            return true;
        }
        line.append((const char *)p, avail);
        f->bpos = f->blen;
    }
}

// one line including its '\n' (if any) as a view: into the read buffer where the whole line lies inside it (no copy: all but
// one line per 4 MiB), assembled in f->line where it crosses a refill.  The view is good until the next call.
bool next_line(ipcr_fasta *f, const uint8_t *&lp, size_t &ln) {
    if (f->bpos < f->blen) {
        const uint8_t *p = f->buf.data() + f->bpos;
        if (const void *nl = memchr(p, '\n', f->blen - f->bpos)) {
            ln = (size_t)((const uint8_t *)nl - p) + 1;
            lp = p;
            f->bpos += ln;
            return true;
        }
    }
    if (!read_line(f, f->line)) return false;
    lp = (const uint8_t *)f->line.data();
    ln = f->line.size();
    return true;
}

void emit_chunk(ipcr_fasta *f, uint64_t start, uint64_t end, const uint8_t *seq, size_t n) { // path_ctx.go:109-124
    char tmp[64];
    snprintf(tmp, sizeof tmp, ":%llu-%llu", (unsigned long long)start, (unsigned long long)end);
    f->out_id = f->id + tmp;
    f->out_seq.assign(seq, seq + n);
    f->last_emitted_end = end;
    f->emitted_chunk = true;
}

// path_ctx.go:126-138 ; returns true when something was placed in out_*
bool flush_record(ipcr_fasta *f) {
    if (!f->have_id) return false;
    f->have_id = false;
    if (!f->emitted_chunk) {
        f->out_id = f->id;
        f->out_seq.swap(f->window); // (the whole record: handed over, not copied -- start_record clears the window)
        f->window.clear();
        return true;
    }
    if (f->last_emitted_end < f->total_len) {
        emit_chunk(f, f->window_start, f->total_len, f->window.data(), f->window.size());
        return true;
    }
    return false;
}

void start_record(ipcr_fasta *f, const std::string &header) { // path_ctx.go:100-107
    f->id = parse_header_id(header);
    f->have_id = !f->id.empty(); // a header without an ID drops its record (path_ctx.go:127-129,142-144)
    f->window.clear();
    f->window_start = f->total_len = f->last_emitted_end = 0;
    f->emitted_chunk = false;
}


// ---------------------------------------------------------------- device-side FASTA decode driver
#define FHIP(call)                                                                                         \
    do {                                                                                                   \
        const hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess) return ipcr_internal_fail(IPCR_ERR_DEVICE, "HIP: %s (%s)", hipGetErrorString(e_), #call); \
    } while (0)

// Pinned slabs the read-ahead cycles through.  Two were one too few: the read of slab j + 2 had to wait until slab j was
// decoded, i.e. for its copy AND the host's round trips behind it, and the link idled 0.5 ms in every 1.7 (1 GB file:
// 29.9 ms; the copies alone take 18).  With four the reader only ever waits for the link.
constexpr int NPIN = 4;

// The loader's big buffers (the pinned slabs, two raw device slabs, the compacted slab) are kept for the next load
// of the process instead of being freed: allocating and pinning them costs ~9 ms, a fifth of loading a 1 GB file.
struct FastaBuffers {
    int device = -1;
    size_t slab = 0;
    uint8_t *pin[NPIN] = {}, *d_raw = nullptr, *d_raw2 = nullptr, *d_out = nullptr;
    uint32_t *d_counts = nullptr;
    bool pinned() const { for (int k = 0; k < NPIN; ++k) if (!pin[k]) return false; return true; }
    void release() {
        for (int k = 0; k < NPIN; ++k) if (pin[k]) (void)hipHostFree(pin[k]);
        if (d_raw) (void)hipFree(d_raw);
        if (d_raw2) (void)hipFree(d_raw2);
        if (d_out) (void)hipFree(d_out);
        if (d_counts) (void)hipFree(d_counts);
        *this = FastaBuffers();
    }
};
std::mutex g_fasta_cache_mu;
FastaBuffers g_fasta_cache; // at most one set (IPCR_FASTA_CACHE=0: none)

struct FastaLoader {
    ipcr_genome *g = nullptr;
    int fd = -1;
    gzFile gz = nullptr; // gzip input (or stdin); plain files are read with read(2) straight into pinned memory
    bool eof = false;
    int64_t fsize = -1; // regular plain file: its size (read with pread at foff)
    uint64_t foff = 0;
    size_t slab = 0;
    hipStream_t st = nullptr;
    uint8_t *pin[NPIN] = {}, *d_raw = nullptr, *d_raw2 = nullptr, *d_out = nullptr, *d_rec = nullptr;
    int phys_device = 0;
    hipStream_t cs = nullptr;                  // the slabs' host-to-device copies: slab j + 1 crosses the link while slab j is decoded
    hipEvent_t ev_h2d[NPIN] = {};              // pinned slab k has crossed the link (its device copy is complete)
    hipEvent_t ev_free[2] = {nullptr, nullptr}; // the kernels that read d_raw[k] have run
    uint32_t *d_counts = nullptr, *d_hdr_off = nullptr, *h_small = nullptr; // h_small: pinned, hdr_off[nh] + total
    ipcr_fasta_range *d_hdr = nullptr;
    size_t hdr_cap = 0;
    uint64_t rec_cap = 0, rec_len = 0;
    std::vector<ipcr_fasta_range> ranges;
    bool at_line_start = true, lead_open = true;
    bool have_id = false;
    bool open_in_slab = false; // the open record began in the slab being processed (its bytes are all in d_out so far)
    std::string id, ids;
    std::vector<uint64_t> b_off, b_len; // records that begin and end inside the current slab: one pack launch for all
    std::vector<std::string> b_id;
    void *d_tab = nullptr;
    size_t tab_cap = 0;
    uint32_t n_added = 0;
    double t_read = 0, t_host = 0, t_decode = 0, t_pack = 0, t_alloc = 0;

    ~FastaLoader() {
        if (gz) gzclose(gz);
        else if (fd >= 0) close(fd);
        if (cs) { (void)hipStreamSynchronize(cs); (void)hipStreamDestroy(cs); }
        for (int k = 0; k < NPIN; ++k) if (ev_h2d[k]) (void)hipEventDestroy(ev_h2d[k]);
        for (int k = 0; k < 2; ++k) if (ev_free[k]) (void)hipEventDestroy(ev_free[k]);
        {   // the big buffers go to the cache (a set already there is dropped), or are freed
            FastaBuffers mine;
            mine.device = phys_device; mine.slab = slab;
            for (int k = 0; k < NPIN; ++k) mine.pin[k] = pin[k];
            mine.d_raw = d_raw; mine.d_raw2 = d_raw2; mine.d_out = d_out; mine.d_counts = d_counts;
            static const bool cache_on = !(getenv("IPCR_FASTA_CACHE") && atoi(getenv("IPCR_FASTA_CACHE")) == 0);
            const bool complete = mine.pinned() && d_raw && d_raw2 && d_out && d_counts;
            if (cache_on && complete) {
                std::lock_guard<std::mutex> lk(g_fasta_cache_mu);
                std::swap(mine, g_fasta_cache);
            }
            mine.release();
        }
        if (h_small) (void)hipHostFree(h_small);
        if (d_rec) (void)hipFree(d_rec);
        if (d_hdr_off) (void)hipFree(d_hdr_off);
        if (d_hdr) (void)hipFree(d_hdr);
        if (d_tab) (void)hipFree(d_tab);
    }

    ipcr_status open(ipcr_genome *genome, const char *path) {
        g = genome;
        st = ipcr_internal_genome_stream(g);
        const char *es = getenv("IPCR_FASTA_SLAB");
        slab = es && *es ? (size_t)strtoull(es, nullptr, 10) : ((size_t)64 << 20);
        if (slab < 64) slab = 64;
        slab = (slab + 15) & ~(size_t)15;
        if (strcmp(path, "-") == 0) {
            gz = gzdopen(0, "rb");
        } else {
            fd = ::open(path, O_RDONLY);
            if (fd < 0) return ipcr_internal_fail(IPCR_ERR_INVALID, "cannot open %s", path);
            unsigned char magic[2] = {0, 0};
            const ssize_t m = pread(fd, magic, 2, 0);
            const size_t pl = strlen(path);
            if ((m == 2 && magic[0] == 0x1f && magic[1] == 0x8b) || (pl > 3 && strcmp(path + pl - 3, ".gz") == 0)) { // open.go:29-50
                gz = gzdopen(fd, "rb");
                if (gz) gzbuffer(gz, 1 << 20);
            } else {
                struct stat sb;
                if (fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode)) fsize = (int64_t)sb.st_size;
            }
        }
        if (fd < 0 && !gz) return ipcr_internal_fail(IPCR_ERR_INVALID, "cannot open %s", path);
        const auto ta = std::chrono::steady_clock::now();
        FHIP(hipGetDevice(&phys_device));
        {
            std::lock_guard<std::mutex> lk(g_fasta_cache_mu);
            if (g_fasta_cache.pinned() && g_fasta_cache.device == phys_device && g_fasta_cache.slab == slab) {
                for (int k = 0; k < NPIN; ++k) pin[k] = g_fasta_cache.pin[k];
                d_raw = g_fasta_cache.d_raw; d_raw2 = g_fasta_cache.d_raw2;
                d_out = g_fasta_cache.d_out; d_counts = g_fasta_cache.d_counts;
                g_fasta_cache = FastaBuffers();
            }
        }
        if (!pin[0]) {
            for (int k = 0; k < NPIN; ++k) FHIP(hipHostMalloc((void **)&pin[k], slab, hipHostMallocDefault));
            FHIP(hipMalloc((void **)&d_raw, slab));
            FHIP(hipMalloc((void **)&d_raw2, slab));
            FHIP(hipMalloc((void **)&d_out, slab));
            FHIP(hipMalloc((void **)&d_counts, (slab / 4096 + 4) * 4));
        }
        FHIP(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
        for (int k = 0; k < NPIN; ++k) FHIP(hipEventCreateWithFlags(&ev_h2d[k], hipEventDisableTiming));
        for (int k = 0; k < 2; ++k) FHIP(hipEventCreateWithFlags(&ev_free[k], hipEventDisableTiming));
        t_alloc = std::chrono::duration<double>(std::chrono::steady_clock::now() - ta).count();
        return IPCR_OK;
    }

    // fill buf[have, slab) from the file; returns bytes now in the buffer.  Plain files are read in pieces of 2 MiB by
    // the process's pool of threads (one pread stream copies from the page cache at ~10 GB/s, the link takes 57; the
    // pool reaches ~70).  The threads must be the same from slab to slab: with 16 threads STARTED for every slab (round
    // 2, and this round until tools/ubench/slab_pipeline.hip took the pipeline apart) every slab copy running meanwhile
    // took 1.7-1.9 ms instead of 1.2 -- creating and ending threads maps and unmaps their stacks, and every change of the
    // address space makes the GPU driver revisit the process's pinned ranges.  And they run on the device's own socket
    // (host.cpp: device_cpus): a slab written by the other socket's cores leaves at 31 GB/s, not 54.
    // IPCR_FASTA_READERS=1: one stream, no pool.
    ipcr_status fill(uint8_t *buf, size_t have, size_t *n) {
        if (!gz && fsize >= 0) {
            const size_t want = (size_t)std::min<uint64_t>(slab - have, (uint64_t)fsize - foff);
            const size_t piece = (size_t)2 << 20;
            static const bool pooled = [] { const char *v = getenv("IPCR_FASTA_READERS"); return !(v && *v && atoi(v) <= 1); }();
            std::atomic<bool> bad{false};
            auto reader = [&](size_t a, size_t b) {
                while (a < b) {
                    const ssize_t r = pread(fd, buf + have + a, b - a, (off_t)(foff + a));
                    if (r <= 0) { bad = true; return; }
                    a += (size_t)r;
                }
            };
            const size_t np = (want + piece - 1) / piece;
            if (np <= 1 || !pooled) reader(0, want);
            else ipcr_internal_pool_run(np, [&](size_t i) { reader(i * piece, std::min(want, (i + 1) * piece)); }, phys_device);
            if (bad) return ipcr_internal_fail(IPCR_ERR_INVALID, "read error in FASTA input");
            foff += want;
            have += want;
            if (foff >= (uint64_t)fsize) eof = true;
            *n = have;
            return IPCR_OK;
        }
        while (have < slab && !eof) {
            long r;
            if (gz) r = gzread(gz, buf + have, (unsigned)std::min<size_t>(slab - have, 1u << 30));
            else r = (long)::read(fd, buf + have, slab - have);
            if (r < 0) return ipcr_internal_fail(IPCR_ERR_INVALID, "read error in FASTA input");
            if (r == 0) eof = true;
            have += (size_t)r;
        }
        *n = have;
        return IPCR_OK;
    }

    void count_record() {
        if (n_added) ids.push_back('\n');
        ids += id;
        ++n_added;
    }

    ipcr_status flush_batch() {
        const size_t n = b_off.size();
        if (n == 0) return IPCR_OK;
        const size_t need = ipcr_internal_batch_table_bytes(n);
        if (need > tab_cap) {
            if (d_tab) (void)hipFree(d_tab);
            d_tab = nullptr;
            tab_cap = need * 2;
            FHIP(hipMalloc(&d_tab, tab_cap));
        }
        const ipcr_status s = ipcr_internal_genome_add_batch(g, d_out, b_off.data(), b_len.data(), b_id.data(), n, d_tab, tab_cap);
        b_off.clear(); b_len.clear(); b_id.clear();
        return s;
    }

    ipcr_status finish_record() {
        if (!have_id) return IPCR_OK;
        const ipcr_status s = ipcr_internal_genome_add_device(g, d_rec ? d_rec : d_out, rec_len, id.c_str());
        if (s != IPCR_OK) return s;
        count_record();
        have_id = false;
        rec_len = 0;
        return IPCR_OK;
    }

    ipcr_status append(uint64_t a, uint64_t b) { // compacted bytes [a, b) of this slab belong to the open record
        if (!have_id || b <= a) return IPCR_OK;
        const uint64_t need = rec_len + (b - a);
        if (need > rec_cap) {
            uint64_t want = std::max<uint64_t>(need + (need >> 2), (uint64_t)1 << 20);
            want = (want + 255) & ~(uint64_t)255;
            uint8_t *nb = nullptr;
            FHIP(hipMalloc((void **)&nb, want));
            if (rec_len) FHIP(hipMemcpyAsync(nb, d_rec, rec_len, hipMemcpyDeviceToDevice, st));
            FHIP(hipStreamSynchronize(st));
            if (d_rec) (void)hipFree(d_rec);
            d_rec = nb;
            rec_cap = want;
        }
        FHIP(hipMemcpyAsync(d_rec + rec_len, d_out + a, b - a, hipMemcpyDeviceToDevice, st));
        rec_len = need;
        return IPCR_OK;
    }

    // header lines of the slab's bytes [0, cut), already on their way to d_raw: found on the device ('>' at a line start,
    // scan.go:27) -- the host reads none of the sequence bytes; it sorts the handful of ranges and parses the IDs out of
    // the pinned slab.  (Round 2 searched the slab on the host with 8 threads: 0.9-1.5 ms per 64 MiB on the critical path.)
    ipcr_status find_headers(const uint8_t *d_raw, size_t cut) {
        ranges.clear();
        for (;;) {
            if (hdr_cap == 0) {
                hdr_cap = 4096;
                FHIP(hipMalloc((void **)&d_hdr, hdr_cap * sizeof(ipcr_fasta_range)));
                FHIP(hipMalloc((void **)&d_hdr_off, hdr_cap * 4));
                FHIP(hipHostMalloc((void **)&h_small, (hdr_cap + 4) * 4 + hdr_cap * sizeof(ipcr_fasta_range), hipHostMallocDefault));
            }
            uint32_t *d_count = d_counts + (slab / 4096 + 1); // one spare word behind the block counts
            FHIP(ipcr::launch_fasta_find_headers(st, d_raw, cut, at_line_start ? 1u : 0u, d_hdr, (uint32_t)hdr_cap, d_count));
            FHIP(hipMemcpyAsync(h_small, d_count, 4, hipMemcpyDeviceToHost, st));
            FHIP(hipStreamSynchronize(st));
            const uint32_t nh = h_small[0];
            if (nh > hdr_cap) { // a slab of very short records: a larger list, again
                (void)hipFree(d_hdr); (void)hipFree(d_hdr_off); (void)hipHostFree(h_small);
                d_hdr = nullptr; d_hdr_off = nullptr; h_small = nullptr;
                hdr_cap = (size_t)nh + (nh >> 2) + 64;
                FHIP(hipMalloc((void **)&d_hdr, hdr_cap * sizeof(ipcr_fasta_range)));
                FHIP(hipMalloc((void **)&d_hdr_off, hdr_cap * 4));
                FHIP(hipHostMalloc((void **)&h_small, (hdr_cap + 4) * 4 + hdr_cap * sizeof(ipcr_fasta_range), hipHostMallocDefault));
                continue;
            }
            if (nh) {
                ipcr_fasta_range *hr = reinterpret_cast<ipcr_fasta_range *>(h_small + hdr_cap + 1 + ((hdr_cap + 1) & 1u)); // 8-byte aligned, behind the offsets
                FHIP(hipMemcpyAsync(hr, d_hdr, (size_t)nh * sizeof(ipcr_fasta_range), hipMemcpyDeviceToHost, st));
                FHIP(hipStreamSynchronize(st));
                ranges.assign(hr, hr + nh);
                std::sort(ranges.begin(), ranges.end(), [](const ipcr_fasta_range &x, const ipcr_fasta_range &y) { return x.start < y.start; });
            }
            return IPCR_OK;
        }
    }

    // What the read-ahead thread hands over: slab j sits in pinned buffer j % NPIN, its bytes [0, cut) are on their way to
    // d_raw[j & 1] (ev_h2d[j % NPIN]); what follows the cut is carried into the next slab.
    struct SlabInfo { size_t n = 0, cut = 0; bool last = false, has_nl = true; ipcr_status st = IPCR_OK; };

    ipcr_status run() {
        uint8_t **buf = pin;
        uint8_t *draw[2] = {d_raw, d_raw2};
        std::mutex mu;
        std::condition_variable cv;
        std::deque<SlabInfo> ready;
        uint64_t consumed = 0; // slabs the main thread has finished with (their pinned buffer may be overwritten)
        uint64_t freed = 0;    // slabs whose decode has been queued: ev_free of their device slab has been recorded
        bool stop = false;
        // IPCR_DEBUG_TIMES: a time line of both threads (label, slab, ms since run() began), printed at the end
        const bool trace = getenv("IPCR_DEBUG_TIMES") != nullptr;
        const auto t_run0 = std::chrono::steady_clock::now();
        std::mutex tmu;
        std::vector<std::tuple<const char *, uint64_t, double>> tl;
        auto mark = [&](const char *what, uint64_t j) {
            if (!trace) return;
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_run0).count();
            std::lock_guard<std::mutex> lk(tmu);
            tl.emplace_back(what, j, ms);
        };
        std::vector<hipEvent_t> tev; // trace: start / stop of every slab copy on the copy stream
        // ---- read-ahead: file -> pinned slab (a few pread threads), cut behind the last line end, copy to the device on
        // the copy stream.  It runs up to NPIN - 1 slabs ahead of the decode: the read of slab j + NPIN waits for slab j to be consumed.
        std::thread reader([&]() {
            (void)hipSetDevice(phys_device);
            (void)ipcr_internal_bind_thread(phys_device); // it reads the file with the pool: on the device's own socket (host.cpp: device_cpus)
            size_t carry = 0, cut_prev = 0;
            for (uint64_t j = 0;; ++j) {
                SlabInfo si;
                uint8_t *b = buf[j % NPIN];
                if (j >= (uint64_t)NPIN) {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return stop || consumed + NPIN > j; }); // slab j - NPIN is done with on the host ...
                    if (stop) return;
                    lk.unlock();
                    (void)hipEventSynchronize(ev_h2d[j % NPIN]);             // ... and has left the pinned buffer
                }
                if (carry) memcpy(b, buf[(j - 1) % NPIN] + cut_prev, carry);
                size_t n = 0;
                const auto t0 = std::chrono::steady_clock::now();
                mark("fill>", j);
                si.st = fill(b, carry, &n);
                mark("fill<", j);
                t_read += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                si.n = n;
                si.last = eof;
                size_t cut = n;
                if (si.st == IPCR_OK && !si.last && n) {
                    // where to cut: behind the last line end; a slab without one is cut in front of its trailing
                    // white space (whether that is kept depends on what follows)
                    const void *nl = memrchr(b, '\n', n);
                    if (nl) cut = (size_t)((const uint8_t *)nl - b) + 1;
                    else {
                        si.has_nl = false; // no line ends in this slab: a header that begins in it does not end in it
                        while (cut > 0 && is_space(b[cut - 1])) --cut;
                        if (cut == 0 && n == slab) si.st = ipcr_internal_fail(IPCR_ERR_UNSUPPORTED, "FASTA: a run of white space longer than the %zu-byte slab", slab);
                    }
                }
                si.cut = cut;
                if (si.st == IPCR_OK) {
                    hipError_t e = hipSuccess;
                    if (j >= 2) { // the kernels over slab j - 2 have read d_raw[j & 1]: their event must have been recorded first
                        std::unique_lock<std::mutex> lk(mu);
                        cv.wait(lk, [&] { return stop || freed + 2 > j; });
                        if (stop) return;
                        lk.unlock();
                        e = hipStreamWaitEvent(cs, ev_free[j & 1], 0);
                    }
                    if (trace && j < 64) { hipEvent_t a, z; (void)hipEventCreate(&a); (void)hipEventCreate(&z); tev.push_back(a); tev.push_back(z); (void)hipEventRecord(a, cs); }
                    // One slab copy at a time: the next one is queued when the one before has finished.  Queued behind a
                    // copy still in flight it was slower on every box tried (1 GB file: 37 -> 46-52 ms on a slow one,
                    // round 3); why, the runtime does not say.  The fill of the next slab has long begun
                    // by the time this thread waits here, so the link idles only for the few microseconds of the hand-over.
                    static const bool one_copy = !(getenv("IPCR_FASTA_COPY_OVERLAP") && atoi(getenv("IPCR_FASTA_COPY_OVERLAP")));
                    if (one_copy && j >= 1 && e == hipSuccess) e = hipEventSynchronize(ev_h2d[(j - 1) % NPIN]);
                    if (e == hipSuccess && cut) e = hipMemcpyAsync(draw[j & 1], b, cut, hipMemcpyHostToDevice, cs);
                    if (trace && j < 64) (void)hipEventRecord(tev.back(), cs);
                    if (e == hipSuccess) e = hipEventRecord(ev_h2d[j % NPIN], cs);
                    mark("h2d queued", j);
                    if (e != hipSuccess) si.st = ipcr_internal_fail(IPCR_ERR_DEVICE, "HIP: %s (FASTA slab copy)", hipGetErrorString(e));
                }
                const bool end = si.st != IPCR_OK || si.last || n == 0;
                {
                    std::lock_guard<std::mutex> lk(mu);
                    ready.push_back(si);
                }
                cv.notify_all();
                if (end) return;
                carry = n - cut;
                cut_prev = cut;
            }
        });
        struct Stopper { // whatever way run() is left: the reader ends and is joined
            std::mutex &mu; std::condition_variable &cv; bool &stop; std::thread &t;
            ~Stopper() { { std::lock_guard<std::mutex> lk(mu); stop = true; } cv.notify_all(); if (t.joinable()) t.join(); }
        } stopper{mu, cv, stop, reader};
        ipcr_status s = IPCR_OK;
        for (uint64_t j = 0;; ++j) {
            SlabInfo si;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return !ready.empty(); });
                si = ready.front();
                ready.pop_front();
            }
            if (si.st != IPCR_OK) return si.st;
            const size_t n = si.n, cut = si.cut;
            if (n == 0) break;
            const bool last = si.last;
            const uint8_t *raw = buf[j % NPIN];
            uint8_t *d_cur = draw[j & 1];
            const auto th0 = std::chrono::steady_clock::now();
            mark("slab taken", j);
            FHIP(hipStreamWaitEvent(st, ev_h2d[j % NPIN], 0));
            s = find_headers(d_cur, cut);
            if (s != IPCR_OK) return s;
            const uint32_t nh = (uint32_t)ranges.size();
            // A slab is cut behind its last line end, so a header that begins in it ends in it -- unless the slab holds no line
            // end at all: then the header line is longer than the slab (the kernel has clamped its range to the cut: the ID
            // would be truncated and the rest of the line decoded as sequence).
            if (nh && !last && !si.has_nl) return ipcr_internal_fail(IPCR_ERR_UNSUPPORTED, "FASTA: a header line longer than the %zu-byte slab", slab);
            uint32_t total = 0;
            const auto td0 = std::chrono::steady_clock::now();
            mark("headers found", j);
            t_host += std::chrono::duration<double>(td0 - th0).count();
            if (cut) {
                const uint32_t nb = (uint32_t)((cut + 4095) / 4096);
                if (nh) FHIP(hipMemcpyAsync(d_hdr, ranges.data(), (size_t)nh * sizeof(ipcr_fasta_range), hipMemcpyHostToDevice, st));
                FHIP(ipcr::launch_fasta_decode(st, d_cur, cut, d_hdr, nh, (at_line_start || lead_open) ? 1u : 0u, d_counts, d_out, d_hdr_off));
                FHIP(hipEventRecord(ev_free[j & 1], st));
                { std::lock_guard<std::mutex> lk(mu); freed = j + 1; }
                cv.notify_all();
                if (nh) FHIP(hipMemcpyAsync(h_small, d_hdr_off, (size_t)nh * 4, hipMemcpyDeviceToHost, st));
                FHIP(hipMemcpyAsync(h_small + nh, d_counts + nb, 4, hipMemcpyDeviceToHost, st));
                FHIP(hipStreamSynchronize(st));
                total = h_small[nh];
            } else {
                FHIP(hipEventRecord(ev_free[j & 1], st));
                { std::lock_guard<std::mutex> lk(mu); freed = j + 1; }
                cv.notify_all();
            }
            const auto tp0 = std::chrono::steady_clock::now();
            mark("decoded", j);
            t_decode += std::chrono::duration<double>(tp0 - td0).count();
            // hand the compacted bytes to the records.  A record that begins and ends in this slab is packed straight
            // out of d_out with all the others like it (one launch); the one that came in open and the one that stays
            // open go through the record buffer
            uint64_t a = 0;
            for (uint32_t k = 0; k < nh; ++k) {
                const uint64_t b = h_small[k];
                if (have_id && open_in_slab) {
                    b_off.push_back(a); b_len.push_back(b - a); b_id.push_back(id);
                    count_record();
                    have_id = false;
                } else {
                    s = append(a, b);
                    if (s == IPCR_OK) s = finish_record(); // path_ctx.go:164-170: a header flushes the open record
                    if (s != IPCR_OK) return s;
                }
                a = b;
                const std::string hdr((const char *)raw + ranges[k].start + 1, (size_t)(ranges[k].end - ranges[k].start - 1));
                id = parse_header_id(hdr);
                have_id = !id.empty(); // a header without an ID drops its record
                open_in_slab = true;
                rec_len = 0;
            }
            if (last && have_id && open_in_slab) { // the file ends here: the open record is complete too
                b_off.push_back(a); b_len.push_back(total - a); b_id.push_back(id);
                count_record();
                have_id = false;
            } else {
                s = append(a, total);
                if (s != IPCR_OK) return s;
            }
            open_in_slab = false; // whatever is still open continues in the record buffer
            s = flush_batch();    // (waits for the stream: the record table is a host vector of that call)
            if (s != IPCR_OK) return s;
            FHIP(hipStreamSynchronize(st)); // d_out is reused by the next slab
            t_pack += std::chrono::duration<double>(std::chrono::steady_clock::now() - tp0).count();
            mark("packed", j);
            if (cut > 0) {
                at_line_start = raw[cut - 1] == '\n';
                lead_open = at_line_start; // a cut inside a line is behind one of its non-blank bytes
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                consumed = j + 1;
            }
            cv.notify_all();
            if (last) break; // cut == n: everything has been consumed
        }
        const ipcr_status fs = finish_record();
        if (getenv("IPCR_DEBUG_TIMES"))
            fprintf(stderr, "fasta loader: buffers %.3f s, read %.3f s (overlapped), cut + header search %.3f s, h2d + decode %.3f s, copy + pack %.3f s\n",
                    t_alloc, t_read, t_host, t_decode, t_pack);
        if (trace && getenv("IPCR_DEBUG_TIMES")[0] == '2') {
            std::sort(tl.begin(), tl.end(), [](const auto &x, const auto &y) { return std::get<2>(x) < std::get<2>(y); });
            (void)hipStreamSynchronize(cs);
            for (size_t k = 0; k + 1 < tev.size(); k += 2) {
                float dur = 0, since0 = 0;
                (void)hipEventElapsedTime(&dur, tev[k], tev[k + 1]);
                (void)hipEventElapsedTime(&since0, tev[0], tev[k]);
                fprintf(stderr, "  copy of slab %2zu: starts %.3f ms after the first, takes %.3f ms\n", k / 2, since0, dur);
            }
            for (hipEvent_t e : tev) (void)hipEventDestroy(e);
            for (const auto &e : tl) fprintf(stderr, "  %8.3f ms  slab %2llu  %s\n", std::get<2>(e), (unsigned long long)std::get<1>(e), std::get<0>(e));
        }
        return fs;
    }
};
} // namespace

extern "C" {

ipcr_status ipcr_fasta_open(const char *path, int64_t chunk_size, int64_t overlap, ipcr_fasta **out) {
    if (!path || !out) return ipcr_internal_fail(IPCR_ERR_INVALID, "ipcr_fasta_open: null argument");
    *out = nullptr;
    gzFile fh = strcmp(path, "-") == 0 ? gzdopen(0, "rb") : gzopen(path, "rb"); // gz or plain, transparently
    if (!fh) return ipcr_internal_fail(IPCR_ERR_INVALID, "cannot open %s", path);
    gzbuffer(fh, 1 << 20);
    int fd = -1;
    if (strcmp(path, "-") != 0 && gzdirect(fh)) { // not gzip: read the file itself
        fd = ::open(path, O_RDONLY);
        if (fd >= 0) { gzclose(fh); fh = nullptr; }
    }
    ipcr_fasta *f = new ipcr_fasta;
    f->fh = fh;
    f->fd = fd;
    f->buf.resize(1 << 22);
    f->chunk_size = chunk_size;
    f->overlap = overlap;
    f->step = chunk_size - overlap;
    if (chunk_size <= 0 || f->step <= 0) f->step = 0; // whole records (path_ctx.go:87-90)
    *out = f;
    return IPCR_OK;
}

void ipcr_fasta_close(ipcr_fasta *f) {
    if (!f) return;
    if (f->fh) gzclose(f->fh);
    if (f->fd >= 0) ::close(f->fd);
    delete f;
}

// next record or chunk: *seq stays valid until the next call; returns IPCR_OK with *got = 0 at EOF
ipcr_status ipcr_fasta_next(ipcr_fasta *f, const char **id, const uint8_t **seq, uint64_t *len, int32_t *got) {
    if (!f || !id || !seq || !len || !got) return ipcr_internal_fail(IPCR_ERR_INVALID, "ipcr_fasta_next: null argument");
    *got = 0;
    auto deliver = [&]() {
        *id = f->out_id.c_str();
        *seq = f->out_seq.data();
        *len = f->out_seq.size();
        *got = 1;
        return IPCR_OK;
    };
    if (f->finished) return IPCR_OK;
    for (;;) {
        if (f->header_pending) { // the previous record has been flushed; open the new one
            f->header_pending = false;
            start_record(f, f->pending_header);
        }
        // a full chunk waiting in the window? (path_ctx.go:148-160)
        if (f->have_id && f->step > 0 && (int64_t)f->window.size() > f->chunk_size) {
            emit_chunk(f, f->window_start, f->window_start + (uint64_t)f->chunk_size, f->window.data(), (size_t)f->chunk_size);
            if ((size_t)f->step >= f->window.size()) f->window.clear();
            else f->window.erase(f->window.begin(), f->window.begin() + f->step);
            f->window_start += (uint64_t)f->step;
            return deliver();
        }
        const uint8_t *lp = nullptr;
        size_t ln = 0;
        if (!next_line(f, lp, ln)) { // EOF
            f->finished = true;
            if (flush_record(f)) return deliver();
            return IPCR_OK;
        }
        if (lp[0] == '>') { // scan.go:27 ('>' counts only at line start; next_line returns whole lines)
            f->pending_header.assign((const char *)lp + 1, ln - 1);
            f->header_pending = true;
            if (flush_record(f)) return deliver();
            continue;
        }
        if (!f->have_id) continue; // sequence before the first header (path_ctx.go:142-144)
        size_t a = 0, b = ln; // normalize.go:5-14: TrimSpace, then a-z -> A-Z
        while (a < b && is_space(lp[a])) ++a;
        while (b > a && is_space(lp[b - 1])) --b;
        const size_t before = f->window.size(), n = b - a;
        f->window.insert(f->window.end(), lp + a, lp + b);
        uint8_t *dst = f->window.data() + before;
        for (size_t i = 0; i < n; ++i) { // (branch-free: the compiler makes vector code of it)
            const uint8_t c = dst[i];
            dst[i] = (uint8_t)(c - (((uint8_t)(c - 'a') < 26u) ? 32u : 0u));
        }
        f->total_len += n;
    }
}

// Load every record of a FASTA file into a resident genome (whole records, no chunking: one
// launch scans them all).  The file is read in slabs into pinned memory and copied to the device as
// it is; the host only finds the header lines and parses their IDs, the device strips line ends /
// white space, folds case and compacts (fasta_kernels.hip), and every finished record goes through
// the pack kernel.  Same record semantics as the streaming reader above (and as the reference):
// header ID up to the first blank, sequence before the first header ignored, a header without an
// ID drops its record, empty records are kept.
// ids: '\n'-joined record IDs written to ids_out (NUL-terminated, truncated to cap);
// *n_added = records appended.  IPCR_FASTA_SLAB = slab bytes (default 64 MiB, tests use tiny ones).
ipcr_status ipcr_genome_add_fasta(ipcr_genome *g, const char *path, uint32_t *n_added, char *ids_out, size_t cap,
                                  size_t *ids_needed) {
    if (!g || !path) return ipcr_internal_fail(IPCR_ERR_INVALID, "ipcr_genome_add_fasta: null argument");
    // the genome's device, whatever the calling thread had selected (put back on return)
    struct OnDevice {
        int prev = -1, want;
        explicit OnDevice(int d) : want(d) { if (hipGetDevice(&prev) == hipSuccess && prev != want) (void)hipSetDevice(want); else prev = want; }
        ~OnDevice() { if (prev != want) (void)hipSetDevice(prev); }
    } on_device(ipcr_internal_genome_phys_device(g));
    if (n_added) *n_added = 0;
    // the fast way in first: the text packed on the host and written through the PCIe BAR (host.cpp, fasta_hostpack.cpp) -- for
    // the files and hosts that allow it; every other file goes through the loader below, as before
    {
        uint32_t added = 0;
        std::string fast_ids;
        int taken = 0;
        const ipcr_status fst = ipcr_internal_genome_add_fasta_hostpacked(g, path, &added, &fast_ids, &taken);
        if (fst != IPCR_OK) return fst;
        if (taken) {
            if (n_added) *n_added = added;
            if (ids_needed) *ids_needed = fast_ids.size() + 1;
            if (ids_out && cap) {
                const size_t m = fast_ids.size() < cap - 1 ? fast_ids.size() : cap - 1;
                memcpy(ids_out, fast_ids.data(), m);
                ids_out[m] = 0;
            }
            return IPCR_OK;
        }
    }
    FastaLoader L;
    ipcr_status st = L.open(g, path);
    if (st == IPCR_OK) st = L.run();
    if (n_added) *n_added = L.n_added;
    if (ids_needed) *ids_needed = L.ids.size() + 1;
    if (ids_out && cap) {
        const size_t m = L.ids.size() < cap - 1 ? L.ids.size() : cap - 1;
        memcpy(ids_out, L.ids.data(), m);
        ids_out[m] = 0;
    }
    return st;
}

} // extern "C"
