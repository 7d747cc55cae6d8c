// fasta_hostpack.cpp -- FASTA file -> resident tiles with the text packed on the HOST and written through the PCIe BAR.
//
// The device loader (fasta.cpp: ipcr_genome_add_fasta) sends the file's bytes as they are, 1.0125 bytes per base: the link is
// what it waits for (17.8 of its 21 ms per Gb).  The page cache, on the other hand, hands sixteen threads 200-360 GB/s
// (tools/exp/mmap_read.cpp) and the BAR takes 44 GB/s of write-combined stores (tools/exp/bar_write.cpp): packed to two bits
// per base on the way, a Gb is 5 ms of reading and 5.7 ms of BAR.  This file does that for the files that allow it -- plain
// files whose records are lines of ONE width (the last one shorter), "\n" or "\r\n" ends, no blank or tab at a line's first or
// last base -- and says "not taken" for every other one: the device loader then loads it, as before.
//
// Same record rules as both other readers (core/fasta/path_ctx.go:126-144, stream.go:125-131, normalize.go:5-14): '>' at a line
// start begins a record, its ID is the header's text up to the first blank, text in front of the first header is ignored, a
// header without an ID drops its record, empty records are kept, a-z fold to A-Z.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include <emmintrin.h>

#include "cpu_pool.h"
#include "fasta_hostpack.h"
#include "hostpack.h"

namespace ipcr {

namespace {

bool is_space(uint8_t c) { return c == ' ' || (c >= '\t' && c <= '\r'); }

std::string header_id(const uint8_t *p, size_t n) { // stream.go:125-131: TrimSpace, then up to the first blank or tab
    size_t a = 0, b = n;
    while (a < b && is_space(p[a])) ++a;
    while (b > a && is_space(p[b - 1])) --b;
    size_t e = a;
    while (e < b && p[e] != ' ' && p[e] != '\t') ++e;
    return std::string((const char *)p + a, e - a);
}

// four tables of `stride` masks, indexed by (offset of a 64-byte block in its record's text) mod stride: where the block holds
// '\n', where '\r', where a line's first base, where its last
const uint64_t *fasta_tables(uint32_t W, uint32_t lt) {
    static std::mutex mu;
    static std::map<uint64_t, std::vector<uint64_t>> cache;
    std::lock_guard<std::mutex> lk(mu);
    std::vector<uint64_t> &t = cache[((uint64_t)W << 8) | lt];
    if (t.empty()) {
        const uint32_t stride = W + lt;
        t.assign((size_t)4 * stride, 0);
        for (uint32_t s = 0; s < stride; ++s)
            for (uint32_t b = 0; b < 64; ++b) {
                const uint32_t in_line = (s + b) % stride;
                if (in_line == W + lt - 1) t[s] |= 1ull << b;                   // '\n'
                if (lt == 2 && in_line == W) t[stride + s] |= 1ull << b;        // '\r'
                if (in_line == 0) t[2 * (size_t)stride + s] |= 1ull << b;       // a line's first base
                if (in_line == W - 1) t[3 * (size_t)stride + s] |= 1ull << b;   // ... and its last
            }
    }
    return t.data();
}

} // namespace

FastaText::~FastaText() {
    if (data && data != MAP_FAILED) munmap(const_cast<uint8_t *>(data), size ? size : 1);
    if (fd >= 0) close(fd);
}

// maps the file and lays its records out; false: not a file for this path (not regular, gzip, empty, irregular at first sight)
bool FastaText::open(const char *path) {
    fd = ::open(path, O_RDONLY);
    if (fd < 0) return false;
    struct stat sb;
    if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode) || sb.st_size < 2) return false;
    size = (size_t)sb.st_size;
    void *m = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (m == MAP_FAILED) { data = nullptr; return false; }
    data = static_cast<const uint8_t *>(m);
    if (data[0] == 0x1f && data[1] == 0x8b) return false; // gzip (core/fasta/open.go:29-50)
    const size_t pl = strlen(path);
    if (pl > 3 && strcmp(path + pl - 3, ".gz") == 0) return false;
    // ---- header lines: '>' at a line start (scan.go:27), found by the pool piece by piece
    const size_t piece = (size_t)4 << 20, np = (size + piece - 1) / piece;
    std::vector<std::vector<uint64_t>> found(np);
    PackPool::get().run(np, [&](size_t i) {
        const size_t a = i * piece, b = std::min(size, a + piece);
        for (const uint8_t *p = data + a; p < data + b;) {
            const void *q = memchr(p, '>', (size_t)(data + b - p));
            if (!q) break;
            const uint8_t *g = static_cast<const uint8_t *>(q);
            if (g == data || g[-1] == '\n') found[i].push_back((uint64_t)(g - data));
            p = g + 1;
        }
    });
    std::vector<uint64_t> hdr;
    for (auto &f : found) hdr.insert(hdr.end(), f.begin(), f.end());
    // ---- records
    for (size_t k = 0; k < hdr.size(); ++k) {
        const uint64_t p = hdr[k], next = k + 1 < hdr.size() ? hdr[k + 1] : (uint64_t)size;
        const void *nl = memchr(data + p, '\n', (size_t)(next - p));
        const uint64_t hend = nl ? (uint64_t)(static_cast<const uint8_t *>(nl) - data) : next; // the header's text is [p + 1, hend)
        FastaRecord r;
        r.id = header_id(data + p + 1, (size_t)(hend - p - 1));
        if (r.id.empty()) continue; // a header without an ID drops its record (path_ctx.go:127-129)
        r.region = nl ? hend + 1 : next;
        r.bytes = next - r.region;
        const uint8_t *reg = data + r.region;
        if (r.bytes) {
            const void *f = memchr(reg, '\n', (size_t)r.bytes);
            if (!f) { // one line without an end (the file's last)
                r.W = (uint32_t)std::min<uint64_t>(r.bytes, 0xFFFFFFFFu); r.lt = 1; r.full = 0;
                if (r.bytes > (1u << 20)) return false;
            } else {
                const uint64_t first = (uint64_t)(static_cast<const uint8_t *>(f) - reg);
                r.lt = first > 0 && reg[first - 1] == '\r' ? 2 : 1;
                r.W = (uint32_t)(first + 1 - r.lt);
                if (r.W == 0 || r.W + r.lt > 4096) return false;
                r.full = r.bytes / (r.W + r.lt);
            }
            // the last, shorter line: what is left behind the whole lines, with or without its end
            const uint64_t area = r.full * (r.W + r.lt), rem = r.bytes - area;
            uint64_t tb = rem;
            if (rem && reg[area + rem - 1] == '\n') {
                tb = rem - 1;
                if (tb && reg[area + tb - 1] == '\r') --tb;
            }
            if (tb > r.W) return false;                        // a last line longer than the others
            if (tb == r.W && rem != tb && r.full) return false; // a whole line with a shorter end than the others: mixed line ends
            for (uint64_t i = 0; i < tb; ++i)
                if (reg[area + i] == '\n' || reg[area + i] == '\r') return false;
            if (tb && (is_space(reg[area]) || is_space(reg[area + tb - 1]))) return false;
            r.tail = tb;
        }
        r.len = r.full * r.W + r.tail;
        records.push_back(std::move(r));
    }
    return true;
}

// one record's bases into linear planes (64-bit words: bit i of word w = base 64 w + i; `words` of them per plane = the
// record's columns x 64): lo / hi may be device memory behind the BAR (every word is written exactly once, none is read), iv is
// host memory.  dirty[g] is set for every group of `group_cols` columns that holds an invalid base.  false: irregular text.
bool FastaText::pack(const FastaRecord &r, uint64_t *lo, uint64_t *hi, uint64_t *iv, uint64_t words, uint8_t *dirty, uint64_t group_cols) const {
    const uint8_t *reg = data + r.region;
    const uint32_t stride = r.W + r.lt;
    const uint64_t area = r.full * stride, nblk = (area + 63) / 64;
    const uint64_t per = 16384; // blocks per piece: 1 MB of text
    const size_t np = (size_t)((nblk + per - 1) / per);
    std::vector<FastaEdge> edges(np);
    std::vector<uint32_t> inval(np, 0);
    std::atomic<bool> ok{true};
    if (np) {
        const uint64_t *tab = fasta_tables(r.W, r.lt);
        PackPool::get().run(np, [&](size_t k) {
            const uint64_t j0 = (uint64_t)k * per, j1 = std::min(nblk, j0 + per);
            if (!pack_fasta_blocks(reg, area, r.W, r.lt, tab, j0, j1, lo, hi, iv, &edges[k], &inval[k])) ok.store(false);
        });
        if (!ok.load()) return false;
    }
    // ---- the words pieces share, the last line, and what lies behind the record's end: composed here, written once
    std::map<uint64_t, std::array<uint64_t, 3>> part;
    for (const FastaEdge &e : edges)
        for (uint32_t i = 0; i < e.n; ++i) {
            auto &v = part[e.word[i]];
            v[0] |= e.val[i][0]; v[1] |= e.val[i][1]; v[2] |= e.val[i][2];
        }
    bool tail_invalid = false;
    for (uint64_t i = 0; i < r.tail; ++i) { // the last line, base by base (at most one line)
        const uint64_t b = r.full * r.W + i;
        const uint8_t c = reg[area + i], u = (uint8_t)(c & 0xDFu);
        auto &v = part[b >> 6];
        const uint64_t bit = 1ull << (b & 63u);
        if (u == 'A' || u == 'C' || u == 'G' || u == 'T') {
            const uint32_t code = u == 'A' ? 0u : u == 'C' ? 1u : u == 'G' ? 2u : 3u;
            if (code & 1u) v[0] |= bit;
            if (code & 2u) v[1] |= bit;
        } else { v[2] |= bit; tail_invalid = true; }
    }
    const uint64_t data_words = (r.len + 63) / 64; // words that hold a base
    if (r.len & 63u) part[r.len >> 6][2] |= ~0ull << (r.len & 63u); // padding in the record's last word: invalid
    for (auto &kv : part) {
        _mm_stream_si64(reinterpret_cast<long long *>(lo + kv.first), (long long)kv.second[0]);
        _mm_stream_si64(reinterpret_cast<long long *>(hi + kv.first), (long long)kv.second[1]);
        iv[kv.first] = kv.second[2];
    }
    for (uint64_t w = data_words; w < words; ++w) { // padding columns: code 0, invalid
        _mm_stream_si64(reinterpret_cast<long long *>(lo + w), 0);
        _mm_stream_si64(reinterpret_cast<long long *>(hi + w), 0);
        iv[w] = ~0ull;
    }
    _mm_sfence();
    // ---- which groups of columns hold an invalid base (their invalid-bit plane has to reach the device)
    const uint64_t group_bases = group_cols * 4096ull;
    for (size_t k = 0; k < np; ++k)
        if (inval[k]) {
            const uint64_t p0 = (uint64_t)k * per * 64u, p1 = std::min(area, p0 + per * 64u);
            const uint64_t b0 = (p0 / stride) * r.W + std::min<uint64_t>(p0 % stride, r.W), b1 = (p1 / stride) * r.W + std::min<uint64_t>(p1 % stride, r.W);
            for (uint64_t g = b0 / group_bases; g <= (b1 ? b1 - 1 : 0) / group_bases; ++g) dirty[g] = 1;
        }
    if (tail_invalid) dirty[(r.len - 1) / group_bases] = 1;
    return true;
}

uint32_t pack_linear(const uint8_t *seq, uint64_t len, uint64_t padded, uint32_t *lo, uint32_t *hi, uint32_t *iv, uint32_t *rs); // hostpack.cpp

} // namespace ipcr

// tests (CPU): the file through this packer, planes in host memory, against the streaming reader's records (ipcr_fasta_next)
// packed by pack_linear.  -1: the file is not taken (not plain / regular at first sight, no AVX-512 + BMI2), -2: refused while
// packing (irregular text), else the number of records that differ (0 = identical: IDs, lengths, every word of every plane)
extern "C" {
typedef struct ipcr_fasta ipcr_fasta;
int ipcr_fasta_open(const char *path, int64_t chunk_size, int64_t overlap, ipcr_fasta **out);
void ipcr_fasta_close(ipcr_fasta *f);
int ipcr_fasta_next(ipcr_fasta *f, const char **id, const uint8_t **seq, uint64_t *len, int32_t *got);

int32_t ipcr_internal_fasta_hostpack_check(const char *path) {
    using namespace ipcr;
    if (!fasta_blocks_supported()) return -1;
    FastaText t;
    if (!t.open(path)) return -1;
    ipcr_fasta *f = nullptr;
    if (ipcr_fasta_open(path, 0, 0, &f) != 0) return -3;
    int32_t bad = 0;
    size_t k = 0;
    for (;; ++k) {
        const char *id = nullptr;
        const uint8_t *seq = nullptr;
        uint64_t len = 0;
        int32_t got = 0;
        if (ipcr_fasta_next(f, &id, &seq, &len, &got) != 0) { bad = -3; break; }
        if (!got) break;
        if (k >= t.records.size()) { ++bad; continue; }
        const FastaRecord &r = t.records[k];
        const uint64_t two = 8192, cols = ((r.len + 128 + two - 1) / two) * 2, words = cols * 64, groups = (cols + 2047) / 2048;
        std::vector<uint64_t> lo(words, 0x5555555555555555ull), hi(words, 0x5555555555555555ull), iv(words, 0x5555555555555555ull);
        std::vector<uint8_t> dirty((size_t)groups, 0);
        if (!t.pack(r, lo.data(), hi.data(), iv.data(), words, dirty.data(), 2048)) { bad = -2; break; } // (the loader would take the device path)
        if (r.id != id || r.len != len) { ++bad; continue; }
        std::vector<uint32_t> wl(words * 2), wh(words * 2), wi(words * 2), wr(words * 2);
        const uint32_t fl = pack_linear(seq, len, cols * 4096, wl.data(), wh.data(), wi.data(), wr.data());
        bool any_dirty = false;
        for (uint8_t d : dirty) any_dirty |= d != 0;
        if (memcmp(wl.data(), lo.data(), words * 8) || memcmp(wh.data(), hi.data(), words * 8) || memcmp(wi.data(), iv.data(), words * 8) ||
            any_dirty != ((fl & 1u) != 0))
            ++bad;
    }
    if (bad >= 0 && k != t.records.size()) ++bad;
    ipcr_fasta_close(f);
    return bad;
}
}
