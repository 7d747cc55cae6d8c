// fasta.cpp -- FASTA record / rolling-chunk stream feeding the device tiles.
//
// Restates the reference's core/fasta package for the one job the scan path needs:
//   open (gzip by magic or .gz suffix, "-" = stdin)      core/fasta/open.go:29-50
//   line scan, '>' headers at line start                 core/fasta/scan.go:10-69
//   header ID = trimmed text up to the first blank/tab   core/fasta/stream.go:125-131
//   sequence lines: TrimSpace, a-z -> A-Z                core/fasta/normalize.go:5-14
//   sequence lines before the first header are ignored   core/fasta/path_ctx.go:142-144
//   rolling chunks "id:start-end", step = chunk-overlap  core/fasta/path_ctx.go:83-179
// The parse runs on the host (zlib inflate is serial anyway); records go to HBM through
// ipcr_genome_add_record's staging copy + pack kernel.
#include <zlib.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "ipcr_hip.h"

struct ipcr_fasta {
    gzFile fh = nullptr;
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

namespace {

bool is_space(uint8_t c) { // bytes.TrimSpace's ASCII set plus the two Latin-1 spaces it knows
    return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f' || c == 0x85 || c == 0xA0;
}

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
            const int n = gzread(f->fh, f->buf.data(), (unsigned)f->buf.size());
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
        f->out_seq = f->window;
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

} // namespace

extern "C" {

ipcr_status ipcr_fasta_open(const char *path, int64_t chunk_size, int64_t overlap, ipcr_fasta **out) {
    if (!path || !out) return ipcr_internal_fail(IPCR_ERR_INVALID, "ipcr_fasta_open: null argument");
    *out = nullptr;
    gzFile fh = strcmp(path, "-") == 0 ? gzdopen(0, "rb") : gzopen(path, "rb"); // gz or plain, transparently
    if (!fh) return ipcr_internal_fail(IPCR_ERR_INVALID, "cannot open %s", path);
    gzbuffer(fh, 1 << 20);
    ipcr_fasta *f = new ipcr_fasta;
    f->fh = fh;
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
        if (!read_line(f, f->line)) { // EOF
            f->finished = true;
            if (flush_record(f)) return deliver();
            return IPCR_OK;
        }
        if (f->line[0] == '>') { // scan.go:27 ('>' counts only at line start; read_line returns whole lines)
            f->pending_header = f->line.substr(1);
            f->header_pending = true;
            if (flush_record(f)) return deliver();
            continue;
        }
        if (!f->have_id) continue; // sequence before the first header (path_ctx.go:142-144)
        size_t a, b;
        trim(f->line, a, b); // normalize.go:5-14
        const size_t before = f->window.size();
        f->window.resize(before + (b - a));
        uint8_t *dst = f->window.data() + before;
        const uint8_t *src = (const uint8_t *)f->line.data() + a;
        for (size_t i = 0; i < b - a; ++i) {
            uint8_t c = src[i];
            if (c >= 'a' && c <= 'z') c = (uint8_t)(c - ('a' - 'A'));
            dst[i] = c;
        }
        f->total_len += b - a;
    }
}

// Load every record of a FASTA file into a resident genome (whole records, no chunking: one
// launch scans them all).  ids: '\n'-joined record IDs written to ids_out (NUL-terminated,
// truncated to cap); *n_added = records appended.
ipcr_status ipcr_genome_add_fasta(ipcr_genome *g, const char *path, uint32_t *n_added, char *ids_out, size_t cap,
                                  size_t *ids_needed) {
    if (!g || !path) return ipcr_internal_fail(IPCR_ERR_INVALID, "ipcr_genome_add_fasta: null argument");
    ipcr_fasta *f = nullptr;
    ipcr_status st = ipcr_fasta_open(path, 0, 0, &f);
    if (st != IPCR_OK) return st;
    std::string ids;
    uint32_t n = 0;
    for (;;) {
        const char *id;
        const uint8_t *seq;
        uint64_t len;
        int32_t got;
        st = ipcr_fasta_next(f, &id, &seq, &len, &got);
        if (st != IPCR_OK || !got) break;
        st = ipcr_genome_add_record(g, seq, len);
        if (st != IPCR_OK) break;
        if (n) ids.push_back('\n');
        ids += id;
        ++n;
    }
    ipcr_fasta_close(f);
    if (n_added) *n_added = n;
    if (ids_needed) *ids_needed = ids.size() + 1;
    if (ids_out && cap) {
        const size_t m = ids.size() < cap - 1 ? ids.size() : cap - 1;
        memcpy(ids_out, ids.data(), m);
        ids_out[m] = 0;
    }
    return st;
}

} // extern "C"
