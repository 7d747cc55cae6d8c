// hostpack.cpp -- ASCII bases -> bit planes on the HOST, for the drop-in call's way over PCIe.
//
// ipcr_scan_chunk gets host ASCII (1 byte per base; Engine.ForEachCompiledProduct, core/engine/compiled.go:162-267) and
// has to touch every byte on the CPU anyway to stage it in pinned memory.  Instead of copying, the staging pass packs:
// per 32 bases one word each of lo, hi (2-bit code A=0 C=1 G=2 T=3), inv (not an upper-case ACGT: core/primer/iupac.go:62-67)
// -- 0.375 bytes per base cross the link instead of 1 -- and, only for a chunk that holds lower-case acgt, rst (byte
// outside ACGTacgt, core/engine/ac.go:16-30; otherwise rst = inv and the device copies it).  The planes are LINEAR (bit i of
// word w = base 32 w + i); the device turns them into strand-major tiles (kernels.hip: tiles_from_linear_kernel).
// Same semantics as kernels.hip: pack_pair, bit for bit: lo/hi are zero for a byte that is not one of ACGTacgt, bases
// past the end are inv = 1, rst = 0.
#include <immintrin.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "hostpack.h"
#include "ipcr_hip.h"

ipcr_status ipcr_internal_fail(ipcr_status st, const char *fmt, ...);

namespace {

inline void classify_scalar(uint8_t c, uint32_t &lo, uint32_t &hi, uint32_t &iv, uint32_t &rs) {
    const uint8_t u = (uint8_t)(c & 0xDFu);
    const bool acgt = u == 'A' || u == 'C' || u == 'G' || u == 'T';
    const bool lower = (c & 0x20u) != 0;
    const uint32_t b1 = (c >> 1) & 1u, b2 = (c >> 2) & 1u;
    lo = acgt ? (b1 ^ b2) : 0u;
    hi = acgt ? b2 : 0u;
    iv = (acgt && !lower) ? 0u : 1u;
    rs = acgt ? 0u : 1u;
}

// bases [0, n) of seq -> words; n a multiple of 32.  Returns OR of (any rst ? 1) | (any lower-case acgt ? 2)
uint32_t pack_scalar(const uint8_t *seq, uint64_t n, uint32_t *lo, uint32_t *hi, uint32_t *iv, uint32_t *rs) {
    uint32_t flags = 0;
    for (uint64_t w = 0; w < n / 32u; ++w) {
        uint32_t a = 0, b = 0, c = 0, d = 0;
        for (uint32_t i = 0; i < 32u; ++i) {
            uint32_t l, h, v, r;
            classify_scalar(seq[w * 32u + i], l, h, v, r);
            a |= l << i; b |= h << i; c |= v << i; d |= r << i;
        }
        lo[w] = a; hi[w] = b; iv[w] = c; rs[w] = d;
        if (d) flags |= 1u;
        if (c & ~d) flags |= 2u;
    }
    return flags;
}

__attribute__((target("avx2"))) uint32_t pack_avx2(const uint8_t *seq, uint64_t n, uint32_t *lo, uint32_t *hi, uint32_t *iv, uint32_t *rs) {
    // the letter a byte would have to be, by its low nibble: A 0x41, C 0x43, T 0x54, G 0x47; the filler of every other
    // slot has another low nibble than the slot, so no byte can equal it
    const __m256i lut = _mm256_setr_epi8(0x01, 'A', 0, 'C', 'T', 0, 0, 'G', 0, 0, 0, 0, 0, 0, 0, 0,
                                         0x01, 'A', 0, 'C', 'T', 0, 0, 'G', 0, 0, 0, 0, 0, 0, 0, 0);
    const __m256i nib = _mm256_set1_epi8(0x0F), up = _mm256_set1_epi8((char)0xDF);
    __m256i any_rst = _mm256_setzero_si256(), any_low = _mm256_setzero_si256();
    const uint64_t words = n / 32u;
    for (uint64_t w = 0; w < words; ++w) {
        const __m256i c = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(seq + w * 32u));
        const __m256i want = _mm256_shuffle_epi8(lut, _mm256_and_si256(c, nib));
        const __m256i acgt = _mm256_cmpeq_epi8(want, _mm256_and_si256(c, up)); // one of ACGTacgt
        const __m256i upper = _mm256_cmpeq_epi8(want, c);                      // one of ACGT
        const uint32_t m_acgt = (uint32_t)_mm256_movemask_epi8(acgt);
        const uint32_t m_up = (uint32_t)_mm256_movemask_epi8(upper);
        const uint32_t b2 = (uint32_t)_mm256_movemask_epi8(_mm256_slli_epi16(c, 5)); // bit 2 of every byte
        const uint32_t b1 = (uint32_t)_mm256_movemask_epi8(_mm256_slli_epi16(c, 6)); // bit 1
        lo[w] = (b1 ^ b2) & m_acgt;
        hi[w] = b2 & m_acgt;
        iv[w] = ~m_up;
        rs[w] = ~m_acgt;
        any_rst = _mm256_or_si256(any_rst, _mm256_xor_si256(acgt, _mm256_set1_epi8((char)0xFF)));
        any_low = _mm256_or_si256(any_low, _mm256_andnot_si256(upper, acgt));
    }
    return (_mm256_testz_si256(any_rst, any_rst) ? 0u : 1u) | (_mm256_testz_si256(any_low, any_low) ? 0u : 2u);
}

// AVX-512BW: byte compares and bit tests give the 64-bit masks directly -- no movemask, two words of every plane per iteration
__attribute__((target("avx512bw,avx512f"))) uint32_t pack_avx512(const uint8_t *seq, uint64_t n, uint32_t *lo, uint32_t *hi, uint32_t *iv, uint32_t *rs) {
    const __m512i lut = _mm512_broadcast_i32x4(_mm_setr_epi8(0x01, 'A', 0, 'C', 'T', 0, 0, 'G', 0, 0, 0, 0, 0, 0, 0, 0));
    const __m512i nib = _mm512_set1_epi8(0x0F), up = _mm512_set1_epi8((char)0xDF), bit1 = _mm512_set1_epi8(2), bit2 = _mm512_set1_epi8(4);
    uint64_t any_rst = 0, any_low = 0;
    const uint64_t pairs = n / 64u;
    // Non-temporal 8-byte stores (the write-combining buffers make whole lines of them) when all four planes are 8-byte
    // aligned: pinned planes a core has written the ordinary way stay dirty in its cache, and a device that reads them
    // next has every line fetched out of that cache -- two hops away if the core sits on the other socket.
    // IPCR_PACK_NT=0: ordinary stores.
    static const bool nt_on = !(getenv("IPCR_PACK_NT") && atoi(getenv("IPCR_PACK_NT")) == 0);
    const bool nt = nt_on && (((uintptr_t)lo | (uintptr_t)hi | (uintptr_t)iv | (uintptr_t)rs) & 7u) == 0;
    for (uint64_t w = 0; w < pairs; ++w) {
        const __m512i c = _mm512_loadu_si512(seq + w * 64u);
        const __m512i want = _mm512_shuffle_epi8(lut, _mm512_and_si512(c, nib));
        const uint64_t m_acgt = _mm512_cmpeq_epi8_mask(want, _mm512_and_si512(c, up));
        const uint64_t m_up = _mm512_cmpeq_epi8_mask(want, c);
        const uint64_t b2 = _mm512_test_epi8_mask(c, bit2), b1 = _mm512_test_epi8_mask(c, bit1);
        const uint64_t l = (b1 ^ b2) & m_acgt, h = b2 & m_acgt, v = ~m_up, r = ~m_acgt;
        if (nt) { // past the caches: the next reader of these words is the device
            _mm_stream_si64(reinterpret_cast<long long *>(lo + 2 * w), (long long)l);
            _mm_stream_si64(reinterpret_cast<long long *>(hi + 2 * w), (long long)h);
            _mm_stream_si64(reinterpret_cast<long long *>(iv + 2 * w), (long long)v);
            _mm_stream_si64(reinterpret_cast<long long *>(rs + 2 * w), (long long)r);
        } else {
            lo[2 * w] = (uint32_t)l; lo[2 * w + 1] = (uint32_t)(l >> 32);
            hi[2 * w] = (uint32_t)h; hi[2 * w + 1] = (uint32_t)(h >> 32);
            iv[2 * w] = (uint32_t)v; iv[2 * w + 1] = (uint32_t)(v >> 32);
            rs[2 * w] = (uint32_t)r; rs[2 * w + 1] = (uint32_t)(r >> 32);
        }
        any_rst |= r;
        any_low |= m_acgt & ~m_up;
    }
    if (nt) _mm_sfence();
    return (any_rst ? 1u : 0u) | (any_low ? 2u : 0u);
}

int simd_level() { // 2: AVX-512BW, 1: AVX2, 0: scalar (IPCR_PACK_SCALAR=1, IPCR_PACK_AVX512=0: tests)
    static const int v = [] {
        if (getenv("IPCR_PACK_SCALAR") && atoi(getenv("IPCR_PACK_SCALAR"))) return 0;
        const bool no512 = getenv("IPCR_PACK_AVX512") && !atoi(getenv("IPCR_PACK_AVX512"));
        if (!no512 && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512f")) return 2;
        return __builtin_cpu_supports("avx2") ? 1 : 0;
    }();
    return v;
}

bool have_avx2() { return simd_level() >= 1; }

} // namespace

namespace ipcr {

// Packs `len` bases of seq followed by padding up to `padded` bases (a multiple of 32; padding = inv 1, rst 0, code 0)
// into four arrays of padded / 32 words.  Returns bit 0: some byte lies outside ACGTacgt, bit 1: some byte is a
// lower-case acgt (only then does rst differ from inv).
uint32_t pack_linear(const uint8_t *seq, uint64_t len, uint64_t padded, uint32_t *lo, uint32_t *hi, uint32_t *iv, uint32_t *rs) {
    const uint64_t full = len & ~31ull; // whole words of real bases
    uint32_t flags = 0;
    if (full) {
        const int lvl = simd_level();
        uint64_t done = 0;
        if (lvl == 2 && full >= 64u) { done = full & ~63ull; flags |= pack_avx512(seq, done, lo, hi, iv, rs); }
        if (done < full) { // (an odd word, or no AVX-512)
            const uint64_t wd = done / 32u;
            flags |= lvl >= 1 ? pack_avx2(seq + done, full - done, lo + wd, hi + wd, iv + wd, rs + wd)
                              : pack_scalar(seq + done, full - done, lo + wd, hi + wd, iv + wd, rs + wd);
        }
    }
    uint64_t w = full / 32u;
    if (full < len) { // the word that holds the record's end
        uint8_t tail[32];
        memset(tail, 'a', sizeof tail);
        memcpy(tail, seq + full, (size_t)(len - full));
        flags |= pack_scalar(tail, 32, lo + w, hi + w, iv + w, rs + w) & 1u; // (the filler is not the record's lower case)
        for (uint64_t i = full; i < len; ++i)
            if (seq[i] == 'a' || seq[i] == 'c' || seq[i] == 'g' || seq[i] == 't') flags |= 2u;
        ++w;
    }
    for (; w < padded / 32u; ++w) { lo[w] = 0; hi[w] = 0; iv[w] = 0xFFFFFFFFu; rs[w] = 0; }
    _mm_sfence(); // the planes may be device memory behind the BAR (write-combining): everything is on its way before anyone is told
    return flags;
}

bool pack_linear_is_simd() { return have_avx2(); }

// ----------------------------------------------------------------------------------------------------------------------
// FASTA text -> bit planes on the host (the resident loader's fast way in: host.cpp: genome_add_fasta_hostpacked).
// A record's sequence region is `full` lines of W base bytes + lt terminator bytes ("\n" or "\r\n") and a shorter last line;
// this packs the bases of its 64-byte blocks [j0, j1) -- the line ends squeezed out of the classification masks with pext --
// to bit offset `bases before block j0` of the record's linear planes: bit i of 64-bit word w = base 64 w + i.  Case is
// folded (core/fasta/normalize.go:5-14), so inv = not one of ACGTacgt and the reset plane equals it.
// Whole words go to lo / hi (device memory through the BAR: written once, never read) with non-temporal stores and to iv
// (host memory) with ordinary ones; the partial words at the piece's ends come back in `edge` for the caller to compose.
// Returns false when the text is not what was assumed -- a line end where a base should be or the other way round, a blank
// or tab at a line's first or last base (the reference trims those) -- and the caller takes the device loader instead.
bool fasta_blocks_supported() {
    return simd_level() == 2 && __builtin_cpu_supports("bmi2");
}

__attribute__((target("avx512bw,avx512f,bmi2,popcnt")))
bool pack_fasta_blocks(const uint8_t *region, uint64_t area /* = full * (W + lt) bytes */, uint32_t W, uint32_t lt, const uint64_t *tab /* 4 x stride masks */,
                       uint64_t j0, uint64_t j1, uint64_t *lo, uint64_t *hi, uint64_t *iv, FastaEdge *edge, uint32_t *any_invalid) {
    const uint32_t stride = W + lt;
    const __m512i lut = _mm512_broadcast_i32x4(_mm_setr_epi8(0x01, 'A', 0, 'C', 'T', 0, 0, 'G', 0, 0, 0, 0, 0, 0, 0, 0));
    const __m512i nib = _mm512_set1_epi8(0x0F), up = _mm512_set1_epi8((char)0xDF), bit1 = _mm512_set1_epi8(2), bit2 = _mm512_set1_epi8(4);
    const __m512i c_nl = _mm512_set1_epi8('\n'), c_cr = _mm512_set1_epi8('\r'), c_sp = _mm512_set1_epi8(' ');
    const uint64_t *t_nl = tab, *t_cr = tab + stride, *t_first = tab + 2 * stride, *t_last = tab + 3 * stride;
    const uint64_t p0 = j0 * 64u;
    const uint64_t before = (p0 / stride) * W + std::min<uint64_t>(p0 % stride, W); // bases in front of block j0
    uint64_t w = before >> 6;          // word the accumulators are filling
    uint32_t fill = (uint32_t)(before & 63u);
    uint64_t al = 0, ah = 0, av = 0, inv_any = 0;
    bool first_word = fill != 0;       // the first word this piece completes is shared with the piece in front
    edge->n = 0;
    auto emit = [&](uint64_t l, uint64_t h, uint64_t v) {
        if (first_word) {
            edge->word[edge->n] = w; edge->val[edge->n][0] = l; edge->val[edge->n][1] = h; edge->val[edge->n][2] = v;
            ++edge->n;
            first_word = false;
        } else {
            _mm_stream_si64(reinterpret_cast<long long *>(lo + w), (long long)l);
            _mm_stream_si64(reinterpret_cast<long long *>(hi + w), (long long)h);
            iv[w] = v;
        }
        ++w;
    };
    uint32_t s = (uint32_t)(p0 % stride);
    const uint64_t jfull = std::min(j1, area / 64u); // blocks that lie wholly inside the text: plain loads
    for (uint64_t j = j0; j < j1; ++j) {
        const uint64_t off = j * 64u;
        const uint64_t live = j < jfull ? ~0ull : ((1ull << (area - off)) - 1ull);
        const __m512i c = j < jfull ? _mm512_loadu_si512(region + off) : _mm512_maskz_loadu_epi8((__mmask64)live, region + off);
        const uint64_t nl = _mm512_cmpeq_epi8_mask(c, c_nl) & live, cr = _mm512_cmpeq_epi8_mask(c, c_cr) & live;
        // blanks, tabs and every other control byte at a line's first or last base (the reference trims white space there): one
        // unsigned compare, line ends taken out
        const uint64_t ctl = _mm512_cmple_epu8_mask(c, c_sp) & live & ~(nl | cr);
        if (((nl ^ t_nl[s]) | (cr ^ t_cr[s])) & live || (ctl & (t_first[s] | t_last[s]))) return false;
        const uint64_t valid = live & ~(nl | cr);
        const __m512i want = _mm512_shuffle_epi8(lut, _mm512_and_si512(c, nib));
        const uint64_t m_acgt = _mm512_cmpeq_epi8_mask(want, _mm512_and_si512(c, up));
        const uint64_t b2 = _mm512_test_epi8_mask(c, bit2), b1 = _mm512_test_epi8_mask(c, bit1);
        const uint64_t l = _pext_u64((b1 ^ b2) & m_acgt, valid), h = _pext_u64(b2 & m_acgt, valid), v = _pext_u64(~m_acgt, valid);
        const uint32_t k = (uint32_t)_mm_popcnt_u64(valid);
        inv_any |= v;
        if (k) {
            al |= l << fill; ah |= h << fill; av |= v << fill;
            if (fill + k >= 64u) {
                emit(al, ah, av);
                const uint32_t used = 64u - fill; // bits of this block that went into the word just finished (1..64)
                al = used < 64u ? l >> used : 0; ah = used < 64u ? h >> used : 0; av = used < 64u ? v >> used : 0;
                fill = fill + k - 64u;
            } else
                fill += k;
        }
        s += 64u; // (no division in the loop unless a line is shorter than a block)
        if (s >= stride) s = stride >= 64u ? s - stride : s % stride;
    }
    // the unfinished word at the piece's end (it is also the piece's first when the piece never finished one)
    if (first_word ? fill > (uint32_t)(before & 63u) : fill > 0u) {
        edge->word[edge->n] = w; edge->val[edge->n][0] = al; edge->val[edge->n][1] = ah; edge->val[edge->n][2] = av;
        ++edge->n;
    }
    *any_invalid = inv_any ? 1u : 0u;
    _mm_sfence();
    return true;
}

} // namespace ipcr

extern "C" ipcr_status ipcr_pack_ascii(const uint8_t *seq, uint64_t len, uint64_t padded_bases, uint32_t *lo, uint32_t *hi,
                                       uint32_t *inv, uint32_t *rst, uint32_t *flags) {
    if ((!seq && len) || !lo || !hi || !inv || !rst) return ipcr_internal_fail(IPCR_ERR_INVALID, "ipcr_pack_ascii: null argument");
    if ((padded_bases & 31u) != 0 || padded_bases < len) return ipcr_internal_fail(IPCR_ERR_INVALID, "ipcr_pack_ascii: padded_bases must be a multiple of 32 and >= len");
    const uint32_t f = ipcr::pack_linear(seq, len, padded_bases, lo, hi, inv, rst);
    if (flags) *flags = f;
    return IPCR_OK;
}
