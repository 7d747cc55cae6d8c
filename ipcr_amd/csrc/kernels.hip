// kernels.hip -- hand-written gfx950 kernels of the ipcr primer matcher.
//
//  pack_kernel            ASCII record -> strand-major bit planes (tile_layout.h)
//  lcg_fill_kernel        reference benchDNA generator on the device (jump-ahead LCG)
//  filter_generic_quad_kernel  table-driven bit-sliced k-mismatch scan (any panel), a wave per row quad
//  verify_kernel          exact per-candidate verification -> ipcr_hit records
//  unpack_kernel          tiles -> ASCII (tests, amplicon extraction)
//  probe_kernel           oligo.BestHit over a batch of amplicons
//
// The panel-specialised filter is generated and compiled at panel-compile time (jit.cpp);
// it shares the queue format and the verifier below.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "device_types.h"
#include "tile_layout.h"
#define IPCR_VERIFY_BLOCKS 1024 // more blocks do not help (latency floor ~30 us), fewer are slower

// ------------------------------------------------------------------------------- pack
// ASCII -> strand-major bit planes.  Bit b of the output word (column, row r) is base r of the column's
// strand b, i.e. a 32 x 128 byte -> bit transpose per column.  One wavefront packs two columns (8 KB of
// sequence): it stages them in its own slice of LDS with coalesced 16-byte loads; then lane (h, q) owns
// row-quad q of column h and walks the 32 strands, four bases (one dword) at a time:
//  * the four bytes are classified together (SWAR): upper-case fold, a 4-entry byte LUT through
//    v_perm_b32 (index = bits 1..3 of the letter) gives the letter the byte would have to be, an exact
//    zero-byte test of the XOR says whether it is one of ACGT -- ~23 ops per dword instead of ~12 per byte;
//  * the per-byte flags are shifted into SWAR accumulators (one bit per strand and row) and moved to the
//    four output words every 8 strands.
// The first version (lane = strand, four wave ballots per row) needed ~30 VALU ops per base and ran at
// 0.9 Tbases/s; no ballots, no atomics here either.
// Semantics: inv = byte is not an upper-case A/C/G/T (core/primer/iupac.go:62-67); rst = byte is outside
// ACGTacgt (core/engine/ac.go:16-30); bases past the record end are inv=1,rst=0 padding (staged as 'a').
__device__ __forceinline__ uint32_t swar_zero_bytes(uint32_t x) { // 0x80 in every byte of x that is zero, exactly
    const uint32_t t = (x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
    return ~(t | x) & 0x80808080u;
}

template <bool ALIGNED>
__device__ __forceinline__ void pack_pair(const uint8_t *__restrict__ seq, uint64_t len, uint64_t col0, uint64_t ncol,
                                          uint64_t pairidx, uint32_t lane, uint32_t *mine,
                                          uint32_t *__restrict__ planes, uint32_t *__restrict__ rst,
                                          uint32_t *__restrict__ rec_flags) {
    const uint64_t base = pairidx * 2u * IPCR_COLUMN_BASES; // record-local first base of my column pair
#pragma unroll
    for (uint32_t it = 0; it < 8u; ++it) {
        const uint32_t idx = it * 64u + lane; // 16-byte piece of the 8 KB
        const uint64_t p0 = base + (uint64_t)idx * 16u;
        uint4 v;
        if (p0 + 16u <= len) {
            if (ALIGNED) v = *reinterpret_cast<const uint4 *>(seq + p0);
            else __builtin_memcpy(&v, seq + p0, 16); // one global_load_dwordx4 at a byte address (records of a batch start anywhere)
        } else {
            uint32_t w[4] = {0x61616161u, 0x61616161u, 0x61616161u, 0x61616161u}; // 'a': inv = 1, rst = 0, code 0
            for (uint32_t t = 0; t < 16u; ++t)
                if (p0 + t < len) w[t >> 2] = (w[t >> 2] & ~(0xFFu << ((t & 3u) * 8u))) | ((uint32_t)seq[p0 + t] << ((t & 3u) * 8u));
            v = make_uint4(w[0], w[1], w[2], w[3]);
        }
        *reinterpret_cast<uint4 *>(mine + idx * 4u) = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // my LDS slice: written and read by this wave only
    const uint32_t h = lane >> 5, q = lane & 31u;
    uint32_t olo[4] = {0, 0, 0, 0}, ohi[4] = {0, 0, 0, 0}, oiv[4] = {0, 0, 0, 0}, ors[4] = {0, 0, 0, 0};
    for (uint32_t g = 0; g < 4u; ++g) {
        uint32_t alo = 0, ahi = 0, aiv = 0, ars = 0;
#pragma unroll
        for (int t = 7; t >= 0; --t) { // strand g*8+7 first: it ends up in bit 7 of its byte lane
            const uint32_t w = mine[(h * 32u + g * 8u + (uint32_t)t) * 32u + q];
            const uint32_t u = w & 0xDFDFDFDFu;
            const uint32_t idx = (u >> 1) & 0x07070707u;                  // A 0, C 1, T 2, G 3 (4..7: no letter of ours)
            const uint32_t e = __builtin_amdgcn_perm(0u, 0x47544341u, idx); // the letter that index stands for
            const uint32_t acgt = swar_zero_bytes(e ^ u) >> 7;            // 1 per byte that is one of ACGTacgt
            const uint32_t lower = (w >> 5) & 0x01010101u;
            const uint32_t c = (idx ^ (idx >> 1)) & 0x03030303u;           // A 0, C 1, G 2, T 3
            alo = (alo << 1) | (c & acgt);
            ahi = (ahi << 1) | ((c >> 1) & acgt);
            aiv = (aiv << 1) | ((acgt & ~lower) ^ 0x01010101u);
            ars = (ars << 1) | (acgt ^ 0x01010101u);
        }
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) {
            olo[k] |= ((alo >> (8u * k)) & 0xFFu) << (8u * g);
            ohi[k] |= ((ahi >> (8u * k)) & 0xFFu) << (8u * g);
            oiv[k] |= ((aiv >> (8u * k)) & 0xFFu) << (8u * g);
            ors[k] |= ((ars >> (8u * k)) & 0xFFu) << (8u * g);
        }
    }
    const uint64_t col = col0 + pairidx * 2u + h;
    if (pairidx * 2u + h < ncol) {
        const uint64_t block = col >> 6;
        const uint32_t ln = (uint32_t)(col & 63u);
        *reinterpret_cast<uint4 *>(planes + ipcr_plane_word(block, q * 4u, 0, ln)) = make_uint4(olo[0], olo[1], olo[2], olo[3]);
        *reinterpret_cast<uint4 *>(planes + ipcr_plane_word(block, q * 4u, 1, ln)) = make_uint4(ohi[0], ohi[1], ohi[2], ohi[3]);
        *reinterpret_cast<uint4 *>(planes + ipcr_plane_word(block, q * 4u, 2, ln)) = make_uint4(oiv[0], oiv[1], oiv[2], oiv[3]);
        *reinterpret_cast<uint4 *>(rst + ipcr_rst_word(block, q * 4u, ln)) = make_uint4(ors[0], ors[1], ors[2], ors[3]);
    }
    const bool saw_rst = (ors[0] | ors[1] | ors[2] | ors[3]) != 0u;
    // bit 0 is the only bit a record's flag word ever gets: a plain store (idempotent, also right when the word lives
    // in pinned host memory, where the chunk path keeps it -- no copy operation brings it back)
    if (__ballot(saw_rst) != 0ull && lane == 0u) __hip_atomic_store(rec_flags, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(256) void pack_kernel(const uint8_t *__restrict__ seq, uint64_t len,
                                                   uint64_t col0, uint64_t ncol,
                                                   uint32_t *__restrict__ planes,
                                                   uint32_t *__restrict__ rst,
                                                   uint32_t *__restrict__ rec_flags,
                                                   uint64_t *__restrict__ rec_start_out,
                                                   uint64_t *__restrict__ rec_len_out) {
    __shared__ uint32_t s_in[4][2048]; // per wave: 2 columns x 32 strands x 32 dwords
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    if (rec_start_out && blockIdx.x == 0 && threadIdx.x == 0) { // chunk path: the one record's table entries, no copy operation
        rec_start_out[0] = col0 * IPCR_COLUMN_BASES;
        rec_len_out[0] = len;
    }
    const uint64_t pairidx = (uint64_t)blockIdx.x * 4u + wv;
    if (pairidx * 2u >= ncol) return; // waves are independent (no workgroup barrier)
    pack_pair<true>(seq, len, col0, ncol, pairidx, lane, s_in[wv], planes, rst, rec_flags);
}

// many records in one launch (a nested-PCR batch or a fragmented assembly has thousands of short records, one launch
// each costs ~17 us); the records may start at any byte of `base`:
// wave = one column pair of one record, found by binary search in the prefix of the records' pair counts
__global__ __launch_bounds__(256) void pack_batch_kernel(const uint8_t *__restrict__ base, const ipcr_pack_rec *__restrict__ recs,
                                                         const uint32_t *__restrict__ pair_prefix, uint32_t nrec,
                                                         uint32_t *__restrict__ planes, uint32_t *__restrict__ rst,
                                                         uint32_t *__restrict__ rec_flags) {
    __shared__ uint32_t s_in[4][2048];
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint64_t gp = (uint64_t)blockIdx.x * 4u + wv;
    if (gp >= pair_prefix[nrec]) return;
    uint32_t lo = 0, hi = nrec; // record whose pair range holds gp
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (pair_prefix[mid] <= gp) lo = mid; else hi = mid;
    }
    const ipcr_pack_rec r = recs[lo];
    pack_pair<false>(base + r.src_off, r.len, r.col0, r.ncol, gp - pair_prefix[lo], lane, s_in[wv], planes, rst, rec_flags + r.flag_idx);
}

// fill columns [col_begin, col_end) with padding (inv=1, everything else 0)
// ------------------------------------------------------------------------------- tiles from host-packed planes
// Linear bit planes (hostpack.cpp: bit i of word w = base 32 w + i) -> strand-major tiles.  A column is 32 strands of
// 128 bases = 4 linear words each; rows 32 j .. 32 j + 31 of its tile words are the 32 x 32 bit transpose of the strands'
// j-th words.  One half wave per (column, plane): lane s loads the four words of strand s (one 16-byte load, 512
// contiguous bytes per half wave), four cross-lane transposes (butterflies over ds_swizzle, as the seed-index kernel's
// k-mer transposes), lane i stores rows i, 32 + i, 64 + i, 96 + i.  Plane 3 = rst; without a fourth linear plane
// (no lower-case acgt in the chunk) rst is inv.  Without a third one either (lin_iv null: every byte of the slice is one of
// ACGT -- most slices of most genomes) the invalid bits are made here: 0.25 bytes per base cross the link instead of 0.375.
template <int D> __device__ __forceinline__ uint32_t tl_stage(uint32_t x, uint32_t lane) {
    constexpr uint32_t m0 = D == 16 ? 0x0000FFFFu : D == 8 ? 0x00FF00FFu : D == 4 ? 0x0F0F0F0Fu : D == 2 ? 0x33333333u : 0x55555555u;
    const uint32_t p = (uint32_t)__builtin_amdgcn_ds_swizzle((int)x, (D << 10) | 0x1F);
    const bool up = (lane & D) != 0;
    const uint32_t r = __builtin_amdgcn_alignbit(p, p, up ? D : 32 - D);
    return __builtin_amdgcn_bitop3_b32(x, r, up ? ~m0 : m0, 0xE4); // (x & m) | (r & ~m)
}
__device__ __forceinline__ uint32_t tl_transpose(uint32_t x, uint32_t lane) {
    x = tl_stage<16>(x, lane);
    x = tl_stage<8>(x, lane);
    x = tl_stage<4>(x, lane);
    x = tl_stage<2>(x, lane);
    x = tl_stage<1>(x, lane);
    return x;
}
__global__ __launch_bounds__(256) void tiles_from_linear_kernel(const uint4 *__restrict__ lin_lo, const uint4 *__restrict__ lin_hi,
                                                               const uint4 *__restrict__ lin_iv, const uint4 *__restrict__ lin_rs,
                                                               const uint32_t *__restrict__ iv_cols, // null, or one bit per column of the launch: its invalid bits are in lin_iv (else: made here)
                                                               uint64_t rec_col0, uint64_t col0, uint64_t ncol, uint64_t len,
                                                               uint32_t *__restrict__ planes, uint32_t *__restrict__ rst,
                                                               uint64_t *__restrict__ rec_start_out, uint64_t *__restrict__ rec_len_out) {
    if (rec_start_out && blockIdx.x == 0 && threadIdx.x == 0) { // chunk path: the one record's table entries, no copy operation
        rec_start_out[0] = rec_col0 * IPCR_COLUMN_BASES;
        rec_len_out[0] = len;
    }
    const uint32_t lane = threadIdx.x & 31u;
    const uint64_t item = ((uint64_t)blockIdx.x * 256u + threadIdx.x) >> 5; // (column, plane)
    const bool live = item < ncol * 4u;
    const uint64_t c = live ? item >> 2 : 0u; // column of this launch's range
    const uint32_t plane = (uint32_t)item & 3u;
    const uint4 *src = plane == 0u ? lin_lo : plane == 1u ? lin_hi : (plane == 2u || !lin_rs) ? lin_iv : lin_rs;
    if (plane >= 2u && src == lin_iv && iv_cols && live && !((iv_cols[c >> 5] >> (uint32_t)(c & 31u)) & 1u)) src = nullptr; // a column of ACGT only
    uint4 v;
    if (src) v = src[c * 32u + lane];
    else { // no invalid-bit plane came over the link: the slice holds ACGT only, what is invalid is what lies behind the record's end
        const uint64_t b0 = ((col0 - rec_col0 + c) * 32u + lane) * 128u; // first base of this lane's strand
        uint32_t w[4];
#pragma unroll
        for (uint32_t j = 0; j < 4u; ++j) {
            const uint64_t b = b0 + 32u * j;
            w[j] = b >= len ? 0xFFFFFFFFu : (b + 32u <= len ? 0u : ~((1u << (uint32_t)(len - b)) - 1u));
        }
        v = make_uint4(w[0], w[1], w[2], w[3]);
    }
    const uint32_t t0 = tl_transpose(v.x, lane), t1 = tl_transpose(v.y, lane), t2 = tl_transpose(v.z, lane), t3 = tl_transpose(v.w, lane);
    if (!live) return;
    const uint64_t col = col0 + c, block = col >> 6;
    const uint32_t ln = (uint32_t)(col & 63u);
    if (plane < 3u) {
        planes[ipcr_plane_word(block, lane, plane, ln)] = t0;
        planes[ipcr_plane_word(block, 32u + lane, plane, ln)] = t1;
        planes[ipcr_plane_word(block, 64u + lane, plane, ln)] = t2;
        planes[ipcr_plane_word(block, 96u + lane, plane, ln)] = t3;
    } else {
        rst[ipcr_rst_word(block, lane, ln)] = t0;
        rst[ipcr_rst_word(block, 32u + lane, ln)] = t1;
        rst[ipcr_rst_word(block, 64u + lane, ln)] = t2;
        rst[ipcr_rst_word(block, 96u + lane, ln)] = t3;
    }
}

__global__ void fill_pad_kernel(uint32_t *__restrict__ planes, uint32_t *__restrict__ rst,
                                uint64_t col_begin, uint64_t col_end) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; // (column, row)
    const uint64_t col = col_begin + (i >> 7);
    const uint32_t row = (uint32_t)(i & 127u);
    if (col >= col_end) return;
    const uint64_t block = col >> 6;
    const uint32_t ln = (uint32_t)(col & 63u);
    planes[ipcr_plane_word(block, row, 0, ln)] = 0u;
    planes[ipcr_plane_word(block, row, 1, ln)] = 0u;
    planes[ipcr_plane_word(block, row, 2, ln)] = 0xFFFFFFFFu;
    rst[ipcr_rst_word(block, row, ln)] = 0u;
}

// -------------------------------------------------------------------------- reset bytes per window
// Does the run of bases [a, b) (padded genome coordinates) hold a reset byte?  One workgroup per window.  The strands that lie
// wholly inside the window are tested a tile word at a time -- a word holds one row of 32 strands: the bits of the strands
// inside, every row -- and the window's first and last strand bit by bit over their rows inside.
// (ipcr_scan_genome_chunked: the reference's forceFallback rule is per ForEachCompiledProduct call, i.e. per window.)
__global__ __launch_bounds__(256) void window_reset_kernel(const uint32_t *__restrict__ rst, const uint64_t *__restrict__ win, // [2 i] = a, [2 i + 1] = b
                                                          uint32_t nwin, uint32_t *__restrict__ flags) {
    const uint32_t w = blockIdx.x;
    if (w >= nwin) return;
    const uint64_t a = win[2u * w], b = win[2u * w + 1u];
    uint32_t any = 0u;
    if (b > a) {
        const uint64_t sa = a >> IPCR_TILE_LOG_N, sb = (b - 1u) >> IPCR_TILE_LOG_N; // first and last strand (inclusive)
        const uint32_t ra = (uint32_t)(a & (IPCR_TILE_N - 1u)), rb = (uint32_t)((b - 1u) & (IPCR_TILE_N - 1u));
        // the boundary strands, row by row
        for (uint32_t t = threadIdx.x; t < 2u * IPCR_TILE_N; t += 256u) {
            const bool last = t >= IPCR_TILE_N;
            const uint32_t r = t & (IPCR_TILE_N - 1u);
            const uint64_t s = last ? sb : sa;
            if (last && sb == sa) continue; // one strand: the first pass takes it
            const uint32_t lo = (s == sa) ? ra : 0u, hi = (s == sb) ? rb : IPCR_TILE_N - 1u;
            if (r < lo || r > hi) continue;
            const uint64_t col = s >> 5;
            any |= (rst[ipcr_rst_word(col >> 6, r, (uint32_t)(col & 63u))] >> (uint32_t)(s & 31u)) & 1u;
        }
        // the strands between them, a word (32 strands of one row) at a time
        if (sb > sa + 1u) {
            const uint64_t s0 = sa + 1u, s1 = sb - 1u; // inclusive
            const uint64_t c0 = s0 >> 5, c1 = s1 >> 5;
            const uint64_t nwords = (c1 - c0 + 1u) * IPCR_TILE_N;
            for (uint64_t i = threadIdx.x; i < nwords; i += 256u) {
                const uint64_t col = c0 + (i >> IPCR_TILE_LOG_N);
                const uint32_t r = (uint32_t)(i & (IPCR_TILE_N - 1u));
                uint32_t m = 0xFFFFFFFFu;
                if (col == c0) m &= 0xFFFFFFFFu << (uint32_t)(s0 & 31u);
                if (col == c1) m &= 0xFFFFFFFFu >> (31u - (uint32_t)(s1 & 31u));
                any |= (rst[ipcr_rst_word(col >> 6, r, (uint32_t)(col & 63u))] & m) ? 1u : 0u;
            }
        }
    }
    any = __syncthreads_or((int)any) ? 1u : 0u;
    if (threadIdx.x == 0u) flags[w] = any;
}

// -------------------------------------------------------------------------- LCG genome
// benchDNA (core/engine/performance_benchmark_test.go:67-76): x = x*1664525 + 1013904223,
// base = "ACGT"[(x>>30)&3].  Each thread jumps ahead to its 64-base run by composing the
// affine map with itself (O(log n)), so the sequence is bit-identical to the serial loop.
__global__ void lcg_fill_kernel(uint8_t *__restrict__ out, uint64_t n, uint32_t seed, uint64_t offset) {
    const uint64_t run = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t start = run * 64u;
    if (start >= n) return;
    uint32_t a = 1664525u, c = 1013904223u; // map for one step
    uint32_t ja = 1u, jc = 0u;              // identity
    uint64_t e = start + offset; // steps already taken by the stream
    while (e) {
        if (e & 1u) { jc = a * jc + c; ja = a * ja; }
        c = a * c + c;
        a = a * a;
        e >>= 1;
    }
    uint32_t x = ja * seed + jc; // state after `start` steps
    const uint64_t end = (start + 64u < n) ? start + 64u : n;
    for (uint64_t i = start; i < end; ++i) {
        x = x * 1664525u + 1013904223u;
        out[i] = (uint8_t)"ACGT"[(x >> 30) & 3u];
    }
}

// ------------------------------------------------------- table-driven filter, a row quad per wave
// Exact bit-sliced k-mismatch count for every pattern of the panel, with the pattern as data (any panel, no hiprtc).  Each lane
// owns one column (a word = 32 strands of one row); pattern position j compares row r + j.  Mismatches outside the protected
// window feed a thermometer counter u[t] = "count >= t + 1"; a mismatch inside it goes straight into the top level (count >=
// K1 = max_mm + 1: out), so there is no separate dead mask.  Survivors go to the candidate queue, a word per entry.
// A wave owns the FOUR start rows of one row quad and walks down quad by quad: one 16-byte load per plane and lane brings four
// rows (the layout's unit), every row is decoded once and then steps 4 starts x PB patterns = 16 independent counters, 8
// vector instructions a step (four and-ors for the mismatch mask, one select + K1 and-ors for the counters at k = 2, every
// and-or a two-cycle v_bitop3_b32).  The pattern's masks live in scalar registers (a word = four positions, read once per
// quad), every test on them is scalar work, and the five masks of a position are made once per offset d = row - start, not
// once per step (the scalar unit bound the walk before that).  K1 is a compile-time constant, PB follows from what 4 x PB x
// K1 counters leave of the registers.  Rows >= 128 continue in the next strand: the same words one bit down, with bit 0 of the
// next column's word on top.  Workgroups are dealt to the XCDs round-robin by the dispatcher: the index is turned so that
// every XCD walks a contiguous eighth of the blocks and the rows a wave shares with the waves below it (it reads ~4 quads past
// its own) come out of that XCD's L2.
// Round 4, before: a wave per ROW (one dword per lane and row out of every 16 bytes, four patterns per walk, run-time counter
// depth): 6.39 ms per 3 Gb for C2's four patterns, 4.1 x the algorithmic bytes from HBM, 2.70 G vector instructions; this
// form 1.08 ms, 1.0005 x the algorithmic bytes, 0.79 G vector + 0.40 G scalar instructions (profiles/r04_c2g_*; DESIGN 4.4
// has the steps in between).
__device__ __forceinline__ uint4 fetch_quad(const uint4 *__restrict__ planes4, uint64_t block, uint32_t quad, uint32_t plane, uint32_t lane) {
    if (quad < 32u) return planes4[((block * 32u + quad) * 3u + plane) * 64u + lane];
    const uint32_t q2 = quad - 32u; // the next strand: the same words one bit down, the next column's bit 0 on top
    const uint4 own = planes4[((block * 32u + q2) * 3u + plane) * 64u + lane];
    const uint4 nxt = (lane < 63u) ? planes4[((block * 32u + q2) * 3u + plane) * 64u + lane + 1u]
                                   : planes4[(((block + 1u) * 32u + q2) * 3u + plane) * 64u];
    return make_uint4((own.x >> 1) | (nxt.x << 31), (own.y >> 1) | (nxt.y << 31), (own.z >> 1) | (nxt.z << 31), (own.w >> 1) | (nxt.w << 31));
}

// (a & b) | c as ONE v_bitop3_b32 (truth table 0xEA): it issues in two cycles where v_and_or_b32 -- the compiler's choice for the
// same expression -- and v_or3_b32 take four (tools/ubench/valu_rates.hip)
__device__ __forceinline__ uint32_t and_or(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0xEA); }

template <int K1, int PB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(K1 <= 3 ? 5 : 3))) void filter_generic_quad_kernel(const uint32_t *__restrict__ planes,
                                                                  uint64_t block0, uint64_t nblocks, // blocks [block0, block0 + nblocks)
                                                                  const ipcr_dev_pattern *__restrict__ pats, uint32_t npat,
                                                                  const uint32_t *__restrict__ sel, // pattern subset or null
                                                                  ipcr_queue_entry *__restrict__ queue, uint64_t qcap,
                                                                  unsigned long long *__restrict__ qcount) {
    const uint4 *__restrict__ planes4 = reinterpret_cast<const uint4 *>(planes);
    const uint32_t lane = threadIdx.x & 63u;
    // the grid is 8 * nblocks workgroups (8 per block: 32 row quads / 4 waves); workgroup w runs on XCD w % 8
    const uint64_t wg = (uint64_t)(blockIdx.x & 7u) * nblocks + (blockIdx.x >> 3);
    const uint64_t tile = wg * 4u + (threadIdx.x >> 6); // (block, row quad)
    if ((tile >> 5) >= nblocks) return;
    const uint64_t block = block0 + (tile >> 5);
    const uint32_t rq = (uint32_t)(tile & 31u);

    for (uint32_t qi0 = 0; qi0 < npat; qi0 += PB) {
        uint32_t u[4][PB][K1]; // [start row of the quad][pattern][t]: count >= t + 1
        uint32_t qid[PB], L[PB], prevw[PB], curw[PB]; // (wave-uniform: the masks of the pattern positions 4 q - 4 ... 4 q + 3, a byte each)
        const uint32_t *mw[PB];
        uint32_t Lmax = 0;
#pragma unroll
        for (int b = 0; b < PB; ++b) {
            const bool on = qi0 + (uint32_t)b < npat; // wave-uniform
            qid[b] = on ? (sel ? sel[qi0 + (uint32_t)b] : qi0 + (uint32_t)b) : 0u;
            mw[b] = reinterpret_cast<const uint32_t *>(pats[qid[b]].mask);
            L[b] = on ? (uint32_t)__builtin_amdgcn_readfirstlane((int)pats[qid[b]].len) : 0u;
            prevw[b] = 0u;
            Lmax = L[b] > Lmax ? L[b] : Lmax;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int t = 0; t < K1; ++t) u[s][b][t] = 0u;
                if (!on) u[s][b][K1 - 1] = 0xFFFFFFFFu;
            }
        }
        const uint32_t nq = (Lmax + 2u) / 4u + 1u; // start row 3 ends at row Lmax + 2 of the walk
        for (uint32_t q = 0; q < nq; ++q) {
            const uint4 lo4 = fetch_quad(planes4, block, rq + q, 0, lane);
            const uint4 hi4 = fetch_quad(planes4, block, rq + q, 1, lane);
            const uint4 iv4 = fetch_quad(planes4, block, rq + q, 2, lane);
            const uint32_t los[4] = {lo4.x, lo4.y, lo4.z, lo4.w}, his[4] = {hi4.x, hi4.y, hi4.z, hi4.w}, ivs[4] = {iv4.x, iv4.y, iv4.z, iv4.w};
#pragma unroll
            for (int b = 0; b < PB; ++b) // (scalar registers: every mask test below is scalar work, the lanes only AND and OR)
                curw[b] = q * 4u < L[b] ? ~(uint32_t)__builtin_amdgcn_readfirstlane((int)mw[b][q]) : 0u; // complemented: bit set = base NOT allowed / not protected
            uint32_t isA[4], isC[4], isG[4], isT[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                isA[i] = ~los[i] & ~his[i]; isC[i] = los[i] & ~his[i]; isG[i] = ~los[i] & his[i]; isT[i] = los[i] & his[i];
            }
            // row i of the quad is position 4 q + (i - s) of the pattern for start row s: the seven offsets d = i - s each name ONE
            // mask byte of a pattern (this quad's word or the one before), so its five scalar masks are made once per offset and
            // serve the 4 - |d| (row, start) pairs that share it (made per step, the scalar instructions bound the walk); the
            // count does not depend on the order its positions arrive in
#pragma unroll
            for (int b = 0; b < PB; ++b) {
#pragma unroll
                for (int d = -3; d <= 3; ++d) {
                    const uint32_t jj = q * 4u + (uint32_t)d; // (wraps below zero in the first quad: not < L)
                    if (jj < L[b]) {                          // wave-uniform
                        const int sh = 8 * (d & 3);
                        const uint32_t w = d >= 0 ? curw[b] : prevw[b];
                        const uint32_t nA = (uint32_t)((int32_t)(w << (31 - sh)) >> 31), nC = (uint32_t)((int32_t)(w << (30 - sh)) >> 31),
                                       nG = (uint32_t)((int32_t)(w << (29 - sh)) >> 31), nT = (uint32_t)((int32_t)(w << (28 - sh)) >> 31);
                        const uint32_t prot = ~(uint32_t)((int32_t)(w << (27 - sh)) >> 31);
                        // inv | (isA & nA) | (isC & nC) | (isG & nG) | (isT & nT): a chain of four and-ors per row, each mask straight from
                        // its scalar register (left to itself the compiler makes four ANDs and two three-way ORs); the rows' chains are
                        // written side by side: they do not depend on each other
                        const int I0 = (d > 0 ? d : 0), I1 = (d < 0 ? 3 + d : 3);
                        uint32_t mm[4];
#pragma unroll
                        for (int i = I0; i <= I1; ++i) mm[i] = and_or(isA[i], nA, ivs[i]);
#pragma unroll
                        for (int i = I0; i <= I1; ++i) mm[i] = and_or(isC[i], nC, mm[i]);
#pragma unroll
                        for (int i = I0; i <= I1; ++i) mm[i] = and_or(isG[i], nG, mm[i]);
#pragma unroll
                        for (int i = I0; i <= I1; ++i) mm[i] = and_or(isT[i], nT, mm[i]);
                        // a mismatch at a protected position goes into the top level whatever the count below it (the lower levels take
                        // it too: the position is out either way) -- no branch, K1 + 1 operations
#pragma unroll
                        for (int i = I0; i <= I1; ++i) {
                            const int s = i - d;
                            if (K1 > 1) {
                                u[s][b][K1 - 1] = and_or(u[s][b][K1 > 1 ? K1 - 2 : 0] | prot, mm[i], u[s][b][K1 - 1]);
#pragma unroll
                                for (int t = K1 - 2; t >= 1; --t) u[s][b][t] = and_or(u[s][b][t - 1], mm[i], u[s][b][t]);
                            }
                            u[s][b][0] |= mm[i];
                        }
                    }
                }
            }
#pragma unroll
            for (int b = 0; b < PB; ++b) prevw[b] = curw[b];
            uint32_t gone = 0xFFFFFFFFu; // positions no start row and no pattern of the pass can still match
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int b = 0; b < PB; ++b) gone &= u[s][b][K1 - 1];
            if (__ballot(gone != 0xFFFFFFFFu) == 0ull) break;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int b = 0; b < PB; ++b) {
                const uint32_t alive = ~u[s][b][K1 - 1];
                if (alive) { // one queue entry per surviving word: 32 strands of this row
                    const uint32_t shard = (uint32_t)block & (IPCR_QUEUE_SHARDS - 1u);
                    const unsigned long long idx = atomicAdd(qcount + shard * IPCR_QUEUE_COUNTER_STRIDE, 1ull);
                    if (idx < qcap) { // qcap = capacity of one shard's segment
                        ipcr_queue_entry e;
                        e.key = ((uint64_t)qid[b] << 48) | ipcr_join_pos(block * 64u + lane, 0, rq * 4u + (uint32_t)s);
                        e.bits = alive;
                        e.pad = 0;
                        queue[(uint64_t)shard * qcap + idx] = e;
                    }
                }
            }
    }
}

// ------------------------------------------------------------------------------ verify
// verifyAt (core/engine/ac.go:186-213) / the inner loop of FindMatches
// (core/primer/match.go:67-84) for one candidate per thread, straight from the tiles.
__device__ __forceinline__ uint32_t base_bits(const uint32_t *__restrict__ planes, uint64_t P) {
    uint64_t col; uint32_t bit, row;
    ipcr_split_pos(P, &col, &bit, &row);
    const uint64_t block = col >> 6;
    const uint32_t ln = (uint32_t)(col & 63u);
    const uint64_t w = ipcr_plane_word(block, row, 0, ln);
    const uint32_t lo = (planes[w] >> bit) & 1u;
    const uint32_t hi = (planes[w + 256u] >> bit) & 1u;  // next plane: +64 lanes * 4
    const uint32_t inv = (planes[w + 512u] >> bit) & 1u;
    return lo | (hi << 1) | (inv << 2);
}

__device__ __forceinline__ uint32_t rst_bit(const uint32_t *__restrict__ rst, uint64_t P) {
    uint64_t col; uint32_t bit, row;
    ipcr_split_pos(P, &col, &bit, &row);
    return (rst[ipcr_rst_word(col >> 6, row, (uint32_t)(col & 63u))] >> bit) & 1u;
}

__global__ __launch_bounds__(256) void verify_kernel(const uint32_t *__restrict__ planes,
                                                     const uint32_t *__restrict__ rst,
                                                     const ipcr_dev_pattern *__restrict__ pats, uint32_t max_mm,
                                                     const uint64_t *__restrict__ rec_start,
                                                     const uint64_t *__restrict__ rec_len, uint32_t nrec,
                                                     uint32_t check_rst,
                                                     const ipcr_queue_entry *__restrict__ queue, uint64_t qcap,
                                                     const unsigned long long *__restrict__ qcount,
                                                     ipcr_hit_rec *__restrict__ hits, uint64_t hcap,
                                                     unsigned long long *__restrict__ hcount,
                                                     unsigned long long *__restrict__ ccount,
                                                     unsigned long long *__restrict__ next_counters,
                                                     unsigned long long *__restrict__ next_qcount, uint32_t single) {
    // single != 0: every queue entry holds ONE candidate (the seed index files a window at a time, bits == 1): one thread
    // per entry instead of one per (entry, strand bit) -- 25 000 threads with work instead of 800 000 of which one in 32 has.
    // The candidate queue is cut into IPCR_QUEUE_SHARDS segments with a counter each (a single
    // counter serialises the filters' pushes at ~12 ns apiece, MI355X_MICROARCH.md "dequeue").
    // qcount = this scan's shard counters, qcap = capacity of one segment.
    // The counters alternate between two sets; this launch clears the set the NEXT scan will use
    // (its values were copied to the host before this scan was enqueued), so no memset launch.
    __shared__ uint32_t s_pref[IPCR_QUEUE_SHARDS]; // inclusive prefix of the (clamped) shard counts
    __shared__ uint32_t s_hits, s_cands;
    __shared__ unsigned long long s_base;
    {
        const uint32_t t = threadIdx.x; // blockDim.x == IPCR_QUEUE_SHARDS == 256
        const unsigned long long raw = qcount[t * IPCR_QUEUE_COUNTER_STRIDE];
        s_pref[t] = (uint32_t)(raw > qcap ? qcap : raw);
        if (blockIdx.x == 0) {
            next_qcount[t * IPCR_QUEUE_COUNTER_STRIDE] = 0ull;
            if (t < 4u) next_counters[t] = 0ull;
            atomicAdd(ccount - 2, raw);           // header[0]: total queue entries pushed
            atomicMax(ccount + 1, raw);           // header[3]: fullest shard (overflow check on the host)
        }
        __syncthreads();
        for (uint32_t d = 1; d < IPCR_QUEUE_SHARDS; d <<= 1) { // Hillis-Steele scan
            const uint32_t v = (t >= d) ? s_pref[t - d] : 0u;
            __syncthreads();
            s_pref[t] += v;
            __syncthreads();
        }
    }
    // hit slots and the candidate count are aggregated per workgroup in LDS: thousands of
    // same-address global atomics would serialise (see above)
    const uint32_t per = single ? 0u : 5u; // log2 of the threads per entry
    unsigned long long n = (unsigned long long)s_pref[IPCR_QUEUE_SHARDS - 1u] << per; // one thread per (entry, strand bit)
    for (uint64_t base = (uint64_t)blockIdx.x * blockDim.x; base < n; base += (uint64_t)gridDim.x * blockDim.x) {
        if (threadIdx.x == 0) { s_hits = 0; s_cands = 0; }
        __syncthreads();
        const uint64_t i = base + threadIdx.x;
        bool hit = false;
        ipcr_hit_rec h;
        if (i < n) {
            const uint32_t eidx = (uint32_t)(i >> per);
            uint32_t slo = 0, shi = IPCR_QUEUE_SHARDS - 1u; // first shard whose inclusive prefix exceeds eidx
            while (slo < shi) {
                const uint32_t mid = (slo + shi) >> 1;
                if (s_pref[mid] > eidx) shi = mid; else slo = mid + 1u;
            }
            const uint32_t within = eidx - (slo ? s_pref[slo - 1u] : 0u);
            const ipcr_queue_entry ent = queue[(uint64_t)slo * qcap + within];
            const uint32_t bit = single ? (ent.bits ? (uint32_t)__builtin_ctz(ent.bits) : 0u) : (uint32_t)(i & 31u);
            if (single || bit == 0u) atomicAdd(&s_cands, (uint32_t)__builtin_popcount(ent.bits));
            if ((ent.bits >> bit) & 1u) {
                const uint32_t q = (uint32_t)(ent.key >> 48);
                const uint64_t P = (ent.key & 0xFFFFFFFFFFFFull) + ((uint64_t)bit << IPCR_TILE_LOG_N);
                // record lookup: last record with start <= P
                uint32_t lo = 0, hi = nrec;
                while (hi - lo > 1u) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (rec_start[mid] <= P) lo = mid; else hi = mid;
                }
                const uint64_t local = P - rec_start[lo];
                const ipcr_dev_pattern *pp = pats + q;
                const uint32_t L = pp->len;
                bool ok = local + L <= rec_len[lo]; // window must stay inside the record (ac.go:188-190)
                uint32_t mm = 0;
                uint64_t m0 = 0, m1 = 0;
                // 8 positions per round: their 24 tile words are loaded back to back (independent
                // addresses), then compared; the early exit is taken between rounds only
                for (uint32_t j0 = 0; j0 < L && ok; j0 += 8u) {
                    uint32_t g[8];
#pragma unroll
                    for (uint32_t t = 0; t < 8u; ++t) g[t] = (j0 + t < L) ? base_bits(planes, P + j0 + t) : 0u;
#pragma unroll
                    for (uint32_t t = 0; t < 8u; ++t) {
                        const uint32_t j = j0 + t;
                        if (j >= L) break;
                        const uint32_t onehot = (g[t] & 4u) ? 0u : (1u << (g[t] & 3u));
                        const uint32_t m = pp->mask[j];
                        if ((m & onehot) == 0u) {
                            if (m & 16u) ok = false;
                            ++mm;
                            if (j < 64u) m0 |= 1ull << j; else m1 |= 1ull << (j - 64u);
                        }
                    }
                    if (mm > max_mm) ok = false;
                }
                if (ok) {
                    uint32_t flag = 0;
                    if (check_rst && pp->seed_len)
                        for (uint32_t j = 0; j < pp->seed_len; ++j) flag |= rst_bit(rst, P + pp->seed_off + j);
                    hit = true;
                    h.pos = local;
                    h.record = lo;
                    h.pattern = pp->global_id | (flag << 31);
                    h.mm_mask[0] = m0;
                    h.mm_mask[1] = m1;
                }
            }
        }
        const uint32_t slot = hit ? atomicAdd(&s_hits, 1u) : 0u;
        __syncthreads();
        if (threadIdx.x == 0) {
            s_base = s_hits ? atomicAdd(hcount, (unsigned long long)s_hits) : 0ull;
            if (s_cands) atomicAdd(ccount, (unsigned long long)s_cands);
        }
        __syncthreads();
        if (hit && s_base + slot < hcap) hits[s_base + slot] = h;
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------ unpack
// tiles -> ASCII for [P0, P0+n): valid -> ACGT, lower-case acgt (inv, !rst) -> acgt, else 'N'
__global__ void unpack_kernel(const uint32_t *__restrict__ planes, const uint32_t *__restrict__ rst,
                              uint64_t P0, uint64_t n, uint8_t *__restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t g = base_bits(planes, P0 + i);
    uint8_t ch;
    if (!(g & 4u)) ch = (uint8_t)"ACGT"[g & 3u];
    else if (!rst_bit(rst, P0 + i)) ch = (uint8_t)"acgt"[g & 3u];
    else ch = 'N';
    out[i] = ch;
}

// gather amplicons of products: seg[i] = {P_start_a, len_a, P_start_b, len_b, out_offset}
__global__ void gather_amplicons_kernel(const uint32_t *__restrict__ planes, const uint32_t *__restrict__ rst,
                                        const ipcr_amp_seg *__restrict__ segs, uint8_t *__restrict__ out) {
    const ipcr_amp_seg s = segs[blockIdx.x];
    const uint64_t total = s.len_a + s.len_b;
    for (uint64_t i = threadIdx.x; i < total; i += blockDim.x) {
        const uint64_t P = (i < s.len_a) ? s.pa + i : s.pb + (i - s.len_a);
        const uint32_t g = base_bits(planes, P);
        uint8_t ch;
        if (!(g & 4u)) ch = (uint8_t)"ACGT"[g & 3u];
        else if (!rst_bit(rst, P)) ch = (uint8_t)"acgt"[g & 3u];
        else ch = 'N';
        out[s.out_off + i] = ch;
    }
}

// ------------------------------------------------------------------------------- probe
// oligo.BestHit (core/oligo/oligo.go:19-77): one wavefront per amplicon, lanes stride the
// start offsets; both strands; best = fewest mismatches, then leftmost; '+' wins exact
// ties; the k=0 strict-ACGT fast path returns the first '+' occurrence when one exists.
// The wave first stages its amplicon in LDS as one-hot codes (A 1, C 2, G 4, T 8, anything else 0 = matches
// nothing; strings.ToUpper folded in: oligo.go:20) with dword loads, and the two mask rows beside it: the
// arguments may lie in pinned HOST memory (ipcr_probe_best_hit hands the amplicon over that way: no copy
// operation, no device allocation), which a lane must not read a byte at a time per comparison.
// tag != 0: the record's `found` word carries it above bit 0 (one 16-byte store; the host spins on it).
#define IPCR_PROBE_LDS_BYTES 16384u
__device__ __forceinline__ uint32_t probe_onehot(uint32_t b) {
    b &= 0xDFu;
    return (b == 'A') ? 1u : (b == 'C') ? 2u : (b == 'G') ? 4u : (b == 'T') ? 8u : 0u;
}
// scan of one amplicon (n one-hot codes in LDS at sb, or bytes in global memory when not staged) by the THREADS threads
// of a workgroup -> the record; the masks are in s_mask.  Shared by the two kernels below.
template <uint32_t THREADS>
__device__ __forceinline__ void probe_scan(const uint8_t *sb, const uint8_t *gamp, bool staged, uint64_t n, const uint8_t *s_mask,
                                           unsigned long long *s_best, uint32_t plen, uint32_t max_mm, uint32_t fastpath, uint32_t tag,
                                           ipcr_probe_rec *__restrict__ out) {
    unsigned long long best[2] = {~0ull, ~0ull}; // key = mm<<40 | pos
    if (plen > 0 && n >= plen) {
        for (uint64_t pos = threadIdx.x; pos + plen <= n; pos += THREADS) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const uint8_t *mk = s_mask + (s ? 128 : 0);
                uint32_t mm = 0;
                if (staged) {
                    for (uint32_t j = 0; j < plen; ++j)
                        if ((mk[j] & sb[pos + j]) == 0u && ++mm > max_mm) break;
                } else {
                    for (uint32_t j = 0; j < plen; ++j)
                        if ((mk[j] & probe_onehot(gamp[pos + j])) == 0u && ++mm > max_mm) break;
                }
                if (mm <= max_mm) {
                    const unsigned long long key = ((unsigned long long)mm << 40) | pos;
                    if (key < best[s]) best[s] = key;
                }
            }
        }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long o = __shfl_xor(best[s], off);
            if (o < best[s]) best[s] = o;
        }
    if (THREADS > 64u) { // the waves of the workgroup meet in LDS
        if ((threadIdx.x & 63u) == 0u) { atomicMin(&s_best[0], best[0]); atomicMin(&s_best[1], best[1]); }
        __syncthreads();
        best[0] = s_best[0];
        best[1] = s_best[1];
    }
    if (threadIdx.x == 0) {
        ipcr_probe_rec r = {0, 0, 0, 0};
        const bool hp = best[0] != ~0ull, hm = best[1] != ~0ull;
        int pick = -1;
        if (fastpath) pick = hp ? 0 : (hm ? 1 : -1);           // oligo.go:33-42
        else if (hp && hm) pick = (best[1] < best[0]) ? 1 : 0;  // oligo.go:48-76, '+' wins ties
        else pick = hp ? 0 : (hm ? 1 : -1);
        if (pick >= 0) {
            r.found = 1;
            r.strand = pick ? '-' : '+';
            r.pos = (int32_t)(best[pick] & 0xFFFFFFFFFFull);
            r.mm = (int32_t)(best[pick] >> 40);
        }
        r.found |= (int32_t)(tag << 1);
        // ONE 16-byte store: a host that spins on the tag never sees half a record
        *reinterpret_cast<int4 *>(out + blockIdx.x) = make_int4(r.found, r.strand, r.pos, r.mm);
    }
}

template <uint32_t THREADS>
__global__ __launch_bounds__(THREADS) void probe_kernel(const uint8_t *__restrict__ amps,
                                                        const uint64_t *__restrict__ amp_off, // n+1 offsets
                                                        const uint8_t *__restrict__ pmask,    // probe IUPAC masks
                                                        const uint8_t *__restrict__ rmask,    // rc(probe) masks
                                                        uint32_t plen, uint32_t max_mm, uint32_t fastpath, uint32_t tag,
                                                        ipcr_probe_rec *__restrict__ out) {
    __shared__ uint8_t s_mask[256];
    __shared__ uint32_t s_amp[IPCR_PROBE_LDS_BYTES / 4u];
    __shared__ unsigned long long s_best[2];
    const uint64_t a0 = amp_off[blockIdx.x], a1 = amp_off[blockIdx.x + 1];
    const uint64_t n = a1 - a0;
    if (plen > 128u) plen = 128u; // (the host refuses longer probes)
    if (threadIdx.x < 2u) s_best[threadIdx.x] = ~0ull;
    for (uint32_t i = threadIdx.x; i < 256u; i += THREADS) s_mask[i] = (i < 128u) ? (i < plen ? pmask[i] : 0) : (i - 128u < plen ? rmask[i - 128u] : 0);
    const bool staged = n <= IPCR_PROBE_LDS_BYTES;
    if (staged && n > 0) {
        // dwords of the buffer that cover [a0, a1): the first may begin up to 3 bytes in front of the amplicon, the last
        // end up to 3 behind it (inside the buffer: every amplicon buffer carries 16 spare bytes)
        const uintptr_t p0 = reinterpret_cast<uintptr_t>(amps + a0);
        const uint32_t head = (uint32_t)(p0 & 3u);
        const uint32_t *w = reinterpret_cast<const uint32_t *>(p0 - head);
        const uint32_t nw = (uint32_t)((n + head + 3u) / 4u);
        uint8_t *sb = reinterpret_cast<uint8_t *>(s_amp);
        for (uint32_t i = threadIdx.x; i < nw; i += THREADS) {
            const uint32_t v = w[i];
#pragma unroll
            for (uint32_t b = 0; b < 4u; ++b) {
                const int64_t idx = (int64_t)(4u * i + b) - (int64_t)head;
                if (idx >= 0 && (uint64_t)idx < n) sb[idx] = (uint8_t)probe_onehot((v >> (8u * b)) & 0xFFu);
            }
        }
    }
    __syncthreads();
    probe_scan<THREADS>(reinterpret_cast<const uint8_t *>(s_amp), amps + a0, staged, n, s_mask, s_best, plen, max_mm, fastpath, tag, out);
}

// The same with the amplicon read straight from the tiles (gather and rescan in ONE launch): segment seg[blockIdx.x] = the
// product's bases in padded coordinates, [pa, pa + len_a) ++ [pb, pb + len_b) (a wrap-around product has both).  Every base
// becomes its one-hot code in LDS: valid -> its base; lower-case acgt (inv, not rst) -> its base too (BestHit upper-cases
// the amplicon, oligo.go:20); anything else 0.  Amplicons beyond the LDS stage are left to the two-kernel form (host.cpp).
template <uint32_t THREADS>
__global__ __launch_bounds__(THREADS) void probe_tiles_kernel(const uint32_t *__restrict__ planes, const uint32_t *__restrict__ rst,
                                                              const ipcr_amp_seg *__restrict__ segs,
                                                              const uint8_t *__restrict__ pmask, const uint8_t *__restrict__ rmask,
                                                              uint32_t plen, uint32_t max_mm, uint32_t fastpath, uint32_t tag,
                                                              ipcr_probe_rec *__restrict__ out) {
    __shared__ uint8_t s_mask[256];
    __shared__ uint32_t s_amp[IPCR_PROBE_LDS_BYTES / 4u];
    __shared__ unsigned long long s_best[2];
    const ipcr_amp_seg sg = segs[blockIdx.x];
    const uint64_t n = sg.len_a + sg.len_b; // <= IPCR_PROBE_LDS_BYTES (the host checks)
    if (plen > 128u) plen = 128u;
    if (threadIdx.x < 2u) s_best[threadIdx.x] = ~0ull;
    for (uint32_t i = threadIdx.x; i < 256u; i += THREADS) s_mask[i] = (i < 128u) ? (i < plen ? pmask[i] : 0) : (i - 128u < plen ? rmask[i - 128u] : 0);
    uint8_t *sb = reinterpret_cast<uint8_t *>(s_amp);
    for (uint64_t i = threadIdx.x; i < n; i += THREADS) {
        const uint64_t P = (i < sg.len_a) ? sg.pa + i : sg.pb + (i - sg.len_a);
        const uint32_t g = base_bits(planes, P);
        sb[i] = (uint8_t)((!(g & 4u) || !rst_bit(rst, P)) ? (1u << (g & 3u)) : 0u);
    }
    __syncthreads();
    probe_scan<THREADS>(sb, nullptr, true, n, s_mask, s_best, plen, max_mm, fastpath, tag, out);
}

// ---------------------------------------------------------------------------- launchers
#include "launch.h"
namespace ipcr {

hipError_t launch_pack(hipStream_t st, const uint8_t *seq, uint64_t len, uint64_t col0, uint64_t ncol,
                       uint32_t *planes, uint32_t *rst, uint32_t *rec_flags, uint64_t *rec_start_out, uint64_t *rec_len_out,
                       hipEvent_t start, hipEvent_t stop) {
    const uint64_t pairs = (ncol + 1u) / 2u;
    const uint64_t grid = (pairs + 3u) / 4u;
    if (grid == 0) return hipSuccess;
    // start / stop ride on the dispatch itself (its begin and end timestamps): no marker packets around the kernel
    hipExtLaunchKernelGGL(pack_kernel, dim3((uint32_t)grid), dim3(256), 0, st, start, stop, 0,
                          seq, len, col0, ncol, planes, rst, rec_flags, rec_start_out, rec_len_out);
    return hipGetLastError();
}

// columns [col0, col0 + ncol) of the tiles from the linear planes of those columns (rec_col0 = the record's first column, len
// its length: the table entries the kernel writes for the chunk path are the record's, whatever slice a launch converts)
hipError_t launch_tiles_from_linear(hipStream_t st, const uint32_t *lin_lo, const uint32_t *lin_hi, const uint32_t *lin_iv,
                                    const uint32_t *lin_rs, uint64_t rec_col0, uint64_t col0, uint64_t ncol, uint64_t len,
                                    uint32_t *planes, uint32_t *rst, uint64_t *rec_start_out, uint64_t *rec_len_out,
                                    hipEvent_t start, hipEvent_t stop, const uint32_t *iv_cols) {
    const uint64_t grid = (ncol * 4u + 7u) / 8u; // 8 half waves per workgroup
    if (grid == 0) return hipSuccess;
    hipExtLaunchKernelGGL(tiles_from_linear_kernel, dim3((uint32_t)grid), dim3(256), 0, st, start, stop, 0,
                          reinterpret_cast<const uint4 *>(lin_lo), reinterpret_cast<const uint4 *>(lin_hi),
                          reinterpret_cast<const uint4 *>(lin_iv), reinterpret_cast<const uint4 *>(lin_rs), iv_cols, rec_col0, col0, ncol, len,
                          planes, rst, rec_start_out, rec_len_out);
    return hipGetLastError();
}

hipError_t launch_pack_batch(hipStream_t st, const uint8_t *base, const ipcr_pack_rec *recs, const uint32_t *pair_prefix,
                             uint32_t nrec, uint64_t total_pairs, uint32_t *planes, uint32_t *rst, uint32_t *rec_flags) {
    const uint64_t grid = (total_pairs + 3u) / 4u;
    if (grid == 0 || nrec == 0) return hipSuccess;
    pack_batch_kernel<<<dim3((uint32_t)grid), dim3(256), 0, st>>>(base, recs, pair_prefix, nrec, planes, rst, rec_flags);
    return hipGetLastError();
}

hipError_t launch_window_reset(hipStream_t st, const uint32_t *rst, const uint64_t *win, uint32_t nwin, uint32_t *flags) {
    if (nwin == 0) return hipSuccess;
    window_reset_kernel<<<dim3(nwin), dim3(256), 0, st>>>(rst, win, nwin, flags);
    return hipGetLastError();
}

hipError_t launch_fill_pad(hipStream_t st, uint32_t *planes, uint32_t *rst, uint64_t col_begin, uint64_t col_end) {
    if (col_end <= col_begin) return hipSuccess;
    const uint64_t n = (col_end - col_begin) * 128u;
    fill_pad_kernel<<<dim3((uint32_t)((n + 255u) / 256u)), dim3(256), 0, st>>>(planes, rst, col_begin, col_end);
    return hipGetLastError();
}

hipError_t launch_lcg(hipStream_t st, uint8_t *out, uint64_t n, uint32_t seed, uint64_t offset) {
    if (n == 0) return hipSuccess;
    const uint64_t runs = (n + 63u) / 64u;
    lcg_fill_kernel<<<dim3((uint32_t)((runs + 255u) / 256u)), dim3(256), 0, st>>>(out, n, seed, offset);
    return hipGetLastError();
}

hipError_t launch_filter_generic(hipStream_t st, const uint32_t *planes, uint64_t block0, uint64_t nblocks,
                                 const ipcr_dev_pattern *pats, uint32_t npat, uint32_t max_mm, const uint32_t *sel,
                                 ipcr_queue_entry *queue, uint64_t qcap, unsigned long long *qcount,
                                 hipEvent_t start, hipEvent_t stop) {
    if (nblocks == 0 || npat == 0) return hipSuccess;
    if (max_mm > 16u || nblocks * 8u > 0x7FFFFFFFull) return hipErrorInvalidValue; // (IPCR_MAX_MM; 2^28 blocks = 7e13 bases)
    const dim3 qgrid((uint32_t)(nblocks * 8u)); // a wave per (block, row quad): 32 row quads per block, 4 waves per workgroup
    // counter depth K1 = max_mm + 1 at compile time; patterns per walk by what 4 start rows x PB x K1 counters leave of the registers
#define IPCR_QUAD_LAUNCH(K1, PB) hipExtLaunchKernelGGL((filter_generic_quad_kernel<K1, PB>), qgrid, dim3(256), 0, st, start, stop, 0, planes, block0, nblocks, \
                                                       pats, npat, sel, queue, qcap, qcount); break
    switch (max_mm) {
    case 0: IPCR_QUAD_LAUNCH(1, 4);
    case 1: IPCR_QUAD_LAUNCH(2, 4);
    case 2: IPCR_QUAD_LAUNCH(3, 4);
    case 3: IPCR_QUAD_LAUNCH(4, 4);
    case 4: IPCR_QUAD_LAUNCH(5, 2);
    case 5: IPCR_QUAD_LAUNCH(6, 2);
    case 6: IPCR_QUAD_LAUNCH(7, 2);
    case 7: IPCR_QUAD_LAUNCH(8, 2);
    case 8: IPCR_QUAD_LAUNCH(9, 1);
    case 9: IPCR_QUAD_LAUNCH(10, 1);
    case 10: IPCR_QUAD_LAUNCH(11, 1);
    case 11: IPCR_QUAD_LAUNCH(12, 1);
    case 12: IPCR_QUAD_LAUNCH(13, 1);
    case 13: IPCR_QUAD_LAUNCH(14, 1);
    case 14: IPCR_QUAD_LAUNCH(15, 1);
    case 15: IPCR_QUAD_LAUNCH(16, 1);
    default: IPCR_QUAD_LAUNCH(17, 1);
    }
#undef IPCR_QUAD_LAUNCH
    return hipGetLastError();
}

hipError_t launch_verify(hipStream_t st, const uint32_t *planes, const uint32_t *rst,
                         const ipcr_dev_pattern *pats, uint32_t max_mm, const uint64_t *rec_start,
                         const uint64_t *rec_len, uint32_t nrec, uint32_t check_rst, const ipcr_queue_entry *queue,
                         uint64_t qcap, const unsigned long long *qcount, ipcr_hit_rec *hits, uint64_t hcap,
                         unsigned long long *hcount, unsigned long long *ccount, unsigned long long *next_counters,
                         unsigned long long *next_qcount, hipEvent_t start, hipEvent_t stop, uint32_t single) {
    if (nrec == 0) return hipSuccess;
    hipExtLaunchKernelGGL(verify_kernel, dim3(IPCR_VERIFY_BLOCKS), dim3(256), 0, st, start, stop, 0, planes, rst, pats, max_mm,
                          rec_start, rec_len, nrec, check_rst, queue, qcap, qcount, hits, hcap, hcount, ccount,
                          next_counters, next_qcount, single);
    return hipGetLastError();
}

hipError_t launch_unpack(hipStream_t st, const uint32_t *planes, const uint32_t *rst, uint64_t P0, uint64_t n,
                         uint8_t *out) {
    if (n == 0) return hipSuccess;
    unpack_kernel<<<dim3((uint32_t)((n + 255u) / 256u)), dim3(256), 0, st>>>(planes, rst, P0, n, out);
    return hipGetLastError();
}

hipError_t launch_gather(hipStream_t st, const uint32_t *planes, const uint32_t *rst, const ipcr_amp_seg *segs,
                         uint32_t nseg, uint8_t *out) {
    if (nseg == 0) return hipSuccess;
    gather_amplicons_kernel<<<dim3(nseg), dim3(256), 0, st>>>(planes, rst, segs, out);
    return hipGetLastError();
}

hipError_t launch_probe(hipStream_t st, const uint8_t *amps, const uint64_t *amp_off, uint32_t namp,
                        const uint8_t *pmask, const uint8_t *rmask, uint32_t plen, uint32_t max_mm,
                        uint32_t fastpath, ipcr_probe_rec *out, uint32_t tag) {
    if (namp == 0) return hipSuccess;
    // few amplicons (a chunk's products, one amplicon of a collector): four waves share each; a large batch has waves enough
    if (namp <= 256u) probe_kernel<256><<<dim3(namp), dim3(256), 0, st>>>(amps, amp_off, pmask, rmask, plen, max_mm, fastpath, tag, out);
    else probe_kernel<64><<<dim3(namp), dim3(64), 0, st>>>(amps, amp_off, pmask, rmask, plen, max_mm, fastpath, tag, out);
    return hipGetLastError();
}

hipError_t launch_probe_tiles(hipStream_t st, const uint32_t *planes, const uint32_t *rst, const ipcr_amp_seg *segs, uint32_t namp,
                              const uint8_t *pmask, const uint8_t *rmask, uint32_t plen, uint32_t max_mm, uint32_t fastpath,
                              ipcr_probe_rec *out, uint32_t tag) {
    if (namp == 0) return hipSuccess;
    if (namp <= 256u) probe_tiles_kernel<256><<<dim3(namp), dim3(256), 0, st>>>(planes, rst, segs, pmask, rmask, plen, max_mm, fastpath, tag, out);
    else probe_tiles_kernel<64><<<dim3(namp), dim3(64), 0, st>>>(planes, rst, segs, pmask, rmask, plen, max_mm, fastpath, tag, out);
    return hipGetLastError();
}

} // namespace ipcr
