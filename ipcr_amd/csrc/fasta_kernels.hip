// fasta_kernels.hip -- raw FASTA text -> contiguous upper-case sequence bytes, on the device.
//
// The reference normalises every sequence line on one CPU thread (core/fasta/scan.go:10-69 splits
// lines, core/fasta/normalize.go:5-14 trims white space at both ends and folds a-z to A-Z).  Here a
// slab of raw file bytes is copied to HBM as it is; the header lines are located there too
// (fasta_find_headers: '>' at a line start), the host sorts the handful of ranges and parses the
// IDs.  Three small kernels then do what the per-line loop does, for all lines at once:
//   fasta_count    keep-mask of every 16-byte group -> kept bytes per 4 KiB block
//   fasta_scan     exclusive prefix sum of the block counts (one workgroup)
//   fasta_scatter  kept bytes, upper-cased, to their compacted position; for every header the
//                  compacted offset at which the record after it begins
// keep(byte) = not inside a header line, not '\n', and not white space that only white space
// separates from its line's start or end (bytes.TrimSpace on the line; white space INSIDE a line
// stays, as in the reference, and later counts as a non-ACGT base).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_types.h"
#include "launch.h"

namespace {

__device__ __forceinline__ bool fasta_ws(uint32_t c) { // the ASCII set of bytes.TrimSpace
    return c == ' ' || (c >= '\t' && c <= '\r');
}

// white-space byte at i (not '\n'): kept only when the line has other bytes on both sides
__device__ bool fasta_ws_kept(const uint8_t *__restrict__ raw, uint64_t n, uint64_t i, uint32_t lead_open0) {
    uint64_t j = i;
    for (;;) { // towards the line start
        if (j == 0) {
            if (lead_open0) return false; // the slab begins at a line start (or in a line that is blank so far)
            break;
        }
        const uint32_t c = raw[j - 1];
        if (c == '\n') return false;
        if (!fasta_ws(c)) break;
        --j;
    }
    for (j = i + 1; j < n; ++j) { // towards the line end
        const uint32_t c = raw[j];
        if (c == '\n') return false;
        if (!fasta_ws(c)) return true;
    }
    return false; // the host never ends a slab inside a run of white space unless the file ends there
}

// first header whose end lies beyond position p
__device__ __forceinline__ uint32_t fasta_first_header(const ipcr_fasta_range *__restrict__ hdr, uint32_t nh, uint64_t p) {
    uint32_t lo = 0, hi = nh;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (hdr[mid].end > p) hi = mid; else lo = mid + 1;
    }
    return lo;
}

// bit t = byte g0+t is kept
__device__ uint32_t fasta_keep_mask(const uint8_t *__restrict__ raw, uint64_t n, uint64_t g0, const uint4 v,
                                    const ipcr_fasta_range *__restrict__ hdr, uint32_t nh, uint32_t lead_open0) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t mask = 0;
    uint32_t h = fasta_first_header(hdr, nh, g0);
#pragma unroll
    for (uint32_t t = 0; t < 16; ++t) {
        const uint64_t i = g0 + t;
        if (i >= n) break;
        while (h < nh && hdr[h].end <= i) ++h;
        if (h < nh && hdr[h].start <= i) continue; // header line
        const uint32_t c = (w[t >> 2] >> ((t & 3u) * 8u)) & 0xFFu;
        if (c == '\n') continue;
        if (!fasta_ws(c) || fasta_ws_kept(raw, n, i, lead_open0)) mask |= 1u << t;
    }
    return mask;
}

__device__ __forceinline__ uint4 fasta_load16(const uint8_t *__restrict__ raw, uint64_t n, uint64_t g0) {
    if (g0 + 16u <= n) return *reinterpret_cast<const uint4 *>(raw + g0); // raw is 16-byte aligned
    uint32_t w[4] = {0, 0, 0, 0};
    for (uint32_t t = 0; t < 16 && g0 + t < n; ++t) w[t >> 2] |= (uint32_t)raw[g0 + t] << ((t & 3u) * 8u);
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// header lines of the slab: '>' at a line start (core/fasta/scan.go:27) up to and including its '\n' (or the slab's end).
// One thread per 16 bytes; the few threads that find one walk to the line end themselves and append the range --
// in no particular order, the host sorts the handful.  list[cap] ranges; *count may exceed cap (the host then retries).
__global__ __launch_bounds__(256) void fasta_find_headers_kernel(const uint8_t *__restrict__ raw, uint64_t n, uint32_t at_line_start,
                                                                 ipcr_fasta_range *__restrict__ list, uint32_t cap, uint32_t *__restrict__ count) {
    const uint64_t g0 = ((uint64_t)blockIdx.x * 256u + threadIdx.x) * 16u;
    if (g0 >= n) return;
    const uint4 v = fasta_load16(raw, n, g0);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t prev = g0 == 0 ? (at_line_start ? (uint32_t)'\n' : 0u) : (uint32_t)raw[g0 - 1];
#pragma unroll
    for (uint32_t t = 0; t < 16; ++t) {
        if (g0 + t >= n) break;
        const uint32_t c = (w[t >> 2] >> ((t & 3u) * 8u)) & 0xFFu;
        if (c == '>' && prev == '\n') {
            uint64_t e = g0 + t + 1;
            while (e < n && raw[e] != '\n') ++e;
            const uint32_t slot = atomicAdd(count, 1u);
            if (slot < cap) { list[slot].start = g0 + t; list[slot].end = e < n ? e + 1 : n; }
        }
        prev = c;
    }
}

__global__ __launch_bounds__(256) void fasta_count_kernel(const uint8_t *__restrict__ raw, uint64_t n,
                                                          const ipcr_fasta_range *__restrict__ hdr, uint32_t nh,
                                                          uint32_t lead_open0, uint32_t *__restrict__ counts) {
    __shared__ uint32_t s_sum[4];
    const uint64_t g0 = ((uint64_t)blockIdx.x * 256u + threadIdx.x) * 16u;
    uint32_t c = 0;
    if (g0 < n) c = (uint32_t)__builtin_popcount(fasta_keep_mask(raw, n, g0, fasta_load16(raw, n, g0), hdr, nh, lead_open0));
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
    if ((threadIdx.x & 63u) == 0) s_sum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
}

// counts[0..nb) -> exclusive prefix in place, total to counts[nb]; one workgroup of 1024 threads
__global__ __launch_bounds__(1024) void fasta_scan_kernel(uint32_t *__restrict__ counts, uint32_t nb) {
    __shared__ uint32_t s_part[1024];
    const uint32_t t = threadIdx.x;
    const uint32_t per = (nb + 1023u) / 1024u;
    const uint32_t b = t * per, e = (b + per < nb) ? b + per : nb;
    uint32_t sum = 0;
    for (uint32_t i = b; i < e; ++i) sum += counts[i];
    s_part[t] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024u; d <<= 1) { // Hillis-Steele over the 1024 partial sums
        const uint32_t v = (t >= d) ? s_part[t - d] : 0u;
        __syncthreads();
        s_part[t] += v;
        __syncthreads();
    }
    uint32_t run = t ? s_part[t - 1] : 0u;
    for (uint32_t i = b; i < e; ++i) {
        const uint32_t c = counts[i];
        counts[i] = run;
        run += c;
    }
    if (t == 1023u) counts[nb] = s_part[1023];
}

__global__ __launch_bounds__(256) void fasta_scatter_kernel(const uint8_t *__restrict__ raw, uint64_t n,
                                                            const ipcr_fasta_range *__restrict__ hdr, uint32_t nh,
                                                            uint32_t lead_open0, const uint32_t *__restrict__ offsets,
                                                            uint8_t *__restrict__ out, uint32_t *__restrict__ hdr_off) {
    __shared__ uint32_t s_wave[4];
    const uint64_t g0 = ((uint64_t)blockIdx.x * 256u + threadIdx.x) * 16u;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    uint4 v = make_uint4(0, 0, 0, 0);
    uint32_t mask = 0;
    if (g0 < n) {
        v = fasta_load16(raw, n, g0);
        mask = fasta_keep_mask(raw, n, g0, v, hdr, nh, lead_open0);
    }
    const uint32_t c = (uint32_t)__builtin_popcount(mask);
    uint32_t incl = c; // inclusive scan inside the wave
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d);
        if ((int)lane >= d) incl += up;
    }
    if (lane == 63u) s_wave[wv] = incl;
    __syncthreads();
    uint32_t base = offsets[blockIdx.x];
    for (uint32_t k = 0; k < wv; ++k) base += s_wave[k];
    base += incl - c;
    if (g0 >= n) return;
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t o = base;
#pragma unroll
    for (uint32_t t = 0; t < 16; ++t) {
        if ((mask >> t) & 1u) {
            uint32_t ch = (w[t >> 2] >> ((t & 3u) * 8u)) & 0xFFu;
            if (ch >= 'a' && ch <= 'z') ch -= 'a' - 'A'; // normalize.go:8-10
            out[o++] = (uint8_t)ch;
        }
    }
    // headers that begin in my group: the record behind header k starts at this compacted offset
    for (uint32_t h = fasta_first_header(hdr, nh, g0); h < nh && hdr[h].start < g0 + 16u; ++h) {
        if (hdr[h].start < g0) continue;
        const uint32_t t = (uint32_t)(hdr[h].start - g0);
        hdr_off[h] = base + (uint32_t)__builtin_popcount(mask & ((1u << t) - 1u));
    }
}

} // namespace

namespace ipcr {

// list: cap ranges, count: one word (cleared here)
hipError_t launch_fasta_find_headers(hipStream_t st, const uint8_t *raw, uint64_t n, uint32_t at_line_start, ipcr_fasta_range *list,
                                     uint32_t cap, uint32_t *count) {
    hipError_t e = hipMemsetAsync(count, 0, 4, st);
    if (e != hipSuccess || n == 0) return e;
    const uint32_t nb = (uint32_t)((n + 4095u) / 4096u);
    hipLaunchKernelGGL(fasta_find_headers_kernel, dim3(nb), dim3(256), 0, st, raw, n, at_line_start, list, cap, count);
    return hipGetLastError();
}

// counts: (nblocks + 1) words of scratch; afterwards counts[nblocks] = kept bytes of the slab
hipError_t launch_fasta_decode(hipStream_t st, const uint8_t *raw, uint64_t n, const ipcr_fasta_range *hdr, uint32_t nh,
                               uint32_t lead_open0, uint32_t *counts, uint8_t *out, uint32_t *hdr_off) {
    if (n == 0) return hipMemsetAsync(counts, 0, 4, st);
    const uint32_t nb = (uint32_t)((n + 4095u) / 4096u);
    hipLaunchKernelGGL(fasta_count_kernel, dim3(nb), dim3(256), 0, st, raw, n, hdr, nh, lead_open0, counts);
    hipLaunchKernelGGL(fasta_scan_kernel, dim3(1), dim3(1024), 0, st, counts, nb);
    hipLaunchKernelGGL(fasta_scatter_kernel, dim3(nb), dim3(256), 0, st, raw, n, hdr, nh, lead_open0, counts, out, hdr_off);
    return hipGetLastError();
}

} // namespace ipcr
