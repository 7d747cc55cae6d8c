// launch.h -- host-callable launchers of the kernels in kernels.hip
#pragma once
#include "hostpack.h"
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "device_types.h"

namespace ipcr {
hipError_t launch_pack(hipStream_t st, const uint8_t *seq, uint64_t len, uint64_t col0, uint64_t ncol,
                       uint32_t *planes, uint32_t *rst, uint32_t *rec_flags, uint64_t *rec_start_out = nullptr,
                       uint64_t *rec_len_out = nullptr, // rec_*_out: the kernel also writes the record's table entries (chunk path)
                       hipEvent_t start = nullptr, hipEvent_t stop = nullptr);
// host-packed linear bit planes (hostpack.cpp) -> tiles; lin_* point at the words of column col0 (the slice's first)
hipError_t launch_tiles_from_linear(hipStream_t st, const uint32_t *lin_lo, const uint32_t *lin_hi, const uint32_t *lin_iv,
                                    const uint32_t *lin_rs, uint64_t rec_col0, uint64_t col0, uint64_t ncol, uint64_t len,
                                    uint32_t *planes, uint32_t *rst, uint64_t *rec_start_out, uint64_t *rec_len_out,
                                    hipEvent_t start, hipEvent_t stop, const uint32_t *iv_cols = nullptr);
// ASCII -> linear planes on the host (returns bit 0: a byte outside ACGTacgt, bit 1: a lower-case acgt)
uint32_t pack_linear(const uint8_t *seq, uint64_t len, uint64_t padded, uint32_t *lo, uint32_t *hi, uint32_t *iv, uint32_t *rs);
bool pack_linear_is_simd();

hipError_t launch_pack_batch(hipStream_t st, const uint8_t *base, const ipcr_pack_rec *recs, const uint32_t *pair_prefix,
                             uint32_t nrec, uint64_t total_pairs, uint32_t *planes, uint32_t *rst, uint32_t *rec_flags);
hipError_t launch_fill_pad(hipStream_t st, uint32_t *planes, uint32_t *rst, uint64_t col_begin, uint64_t col_end);
// flags[i] = does [win[2 i], win[2 i + 1]) (padded genome coordinates) hold a reset byte (win, flags: device memory)
hipError_t launch_window_reset(hipStream_t st, const uint32_t *rst, const uint64_t *win, uint32_t nwin, uint32_t *flags);
hipError_t launch_lcg(hipStream_t st, uint8_t *out, uint64_t n, uint32_t seed, uint64_t offset);
hipError_t launch_filter_generic(hipStream_t st, const uint32_t *planes, uint64_t block0, uint64_t nblocks,
                                 const ipcr_dev_pattern *pats, uint32_t npat, uint32_t max_mm, const uint32_t *sel,
                                 ipcr_queue_entry *queue, uint64_t qcap, unsigned long long *qcount,
                                 hipEvent_t start, hipEvent_t stop);
hipError_t launch_verify(hipStream_t st, const uint32_t *planes, const uint32_t *rst,
                         const ipcr_dev_pattern *pats, uint32_t max_mm, const uint64_t *rec_start,
                         const uint64_t *rec_len, uint32_t nrec, uint32_t check_rst, const ipcr_queue_entry *queue,
                         uint64_t qcap, const unsigned long long *qcount, ipcr_hit_rec *hits, uint64_t hcap,
                         unsigned long long *hcount, unsigned long long *ccount, unsigned long long *next_counters,
                         unsigned long long *next_qcount, hipEvent_t start, hipEvent_t stop,
                         uint32_t single = 0); // single: every queue entry holds one candidate (bits has one bit set)
hipError_t launch_unpack(hipStream_t st, const uint32_t *planes, const uint32_t *rst, uint64_t P0, uint64_t n,
                         uint8_t *out);
hipError_t launch_gather(hipStream_t st, const uint32_t *planes, const uint32_t *rst, const ipcr_amp_seg *segs,
                         uint32_t nseg, uint8_t *out);
hipError_t launch_probe(hipStream_t st, const uint8_t *amps, const uint64_t *amp_off, uint32_t namp,
                        const uint8_t *pmask, const uint8_t *rmask, uint32_t plen, uint32_t max_mm,
                        uint32_t fastpath, ipcr_probe_rec *out,
                        uint32_t tag = 0); // tag != 0: out[i].found = found | tag << 1 (the host spins on it in pinned memory)
// gather + rescan in one launch: every amplicon (<= launch_probe_tiles_max() bases) read straight from the tiles
hipError_t launch_probe_tiles(hipStream_t st, const uint32_t *planes, const uint32_t *rst, const ipcr_amp_seg *segs, uint32_t namp,
                              const uint8_t *pmask, const uint8_t *rmask, uint32_t plen, uint32_t max_mm, uint32_t fastpath,
                              ipcr_probe_rec *out, uint32_t tag);
inline uint64_t launch_probe_tiles_max() { return 16384u; } // IPCR_PROBE_LDS_BYTES (kernels.hip)
// header lines ('>' at a line start .. its line end) of a raw FASTA slab, unordered; *count may exceed cap
hipError_t launch_fasta_find_headers(hipStream_t st, const uint8_t *raw, uint64_t n, uint32_t at_line_start, ipcr_fasta_range *list,
                                     uint32_t cap, uint32_t *count);
// raw FASTA slab -> compacted upper-case sequence bytes (fasta_kernels.hip); counts = nblocks(4 KiB) + 1 words
hipError_t launch_fasta_decode(hipStream_t st, const uint8_t *raw, uint64_t n, const ipcr_fasta_range *hdr, uint32_t nh,
                               uint32_t lead_open0, uint32_t *counts, uint8_t *out, uint32_t *hdr_off);
} // namespace ipcr
