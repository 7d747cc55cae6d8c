// tile_layout.h -- how a genome is laid out in HBM for the bit-sliced scan.
//
// Padded genome coordinate P (all records concatenated, each record start aligned to a
// column and followed by >= IPCR_MAX_PRIMER_LEN invalid pad bases) is cut into STRANDS of
// N = 128 consecutive bases.  32 consecutive strands form a COLUMN (one 32-bit word per
// row: bit i of the word of row r = base r of strand 32*column + i); 64 consecutive
// columns form a BLOCK (what one wavefront scans: lane = column).  So the base that follows
// a word's bit at row r sits in the SAME bit of the SAME lane at row r+1: a primer window
// is a run of rows, and the k-mismatch filter needs no shifts at all.
//
// Planes per base: lo, hi (2-bit code A=0,C=1,G=2,T=3) and inv (1 = not an upper-case
// A/C/G/T, core/primer/iupac.go:62-67) -- 0.375 B/base, the bytes the filter kernel reads
// once.  A fourth plane rst (1 = byte outside ACGTacgt, the automaton-reset bytes of
// core/engine/ac.go:16-30,141-148) lives in its own buffer and is read by the verifier only.
//
// Word order inside a block: [row/4][plane][lane][row%4], so a wave loads 4 rows of one
// plane as one contiguous 1 KiB global_load_dwordx4.
#pragma once
#include <stdint.h>

#define IPCR_TILE_N 128u             // rows per strand
#define IPCR_TILE_LOG_N 7u
#define IPCR_COLUMN_BASES 4096u      // 32 strands * 128
#define IPCR_BLOCK_COLUMNS 64u
#define IPCR_BLOCK_BASES 262144u     // 64 * 4096
#define IPCR_BLOCK_PLANE_WORDS 24576u // 32 row-quads * 3 planes * 64 lanes * 4
#define IPCR_BLOCK_RST_WORDS 8192u   // 32 row-quads * 64 lanes * 4
#define IPCR_PAD_BASES 128u          // invalid bases kept after every record

#if defined(__HIPCC__) || defined(__HIP_DEVICE_COMPILE__)
#define IPCR_HD __host__ __device__ __forceinline__
#else
#define IPCR_HD static inline
#endif

// word index of (block, row, plane, lane) in the lo/hi/inv buffer
IPCR_HD uint64_t ipcr_plane_word(uint64_t block, uint32_t row, uint32_t plane, uint32_t lane) {
    return ((((block * 32u + (row >> 2)) * 3u + plane) * 64u + lane) << 2) + (row & 3u);
}
// word index of (block, row, lane) in the rst buffer
IPCR_HD uint64_t ipcr_rst_word(uint64_t block, uint32_t row, uint32_t lane) {
    return (((block * 32u + (row >> 2)) * 64u + lane) << 2) + (row & 3u);
}
// padded position -> (column, bit, row)
IPCR_HD void ipcr_split_pos(uint64_t P, uint64_t *column, uint32_t *bit, uint32_t *row) {
    uint64_t strand = P >> IPCR_TILE_LOG_N;
    *row = (uint32_t)(P & (IPCR_TILE_N - 1u));
    *column = strand >> 5;
    *bit = (uint32_t)(strand & 31u);
}
IPCR_HD uint64_t ipcr_join_pos(uint64_t column, uint32_t bit, uint32_t row) {
    return (((column << 5) + bit) << IPCR_TILE_LOG_N) + row;
}
