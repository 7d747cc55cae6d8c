// hostpack.h -- FASTA text -> bit planes on the host (hostpack.cpp), shared with host.cpp
#pragma once
#include <stdint.h>

namespace ipcr {

// the words at a piece's two ends that it shares with its neighbours (at most two): values hold this piece's bits only, the
// caller ORs entries of the same word and writes the word once
struct FastaEdge {
    uint32_t n;
    uint64_t word[2];
    uint64_t val[2][3]; // lo, hi, iv
};
bool fasta_blocks_supported();
// region: a record's sequence text, `area` = full * (W + lt) bytes of whole lines (W base bytes + lt terminator bytes each);
// tab: four tables of (W + lt) masks (host.cpp: fasta_tables); blocks [j0, j1) of 64 bytes; lo / hi / iv: the record's linear
// planes as 64-bit words.  false: the text is not that regular.
bool pack_fasta_blocks(const uint8_t *region, uint64_t area, uint32_t W, uint32_t lt, const uint64_t *tab, uint64_t j0, uint64_t j1,
                       uint64_t *lo, uint64_t *hi, uint64_t *iv, FastaEdge *edge, uint32_t *any_invalid);

} // namespace ipcr
