// device_types.h -- PODs shared by the host runtime and the kernels.
#pragma once
#include <stdint.h>

// One distinct scanned pattern (primer orientation).  mask[j]: low 4 bits = IUPAC mask of
// pattern base j (A=1,C=2,G=4,T=8; core/primer/iupac.go:6-58), bit 4 = position j lies in the
// protected terminal window (a mismatch there rejects; core/engine/ac.go:197-205).
// seed_off/seed_len: the span the reference would seed (core/engine/seed.go:260-283), used
// only to reproduce its hit ORDER when a record holds non-ACGT bytes and HitCap == 0.
struct ipcr_dev_pattern {
    uint16_t len;
    uint16_t seed_off;
    uint16_t seed_len;
    uint16_t reserved;
    uint32_t global_id; // panel-wide pattern id written into ipcr_hit.pattern
    uint8_t mask[128];
};

// filter -> verifier queue: one surviving WORD: pattern (set-local index) in key bits 48..63,
// padded position of strand bit 0 in bits 0..47; bit b of `bits` = strand b survived
struct ipcr_queue_entry {
    uint64_t key;
    uint32_t bits;
    uint32_t pad;
};

// the queue is cut into segments, one push counter each, 128 B apart
#define IPCR_QUEUE_SHARDS 256u
#define IPCR_QUEUE_COUNTER_STRIDE 16u // in 8-byte words

// layout-identical to ipcr_hit in include/ipcr_hip.h
struct ipcr_hit_rec {
    uint64_t pos;
    uint32_t record;
    uint32_t pattern;
    uint64_t mm_mask[2];
};

struct ipcr_amp_seg { // amplicon = [pa, pa+len_a) ++ [pb, pb+len_b) in padded coordinates
    uint64_t pa, len_a, pb, len_b, out_off;
};

struct ipcr_probe_rec { // layout-identical to ipcr_probe_hit
    int32_t found, strand, pos, mm;
};

// ---- seed-index filter for large panels (jit.cpp: jit_index_source) ----
// One shape = one exact "key" every window of a pattern group must contain if it is to match
// with <= k mismatches: protected terminal bases plus one of k+1 pigeonhole blocks, both
// relative to the anchored end of the window.  Keys are read out of a rolling 2-bit k-mer
// (newest base in bits 1:0).  Right-anchored groups (A/B orientations, 3' window at the window end)
// are tested when their window ENDS at the newest base; left-anchored groups (rc orientations) when
// their window STARTED `dl` bases ago, dl = longest left pattern - 1, so that the whole window is
// already in the k-mer.
//   key = ((kmer >> tw_shift) & tw_mask) | (((kmer >> blk_shift) & blk_mask) << tw_bits)      (<= 17 bits)
// A shape's keys are a bitmap in LDS, addressed as 64-bit words: word key >> 6, bit key & 63 (2^(key bits - 6)
// words, at least 16; the shapes' bitmaps lie one after the other: ipcr_index_words64).  Every shape is looked
// up at the base step that completes its window -- all its fields are in the 32-base k-mer register by then.
// In a FAST group the low six key bits (three protected bases next to the anchor) are common to all its
// shapes: they are extracted once per step and select the bit, the shape's block field selects the word.
// A block field is a run of bits, not of bases: with a spare base next to a five-base block the key takes one
// bit of it too (11 block bits, half the false hits); no base ever feeds two blocks, which is all the
// pigeonhole argument needs.
struct ipcr_index_shape {
    uint8_t left;       // 1: anchored at the window start (rc orientations), 0: at its end
    uint8_t tw_shift;   // k-mer bit offset of the protected part
    uint8_t blk_shift;  // k-mer bit offset of the block part
    uint8_t tw_bits;    // 2 * bases of the protected part used in the key
    uint32_t tw_mask;   // (1 << tw_bits) - 1
    uint32_t blk_mask;  // (1 << block bits) - 1
    uint8_t group;      // shapes of one (anchored end, protected length) share a group
    uint8_t paired;     // 1: the shape's table serves TWO consecutive base steps per 32-bit word (jit.cpp: "two steps per lookup")
    uint8_t fast;       // the group's shapes share their low six key bits (else the whole key is computed per shape)
    uint8_t dl;         // left shapes: the window start lies dl bases behind the newest base
    uint64_t valid_mask; // even bits of the k-mer bases the key reads (a base that gives one bit counts)
};

static inline uint32_t ipcr_index_key_bits(const ipcr_index_shape &s) { return (uint32_t)s.tw_bits + (uint32_t)__builtin_popcount(s.blk_mask); }
// 64-bit words of a shape's bitmap: never below 16 (so that word offsets fit the 11 bits the drain's constants have, in units of 16)
static inline uint32_t ipcr_index_words64(const ipcr_index_shape &s) {
    const uint32_t kb = ipcr_index_key_bits(s);
    if (s.paired) return 2048u; // 2^12 words of 32 bits: 16 bits per step of the pair (16-bit keys only)
    return kb <= 10u ? 16u : (1u << (kb - 6u));
}

struct ipcr_index_entry { // 64 B; entry r belongs to the r-th distinct (shape, key) in sorted order
    uint32_t next;    // further pattern with the same key (index into the same array), 0xFFFFFFFF = none
    uint32_t pattern; // set-local pattern index
    uint64_t seq2;    // pure ACGT patterns: the primer as 2-bit codes, position j in bits 2*(L-1-j)+1 : 2*(L-1-j)
    uint64_t prot2;   // even bit 2*(L-1-j) set when position j is protected
    uint32_t len;
    uint32_t flags;   // bit 0: left-anchored, bit 1: pure ACGT (seq2 valid: the first 32 bytes suffice for the check)
    uint64_t ok[4];   // IUPAC masks as four position sets: even bit 2*(L-1-j) of ok[b] set when base b
                      // (A,C,G,T) is allowed at pattern position j -- pure ACGT primers are the one-hot case
};
static_assert(sizeof(ipcr_index_entry) == 64, "index entries are read as four 16-byte loads");

#define IPCR_INDEX_MAX_SHAPES 12
#define IPCR_INDEX_MAX_KEY_BITS 17u

// raw byte range [start, end) of one FASTA header line (its '\n' included) inside a slab
struct ipcr_fasta_range {
    uint64_t start, end;
};

// one record of a batched pack launch: 16-byte aligned source bytes, destination columns, index of its flag word
struct ipcr_pack_rec {
    uint64_t src_off, len, col0;
    uint32_t ncol, flag_idx;
};
