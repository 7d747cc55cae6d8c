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
