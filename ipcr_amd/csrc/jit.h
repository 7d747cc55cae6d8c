// jit.h -- panel-specialised filter kernel: generated as HIP source for the concrete primer
// panel and compiled for gfx950 with hiprtc when the panel is first scanned.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "device_types.h"

namespace ipcr {

struct JitFilter;

// HIP source of the specialised filter for one group of patterns (also used by the build
// check); queue entries carry pattern index qbase + position in `pats`
// ids: the patterns' indices in the panel's device table when they are not qbase, qbase + 1, ... (a subset of a panel);
// spill_only: every survivor goes to the candidate queue (the stand-alone verifier follows), nothing is verified in the kernel
// segments > 1: the form for small launches -- a block is shared by that many waves (jit.cpp); "" when the panel's loop cannot be cut so
std::string jit_source(const std::vector<ipcr_dev_pattern> &pats, int max_mm, unsigned qbase = 0,
                       const std::vector<uint32_t> *ids = nullptr, bool spill_only = false, int segments = 1);
// patterns per kernel for this panel; 0 = not specialisable (table-driven filter)
size_t jit_group_size(const std::vector<ipcr_dev_pattern> &pats, int max_mm);
// one kernel per pattern group, compiled in parallel; empty (and `err` set) when the panel
// cannot be specialised or hiprtc fails
// subset: build for these patterns of `pats` only, as spill-only kernels (what a seed-index panel cannot key)
std::vector<JitFilter *> jit_build(const std::vector<ipcr_dev_pattern> &pats, int max_mm, std::string &err,
                                   const std::vector<uint32_t> *subset = nullptr);
// The specialised filter verifies its own survivors (each wave, when its block is done) and appends
// the hit records itself: these are the exact verifier's operands.  counts = this scan's counter
// set ([0] survivor words spilled to the queue because a wave's list was full -> the stand-alone
// verifier must run over the queue, [1] hits, [2] candidate windows); next_* = the sets of the next
// scan, cleared by workgroup 0 (null for every kernel of a scan but the first).
struct JitVerify {
    const uint32_t *rst = nullptr;
    const ipcr_dev_pattern *pats = nullptr;
    const uint64_t *rec_start = nullptr, *rec_len = nullptr;
    const uint32_t *block_rec = nullptr; // per block: last record starting at or before it
    uint32_t nrec = 0, max_mm = 0, check_rst = 0;
    ipcr_hit_rec *hits = nullptr;
    uint64_t hcap = 0;
    unsigned long long *counts = nullptr, *next_counts = nullptr, *next_qcount = nullptr;
    // hand-over: every kernel also writes the first `pre` hit records to pub_hits (pinned host memory) as they are
    // found.  Stores of waves on different XCDs are not ordered with the last wave's sequence word -- with several
    // processes on the GPU the host saw it before 40 % of the records -- and the two 16-byte halves of a record are
    // separate stores, so EACH half carries the scan's tag (seq in bits 40..63 of pos; slot | seq << 32 in the upper
    // mask word, which is zero for patterns <= 64 nt) and is one dwordx4 store; the host takes a record only when both
    // tags are there, waits for stragglers and strips the tags.  In the last kernel of a scan (pub != null) the last wave to finish writes the counter set to
    // pub[0..3] and then `seq` to *pub_seq (one wave, system fence in between).
    // tickets: 65 zeroed counters, 32 words apart, left zeroed again (+ one statistics counter next to each).
    uint32_t *tickets = nullptr;
    unsigned long long *pub = nullptr;
    ipcr_hit_rec *pub_hits = nullptr;
    uint32_t pre = 0;
    uint32_t *pub_seq = nullptr;
    uint32_t seq = 0;
    uint32_t withhold = 0; // tests: slot + 1 of a record published with a stale tag in its first half (a torn record)
};
// blocks [block0, block0 + nblocks) of the tiles
hipError_t jit_launch(JitFilter *f, hipStream_t st, const uint32_t *planes, uint64_t block0, uint64_t nblocks, void *queue,
                      uint64_t qcap, unsigned long long *qcount, const JitVerify &v, hipEvent_t start, hipEvent_t stop);
// launches that took the form for small launches (a block shared by several waves) so far, in this process: tests
uint64_t jit_small_launches();
// seed-index filter for large panels, with the panel's key shapes baked in (host.cpp: build_index)
struct IndexGeom {
    int tail_rows = 0;     // bases of history a window can reach behind the newest base (longest pattern - 1)
    bool all_acgt = false; // no indexed pattern holds an IUPAC code (the exact check then needs half an entry)
    int uniform_len = 0;   // every indexed pattern has this length (0: mixed); shifts and masks of the check become constants
    int dl = 0;            // left-anchored windows are tested dl bases after their start (one value for the panel)
    uint32_t table_entries = 0; // entries of the panel's table: chained patterns of a key go back into the queue when an entry index fits the bits a queue entry has for it
};
std::string jit_index_source(const std::vector<ipcr_index_shape> &shapes, const IndexGeom &geom);
// can a panel of n_shapes shapes (all "3 protected bases + 5 block bases") whose windows reach tail_rows bases back take
// the two-steps-per-lookup tables?  (host.cpp: build_index asks before it lays the tables out; IPCR_INDEX_TWO_STEP=0: no)
bool jit_index_pairable(size_t n_shapes, int tail_rows);
// bytes of the LDS image of a set of shapes (tables + rank prefixes + constants)
unsigned jit_index_image_bytes(const std::vector<ipcr_index_shape> &shapes);
JitFilter *jit_build_index(const std::vector<ipcr_index_shape> &shapes, const IndexGeom &geom, std::string &err);
hipError_t jit_launch_index(JitFilter *f, hipStream_t st, const uint32_t *planes, uint64_t block0, uint64_t nblocks, uint32_t nshapes,
                            const uint32_t *lds_image, const void *table, uint32_t max_mm, void *queue,
                            uint64_t qcap, unsigned long long *qcount, uint32_t *work, hipEvent_t start, hipEvent_t stop,
                            const JitVerify *fused = nullptr);
// work: two zeroed counters 128 B apart (unit counter, leavers); the kernel leaves them zeroed again
// fused: the index kernel writes the hit records itself and its last wave hands the counters to the host (as the specialised
// filter does: JitVerify; tickets / withhold unused) -- for panels whose every pattern the index serves (no leftovers) and
// when jit_index_fusable() (the dynamic unit hand-out is on: the last wave to leave is what publishes)
bool jit_index_fusable();
void jit_destroy(JitFilter *f);

} // namespace ipcr
