/*
 * ipcr_hip.h -- C ABI of the MI355X-native ipcr primer matcher (libipcr_hip.so).
 *
 * This is the drop-in boundary for ipcr's seeded-scan + per-hit-verify path.  The
 * reference has no FFI today; its seam is the Go interface family in
 * internal/pipeline/sim.go:11-39, constructed at internal/appcore/core.go:108-117.
 * Every entry point below names the reference symbol it replaces; INTEGRATION.md
 * shows the cgo shim that binds them behind pipeline.StreamingCompiledSimulator.
 *
 * Conventions: plain pointers and sizes only; every function returns an
 * ipcr_status (0 = ok) unless stated; no pointer passed in is retained after the
 * call returns (cgo rule); all handles are opaque; errors never abort the
 * process (the reference panics; see ipcr_last_error()).  There is no CPU
 * fallback: every scan runs on the HIP device or fails with IPCR_ERR_DEVICE.
 */
#ifndef IPCR_HIP_H
#define IPCR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IPCR_ABI_VERSION 4
#define IPCR_MAX_PRIMER_LEN 128 /* longest primer/probe the device path accepts */
#define IPCR_MAX_MM 16          /* largest --mismatches the device path accepts */

typedef enum {
    IPCR_OK = 0,
    IPCR_ERR_INVALID = 1,     /* bad argument (NULL handle, negative max_mm, ...) */
    IPCR_ERR_PRIMER = 2,      /* primer/probe is not upper-case IUPAC DNA (core/primer/rc.go:27-34 panics here) */
    IPCR_ERR_UNSUPPORTED = 3, /* primer longer than IPCR_MAX_PRIMER_LEN, max_mm > IPCR_MAX_MM */
    IPCR_ERR_DEVICE = 4,      /* HIP error, no device, kernel build failure */
    IPCR_ERR_CAPACITY = 5,    /* more hits than the device buffers can hold even after regrowth */
    IPCR_ERR_ABORTED = 6      /* emit callback returned non-zero (ForEachCompiledProduct's emit error) */
} ipcr_status;

/* engine.Config -- core/engine/engine.go:10-19.  need_sites only affects presentation
 * (FwdSite/RevSite) and is ignored by the scan. */
typedef struct {
    int32_t max_mm;
    int32_t terminal_window;
    int32_t min_len;
    int32_t max_len;
    int32_t hit_cap;
    int32_t seed_len;
    int32_t circular;
    int32_t need_sites;
} ipcr_config;

/* primer.Pair -- core/primer/pair.go:4-10 */
typedef struct {
    const char *id;
    const char *forward;
    const char *reverse;
    int32_t min_product;
    int32_t max_product;
} ipcr_pair;

/* engine.Product -- core/engine/product.go:4-35 (scan-produced fields).  ExperimentID is
 * pairs[pair].id, SequenceID is the caller's id for `record`.  Mismatch indices are in
 * primer 5'->3' coordinates exactly as the reference emits them (rev_idx descending). */
typedef struct {
    int64_t start;
    int64_t end;
    int64_t length;
    int32_t pair;
    int32_t record;
    int32_t type; /* 0 = "forward", 1 = "revcomp" */
    int32_t fwd_mm;
    int32_t rev_mm;
    int32_t n_fwd_idx;
    int32_t n_rev_idx;
    uint8_t fwd_idx[IPCR_MAX_MM];
    uint8_t rev_idx[IPCR_MAX_MM];
} ipcr_product;

/* primer.Match -- core/primer/match.go:8-13, as produced on the device.  One record per
 * (distinct pattern, start).  `mm_mask` has bit j set when pattern position j mismatches
 * (ascending MismatchIdx = set bits in order).  This is the record exchanged between
 * GPUs by the RCCL all-gatherv. */
typedef struct {
    uint64_t pos;        /* 0-based start on the forward strand, record-local */
    uint32_t record;     /* record (or chunk) index inside the scanned genome */
    uint32_t pattern;    /* low 24 bits: distinct-pattern id; bit 31: seed span touched a non-ACGTacgt byte */
    uint64_t mm_mask[2];
} ipcr_hit;

/* oligo.Hit -- core/oligo/oligo.go:9-15 */
typedef struct {
    int32_t found;
    int32_t strand; /* '+' or '-' */
    int32_t pos;
    int32_t mm;
} ipcr_probe_hit;

/* per-scan measurements, filled by the last ipcr_scan_* call on a scratch */
typedef struct {
    double pack_ms;       /* ASCII -> tile pack kernel (0 when the genome was already resident) */
    double filter_ms;     /* dominant kernel: bit-sliced k-mismatch filter over the tiles (HIP events) */
    double verify_ms;     /* stand-alone verify kernel (0 when the specialised filter verified its survivors itself) */
    double total_ms;      /* host wall time of the call */
    uint64_t bases;       /* genome bases scanned */
    uint64_t tile_bytes;  /* bytes of encoded tiles the filter kernel reads once */
    uint64_t candidates;  /* windows that reached the exact verifier (in-kernel or stand-alone) */
    uint64_t hits;        /* verified primer.Match records */
    uint64_t products;
    int32_t kernel_kind;  /* 1 = panel-specialised (runtime-compiled) filter, 2 = table-driven filter, 3 = seed-index filter */
    int32_t n_patterns;
    double enqueue_ms;    /* host time to enqueue the kernels (+ copy and marker where the kernel does not publish itself) */
    double wait_ms;       /* host time waiting for the results (sequence word in pinned memory, or event / stream) */
    double sort_ms;       /* host: hit records -> (record, pattern, pos) order */
    double join_ms;       /* host: match lists + amplicon join */
    /* in-kernel hand-over (specialised filter): hit records reach pinned host memory as two tagged 16-byte halves */
    uint64_t handover_refetched;   /* scans whose records were fetched from device memory instead: a record's tags had
                                      not both arrived within 2 ms */
    uint64_t handover_checked;     /* IPCR_DEBUG_PUBLISH_CHECK=1: scans whose pinned records were compared with device memory */
    uint64_t handover_check_diffs; /* ... records that differed (must be 0) */
    /* seed-index panels: patterns the index cannot key (primers > 32 nt, too many IUPAC expansions in a key) */
    uint32_t leftover_patterns;    /* how many of the panel's patterns those are */
    uint32_t leftover_kernels;     /* specialised spill-only filters that took them (0: the table-driven kernel did) */
    double hostpack_ms;            /* ipcr_scan_chunk: host time packing the caller's ASCII into bit planes (0: the bases went over the link as ASCII) */
    uint32_t segmented;            /* 1: a capped scan with more raw matches than the hit buffer may take was repeated in position
                                      order, range of blocks by range of blocks, keeping what HitCap can use (host.cpp: scan_segmented) */
    uint32_t pattern_set;          /* 1: the rc orientations were scanned without their 5' window and the host applied it after the
                                      cap (core/engine/compiled.go:249-256: what a record with a non-ACGT byte takes -- and a chunk whose
                                      bytes the device packs, where that is not known at launch); 0: every orientation kept its window */
} ipcr_scan_stats;

typedef struct ipcr_panel ipcr_panel;     /* engine.CompiledPanel + device tables */
typedef struct ipcr_scratch ipcr_scratch; /* engine.SimulationScratch: one HIP stream + staging per worker */
typedef struct ipcr_genome ipcr_genome;   /* packed reference tiles resident in HBM */

/* ---- process / device ---- */
const char *ipcr_version(void);             /* internal/version/version.go:12-15 analogue */
const char *ipcr_last_error(void);          /* thread-local message of the last failing call */
/* Devices.  One host process may drive every GPU of a node: a scratch and a genome belong to the device they were
 * created on, a panel keeps one set of device tables and kernels per device it is scanned on (built at the first scan
 * there, under the panel's lock), and EVERY entry point selects its object's device for the calling thread itself and
 * puts the thread's previous device back -- HIP's current device is a per-thread setting, and a Go worker
 * (internal/pipeline/pipeline.go:60-125) may run on any thread.  Worker i of a pool uses device i mod N
 * (ipcr_scratch_create_on); chunks are independent, so there is no collective.
 * ipcr_set_device(d): the default device of ipcr_scratch_create / ipcr_genome_create from now on, for every thread of
 * the process (one process per GPU calls it once); it also selects d for the calling thread.  Without it the default
 * is the calling thread's current HIP device.
 * IPCR_DEVICE_SLOTS=N in the environment (tests, rehearsals on a one-GPU box): ipcr_device_count() = max(N, GPUs);
 * device d then runs on GPU d mod GPUs, with tables and kernels of its own. */
ipcr_status ipcr_set_device(int device);
int ipcr_device_count(void);                /* 0 when no HIP device is visible */
/* Moves the CALLING thread onto the CPUs next to a device (its PCI function's local_cpulist, within the process's own
 * mask); device < 0: the default device.  For the threads of a worker pool that hand sequence to ipcr_scan_chunk (the
 * workers of internal/pipeline/pipeline.go:60-125; a goroutine first calls runtime.LockOSThread): bytes packed into
 * pinned memory by a core of the other socket cross the link at 31 GB/s instead of 54.
 * Returns 1 when the thread was moved, 0 when there is nothing to choose (one socket, no such list, IPCR_BIND_THREADS=0).
 * The library's own threads (FASTA reader, pack pool) do this themselves. */
int ipcr_bind_thread_to_device(int device);

/* ---- core/primer helpers used by callers of the path ---- */
uint8_t ipcr_iupac_mask(uint8_t c);                                  /* core/primer/iupac.go:6-58 */
int ipcr_base_match(uint8_t g, uint8_t p);                           /* core/primer/iupac.go:62-67 */
ipcr_status ipcr_revcomp(const char *seq, size_t n, char *out);      /* core/primer/rc.go:37-56 (RevCompStrict) */

/* ---- engine.New + Engine.CompilePanel -- core/engine/engine.go:27, compiled.go:96-136 ---- */
ipcr_status ipcr_panel_create(const ipcr_config *cfg, const ipcr_pair *pairs, int32_t n_pairs,
                              ipcr_panel **out);
void ipcr_panel_destroy(ipcr_panel *p);
int32_t ipcr_panel_num_pairs(const ipcr_panel *p);
int32_t ipcr_panel_num_patterns(const ipcr_panel *p); /* distinct (sequence, protected side) patterns */
int32_t ipcr_panel_max_primer_len(const ipcr_panel *p);
/* compiledHas(cp.Have, pair, which) -- core/engine/compiled.go:35-37; which in 'A','B','a','b' */
int32_t ipcr_panel_have(const ipcr_panel *p, int32_t pair, char which);
/* 0 = table-driven filter only, 1 = allow the panel-specialised filter (default) */
ipcr_status ipcr_panel_set_specialize(ipcr_panel *p, int32_t enable);
/* A small panel (<= 16 distinct patterns) does not make its first scans wait for the kernel build (hiprtc, ~0.8 s): the
 * build runs on a thread of its own and the scans before it is done take the table-driven kernel -- same results,
 * ipcr_scan_stats.kernel_kind says which ran.  ipcr_panel_wait_ready blocks until the kernels of every device the panel
 * has been scanned on are built (measurements; IPCR_JIT_ASYNC=0 in the environment makes every build synchronous). */
ipcr_status ipcr_panel_wait_ready(const ipcr_panel *p);
/* devices this panel holds tables and kernels on (one per device it has been scanned on) */
int32_t ipcr_panel_device_slots(const ipcr_panel *p);
/* Pattern-axis sharding (one genome x a huge panel over several GPUs): this panel object scans only
 * every count-th distinct pattern of its scanned-pattern list, starting at `index` (the orientations of a pair are
 * independent until the per-pair join, core/engine/compiled.go:192-207,260-265).  Call before the first scan.  The
 * hits keep the panel-wide pattern ids, so the hit lists of all shards over the SAME records, concatenated (e.g. by
 * the all-gatherv), joined with ipcr_join_hits give exactly the products of the unsharded scan. */
ipcr_status ipcr_panel_set_shard(ipcr_panel *p, int32_t index, int32_t count);
/* distinct-pattern ids this panel object scans in `mode` (as in ipcr_panel_filter_source: 0 / 1), ascending; returns
 * their number, fills at most cap */
int32_t ipcr_panel_scanned_patterns(const ipcr_panel *p, int32_t mode, int32_t *out, int32_t cap);
/* HIP source of the panel-specialised filter kernel (what hiprtc compiles at first scan);
 * mode 0 = records without non-ACGT bytes, 1 = with; 2 / 3 = the seed-index filter's source (the
 * kernel large panels use) for mode 0 / 1.  Writes at most cap bytes (NUL-terminated),
 * *needed = full length + 1; an empty string means the panel is not specialisable (too many
 * pattern groups).  Large panels are cut into groups of patterns, one kernel each; this
 * returns the first group's source. */
ipcr_status ipcr_panel_filter_source(const ipcr_panel *p, int32_t mode, char *out, size_t cap, size_t *needed);

/* introspection of the compiled panel: the distinct patterns the device scans.
 * A pattern is one primer orientation string plus which end carries the protected terminal
 * window and how many of its bases the device enforces (tw_dev; 0 = the host filters).
 * seed_off/seed_len: the span the reference would seed (core/engine/seed.go:260-283), 0 length
 * when it would leave the orientation unseeded. */
int32_t ipcr_panel_num_patterns_total(const ipcr_panel *p);
ipcr_status ipcr_panel_pattern_info(const ipcr_panel *p, int32_t pattern, char *seq_out, size_t cap,
                                    int32_t *left_window, int32_t *tw_dev, int32_t *seed_off,
                                    int32_t *seed_len);
/* pattern id scanned for orientation `which` of `pair`; mode as in ipcr_panel_filter_source */
int32_t ipcr_panel_slot_pattern(const ipcr_panel *p, int32_t pair, char which, int32_t mode);

/* ---- Engine.NewSimulationScratch -- core/engine/hit_collect.go:21-34 ---- */
ipcr_status ipcr_scratch_create(const ipcr_panel *p, ipcr_scratch **out);          /* on the default device */
ipcr_status ipcr_scratch_create_on(const ipcr_panel *p, int32_t device, ipcr_scratch **out);
int32_t ipcr_scratch_device(const ipcr_scratch *s);                                  /* -1 for a host-only scratch */
/* host-only scratch: holds results of ipcr_join_hits, owns no device resources; every
 * ipcr_scan_* call on it fails with IPCR_ERR_DEVICE */
ipcr_status ipcr_scratch_create_host(const ipcr_panel *p, ipcr_scratch **out);
void ipcr_scratch_destroy(ipcr_scratch *s);
ipcr_status ipcr_scratch_stats(const ipcr_scratch *s, ipcr_scan_stats *out);
/* results of the last scan on this scratch; pointers stay valid until the next scan/destroy */
ipcr_status ipcr_scratch_products(const ipcr_scratch *s, const ipcr_product **out, int64_t *n);
ipcr_status ipcr_scratch_hits(const ipcr_scratch *s, const ipcr_hit **out, int64_t *n);
/* the same hit records where the kernels left them in device memory, for a device-to-device exchange
 * (RCCL all-gather straight out of this buffer, no host staging): *dev_block points at a 64-byte
 * header followed by `capacity` ipcr_hit slots, the first *n_hits of them valid, in device append
 * order (unsorted, and a window found through several keys may appear twice: ipcr_join_hits sorts
 * and removes duplicates).  Header = two sets of four uint64 {queue words, hits, candidate windows,
 * fullest queue segment}; the set the last scan used is the non-zero one.  Valid until the next scan
 * on this scratch; the address changes when the buffer regrows.  IPCR_ERR_UNSUPPORTED after a scan that ran in segments
 * (ipcr_scan_stats.segmented: a capped scan with more raw matches than the buffer may take): the device buffer then holds the
 * last range only and the hits live in the host list (ipcr_scratch_hits); ipcr_exchange_begin sends them from there. */
ipcr_status ipcr_scratch_device_hits(const ipcr_scratch *s, const void **dev_block, uint64_t *n_hits, uint64_t *capacity);

/* ---- Engine.ForEachCompiledProduct / SimulateCompiledWithScratch -- compiled.go:141-267 ----
 * One record or chunk of upper-cased ASCII (host memory), chunk-local coordinates, products
 * in the reference's emission order.  emit may be NULL (fetch with ipcr_scratch_products);
 * a non-zero return from emit aborts and yields IPCR_ERR_ABORTED. */
typedef int (*ipcr_emit_fn)(const ipcr_product *product, void *user);
ipcr_status ipcr_scan_chunk(const ipcr_panel *p, ipcr_scratch *s, const uint8_t *seq, uint64_t len,
                            ipcr_emit_fn emit, void *user);

/* ASCII bases -> the linear bit planes ipcr_scan_chunk sends over the link (bit i of word w = base 32 w + i; lo, hi = 2-bit
 * code A 0 C 1 G 2 T 3, zero for other bytes; inv = not an upper-case ACGT, core/primer/iupac.go:62-67; rst = outside
 * ACGTacgt, core/engine/ac.go:16-30).  padded_bases: a multiple of 32 >= len; bases past len are inv 1, rst 0.  Every
 * array takes padded_bases / 32 words.  *flags: bit 0 = some byte lies outside ACGTacgt, bit 1 = some lower-case acgt. */
ipcr_status ipcr_pack_ascii(const uint8_t *seq, uint64_t len, uint64_t padded_bases, uint32_t *lo, uint32_t *hi,
                            uint32_t *inv, uint32_t *rst, uint32_t *flags);

/* ---- resident genome: many records packed once, scanned by any panel ---- */
ipcr_status ipcr_genome_create(uint64_t capacity_bases, uint32_t max_records, ipcr_genome **out); /* on the default device */
ipcr_status ipcr_genome_create_on(uint64_t capacity_bases, uint32_t max_records, int32_t device, ipcr_genome **out);
int32_t ipcr_genome_device(const ipcr_genome *g);
void ipcr_genome_destroy(ipcr_genome *g);
/* append one record: ASCII in host memory, or in device memory (16-byte aligned) */
ipcr_status ipcr_genome_add_record(ipcr_genome *g, const uint8_t *seq, uint64_t len);
ipcr_status ipcr_genome_add_record_device(ipcr_genome *g, const void *dev_seq, uint64_t len);
/* fill a device buffer with the reference's benchDNA LCG stream, generated on the device with
 * jump-ahead (core/engine/performance_benchmark_test.go:67-76); bit-identical to bases
 * [stream_offset, stream_offset + len) of the serial loop started from `seed` */
ipcr_status ipcr_lcg_fill_device(void *dev_out, uint64_t len, uint32_t seed, uint64_t stream_offset);
/* copy bases of a resident record back to the host (decoded from the tiles; invalid -> 'N') */
ipcr_status ipcr_genome_read(const ipcr_genome *g, uint32_t record, uint64_t pos, uint8_t *out, uint64_t len);
uint32_t ipcr_genome_num_records(const ipcr_genome *g);
uint64_t ipcr_genome_record_len(const ipcr_genome *g, uint32_t record);
/* ID of a record loaded by ipcr_genome_add_fasta (header text up to the first blank,
 * core/fasta/stream.go:125-131); "" for records added as bytes.  Valid until the genome changes. */
const char *ipcr_genome_record_id(const ipcr_genome *g, uint32_t record);
uint64_t ipcr_genome_total_bases(const ipcr_genome *g);
uint64_t ipcr_genome_tile_bytes(const ipcr_genome *g);
double ipcr_genome_pack_ms(const ipcr_genome *g); /* accumulated pack-kernel time */

/* ---- FASTA record / rolling-chunk stream -- core/fasta/path_ctx.go:19-179 ----
 * open: gzip by content, "-" = stdin (open.go:29-50).  chunk_size <= 0 or chunk_size <= overlap
 * streams whole records; else windows "id:start-end" advancing by chunk_size - overlap.
 * Sequence is TrimSpace'd per line and a-z upper-cased (normalize.go:5-14). */
typedef struct ipcr_fasta ipcr_fasta;
ipcr_status ipcr_fasta_open(const char *path, int64_t chunk_size, int64_t overlap, ipcr_fasta **out);
void ipcr_fasta_close(ipcr_fasta *f);
/* next record/chunk; *got = 0 at end of input; *id and *seq stay valid until the next call */
ipcr_status ipcr_fasta_next(ipcr_fasta *f, const char **id, const uint8_t **seq, uint64_t *len, int32_t *got);
/* pack every record of a FASTA file (plain or gzip, "-" = stdin) into a resident genome (same record semantics as the
 * stream above); record IDs come back '\n'-joined and via ipcr_genome_record_id.  Two loaders behind it: a plain file whose
 * records are lines of one width is mapped, packed on the host (2 bits per base) and written straight into device memory through
 * the PCIe BAR where the host has a large one (IPCR_FASTA_HOSTPACK=0: never); every other file, and every host without, sends
 * raw slabs to the device, which strips line ends / white space and folds case itself.  The mapped file must not be truncated
 * while it is loaded. */
ipcr_status ipcr_genome_add_fasta(ipcr_genome *g, const char *path, uint32_t *n_added, char *ids_out, size_t cap,
                                  size_t *ids_needed);

/* scan every record of a resident genome with one launch; products carry `record`.  Scratch and genome must live on
 * the same device (IPCR_ERR_INVALID otherwise) */
ipcr_status ipcr_scan_genome(const ipcr_panel *p, ipcr_scratch *s, const ipcr_genome *g,
                             ipcr_emit_fn emit, void *user);
/* split form of ipcr_scan_genome, for pipelining as the reference's worker pool + collector do
 * (internal/pipeline/pipeline.go:60-161): begin() enqueues the kernels and the read-back on the
 * scratch's stream and returns at once; end() waits, rebuilds the match lists and joins.  The
 * scratch must not be used in between; other scratches may scan (and be joined) meanwhile. */
ipcr_status ipcr_scan_genome_begin(const ipcr_panel *p, ipcr_scratch *s, const ipcr_genome *g);

/* ---- a resident genome scanned the way the pipeline scans it under --chunk-size ----
 * The reference cuts every record into rolling windows (core/fasta/path_ctx.go:83-179: windows of chunk_size bases, step
 * chunk_size - overlap; a record that never fills a window goes whole, under its own ID) and every window is ONE
 * Engine.ForEachCompiledProduct call: HitCap, the reset-byte rules (core/engine/compiled.go:185-190) and the join apply per
 * window, coordinates are window-local (internal/pipeline/pipeline.go:60-161 puts them back).  Here the tiles are swept ONCE;
 * the hits of every window are then taken out of the record's list (a window's hits lie wholly inside it), the windows of a
 * record with a reset byte are asked on the device whether they hold one, and every window is joined as its own call.
 * Products: `record` = index into the window list (ipcr_scratch_chunk_windows), start / end window-local.
 * IPCR_ERR_INVALID for a circular panel (internal/runutil/runutil.go:46-49 disables chunking there); IPCR_ERR_UNSUPPORTED when a
 * capped scan had to run in segments (ipcr_scan_stats.segmented: the device kept per RECORD what HitCap can use -- the caller
 * streams the chunks through ipcr_scan_chunk instead). */
typedef struct {
    uint32_t record;   /* record of the genome */
    uint32_t plain;    /* 1: the whole record under its own ID (no window was ever emitted); 0: ID "id:start-end" */
    uint64_t start, end; /* [start, end) in the record */
    uint32_t reset;    /* the window holds a byte outside ACGTacgt */
    uint32_t reserved0;
} ipcr_chunk_window;
ipcr_status ipcr_scan_genome_chunked(const ipcr_panel *p, ipcr_scratch *s, const ipcr_genome *g, int64_t chunk_size, int64_t overlap,
                                     ipcr_emit_fn emit, void *user);
/* the windows of the last ipcr_scan_genome_chunked on this scratch, in order; valid until its next scan */
ipcr_status ipcr_scratch_chunk_windows(const ipcr_scratch *s, const ipcr_chunk_window **out, int64_t *n);
/* the windows of ONE record of `len` bases (record, reset left 0): what the streaming reader (ipcr_fasta_next) emits for it;
 * *n = how many there are (out may be null or shorter: the first `cap` are written) */
ipcr_status ipcr_chunk_windows(uint64_t len, int64_t chunk_size, int64_t overlap, ipcr_chunk_window *out, int64_t cap, int64_t *n);

/* order two pipelined scans on the device: the next scan begun on `s` runs its filter sweep
 * directly after the filter sweep of the scan most recently begun on `prev` (the two sweeps share
 * one in-order stream, so they never compete for HBM); prev's verify kernel, read-back and
 * host-side join overlap it.  Call after prev's begin() and before s's begin(); scratches of a
 * chain should be destroyed only after their scans have ended. */
ipcr_status ipcr_scratch_chain_after(ipcr_scratch *s, const ipcr_scratch *prev);
ipcr_status ipcr_scan_genome_end(const ipcr_panel *p, ipcr_scratch *s, const ipcr_genome *g,
                                 ipcr_emit_fn emit, void *user);
/* scan only (rows 10-15 of SURVEY section 8a): verified hits, no join */
ipcr_status ipcr_scan_genome_hits(const ipcr_panel *p, ipcr_scratch *s, const ipcr_genome *g);
/* join step alone (core/engine/engine.go:108-404) over an arbitrary hit list, e.g. the
 * all-gathered hits of several GPUs; record_len[r] is the length of record r,
 * record_flags[r] bit0 = record contains a non-ACGTacgt byte (may be NULL = none). */
ipcr_status ipcr_join_hits(const ipcr_panel *p, ipcr_scratch *s, const ipcr_hit *hits, int64_t n_hits,
                           const uint64_t *record_len, const uint8_t *record_flags, uint32_t n_records,
                           ipcr_emit_fn emit, void *user);
uint8_t ipcr_genome_record_flags(const ipcr_genome *g, uint32_t record);

/* ---- several GPUs, one process each: the all-gatherv of hit records (RCCL over xGMI) ----
 * The reference has no distributed mode; its unit of parallelism is the independent record / chunk
 * (internal/pipeline/pipeline.go:60-125).  Every rank scans its own records with the whole panel (or, ipcr_panel_set_shard,
 * its share of the patterns over the same records) -- no data-path collective -- and only the hit records are
 * exchanged: ncclAllGather straight out of every scratch's DEVICE hit buffer.  The gathered list goes to ipcr_join_hits.
 * (One process driving several devices needs none of this: the hits of all its scratches are in host memory already.)
 *   rank 0: ipcr_exchange_unique_id -> send the 128 bytes to every rank over the host's own channel
 *   every rank: ipcr_exchange_create (collective), ipcr_exchange_set_records (collective) once the genome is loaded
 *   per pass: scan on scratch s (ipcr_scan_genome_hits / _end) -> ipcr_exchange_begin(x, s, &t) -> ... -> ipcr_exchange_end
 * Up to two exchanges may be in flight; end them in the order they began (an overflow redo of one does not disturb the other).  The scratch's next scan must not begin
 * before its exchange has ended (the all-gather reads the scratch's hit buffer).  Lock step: a rank with more hits
 * than the capacity still enters the collective; every rank then sees the same counts, and all regrow and repeat that
 * exchange together inside ipcr_exchange_end (ipcr_exchange_redone counts those). */
#define IPCR_EXCHANGE_ID_BYTES 128
typedef struct ipcr_exchange ipcr_exchange;
/* 1 when ipcr_exchange_create can reach ncclCommInitRank on this rank: librccl opens and has the entry points, the device
 * exists (ipcr_last_error says why not).  Local, no side effect on the job.  Take the MINIMUM over all ranks on the host's own
 * channel before ANY rank calls ipcr_exchange_create: a rank that fails early while the others are already inside
 * ncclCommInitRank leaves them waiting for ever. */
int32_t ipcr_exchange_available(int32_t device);
ipcr_status ipcr_exchange_unique_id(uint8_t *id_out /* IPCR_EXCHANGE_ID_BYTES */);
/* same_records: every rank scanned the SAME records (pattern shards): record indices are not rebased */
ipcr_status ipcr_exchange_create(const uint8_t *id, int32_t world, int32_t rank, int32_t device, uint64_t cap_hits,
                                 int32_t same_records, ipcr_exchange **out);
void ipcr_exchange_destroy(ipcr_exchange *x);
ipcr_status ipcr_exchange_set_records(ipcr_exchange *x, uint32_t n_local_records);            /* collective */
ipcr_status ipcr_exchange_set_record_counts(ipcr_exchange *x, const uint32_t *counts /* world */); /* local: the host knows them */
ipcr_status ipcr_exchange_begin(ipcr_exchange *x, const ipcr_scratch *s, int32_t *ticket);
/* every rank's hits in rank order, `record` rebased to the job-global record index; rank r's hits are
 * [rank_hit_start[r], rank_hit_start[r + 1]), its records start at rank_record_offset[r].  Valid until the next end. */
ipcr_status ipcr_exchange_end(ipcr_exchange *x, int32_t ticket, const ipcr_hit **hits, int64_t *n_hits,
                              const uint64_t **rank_hit_start, const uint32_t **rank_record_offset);
/* The host side of ipcr_exchange_end as a pure function over a gathered buffer in HOST memory -- `world` blocks of
 * 64 + cap * 32 bytes (header: two sets of four uint64, the hit count is word 1 of the non-zero set; then cap ipcr_hit slots),
 * as ncclAllGather of every rank's ipcr_scratch_device_hits block leaves them.  rec_counts[r] = records of rank r (ignored with
 * same_records).  Writes rank r's valid records to out[rank_hit_start[r] .. rank_hit_start[r + 1]) with `record` rebased by
 * rank_record_offset[r] (both arrays: world + 1 entries, may be NULL).  *need = the largest count any rank reported.
 * IPCR_ERR_CAPACITY: some rank reported more than cap -- every rank reads the same headers, gets the same status and repeats
 * the exchange with a capacity >= *need -- or the records do not fit out_cap.  No device, no communicator. */
ipcr_status ipcr_exchange_unpack(const void *gathered, int32_t world, uint64_t cap, const uint32_t *rec_counts, int32_t same_records,
                                 ipcr_hit *out, uint64_t out_cap, uint64_t *rank_hit_start, uint32_t *rank_record_offset, uint64_t *need);
/* hit slots every rank sends: grows by itself on overflow; a host that knows what it needs says so (the SAME value on every rank) */
ipcr_status ipcr_exchange_reserve(ipcr_exchange *x, uint64_t cap_hits);
uint64_t ipcr_exchange_capacity(const ipcr_exchange *x);
uint64_t ipcr_exchange_redone(const ipcr_exchange *x);

/* ---- oligo.BestHit / probe.AnnotateAmplicon -- core/oligo/oligo.go:19-77 ----
 * amplicon in host memory; runs the probe rescan on the default device (ipcr_set_device).  Safe to call from any thread
 * next to running scans, and cheap enough to call per product as visitors.Probe.Visit does on the collector goroutine
 * (internal/visitors/probe.go:18-33): no device allocation, no copy operation, not the null stream -- the call borrows
 * a pinned block and a stream from a free list, the kernel stages the amplicon in LDS straight out of pinned memory and
 * the caller spins on the tagged 16-byte result.  (ipcr_probe_scratch_products is the batched form for a worker.) */
ipcr_status ipcr_probe_best_hit(const uint8_t *amplicon, uint64_t len, const char *probe, int32_t max_mm,
                                ipcr_probe_hit *out);
/* ipcr-probe behind the drop-in call: every product of the LAST ipcr_scan_chunk on `s`, rescanned for the probe from the
 * tiles that call packed (the scratch's private chunk genome keeps them until its next scan; no allocation in steady
 * state; the scratch's device and probe lane).  out[i] corresponds to product i of ipcr_scratch_products.  The amplicon
 * is what the pipeline would slice into Product.Seq on the worker (internal/pipeline/pipeline.go:80-89): chunk-local
 * [start, end), or record[start:] ++ record[:end] for a wrap-around product of a circular record -- so a worker
 * annotates its own chunk's products and the collector only formats (hipprobe).  _begin / ipcr_probe_products_end
 * split it as below.  IPCR_ERR_INVALID when the scratch's last scan was not an ipcr_scan_chunk. */
ipcr_status ipcr_probe_scratch_products_begin(ipcr_scratch *s, const char *probe, int32_t max_mm);
ipcr_status ipcr_probe_scratch_products(ipcr_scratch *s, const char *probe, int32_t max_mm, ipcr_probe_hit *out, int64_t n_out);
/* batched form for ipcr-probe: every product of the last scan on `s` against the resident
 * genome; out[i] corresponds to product i (internal/visitors/probe.go:18-33) */
/* the same in two halves: _begin queues the rescan of the scratch's current products on a lane of its own and returns,
 * _end waits for it and hands the results out -- the collector's work on the NEXT chunk's products can run in between
 * (internal/pipeline/pipeline.go:127-161 merges while the workers scan).  The scratch must not be scanned with between
 * the two; one rescan per scratch at a time. */
ipcr_status ipcr_probe_products_begin(ipcr_scratch *s, const ipcr_genome *g, const char *probe, int32_t max_mm);
ipcr_status ipcr_probe_products_end(ipcr_scratch *s, ipcr_probe_hit *out, int64_t n_out);
ipcr_status ipcr_probe_products(ipcr_scratch *s, const ipcr_genome *g, const char *probe, int32_t max_mm,
                                ipcr_probe_hit *out, int64_t n_out);

/* ---- nested PCR: visitors.Nested.Visit -- internal/visitors/nested.go:17-66 ----
 * The reference builds a new engine and scans the amplicon of every outer product on the collector
 * goroutine.  Batched form: the amplicons of all windows are gathered on the device, packed as the
 * records of a scratch-private genome and scanned with the compiled inner panel in one launch; per
 * window the best inner product is chosen by the reference's rule (fewest total mismatches, longest,
 * leftmost start, end, pair ID).  Coordinates of the inner product are relative to the amplicon.
 * A window with start > end is an origin-spanning amplicon (record[start:] ++ record[:end]). */
typedef struct ipcr_window {
    int64_t start, end;
    int32_t record;
    int32_t reserved;
} ipcr_window;
typedef struct ipcr_nested_hit {
    int32_t found;          /* 0: no inner product in this amplicon */
    int32_t pair;           /* index into the inner panel's pairs */
    int32_t type;           /* 0 forward, 1 revcomp */
    int32_t fwd_mm, rev_mm;
    int32_t reserved;
    int64_t start, end, length;
} ipcr_nested_hit;
ipcr_status ipcr_nested_windows(const ipcr_genome *g, const ipcr_window *windows, int64_t n, const ipcr_panel *inner,
                                ipcr_scratch *inner_scratch, ipcr_nested_hit *out);
/* every product of the last scan on `outer` (out[i] <-> product i) */
ipcr_status ipcr_nested_products(const ipcr_scratch *outer, const ipcr_genome *g, const ipcr_panel *inner,
                                 ipcr_scratch *inner_scratch, ipcr_nested_hit *out, int64_t n_out);

#ifdef __cplusplus
}
#endif
#endif
