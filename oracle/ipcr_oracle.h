/*
 * ipcr_oracle.h -- CPU restatement of ipcr's primer matcher (TEST INFRASTRUCTURE ONLY).
 *
 * This library is the parity checker for the HIP path.  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may load it; the product
 * (ipcr_amd/) never links, imports or calls anything under oracle/.
 *
 * Every function cites the reference file:line (relative to the ipcr checkout)
 * whose behaviour it restates.  The reference is Go and there is no Go
 * toolchain in the build container, so the oracle is pinned by the reference's
 * own literal known-answer tests (see tests/test_oracle_golden.py), not by
 * running the reference.
 */
#ifndef IPCR_ORACLE_H
#define IPCR_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* engine.Config -- core/engine/engine.go:10-19 (NeedSites is presentation only) */
typedef struct {
    int32_t max_mm;
    int32_t terminal_window;
    int32_t min_len;
    int32_t max_len;
    int32_t hit_cap;
    int32_t seed_len;
    int32_t circular;
} or_config;

/* primer.Match -- core/primer/match.go:8-13 ; idx points into an owned array */
typedef struct {
    int32_t pos;
    int32_t mm;
    int32_t len;
    int32_t nidx;
    int32_t *idx;
} or_match;

typedef struct {
    or_match *v;
    int32_t n;
    int32_t cap;
} or_matches;

/* engine.Product -- core/engine/product.go:4-35 (scan-relevant fields only) */
typedef struct {
    int32_t pair;      /* index into the pair list (ExperimentID = pairs[pair].ID) */
    int32_t start;
    int32_t end;
    int32_t length;
    int32_t type;      /* 0 = "forward", 1 = "revcomp" */
    int32_t fwd_mm;
    int32_t rev_mm;
    int32_t nf;        /* len(FwdMismatchIdx) */
    int32_t nr;        /* len(RevMismatchIdx) */
    int32_t *fidx;
    int32_t *ridx;
} or_product;

typedef struct {
    or_product *v;
    int32_t n;
    int32_t cap;
} or_products;

typedef struct or_panel or_panel; /* engine.CompiledPanel -- core/engine/compiled.go:77-91 */

/* ---- core/primer ---- */
uint8_t or_iupac_mask(uint8_t c);                       /* iupac.go:6-58 */
int or_base_match(uint8_t g, uint8_t p);                /* iupac.go:62-67 */
int or_revcomp(const uint8_t *in, int n, uint8_t *out); /* rc.go:37-56 ; 0 ok, else 1-based bad position */
int or_mismatch_count(const uint8_t *g, const uint8_t *p, int n); /* mismatch.go:4-15 */
void or_find_matches(const uint8_t *seq, int n, const uint8_t *primer, int pl,
                     int max_mm, int cap_hits, int tw, or_matches *out); /* match.go:30-90 */
void or_matches_free(or_matches *m);
/* engine.go:70-93 on bare positions (match_search_test.go:8-31): sorted_out gets (pos, input index) pairs */
void or_sort_and_bounds(const int32_t *pos, int n, int query, int32_t *sorted_out, int *lo, int *hi);
/* core/primer/validate.go:12-37: normalised length, 0 = empty, -(1-based position) = unsupported character */
int or_validate_primer(const char *raw, char *out, int cap);

/* ---- core/engine ---- */
void or_products_free(or_products *p);
/* bruteforce.go:11-38 */
void or_simulate_bruteforce(const or_config *cfg, const uint8_t *seq, int n,
                            int npairs, const char *const *fwd, const char *const *rev,
                            const int32_t *minp, const int32_t *maxp, or_products *out);
/* compiled.go:96-136 */
or_panel *or_panel_create(const or_config *cfg, int npairs, const char *const *fwd,
                          const char *const *rev, const int32_t *minp, const int32_t *maxp);
void or_panel_free(or_panel *p);
int or_panel_have(const or_panel *p, int pair, char which);  /* compiled.go:35-37 */
int or_panel_num_seed_patterns(const or_panel *p);
int or_panel_num_nodes(const or_panel *p);
int or_panel_seed_pattern(const or_panel *p, int i, char *pat_out, int cap, int *npayloads);
/* compiled.go:162-267 (production path: AC seeds + halo + fallback + join) */
void or_panel_scan(const or_panel *p, const uint8_t *seq, int n, or_products *out);
/* per-orientation matches of the production path, before the join (for hit-level parity) */
void or_panel_scan_matches(const or_panel *p, const uint8_t *seq, int n,
                           int pair, char which, or_matches *out);

/* ac.go:151-180 over an ad-hoc pattern list: writes (endPos, patternIdx) pairs */
int or_ac_scan(int npat, const char *const *pats, const uint8_t *seq, int n,
               int32_t *out_pairs, int cap_pairs);
/* halo.go:9-24 */
int or_non_acgt_ranges(const uint8_t *seq, int n, int32_t *out_pairs, int cap_pairs);
/* halo.go:26-74 */
int or_halo_starts(int seq_len, int primer_len, int nranges, const int32_t *ranges,
                   int32_t *out, int cap);
/* seed.go:152-231 : number of unique seed patterns for a pair list */
int or_build_seed_patterns_count(int npairs, const char *const *fwd, const char *const *rev,
                                 int seed_len, int tw, int max_mm);

/* ---- core/oligo ---- oligo.go:19-77 ; strand '+','-' or 0 when not found */
typedef struct {
    int32_t found;
    int32_t strand;
    int32_t pos;
    int32_t mm;
} or_hit;
or_hit or_best_hit(const uint8_t *amplicon, int n, const char *probe, int max_mm);

/* ---- fixtures: core/engine/performance_benchmark_test.go:20-106 ---- */
void or_bench_dna(uint8_t *out, int64_t n, uint32_t seed);  /* :67-76 */
void or_bench_primer(int idx, int n, char *out);            /* :78-93 (out gets n+1 bytes) */
uint8_t or_different_base(uint8_t b);                       /* :95-106 */
/* :25-65 ; returns actual genome length; seq must hold max(genome_len, 256+pairs*256+180) */
int64_t or_make_bench_fixture(int pair_count, int64_t genome_len, int mutate_forward,
                              int reference_n, uint8_t *seq, char *fwd_out, char *rev_out);

/* ---- CPU baseline: internal/pipeline/pipeline.go:60-125 worker pool over rolling chunks
 * (core/fasta/path_ctx.go:83-179), one or_panel_scan per chunk; returns product count
 * after the collector's (base,start,end,type,exp) de-dup (pipeline.go:140-149, exact set
 * instead of the bounded LRU). */
int64_t or_baseline_scan_mt(const or_panel *p, const uint8_t *seq, int64_t n,
                            int chunk_size, int overlap, int threads);
/* the same with `passes` passes over the record queued to one worker pool; *busy = workers that got a chunk */
int64_t or_baseline_scan_pool(const or_panel *p, const uint8_t *seq, int64_t n, int chunk_size, int overlap,
                              int threads, int passes, int *busy, int *nchunks_out);

#ifdef __cplusplus
}
#endif
#endif
