/*
 * ipcr_oracle.c -- CPU restatement of ipcr's primer matcher.  TEST INFRASTRUCTURE ONLY.
 *
 * See ipcr_oracle.h.  Each block names the reference file:line it follows
 * (paths relative to the ipcr checkout).  Plain C11, no dependencies.
 * Parity status: pinned by the reference's own known-answer tests
 * (tests/test_oracle_golden.py); the Go reference itself cannot be built here.
 */
#include "ipcr_oracle.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ utils */

static void *xmalloc(size_t n) {
    void *p = malloc(n ? n : 1);
    if (!p) { fprintf(stderr, "oracle: out of memory\n"); abort(); }
    return p;
}
static void *xrealloc(void *q, size_t n) {
    void *p = realloc(q, n ? n : 1);
    if (!p) { fprintf(stderr, "oracle: out of memory\n"); abort(); }
    return p;
}
static void *xcalloc(size_t a, size_t b) {
    void *p = calloc(a ? a : 1, b ? b : 1);
    if (!p) { fprintf(stderr, "oracle: out of memory\n"); abort(); }
    return p;
}

/* ------------------------------------------------- core/primer/iupac.go:6-67 */

static uint8_t g_mask[256];
static uint8_t g_comp[256];
static int8_t g_basecode[256];
static int g_init_done = 0;

static void or_init(void) {
    if (g_init_done) return;
    memset(g_mask, 0, sizeof g_mask);
    const uint8_t A = 1, C = 2, G = 4, T = 8;
    struct { char c; uint8_t m; } tab[] = {
        {'A', A}, {'C', C}, {'G', G}, {'T', T}, {'U', T},
        {'R', A | G}, {'Y', C | T}, {'S', C | G}, {'W', A | T}, {'K', G | T}, {'M', A | C},
        {'B', C | G | T}, {'D', A | G | T}, {'H', A | C | T}, {'V', A | C | G},
        {'N', A | C | G | T},
    };
    for (size_t i = 0; i < sizeof tab / sizeof tab[0]; i++) {
        g_mask[(uint8_t)tab[i].c] = tab[i].m;
        g_mask[(uint8_t)(tab[i].c + ('a' - 'A'))] = tab[i].m; /* iupac.go:41-57 lower-case mirrors */
    }
    /* core/primer/rc.go:8-24 : upper-case IUPAC only */
    memset(g_comp, 0, sizeof g_comp);
    const char *from = "ACGTRYSWKMBVDHN";
    const char *to = "TGCAYRSWMKVBHDN";
    for (int i = 0; from[i]; i++) g_comp[(uint8_t)from[i]] = (uint8_t)to[i];
    /* core/engine/ac.go:16-30 */
    for (int i = 0; i < 256; i++) g_basecode[i] = -1;
    g_basecode['A'] = 0; g_basecode['C'] = 1; g_basecode['G'] = 2; g_basecode['T'] = 3;
    g_basecode['a'] = 0; g_basecode['c'] = 1; g_basecode['g'] = 2; g_basecode['t'] = 3;
    g_init_done = 1;
}

__attribute__((constructor)) static void or_ctor(void) { or_init(); }

uint8_t or_iupac_mask(uint8_t c) { return g_mask[c]; }

/* iupac.go:62-67 : genome byte must be exactly upper-case A/C/G/T */
int or_base_match(uint8_t g, uint8_t p) {
    if (g != 'A' && g != 'C' && g != 'G' && g != 'T') return 0;
    return (g_mask[p] & g_mask[g]) != 0;
}

/* rc.go:37-56 */
int or_revcomp(const uint8_t *in, int n, uint8_t *out) {
    for (int i = 0; i < n; i++) {
        int src = n - 1 - i;
        uint8_t c = g_comp[in[src]];
        if (c == 0) return src + 1;
        out[i] = c;
    }
    return 0;
}

/* mismatch.go:4-15 */
int or_mismatch_count(const uint8_t *g, const uint8_t *p, int n) {
    int mm = 0;
    for (int i = 0; i < n; i++)
        if (!or_base_match(g[i], p[i])) mm++;
    return mm;
}

/* ------------------------------------------------------------ match lists */

static void matches_push(or_matches *m, int pos, int mm, int len, const int32_t *idx, int nidx) {
    if (m->n == m->cap) {
        m->cap = m->cap ? m->cap * 2 : 8;
        m->v = xrealloc(m->v, (size_t)m->cap * sizeof(or_match));
    }
    or_match *o = &m->v[m->n++];
    o->pos = pos; o->mm = mm; o->len = len; o->nidx = nidx; o->idx = NULL;
    if (nidx > 0) {
        o->idx = xmalloc((size_t)nidx * sizeof(int32_t));
        memcpy(o->idx, idx, (size_t)nidx * sizeof(int32_t));
    }
}

void or_matches_free(or_matches *m) {
    if (!m) return;
    for (int i = 0; i < m->n; i++) free(m->v[i].idx);
    free(m->v);
    m->v = NULL; m->n = 0; m->cap = 0;
}

static int is_unambiguous(const uint8_t *p, int n) { /* match.go:17-24 */
    for (int i = 0; i < n; i++)
        if (p[i] != 'A' && p[i] != 'C' && p[i] != 'G' && p[i] != 'T') return 0;
    return 1;
}

/* match.go:30-90 */
void or_find_matches(const uint8_t *seq, int n, const uint8_t *primer, int pl,
                     int max_mm, int cap_hits, int tw, or_matches *out) {
    if (pl == 0 || n < pl) return;

    if (max_mm == 0 && is_unambiguous(primer, pl)) { /* :38-53 bytes.Index loop */
        for (int i = 0; i + pl <= n; i++) {
            if (seq[i] == primer[0] && memcmp(seq + i, primer, (size_t)pl) == 0) {
                matches_push(out, i, 0, pl, NULL, 0);
                if (cap_hits > 0 && out->n >= cap_hits) break;
            }
        }
        return;
    }

    int end = n - pl;
    int cutoff = pl - tw; /* :59-65 */
    if (tw <= 0) cutoff = pl + 1;
    if (cutoff < 0) cutoff = 0;

    int32_t *idx = xmalloc((size_t)(pl + 1) * sizeof(int32_t));
    for (int pos = 0; pos <= end; pos++) {
        int mm = 0, ok = 1;
        for (int j = 0; j < pl; j++) {
            if (!or_base_match(seq[pos + j], primer[j])) {
                if (j >= cutoff) { ok = 0; break; }
                idx[mm] = j;
                mm++;
                if (mm > max_mm) { ok = 0; break; }
            }
        }
        if (!ok) continue;
        matches_push(out, pos, mm, pl, idx, mm);
        if (cap_hits > 0 && out->n >= cap_hits) break;
    }
    free(idx);
}

/* core/engine/engine.go:53-68 : in-place filter keeping order */
static void filter_left_tw(or_matches *ms, int tw) {
    if (tw <= 0) return;
    int w = 0;
    for (int i = 0; i < ms->n; i++) {
        int drop = 0;
        for (int t = 0; t < ms->v[i].nidx; t++)
            if (ms->v[i].idx[t] < tw) { drop = 1; break; }
        if (drop) { free(ms->v[i].idx); continue; }
        ms->v[w++] = ms->v[i];
    }
    ms->n = w;
}

/* ----------------------------------------------------------- product lists */

static void products_push(or_products *ps, int pair, int start, int end, int length, int type,
                          const or_match *fm, const or_match *rm, int rlen) {
    if (ps->n == ps->cap) {
        ps->cap = ps->cap ? ps->cap * 2 : 8;
        ps->v = xrealloc(ps->v, (size_t)ps->cap * sizeof(or_product));
    }
    or_product *p = &ps->v[ps->n++];
    p->pair = pair; p->start = start; p->end = end; p->length = length; p->type = type;
    p->fwd_mm = fm->mm; p->rev_mm = rm->mm;
    p->nf = fm->nidx; p->nr = rm->nidx;
    p->fidx = NULL; p->ridx = NULL;
    if (p->nf > 0) {
        p->fidx = xmalloc((size_t)p->nf * sizeof(int32_t));
        memcpy(p->fidx, fm->idx, (size_t)p->nf * sizeof(int32_t));
    }
    if (p->nr > 0) { /* engine.go:127-136 flip: n-1-v in original order */
        p->ridx = xmalloc((size_t)p->nr * sizeof(int32_t));
        for (int i = 0; i < p->nr; i++) p->ridx[i] = rlen - 1 - rm->idx[i];
    }
}

void or_products_free(or_products *ps) {
    if (!ps) return;
    for (int i = 0; i < ps->n; i++) { free(ps->v[i].fidx); free(ps->v[i].ridx); }
    free(ps->v);
    ps->v = NULL; ps->n = 0; ps->cap = 0;
}

/* engine.go:70-85 : stable sort by Pos unless already sorted */
static void sort_matches_by_pos(or_matches *ms) {
    int sorted = 1;
    for (int i = 1; i < ms->n; i++)
        if (ms->v[i].pos < ms->v[i - 1].pos) { sorted = 0; break; }
    if (sorted) return;
    for (int i = 1; i < ms->n; i++) { /* insertion sort = stable */
        or_match key = ms->v[i];
        int j = i - 1;
        while (j >= 0 && ms->v[j].pos > key.pos) { ms->v[j + 1] = ms->v[j]; j--; }
        ms->v[j + 1] = key;
    }
}

/* engine.go:87-93 */
static int lower_bound_pos(const or_matches *ms, int pos) {
    int lo = 0, hi = ms->n;
    while (lo < hi) { int mid = lo + (hi - lo) / 2; if (ms->v[mid].pos >= pos) hi = mid; else lo = mid + 1; }
    return lo;
}
static int upper_bound_pos(const or_matches *ms, int pos) {
    int lo = 0, hi = ms->n;
    while (lo < hi) { int mid = lo + (hi - lo) / 2; if (ms->v[mid].pos > pos) hi = mid; else lo = mid + 1; }
    return lo;
}

/* test hook for core/engine/match_search_test.go:8-31: sortMatchesByPos + lower/upperBoundMatchPos on bare positions */
void or_sort_and_bounds(const int32_t *pos, int n, int query, int32_t *sorted_out, int *lo, int *hi) {
    or_matches ms;
    ms.v = (or_match *)calloc((size_t)(n > 0 ? n : 1), sizeof(or_match));
    ms.n = n; ms.cap = n;
    for (int i = 0; i < n; i++) { ms.v[i].pos = pos[i]; ms.v[i].mm = i; } /* mm carries the input order: stability is visible */
    sort_matches_by_pos(&ms);
    for (int i = 0; i < n; i++) { sorted_out[2 * i] = ms.v[i].pos; sorted_out[2 * i + 1] = ms.v[i].mm; }
    *lo = lower_bound_pos(&ms, query);
    *hi = upper_bound_pos(&ms, query);
    free(ms.v);
}

/* core/primer/validate.go:12-37 : Normalize (drop white space and quotes, upper-case) then accept only
 * ACGTRYSWKMBDHVN ('U' is rejected at the input boundary).  Returns the normalised length (>= 1), 0 for an empty
 * primer, or -(1-based position) of the first unsupported character.  ASCII input (the reference walks runes:
 * multi-byte white space / letters do not occur in primer TSVs and are reported as unsupported here). */
int or_validate_primer(const char *raw, char *out, int cap) {
    int n = 0;
    for (const unsigned char *q = (const unsigned char *)raw; *q; ++q) {
        unsigned char ch = *q;
        if (ch == ' ' || (ch >= 9 && ch <= 13) || ch == '\'' || ch == '"') continue; /* unicode.IsSpace on ASCII: \t \n \v \f \r space */
        if (ch >= 'a' && ch <= 'z') ch = (unsigned char)(ch - 32);
        if (n + 1 < cap) out[n] = (char)ch;
        n++;
    }
    if (n < cap) out[n] = 0; else if (cap > 0) out[cap - 1] = 0;
    if (n == 0) return 0;
    for (int i = 0; i < n && i + 1 < cap; i++)
        if (!strchr("ACGTRYSWKMBDHVN", out[i]) || out[i] == 0) return -(i + 1);
    return n;
}

/* One direction of engine.go:108-404.  `left` are the forward-strand hits of the
 * left primer (fwdA, or fwdB), `right` the rc hits of the other primer (revB, or
 * revA); rlen is the right primer length; type 0 forward / 1 revcomp. */
static void join_direction(const or_config *cfg, int seqlen, int pair, int minL, int maxL,
                           const or_matches *left, or_matches *right, int rlen, int type,
                           or_products *out) {
    sort_matches_by_pos(right); /* engine.go:144 / :275 */
    for (int a = 0; a < left->n; a++) {
        const or_match *ma = &left->v[a];
        int last = seqlen - rlen;
        int lo = ma->pos + 1; /* :147-156 */
        if (minL > 0) {
            lo = ma->pos + minL - rlen;
            if (lo <= ma->pos) lo = ma->pos + 1;
        }
        if (lo < 0) lo = 0;
        int hi = last; /* :157-163 */
        if (maxL > 0) {
            hi = ma->pos + maxL - rlen;
            if (hi > last) hi = last;
        }
        if (hi >= lo) {
            int iMin = lower_bound_pos(right, lo);
            int iMax = upper_bound_pos(right, hi) - 1;
            for (int j = iMax; j >= iMin; j--) { /* :169 descending */
                const or_match *mb = &right->v[j];
                int end = mb->pos + rlen;
                int length = end - ma->pos;
                if ((minL != 0 && length < minL) || (maxL != 0 && length > maxL)) continue;
                products_push(out, pair, ma->pos, end, length, type, ma, mb, rlen);
            }
        }
        if (cfg->circular) { /* :208-271 */
            int X = seqlen - ma->pos;
            int loWrap = 0;
            if (minL > 0) {
                int needed = minL - X - rlen;
                if (needed < 0) needed = 0;
                loWrap = needed;
            }
            int hiWrap = ma->pos - 1;
            if (maxL > 0) {
                int allowed = maxL - X - rlen;
                if (allowed < hiWrap) hiWrap = allowed;
            }
            if (hiWrap >= loWrap) {
                int iMinW = lower_bound_pos(right, loWrap);
                int iMaxW = upper_bound_pos(right, hiWrap) - 1;
                for (int j = iMaxW; j >= iMinW; j--) {
                    const or_match *mb = &right->v[j];
                    if (mb->pos >= ma->pos) continue;
                    int end = mb->pos + rlen;
                    int length = (seqlen - ma->pos) + end;
                    if ((minL != 0 && length < minL) || (maxL != 0 && length > maxL)) continue;
                    products_push(out, pair, ma->pos, end, length, type, ma, mb, rlen);
                }
            }
        }
    }
}

/* engine.go:108-404 */
static void join_pair(const or_config *cfg, int seqlen, int pair, int alen, int blen,
                      int pmin, int pmax, or_matches *fwdA, or_matches *fwdB,
                      or_matches *revA, or_matches *revB, or_products *out) {
    int minL = pmin, maxL = pmax; /* :113-120 */
    if (minL == 0) minL = cfg->min_len;
    if (maxL == 0) maxL = cfg->max_len;
    join_direction(cfg, seqlen, pair, minL, maxL, fwdA, revB, blen, 0, out);
    join_direction(cfg, seqlen, pair, minL, maxL, fwdB, revA, alen, 1, out);
}

/* ------------------------------------------- core/engine/bruteforce.go:11-38 */

void or_simulate_bruteforce(const or_config *cfg, const uint8_t *seq, int n,
                            int npairs, const char *const *fwd, const char *const *rev,
                            const int32_t *minp, const int32_t *maxp, or_products *out) {
    or_init();
    for (int i = 0; i < npairs; i++) {
        int alen = (int)strlen(fwd[i]), blen = (int)strlen(rev[i]);
        uint8_t *rcA = xmalloc((size_t)alen + 1), *rcB = xmalloc((size_t)blen + 1);
        if (or_revcomp((const uint8_t *)fwd[i], alen, rcA) || or_revcomp((const uint8_t *)rev[i], blen, rcB)) {
            fprintf(stderr, "oracle: invalid reverse-complement base (rc.go:27-34 panics)\n");
            abort();
        }
        or_matches fA = {0}, fB = {0}, rA = {0}, rB = {0};
        or_find_matches(seq, n, (const uint8_t *)fwd[i], alen, cfg->max_mm, cfg->hit_cap, cfg->terminal_window, &fA);
        or_find_matches(seq, n, (const uint8_t *)rev[i], blen, cfg->max_mm, cfg->hit_cap, cfg->terminal_window, &fB);
        or_find_matches(seq, n, rcA, alen, cfg->max_mm, cfg->hit_cap, 0, &rA);
        filter_left_tw(&rA, cfg->terminal_window);
        or_find_matches(seq, n, rcB, blen, cfg->max_mm, cfg->hit_cap, 0, &rB);
        filter_left_tw(&rB, cfg->terminal_window);
        join_pair(cfg, n, i, alen, blen, minp ? minp[i] : 0, maxp ? maxp[i] : 0, &fA, &fB, &rA, &rB, out);
        or_matches_free(&fA); or_matches_free(&fB); or_matches_free(&rA); or_matches_free(&rB);
        free(rcA); free(rcB);
    }
}

/* -------------------------------------------------- core/engine/seed.go */

#define OR_MAX_VARIANTS 50000 /* seed.go:97 */

typedef struct {
    int32_t pair;
    char which;
    int32_t primer_len;
    int32_t seed_offset;
    int32_t next; /* next payload of the same pattern, -1 = end */
} seed_payload;

typedef struct {
    uint64_t code; /* 2-bit key, seed.go:29-55 */
    uint8_t len;
    int32_t head, tail, npay;
} seed_pattern;

typedef struct {
    seed_pattern *pat; int32_t npat, cappat;
    seed_payload *pay; int32_t npay, cappay;
    int32_t *table; uint32_t tsize; /* open addressing over (code,len) -> pattern idx */
} seed_builder;

static uint64_t seed_hash(uint64_t code, uint8_t len) {
    uint64_t x = code ^ ((uint64_t)len << 58);
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}

static void builder_grow(seed_builder *b) {
    uint32_t ns = b->tsize ? b->tsize * 2 : 1024;
    int32_t *nt = xmalloc((size_t)ns * sizeof(int32_t));
    for (uint32_t i = 0; i < ns; i++) nt[i] = -1;
    for (int32_t i = 0; i < b->npat; i++) {
        uint32_t h = (uint32_t)(seed_hash(b->pat[i].code, b->pat[i].len) & (ns - 1));
        while (nt[h] >= 0) h = (h + 1) & (ns - 1);
        nt[h] = i;
    }
    free(b->table);
    b->table = nt; b->tsize = ns;
}

/* seed.go:33-55 */
static int encode_seed(const uint8_t *p, int n, uint64_t *code) {
    if (n == 0 || n > 32) return 0;
    uint64_t c = 0;
    for (int i = 0; i < n; i++) {
        uint64_t v;
        switch (p[i]) {
        case 'A': v = 0; break;
        case 'C': v = 1; break;
        case 'G': v = 2; break;
        case 'T': v = 3; break;
        default: return 0;
        }
        c = (c << 2) | v;
    }
    *code = c;
    return 1;
}

/* seed.go:245-258 */
static int builder_add(seed_builder *b, const uint8_t *pat, int n, const seed_payload *pl) {
    uint64_t code;
    if (!encode_seed(pat, n, &code)) return 0;
    if ((uint64_t)(b->npat + 1) * 2 > b->tsize) builder_grow(b);
    uint32_t h = (uint32_t)(seed_hash(code, (uint8_t)n) & (b->tsize - 1));
    int32_t idx = -1;
    while (b->table[h] >= 0) {
        seed_pattern *sp = &b->pat[b->table[h]];
        if (sp->code == code && sp->len == (uint8_t)n) { idx = b->table[h]; break; }
        h = (h + 1) & (b->tsize - 1);
    }
    if (idx < 0) {
        if (b->npat == b->cappat) {
            b->cappat = b->cappat ? b->cappat * 2 : 256;
            b->pat = xrealloc(b->pat, (size_t)b->cappat * sizeof(seed_pattern));
        }
        idx = b->npat++;
        b->pat[idx].code = code; b->pat[idx].len = (uint8_t)n;
        b->pat[idx].head = b->pat[idx].tail = -1; b->pat[idx].npay = 0;
        b->table[h] = idx;
    }
    if (b->npay == b->cappay) {
        b->cappay = b->cappay ? b->cappay * 2 : 256;
        b->pay = xrealloc(b->pay, (size_t)b->cappay * sizeof(seed_payload));
    }
    int32_t pi = b->npay++;
    b->pay[pi] = *pl;
    b->pay[pi].next = -1;
    if (b->pat[idx].tail >= 0) b->pay[b->pat[idx].tail].next = pi; else b->pat[idx].head = pi;
    b->pat[idx].tail = pi;
    b->pat[idx].npay++;
    return 1;
}

static void builder_free(seed_builder *b) {
    free(b->pat); free(b->pay); free(b->table);
    memset(b, 0, sizeof *b);
}

/* seed.go:104-115 */
static int configured_seed_len(int primer_len, int cfg) {
    if (cfg <= 0) return primer_len < 12 ? primer_len : 12;
    return cfg > primer_len ? primer_len : cfg;
}

/* seed.go:285-303 */
static uint64_t exact_variant_estimate(const uint8_t *seed, int n) {
    uint64_t score = 1;
    for (int i = 0; i < n; i++) {
        uint64_t c = 0;
        for (int b = 0; b < 4; b++)
            if (or_base_match((uint8_t)"ACGT"[b], seed[i])) c++;
        if (c == 0) c = 5;
        score *= c; /* uint64 wrap-around like Go */
    }
    return score;
}

/* seed.go:260-283 */
static int choose_seed_span(const uint8_t *pat, int plen, int seed_len, int prefer_right,
                            int *offset, int *length) {
    int len = configured_seed_len(plen, seed_len);
    if (len <= 0 || len > plen) return 0;
    int best_off = 0;
    uint64_t best_score = 0;
    for (int off = 0; off <= plen - len; off++) {
        uint64_t score = exact_variant_estimate(pat + off, len);
        int tie = prefer_right ? (off > best_off) : (off < best_off);
        if (off == 0 || score < best_score || (score == best_score && tie)) {
            best_off = off; best_score = score;
        }
    }
    *offset = best_off; *length = len;
    return 1;
}

typedef struct {
    const uint8_t *seed; int slen;
    int seed_offset, primer_len, max_mm, left_tw, right_tw, max_variants;
    uint8_t *buf;
    uint8_t *store; int emitted; int ok; /* variants stored back to back, slen bytes each */
    size_t store_cap;
} enum_ctx;

/* seed.go:331-363 */
static void enum_walk(enum_ctx *c, int pos, int mm) {
    if (!c->ok || mm > c->max_mm) return;
    if (pos == c->slen) {
        if (c->emitted >= c->max_variants) { c->ok = 0; return; }
        size_t need = (size_t)(c->emitted + 1) * (size_t)c->slen;
        if (need > c->store_cap) {
            c->store_cap = c->store_cap ? c->store_cap * 2 : 4096;
            while (c->store_cap < need) c->store_cap *= 2;
            c->store = xrealloc(c->store, c->store_cap);
        }
        memcpy(c->store + (size_t)c->emitted * (size_t)c->slen, c->buf, (size_t)c->slen);
        c->emitted++;
        return;
    }
    int full = c->seed_offset + pos;
    int prot = (c->left_tw > 0 && full < c->left_tw) ||
               (c->right_tw > 0 && full >= c->primer_len - c->right_tw);
    for (int b = 0; b < 4; b++) {
        uint8_t ch = (uint8_t)"ACGT"[b];
        int cost = or_base_match(ch, c->seed[pos]) ? 0 : 1;
        if (prot && cost != 0) continue;
        if (mm + cost > c->max_mm) continue;
        c->buf[pos] = ch;
        enum_walk(c, pos + 1, mm + cost);
    }
}

/* seed.go:304-367 ; returns ok, variants in *store (caller frees), count in *count */
static int enumerate_seed_variants(const uint8_t *seed, int slen, int seed_offset, int primer_len,
                                   int max_mm, int left_tw, int right_tw, int max_variants,
                                   uint8_t **store, int *count) {
    *store = NULL; *count = 0;
    if (slen == 0 || max_variants <= 0) return 0;
    enum_ctx c;
    memset(&c, 0, sizeof c);
    c.seed = seed; c.slen = slen; c.seed_offset = seed_offset; c.primer_len = primer_len;
    c.max_mm = max_mm < 0 ? 0 : max_mm;
    c.left_tw = left_tw < 0 ? 0 : left_tw;
    c.right_tw = right_tw < 0 ? 0 : right_tw;
    c.max_variants = max_variants;
    c.buf = xmalloc((size_t)slen);
    c.ok = 1;
    enum_walk(&c, 0, 0);
    free(c.buf);
    *store = c.store; *count = c.emitted;
    return c.ok && c.emitted > 0;
}

/* seed.go:174-215 addOrientation */
static int add_orientation(seed_builder *b, int pair, char which, const uint8_t *pat, int plen,
                           int seed_len, int max_mm, int prefer_right, int left_tw, int right_tw) {
    if (plen == 0) return 0;
    int off, sl;
    if (!choose_seed_span(pat, plen, seed_len, prefer_right, &off, &sl)) return 0;
    uint8_t *store; int count;
    int ok = enumerate_seed_variants(pat + off, sl, off, plen, max_mm, left_tw, right_tw,
                                     OR_MAX_VARIANTS, &store, &count);
    if (!ok || count == 0) { free(store); return 0; }
    seed_payload pl = { pair, which, plen, off, -1 };
    int added = 0;
    for (int i = 0; i < count; i++) {
        if (!builder_add(b, store + (size_t)i * (size_t)sl, sl, &pl)) { free(store); return 0; }
        added = 1;
    }
    free(store);
    return added;
}

/* ---------------------------------------------------- core/engine/ac.go */

typedef struct {
    uint32_t next[4];
    uint32_t fail;
    uint32_t out_start;
    uint32_t out_len;
} ac_node;

typedef struct {
    ac_node *nodes; uint32_t nnodes;
    uint32_t *out; uint32_t nout;
} automaton;

/* ac.go:64-139 ; pattern i = pats[i] (plen[i] bytes of ACGT) */
static void build_ac(automaton *a, int npat, const uint8_t *const *pats, const int *plen) {
    memset(a, 0, sizeof *a);
    uint32_t cap = 1024, n = 1;
    uint32_t (*next)[4] = xcalloc(cap, sizeof *next);
    /* own pattern lists as singly linked lists in append order */
    int32_t *own_head = xmalloc(cap * sizeof(int32_t)), *own_tail = xmalloc(cap * sizeof(int32_t));
    own_head[0] = own_tail[0] = -1;
    int32_t *own_next = xmalloc((size_t)(npat ? npat : 1) * sizeof(int32_t));
    for (int i = 0; i < npat; i++) {
        uint32_t cur = 0;
        for (int j = 0; j < plen[i]; j++) {
            int code = g_basecode[pats[i][j]];
            if (code < 0) { fprintf(stderr, "oracle: buildAC received non-ACGT seed pattern (ac.go:71-76 panics)\n"); abort(); }
            if (next[cur][code] == 0) {
                if (n == cap) {
                    cap *= 2;
                    next = xrealloc(next, cap * sizeof *next);
                    own_head = xrealloc(own_head, cap * sizeof(int32_t));
                    own_tail = xrealloc(own_tail, cap * sizeof(int32_t));
                }
                memset(next[n], 0, sizeof next[n]);
                own_head[n] = own_tail[n] = -1;
                next[cur][code] = n++;
            }
            cur = next[cur][code];
        }
        own_next[i] = -1;
        if (own_tail[cur] >= 0) own_next[own_tail[cur]] = i; else own_head[cur] = i;
        own_tail[cur] = i;
    }
    /* BFS fail links, outputs = own ++ out(fail)  (ac.go:85-118) */
    uint32_t *fail = xcalloc(n, sizeof(uint32_t));
    uint32_t *ostart = xcalloc(n, sizeof(uint32_t)), *olen = xcalloc(n, sizeof(uint32_t));
    uint32_t ocap = (uint32_t)(npat ? npat * 2 : 4), on = 0;
    uint32_t *obuf = xmalloc(ocap * sizeof(uint32_t));
    uint32_t *queue = xmalloc(n * sizeof(uint32_t));
    uint32_t qh = 0, qt = 0;
    /* per-node temporary out list built in BFS order into obuf (not the final order yet) */
#define EMIT_OUT(node, failnode)                                                          \
    do {                                                                                  \
        uint32_t need = on;                                                               \
        for (int32_t t = own_head[node]; t >= 0; t = own_next[t]) need++;                 \
        need += olen[failnode];                                                           \
        if (need > ocap) { while (ocap < need) ocap *= 2; obuf = xrealloc(obuf, ocap * sizeof(uint32_t)); } \
        ostart[node] = on;                                                                \
        for (int32_t t = own_head[node]; t >= 0; t = own_next[t]) obuf[on++] = (uint32_t)t; \
        for (uint32_t t = 0; t < olen[failnode]; t++) obuf[on++] = obuf[ostart[failnode] + t]; \
        olen[node] = on - ostart[node];                                                   \
    } while (0)
    /* root: own outputs only (empty patterns are rejected upstream) */
    ostart[0] = 0; olen[0] = 0;
    for (int c = 0; c < 4; c++) {
        uint32_t child = next[0][c];
        if (child != 0) {
            fail[child] = 0;
            /* ac.go:88-93: depth-1 nodes keep their own outputs; fail = root has none */
            EMIT_OUT(child, 0);
            queue[qt++] = child;
        }
    }
    while (qh < qt) {
        uint32_t r = queue[qh++];
        for (int c = 0; c < 4; c++) {
            uint32_t s = next[r][c];
            if (s == 0) continue;
            queue[qt++] = s;
            uint32_t f = fail[r];
            while (f > 0 && next[f][c] == 0) f = fail[f];
            if (next[f][c] != 0) f = next[f][c];
            fail[s] = f;
            EMIT_OUT(s, f);
        }
    }
#undef EMIT_OUT
    /* flatten in node-index order (ac.go:120-137) */
    a->nodes = xmalloc((size_t)n * sizeof(ac_node));
    a->nnodes = n;
    a->out = xmalloc((size_t)(on ? on : 1) * sizeof(uint32_t));
    uint32_t w = 0;
    for (uint32_t i = 0; i < n; i++) {
        memcpy(a->nodes[i].next, next[i], sizeof next[i]);
        a->nodes[i].fail = fail[i];
        a->nodes[i].out_start = w;
        a->nodes[i].out_len = olen[i];
        memcpy(a->out + w, obuf + ostart[i], (size_t)olen[i] * sizeof(uint32_t));
        w += olen[i];
    }
    a->nout = w;
    free(next); free(own_head); free(own_tail); free(own_next);
    free(fail); free(ostart); free(olen); free(obuf); free(queue);
}

static void automaton_free(automaton *a) {
    free(a->nodes); free(a->out);
    memset(a, 0, sizeof *a);
}

/* ac.go:141-148 */
static int sequence_has_reset(const uint8_t *seq, int n) {
    for (int i = 0; i < n; i++)
        if (g_basecode[seq[i]] < 0) return 1;
    return 0;
}

typedef void (*ac_cb)(void *ud, int end_pos, int pattern_idx);

/* ac.go:151-180 */
static void scan_ac_each(const uint8_t *seq, int n, const automaton *a, ac_cb fn, void *ud) {
    if (a->nnodes == 0 || !fn) return;
    uint32_t state = 0;
    for (int i = 0; i < n; i++) {
        int code = g_basecode[seq[i]];
        if (code < 0) { state = 0; continue; }
        while (state > 0 && a->nodes[state].next[code] == 0) state = a->nodes[state].fail;
        uint32_t nx = a->nodes[state].next[code];
        if (nx != 0) state = nx;
        const ac_node *node = &a->nodes[state];
        if (node->out_len == 0) continue;
        for (uint32_t t = 0; t < node->out_len; t++) fn(ud, i, (int)a->out[node->out_start + t]);
    }
}

/* ac.go:186-213 ; idx must hold n entries */
static int verify_at(const uint8_t *seq, int seqlen, int pos, const uint8_t *pat, int n,
                     int max_mm, int left_tw, int right_tw, int32_t *idx, int *mm_out) {
    if (pos < 0 || pos + n > seqlen) return 0;
    int mm = 0;
    int left_cut = left_tw, right_cut = n - right_tw;
    for (int j = 0; j < n; j++) {
        if (!or_base_match(seq[pos + j], pat[j])) {
            if (j < left_cut || j >= right_cut) return 0;
            idx[mm] = j;
            mm++;
            if (max_mm >= 0 && mm > max_mm) return 0;
        }
    }
    *mm_out = mm;
    return 1;
}

/* -------------------------------------------------- core/engine/halo.go */

typedef struct { int start, end; } seq_range;

/* halo.go:9-24 */
static seq_range *non_acgt_ranges(const uint8_t *seq, int n, int *count) {
    seq_range *r = NULL; int nr = 0, cap = 0;
    for (int i = 0; i < n;) {
        if (g_basecode[seq[i]] >= 0) { i++; continue; }
        int start = i;
        while (i < n && g_basecode[seq[i]] < 0) i++;
        if (nr == cap) { cap = cap ? cap * 2 : 16; r = xrealloc(r, (size_t)cap * sizeof(seq_range)); }
        r[nr].start = start; r[nr].end = i; nr++;
    }
    *count = nr;
    return r;
}

typedef void (*start_cb)(void *ud, int start);

/* halo.go:26-74 */
static void for_each_halo_start(int seq_len, int primer_len, const seq_range *ranges, int nr,
                                start_cb fn, void *ud) {
    if (seq_len <= 0 || primer_len <= 0 || primer_len > seq_len || nr == 0 || !fn) return;
    int limit = seq_len - primer_len;
    int cur_lo = -1, cur_hi = -1;
    for (int i = 0; i < nr; i++) {
        if (ranges[i].end <= ranges[i].start) continue;
        int lo = ranges[i].start - primer_len + 1;
        if (lo < 0) lo = 0;
        int hi = ranges[i].end - 1;
        if (hi > limit) hi = limit;
        if (hi < lo) continue;
        if (cur_lo < 0) { cur_lo = lo; cur_hi = hi; continue; }
        if (lo <= cur_hi + 1) { if (hi > cur_hi) cur_hi = hi; continue; }
        for (int s = cur_lo; s <= cur_hi; s++) fn(ud, s);
        cur_lo = lo; cur_hi = hi;
    }
    if (cur_lo >= 0)
        for (int s = cur_lo; s <= cur_hi; s++) fn(ud, s);
}

/* ------------------------------------------- core/engine/hit_collect.go */

typedef struct {
    or_matches matches;
    int32_t starts[8]; int nstarts;          /* hit_collect.go:9 linear limit */
    int32_t *set; uint32_t set_size, set_n;  /* promoted visited set (open addressing, -1 empty) */
} collector;

static int set_has(const collector *c, int start) {
    uint32_t h = ((uint32_t)start * 2654435761u) & (c->set_size - 1);
    while (c->set[h] != -1) { if (c->set[h] == start) return 1; h = (h + 1) & (c->set_size - 1); }
    return 0;
}
static void set_add(collector *c, int start) {
    if ((c->set_n + 1) * 2 > c->set_size) {
        uint32_t ns = c->set_size ? c->set_size * 2 : 32;
        int32_t *nt = xmalloc((size_t)ns * sizeof(int32_t));
        for (uint32_t i = 0; i < ns; i++) nt[i] = -1;
        for (uint32_t i = 0; i < c->set_size; i++)
            if (c->set[i] != -1) {
                uint32_t h = ((uint32_t)c->set[i] * 2654435761u) & (ns - 1);
                while (nt[h] != -1) h = (h + 1) & (ns - 1);
                nt[h] = c->set[i];
            }
        free(c->set);
        c->set = nt; c->set_size = ns;
    }
    uint32_t h = ((uint32_t)start * 2654435761u) & (c->set_size - 1);
    while (c->set[h] != -1) h = (h + 1) & (c->set_size - 1);
    c->set[h] = start;
    c->set_n++;
}

/* hit_collect.go:100-129 */
static int collector_seen(const collector *c, int start) {
    if (c->set) return set_has(c, start);
    for (int i = 0; i < c->nstarts; i++)
        if (c->starts[i] == start) return 1;
    return 0;
}
static void collector_mark(collector *c, int start) {
    if (c->set) { set_add(c, start); return; }
    if (c->nstarts < 8) { c->starts[c->nstarts++] = start; return; }
    for (int i = 0; i < c->nstarts; i++) set_add(c, c->starts[i]);
    c->nstarts = 0;
    set_add(c, start);
}

/* hit_collect.go:79-98 */
static void collector_add_verified(collector *c, const uint8_t *seq, int seqlen, int start,
                                   const uint8_t *pat, int plen, int max_mm, int left_tw,
                                   int right_tw, int hit_cap, int32_t *idx_scratch) {
    if (hit_cap > 0 && c->matches.n >= hit_cap) return;
    if (collector_seen(c, start)) return;
    collector_mark(c, start);
    int mm;
    if (!verify_at(seq, seqlen, start, pat, plen, max_mm, left_tw, right_tw, idx_scratch, &mm)) return;
    matches_push(&c->matches, start, mm, plen, idx_scratch, mm);
}

static void collector_free(collector *c) {
    or_matches_free(&c->matches);
    free(c->set);
    memset(c, 0, sizeof *c);
}

/* -------------------------------------------- core/engine/compiled.go */

struct or_panel {
    or_config cfg;
    int npairs;
    uint8_t **seq[4];   /* [0]=A, [1]=B, [2]=rc(A), [3]=rc(B) ; compiled.go:108-117 */
    int *len[4];
    int32_t *minp, *maxp;
    seed_builder sb;    /* SeedPatterns */
    uint8_t **pat_bytes; int *pat_len; /* decoded patterns for the automaton */
    automaton ac;
    uint8_t *have;      /* orientationMask per pair: bit0 A, bit1 B, bit2 a, bit3 b */
    int max_plen;
};

static int which_index(char w) {
    switch (w) { case 'A': return 0; case 'B': return 1; case 'a': return 2; case 'b': return 3; }
    return -1;
}

static void decode_seed(uint64_t code, int len, uint8_t *out) { /* seed.go:57-78 */
    for (int i = len - 1; i >= 0; i--) { out[i] = (uint8_t)"ACGT"[code & 3]; code >>= 2; }
}

/* seed.go:152-231 into a builder; have[] bits set per seeded orientation */
static void build_seed_patterns(seed_builder *sb, uint8_t *have, int npairs,
                                uint8_t **seqs[4], int *lens[4], int seed_len, int tw, int max_mm) {
    if (seed_len < 0) return; /* :156-159 */
    for (int i = 0; i < npairs; i++) {
        if (add_orientation(sb, i, 'A', seqs[0][i], lens[0][i], seed_len, max_mm, 1, 0, tw)) have[i] |= 1;
        if (add_orientation(sb, i, 'B', seqs[1][i], lens[1][i], seed_len, max_mm, 1, 0, tw)) have[i] |= 2;
        if (add_orientation(sb, i, 'a', seqs[2][i], lens[2][i], seed_len, max_mm, 0, tw, 0)) have[i] |= 4;
        if (add_orientation(sb, i, 'b', seqs[3][i], lens[3][i], seed_len, max_mm, 0, tw, 0)) have[i] |= 8;
    }
}

or_panel *or_panel_create(const or_config *cfg, int npairs, const char *const *fwd,
                          const char *const *rev, const int32_t *minp, const int32_t *maxp) {
    or_init();
    or_panel *p = xcalloc(1, sizeof *p);
    p->cfg = *cfg;
    p->npairs = npairs;
    for (int o = 0; o < 4; o++) {
        p->seq[o] = xcalloc((size_t)npairs, sizeof(uint8_t *));
        p->len[o] = xcalloc((size_t)npairs, sizeof(int));
    }
    p->minp = xcalloc((size_t)npairs, sizeof(int32_t));
    p->maxp = xcalloc((size_t)npairs, sizeof(int32_t));
    p->have = xcalloc((size_t)npairs, 1);
    for (int i = 0; i < npairs; i++) {
        int alen = (int)strlen(fwd[i]), blen = (int)strlen(rev[i]);
        p->len[0][i] = p->len[2][i] = alen;
        p->len[1][i] = p->len[3][i] = blen;
        if (alen > p->max_plen) p->max_plen = alen;
        if (blen > p->max_plen) p->max_plen = blen;
        p->seq[0][i] = xmalloc((size_t)alen + 1); memcpy(p->seq[0][i], fwd[i], (size_t)alen + 1);
        p->seq[1][i] = xmalloc((size_t)blen + 1); memcpy(p->seq[1][i], rev[i], (size_t)blen + 1);
        p->seq[2][i] = xcalloc((size_t)alen + 1, 1);
        p->seq[3][i] = xcalloc((size_t)blen + 1, 1);
        if (or_revcomp(p->seq[0][i], alen, p->seq[2][i]) || or_revcomp(p->seq[1][i], blen, p->seq[3][i])) {
            fprintf(stderr, "oracle: invalid reverse-complement base (rc.go:27-34 panics)\n");
            abort();
        }
        p->minp[i] = minp ? minp[i] : 0;
        p->maxp[i] = maxp ? maxp[i] : 0;
    }
    if (npairs == 0) return p; /* compiled.go:101-103 */
    build_seed_patterns(&p->sb, p->have, npairs, p->seq, p->len, cfg->seed_len,
                        cfg->terminal_window, cfg->max_mm);
    int np = p->sb.npat;
    p->pat_bytes = xcalloc((size_t)(np ? np : 1), sizeof(uint8_t *));
    p->pat_len = xcalloc((size_t)(np ? np : 1), sizeof(int));
    for (int i = 0; i < np; i++) {
        p->pat_len[i] = p->sb.pat[i].len;
        p->pat_bytes[i] = xmalloc((size_t)p->pat_len[i]);
        decode_seed(p->sb.pat[i].code, p->pat_len[i], p->pat_bytes[i]);
    }
    if (np > 0) build_ac(&p->ac, np, (const uint8_t *const *)p->pat_bytes, p->pat_len);
    return p;
}

void or_panel_free(or_panel *p) {
    if (!p) return;
    for (int o = 0; o < 4; o++) {
        for (int i = 0; i < p->npairs; i++) free(p->seq[o][i]);
        free(p->seq[o]); free(p->len[o]);
    }
    for (int i = 0; i < p->sb.npat; i++) free(p->pat_bytes[i]);
    free(p->pat_bytes); free(p->pat_len);
    builder_free(&p->sb);
    automaton_free(&p->ac);
    free(p->minp); free(p->maxp); free(p->have);
    free(p);
}

int or_panel_have(const or_panel *p, int pair, char which) {
    int w = which_index(which);
    if (pair < 0 || pair >= p->npairs || w < 0) return 0;
    return (p->have[pair] >> w) & 1;
}
int or_panel_num_seed_patterns(const or_panel *p) { return p->sb.npat; }
int or_panel_num_nodes(const or_panel *p) { return (int)p->ac.nnodes; }
int or_panel_seed_pattern(const or_panel *p, int i, char *pat_out, int cap, int *npayloads) {
    if (i < 0 || i >= p->sb.npat) return -1;
    int n = p->pat_len[i];
    if (cap < n + 1) return -1;
    memcpy(pat_out, p->pat_bytes[i], (size_t)n);
    pat_out[n] = 0;
    if (npayloads) *npayloads = p->sb.pat[i].npay;
    return n;
}

typedef struct {
    const or_panel *p;
    const uint8_t *seq; int n;
    collector *col; /* [npairs*4] */
    int32_t *idx_scratch;
} scan_ctx;

/* compiled.go:192-207 addHit */
static void add_hit(scan_ctx *c, int pair, char which, int start) {
    const or_panel *p = c->p;
    if (pair < 0 || pair >= p->npairs) return;
    int w = which_index(which);
    if (w < 0) return;
    int tw = p->cfg.terminal_window;
    int left = (w >= 2) ? tw : 0, right = (w >= 2) ? 0 : tw;
    collector_add_verified(&c->col[pair * 4 + w], c->seq, c->n, start, p->seq[w][pair],
                           p->len[w][pair], p->cfg.max_mm, left, right, p->cfg.hit_cap,
                           c->idx_scratch);
}

static void ac_hit_cb(void *ud, int end_pos, int pattern_idx) { /* compiled.go:212-222 */
    scan_ctx *c = ud;
    const seed_builder *sb = &c->p->sb;
    const seed_pattern *sp = &sb->pat[pattern_idx];
    for (int32_t t = sp->head; t >= 0; t = sb->pay[t].next) {
        const seed_payload *pl = &sb->pay[t];
        int start = end_pos - pl->seed_offset - ((int)sp->len - 1);
        add_hit(c, pl->pair, pl->which, start);
    }
}

typedef struct { scan_ctx *c; int pair; char which; } halo_ud;
static void halo_cb(void *ud, int start) {
    halo_ud *h = ud;
    add_hit(h->c, h->pair, h->which, start);
}

/* compiled.go:162-267 up to (excluding) the join: fills per[pair*4+w] (owned by caller) */
static void scan_collect(const or_panel *p, const uint8_t *seq, int n, or_matches *per) {
    const or_config *cfg = &p->cfg;
    int np = p->npairs;
    scan_ctx c;
    c.p = p; c.seq = seq; c.n = n;
    c.col = xcalloc((size_t)np * 4, sizeof(collector));
    c.idx_scratch = xmalloc((size_t)(p->max_plen + 1) * sizeof(int32_t));

    int has_reset = cfg->max_mm > 0 && sequence_has_reset(seq, n); /* :189 */
    int force_fallback = has_reset && cfg->hit_cap > 0;            /* :190 */

    if (p->sb.npat > 0 && p->ac.nnodes != 0 && !force_fallback)    /* :211 */
        scan_ac_each(seq, n, &p->ac, ac_hit_cb, &c);

    if (has_reset && !force_fallback && cfg->max_mm > 0) {         /* :230-232, halo.go:76-108 */
        int nr;
        seq_range *ranges = non_acgt_ranges(seq, n, &nr);
        if (nr > 0) {
            static const char W[4] = { 'A', 'B', 'a', 'b' };
            for (int i = 0; i < np; i++)
                for (int w = 0; w < 4; w++)
                    if ((p->have[i] >> w) & 1) {
                        halo_ud h = { &c, i, W[w] };
                        for_each_halo_start(n, p->len[w][i], ranges, nr, halo_cb, &h);
                    }
        }
        free(ranges);
    }

    for (int i = 0; i < np; i++) {                                  /* :238-258 */
        for (int w = 0; w < 4; w++) {
            or_matches *dst = &per[i * 4 + w];
            if (force_fallback || !((p->have[i] >> w) & 1)) {
                if (w < 2) {
                    or_find_matches(seq, n, p->seq[w][i], p->len[w][i], cfg->max_mm, cfg->hit_cap,
                                    cfg->terminal_window, dst);
                } else {
                    or_find_matches(seq, n, p->seq[w][i], p->len[w][i], cfg->max_mm, cfg->hit_cap, 0, dst);
                    filter_left_tw(dst, cfg->terminal_window);
                }
            } else {
                *dst = c.col[i * 4 + w].matches; /* move */
                memset(&c.col[i * 4 + w].matches, 0, sizeof(or_matches));
            }
        }
    }
    for (int i = 0; i < np * 4; i++) collector_free(&c.col[i]);
    free(c.col);
    free(c.idx_scratch);
}

void or_panel_scan(const or_panel *p, const uint8_t *seq, int n, or_products *out) {
    if (!p || p->npairs == 0) return; /* :163-165 */
    int np = p->npairs;
    or_matches *per = xcalloc((size_t)np * 4, sizeof(or_matches));
    scan_collect(p, seq, n, per);
    for (int i = 0; i < np; i++) /* :260-265 */
        join_pair(&p->cfg, n, i, p->len[0][i], p->len[1][i], p->minp[i], p->maxp[i],
                  &per[i * 4 + 0], &per[i * 4 + 1], &per[i * 4 + 2], &per[i * 4 + 3], out);
    for (int i = 0; i < np * 4; i++) or_matches_free(&per[i]);
    free(per);
}

void or_panel_scan_matches(const or_panel *p, const uint8_t *seq, int n, int pair, char which,
                           or_matches *out) {
    int w = which_index(which);
    if (!p || pair < 0 || pair >= p->npairs || w < 0) return;
    int np = p->npairs;
    or_matches *per = xcalloc((size_t)np * 4, sizeof(or_matches));
    scan_collect(p, seq, n, per);
    *out = per[pair * 4 + w];
    memset(&per[pair * 4 + w], 0, sizeof(or_matches));
    for (int i = 0; i < np * 4; i++) or_matches_free(&per[i]);
    free(per);
}

/* ------------------------------------------------------- test entry points */

typedef struct { int32_t *out; int cap; int n; } pair_sink;
static void pair_sink_cb(void *ud, int a, int b) {
    pair_sink *s = ud;
    if (s->n < s->cap) { s->out[2 * s->n] = a; s->out[2 * s->n + 1] = b; }
    s->n++;
}

int or_ac_scan(int npat, const char *const *pats, const uint8_t *seq, int n,
               int32_t *out_pairs, int cap_pairs) {
    or_init();
    int *plen = xmalloc((size_t)(npat ? npat : 1) * sizeof(int));
    for (int i = 0; i < npat; i++) plen[i] = (int)strlen(pats[i]);
    automaton a;
    build_ac(&a, npat, (const uint8_t *const *)pats, plen);
    pair_sink s = { out_pairs, cap_pairs, 0 };
    scan_ac_each(seq, n, &a, pair_sink_cb, &s);
    automaton_free(&a);
    free(plen);
    return s.n;
}

int or_non_acgt_ranges(const uint8_t *seq, int n, int32_t *out_pairs, int cap_pairs) {
    or_init();
    int nr;
    seq_range *r = non_acgt_ranges(seq, n, &nr);
    for (int i = 0; i < nr && i < cap_pairs; i++) { out_pairs[2 * i] = r[i].start; out_pairs[2 * i + 1] = r[i].end; }
    free(r);
    return nr;
}

typedef struct { int32_t *out; int cap; int n; } int_sink;
static void int_sink_cb(void *ud, int v) {
    int_sink *s = ud;
    if (s->n < s->cap) s->out[s->n] = v;
    s->n++;
}

int or_halo_starts(int seq_len, int primer_len, int nranges, const int32_t *ranges,
                   int32_t *out, int cap) {
    seq_range *r = xmalloc((size_t)(nranges ? nranges : 1) * sizeof(seq_range));
    for (int i = 0; i < nranges; i++) { r[i].start = ranges[2 * i]; r[i].end = ranges[2 * i + 1]; }
    int_sink s = { out, cap, 0 };
    for_each_halo_start(seq_len, primer_len, r, nranges, int_sink_cb, &s);
    free(r);
    return s.n;
}

int or_build_seed_patterns_count(int npairs, const char *const *fwd, const char *const *rev,
                                 int seed_len, int tw, int max_mm) {
    or_config cfg = { max_mm, tw, 0, 0, 0, seed_len, 0 };
    or_panel *p = or_panel_create(&cfg, npairs, fwd, rev, NULL, NULL);
    int n = p->sb.npat;
    or_panel_free(p);
    return n;
}

/* ---------------------------------------------- core/oligo/oligo.go:19-77 */

static int is_strict_acgt(const uint8_t *s, int n) { /* oligo.go:79-91 */
    if (n == 0) return 0;
    return is_unambiguous(s, n);
}

or_hit or_best_hit(const uint8_t *amplicon, int n, const char *probe, int max_mm) {
    or_init();
    or_hit none = { 0, 0, 0, 0 };
    /* oligo.go:20-29 : amplicon upper-cased; probe normalised (whitespace/quotes stripped,
     * upper-cased) and validated against ACGTRYSWKMBDHVN (validate.go) */
    uint8_t *amp = xmalloc((size_t)n + 1);
    for (int i = 0; i < n; i++) amp[i] = (amplicon[i] >= 'a' && amplicon[i] <= 'z') ? (uint8_t)(amplicon[i] - 32) : amplicon[i];
    int plen = 0;
    uint8_t *prb = xmalloc(strlen(probe) + 1);
    for (const char *q = probe; *q; q++) {
        char ch = *q;
        if (ch == ' ' || ch == '\t' || ch == '\n' || ch == '\r' || ch == '\v' || ch == '\f' || ch == '\'' || ch == '"') continue;
        if (ch >= 'a' && ch <= 'z') ch = (char)(ch - 32);
        prb[plen++] = (uint8_t)ch;
    }
    if (plen == 0) { free(amp); free(prb); return none; }
    for (int i = 0; i < plen; i++)
        if (!strchr("ACGTRYSWKMBDHVN", prb[i])) { fprintf(stderr, "oracle: invalid probe (oligo.go:26-28 panics)\n"); abort(); }
    uint8_t *rc = xmalloc((size_t)plen + 1);
    or_revcomp(prb, plen, rc);

    or_hit best = none;
    if (max_mm == 0 && is_strict_acgt(prb, plen)) { /* :33-42 */
        for (int i = 0; i + plen <= n; i++)
            if (memcmp(amp + i, prb, (size_t)plen) == 0) { best = (or_hit){ 1, '+', i, 0 }; goto done; }
        for (int i = 0; i + plen <= n; i++)
            if (memcmp(amp + i, rc, (size_t)plen) == 0) { best = (or_hit){ 1, '-', i, 0 }; goto done; }
        goto done;
    }
    {
        or_matches plus = {0}, minus = {0};
        or_find_matches(amp, n, prb, plen, max_mm, 0, 0, &plus);
        or_find_matches(amp, n, rc, plen, max_mm, 0, 0, &minus);
        const or_matches *lists[2] = { &plus, &minus };
        const int strands[2] = { '+', '-' };
        for (int s = 0; s < 2; s++) { /* :58-75 */
            if (lists[s]->n == 0) continue;
            or_match bl = lists[s]->v[0];
            for (int i = 1; i < lists[s]->n; i++) {
                or_match h = lists[s]->v[i];
                if (h.mm < bl.mm || (h.mm == bl.mm && h.pos < bl.pos)) bl = h;
            }
            if (!best.found || bl.mm < best.mm || (bl.mm == best.mm && bl.pos < best.pos))
                best = (or_hit){ 1, strands[s], bl.pos, bl.mm };
        }
        or_matches_free(&plus); or_matches_free(&minus);
    }
done:
    free(amp); free(prb); free(rc);
    return best;
}

/* ------------- fixtures: core/engine/performance_benchmark_test.go:20-106 */

void or_bench_dna(uint8_t *out, int64_t n, uint32_t seed) { /* :67-76 */
    uint32_t x = seed;
    for (int64_t i = 0; i < n; i++) {
        x = x * 1664525u + 1013904223u;
        out[i] = (uint8_t)"ACGT"[(x >> 30) & 3];
    }
}

void or_bench_primer(int idx, int n, char *out) { /* :78-93 */
    uint32_t x = (uint32_t)(0x9e3779b9u ^ ((uint32_t)idx * 0x45d9f3bu));
    for (int i = 0; i < n; i++) {
        x = x * 1103515245u + 12345u + (uint32_t)(i * 97);
        out[i] = "ACGT"[(x >> 29) & 3];
    }
    out[0] = "ACGT"[idx & 3];
    out[1] = "ACGT"[(idx + 1) & 3];
    out[2] = "ACGT"[(idx + 2) & 3];
    out[n - 1] = "ACGT"[(idx + 3) & 3];
    out[n] = 0;
}

uint8_t or_different_base(uint8_t b) { /* :95-106 */
    switch (b) {
    case 'A': return 'C';
    case 'C': return 'G';
    case 'G': return 'T';
    default: return 'A';
    }
}

/* :25-65 ; fwd_out/rev_out hold pair_count records of 21 bytes (20-mer + NUL) */
int64_t or_make_bench_fixture(int pair_count, int64_t genome_len, int mutate_forward,
                              int reference_n, uint8_t *seq, char *fwd_out, char *rev_out) {
    const int product_len = 180, primer_len = 20;
    if (pair_count < 1) pair_count = 1;
    int64_t needed = 256 + (int64_t)pair_count * 256 + product_len;
    if (genome_len < needed) genome_len = needed;
    or_bench_dna(seq, genome_len, 0x5eed1234u);
    for (int i = 0; i < pair_count; i++) {
        char *fwd = fwd_out + (size_t)i * 21, *rev = rev_out + (size_t)i * 21;
        or_bench_primer(i * 2, primer_len, fwd);
        or_bench_primer(i * 2 + 1, primer_len, rev);
        int64_t start = 128 + (int64_t)i * 256;
        uint8_t planted[20];
        memcpy(planted, fwd, 20);
        if (mutate_forward) planted[10] = or_different_base(planted[10]);
        if (reference_n) planted[11] = 'N';
        memcpy(seq + start, planted, 20);
        uint8_t rc[20];
        or_revcomp((const uint8_t *)rev, 20, rc);
        memcpy(seq + start + product_len - 20, rc, 20);
    }
    return genome_len;
}

/* -------------------- CPU baseline: worker pool over rolling chunks (see header) */

typedef struct { int64_t start, end; } chunk_t;
typedef struct { int64_t gs, ge; int32_t type, pair; } pkey;

typedef struct {
    const or_panel *p; const uint8_t *seq;
    const chunk_t *chunks; int nchunks;
    int64_t njobs;   /* passes x nchunks: every pass of the record is queued at once, one pool serves them all */
    int64_t next; pthread_mutex_t mu;
    pkey *keys; int64_t nkeys, capkeys;
    int busy;        /* workers that got at least one chunk */
} mt_ctx;

static void *mt_worker(void *arg) {
    mt_ctx *c = arg;
    int mine = 0;
    for (;;) {
        pthread_mutex_lock(&c->mu);
        int64_t job = c->next++;
        if (job < c->njobs && mine++ == 0) c->busy++;
        pthread_mutex_unlock(&c->mu);
        if (job >= c->njobs) break;
        const int i = (int)(job % c->nchunks);
        or_products ps = {0};
        int64_t off = c->chunks[i].start;
        or_panel_scan(c->p, c->seq + off, (int)(c->chunks[i].end - off), &ps);
        if (ps.n > 0 && job < c->nchunks) { /* the products of the first pass are the answer; later passes only cost time */
            pthread_mutex_lock(&c->mu);
            if (c->nkeys + ps.n > c->capkeys) {
                while (c->nkeys + ps.n > c->capkeys) c->capkeys = c->capkeys ? c->capkeys * 2 : 1024;
                c->keys = xrealloc(c->keys, (size_t)c->capkeys * sizeof(pkey));
            }
            for (int k = 0; k < ps.n; k++) {
                pkey *key = &c->keys[c->nkeys++];
                key->gs = ps.v[k].start + off; key->ge = ps.v[k].end + off;
                key->type = ps.v[k].type; key->pair = ps.v[k].pair;
            }
            pthread_mutex_unlock(&c->mu);
        }
        or_products_free(&ps);
    }
    return NULL;
}

int64_t or_baseline_scan_pool(const or_panel *p, const uint8_t *seq, int64_t n, int chunk_size, int overlap,
                              int threads, int passes, int *busy, int *nchunks_out);

static int pkey_cmp(const void *a, const void *b) {
    const pkey *x = a, *y = b;
    if (x->gs != y->gs) return x->gs < y->gs ? -1 : 1;
    if (x->ge != y->ge) return x->ge < y->ge ? -1 : 1;
    if (x->type != y->type) return x->type < y->type ? -1 : 1;
    if (x->pair != y->pair) return x->pair < y->pair ? -1 : 1;
    return 0;
}

int64_t or_baseline_scan_mt(const or_panel *p, const uint8_t *seq, int64_t n,
                            int chunk_size, int overlap, int threads) {
    return or_baseline_scan_pool(p, seq, n, chunk_size, overlap, threads, 1, NULL, NULL);
}

/* `passes` passes over the record served by ONE pool of `threads` workers (all chunks of all passes queued together,
 * the way internal/pipeline/pipeline.go:60-125 feeds its workers from one job channel): the pool stays busy when one
 * pass alone has fewer chunks than there are threads.  *busy = workers that scanned at least one chunk. */
int64_t or_baseline_scan_pool(const or_panel *p, const uint8_t *seq, int64_t n, int chunk_size, int overlap,
                              int threads, int passes, int *busy, int *nchunks_out) {
    /* core/fasta/path_ctx.go:83-179 chunk schedule for one record of n bases */
    int64_t step = (int64_t)chunk_size - overlap;
    chunk_t *chunks = NULL; int nch = 0, cap = 0;
#define PUSH_CHUNK(s, e) do { if (nch == cap) { cap = cap ? cap * 2 : 64; chunks = xrealloc(chunks, (size_t)cap * sizeof(chunk_t)); } chunks[nch].start = (s); chunks[nch].end = (e); nch++; } while (0)
    if (chunk_size <= 0 || step <= 0 || n <= chunk_size) {
        PUSH_CHUNK(0, n);
    } else {
        int64_t ws = 0, last_end = 0;
        while (n - ws > chunk_size) { PUSH_CHUNK(ws, ws + chunk_size); last_end = ws + chunk_size; ws += step; }
        if (last_end < n) PUSH_CHUNK(ws, n);
    }
#undef PUSH_CHUNK
    if (threads < 1) threads = 1;
    mt_ctx c;
    memset(&c, 0, sizeof c);
    c.p = p; c.seq = seq; c.chunks = chunks; c.nchunks = nch;
    c.njobs = (int64_t)nch * (passes < 1 ? 1 : passes);
    pthread_mutex_init(&c.mu, NULL);
    pthread_t *th = xmalloc((size_t)threads * sizeof(pthread_t));
    for (int i = 0; i < threads; i++) pthread_create(&th[i], NULL, mt_worker, &c);
    for (int i = 0; i < threads; i++) pthread_join(th[i], NULL);
    free(th);
    pthread_mutex_destroy(&c.mu);
    int64_t uniq = 0;
    if (c.nkeys > 0) {
        qsort(c.keys, (size_t)c.nkeys, sizeof(pkey), pkey_cmp);
        uniq = 1;
        for (int64_t i = 1; i < c.nkeys; i++)
            if (pkey_cmp(&c.keys[i], &c.keys[i - 1]) != 0) uniq++;
    }
    free(c.keys);
    free(chunks);
    if (busy) *busy = c.busy;
    if (nchunks_out) *nchunks_out = nch;
    return uniq;
}
