/*
 * trew_oracle.c -- CPU ORACLE for the TREW per-read tandem-repeat scan.
 *
 * TEST INFRASTRUCTURE ONLY: a plain-C restatement of the reference algorithm
 * (Chemical118/TREW, src/kmer.cpp) used as the checker for the HIP path and,
 * timed, as bench.py's "port" CPU baseline.  Nothing under trew_amd/ or
 * include/ may link, load or call it.  Every function cites the reference
 * lines it follows.  Pinning: see trew_oracle.h and DESIGN.md.
 */
#include "trew_oracle.h"

#include <alloca.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

/* codes[256], kmer.cpp:14-31: T=0 G=1 C=2 A=3 (upper and lower case), anything
 * else -1.  The reference indexes with a signed char (UB for bytes >= 0x80,
 * SURVEY G8); those bytes are treated as invalid here. */
int trew_oracle_code(unsigned char c) {
    switch (c) {
    case 'T': case 't': return 0;
    case 'G': case 'g': return 1;
    case 'C': case 'c': return 2;
    case 'A': case 'a': return 3;
    default: return -1;
    }
}

/* ---------- small (k, word) -> count list: the role of a ResultMap used as a
 * temporary (temp_result_left/right, kmer.cpp:91-92) ---------- */
typedef struct {
    int k;
    u128 w;
    uint64_t c;
} rent;
typedef struct {
    rent *e;
    int n, cap;
} rlist;

static void rlist_init(rlist *l) {
    l->e = NULL;
    l->n = l->cap = 0;
}
static void rlist_clear(rlist *l) { l->n = 0; }
static void rlist_free(rlist *l) {
    free(l->e);
    l->e = NULL;
    l->n = l->cap = 0;
}
static void rlist_add(rlist *l, int k, u128 w, uint64_t c) {
    for (int i = 0; i < l->n; i++)
        if (l->e[i].k == k && l->e[i].w == w) {
            l->e[i].c += c;
            return;
        }
    if (l->n == l->cap) {
        l->cap = l->cap ? 2 * l->cap : 64;
        l->e = (rent *) realloc(l->e, sizeof(rent) * (size_t) l->cap);
    }
    l->e[l->n].k = k;
    l->e[l->n].w = w;
    l->e[l->n].c = c;
    l->n++;
}

#define WORD uint64_t
#define WBITS 64
#define SFX 64
#include "trew_oracle_core.inc"
#undef WORD
#undef WBITS
#undef SFX

#define WORD u128
#define WBITS 128
#define SFX 128
#include "trew_oracle_core.inc"
#undef WORD
#undef WBITS
#undef SFX

/* ---------- the six result tables (ResultMapData, kmer.h:79-81) ---------- */
typedef struct {
    int32_t *k; /* 0 = empty slot */
    u128 *w;
    uint64_t *c;
    uint64_t cap, n;
} gtable;

static void gtable_init(gtable *t) {
    t->cap = 1024;
    t->n = 0;
    t->k = (int32_t *) calloc(t->cap, sizeof(int32_t));
    t->w = (u128 *) malloc(sizeof(u128) * t->cap);
    t->c = (uint64_t *) malloc(sizeof(uint64_t) * t->cap);
}
static void gtable_free(gtable *t) {
    free(t->k);
    free(t->w);
    free(t->c);
}
static uint64_t gtable_hash(int k, u128 w) {
    uint64_t h = (uint64_t) w ^ ((uint64_t) (w >> 64) * 0xD6E8FEB86659FD93ull) ^ ((uint64_t) k << 57);
    h ^= h >> 29;
    h *= 0x9E3779B97F4A7C15ull;
    h ^= h >> 32;
    return h;
}
static void gtable_add(gtable *t, int k, u128 w, uint64_t c);
static void gtable_grow(gtable *t) {
    gtable o = *t;
    t->cap = o.cap * 2;
    t->n = 0;
    t->k = (int32_t *) calloc(t->cap, sizeof(int32_t));
    t->w = (u128 *) malloc(sizeof(u128) * t->cap);
    t->c = (uint64_t *) malloc(sizeof(uint64_t) * t->cap);
    for (uint64_t i = 0; i < o.cap; i++)
        if (o.k[i]) gtable_add(t, o.k[i], o.w[i], o.c[i]);
    gtable_free(&o);
}
static void gtable_add(gtable *t, int k, u128 w, uint64_t c) {
    if (2 * (t->n + 1) > t->cap) gtable_grow(t);
    uint64_t i = gtable_hash(k, w) & (t->cap - 1);
    for (;;) {
        if (t->k[i] == 0) {
            t->k[i] = k;
            t->w[i] = w;
            t->c[i] = c;
            t->n++;
            return;
        }
        if (t->k[i] == k && t->w[i] == w) {
            t->c[i] += c;
            return;
        }
        i = (i + 1) & (t->cap - 1);
    }
}

struct trew_oracle_ctx {
    trew_oracle_params p;
    gtable t[TREW_NUM_TABLES];
    segmap_64 m64;
    segmap_128 m128;
    rlist tl[2], tr[2]; /* temp_result_left / temp_result_right, [0]=high(first) [1]=low(second) */
};

trew_oracle_ctx *trew_oracle_new(const trew_oracle_params *p) {
    trew_oracle_ctx *c = (trew_oracle_ctx *) calloc(1, sizeof(*c));
    c->p = *p;
    for (int i = 0; i < TREW_NUM_TABLES; i++) gtable_init(&c->t[i]);
    segmap_init_64(&c->m64, 1u << 14);
    segmap_init_128(&c->m128, 1u << 14);
    for (int i = 0; i < 2; i++) {
        rlist_init(&c->tl[i]);
        rlist_init(&c->tr[i]);
    }
    return c;
}
void trew_oracle_free(trew_oracle_ctx *c) {
    if (!c) return;
    for (int i = 0; i < TREW_NUM_TABLES; i++) gtable_free(&c->t[i]);
    segmap_free_64(&c->m64);
    segmap_free_128(&c->m128);
    for (int i = 0; i < 2; i++) {
        rlist_free(&c->tl[i]);
        rlist_free(&c->tr[i]);
    }
    free(c);
}

/* segments longer than the counter capacity / 2 are refused by the callers
 * (short mode aborts above MAX_SEQ=1000, kmer.cpp:1006-1009). */
#define ORACLE_MAX_SEG 8000

typedef struct {
    int k_high, k_low;
    u128 seq_high, seq_low;
} cres;

/* dispatch on MAX_MER exactly as the reference does (kmer.cpp:100,180,298,774) */
static cres ctx_check(trew_oracle_ctx *c, const char *seq, int64_t st, int64_t nd, int min_mer, int max_mer,
                      rlist *high, rlist *low) {
    cres r;
    /* rebase so the core can use int indices */
    const char *base = seq + st;
    int n = (int) (nd - st + 1);
    if (c->p.max_mer <= 32) {
        check_res_64 q = kmer_check_64(&c->p, base, 0, n - 1, min_mer, max_mer, high, low, &c->m64);
        r.k_high = q.k_high;
        r.k_low = q.k_low;
        r.seq_high = q.seq_high;
        r.seq_low = q.seq_low;
    } else {
        check_res_128 q = kmer_check_128(&c->p, base, 0, n - 1, min_mer, max_mer, high, low, &c->m128);
        r.k_high = q.k_high;
        r.k_low = q.k_low;
        r.seq_high = q.seq_high;
        r.seq_low = q.seq_low;
    }
    return r;
}
static void ctx_target(trew_oracle_ctx *c, const char *seq, int64_t st, int64_t nd, int k, int is_high, rlist *out) {
    const char *base = seq + st;
    int n = (int) (nd - st + 1);
    if (c->p.max_mer <= 32)
        kmer_target_64(&c->p, base, 0, n - 1, k, is_high, out, &c->m64);
    else
        kmer_target_128(&c->p, base, 0, n - 1, k, is_high, out, &c->m128);
}
static u128 ctx_rot_rc(const trew_oracle_ctx *c, u128 w, int k) {
    (void) c;
    return rot_seq_128(revcomp_128(w, k), k);
}
static void flush(trew_oracle_ctx *c, rlist *l, int table) {
    for (int i = 0; i < l->n; i++) gtable_add(&c->t[table], l->e[i].k, l->e[i].w, l->e[i].c);
}
/* strand-canonical flush: KmerSeq{k, MIN(w, rot(rc(w)))}, e.g. kmer.cpp:379-382, 820-823 */
static void flush_canon(trew_oracle_ctx *c, rlist *l, int table) {
    for (int i = 0; i < l->n; i++) {
        u128 rv = ctx_rot_rc(c, l->e[i].w, l->e[i].k);
        gtable_add(&c->t[table], l->e[i].k, l->e[i].w < rv ? l->e[i].w : rv, l->e[i].c);
    }
}

#define IMIN(a, b) ((a) < (b) ? (a) : (b))
#define IMAX(a, b) ((a) > (b) ? (a) : (b))

/* buffer_task, kmer.cpp:111-173 (64-bit branch) == 188-250 (128-bit branch) */
void trew_oracle_add_short(trew_oracle_ctx *c, const char *buf, const int64_t *stv, const int64_t *ndv, int64_t nreads) {
    const int MIN_MER = c->p.min_mer, MAX_MER = c->p.max_mer;
    rlist *tl = c->tl, *tr = c->tr;
    for (int64_t r = 0; r < nreads; r++) {
        int64_t st = stv[r], nd = ndv[r];
        int n = (int) (nd - st + 1);
        if (n > ORACLE_MAX_SEG) continue;
        if (2 * MIN_MER <= n) { /* kmer.cpp:115 */
            cres left = {0, 0, 0, 0}, right = {0, 0, 0, 0};
            if (4 * MIN_MER <= n) { /* kmer.cpp:119 */
                left = ctx_check(c, buf, st, st + (n / 2) - 1, MIN_MER, IMIN(n / 4, MAX_MER), &tl[0], &tl[1]);
                if (left.k_high > 0 || left.k_low > 0) { /* kmer.cpp:123 */
                    right = ctx_check(c, buf, nd - ((n + 1) / 2) + 1, nd, MIN_MER, IMIN(n / 4, MAX_MER),
                                      left.k_high > 0 ? NULL : &tr[0], left.k_low > 0 ? NULL : &tr[1]);
                    if (left.k_high == right.k_high && left.k_high > 0) { /* kmer.cpp:128-130 */
                        rlist tmp;
                        rlist_init(&tmp);
                        ctx_target(c, buf, st, nd, left.k_high, 1, &tmp);
                        flush(c, &tmp, TREW_T_BOTH_HIGH);
                        rlist_free(&tmp);
                    } else { /* kmer.cpp:131-139 */
                        flush(c, &tl[0], TREW_T_FORWARD_HIGH);
                        flush(c, &tr[0], TREW_T_BACKWARD_HIGH);
                    }
                    if (left.k_low == right.k_low && left.k_low > 0) { /* kmer.cpp:141-143 */
                        rlist tmp;
                        rlist_init(&tmp);
                        ctx_target(c, buf, st, nd, left.k_low, 0, &tmp);
                        flush(c, &tmp, TREW_T_BOTH_LOW);
                        rlist_free(&tmp);
                    } else { /* kmer.cpp:144-152 */
                        flush(c, &tl[1], TREW_T_FORWARD_LOW);
                        flush(c, &tr[1], TREW_T_BACKWARD_LOW);
                    }
                    rlist_clear(&tr[0]);
                    rlist_clear(&tr[1]);
                } else { /* kmer.cpp:156-158: right half straight into result.backward */
                    right = ctx_check(c, buf, nd - ((n + 1) / 2) + 1, nd, MIN_MER, IMIN(n / 4, MAX_MER), &tr[0], &tr[1]);
                    flush(c, &tr[0], TREW_T_BACKWARD_HIGH);
                    flush(c, &tr[1], TREW_T_BACKWARD_LOW);
                    rlist_clear(&tr[0]);
                    rlist_clear(&tr[1]);
                }
                rlist_clear(&tl[0]);
                rlist_clear(&tl[1]);
            }
            int high_half_check = left.k_high == 0 && right.k_high == 0; /* kmer.cpp:165-166 */
            int low_half_check = left.k_low == 0 && right.k_low == 0;
            if (4 * MAX_MER > n && (high_half_check || low_half_check)) { /* kmer.cpp:168-171 */
                ctx_check(c, buf, st, nd, IMAX(n / 4 + 1, MIN_MER), IMIN(n / 2, MAX_MER),
                          high_half_check ? &tl[0] : NULL, low_half_check ? &tl[1] : NULL);
                flush(c, &tl[0], TREW_T_BOTH_HIGH);
                flush(c, &tl[1], TREW_T_BOTH_LOW);
                rlist_clear(&tl[0]);
                rlist_clear(&tl[1]);
            }
        }
    }
}

/* buffer_task_pair, kmer.cpp:322-507.  The 64-bit branch never clears
 * temp_result_left after the whole-read block (467-505) while its 128-bit twin
 * does (722-723); the stale entries leak into the next pair handled by the same
 * worker thread (a reference bug that makes results depend on thread
 * scheduling).  By default this restatement implements the cleared (128-bit twin)
 * semantics; with params.compat_g1 the 64-bit branch is followed as written (the
 * lists tl[] live in the context, so the entries reach the next pair exactly as
 * temp_result_left does in a single consumer thread).  It cannot trigger when
 * min(n1,n2) >= 4*MAX_MER. */
void trew_oracle_add_pair(trew_oracle_ctx *c, const char *buf1, const int64_t *st1v, const int64_t *nd1v,
                          const char *buf2, const int64_t *st2v, const int64_t *nd2v, int64_t npairs) {
    const int MIN_MER = c->p.min_mer, MAX_MER = c->p.max_mer;
    rlist *tl = c->tl, *tr = c->tr;
    for (int64_t r = 0; r < npairs; r++) {
        int64_t st1 = st1v[r], nd1 = nd1v[r], st2 = st2v[r], nd2 = nd2v[r];
        int n1 = (int) (nd1 - st1 + 1), n2 = (int) (nd2 - st2 + 1);
        int n = IMIN(n1, n2); /* kmer.cpp:331 */
        if (n1 > ORACLE_MAX_SEG || n2 > ORACLE_MAX_SEG) continue;
        if (2 * MIN_MER <= n) {
            int lef_k[2] = {0, 0};
            int k_mer[2] = {0, 0};
            u128 k_mer_seq[2] = {0, 0};
            if (4 * MIN_MER <= n) {
                /* fragment order R1-left, R1-right, R2-right, R2-left; kmer.cpp:338-340 */
                const char *bufv[4] = {buf1, buf1, buf2, buf2};
                int64_t sst[4] = {st1, nd1 - ((n1 + 1) / 2) + 1, nd2 - ((n2 + 1) / 2) + 1, st2};
                int64_t snd[4] = {st1 + (n1 / 2) - 1, nd1, nd2, st2 + (n2 / 2) - 1};
                const int snum = 4;
                int si[2] = {1, 1};
                int rend[2] = {0, 0};
                int ti, tj;
                for (ti = 1; ti <= snum && (!rend[0] || !rend[1]); ti++) { /* kmer.cpp:347-374 */
                    cres t = ctx_check(c, bufv[ti - 1], sst[ti - 1], snd[ti - 1], MIN_MER, IMIN(n / 4, MAX_MER),
                                       rend[0] ? NULL : (ti <= 2 ? &tl[0] : &tr[0]),
                                       rend[1] ? NULL : (ti <= 2 ? &tl[1] : &tr[1]));
                    int tk[2] = {t.k_high, t.k_low};
                    u128 ts[2] = {t.seq_high, t.seq_low};
                    for (int b = 0; b < 2; b++) {
                        if (!rend[b] && tk[b] > 0 &&
                            ((k_mer[b] == tk[b] && k_mer_seq[b] == dir_seq_128(ti, tk[b], ts[b], 1)) || ti == 1)) {
                            si[b] += 1;
                            k_mer[b] = tk[b];
                            if (ti == 1) k_mer_seq[b] = ts[b];
                            rend[b] = 0;
                        } else {
                            rend[b] = 1;
                        }
                    }
                }
                lef_k[0] = k_mer[0];
                lef_k[1] = k_mer[1];
                for (int b = 0; b < 2; b++) /* kmer.cpp:378-399 */
                    if (si[b] == snum + 1) {
                        flush_canon(c, &tl[b], b == 0 ? TREW_T_BOTH_HIGH : TREW_T_BOTH_LOW);
                        flush_canon(c, &tr[b], b == 0 ? TREW_T_BOTH_HIGH : TREW_T_BOTH_LOW);
                    }
                if (si[0] <= snum || si[1] <= snum) { /* kmer.cpp:401-436 */
                    int sj[2] = {snum, snum};
                    k_mer[0] = k_mer[1] = 0;
                    rend[0] = rend[1] = 0;
                    for (tj = snum; !rend[0] || !rend[1]; tj--) {
                        cres t = ctx_check(c, bufv[tj - 1], sst[tj - 1], snd[tj - 1], MIN_MER, IMIN(n / 4, MAX_MER),
                                           rend[0] ? NULL : (tj <= 2 ? &tr[0] : &tl[0]),
                                           rend[1] ? NULL : (tj <= 2 ? &tr[1] : &tl[1]));
                        int tk[2] = {t.k_high, t.k_low};
                        u128 ts[2] = {t.seq_high, t.seq_low};
                        for (int b = 0; b < 2; b++) {
                            if (sj[b] >= si[b] && !rend[b] && tk[b] > 0 &&
                                ((k_mer[b] == tk[b] && k_mer_seq[b] == dir_seq_128(tj, tk[b], ts[b], 0)) || tj == snum)) {
                                sj[b] -= 1;
                                k_mer[b] = tk[b];
                                if (tj == snum) k_mer_seq[b] = ts[b];
                                rend[b] = 0;
                            } else {
                                rend[b] = 1;
                            }
                        }
                    }
                }
                for (int b = 0; b < 2; b++) /* kmer.cpp:438-455 */
                    if (si[b] <= snum) {
                        flush(c, &tl[b], b == 0 ? TREW_T_FORWARD_HIGH : TREW_T_FORWARD_LOW);
                        flush(c, &tr[b], b == 0 ? TREW_T_BACKWARD_HIGH : TREW_T_BACKWARD_LOW);
                    }
                for (int b = 0; b < 2; b++) {
                    rlist_clear(&tl[b]);
                    rlist_clear(&tr[b]);
                }
            }
            if (4 * MAX_MER > n) { /* kmer.cpp:467-505 */
                cres lt = {0, 0, 0, 0}, rt = {0, 0, 0, 0};
                if (lef_k[0] == 0 || lef_k[1] == 0)
                    lt = ctx_check(c, buf1, st1, nd1, IMAX(n / 4 + 1, MIN_MER), IMIN(n / 2, MAX_MER),
                                   lef_k[0] == 0 ? &tl[0] : NULL, lef_k[1] == 0 ? &tl[1] : NULL);
                if (k_mer[0] == 0 || k_mer[1] == 0)
                    rt = ctx_check(c, buf2, st2, nd2, IMAX(n / 4 + 1, MIN_MER), IMIN(n / 2, MAX_MER),
                                   k_mer[0] == 0 ? &tl[0] : NULL, k_mer[1] == 0 ? &tl[1] : NULL);
                int ltk[2] = {lt.k_high, lt.k_low}, rtk[2] = {rt.k_high, rt.k_low};
                u128 lts[2] = {lt.seq_high, lt.seq_low}, rts[2] = {rt.seq_high, rt.seq_low};
                for (int b = 0; b < 2; b++)
                    if (lef_k[b] == 0 && k_mer[b] == 0 && ltk[b] == rtk[b] && ltk[b] > 0 &&
                        lts[b] == ctx_rot_rc(c, rts[b], rtk[b]))
                        flush_canon(c, &tl[b], b == 0 ? TREW_T_BOTH_HIGH : TREW_T_BOTH_LOW);
                flush(c, &tl[0], TREW_T_FORWARD_HIGH);
                flush(c, &tl[1], TREW_T_FORWARD_LOW);
                if (!(c->p.compat_g1 && MAX_MER <= 32)) { /* the 128-bit twin's clear, kmer.cpp:722-723; the 64-bit branch has none */
                    rlist_clear(&tl[0]);
                    rlist_clear(&tl[1]);
                }
            }
        }
    }
}

/* buffer_task_long, kmer.cpp:785-871 */
void trew_oracle_add_long(trew_oracle_ctx *c, const char *buf, const int64_t *stv, const int64_t *ndv, int64_t nreads) {
    const int MIN_MER = c->p.min_mer, MAX_MER = c->p.max_mer, SLICE_LENGTH = c->p.slice_len;
    rlist *tl = c->tl, *tr = c->tr;
    for (int64_t r = 0; r < nreads; r++) {
        int64_t tst = stv[r], tnd = ndv[r];
        int64_t len = tnd - tst + 1;
        int snum = (int) (len / SLICE_LENGTH); /* kmer.cpp:790-792 */
        int mid = (snum + 1) / 2;
        int mid_bonus_sl = (int) (len % SLICE_LENGTH);
        int si[2] = {1, 1}, k_mer[2] = {0, 0}, rend[2] = {0, 0};
        int sl = 0, ti, tj;
        for (ti = 1; ti <= snum && (!rend[0] || !rend[1]); ti++, tst += sl) { /* kmer.cpp:797-817 */
            sl = SLICE_LENGTH + (ti == mid ? mid_bonus_sl : 0);
            cres t = ctx_check(c, buf, tst, tst + sl - 1, MIN_MER, MAX_MER, rend[0] ? NULL : &tl[0], rend[1] ? NULL : &tl[1]);
            int tk[2] = {t.k_high, t.k_low};
            for (int b = 0; b < 2; b++) {
                if (!rend[b] && tk[b] > 0 && (k_mer[b] == tk[b] || ti == 1)) {
                    si[b] += 1;
                    k_mer[b] = tk[b];
                    rend[b] = 0;
                } else {
                    rend[b] = 1;
                }
            }
        }
        for (int b = 0; b < 2; b++) /* kmer.cpp:819-830 */
            if (si[b] == snum + 1) flush_canon(c, &tl[b], b == 0 ? TREW_T_BOTH_HIGH : TREW_T_BOTH_LOW);
        if (si[0] <= snum || si[1] <= snum) { /* kmer.cpp:832-868 */
            int sj[2] = {snum, snum};
            k_mer[0] = k_mer[1] = 0;
            rend[0] = rend[1] = 0;
            for (tj = snum; !rend[0] || !rend[1]; tj--, tnd -= sl) {
                sl = SLICE_LENGTH + (tj == mid ? mid_bonus_sl : 0);
                cres t = ctx_check(c, buf, tnd - sl + 1, tnd, MIN_MER, MAX_MER, rend[0] ? NULL : &tr[0], rend[1] ? NULL : &tr[1]);
                flush(c, &tr[0], TREW_T_BACKWARD_HIGH); /* recorded straight into result.backward, kmer.cpp:840 */
                flush(c, &tr[1], TREW_T_BACKWARD_LOW);
                rlist_clear(&tr[0]);
                rlist_clear(&tr[1]);
                int tk[2] = {t.k_high, t.k_low};
                for (int b = 0; b < 2; b++) {
                    if (sj[b] >= si[b] && !rend[b] && tk[b] > 0 && (k_mer[b] == tk[b] || tj == snum)) {
                        sj[b] -= 1;
                        k_mer[b] = tk[b];
                        rend[b] = 0;
                    } else {
                        rend[b] = 1;
                    }
                }
            }
            if (si[0] <= snum) flush(c, &tl[0], TREW_T_FORWARD_HIGH);
            if (si[1] <= snum) flush(c, &tl[1], TREW_T_FORWARD_LOW);
        }
        rlist_clear(&tl[0]);
        rlist_clear(&tl[1]);
    }
}

/* thread merge, kmer.cpp:1486-1515 */
void trew_oracle_merge(trew_oracle_ctx *dst, const trew_oracle_ctx *src) {
    for (int t = 0; t < TREW_NUM_TABLES; t++) {
        const gtable *g = &src->t[t];
        for (uint64_t i = 0; i < g->cap; i++)
            if (g->k[i]) gtable_add(&dst->t[t], g->k[i], g->w[i], g->c[i]);
    }
}

int64_t trew_oracle_table_size(const trew_oracle_ctx *c, int table) { return (int64_t) c->t[table].n; }

int64_t trew_oracle_table_rows(const trew_oracle_ctx *c, int table, trew_oracle_row *rows, int64_t cap) {
    const gtable *g = &c->t[table];
    int64_t n = 0;
    for (uint64_t i = 0; i < g->cap && n < cap; i++)
        if (g->k[i]) {
            rows[n].k = g->k[i];
            rows[n].pad = 0;
            rows[n].word_lo = (uint64_t) g->w[i];
            rows[n].word_hi = (uint64_t) (g->w[i] >> 64);
            rows[n].count = g->c[i];
            n++;
        }
    return n;
}

/* ---------- exported primitives ---------- */
static u128 mk128(uint64_t lo, uint64_t hi) { return ((u128) hi << 64) | lo; }

void trew_oracle_rot_seq(uint64_t lo, uint64_t hi, int k, uint64_t *out_lo, uint64_t *out_hi) {
    u128 r = rot_seq_128(mk128(lo, hi), k);
    *out_lo = (uint64_t) r;
    *out_hi = (uint64_t) (r >> 64);
}
void trew_oracle_revcomp(uint64_t lo, uint64_t hi, int k, uint64_t *out_lo, uint64_t *out_hi) {
    u128 r = revcomp_128(mk128(lo, hi), k);
    *out_lo = (uint64_t) r;
    *out_hi = (uint64_t) (r >> 64);
}
int trew_oracle_repeat_check(uint64_t lo, uint64_t hi, int k) { return repeat_check_128(mk128(lo, hi), k); }

/* check_ans_seq, kmer.cpp:2549-2569: reject a k-mer that is a pure repetition
 * of a shorter unit of length j in [ABS_MIN_MER(=3), MIN_MER): for such j all
 * k-j+1 length-j windows of the word fall into one rotation class. */
int trew_oracle_check_ans_seq(uint64_t lo, uint64_t hi, int k, int min_mer) {
    u128 seq = mk128(lo, hi);
    for (int j = 3; j < min_mer; j++) {
        u128 num = seq, bef = 0;
        int i;
        for (i = 0; i < k - j + 1; i++) {
            u128 t = rot_seq_128(num & kmask_128(j), j);
            if (i > 0 && t != bef) break;
            bef = t;
            num >>= 2;
        }
        if (i == k - j + 1) return 0;
    }
    return 1;
}

int trew_oracle_segment_check(const trew_oracle_params *p, const char *seq, int st, int nd, int min_mer, int max_mer,
                              int *k_high, int *k_low, uint64_t seq_high[2], uint64_t seq_low[2],
                              trew_oracle_row *hist_high, int *n_high, trew_oracle_row *hist_low, int *n_low, int cap) {
    trew_oracle_ctx *c = trew_oracle_new(p);
    rlist h, l;
    rlist_init(&h);
    rlist_init(&l);
    cres r = ctx_check(c, seq, st, nd, min_mer, max_mer, &h, &l);
    *k_high = r.k_high;
    *k_low = r.k_low;
    seq_high[0] = (uint64_t) r.seq_high;
    seq_high[1] = (uint64_t) (r.seq_high >> 64);
    seq_low[0] = (uint64_t) r.seq_low;
    seq_low[1] = (uint64_t) (r.seq_low >> 64);
    *n_high = *n_low = 0;
    for (int i = 0; i < h.n && i < cap; i++, (*n_high)++) {
        hist_high[i].k = h.e[i].k;
        hist_high[i].pad = 0;
        hist_high[i].word_lo = (uint64_t) h.e[i].w;
        hist_high[i].word_hi = (uint64_t) (h.e[i].w >> 64);
        hist_high[i].count = h.e[i].c;
    }
    for (int i = 0; i < l.n && i < cap; i++, (*n_low)++) {
        hist_low[i].k = l.e[i].k;
        hist_low[i].pad = 0;
        hist_low[i].word_lo = (uint64_t) l.e[i].w;
        hist_low[i].word_hi = (uint64_t) (l.e[i].w >> 64);
        hist_low[i].count = l.e[i].c;
    }
    rlist_free(&h);
    rlist_free(&l);
    trew_oracle_free(c);
    return 0;
}

int trew_oracle_segment_stats(const trew_oracle_params *p, const char *seq, int st, int nd, int min_mer, int max_mer,
                              uint32_t *count, uint32_t *maxc, uint64_t *maxseq_lo, uint64_t *maxseq_hi) {
    segmap_128 m;
    segmap_init_128(&m, 1u << 14);
    (void) p;
    for (int k = min_mer; k <= max_mer; k++) {
        kstat_128 s = count_k_128(seq, st, nd, k, 0, -1.0, &m);
        count[k - min_mer] = s.count;
        maxc[k - min_mer] = s.maxc;
        maxseq_lo[k - min_mer] = (uint64_t) s.maxseq;
        maxseq_hi[k - min_mer] = (uint64_t) (s.maxseq >> 64);
    }
    segmap_free_128(&m);
    return 0;
}

/* ---------- multi-threaded short-mode run (timed CPU baseline) ---------- */
typedef struct {
    trew_oracle_ctx *c;
    const char *buf;
    const int64_t *st, *nd;
    int64_t n;
    const trew_oracle_params *p;
} mt_arg;

static void *mt_worker(void *a_) {
    mt_arg *a = (mt_arg *) a_;
    a->c = trew_oracle_new(a->p); /* per-thread tables are allocated and cleared by the thread that uses them */
    trew_oracle_add_short(a->c, a->buf, a->st, a->nd, a->n);
    return NULL;
}

trew_oracle_ctx *trew_oracle_run_short_mt(const trew_oracle_params *p, const char *buf, const int64_t *st,
                                          const int64_t *nd, int64_t n, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    pthread_t *th = (pthread_t *) malloc(sizeof(pthread_t) * (size_t) nthreads);
    mt_arg *args = (mt_arg *) malloc(sizeof(mt_arg) * (size_t) nthreads);
    int64_t per = (n + nthreads - 1) / nthreads;
    for (int t = 0; t < nthreads; t++) {
        int64_t lo = per * t, hi = lo + per;
        if (lo > n) lo = n;
        if (hi > n) hi = n;
        args[t].c = NULL;
        args[t].p = p;
        args[t].buf = buf;
        args[t].st = st + lo;
        args[t].nd = nd + lo;
        args[t].n = hi - lo;
        pthread_create(&th[t], NULL, mt_worker, &args[t]);
    }
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    for (int t = 1; t < nthreads; t++) {
        trew_oracle_merge(args[0].c, args[t].c);
        trew_oracle_free(args[t].c);
    }
    trew_oracle_ctx *r = args[0].c;
    free(th);
    free(args);
    return r;
}
