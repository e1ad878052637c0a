/*
 * trew_oracle.h -- CPU ORACLE for the TREW per-read tandem-repeat scan.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference
 * algorithm (Chemical118/TREW @ 2025-02-18, src/kmer.cpp) used as the checker
 * for the HIP path.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; the product (trew_amd/, include/) never does.
 *
 * Pinning: the reference itself is unbuildable in the build image (it needs
 * abseil, oneTBB and argparse, none of which are installed and none of which
 * may be replaced by stand-ins), so this restatement is pinned by
 *   - the reference's own known-answer tests  test/test.cpp:83-97, 172-258,
 *   - the bundled fixtures test/test.fastq, test/test_long.fastq (copied as
 *     data under tests/golden/), whose expected output shape and rows are
 *     recorded in SURVEY.md section 8(c),
 *   - the per-segment example vector of SURVEY.md section 7.
 * See DESIGN.md "Oracle".
 *
 * Table ids (reference ResultMapData = {forward,backward,both} x {high(first),
 * low(second)}, kmer.h:79-81):
 */
#ifndef TREW_ORACLE_H
#define TREW_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    TREW_T_FORWARD_HIGH = 0,
    TREW_T_FORWARD_LOW = 1,
    TREW_T_BACKWARD_HIGH = 2,
    TREW_T_BACKWARD_LOW = 3,
    TREW_T_BOTH_HIGH = 4,
    TREW_T_BOTH_LOW = 5,
    TREW_NUM_TABLES = 6
};

typedef struct {
    int min_mer;      /* MIN_MER  (trew.cpp:165,246) */
    int max_mer;      /* MAX_MER                      */
    double low;       /* LOW_BASELINE  (-L, default 0.5) */
    double high;      /* HIGH_BASELINE (-H, default 0.8) */
    int slice_len;    /* SLICE_LENGTH (-s, long mode, default 150) */
    int use_break;    /* 1 = keep the reference's early break (kmer.cpp:2207-2210) */
    int compat_g1;    /* 1 = the 64-bit pair branch as written: temp_result_left is NOT cleared after the whole-read block
                         (kmer.cpp:467-505), so its entries are added once more by the next pair this context is given
                         (SURVEY G1).  Pairs must then be added in file order to ONE context: that is the reference with a
                         single consumer thread.  No effect for MAX_MER > 32 (the 128-bit twin clears, kmer.cpp:722-723). */
} trew_oracle_params;

/* one (k, word) -> count row; word is the 2k-bit k-mer, first base most
 * significant, split into two 64-bit halves (hi is 0 for k <= 32). */
typedef struct {
    int32_t k;
    int32_t pad;
    uint64_t word_lo;
    uint64_t word_hi;
    uint64_t count;
} trew_oracle_row;

/* ---- primitives (kmer.cpp:14-70, 1815-1884) ---- */
int trew_oracle_code(unsigned char c);                       /* codes[] kmer.cpp:14-31 */
void trew_oracle_rot_seq(uint64_t lo, uint64_t hi, int k, uint64_t *out_lo, uint64_t *out_hi);  /* get_rot_seq(_128) */
void trew_oracle_revcomp(uint64_t lo, uint64_t hi, int k, uint64_t *out_lo, uint64_t *out_hi);  /* reverse_complement_* >> 2*(W-k) */
int trew_oracle_repeat_check(uint64_t lo, uint64_t hi, int k);                                   /* get_repeat_check */
int trew_oracle_check_ans_seq(uint64_t lo, uint64_t hi, int k, int min_mer);                     /* check_ans_seq kmer.cpp:2549 */

/* ---- per-segment detector, k_mer_check (kmer.cpp:2144-2344 / 2346-2547) ----
 * seq[st..nd] inclusive.  Returns 0.  hist_* receive every class of the
 * winning k (rotation-canonical key), at most cap rows each.               */
int trew_oracle_segment_check(const trew_oracle_params *p, const char *seq, int st, int nd,
                              int min_mer, int max_mer,
                              int *k_high, int *k_low,
                              uint64_t seq_high[2], uint64_t seq_low[2],
                              trew_oracle_row *hist_high, int *n_high,
                              trew_oracle_row *hist_low, int *n_low, int cap);

/* per-k statistics of one segment: COUNT, MAX, MAX_SEQ (kmer.cpp:2183-2216),
 * without the early break.  arrays are indexed k - min_mer.                 */
int trew_oracle_segment_stats(const trew_oracle_params *p, const char *seq, int st, int nd,
                              int min_mer, int max_mer,
                              uint32_t *count, uint32_t *maxc, uint64_t *maxseq_lo, uint64_t *maxseq_hi);

/* ---- per-read drivers (buffer_task*, kmer.cpp:80-985) accumulating the six tables ---- */
typedef struct trew_oracle_ctx trew_oracle_ctx;

trew_oracle_ctx *trew_oracle_new(const trew_oracle_params *p);
void trew_oracle_free(trew_oracle_ctx *c);
/* short single-end: buffer_task kmer.cpp:111-173 */
void trew_oracle_add_short(trew_oracle_ctx *c, const char *buf, const int64_t *st, const int64_t *nd, int64_t n);
/* short paired-end: buffer_task_pair kmer.cpp:322-507 (with the 128-bit twin's clear, 722-723, unless compat_g1) */
void trew_oracle_add_pair(trew_oracle_ctx *c, const char *buf1, const int64_t *st1, const int64_t *nd1,
                          const char *buf2, const int64_t *st2, const int64_t *nd2, int64_t n);
/* long: buffer_task_long kmer.cpp:785-871 */
void trew_oracle_add_long(trew_oracle_ctx *c, const char *buf, const int64_t *st, const int64_t *nd, int64_t n);
/* merge src into dst (thread merge, kmer.cpp:1486-1515) */
void trew_oracle_merge(trew_oracle_ctx *dst, const trew_oracle_ctx *src);
int64_t trew_oracle_table_size(const trew_oracle_ctx *c, int table);
int64_t trew_oracle_table_rows(const trew_oracle_ctx *c, int table, trew_oracle_row *rows, int64_t cap);

/* multi-threaded short-mode run over in-memory reads (used for the timed CPU
 * baseline): splits the reads into nthreads contiguous ranges, one ctx each,
 * merges into the returned ctx.                                             */
trew_oracle_ctx *trew_oracle_run_short_mt(const trew_oracle_params *p, const char *buf,
                                          const int64_t *st, const int64_t *nd, int64_t n, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
