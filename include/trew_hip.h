/*
 * trew_hip.h -- C ABI of the MI355X-native TREW scan (libtrew_hip.so).
 *
 * Drop-in boundary for the hot path of Chemical118/TREW (reference @ 2025-02-18):
 * the per-read tandem-repeat detector of src/kmer.cpp.  The reference has no
 * FFI; its de-facto seam is the kmer.h function set between trew.cpp and
 * kmer.cpp.  Each entry point below names the
 * reference interface it replaces.  Plain pointers and sizes only; no C++ or
 * torch types.  All functions return 0 on success, non-zero on error
 * (trew_hip_last_error() gives the text); the CLI layer turns a non-zero status
 * into the reference's "message on stderr + exit(EXIT_FAILURE)" convention
 * (kmer.cpp:84-87, 1007-1008).
 *
 * Packed read format ("bit planes", produced by trew_pack_reads or by the
 * device-side generator): a read of n bases occupies 3*ceil(n/32) 32-bit words,
 * one {lo, hi, nmask} triple per 32 bases; bit i of a triple's words describes
 * base 32*j+i.  code = 2*hi+lo with T=0 G=1 C=2 A=3 (codes[], kmer.cpp:14-31);
 * nmask bit = 1 for any other byte (N, lower/upper IUPAC, '\r', bytes >= 0x80).
 * Bits past the end of the read are zero in lo/hi and may be anything in nmask.
 */
#ifndef TREW_HIP_H
#define TREW_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TREW_HIP_ABI_VERSION 4

/* scan modes: which per-read driver of the reference is reproduced */
enum {
    TREW_MODE_SHORT = 0,   /* buffer_task       kmer.cpp:80-266  (trew short)              */
    TREW_MODE_PAIR = 1,    /* buffer_task_pair  kmer.cpp:268-745 (trew short --paired_end) */
    TREW_MODE_LONG = 2,    /* buffer_task_long  kmer.cpp:747-985 (trew long)               */
    TREW_MODE_SEGMENT = 3  /* k_mer_check on every read as one segment, kmer.h:232-236     */
};

/* result tables: ResultMapData = {forward, backward, both} x {high(first), low(second)}, kmer.h:79-81 */
enum {
    TREW_TABLE_FORWARD_HIGH = 0,
    TREW_TABLE_FORWARD_LOW = 1,
    TREW_TABLE_BACKWARD_HIGH = 2,
    TREW_TABLE_BACKWARD_LOW = 3,
    TREW_TABLE_BOTH_HIGH = 4,
    TREW_TABLE_BOTH_LOW = 5,
    TREW_NUM_TABLES = 6
};

/* debug / test flags for trew_hip_params.flags */
enum {
    TREW_FLAG_NO_FILTER = 1, /* skip the bucket-bound prefilter: every k is a candidate (exact path only) */
    TREW_FLAG_DEBUG_NO_EMIT = 2, /* timing experiments only: drop every table update (results are empty) */
    TREW_FLAG_DEBUG_NO_KLOOP = 4, /* timing experiments only: prefilter without its k loop (nothing is flagged) */
    TREW_FLAG_DEBUG_POISON_LDS = 32, /* tests: the exact kernel starts from garbage-filled LDS */
    TREW_FLAG_DEBUG_WIDE_NO_WAIT = 128, /* tests: the wide table (k > 32) never waits for a claimed slot's ready bit -- every such
                                  wait counts as timed out (trew_hip_debug_counters), duplicates are left for collect to merge */
    TREW_FLAG_NO_TIMING = 64, /* no HIP events around the kernels (trew_hip_last_timing is unavailable): for hosts that
                                submit ~10^4 small batches a second and are bound by API calls */
    TREW_FLAG_COMPAT_G1 = 512, /* pair mode, MAX_MER <= 32, n_slots = 1 only: follow the reference's 64-bit pair branch as written --
                                temp_result_left is not cleared after the whole-read block (kmer.cpp:467-505; the 128-bit twin
                                clears it, 722-723), so what that block recorded is added once more by the next pair: to `both`
                                if that pair's four segments chain, else to `forward` (kmer.cpp:378-399, 438-455).  Batches must
                                be submitted in file order; this reproduces the reference run with ONE consumer thread (with
                                more its output depends on scheduling).  Default: the cleared semantics. */
    TREW_FLAG_DEBUG_NO_JOINT = 2048, /* tests: the prefilter's uniform path judges every segment in a k loop of its own instead of both
                                halves of a read in one (filter_halves_uni); the flagged reads may differ by a few (odd lengths use a
                                joint threshold), the tables never */
    TREW_FLAG_DEBUG_NO_GROUP = 1024, /* tests and A/B runs: the exact kernel gives every segment a wave of its own (decide()) instead
                                of deciding four segments in lock step, 16 lanes each (decide_group); results are identical */
    TREW_FLAG_TRACK_PRESSURE = 256 /* every batch ends with a copy of the table's fill counters into pinned host memory, and
                                trew_hip_table_pressure answers from those copies (and from what collect / add_rows /
                                reset read since) instead of asking the device: for hosts that ask before every batch.
                                Like the device query, the answer does not include batches still in flight. */
};

/* Replaces the eight configuration globals MIN_MER ... HIGH_BASELINE
 * (kmer.h:55-63, set in trew.cpp:165-172, 246-253). */
typedef struct {
    int32_t min_mer;          /* MIN_MER, >= 3 (ABS_MIN_MER)                                  */
    int32_t max_mer;          /* MAX_MER, <= 64; > 32 selects the 128-bit-word kernels         */
    double low_baseline;      /* LOW_BASELINE  (-L)                                           */
    double high_baseline;     /* HIGH_BASELINE (-H)                                           */
    int32_t slice_length;     /* SLICE_LENGTH (-s), long mode only                            */
    int32_t mode;             /* TREW_MODE_*                                                  */
    int32_t device;           /* HIP device ordinal                                           */
    int32_t n_slots;          /* batch slots (one HIP stream each), 1 .. 512                  */
    uint64_t max_batch_words; /* capacity of one slot's packed-read buffer, 32-bit words      */
    uint64_t max_batch_reads; /* capacity of one slot, reads (pairs count as two)             */
    uint32_t table_log2_slots;/* device count table: 2^table_log2_slots entries (>= 12)       */
    uint32_t flags;           /* TREW_FLAG_*                                                  */
    uint64_t max_batch_ascii_bytes; /* ABI 3: capacity of one slot for trew_hip_submit_ascii (sequence bytes + 12 B per
                                 read of index arrays); 0 = text batches are not used            */
} trew_hip_params;

/* Replaces QueueData / PairQueueData + LocationVector (kmer.h:73, 93-103): one
 * chunk of reads handed to the consumer.  The caller keeps ownership of every
 * pointer until trew_hip_wait(slot) returns.  With on_device != 0 the pointers
 * are device pointers and nothing is copied.
 * Host batches: when the three arrays lie back to back in one (pinned) buffer, laid out [offsets][lengths][words]
 * (lengths == offsets + n_reads, words == lengths + n_reads) or [words][offsets][lengths] (offsets == words +
 * n_words, lengths == offsets + n_reads), the batch is shipped with a single asynchronous copy; any other layout
 * works too and costs three. */
typedef struct {
    const uint32_t *words;    /* packed triples                                               */
    uint64_t n_words;
    const uint32_t *offsets;  /* word offset of each read, or NULL: read r starts at r*uniform_stride */
    const uint32_t *lengths;  /* bases of each read, or NULL: every read has uniform_length   */
    uint32_t uniform_length;
    uint32_t uniform_stride;
    uint64_t n_reads;         /* number of reads; in pair mode reads 2i and 2i+1 are mates (R1, R2) */
    int32_t on_device;
    int32_t max_length;       /* on_device + lengths only: longest read of the batch (0 = unknown) */
} trew_hip_batch;

/* The same chunk as TEXT, closest to what the reference's consumers pop (QueueData: a char buffer plus the [st, nd]
 * locations of the sequence lines, kmer.h:93-96): the bytes of the sequence lines only, in one pinned host buffer, and the
 * DEVICE applies codes[] (kmer.cpp:14-31) -- a pack kernel in front of the prefilter -- so that host threads only locate
 * lines and copy bytes.  Two shapes:
 *   uniform  byte_offsets == lengths == word_offsets == NULL: read r is bases[r * uniform_length .. + uniform_length)
 *   ragged   read r is bases[byte_offsets[r] .. + lengths[r]); word_offsets[r] = 3 * sum_{q<r} ceil(lengths[q] / 32), the
 *            place of its first packed triple (a running sum the host has anyway)
 * When the arrays lie back to back as [word_offsets][byte_offsets][lengths][bases] the batch is shipped with one copy.
 * The caller keeps ownership until trew_hip_wait(slot).  n_bytes + 12 * n_reads <= max_batch_ascii_bytes. */
typedef struct {
    const char *bases;
    uint64_t n_bytes;
    const uint32_t *byte_offsets;
    const uint32_t *lengths;
    const uint32_t *word_offsets;
    uint32_t uniform_length;
    uint32_t reserved;
    uint64_t n_reads;         /* pair mode: reads 2i and 2i+1 are mates                       */
} trew_hip_ascii_batch;

/* one (k, word) -> count row; word = the 2k-bit k-mer, first base most
 * significant (KmerSeq, kmer.h:77); word_hi is 0 for k <= 32. */
typedef struct {
    int32_t k;
    int32_t table;
    uint64_t word_lo;
    uint64_t word_hi;
    uint64_t count;
} trew_hip_row;

typedef struct trew_hip_ctx trew_hip_ctx;

/* Replaces set_extract_k_mer / set_rotation_table / ThreadData set-up,
 * trew.cpp:382-406: allocates streams, device buffers, the count table. */
int trew_hip_init(const trew_hip_params *params, trew_hip_ctx **out);
void trew_hip_destroy(trew_hip_ctx *ctx);
/* Text of the calling thread's last failing call (several host threads may share a context, one slot each);
 * ctx may be NULL: last trew_hip_init error. */
const char *trew_hip_last_error(const trew_hip_ctx *ctx);

/* Replaces one pop+process iteration of buffer_task* (kmer.cpp:106-177):
 * asynchronously copies the batch (unless on_device), runs the prefilter and
 * the exact kernel on the slot's stream, accumulating into the device tables. */
int trew_hip_submit(trew_hip_ctx *ctx, const trew_hip_batch *batch, int slot);
/* trew_hip_submit for a chunk of text: copies the batch, packs it on the device (bit planes identical to
 * trew_pack_reads, word for word) and runs the two kernels on the packed reads. */
int trew_hip_submit_ascii(trew_hip_ctx *ctx, const trew_hip_ascii_batch *batch, int slot);
/* Diagnostic: packs `batch` on the device exactly as trew_hip_submit_ascii does and copies the packed words back
 * (words_cap words available; *n_words = words the batch packs to).  Tests compare them with trew_pack_reads. */
int trew_hip_pack_ascii(trew_hip_ctx *ctx, const trew_hip_ascii_batch *batch, uint32_t *words, uint64_t words_cap, uint64_t *n_words);
/* Blocks until the slot's work is done (tasks.wait, kmer.cpp:1323-1325). */
int trew_hip_wait(trew_hip_ctx *ctx, int slot);

/* Replaces the thread merge of process_output (kmer.cpp:1486-1515): waits for
 * every slot and returns the rows of one table (any order).  *n_rows receives
 * the number of rows in the table even when it exceeds cap. */
int trew_hip_collect(trew_hip_ctx *ctx, int table, trew_hip_row *rows, uint64_t cap, uint64_t *n_rows);
/* Clears the six tables (start of a new file, kmer.cpp:89). */
int trew_hip_reset_tables(trew_hip_ctx *ctx);
/* Adds rows (e.g. another rank's tables) into the device tables. */
int trew_hip_add_rows(trew_hip_ctx *ctx, const trew_hip_row *rows, uint64_t n_rows);

/* ---- cross-GPU reduction of the tables; the reference's only counterpart is the
 * single-threaded map merge of process_output, kmer.cpp:1486-1515, and the cross-file merge, trew.cpp:454-467 ----
 * Counts are sums, so a table is reduced by adding every other table's rows into it.
 *
 * trew_hip_collect_device: device-to-device form of trew_hip_collect(table = -1) for an exchange that never
 * leaves HBM (one process per GPU: the caller all_gathers d_rows with RCCL).  d_rows is a device buffer of cap
 * rows on the context's GPU; *n_rows receives the number of rows there are (nothing is written past cap, call
 * again with a larger buffer).  Rows may repeat a key (spilled rows); every consumer below merges by adding.
 * trew_hip_add_rows_device: trew_hip_add_rows for rows that already live on the context's GPU.
 * trew_hip_merge: one process driving several GPUs (`trew --devices 0,1,...`): adds every row of src's tables
 * into dst's tables with one peer copy (xGMI between two GPUs); src is left unchanged. */
int trew_hip_collect_device(trew_hip_ctx *ctx, trew_hip_row *d_rows, uint64_t cap, uint64_t *n_rows);
int trew_hip_add_rows_device(trew_hip_ctx *ctx, const trew_hip_row *d_rows, uint64_t n_rows);
int trew_hip_merge(trew_hip_ctx *dst, trew_hip_ctx *src);
/* The whole exchange behind ONE collective (ABI 3).  Every rank owns one slice of 1 + slice_rows rows of a gather
 * buffer: row 0 is the slice's header (count = number of rows the rank has, everything else 0), rows 1.. are what
 * trew_hip_collect_device wrote.  After a single all_gather of the slices, d_buf holds n_slices of them and this call
 * adds the rows of every slice but own_slice into the context's tables with one kernel over the whole buffer
 * (preceded by a validation pass on the same stream: a row out of range fails the call and NOTHING is added).
 * producer_stream: the HIP stream (hipStream_t) the collective ran on -- the kernels are ordered behind it on the
 * device, no host synchronisation in between; NULL = wait for the whole device first.
 * *max_rows receives the largest header count.  If it exceeds slice_rows some rank's rows did not fit: nothing was
 * added anywhere (every rank sees the same headers), the call returns 0 and the caller repeats the exchange with
 * larger slices.  Matches the thread merge of process_output, kmer.cpp:1486-1515 (sums over contributors). */
int trew_hip_add_gathered_device(trew_hip_ctx *ctx, const trew_hip_row *d_buf, uint32_t n_slices, uint32_t own_slice,
                                 uint64_t slice_rows, void *producer_stream, uint64_t *max_rows);
/* The producing side of that exchange with no host hop (ABI 4): compacts the context's tables straight into rows 1.. of
 * d_slice (1 + slice_rows rows of device memory) and writes the header row -- count = rows the rank has, which may
 * exceed slice_rows (then only slice_rows of them are there and trew_hip_add_gathered_device reports it on every rank) --
 * with a kernel of its own behind the compaction; the spill log is appended on the device as well.  consumer_stream:
 * the HIP stream the collective will be issued on; it is made to wait for the header on the device (and this call first
 * waits, on the device, for what that stream still has queued on the slice), so the host neither reads the count nor
 * writes the header.  NULL = synchronise the device before and after instead.  n_rows may be NULL; asking for the count
 * costs one host synchronisation.  Replaces the size exchange a merge of per-thread maps needs (kmer.cpp:1486-1515). */
int trew_hip_collect_slice_device(trew_hip_ctx *ctx, trew_hip_row *d_slice, uint64_t slice_rows, void *consumer_stream, uint64_t *n_rows);

/* Fill state of the device tables (a snapshot; does not wait for running batches).  The reference's hash maps
 * grow without bound (absl::flat_hash_map, kmer.h:79); the device table has a fixed number of slots, rows that
 * find their partition full go to a spill log of spill_capacity rows, and only a full log loses counts (then
 * trew_hip_collect fails).  A host that scans unbounded input calls this between batches and, when
 * used_slots nears total_slots or spilled_rows > 0, drains: trew_hip_collect, keep the rows, trew_hip_reset_tables.
 * Any pointer may be NULL.  With TREW_FLAG_TRACK_PRESSURE the call touches no device (see the flag). */
int trew_hip_table_pressure(trew_hip_ctx *ctx, uint64_t *used_slots, uint64_t *total_slots, uint64_t *spilled_rows,
                            uint64_t *spill_capacity);

/* How often the kernels took their rare fall-back paths since the last trew_hip_reset_tables (waits for every slot).
 * Diagnostic: tests assert that each path is live code and that results still equal the oracle when it runs.
 * (The three fall-back counters are kept per DEVICE: contexts that share a GPU share them.)
 * out[0] decide(): speculative skip refused, segment decided again with every k counted
 * out[1] eval_runs(): more than 64 runs of adjacent same-class windows, classes counted window by window
 * out[2] wide table (k > 32): gave up waiting for a slot's ready bit (collect merges the duplicate slot this can leave)
 * out[3] keys inserted into the narrow table, out[4] into the wide table
 * out[5] decide_group(): a 16-lane row gave its segment back to decide() (an N where a class count was needed, a k with
 *        more than 16 runs and no skip slot left, a failed skip check)
 * out[6] reads of the group pass routed and recorded by the wave-per-segment code, out[7] whose k_mer_target was counted by it.
 * n <= TREW_DEBUG_COUNTERS entries are written. */
#define TREW_DEBUG_COUNTERS 8
int trew_hip_debug_counters(trew_hip_ctx *ctx, uint64_t *out, int n);

/* Diagnostic: the unit indices (reads, or pairs in pair mode) the prefilter of the last submit on `slot` handed to the exact
 * kernel, in worklist order (after trew_hip_wait).  *n receives their number even when it exceeds cap.  Tests use it to assert
 * that the prefilter is sound: every read with a (segment, k) that k_mer_check accepts (kmer.cpp:2221-2258) must be there. */
int trew_hip_debug_worklist(trew_hip_ctx *ctx, int slot, uint32_t *units, uint64_t cap, uint64_t *n);

/* Per-read results of the last submit on `slot` (after trew_hip_wait): for
 * TREW_MODE_SEGMENT the (k_high, k_low, MAX_SEQ at k_high, MAX_SEQ at k_low)
 * that k_mer_check returns / reports through repeat_seq (kmer.cpp:2260-2262,
 * 2327).  Arrays of n_reads entries; any may be NULL.  seq_* hold the low 64 bits of
 * the word, seq_*_hi the high 64 bits (non-zero only for k > 32). */
int trew_hip_segment_results(trew_hip_ctx *ctx, int slot, int32_t *k_high, int32_t *k_low,
                             uint64_t *seq_high, uint64_t *seq_low, uint64_t *seq_high_hi, uint64_t *seq_low_hi,
                             uint64_t n_reads);
/* Candidate-k masks of the prefilter for the last submit on `slot`: bit (k-1)
 * of cand[r*slots_per_read + s] is set when k survived for segment s of read r.
 * Diagnostic: used to test that the prefilter never drops a passing k. */
int trew_hip_filter_masks(trew_hip_ctx *ctx, const trew_hip_batch *batch, uint64_t *cand, int slots_per_read);

/* Mean kernel timings of the submits on `slot` since the previous call (HIP events on the
 * slot's own stream, at most the last 128 submits), milliseconds; n_flagged (optional) = reads
 * the prefilter passed to the exact kernel in the last submit. */
int trew_hip_last_timing(trew_hip_ctx *ctx, int slot, float *ms_filter, float *ms_exact, uint64_t *n_flagged);

/* ---- host-side packing: the codes[] lookup of kmer.cpp:14-31 applied once per base ---- */
/* words needed for a read of n bases */
uint64_t trew_pack_words(uint64_t n_bases);
/* Packs n reads given as inclusive [st,nd] byte ranges of buf (LocationVector,
 * kmer.h:73) into words/offsets/lengths; returns the words written, or
 * (uint64_t)-1 if words_cap is too small. */
uint64_t trew_pack_reads(const char *buf, const int64_t *st, const int64_t *nd, uint64_t n_reads,
                         uint32_t *words, uint64_t words_cap, uint32_t *offsets, uint32_t *lengths);

/* Same for mate pairs (PairQueueData, kmer.h:98-103): pair i is read i of each buffer; the
 * output holds reads 2i (mate 1) and 2i+1 (mate 2), the layout TREW_MODE_PAIR expects. */
uint64_t trew_pack_pairs(const char *buf1, const int64_t *st1, const int64_t *nd1,
                         const char *buf2, const int64_t *st2, const int64_t *nd2, uint64_t n_pairs,
                         uint32_t *words, uint64_t words_cap, uint32_t *offsets, uint32_t *lengths);

/* ---- synthetic workloads of SURVEY.md section 8(d); identical on host and device ---- */
/* short reads, TTAGGG-seeded: 1.0 % telomeric, 0.5 % junction, 1 % substitutions in those,
 * N with p = 5e-4.  Host: ASCII rows of read_len bytes + '\n'. */
int trew_synth_short_ascii(uint64_t seed, uint64_t first_read, uint64_t n_reads, uint32_t read_len, char *out);
/* Device: packed triples, read r at word offset r*3*ceil(read_len/32), written to device memory. */
int trew_synth_short_device(trew_hip_ctx *ctx, uint64_t seed, uint64_t first_read, uint64_t n_reads,
                            uint32_t read_len, uint32_t *d_words);
/* paired fragments (config 3): R1 = first read_len bases of a 2*read_len fragment, R2 = revcomp of the rest.
 * Host: mate 1 and mate 2 rows; device: reads 2i, 2i+1 are the mates. */
int trew_synth_pair_ascii(uint64_t seed, uint64_t first_pair, uint64_t n_pairs, uint32_t read_len, char *out1, char *out2);
int trew_synth_pair_device(trew_hip_ctx *ctx, uint64_t seed, uint64_t first_pair, uint64_t n_pairs,
                           uint32_t read_len, uint32_t *d_words);

/* long reads (config 4): lengths from clip(lognormal(9.413, 0.7), 1000, 200000), 5 % with a 2-6 kb
 * (TTAGGG)n 3' tail (5 % substitutions), half reverse-complemented.  lengths first, then the caller
 * lays the reads out (byte offsets of the ASCII rows / u32 word offsets of the packed triples). */
int trew_synth_long_lengths(uint64_t seed, uint64_t first_read, uint64_t n_reads, uint32_t *lengths);
int trew_synth_long_ascii(uint64_t seed, uint64_t first_read, uint64_t n_reads, const uint64_t *byte_offsets, char *out);
int trew_synth_long_device(trew_hip_ctx *ctx, uint64_t seed, uint64_t first_read, uint64_t n_reads,
                           const uint32_t *d_offsets, uint32_t *d_words);

/* device memory helpers so that a non-C++ host can keep batches resident */
int trew_hip_malloc(trew_hip_ctx *ctx, uint64_t bytes, void **d_ptr);
int trew_hip_free(trew_hip_ctx *ctx, void *d_ptr);
int trew_hip_memcpy_h2d(trew_hip_ctx *ctx, void *d_dst, const void *h_src, uint64_t bytes);
int trew_hip_memcpy_d2h(trew_hip_ctx *ctx, void *h_dst, const void *d_src, uint64_t bytes);
/* pinned host memory for the double-buffered H2D copies of trew_hip_submit */
int trew_hip_host_alloc(trew_hip_ctx *ctx, uint64_t bytes, void **h_ptr);
int trew_hip_host_free(trew_hip_ctx *ctx, void *h_ptr);
int trew_hip_device_count(void);
int trew_hip_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* TREW_HIP_H */
