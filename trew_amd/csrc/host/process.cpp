// process.cpp -- per-file pipelines of the `trew` host: FASTQ decode on CPU threads, packing into
// pinned buffers, one device slot (HIP stream) per worker thread, device tables collected per file.
//
// Two pipelines, same results:
//
//  * serial reader (gzip / BGZF input): the shape of the reference
//    (kmer.cpp:987-1476) -- one producer reading 4 MiB chunks (LENGTH, kmer.h:8), sequence lines found by
//    counting newlines (num & 3 == 2), a sequence line split across two chunks carried over, pairs
//    re-synchronised by read index; NUM_THREAD-1 consumers.  A consumer does not scan the reads itself: it
//    packs the chunk into pinned memory and submits it to its own HIP stream (trew_hip_submit), so
//    decode/pack of chunk i+1 overlaps the device scan of chunk i.
//
//  * block-parallel reader (plain FASTQ; paired files have their own variant further down): the file is mapped and cut into 4 MiB
//    blocks that the NUM_THREAD-1 workers claim in file order.  A worker records the newlines of its block,
//    learns the line number its block starts at from its predecessor (a chain of additions, the only serial
//    part), and then packs the sequence lines that END in its block -- a line that starts in an earlier
//    block is simply read from there, which is what the reference's carry-over achieves -- straight into its
//    pinned buffers.  Line numbering is exact (no record-boundary heuristics): the same reads as
//    read_fastq_thread, in any block order, since counts are sums.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <atomic>
#include <cerrno>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <thread>

#include "bgzf_reader.hpp"
#include "fastq_blocks.hpp"
#include "trew_host.hpp"

namespace trew_host {

static const int LENGTH = 1 << 22;  // kmer.h:8
static const int MAX_SEQ = 1000;    // kmer.h:10
// Text a worker of the block-parallel reader gathers (over several 4 MiB blocks) before it submits one batch.  The
// device scans a batch of this size in well under a millisecond; what a submit costs is its handful of HIP calls and
// the stream's hand-overs (copy -> prefilter -> exact kernel), which at 4 MiB per batch and 15 workers were the whole
// end-to-end time (tools/e2e_cli.py, profiles/r02/README.md).
static const size_t kSubmitBytes = (size_t) 32 << 20;  // capacity; Config::batch_mib (<= 32) is what a batch is closed at

// Errors follow the reference: message on stderr, exit status 1 (kmer.cpp:84-87, 1007-1008).  The other host
// threads may be inside HIP calls on live streams at this moment; running atexit handlers and the HIP runtime's
// teardown under them can hang, so the process leaves through _exit after flushing its own output.
[[noreturn]] static void die(const char *msg) {
    fprintf(stderr, "%s\n", msg);
    fflush(stdout);
    fflush(stderr);
    _exit(EXIT_FAILURE);
}

// FileReader, kmer.h:157-204
struct FileReader {
    bool is_gz = false;
    FILE *fp = nullptr;
    gzFile gz_fp = nullptr;
    BgzfReader *bgzf = nullptr;  // block-gzip input: members inflated on several threads
    int read(char *buffer, int length) {
        if (bgzf) return bgzf->read(buffer, length);
        return is_gz ? gzread(gz_fp, buffer, (unsigned) length) : (int) fread(buffer, 1, (size_t) length, fp);
    }
    bool eof() {
        if (bgzf) return bgzf->eof();
        return is_gz ? gzeof(gz_fp) != 0 : feof(fp) != 0;
    }
    const char *error() {
        if (bgzf) return bgzf->error();
        if (is_gz) {
            int err_num;
            return gzerror(gz_fp, &err_num);
        }
        return strerror(errno);
    }
    void close() {
        if (bgzf)
            delete bgzf;
        else if (is_gz)
            gzclose(gz_fp);
        else
            fclose(fp);
        bgzf = nullptr;
    }
};

[[noreturn]] static void open_failed(const char *file_name) {  // kmer.cpp:1288-1289
    fprintf(stderr, "File open failed: %s\n", file_name);
    fflush(stdout);
    fflush(stderr);
    _exit(EXIT_FAILURE);
}

static FileReader open_reader(const char *file_name, bool is_gz, int n_threads) {
    FileReader r;
    r.is_gz = is_gz;
    if (is_gz && BgzfReader::sniff(file_name)) {
        // inflate is the bottleneck of compressed input and the packers are nearly idle beside it: as many inflate
        // threads as the user gave the run (-t), within what the machine has
        const unsigned hw = std::max(2u, std::thread::hardware_concurrency());
        r.bgzf = new BgzfReader(file_name, (int) std::min<unsigned>(hw, (unsigned) std::max(2, std::min(n_threads, 32))));
        if (!r.bgzf->ok()) open_failed(file_name);
        return r;
    }
    if (is_gz) {
        r.gz_fp = gzopen(file_name, "r");
        if (!r.gz_fp) open_failed(file_name);
        gzbuffer(r.gz_fp, 1 << 20);
    } else {
        r.fp = fopen(file_name, "r");
        if (!r.fp) open_failed(file_name);
    }
    return r;
}

// QueueData / PairQueueData, kmer.h:93-103
struct Chunk {
    char *buffer1 = nullptr, *buffer2 = nullptr;
    std::vector<int64_t> st1, nd1, st2, nd2;
    bool sentinel = false;
    // single-file chunks leave the reader with their lines only COUNTED (the reader is the serial part of a compressed
    // run: it inflates, counts newlines and carries a split sequence line over); the consumer that gets the chunk finds
    // the sequence lines itself from `total` bytes and the number of newlines that precede the chunk in the file
    bool located = true;
    int total = 0;
    int64_t num_before = 0;
};

class ChunkQueue {  // the role of tbb::concurrent_bounded_queue (kmer.h:111-112)
public:
    explicit ChunkQueue(size_t cap) : cap_(cap) {}
    void push(Chunk *c) {
        std::unique_lock<std::mutex> lk(m_);
        not_full_.wait(lk, [&] { return q_.size() < cap_; });
        q_.push_back(c);
        not_empty_.notify_one();
    }
    Chunk *pop() {
        std::unique_lock<std::mutex> lk(m_);
        not_empty_.wait(lk, [&] { return !q_.empty(); });
        Chunk *c = q_.front();
        q_.pop_front();
        not_full_.notify_one();
        return c;
    }

private:
    std::mutex m_;
    std::condition_variable not_full_, not_empty_;
    std::deque<Chunk *> q_;
    size_t cap_;
};

struct Worker {
    int dev_index = 0;  // index into Scanner::dev
    int slot = 0;
    // the block-parallel reader alternates between two slots (and two pinned buffers): it fills one batch while the previous
    // one is still on its way to the device, instead of waiting for the copy (13 % of a worker's time with one, --stats)
    int slot_b = 0;
    uint32_t *h_buf_b = nullptr;
    // one pinned allocation; a batch is laid out [offsets n][lengths n][words] (serial reader: n is known before
    // packing) or [words][offsets n][lengths n] (block reader: reads accumulate), either of which trew_hip_submit
    // ships with a single copy (include/trew_hip.h, trew_hip_batch)
    uint32_t *h_buf = nullptr;
    std::vector<uint32_t> tmp_off, tmp_len;  // block reader: offsets / lengths until the batch is closed
    uint64_t words_cap = 0, reads_cap = 0, text_cap = 0;
    std::vector<uint32_t> tmp_woff;  // text batches: word offset of every read (running sum of the packed sizes)
    uint64_t reads = 0, bases = 0, submits = 0;
    double t_scan = 0, t_wait = 0, t_pack = 0, t_submit = 0, t_pressure = 0;  // seconds per phase, for --stats (t_pressure is part of t_submit)
    std::vector<int64_t> st, nd;      // block-parallel reader: sequence lines of the current block
    std::vector<uint32_t> nl;         // newline offsets of the current block
};

// one GPU: its context and the lock that keeps a table drain apart from running submits
struct Device {
    trew_hip_ctx *ctx = nullptr;
    std::shared_mutex drain_mu;  // shared: trew_hip_submit; exclusive: collect + reset
};

struct Scanner {
    Config cfg;
    int mode = TREW_MODE_SHORT;
    std::vector<std::unique_ptr<Device>> dev;
    std::vector<Worker> workers;
    RunStats stats;
    // rows drained from the device tables in the middle of a file (the reference's hash maps grow without
    // bound; the device table is fixed-size, so it is emptied into host memory whenever it runs half full)
    std::mutex pending_mu;
    ResultMapData pending;
    std::atomic<uint64_t> drains{0};
    double t_map = 0, t_workers = 0, t_unmap = 0, t_collect = 0;  // --stats: wall seconds of the block-parallel reader's stages
};

[[noreturn]] static void hip_die(trew_hip_ctx *ctx, const char *what) {
    fprintf(stderr, "%s: %s\n", what, trew_hip_last_error(ctx));
    fflush(stdout);
    fflush(stderr);
    _exit(EXIT_FAILURE);
}

Scanner *scanner_create(const Config &cfg, int mode) {
    Scanner *s = new Scanner();
    s->cfg = cfg;
    s->mode = mode;
    // --compat_g1: the reference's 64-bit pair branch with its un-cleared temp_result_left is only defined for ONE
    // consumer that sees the pairs in file order: one worker, one slot.  With MAX_MER > 32 the reference runs the 128-bit
    // branch, which clears the map: nothing to reproduce, the flag is not passed on.
    const bool compat_g1 = cfg.compat_g1 && mode == TREW_MODE_PAIR && cfg.MAX_MER <= 32;
    const int n_workers = compat_g1 ? 1 : std::max(1, cfg.NUM_THREAD - 1);  // the caller's thread is the reader (kmer.cpp:1278-1301)
    const int ndev = (int) cfg.devices.size();
    const bool pair = mode == TREW_MODE_PAIR;
    const uint64_t reads_cap = (pair ? 2ull : 1ull) << 20;
    // serial reader: one 4 MiB chunk per mate; block reader: kSubmitBytes of text per batch plus the line that ends in a
    // block but started before it; every read may waste up to one triple of padding
    const uint64_t words_cap = 3ull * ((uint64_t) (kSubmitBytes + 2 * LENGTH) / 32 + reads_cap) + 64;
    // text batches of the block-parallel reader (the device packs): the sequence bytes of a batch + 12 B of index arrays per read
    const uint64_t text_cap = (uint64_t) (kSubmitBytes + 2 * LENGTH) + 12ull * reads_cap;
    std::vector<int> slots_on_dev((size_t) ndev, 0);
    for (int w = 0; w < n_workers; w++) slots_on_dev[(size_t) (w % ndev)]++;
    for (int d = 0; d < ndev; d++) {
        trew_hip_params p;
        memset(&p, 0, sizeof(p));
        p.min_mer = cfg.MIN_MER;
        p.max_mer = cfg.MAX_MER;
        p.low_baseline = cfg.LOW_BASELINE;
        p.high_baseline = cfg.HIGH_BASELINE;
        p.slice_length = cfg.SLICE_LENGTH;
        p.mode = mode;
        p.device = cfg.devices[(size_t) d];
        p.n_slots = compat_g1 ? 1 : 2 * std::max(1, slots_on_dev[(size_t) d]);
        p.max_batch_words = words_cap;
        p.max_batch_reads = reads_cap;
        p.table_log2_slots = (uint32_t) cfg.table_log2_slots;
        p.max_batch_ascii_bytes = cfg.host_pack ? 0 : text_cap;
        // ~10^4 small batches a second: every HIP call per batch counts -- no timing events, and the table's fill state comes
        // back with every batch instead of being asked for before every batch (0.4 ms a query with 15 threads asking)
        p.flags = TREW_FLAG_NO_TIMING | TREW_FLAG_TRACK_PRESSURE | (compat_g1 ? TREW_FLAG_COMPAT_G1 : 0);
        trew_hip_ctx *c = nullptr;
        if (trew_hip_init(&p, &c) != 0) die(trew_hip_last_error(nullptr));
        s->dev.emplace_back(new Device());
        s->dev.back()->ctx = c;
    }
    std::vector<int> next_slot((size_t) ndev, 0);
    for (int w = 0; w < n_workers; w++) {
        Worker wk;
        wk.dev_index = w % ndev;
        wk.slot = next_slot[(size_t) wk.dev_index]++;
        wk.slot_b = compat_g1 ? wk.slot : next_slot[(size_t) wk.dev_index]++;
        wk.words_cap = words_cap;
        wk.reads_cap = reads_cap;
        trew_hip_ctx *c = s->dev[(size_t) wk.dev_index]->ctx;
        wk.text_cap = text_cap;
        if (trew_hip_host_alloc(c, std::max<uint64_t>((words_cap + 2 * reads_cap) * 4, text_cap + 64), (void **) &wk.h_buf)) hip_die(c, "pinned allocation");
        if (trew_hip_host_alloc(c, std::max<uint64_t>((words_cap + 2 * reads_cap) * 4, text_cap + 64), (void **) &wk.h_buf_b)) hip_die(c, "pinned allocation");
        s->workers.push_back(std::move(wk));
    }
    return s;
}

void scanner_destroy(Scanner *s) {
    if (!s) return;
    for (auto &w : s->workers) {
        trew_hip_ctx *c = s->dev[(size_t) w.dev_index]->ctx;
        trew_hip_host_free(c, w.h_buf);
        trew_hip_host_free(c, w.h_buf_b);
    }
    for (auto &d : s->dev) trew_hip_destroy(d->ctx);
    delete s;
}

const RunStats &last_stats(const Scanner *s) { return s->stats; }

static void add_rows_to(ResultMapData &r, const std::vector<trew_hip_row> &rows, uint64_t n) {
    for (uint64_t i = 0; i < n; i++) {
        const uint128_t w = ((uint128_t) rows[i].word_hi << 64) | rows[i].word_lo;
        r.table[rows[i].table][KmerSeq{rows[i].k, w}] += rows[i].count;  // thread merge, kmer.cpp:1486-1515
    }
}

// empties one device's tables into host memory; the caller holds the device's drain lock exclusively
// (or no worker is running)
static void drain_device(Scanner *s, Device *d) {
    uint64_t n = 0;
    if (trew_hip_collect(d->ctx, -1, nullptr, 0, &n)) hip_die(d->ctx, "trew_hip_collect");
    std::vector<trew_hip_row> rows((size_t) std::max<uint64_t>(n, 1));
    if (trew_hip_collect(d->ctx, -1, rows.data(), n, &n)) hip_die(d->ctx, "trew_hip_collect");
    {
        std::lock_guard<std::mutex> lk(s->pending_mu);
        add_rows_to(s->pending, rows, n);
    }
    if (trew_hip_reset_tables(d->ctx)) hip_die(d->ctx, "trew_hip_reset_tables");
}

// true when the device table should be emptied before more rows arrive
static bool under_pressure(trew_hip_ctx *c) {
    uint64_t used = 0, total = 0, spilled = 0, spill_cap = 0;
    if (trew_hip_table_pressure(c, &used, &total, &spilled, &spill_cap)) hip_die(c, "trew_hip_table_pressure");
    return used * 2 > total || spilled > 0;
}

// hand one packed batch (already in the worker's pinned buffers) to the worker's device slot
static void submit_packed(Scanner *s, Worker *w, uint32_t *h_words, uint32_t *h_offsets, uint32_t *h_lengths, uint64_t n_reads, uint64_t n_words) {
    Device *d = s->dev[(size_t) w->dev_index].get();
    trew_hip_ctx *c = d->ctx;
    if (n_words == (uint64_t) -1) die("internal error: packed chunk exceeds the slot buffer");
    uint32_t max_len = 0;
    uint64_t bases = 0;
    for (uint64_t i = 0; i < n_reads; i++) {
        bases += h_lengths[i];
        max_len = std::max(max_len, h_lengths[i]);
    }
    w->bases += bases;
    w->reads += n_reads;
    if (n_reads == 0) return;
    // before every batch: has the fixed-size device table run half full (or started to spill)?  Then it is emptied
    // into host memory before this batch adds to it.  Counts can only be lost if one batch fills what is left of the
    // table AND the spill log (>= 64 k rows, 1 M at the default table size); trew_hip_collect reports that as an error.
    w->submits++;
    const auto tp0 = std::chrono::steady_clock::now();
    const bool pressed = under_pressure(c);
    w->t_pressure += std::chrono::duration<double>(std::chrono::steady_clock::now() - tp0).count();
    if (pressed) {
        // emptying the table mid-file also empties the stale map of --compat_g1 (trew_hip_reset_tables starts a new input): say so
        if (s->cfg.compat_g1 && s->mode == TREW_MODE_PAIR && s->cfg.MAX_MER <= 32)
            die("--compat_g1: the device count table filled up in the middle of a file; raise --table_log2_slots");
        std::unique_lock<std::shared_mutex> lk(d->drain_mu);  // waits for running submits, blocks new ones
        if (under_pressure(c)) {                              // nobody drained in the meantime
            drain_device(s, d);
            s->drains++;
        }
    }
    trew_hip_batch b;
    memset(&b, 0, sizeof(b));
    b.words = h_words;
    b.n_words = n_words;
    b.offsets = h_offsets;
    b.lengths = h_lengths;
    b.n_reads = n_reads;
    b.max_length = (int32_t) max_len;
    std::shared_lock<std::shared_mutex> lk(d->drain_mu);
    if (trew_hip_submit(c, &b, w->slot)) hip_die(c, "trew_hip_submit");
}

// the same for a batch of TEXT (sequence bytes + index arrays, see trew_hip_ascii_batch): the device packs
static void submit_text(Scanner *s, Worker *w, const trew_hip_ascii_batch &b, uint64_t bases) {
    Device *d = s->dev[(size_t) w->dev_index].get();
    trew_hip_ctx *c = d->ctx;
    w->bases += bases;
    w->reads += b.n_reads;
    if (b.n_reads == 0) return;
    w->submits++;
    const auto tp0 = std::chrono::steady_clock::now();
    const bool pressed = under_pressure(c);
    w->t_pressure += std::chrono::duration<double>(std::chrono::steady_clock::now() - tp0).count();
    if (pressed) {
        // emptying the table mid-file also empties the stale map of --compat_g1 (trew_hip_reset_tables starts a new input): say so
        if (s->cfg.compat_g1 && s->mode == TREW_MODE_PAIR && s->cfg.MAX_MER <= 32)
            die("--compat_g1: the device count table filled up in the middle of a file; raise --table_log2_slots");
        std::unique_lock<std::shared_mutex> lk(d->drain_mu);  // waits for running submits, blocks new ones
        if (under_pressure(c)) {                              // nobody drained in the meantime
            drain_device(s, d);
            s->drains++;
        }
    }
    std::shared_lock<std::shared_mutex> lk(d->drain_mu);
    if (trew_hip_submit_ascii(c, &b, w->slot)) hip_die(c, "trew_hip_submit_ascii");
}

// the consumer: buffer_task* (kmer.cpp:80-985) with the scan itself moved to the device
static void worker_loop(Scanner *s, Worker *w, ChunkQueue *q) {
    trew_hip_ctx *c = s->dev[(size_t) w->dev_index]->ctx;
    for (;;) {
        Chunk *ch = q->pop();
        if (ch->sentinel) {  // loc_vector == nullptr, kmer.cpp:108-110
            delete ch;
            break;
        }
        if (!ch->located) {
            // the newline that makes num & 3 == 2 closes a sequence line (read_fastq_thread, kmer.cpp:1002-1011); a chunk
            // starts at a line start or, after a carry-over, at the start of a sequence line
            if (w->nl.size() < (size_t) LENGTH + 2) w->nl.resize((size_t) LENGTH + 2);  // + 2: scan_newlines stores two slots ahead
            const size_t cnt = scan_newlines(ch->buffer1, (size_t) ch->total, w->nl.data());
            const bool long_mode = s->mode == TREW_MODE_LONG;
            for (size_t j = (size_t) ((1 - ch->num_before) & 3); j < cnt; j += 4) {
                const int64_t start = j > 0 ? (int64_t) w->nl[j - 1] + 1 : 0, len = (int64_t) w->nl[j] - start;
                if (long_mode) {
                    if (len < s->cfg.SLICE_LENGTH) continue;  // kmer.cpp:1184
                } else if (len > MAX_SEQ) {
                    die("This mode is designed for short-read sequencing. Please use 'trew long'.");  // kmer.cpp:1006-1009
                }
                ch->st1.push_back(start);
                ch->nd1.push_back(start + len - 1);
            }
        }
        if (trew_hip_wait(c, w->slot)) hip_die(c, "trew_hip_wait");  // the slot's pinned buffers are free again
        uint64_t nw, n_reads;
        if (s->mode == TREW_MODE_PAIR) {
            const uint64_t np = std::min(ch->st1.size(), ch->st2.size());  // kmer.cpp:321
            n_reads = 2 * np;
            nw = trew_pack_pairs(ch->buffer1, ch->st1.data(), ch->nd1.data(), ch->buffer2, ch->st2.data(), ch->nd2.data(), np, w->h_buf + 2 * n_reads,
                                 w->words_cap, w->h_buf, w->h_buf + n_reads);
        } else {
            n_reads = ch->st1.size();
            nw = trew_pack_reads(ch->buffer1, ch->st1.data(), ch->nd1.data(), n_reads, w->h_buf + 2 * n_reads, w->words_cap, w->h_buf, w->h_buf + n_reads);
        }
        submit_packed(s, w, w->h_buf + 2 * n_reads, w->h_buf, w->h_buf + n_reads, n_reads, nw);
        free(ch->buffer1);  // the consumer owns and frees the chunk, kmer.cpp:175-176
        free(ch->buffer2);
        delete ch;
    }
    if (trew_hip_wait(c, w->slot)) hip_die(c, "trew_hip_wait");
}

static char *alloc_buffer() {
    char *b = (char *) malloc(LENGTH);
    if (!b) die("memory allocation failure");
    return b;
}

// read_fastq_thread (kmer.cpp:987-1038) and read_fastq_long_thread (1166-1213): 4 MiB chunks, sequence lines by counting
// newlines, a sequence line split across two chunks carried over.  This thread is the serial part of a compressed run, so it
// only COUNTS the newlines of a chunk (AVX2) and leaves finding the lines to the consumer (Chunk::located).
static void read_fastq_thread(FileReader &fr, ChunkQueue *q) {
    int64_t num = 0;
    int shift = 0;
    char *buffer = alloc_buffer();
    for (;;) {
        Chunk *ch = new Chunk();
        const int bytes_read = fr.read(buffer + shift, LENGTH - 1 - shift);
        const int total = (bytes_read > 0 ? bytes_read : 0) + shift;
        buffer[total] = '\0';
        ch->buffer1 = buffer;
        ch->located = false;
        ch->total = total;
        ch->num_before = num;
        num += (int64_t) count_newlines(buffer, (size_t) total);
        if (bytes_read <= 0) {
            q->push(ch);
            if (fr.eof()) break;
            fprintf(stderr, "File-IO Error: %s.\n", fr.error());  // kmer.cpp:1021-1022
            fflush(stdout);
            fflush(stderr);
            _exit(EXIT_FAILURE);
        }
        char *buffer_new = alloc_buffer();
        shift = 0;
        if ((num & 3) == 1) {  // inside a sequence line: carry it over (kmer.cpp:1026-1029)
            const char *last = (const char *) memrchr(buffer, '\n', (size_t) total);
            const int idx = last ? (int) (last - buffer) : -1;
            const int rest = total - idx - 1;
            memcpy(buffer_new, buffer + idx + 1, (size_t) rest);
            shift = rest;
            if (shift >= LENGTH - 2) die("a read does not fit one 4 MiB chunk");
        }
        q->push(ch);
        buffer = buffer_new;
    }
}

// read_pair_fastq_thread, kmer.cpp:1040-1164: mates are matched by read index; the side that
// decoded more reads in a chunk carries the surplus over to the next one
static void read_pair_fastq_thread(FileReader &f1, FileReader &f2, ChunkQueue *q) {
    struct Side {
        FileReader *fr;
        int num = 0, shift = 0, idx = -1, bytes_read = 0;
        bool is_end = false;
        char *buffer = nullptr;
        std::vector<int64_t> st, nd;
    } s[2];
    s[0].fr = &f1;
    s[1].fr = &f2;
    s[0].buffer = alloc_buffer();
    s[1].buffer = alloc_buffer();
    for (;;) {
        for (int m = 0; m < 2; m++) {
            Side &x = s[m];
            x.st.clear();
            x.nd.clear();
            if (!x.is_end) {
                x.bytes_read = x.fr->read(x.buffer + x.shift, LENGTH - 1 - x.shift);
                if (x.bytes_read <= 0) {
                    if (x.fr->eof()) {
                        x.is_end = true;
                        x.bytes_read = 0;
                    } else {
                        fprintf(stderr, "File %d IO Error: %s.\n", m + 1, x.fr->error());  // kmer.cpp:1065,1080
                        fflush(stdout);
                        fflush(stderr);
                        _exit(EXIT_FAILURE);
                    }
                }
            } else {
                x.bytes_read = 0;
            }
            const int total = x.bytes_read + x.shift;
            x.buffer[total] = '\0';
            for (const char *nl = (const char *) memchr(x.buffer, '\n', (size_t) total); nl;
                 nl = (const char *) memchr(nl + 1, '\n', (size_t) (x.buffer + total - nl - 1))) {
                const int i = (int) (nl - x.buffer);
                x.num += 1;
                if ((x.num & 3) == 2) {
                    if ((i - 1) - (x.idx + 1) + 1 > MAX_SEQ)  // the reference leaves pair mode unchecked
                        die("This mode is designed for short-read sequencing. Please use 'trew long'.");
                    x.st.push_back(x.idx + 1);
                    x.nd.push_back(i - 1);
                }
                x.idx = i;
            }
        }
        Chunk *ch = new Chunk();
        if (s[0].is_end && s[1].is_end) {
            if (s[0].num != s[1].num) {  // kmer.cpp:1112-1114
                fprintf(stderr, "Error: Mismatched record counts between files (num1: %d, num2: %d).\n", s[0].num, s[1].num);
                fflush(stdout);
                fflush(stderr);
                _exit(EXIT_FAILURE);
            }
            ch->buffer1 = s[0].buffer;
            ch->buffer2 = s[1].buffer;
            ch->st1 = s[0].st;
            ch->nd1 = s[0].nd;
            ch->st2 = s[1].st;
            ch->nd2 = s[1].nd;
            q->push(ch);
            break;
        }
        if ((s[0].st.empty() && !s[1].st.empty()) || (!s[0].st.empty() && s[1].st.empty())) die("Paired-end error");  // kmer.cpp:1120-1124
        const size_t min_size = std::min(s[0].st.size(), s[1].st.size());
        char *nb[2] = {alloc_buffer(), alloc_buffer()};
        for (int m = 0; m < 2; m++) {
            Side &x = s[m];
            const int total = x.bytes_read + x.shift;
            if (x.st.size() > min_size) {  // surplus reads: restart the next chunk at the first unmatched sequence line
                x.idx = (int) x.st[min_size] - 1;
                const int rest = total - x.idx - 1;
                memcpy(nb[m], x.buffer + x.idx + 1, (size_t) rest);
                x.num = ((x.num - 2) / 4) * 4 + 1 - 4 * (int) (x.st.size() - min_size - 1);  // kmer.cpp:1135
                x.shift = rest;
                x.idx = -1;
            } else if ((x.num & 3) == 1) {
                const int rest = total - x.idx - 1;
                memcpy(nb[m], x.buffer + x.idx + 1, (size_t) rest);
                x.shift = rest;
                x.idx = -1;
            } else {
                x.shift = 0;
                x.idx = -1;
            }
        }
        ch->buffer1 = s[0].buffer;
        ch->buffer2 = s[1].buffer;
        ch->st1.assign(s[0].st.begin(), s[0].st.begin() + (long) min_size);
        ch->nd1.assign(s[0].nd.begin(), s[0].nd.begin() + (long) min_size);
        ch->st2.assign(s[1].st.begin(), s[1].st.begin() + (long) min_size);
        ch->nd2.assign(s[1].nd.begin(), s[1].nd.begin() + (long) min_size);
        q->push(ch);
        s[0].buffer = nb[0];
        s[1].buffer = nb[1];
    }
}

// ------------------------------------------------------------------ block-parallel reader (plain FASTQ)
struct BlockJob {
    BlockScan scan;
    bool long_mode = false;
    int slice_length = 0;
};

#if defined(__x86_64__)
__attribute__((target("avx2"))) static inline void copy_short_line(unsigned char *dst, const char *src, uint32_t len) {
    for (uint32_t o = 0; o < len; o += 32) _mm256_storeu_si256((__m256i *) (dst + o), _mm256_loadu_si256((const __m256i *) (src + o)));
}
#else
static inline void copy_short_line(unsigned char *dst, const char *src, uint32_t len) { memcpy(dst, src, len); }
#endif

static void block_worker_loop(Scanner *s, Worker *w, BlockJob *job) {
    trew_hip_ctx *c = s->dev[(size_t) w->dev_index]->ctx;
    typedef std::chrono::steady_clock clk;
    auto secs = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    // ship the sequence bytes, the device applies codes[] (pack kernel); else trew_pack_reads here.  Long reads are packed on the
    // host: their kernels take longer per base and a byte per base over PCIe (against 3 bits) then makes the workers wait for
    // their slots -- 29.0 against 30.7 Gbases/s on 300 k ONT-like reads (tools/e2e_long.py, profiles/r03/README.md)
    const bool text = !s->cfg.host_pack && !job->long_mode;
#if defined(__x86_64__)
    const bool inline_copy = __builtin_cpu_supports("avx2") && !getenv("TREW_AB_MEMCPY");
#else
    const bool inline_copy = false;
#endif
    uint64_t acc_reads = 0, acc_words = 0, acc_bytes = 0, acc_bases = 0;
    size_t acc_text = 0;
    bool buffers_free = false;  // has the slot's previous submit been waited for
    bool same_len = true;       // text batches: every read so far has the length of the first one
    if (w->tmp_off.size() < w->reads_cap) {
        w->tmp_off.resize(w->reads_cap);
        w->tmp_len.resize(w->reads_cap);
        w->tmp_woff.resize(w->reads_cap);
    }
    // text batches: the bases start at a fixed place of the pinned buffer, the three index arrays are put right in front of
    // them when the batch is closed -- [word_offsets][byte_offsets][lengths][bases], which trew_hip_submit_ascii ships in one copy
    unsigned char *h_bases = (unsigned char *) w->h_buf + 12ull * w->reads_cap;
    const uint64_t bytes_cap = w->text_cap - 12ull * w->reads_cap;
    auto close_batch = [&]() {
        if (acc_reads == 0) return;
        const clk::time_point t3 = clk::now();
        if (text) {
            trew_hip_ascii_batch b;
            memset(&b, 0, sizeof(b));
            b.bases = (const char *) h_bases;
            b.n_bytes = acc_bytes;
            b.n_reads = acc_reads;
            // an ordinary Illumina run: no index arrays, and the prefilter takes its uniform-geometry path.  (A batch of EMPTY
            // sequence lines has one length too, but a uniform batch needs a length: it goes with index arrays and yields nothing,
            // as in the reference.)
            if (same_len && w->tmp_len[0] > 0 && !job->long_mode) {
                b.uniform_length = w->tmp_len[0];
            } else {
                uint32_t *arr = (uint32_t *) (h_bases - 12ull * acc_reads);
                memcpy(arr, w->tmp_woff.data(), acc_reads * 4);
                memcpy(arr + acc_reads, w->tmp_off.data(), acc_reads * 4);
                memcpy(arr + 2 * acc_reads, w->tmp_len.data(), acc_reads * 4);
                b.word_offsets = arr;
                b.byte_offsets = arr + acc_reads;
                b.lengths = arr + 2 * acc_reads;
            }
            submit_text(s, w, b, acc_bases);
        } else {
            uint32_t *h_off = w->h_buf + acc_words, *h_len = h_off + acc_reads;  // [words][offsets][lengths]
            memcpy(h_off, w->tmp_off.data(), acc_reads * 4);
            memcpy(h_len, w->tmp_len.data(), acc_reads * 4);
            submit_packed(s, w, w->h_buf, h_off, h_len, acc_reads, acc_words);
        }
        acc_reads = acc_words = acc_bytes = acc_bases = 0;
        acc_text = 0;
        same_len = true;
        // the next batch goes to the other slot and the other pinned buffer (that slot's previous batch was submitted one
        // batch ago: the wait below rarely blocks)
        std::swap(w->slot, w->slot_b);
        std::swap(w->h_buf, w->h_buf_b);
        h_bases = (unsigned char *) w->h_buf + 12ull * w->reads_cap;
        buffers_free = false;
        w->t_submit += secs(t3, clk::now());
    };
    for (;;) {
        const clk::time_point t0 = clk::now();
        size_t blk = 0;
        if (!job->scan.claim(w->nl, w->st, w->nd, &blk)) break;
        // the limits of the chunk reader, applied to the same lines
        size_t keep = 0;
        uint64_t need_words = 0, need_bytes = 0;
        for (size_t i = 0; i < w->st.size(); i++) {
            const int64_t len = w->nd[i] - w->st[i] + 1;
            if (job->long_mode) {
                if (len >= LENGTH - 2) die("a read does not fit one 4 MiB chunk");
                if (len < job->slice_length) continue;                              // kmer.cpp:1184
            } else if (len > MAX_SEQ) {
                die("This mode is designed for short-read sequencing. Please use 'trew long'.");  // kmer.cpp:1006-1009
            }
            w->st[keep] = w->st[i];
            w->nd[keep] = w->nd[i];
            need_words += 3ull * (((uint64_t) len + 31) / 32);
            need_bytes += (uint64_t) (len > 0 ? len : 0);
            keep++;
        }
        const clk::time_point t1 = clk::now();
        w->t_scan += secs(t0, t1);
        // a block holds at most reads_cap sequence lines and words_cap/2 words: it always fits an empty batch
        if (acc_reads + keep > w->reads_cap || acc_words + need_words > w->words_cap || (text && acc_bytes + need_bytes > bytes_cap)) close_batch();
        if (keep > w->reads_cap || need_words > w->words_cap || (text && need_bytes > bytes_cap)) die("internal error: one block's reads do not fit the slot buffer");
        if (!buffers_free) {
            const clk::time_point tw = clk::now();
            if (trew_hip_wait(c, w->slot)) hip_die(c, "trew_hip_wait");  // the slot's pinned buffer is free again
            buffers_free = true;
            w->t_wait += secs(tw, clk::now());
        }
        const clk::time_point t2 = clk::now();
        if (text) {
            // locate + copy: the bytes of the sequence lines back to back; packing is the device's job
            const char *base = job->scan.base;
            uint64_t words = acc_words, bytes = acc_bytes;
            const uint32_t first_len = acc_reads ? w->tmp_len[0] : (keep ? (uint32_t) (w->nd[0] - w->st[0] + 1 > 0 ? w->nd[0] - w->st[0] + 1 : 0) : 0u);
            // short lines are copied 32 bytes at a time past their end (the next line overwrites the excess, the pinned buffer
            // has 64 bytes of slack, and the source stays inside the mapping): a libc memcpy call per 150-byte line costs as
            // much as finding the line (profiles/r03/README.md, end to end)
            const int64_t safe_src = (int64_t) job->scan.size - 288;
            for (size_t i = 0; i < keep; i++) {
                const int64_t n = w->nd[i] - w->st[i] + 1;
                const uint32_t len = n > 0 ? (uint32_t) n : 0u;
                if (inline_copy && len <= 256 && w->st[i] <= safe_src)
                    copy_short_line(h_bases + bytes, base + w->st[i], len);
                else
                    memcpy(h_bases + bytes, base + w->st[i], len);
                w->tmp_woff[acc_reads + i] = (uint32_t) words;
                w->tmp_off[acc_reads + i] = (uint32_t) bytes;
                w->tmp_len[acc_reads + i] = len;
                same_len = same_len && len == first_len;
                words += 3ull * (((uint64_t) len + 31) / 32);
                bytes += len;
            }
            acc_bases += bytes - acc_bytes;
            acc_words = words;
            acc_bytes = bytes;
            acc_reads += keep;
        } else {
            const uint64_t nw = trew_pack_reads(job->scan.base, w->st.data(), w->nd.data(), keep, w->h_buf + acc_words, w->words_cap - acc_words,
                                                w->tmp_off.data() + acc_reads, w->tmp_len.data() + acc_reads);
            if (nw == (uint64_t) -1) die("internal error: packed block exceeds the slot buffer");
            for (size_t i = 0; i < keep; i++) w->tmp_off[acc_reads + i] += (uint32_t) acc_words;  // offsets count from the batch's first word
            acc_reads += keep;
            acc_words += nw;
        }
        acc_text += job->scan.block;
        w->t_pack += secs(t2, clk::now());
        job->scan.release(blk);  // the block's bytes are in the pinned buffer: drop its page-table entries (see BlockScan::release)
        if (acc_text >= (size_t) s->cfg.batch_mib << 20) close_batch();
    }
    close_batch();
    if (trew_hip_wait(c, w->slot) || trew_hip_wait(c, w->slot_b)) hip_die(c, "trew_hip_wait");
}

// ------------------------------------------------------------------ block-parallel reader, paired files
// Mates are matched by read index (read r = lines 4r .. 4r+3 of each file, read_pair_fastq_thread, kmer.cpp:1040-1164).
// Pass 1 counts the newlines of every 4 MiB block of both files (all workers, any order); with the prefix sums a worker
// finds where any line starts.  Pass 2: workers claim ranges of read indices, walk the two files side by side from the
// start of that range and pack the pairs.  No carry-over, no re-synchronisation: both fall out of the line numbers.
struct PairJob {
    LineIndex index[2];
    std::atomic<size_t> next_block{0}, next_item{0};
    size_t n_pairs = 0, pairs_per_item = 1 << 16;  // run_pair_blocks picks the item size (text batches: small items, see pair_worker_loop)
};

// page-table entries of [lo, hi) of a mapped file: in with one call instead of a fault per 64 KiB (before the first pass), out
// again once a worker is done with the range (one munmap of both files at the end is single-threaded: 0.1 s for 10 GB)
static void map_range(const LineIndex &ix, size_t lo, size_t hi, bool in) {
    const size_t page = 4096;
    hi = std::min(hi, ix.size);
    if (in) {
#ifdef MADV_POPULATE_READ
        const size_t plo = lo & ~(page - 1);
        if (hi > plo) (void) madvise(const_cast<char *>(ix.base) + plo, hi - plo, MADV_POPULATE_READ);
#endif
    } else {
        const size_t plo = (lo + page - 1) & ~(page - 1), phi = hi & ~(page - 1);
        if (phi > plo) (void) madvise(const_cast<char *>(ix.base) + plo, phi - plo, MADV_DONTNEED);
    }
}

static void pair_count_loop(PairJob *job) {
    const size_t n0 = job->index[0].n_blocks, total = n0 + job->index[1].n_blocks;
    for (;;) {
        const size_t i = job->next_block.fetch_add(1);
        if (i >= total) break;
        LineIndex &ix = job->index[i < n0 ? 0 : 1];
        const size_t b = i < n0 ? i : i - n0;
        map_range(ix, b * ix.block, (b + 1) * ix.block, true);
        ix.count_block(b);
    }
}

static void pair_worker_loop(Scanner *s, Worker *w, PairJob *job) {
    trew_hip_ctx *c = s->dev[(size_t) w->dev_index]->ctx;
    typedef std::chrono::steady_clock clk;
    auto secs = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    LineCursor cur[2];
    std::vector<int64_t> st[2], nd[2];
    // Text batches (default): the sequence bytes of both mates, interleaved, accumulate over several work items until a
    // batch is full; the device packs (as block_worker_loop).  An item is small enough (pairs_per_item) for the copy to find
    // the lines that the cursor has just scanned still in cache.
    const bool text = !s->cfg.host_pack;
    unsigned char *h_bases = (unsigned char *) w->h_buf + 12ull * w->reads_cap;
    const uint64_t bytes_cap = w->text_cap - 12ull * w->reads_cap;
    uint64_t acc_reads = 0, acc_words = 0, acc_bytes = 0;
    bool same_len = true, buffers_free = false;
    uint32_t first_len = 0;
    if (text && w->tmp_off.size() < w->reads_cap) {
        w->tmp_off.resize(w->reads_cap);
        w->tmp_len.resize(w->reads_cap);
        w->tmp_woff.resize(w->reads_cap);
    }
    auto close_text_batch = [&]() {
        if (acc_reads == 0) return;
        const clk::time_point t3 = clk::now();
        trew_hip_ascii_batch b;
        memset(&b, 0, sizeof(b));
        b.bases = (const char *) h_bases;
        b.n_bytes = acc_bytes;
        b.n_reads = acc_reads;
        if (same_len && first_len > 0) {
            b.uniform_length = first_len;
        } else {
            uint32_t *arr = (uint32_t *) (h_bases - 12ull * acc_reads);
            memcpy(arr, w->tmp_woff.data(), acc_reads * 4);
            memcpy(arr + acc_reads, w->tmp_off.data(), acc_reads * 4);
            memcpy(arr + 2 * acc_reads, w->tmp_len.data(), acc_reads * 4);
            b.word_offsets = arr;
            b.byte_offsets = arr + acc_reads;
            b.lengths = arr + 2 * acc_reads;
        }
        submit_text(s, w, b, acc_bytes);
        acc_reads = acc_words = acc_bytes = 0;
        same_len = true;
        std::swap(w->slot, w->slot_b);  // two slots, two pinned buffers: fill the next batch while this one travels
        std::swap(w->h_buf, w->h_buf_b);
        h_bases = (unsigned char *) w->h_buf + 12ull * w->reads_cap;
        buffers_free = false;
        w->t_submit += secs(t3, clk::now());
    };
    for (;;) {
        const size_t item = job->next_item.fetch_add(1);
        const size_t r0 = item * job->pairs_per_item;
        if (r0 >= job->n_pairs) break;
        const size_t r1 = std::min(job->n_pairs, r0 + job->pairs_per_item);
        const clk::time_point tc = clk::now();
        for (int m = 0; m < 2; m++) cur[m].init(job->index[m].base, job->index[m].size, job->index[m].line_start((int64_t) (4 * r0), w->nl));
        w->t_scan += secs(tc, clk::now());
        int64_t first_byte[2] = {-1, -1}, last_byte[2] = {-1, -1};
        // text batches: the item is walked in pieces small enough for the copy to find the lines just scanned in cache (the
        // cursors carry on from piece to piece; locating a line from the block counts costs a scan of its 4 MiB block)
        const size_t piece = text ? (size_t) 8192 : job->pairs_per_item;
        for (size_t q0 = r0; q0 < r1; q0 += piece) {
        const size_t q1 = std::min(r1, q0 + piece);
        const clk::time_point t0 = clk::now();
        uint64_t need_words = 0, need_bytes = 0;
        for (int m = 0; m < 2; m++) {
            st[m].clear();
            nd[m].clear();
            for (size_t r = q0; r < q1; r++) {
                const int64_t a = cur[m].next(), b = cur[m].next(), x = cur[m].next(), y = cur[m].next();
                if (a < 0 || b < 0) die("internal error: the paired reader ran past the end of a file");
                (void) x;
                (void) y;  // the last record may lack its final newline(s): the sequence line is complete
                if (b - a - 1 > MAX_SEQ)  // the reference leaves pair mode unchecked
                    die("This mode is designed for short-read sequencing. Please use 'trew long'.");
                st[m].push_back(a + 1);
                nd[m].push_back(b - 1);
                const uint64_t len = (uint64_t) std::max<int64_t>(0, b - a - 1);
                need_words += 3ull * ((len + 31) / 32);
                need_bytes += len;
            }
            if (!st[m].empty()) {
                if (first_byte[m] < 0) first_byte[m] = st[m].front();
                last_byte[m] = nd[m].back();
            }
        }
        const clk::time_point t1 = clk::now();
        w->t_scan += secs(t0, t1);
        if (text) {
            const uint64_t item_reads = 2 * (uint64_t) st[0].size();
            if (acc_reads + item_reads > w->reads_cap || acc_words + need_words > w->words_cap || acc_bytes + need_bytes > bytes_cap) close_text_batch();
            if (item_reads > w->reads_cap || need_words > w->words_cap || need_bytes > bytes_cap) die("internal error: one item's pairs do not fit the slot buffer");
            if (!buffers_free) {
                const clk::time_point tw = clk::now();
                if (trew_hip_wait(c, w->slot)) hip_die(c, "trew_hip_wait");  // the slot's pinned buffer is free again
                buffers_free = true;
                w->t_wait += secs(tw, clk::now());
            }
            const clk::time_point t2 = clk::now();
            if (acc_reads == 0 && !st[0].empty()) first_len = (uint32_t) std::max<int64_t>(0, nd[0][0] - st[0][0] + 1);
            uint64_t wo = acc_words, bo = acc_bytes, rr = acc_reads;
            const int64_t safe_src[2] = {(int64_t) job->index[0].size - 288, (int64_t) job->index[1].size - 288};
            for (size_t i = 0; i < st[0].size(); i++) {
                for (int m = 0; m < 2; m++) {
                    const int64_t a = st[m][i], len64 = nd[m][i] - a + 1;
                    const uint32_t len = len64 > 0 ? (uint32_t) len64 : 0u;
                    if (len <= 256 && a <= safe_src[m])
                        copy_short_line(h_bases + bo, job->index[m].base + a, len);
                    else
                        memcpy(h_bases + bo, job->index[m].base + a, len);
                    w->tmp_woff[rr] = (uint32_t) wo;
                    w->tmp_off[rr] = (uint32_t) bo;
                    w->tmp_len[rr] = len;
                    rr++;
                    same_len = same_len && len == first_len;
                    wo += 3ull * (((uint64_t) len + 31) / 32);
                    bo += len;
                }
            }
            acc_words = wo;
            acc_bytes = bo;
            acc_reads = rr;
            w->t_pack += secs(t2, clk::now());
            if (acc_bytes >= ((uint64_t) s->cfg.batch_mib << 19)) close_text_batch();  // batch_mib of FASTQ text is about half as many sequence bytes
        } else {
            // as many pairs per batch as the slot's buffers hold
            for (size_t at = 0; at < st[0].size();) {
                size_t n = 0;
                uint64_t words = 0;
                while (at + n < st[0].size() && 2 * (n + 1) <= w->reads_cap) {
                    const uint64_t need = 3ull * (((uint64_t) (nd[0][at + n] - st[0][at + n] + 1) + 31) / 32) + 3ull * (((uint64_t) (nd[1][at + n] - st[1][at + n] + 1) + 31) / 32);
                    if (words + need > w->words_cap) break;
                    words += need;
                    n++;
                }
                if (n == 0) die("internal error: a pair does not fit the slot buffer");  // MAX_SEQ bounds a pair far below words_cap; never spin
                const clk::time_point tw = clk::now();
                if (trew_hip_wait(c, w->slot)) hip_die(c, "trew_hip_wait");  // the slot's pinned buffer is free again
                const clk::time_point t2 = clk::now();
                const uint64_t n_reads = 2 * n;
                const uint64_t nw = trew_pack_pairs(job->index[0].base, st[0].data() + at, nd[0].data() + at, job->index[1].base, st[1].data() + at, nd[1].data() + at, n,
                                                    w->h_buf + 2 * n_reads, w->words_cap, w->h_buf, w->h_buf + n_reads);
                const clk::time_point t3 = clk::now();
                submit_packed(s, w, w->h_buf + 2 * n_reads, w->h_buf, w->h_buf + n_reads, n_reads, nw);
                w->t_wait += secs(tw, t2);
                w->t_pack += secs(t2, t3);
                w->t_submit += secs(t3, clk::now());
                at += n;
                std::swap(w->slot, w->slot_b);  // two slots, two pinned buffers: pack the next batch while this one travels
                std::swap(w->h_buf, w->h_buf_b);
            }
        }
        }  // pieces
        // the item's bytes are in pinned buffers: give the page-table entries of its stretch of both files back
        for (int m = 0; m < 2; m++)
            if (first_byte[m] >= 0) map_range(job->index[m], (size_t) first_byte[m], (size_t) last_byte[m], false);
    }
    if (text) close_text_batch();
    if (trew_hip_wait(c, w->slot) || trew_hip_wait(c, w->slot_b)) hip_die(c, "trew_hip_wait");
}

struct Mapping {
    void *p = MAP_FAILED;
    size_t size = 0;
    bool map(const char *name) {
        const int fd = open(name, O_RDONLY);
        if (fd < 0) open_failed(name);
        struct stat st;
        if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size <= 0) {
            close(fd);
            return false;
        }
        size = (size_t) st.st_size;
        p = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
        close(fd);
        if (p == MAP_FAILED) return false;
        (void) madvise(p, size, MADV_SEQUENTIAL);
        (void) madvise(p, size, MADV_WILLNEED);
        return true;
    }
    ~Mapping() {
        if (p != MAP_FAILED) munmap(p, size);
    }
};

// false when a file cannot be mapped: the caller falls back to the serial reader
static bool run_pair_blocks(Scanner *s, const char *name1, const char *name2) {
    typedef std::chrono::steady_clock clk;
    auto secs = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    const clk::time_point t0 = clk::now();
    {
        Mapping m1, m2;
        if (!m1.map(name1) || !m2.map(name2)) return false;
        PairJob job;
        job.index[0].init((const char *) m1.p, m1.size, (size_t) LENGTH);
        job.index[1].init((const char *) m2.p, m2.size, (size_t) LENGTH);
        {
            std::vector<std::thread> th;
            for (size_t i = 0; i < s->workers.size(); i++) th.emplace_back(pair_count_loop, &job);
            for (auto &t : th) t.join();
        }
        job.index[0].finish();
        job.index[1].finish();
        const long long num1 = job.index[0].total(), num2 = job.index[1].total();
        if (num1 != num2) {  // kmer.cpp:1112-1114
            fprintf(stderr, "Error: Mismatched record counts between files (num1: %lld, num2: %lld).\n", num1, num2);
            fflush(stdout);
            fflush(stderr);
            _exit(EXIT_FAILURE);
        }
        job.n_pairs = (size_t) ((num1 + 2) / 4);  // sequence lines closed by a newline: line numbers 1, 5, 9, ...
        if (const char *e = getenv("TREW_PAIR_ITEM")) job.pairs_per_item = (size_t) std::max(1L, atol(e));  // experiments
        const clk::time_point t1 = clk::now();
        std::vector<std::thread> th;
        for (auto &w : s->workers) th.emplace_back(pair_worker_loop, s, &w, &job);
        for (auto &t : th) t.join();
        const clk::time_point t2 = clk::now();
        s->t_map = secs(t0, t1);  // mapping + the counting pass over both files
        s->t_workers = secs(t1, t2);
        s->t_unmap = -secs(t0, t2);  // completed below, once the mappings are gone
    }
    s->t_unmap += secs(t0, clk::now());
    return true;
}

// maps the file and runs the block workers; false when the file cannot be mapped (empty file, special file):
// the caller falls back to the serial reader
static bool run_blocks(Scanner *s, const char *name, bool long_mode, int slice_length) {
    typedef std::chrono::steady_clock clk;
    auto secs = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    const clk::time_point t0 = clk::now();
    {
        Mapping m;
        if (!m.map(name)) return false;
        BlockJob job;
        // Blocks the workers claim: the reference's chunk size (LENGTH, 4 MiB).  Smaller blocks were measured on the GPU box
        // (TREW_SCAN_BLOCK_KIB, profiles/r03/README.md) in the hope that the copy of the sequence lines would still find the
        // block in L2: 2 MiB is 10 % slower, 1 MiB 35 %, 256 KiB five times -- one madvise pair and one hand-over of the line
        // count per block cost more than the cache misses they save.
        size_t blk = (size_t) LENGTH;
        if (const char *e = getenv("TREW_SCAN_BLOCK_KIB")) {
            const long v = atol(e);
            if (v >= 64 && v <= 4096) blk = (size_t) v << 10;
        }
        job.scan.init((const char *) m.p, m.size, blk);
        job.scan.populate = true;
        job.long_mode = long_mode;
        job.slice_length = slice_length;
        const clk::time_point t1 = clk::now();
        std::vector<std::thread> th;
        for (auto &w : s->workers) th.emplace_back(block_worker_loop, s, &w, &job);
        for (auto &t : th) t.join();
        const clk::time_point t2 = clk::now();
        s->t_map = secs(t0, t1);
        s->t_workers = secs(t1, t2);
        s->t_unmap = -secs(t0, t2);  // completed below, once the mapping is gone
    }
    s->t_unmap += secs(t0, clk::now());
    return true;
}

// The same for block-gzip input (round 4): the decompressed stream is given an address range of its own (reserved, not
// committed); a block is a group of consecutive members with ~4 MiB of text, and the worker that claims it inflates its
// members into their place before it scans them (BlockScan::fill) -- inflate, newline scan, line copy and submit all run on the
// workers, nothing is serial but the chain of line counts.  false when the file is not BGZF from end to end (a plain gzip
// member, trailing bytes, a damaged member: the serial reader handles or reports those as before) or cannot be mapped.
static bool run_blocks_bgzf(Scanner *s, const char *name, bool long_mode, int slice_length) {
    typedef std::chrono::steady_clock clk;
    auto secs = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    const clk::time_point t0 = clk::now();
    {
        Mapping m;
        if (!m.map(name)) return false;
        BgzfIndex idx;
        std::string why;
        if (!idx.build((const unsigned char *) m.p, m.size, &why)) return false;
        if (idx.text_size == 0) return true;  // nothing in it
        const size_t span = (size_t) idx.text_size + 8192;
        void *text = mmap(nullptr, span, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (text == MAP_FAILED) return false;
        std::vector<size_t> bounds, first_member;
        size_t acc = 0;
        for (size_t i = 0; i < idx.members.size(); i++) {
            if (i == 0 || acc >= (size_t) LENGTH - 65536) {
                bounds.push_back((size_t) idx.members[i].toff);
                first_member.push_back(i);
                acc = 0;
            }
            acc += idx.members[i].isize;
        }
        bounds.push_back((size_t) idx.text_size);
        first_member.push_back(idx.members.size());
        BlockJob job;
        job.scan.init_bounds((const char *) text, (size_t) idx.text_size, bounds);
        job.scan.anonymous = true;
        job.scan.fill = [&](size_t b, size_t, size_t) {
            for (size_t i = first_member[b]; i < first_member[b + 1]; i++)
                if (!idx.inflate_member((const unsigned char *) m.p, i, (unsigned char *) text + idx.members[i].toff)) {
                    fprintf(stderr, "File-IO Error: BGZF member fails to inflate (data error).\n");  // as the serial reader reports it (kmer.cpp:1021-1022)
                    fflush(stdout);
                    fflush(stderr);
                    _exit(EXIT_FAILURE);
                }
        };
        job.long_mode = long_mode;
        job.slice_length = slice_length;
        const clk::time_point t1 = clk::now();
        std::vector<std::thread> th;
        for (auto &w : s->workers) th.emplace_back(block_worker_loop, s, &w, &job);
        for (auto &t : th) t.join();
        const clk::time_point t2 = clk::now();
        munmap(text, span);
        s->t_map = secs(t0, t1);
        s->t_workers = secs(t1, t2);
        s->t_unmap = -secs(t0, t2);
    }
    s->t_unmap += secs(t0, clk::now());
    return true;
}

// thread merge of process_output (kmer.cpp:1486-1515): what was drained during the file + what the devices
// still hold.  With several devices the tables are first reduced on the GPUs (trew_hip_merge, one peer copy
// per device) and only the merged rows cross PCIe.
static ResultMapData collect_tables(Scanner *s) {
    for (size_t d = 1; d < s->dev.size(); d++) {
        // Mid-file drains only keep EACH device at or below half full; the union of several devices' keys can exceed
        // device 0's partitions plus its spill log.  So device 0 is emptied into host memory whenever it is under
        // pressure before it takes another device's rows (no worker is running any more: no lock needed).
        if (under_pressure(s->dev[0]->ctx)) {
            drain_device(s, s->dev[0].get());
            s->drains++;
        }
        if (trew_hip_merge(s->dev[0]->ctx, s->dev[d]->ctx)) hip_die(s->dev[0]->ctx, "trew_hip_merge");
        if (trew_hip_reset_tables(s->dev[d]->ctx)) hip_die(s->dev[d]->ctx, "trew_hip_reset_tables");
    }
    drain_device(s, s->dev[0].get());
    ResultMapData r;
    std::lock_guard<std::mutex> lk(s->pending_mu);
    std::swap(r, s->pending);
    return r;
}

static FinalFastqOutput run_file(Scanner *s, const Config &cfg, const char *name1, const char *name2, bool gz1, bool gz2) {
    const auto t0 = std::chrono::steady_clock::now();
    for (auto &w : s->workers) {
        w.reads = 0;
        w.bases = 0;
        w.t_scan = w.t_wait = w.t_pack = w.t_submit = w.t_pressure = 0;
        w.submits = 0;
    }
    const uint64_t drains0 = s->drains;
    bool done = false;
    const char *how = "serial reader";
    if (s->mode != TREW_MODE_PAIR && !gz1 && !cfg.serial_reader) {
        done = run_blocks(s, name1, s->mode == TREW_MODE_LONG, cfg.SLICE_LENGTH);
        if (done) how = "block-parallel reader";
    } else if (s->mode != TREW_MODE_PAIR && gz1 && !cfg.serial_reader && BgzfReader::sniff(name1)) {
        done = run_blocks_bgzf(s, name1, s->mode == TREW_MODE_LONG, cfg.SLICE_LENGTH);
        if (done) how = "block-parallel BGZF reader";
    } else if (s->mode == TREW_MODE_PAIR && !gz1 && !gz2 && !cfg.serial_reader) {
        done = run_pair_blocks(s, name1, name2);
        if (done) how = "block-parallel paired reader";
    }
    if (!done) {
        size_t qcap = cfg.QUEUE_SIZE >= 4 ? (size_t) (cfg.QUEUE_SIZE / 4) : 256;  // kmer.cpp:1274-1276; "unlimited" is capped at 1 GiB
        ChunkQueue q(qcap);
        std::vector<std::thread> th;
        for (auto &w : s->workers) th.emplace_back(worker_loop, s, &w, &q);
        const int inflaters = s->mode == TREW_MODE_PAIR ? std::max(2, cfg.NUM_THREAD / 2) : cfg.NUM_THREAD;
        FileReader f1 = open_reader(name1, gz1, inflaters);
        if (s->mode == TREW_MODE_PAIR) {
            FileReader f2 = open_reader(name2, gz2, inflaters);
            read_pair_fastq_thread(f1, f2, &q);
            f2.close();
        } else {
            read_fastq_thread(f1, &q);
        }
        f1.close();
        for (size_t i = 0; i < s->workers.size(); i++) {  // sentinels, kmer.cpp:1304-1310
            Chunk *c = new Chunk();
            c->sentinel = true;
            q.push(c);
        }
        for (auto &t : th) t.join();
    }
    const auto tc0 = std::chrono::steady_clock::now();
    ResultMapData r = collect_tables(s);
    s->t_collect = std::chrono::duration<double>(std::chrono::steady_clock::now() - tc0).count();
    s->stats = RunStats();
    for (auto &w : s->workers) {
        s->stats.reads += w.reads;
        s->stats.bases += w.bases;
    }
    s->stats.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (cfg.stats)
        fprintf(stderr, "[trew] %s: %llu reads, %llu bases, %.3f s, %.3f Gbases/s end-to-end (decode + pack + scan; %s, %d worker(s), %llu table drain(s))\n",
                name1, (unsigned long long) s->stats.reads, (unsigned long long) s->stats.bases, s->stats.seconds,
                s->stats.bases / s->stats.seconds / 1e9, how, (int) s->workers.size(), (unsigned long long) (s->drains - drains0));
    if (cfg.stats && done) {
        double a = 0, b = 0, c = 0, d = 0, e = 0;
        uint64_t nsub = 0;
        for (auto &w : s->workers) a += w.t_scan, b += w.t_wait, c += w.t_pack, d += w.t_submit, e += w.t_pressure, nsub += w.submits;
        const double n = (double) s->workers.size();
        fprintf(stderr,
                "[trew]   per worker (mean seconds): newline scan + line chain %.3f, wait for the slot %.3f, pack %.3f, submit %.3f (of which table "
                "pressure query %.3f); %llu batches\n",
                a / n, b / n, c / n, d / n, e / n, (unsigned long long) nsub);
        fprintf(stderr, "[trew]   wall seconds: map the file %.3f, workers %.3f, unmap %.3f, collect the tables %.3f\n", s->t_map, s->t_workers, s->t_unmap, s->t_collect);
    }
    return process_output(name1, r, cfg.MIN_MER, stdout);  // pair mode prints file 1 only (kmer.cpp:1409)
}

FinalFastqOutput process_kmer(Scanner *s, const Config &cfg, const char *file_name, bool is_gz) {
    return run_file(s, cfg, file_name, nullptr, is_gz, false);
}
FinalFastqOutput process_kmer_pair(Scanner *s, const Config &cfg, const char *file_name1, const char *file_name2, bool is_gz1, bool is_gz2) {
    return run_file(s, cfg, file_name1, file_name2, is_gz1, is_gz2);
}
FinalFastqOutput process_kmer_long(Scanner *s, const Config &cfg, const char *file_name, bool is_gz) {
    return run_file(s, cfg, file_name, nullptr, is_gz, false);
}

}  // namespace trew_host
