// process.cpp -- per-file pipelines of the `trew` host: FASTQ/.gz chunk reader on the caller's
// thread, packer threads that each own one device slot, device tables collected per file.
//
// Shape of the reference (kmer.cpp:987-1476): one producer reading 4 MiB chunks (LENGTH,
// kmer.h:8), sequence lines found by counting newlines (num & 3 == 2), a sequence line split
// across two chunks carried over, pairs re-synchronised by read index; NUM_THREAD-1 consumers.
// What differs: a consumer does not scan the reads itself -- it packs the chunk into pinned
// memory and submits it to its own HIP stream (trew_hip_submit), so decode/pack of chunk i+1
// overlaps the device scan of chunk i.
#include <zlib.h>

#include <cerrno>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>

#include "bgzf_reader.hpp"
#include "trew_host.hpp"

namespace trew_host {

static const int LENGTH = 1 << 22;  // kmer.h:8
static const int MAX_SEQ = 1000;    // kmer.h:10

[[noreturn]] static void die(const char *msg) {
    fprintf(stderr, "%s\n", msg);
    exit(EXIT_FAILURE);
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

static FileReader open_reader(const char *file_name, bool is_gz) {
    FileReader r;
    r.is_gz = is_gz;
    if (is_gz && BgzfReader::sniff(file_name)) {
        const unsigned hw = std::thread::hardware_concurrency();
        r.bgzf = new BgzfReader(file_name, (int) std::min(8u, std::max(2u, hw / 2)));
        if (!r.bgzf->ok()) {
            fprintf(stderr, "File open failed: %s\n", file_name);
            exit(EXIT_FAILURE);
        }
        return r;
    }
    if (is_gz) {
        r.gz_fp = gzopen(file_name, "r");
        if (!r.gz_fp) {  // kmer.cpp:1288-1289
            fprintf(stderr, "File open failed: %s\n", file_name);
            exit(EXIT_FAILURE);
        }
        gzbuffer(r.gz_fp, 1 << 20);
    } else {
        r.fp = fopen(file_name, "r");
        if (!r.fp) {
            fprintf(stderr, "File open failed: %s\n", file_name);
            exit(EXIT_FAILURE);
        }
    }
    return r;
}

// QueueData / PairQueueData, kmer.h:93-103
struct Chunk {
    char *buffer1 = nullptr, *buffer2 = nullptr;
    std::vector<int64_t> st1, nd1, st2, nd2;
    bool sentinel = false;
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
    int dev_index = 0;  // index into Scanner::ctx
    int slot = 0;
    uint32_t *h_words = nullptr, *h_offsets = nullptr, *h_lengths = nullptr;
    uint64_t words_cap = 0, reads_cap = 0;
    uint64_t reads = 0, bases = 0;
};

struct Scanner {
    Config cfg;
    int mode = TREW_MODE_SHORT;
    std::vector<trew_hip_ctx *> ctx;
    std::vector<Worker> workers;
    RunStats stats;
};

static void hip_die(trew_hip_ctx *ctx, const char *what) {
    fprintf(stderr, "%s: %s\n", what, trew_hip_last_error(ctx));
    exit(EXIT_FAILURE);
}

Scanner *scanner_create(const Config &cfg, int mode) {
    Scanner *s = new Scanner();
    s->cfg = cfg;
    s->mode = mode;
    const int n_workers = std::max(1, cfg.NUM_THREAD - 1);  // the caller's thread is the reader (kmer.cpp:1278-1301)
    const int ndev = (int) cfg.devices.size();
    const bool pair = mode == TREW_MODE_PAIR;
    const uint64_t reads_cap = (pair ? 2ull : 1ull) << 20;
    const uint64_t words_cap = 3ull * ((uint64_t) (pair ? 2 : 1) * (LENGTH / 32) + reads_cap) + 64;
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
        p.n_slots = std::max(1, slots_on_dev[(size_t) d]);
        p.max_batch_words = words_cap;
        p.max_batch_reads = reads_cap;
        p.table_log2_slots = 22;
        p.flags = 0;
        trew_hip_ctx *c = nullptr;
        if (trew_hip_init(&p, &c) != 0) {
            fprintf(stderr, "%s\n", trew_hip_last_error(nullptr));
            exit(EXIT_FAILURE);
        }
        s->ctx.push_back(c);
    }
    std::vector<int> next_slot((size_t) ndev, 0);
    for (int w = 0; w < n_workers; w++) {
        Worker wk;
        wk.dev_index = w % ndev;
        wk.slot = next_slot[(size_t) wk.dev_index]++;
        wk.words_cap = words_cap;
        wk.reads_cap = reads_cap;
        trew_hip_ctx *c = s->ctx[(size_t) wk.dev_index];
        if (trew_hip_host_alloc(c, words_cap * 4, (void **) &wk.h_words) || trew_hip_host_alloc(c, reads_cap * 4, (void **) &wk.h_offsets) ||
            trew_hip_host_alloc(c, reads_cap * 4, (void **) &wk.h_lengths))
            hip_die(c, "pinned allocation");
        s->workers.push_back(wk);
    }
    return s;
}

void scanner_destroy(Scanner *s) {
    if (!s) return;
    for (auto &w : s->workers) {
        trew_hip_ctx *c = s->ctx[(size_t) w.dev_index];
        trew_hip_host_free(c, w.h_words);
        trew_hip_host_free(c, w.h_offsets);
        trew_hip_host_free(c, w.h_lengths);
    }
    for (auto c : s->ctx) trew_hip_destroy(c);
    delete s;
}

const RunStats &last_stats(const Scanner *s) { return s->stats; }

// the consumer: buffer_task* (kmer.cpp:80-985) with the scan itself moved to the device
static void worker_loop(Scanner *s, Worker *w, ChunkQueue *q) {
    trew_hip_ctx *c = s->ctx[(size_t) w->dev_index];
    for (;;) {
        Chunk *ch = q->pop();
        if (ch->sentinel) {  // loc_vector == nullptr, kmer.cpp:108-110
            delete ch;
            break;
        }
        if (trew_hip_wait(c, w->slot)) hip_die(c, "trew_hip_wait");  // the slot's pinned buffers are free again
        trew_hip_batch b;
        memset(&b, 0, sizeof(b));
        uint64_t nw;
        if (s->mode == TREW_MODE_PAIR) {
            const uint64_t np = std::min(ch->st1.size(), ch->st2.size());  // kmer.cpp:321
            nw = trew_pack_pairs(ch->buffer1, ch->st1.data(), ch->nd1.data(), ch->buffer2, ch->st2.data(), ch->nd2.data(), np, w->h_words,
                                 w->words_cap, w->h_offsets, w->h_lengths);
            b.n_reads = 2 * np;
        } else {
            nw = trew_pack_reads(ch->buffer1, ch->st1.data(), ch->nd1.data(), ch->st1.size(), w->h_words, w->words_cap, w->h_offsets, w->h_lengths);
            b.n_reads = ch->st1.size();
        }
        if (nw == (uint64_t) -1) die("internal error: packed chunk exceeds the slot buffer");
        for (uint64_t i = 0; i < b.n_reads; i++) w->bases += w->h_lengths[i];
        w->reads += b.n_reads;
        b.words = w->h_words;
        b.n_words = nw;
        b.offsets = w->h_offsets;
        b.lengths = w->h_lengths;
        if (b.n_reads && trew_hip_submit(c, &b, w->slot)) hip_die(c, "trew_hip_submit");
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

// read_fastq_thread (kmer.cpp:987-1038) and read_fastq_long_thread (1166-1213)
static void read_fastq_thread(FileReader &fr, ChunkQueue *q, bool long_mode, int slice_length) {
    int num = 0, shift = 0, idx = -1;
    char *buffer = alloc_buffer();
    for (;;) {
        Chunk *ch = new Chunk();
        const int bytes_read = fr.read(buffer + shift, LENGTH - 1 - shift);
        const int total = (bytes_read > 0 ? bytes_read : 0) + shift;
        buffer[total] = '\0';
        for (const char *nl = (const char *) memchr(buffer, '\n', (size_t) total); nl; nl = (const char *) memchr(nl + 1, '\n', (size_t) (buffer + total - nl - 1))) {
            const int i = (int) (nl - buffer);
            num += 1;
            if ((num & 3) == 2) {
                const int len = (i - 1) - (idx + 1) + 1;
                if (long_mode) {
                    if (len >= slice_length) {  // kmer.cpp:1184
                        ch->st1.push_back(idx + 1);
                        ch->nd1.push_back(i - 1);
                    }
                } else {
                    if (len > MAX_SEQ) die("This mode is designed for short-read sequencing. Please use 'trew long'.");  // kmer.cpp:1006-1009
                    ch->st1.push_back(idx + 1);
                    ch->nd1.push_back(i - 1);
                }
            }
            idx = i;
        }
        ch->buffer1 = buffer;
        if (bytes_read <= 0) {
            q->push(ch);
            if (fr.eof()) break;
            fprintf(stderr, "File-IO Error: %s.\n", fr.error());  // kmer.cpp:1021-1022
            exit(EXIT_FAILURE);
        }
        char *buffer_new = alloc_buffer();
        if ((num & 3) == 1) {  // inside a sequence line: carry it over (kmer.cpp:1026-1029)
            const int rest = total - idx - 1;
            memcpy(buffer_new, buffer + idx + 1, (size_t) rest);
            shift = rest;
            idx = -1;
            if (shift >= LENGTH - 2) die("a read does not fit one 4 MiB chunk");
        } else {
            shift = 0;
            idx = -1;  // positions restart in the new buffer
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
                        exit(EXIT_FAILURE);
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
                exit(EXIT_FAILURE);
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

static ResultMapData collect_tables(Scanner *s) {
    ResultMapData r;
    for (auto c : s->ctx) {
        uint64_t n = 0;
        if (trew_hip_collect(c, -1, nullptr, 0, &n)) hip_die(c, "trew_hip_collect");
        std::vector<trew_hip_row> rows((size_t) std::max<uint64_t>(n, 1));
        if (trew_hip_collect(c, -1, rows.data(), n, &n)) hip_die(c, "trew_hip_collect");
        for (uint64_t i = 0; i < n; i++) {
            const uint128_t w = ((uint128_t) rows[i].word_hi << 64) | rows[i].word_lo;
            r.table[rows[i].table][KmerSeq{rows[i].k, w}] += rows[i].count;  // thread merge, kmer.cpp:1486-1515
        }
        if (trew_hip_reset_tables(c)) hip_die(c, "trew_hip_reset_tables");
    }
    return r;
}

static FinalFastqOutput run_file(Scanner *s, const Config &cfg, const char *name1, const char *name2, bool gz1, bool gz2) {
    const auto t0 = std::chrono::steady_clock::now();
    size_t qcap = cfg.QUEUE_SIZE >= 4 ? (size_t) (cfg.QUEUE_SIZE / 4) : 256;  // kmer.cpp:1274-1276; "unlimited" is capped at 1 GiB
    ChunkQueue q(qcap);
    for (auto &w : s->workers) {
        w.reads = 0;
        w.bases = 0;
    }
    std::vector<std::thread> th;
    for (auto &w : s->workers) th.emplace_back(worker_loop, s, &w, &q);
    FileReader f1 = open_reader(name1, gz1);
    if (s->mode == TREW_MODE_PAIR) {
        FileReader f2 = open_reader(name2, gz2);
        read_pair_fastq_thread(f1, f2, &q);
        f2.close();
    } else {
        read_fastq_thread(f1, &q, s->mode == TREW_MODE_LONG, cfg.SLICE_LENGTH);
    }
    f1.close();
    for (size_t i = 0; i < s->workers.size(); i++) {  // sentinels, kmer.cpp:1304-1310
        Chunk *c = new Chunk();
        c->sentinel = true;
        q.push(c);
    }
    for (auto &t : th) t.join();
    ResultMapData r = collect_tables(s);
    s->stats = RunStats();
    for (auto &w : s->workers) {
        s->stats.reads += w.reads;
        s->stats.bases += w.bases;
    }
    s->stats.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (cfg.stats)
        fprintf(stderr, "[trew] %s: %llu reads, %llu bases, %.3f s, %.3f Gbases/s end-to-end (decode + pack + scan)\n", name1,
                (unsigned long long) s->stats.reads, (unsigned long long) s->stats.bases, s->stats.seconds,
                s->stats.bases / s->stats.seconds / 1e9);
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
