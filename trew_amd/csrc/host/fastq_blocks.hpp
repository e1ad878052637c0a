// fastq_blocks.hpp -- block-parallel location of the sequence lines of a plain (uncompressed) FASTQ file.
//
// The reference finds sequence lines by counting newlines from the start of the file: the newline that makes
// `num & 3 == 2` closes one (read_fastq_thread, kmer.cpp:987-1038), i.e. line number 1 mod 4 (0-based), whatever
// the lines contain.  Here the mapped file is cut into fixed-size blocks that worker threads claim in file order.
// A worker records the newlines of its block, learns how many newlines precede the block from its predecessor
// (a chain of additions -- the only serial part), and reports the sequence lines that END in its block; a line
// that starts in an earlier block is simply read from there, which is what the reference's carry-over of a split
// line achieves.  Line numbering is exact: no record-boundary heuristics, the same reads as the serial reader.
#pragma once
#include <sys/mman.h>

#include <atomic>
#include <cstdint>
#include <functional>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <memory>
#include <thread>
#include <vector>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace trew_host {

// offsets (relative to p) of every '\n' in p[0, n), n < 2^32; returns how many
#if defined(__x86_64__)
// (A version that stores the first two set bits of every 64-byte chunk unconditionally, to spare the branch of the bit
// loop, was measured and is slower on FASTQ text -- 5.2-6.2 against 7.7 GB/s per thread: records are regular enough for
// the predictor, and the scan is close to what one thread streams from DRAM.)
__attribute__((target("avx2"))) inline size_t scan_newlines_avx2(const char *p, size_t n, uint32_t *out) {
    size_t cnt = 0, i = 0;
    const __m256i nl = _mm256_set1_epi8('\n');
    for (; i + 64 <= n; i += 64) {
        const uint32_t m0 = (uint32_t) _mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i *) (p + i)), nl));
        const uint32_t m1 = (uint32_t) _mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i *) (p + i + 32)), nl));
        uint64_t m = ((uint64_t) m1 << 32) | m0;
        while (m) {
            out[cnt++] = (uint32_t) (i + (size_t) __builtin_ctzll(m));
            m &= m - 1;
        }
    }
    for (; i < n; i++)
        if (p[i] == '\n') out[cnt++] = (uint32_t) i;
    return cnt;
}
// the same with one 64-byte compare per chunk (the mask comes straight out of the compare: no movemask, no merge).  Opt-in
// (TREW_SCAN_ISA=avx512): on the EPYC 9575F of the MI355X hosts it is SLOWER than the AVX2 loop -- 0.103 against 0.076 s of
// scan per worker, 29.5 against 32.3 Gbases/s end to end (profiles/r03/README.md).
__attribute__((target("avx512bw,avx512f"))) inline size_t scan_newlines_avx512(const char *p, size_t n, uint32_t *out) {
    size_t cnt = 0, i = 0;
    const __m512i nl = _mm512_set1_epi8('\n');
    for (; i + 64 <= n; i += 64) {
        uint64_t m = (uint64_t) _mm512_cmpeq_epi8_mask(_mm512_loadu_si512((const void *) (p + i)), nl);
        while (m) {
            out[cnt++] = (uint32_t) (i + (size_t) __builtin_ctzll(m));
            m &= m - 1;
        }
    }
    for (; i < n; i++)
        if (p[i] == '\n') out[cnt++] = (uint32_t) i;
    return cnt;
}
inline int scan_isa() {  // 2: AVX-512BW, 1: AVX2 (the default where present), 0: memchr; TREW_SCAN_ISA=avx512|scalar overrides
    static const int isa = [] {
        int v = __builtin_cpu_supports("avx2") ? 1 : 0;
        if (const char *e = getenv("TREW_SCAN_ISA")) {
            if (!strcmp(e, "avx512") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512f")) v = 2;
            if (!strcmp(e, "scalar")) v = 0;
        }
        return v;
    }();
    return isa;
}
#endif
inline size_t scan_newlines(const char *p, size_t n, uint32_t *out) {
#if defined(__x86_64__)
    const int isa = scan_isa();
    if (isa == 2) return scan_newlines_avx512(p, n, out);
    if (isa == 1) return scan_newlines_avx2(p, n, out);
#endif
    size_t cnt = 0;
    for (const char *q = (const char *) memchr(p, '\n', n); q; q = (const char *) memchr(q + 1, '\n', (size_t) (p + n - q - 1))) out[cnt++] = (uint32_t) (q - p);
    return cnt;
}

// number of '\n' in p[0, n)
#if defined(__x86_64__)
__attribute__((target("avx2,popcnt"))) inline size_t count_newlines_avx2(const char *p, size_t n) {
    size_t cnt = 0, i = 0;
    const __m256i nl = _mm256_set1_epi8('\n');
    for (; i + 32 <= n; i += 32) cnt += (size_t) __builtin_popcount((uint32_t) _mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i *) (p + i)), nl)));
    for (; i < n; i++) cnt += p[i] == '\n';
    return cnt;
}
#endif
inline size_t count_newlines(const char *p, size_t n) {
#if defined(__x86_64__)
    static const bool have_avx2 = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("popcnt");
    if (have_avx2) return count_newlines_avx2(p, n);
#endif
    size_t cnt = 0;
    for (size_t i = 0; i < n; i++) cnt += p[i] == '\n';
    return cnt;
}

// Walks the newlines of a mapped file from a byte position on, a window at a time (the paired reader: a worker
// needs lines 4r .. 4r+3 of read r in BOTH files, and knows from the per-block newline counts where line 4r starts).
struct LineCursor {
    const char *base = nullptr;
    size_t size = 0, win_lo = 0, win_hi = 0, idx = 0, cnt = 0;
    std::vector<uint32_t> nl;
    static constexpr size_t kWindow = (size_t) 1 << 18;
    void init(const char *b, size_t n, size_t pos) {
        base = b;
        size = n;
        win_lo = win_hi = pos;
        idx = cnt = 0;
        if (nl.size() < kWindow + 2) nl.resize(kWindow + 2);  // + 2: scan_newlines stores two slots ahead
    }
    // absolute offset of the next '\n' at or after the cursor, -1 at the end of the file
    int64_t next() {
        while (idx == cnt) {
            if (win_hi >= size) return -1;
            win_lo = win_hi;
            win_hi = win_lo + kWindow < size ? win_lo + kWindow : size;
            cnt = scan_newlines(base + win_lo, win_hi - win_lo, nl.data());
            idx = 0;
        }
        return (int64_t) (win_lo + nl[idx++]);
    }
};

// Per-block newline counts of one mapped file (the paired reader's first pass) and the lookup they allow.
struct LineIndex {
    const char *base = nullptr;
    size_t size = 0, block = 0, n_blocks = 0;
    std::vector<int64_t> before;  // before[b] = newlines in [0, start of block b); before[n_blocks] = total
    void init(const char *b, size_t n, size_t blk) {
        base = b;
        size = n;
        block = blk;
        n_blocks = (n + blk - 1) / blk;
        before.assign(n_blocks + 1, 0);
    }
    void count_block(size_t b) {  // any thread, any order; before[b + 1] holds the block's own count until finish()
        const size_t lo = b * block, hi = lo + block < size ? lo + block : size;
        before[b + 1] = (int64_t) count_newlines(base + lo, hi - lo);
    }
    void finish() {
        for (size_t b = 0; b < n_blocks; b++) before[b + 1] += before[b];
    }
    int64_t total() const { return before[n_blocks]; }
    // byte offset at which line number `line` (0-based) starts; the line must exist or be the one past the last newline
    size_t line_start(int64_t line, std::vector<uint32_t> &scratch) const {
        if (line <= 0) return 0;
        const int64_t want = line - 1;  // the newline that closes the previous line
        size_t lo = 0, hi = n_blocks;   // block with before[b] <= want < before[b + 1]
        while (lo + 1 < hi) {
            const size_t mid = (lo + hi) / 2;
            if (before[mid] <= want)
                lo = mid;
            else
                hi = mid;
        }
        const size_t blo = lo * block, bhi = blo + block < size ? blo + block : size;
        if (scratch.size() < block + 2) scratch.resize(block + 2);
        const size_t cnt = scan_newlines(base + blo, bhi - blo, scratch.data());
        const size_t j = (size_t) (want - before[lo]);
        return j < cnt ? blo + scratch[j] + 1 : size;
    }
};

struct BlockScan {
    const char *base = nullptr;
    size_t size = 0, block = 0, n_blocks = 0;
    // blocks of unequal size (BGZF input: a block is a group of members): block b = [bounds[b], bounds[b+1]); empty = b * block.
    // `block` is then the size of the largest one.
    std::vector<size_t> bounds;
    // called with the claimed block before it is scanned (BGZF input: inflate the block's members into [lo, hi)); may be empty
    std::function<void(size_t b, size_t lo, size_t hi)> fill;
    bool anonymous = false;  // base is anonymous memory: what release() drops is gone for good (see release)
    std::atomic<size_t> next{0};
    // published by the worker of block b once its predecessor's values are known:
    // lines_end[b] = newlines in [0, end of block b) (-1: not yet known);
    // last_nl[b]   = offset of the last newline before the end of block b (-1: none).
    std::unique_ptr<std::atomic<int64_t>[]> lines_end;
    std::unique_ptr<int64_t[]> last_nl;
    bool populate = false;  // base is a page-aligned file mapping: pre-fault each block when it is claimed

    // (call init_bounds instead for blocks of unequal size)
    void init_bounds(const char *base_, size_t size_, std::vector<size_t> bounds_) {
        size_t big = 1;
        for (size_t i = 0; i + 1 < bounds_.size(); i++) big = std::max(big, bounds_[i + 1] - bounds_[i]);
        init(base_, size_, big);
        bounds = std::move(bounds_);
        n_blocks = bounds.size() - 1;
        lines_end.reset(new std::atomic<int64_t>[n_blocks ? n_blocks : 1]);
        last_nl.reset(new int64_t[n_blocks ? n_blocks : 1]);
        for (size_t i = 0; i < n_blocks; i++) lines_end[i].store(-1, std::memory_order_relaxed);
    }
    void init(const char *base_, size_t size_, size_t block_) {
        base = base_;
        size = size_;
        block = block_;
        bounds.clear();
        n_blocks = (size + block - 1) / block;
        lines_end.reset(new std::atomic<int64_t>[n_blocks]);
        last_nl.reset(new int64_t[n_blocks]);
        for (size_t i = 0; i < n_blocks; i++) lines_end[i].store(-1, std::memory_order_relaxed);
        next.store(0);
    }

    // Claims the next block; false when none is left.  st / nd receive the inclusive byte ranges [st, nd] (offsets
    // into the file, the LocationVector convention of kmer.h:73) of the sequence lines whose closing newline lies
    // in the block; nl is scratch for the block's newline offsets (at least `block` entries).
    bool claim(std::vector<uint32_t> &nl, std::vector<int64_t> &st, std::vector<int64_t> &nd, size_t *claimed = nullptr) {
        const size_t b = next.fetch_add(1);
        if (b >= n_blocks) return false;
        if (claimed) *claimed = b;
        if (nl.size() < block + 2) nl.resize(block + 2);
        const size_t lo = bounds.empty() ? b * block : bounds[b], hi = bounds.empty() ? (lo + block < size ? lo + block : size) : bounds[b + 1];
        if (fill) fill(b, lo, hi);
        if (populate) {
            // map the block's pages with one call instead of a page fault per 64 KiB of a cold mapping (Linux >= 5.14;
            // any failure just leaves the faults to happen)
#ifdef MADV_POPULATE_READ
            const size_t page = 4096, plo = lo & ~(page - 1);
            (void) madvise(const_cast<char *>(base) + plo, hi - plo, MADV_POPULATE_READ);
#endif
        }
        const size_t cnt = scan_newlines(base + lo, hi - lo, nl.data());
        // the number of newlines before this block: a chain of additions through the blocks.  Blocks are claimed in
        // file order, so every predecessor is already running on some thread, and it publishes right after its scan.
        int64_t lines_before = 0, prev_nl = -1;
        if (b > 0) {
            int64_t v;
            for (unsigned spin = 0; (v = lines_end[b - 1].load(std::memory_order_acquire)) < 0; spin++)
                if (spin > 64) std::this_thread::yield();
            lines_before = v;
            prev_nl = last_nl[b - 1];
        }
        last_nl[b] = cnt ? (int64_t) (lo + nl[cnt - 1]) : prev_nl;
        lines_end[b].store(lines_before + (int64_t) cnt, std::memory_order_release);
        st.clear();
        nd.clear();
        // newline j of the block closes line number lines_before + j; sequence lines are the numbers 1 mod 4
        for (size_t j = (size_t) ((1 - lines_before) & 3); j < cnt; j += 4) {
            const int64_t start = j > 0 ? (int64_t) (lo + nl[j - 1]) + 1 : prev_nl + 1;
            st.push_back(start);
            nd.push_back((int64_t) (lo + nl[j]) - 1);
        }
        return true;
    }

    // The caller is done with block b (its sequence bytes are copied or packed): give its page-table entries back now.
    // Tearing down the mapping of a large file in one munmap at the end costs as long as scanning it (0.22 s for 19.6 GB of
    // 4-KiB pages, single-threaded, measured); MADV_DONTNEED only takes the mapping's lock for reading, so the workers pay
    // for it in parallel, a block at a time.  A line that started in this block and ends in the next simply faults its
    // pages in again (they are still in the page cache).
    // Anonymous memory (BGZF text) does not come back: there the pages from the block's last line start on are kept -- the next
    // block reads the line that straddles the boundary from them -- and never given back (a page or two per block; with long
    // reads up to a line's length, a few per cent of the text).
    void release(size_t b) const {
        if ((!populate && !anonymous) || b >= n_blocks) return;
        const size_t page = 4096, lo = bounds.empty() ? b * block : bounds[b];
        size_t hi = bounds.empty() ? (lo + block < size ? lo + block : size) : bounds[b + 1];
        if (anonymous) {
            const int64_t ln = last_nl[b];
            hi = ln >= 0 && (size_t) ln + 1 > lo ? (size_t) ln + 1 : lo;
        }
        const size_t plo = (lo + page - 1) & ~(page - 1), phi = hi & ~(page - 1);
        if (phi > plo) (void) madvise(const_cast<char *>(base) + plo, phi - plo, MADV_DONTNEED);
    }
};

}  // namespace trew_host
