// bgzf_reader.hpp -- parallel inflate of block-gzip (BGZF, the .gz flavour written by bgzip /
// htslib) input.  A BGZF file is a series of independent gzip members of <= 64 KiB, each carrying
// its compressed size in a 'BC' extra subfield, so members can be inflated concurrently; gzread
// (what the reference uses, kmer.h:157-204) inflates the same bytes on one thread.  Plain gzip files
// do not have the subfield and keep going through zlib's gzread.  A file may also mix the two (`cat a.bgz b.gz`,
// which gzread reads without trouble): when a member without the subfield is met after BGZF members, the rest of
// the file is handed to zlib from that offset.
#pragma once
#include <fcntl.h>
#include <unistd.h>
#include <zlib.h>

#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace trew_host {

class BgzfReader {
public:
    // true when the file starts with a BGZF member header
    static bool sniff(const char *file_name) {
        FILE *f = fopen(file_name, "rb");
        if (!f) return false;
        std::vector<unsigned char> h;
        const int bs = read_header(f, h);
        fclose(f);
        return bs > 0;
    }

    BgzfReader(const char *file_name, int n_threads) : name_(file_name) {
        fp_ = fopen(file_name, "rb");
        if (!fp_) {
            fail("cannot open file");
            return;
        }
        if (n_threads < 1) n_threads = 1;
        producer_ = std::thread([this] { produce(); });
        for (int i = 0; i < n_threads; i++) workers_.emplace_back([this] { work(); });
    }
    ~BgzfReader() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_in_.notify_all();
        cv_out_.notify_all();
        cv_space_.notify_all();
        if (producer_.joinable()) producer_.join();
        for (auto &t : workers_) t.join();
        if (tail_) gzclose(tail_);
        if (fp_) fclose(fp_);
    }
    bool ok() const { return fp_ != nullptr; }

    // same contract as gzread: up to `length` decompressed bytes, 0 at end of file, -1 on error
    int read(char *buffer, int length) {
        int got = 0;
        while (got < length) {
            if (tail_) {  // the rest of the file is ordinary gzip: zlib reads it, as the reference does
                const int r = gzread(tail_, buffer + got, (unsigned) (length - got));
                if (r < 0) {
                    int e;
                    fail(std::string("gzip tail: ") + gzerror(tail_, &e));
                    return -1;
                }
                got += r;
                if (r == 0 || gzeof(tail_)) {
                    eof_ = gzeof(tail_) != 0;
                    if (r == 0) return got;
                }
                continue;
            }
            if (cur_pos_ == cur_.size()) {
                std::unique_lock<std::mutex> lk(m_);
                cv_out_.wait(lk, [&] { return stop_ || !error_.empty() || done_.count(next_out_) || (produced_all_ && next_out_ == n_blocks_); });
                if (!error_.empty()) return -1;
                if (stop_) return got;
                auto it = done_.find(next_out_);
                if (it == done_.end()) {  // every block consumed
                    if (tail_offset_ >= 0) {  // ... and a plain gzip member follows
                        const long long off = tail_offset_;
                        tail_offset_ = -1;
                        lk.unlock();
                        const int fd = open(name_.c_str(), O_RDONLY);
                        if (fd < 0 || lseek(fd, (off_t) off, SEEK_SET) != (off_t) off || !(tail_ = gzdopen(fd, "r"))) {
                            if (fd >= 0) close(fd);
                            fail("cannot reopen the file for its gzip tail");
                            return -1;
                        }
                        gzbuffer(tail_, 1 << 20);
                        continue;
                    }
                    eof_ = true;
                    return got;
                }
                cur_.swap(it->second);
                cur_pos_ = 0;
                done_.erase(it);
                next_out_++;
                lk.unlock();
                cv_space_.notify_all();
                continue;
            }
            const size_t take = std::min<size_t>((size_t) (length - got), cur_.size() - cur_pos_);
            memcpy(buffer + got, cur_.data() + cur_pos_, take);
            cur_pos_ += take;
            got += (int) take;
        }
        return got;
    }
    bool eof() const { return eof_; }
    const char *error() {
        std::lock_guard<std::mutex> lk(m_);
        return error_.empty() ? "" : error_.c_str();
    }

private:
    struct Job {
        uint64_t seq;
        std::vector<unsigned char> comp;  // one whole member
    };
    // Reads one gzip member header (the 12 fixed bytes + the extra field) from f into h.  Returns the total size
    // of the member when the extra field carries a 'BC' subfield (RFC 1952 2.3.1.1: any position among the
    // subfields), 0 when the bytes are a gzip header without it, -1 when they are not a gzip header at all,
    // -2 at a clean end of file, -3 when the file ends inside the header.
    static int read_header(FILE *f, std::vector<unsigned char> &h) {
        h.resize(12);
        const size_t n = fread(h.data(), 1, 12, f);
        if (n == 0) return -2;
        if (n < 2 || h[0] != 0x1f || h[1] != 0x8b) return -1;
        if (n < 10) return -3;  // the magic, then the file ends inside the header
        if (h[2] != 8 || !(h[3] & 4)) return 0;
        if (n < 12) return -3;
        const size_t xlen = (size_t) h[10] | ((size_t) h[11] << 8);
        h.resize(12 + xlen);
        if (fread(h.data() + 12, 1, xlen, f) != xlen) return -3;
        for (size_t p = 12; p + 4 <= 12 + xlen;) {
            const size_t slen = (size_t) h[p + 2] | ((size_t) h[p + 3] << 8);
            if (h[p] == 'B' && h[p + 1] == 'C' && slen == 2 && p + 6 <= 12 + xlen) return ((int) h[p + 4] | ((int) h[p + 5] << 8)) + 1;
            p += 4 + slen;
        }
        return 0;
    }
    void fail(const std::string &msg) {
        std::lock_guard<std::mutex> lk(m_);
        if (error_.empty()) error_ = msg;
        cv_out_.notify_all();
        cv_in_.notify_all();
        cv_space_.notify_all();
    }
    void produce() {
        uint64_t seq = 0;
        std::vector<unsigned char> h;
        for (;;) {
            const long long at = ftello(fp_);
            const int bs = read_header(fp_, h);
            if (bs == -2) break;  // clean end of file
            if (bs == -1) break;  // not gzip: trailing garbage, which gzread ignores as well
            if (bs == -3) {
                fail("truncated BGZF member");
                return;
            }
            if (bs == 0) {        // a gzip member without the BC subfield: the rest goes through zlib
                std::lock_guard<std::mutex> lk(m_);
                tail_offset_ = at;
                break;
            }
            if ((size_t) bs < h.size() + 8) {
                fail("corrupt BGZF member (size field smaller than its header)");
                return;
            }
            Job j;
            j.seq = seq++;
            j.comp.resize((size_t) bs);
            memcpy(j.comp.data(), h.data(), h.size());
            if (fread(j.comp.data() + h.size(), 1, (size_t) bs - h.size(), fp_) != (size_t) bs - h.size()) {
                fail("truncated BGZF member");
                return;
            }
            std::unique_lock<std::mutex> lk(m_);
            // bounded look-ahead: at most kWindow members between the consumer and the producer
            cv_space_.wait(lk, [&] { return stop_ || !error_.empty() || j.seq < next_out_ + kWindow; });
            if (stop_ || !error_.empty()) return;
            in_.push_back(std::move(j));
            lk.unlock();
            cv_in_.notify_one();
        }
        std::lock_guard<std::mutex> lk(m_);
        n_blocks_ = seq;
        produced_all_ = true;
        cv_in_.notify_all();
        cv_out_.notify_all();
    }
    void work() {
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_in_.wait(lk, [&] { return stop_ || !error_.empty() || !in_.empty() || produced_all_; });
                if (stop_ || !error_.empty()) return;
                if (in_.empty()) {
                    if (produced_all_) return;
                    continue;
                }
                j = std::move(in_.front());
                in_.pop_front();
            }
            const unsigned char *c = j.comp.data();
            const size_t n = j.comp.size();
            const int xlen = c[10] | (c[11] << 8);
            const size_t off = 12u + (size_t) xlen;
            if (n < off + 8) {
                fail("corrupt BGZF member");
                return;
            }
            const uint32_t isize = (uint32_t) c[n - 4] | ((uint32_t) c[n - 3] << 8) | ((uint32_t) c[n - 2] << 16) | ((uint32_t) c[n - 1] << 24);
            const uint32_t crc = (uint32_t) c[n - 8] | ((uint32_t) c[n - 7] << 8) | ((uint32_t) c[n - 6] << 16) | ((uint32_t) c[n - 5] << 24);
            if (isize > 65536u) {  // a BGZF member holds at most 64 KiB: never allocate from a corrupt trailer
                fail("corrupt BGZF member (uncompressed size above 64 KiB)");
                return;
            }
            std::vector<unsigned char> out(isize);
            if (isize) {
                z_stream zs;
                memset(&zs, 0, sizeof(zs));
                if (inflateInit2(&zs, -15) != Z_OK) {
                    fail("inflateInit2 failed");
                    return;
                }
                zs.next_in = const_cast<unsigned char *>(c + off);
                zs.avail_in = (uInt) (n - off - 8);
                zs.next_out = out.data();
                zs.avail_out = isize;
                const int rc = inflate(&zs, Z_FINISH);
                inflateEnd(&zs);
                if (rc != Z_STREAM_END || zs.total_out != isize || (uint32_t) crc32(crc32(0L, Z_NULL, 0), out.data(), isize) != crc) {
                    fail("BGZF member fails to inflate (data error)");
                    return;
                }
            }
            {
                std::lock_guard<std::mutex> lk(m_);
                done_[j.seq] = std::move(out);
            }
            cv_out_.notify_all();
        }
    }

    static constexpr uint64_t kWindow = 256;  // members in flight (<= 16 MiB of output)
    std::string name_;
    FILE *fp_ = nullptr;
    gzFile tail_ = nullptr;       // plain gzip remainder of a mixed file (consumer thread only)
    long long tail_offset_ = -1;  // where it starts (set by the producer under m_)
    std::thread producer_;
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_in_, cv_out_, cv_space_;
    std::deque<Job> in_;
    std::map<uint64_t, std::vector<unsigned char>> done_;
    uint64_t next_out_ = 0, n_blocks_ = 0;
    bool produced_all_ = false, stop_ = false, eof_ = false;
    std::string error_;
    std::vector<unsigned char> cur_;
    size_t cur_pos_ = 0;
};

// ------------------------------------------------------------------ BGZF as random-access text (round 4)
// Every BGZF member says how long it is compressed ('BC' subfield) and uncompressed (ISIZE, its last four bytes), so one
// cheap walk over the headers of the mapped file gives the position of every member's text in the decompressed stream without
// inflating anything.  The block-parallel FASTQ reader (fastq_blocks.hpp, process.cpp) then treats the decompressed stream as
// a file it can address: worker threads claim groups of consecutive members ("blocks" of ~4 MiB of text), inflate them straight
// into their place in a reserved address range and locate the sequence lines there -- no serial reader thread any more.
struct BgzfIndex {
    struct Member {
        uint64_t coff;   // offset of the member in the file
        uint32_t csize;  // its size there
        uint32_t hsize;  // of which header (the deflate stream starts behind it)
        uint32_t isize;  // uncompressed size
        uint64_t toff;   // offset of its text in the decompressed stream
    };
    std::vector<Member> members;
    uint64_t text_size = 0;

    // false (with a reason) when the file is not BGZF from end to end -- a plain gzip member, trailing bytes, a truncated or
    // implausible member: the caller takes the serial reader, which handles (or reports) all of these as before
    bool build(const unsigned char *p, size_t n, std::string *why) {
        members.clear();
        text_size = 0;
        size_t at = 0;
        while (at < n) {
            if (n - at < 18 || p[at] != 0x1f || p[at + 1] != 0x8b || p[at + 2] != 8 || !(p[at + 3] & 4)) {
                *why = "a member without the BGZF extra field";
                return false;
            }
            if (p[at + 3] & ~4u) {  // FNAME / FCOMMENT / FHCRC: legal gzip, never written by bgzip
                *why = "a member with optional gzip header fields";
                return false;
            }
            const size_t xlen = (size_t) p[at + 10] | ((size_t) p[at + 11] << 8);
            if (at + 12 + xlen > n) {
                *why = "truncated member header";
                return false;
            }
            size_t bsize = 0;
            for (size_t q = at + 12; q + 4 <= at + 12 + xlen;) {
                const size_t slen = (size_t) p[q + 2] | ((size_t) p[q + 3] << 8);
                if (p[q] == 'B' && p[q + 1] == 'C' && slen == 2 && q + 6 <= at + 12 + xlen) bsize = ((size_t) p[q + 4] | ((size_t) p[q + 5] << 8)) + 1;
                q += 4 + slen;
            }
            if (bsize == 0) {
                *why = "a member without the BC subfield";
                return false;
            }
            if (bsize < 12 + xlen + 8 || at + bsize > n) {
                *why = "truncated or corrupt member";
                return false;
            }
            const unsigned char *t = p + at + bsize - 4;
            const uint32_t isize = (uint32_t) t[0] | ((uint32_t) t[1] << 8) | ((uint32_t) t[2] << 16) | ((uint32_t) t[3] << 24);
            if (isize > 65536u) {
                *why = "a member with more than 64 KiB of text";
                return false;
            }
            Member m;
            m.coff = at;
            m.csize = (uint32_t) bsize;
            m.hsize = (uint32_t) (12 + xlen);
            m.isize = isize;
            m.toff = text_size;
            members.push_back(m);
            text_size += isize;
            at += bsize;
        }
        return true;
    }

    // inflates member i into dst (isize bytes), checking length and CRC as BgzfReader does; false on a data error
    bool inflate_member(const unsigned char *file, size_t i, unsigned char *dst) const {
        const Member &m = members[i];
        if (m.isize == 0) return true;
        const unsigned char *c = file + m.coff;
        const uint32_t crc = (uint32_t) c[m.csize - 8] | ((uint32_t) c[m.csize - 7] << 8) | ((uint32_t) c[m.csize - 6] << 16) | ((uint32_t) c[m.csize - 5] << 24);
        z_stream zs;
        memset(&zs, 0, sizeof(zs));
        if (inflateInit2(&zs, -15) != Z_OK) return false;
        zs.next_in = const_cast<unsigned char *>(c + m.hsize);
        zs.avail_in = (uInt) (m.csize - m.hsize - 8);
        zs.next_out = dst;
        zs.avail_out = m.isize;
        const int rc = inflate(&zs, Z_FINISH);
        const bool ok = rc == Z_STREAM_END && zs.total_out == m.isize;
        inflateEnd(&zs);
        return ok && (uint32_t) crc32(crc32(0L, Z_NULL, 0), dst, m.isize) == crc;
    }
};

}  // namespace trew_host
