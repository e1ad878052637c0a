// bgzf_reader.hpp -- parallel inflate of block-gzip (BGZF, the .gz flavour written by bgzip /
// htslib) input.  A BGZF file is a series of independent gzip members of <= 64 KiB, each carrying
// its compressed size in a 'BC' extra subfield, so members can be inflated concurrently; gzread
// (what the reference uses, kmer.h:157-204) inflates the same bytes on one thread.  Plain gzip files
// do not have the subfield and keep going through zlib's gzread.
#pragma once
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
        unsigned char h[18];
        const size_t n = fread(h, 1, sizeof(h), f);
        fclose(f);
        return n == sizeof(h) && block_size(h) > 0;
    }

    BgzfReader(const char *file_name, int n_threads) {
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
        if (fp_) fclose(fp_);
    }
    bool ok() const { return fp_ != nullptr; }

    // same contract as gzread: up to `length` decompressed bytes, 0 at end of file, -1 on error
    int read(char *buffer, int length) {
        int got = 0;
        while (got < length) {
            if (cur_pos_ == cur_.size()) {
                std::unique_lock<std::mutex> lk(m_);
                cv_out_.wait(lk, [&] { return stop_ || !error_.empty() || done_.count(next_out_) || (produced_all_ && next_out_ == n_blocks_); });
                if (!error_.empty()) return -1;
                if (stop_) return got;
                auto it = done_.find(next_out_);
                if (it == done_.end()) {  // every block consumed
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
    // total size of the member whose first 18 bytes are h, or 0 when it is not a BGZF header
    static int block_size(const unsigned char *h) {
        if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) return 0;
        const int xlen = h[10] | (h[11] << 8);
        if (xlen < 6 || h[12] != 'B' || h[13] != 'C' || h[14] != 2 || h[15] != 0) return 0;
        return (h[16] | (h[17] << 8)) + 1;
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
        for (;;) {
            unsigned char h[18];
            const size_t n = fread(h, 1, sizeof(h), fp_);
            if (n == 0) break;  // clean end of file
            const int bs = n == sizeof(h) ? block_size(h) : 0;
            if (bs < 26) {
                fail("not a BGZF member (truncated or mixed gzip file)");
                return;
            }
            Job j;
            j.seq = seq++;
            j.comp.resize((size_t) bs);
            memcpy(j.comp.data(), h, sizeof(h));
            if (fread(j.comp.data() + sizeof(h), 1, (size_t) bs - sizeof(h), fp_) != (size_t) bs - sizeof(h)) {
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
    FILE *fp_ = nullptr;
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

}  // namespace trew_host
