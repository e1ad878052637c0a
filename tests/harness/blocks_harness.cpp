// Test harness (CPU only): runs the product's block-parallel FASTQ line locator (host/fastq_blocks.hpp) over a
// file with a given block size on N threads and prints "start length" of every sequence line it reports,
// sorted by start.  Driven by tests/test_fastq_blocks_cpu.py.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <vector>

#include "../../trew_amd/csrc/host/fastq_blocks.hpp"

using namespace trew_host;

int main(int argc, char **argv) {
    if (argc < 4) return 2;
    const size_t block = (size_t) atoll(argv[2]);
    const int threads = atoi(argv[3]);
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    std::vector<char> data;
    char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) data.insert(data.end(), buf, buf + n);
    fclose(f);
    std::vector<std::pair<long long, long long>> all;
    if (!data.empty()) {
        BlockScan bs;
        bs.init(data.data(), data.size(), block);
        std::mutex mu;
        std::vector<std::thread> th;
        for (int t = 0; t < threads; t++)
            th.emplace_back([&] {
                std::vector<uint32_t> nl;
                std::vector<int64_t> st, nd;
                std::vector<std::pair<long long, long long>> mine;
                while (bs.claim(nl, st, nd))
                    for (size_t i = 0; i < st.size(); i++) mine.emplace_back((long long) st[i], (long long) (nd[i] - st[i] + 1));
                std::lock_guard<std::mutex> lk(mu);
                all.insert(all.end(), mine.begin(), mine.end());
            });
        for (auto &t : th) t.join();
    }
    std::sort(all.begin(), all.end());
    for (auto &p : all) printf("%lld %lld\n", p.first, p.second);
    return 0;
}
