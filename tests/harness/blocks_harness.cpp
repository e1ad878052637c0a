// Test harness (CPU only): runs the product's block-parallel FASTQ line locator (host/fastq_blocks.hpp) over a
// file with a given block size on N threads and prints "start length" of every sequence line it reports,
// sorted by start.  Driven by tests/test_fastq_blocks_cpu.py.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../trew_amd/csrc/host/fastq_blocks.hpp"

using namespace trew_host;

static std::vector<char> slurp(const char *path) {
    std::vector<char> data;
    FILE *f = fopen(path, "rb");
    if (!f) exit(2);
    char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) data.insert(data.end(), buf, buf + n);
    fclose(f);
    return data;
}

// pair FILE1 FILE2 BLOCK PAIRS_PER_ITEM: the paired reader's helpers (LineIndex + LineCursor): prints
// "newlines1 newlines2" and then "start1 len1 start2 len2" per pair, items walked in reverse order.
static int pair_mode(char **argv) {
    std::vector<char> d[2] = {slurp(argv[2]), slurp(argv[3])};
    const size_t block = (size_t) atoll(argv[4]), per_item = (size_t) atoll(argv[5]);
    LineIndex ix[2];
    for (int m = 0; m < 2; m++) {
        ix[m].init(d[m].data(), d[m].size(), block);
        for (size_t b = ix[m].n_blocks; b-- > 0;) ix[m].count_block(b);  // any order
        ix[m].finish();
    }
    printf("%lld %lld\n", (long long) ix[0].total(), (long long) ix[1].total());
    const size_t n_pairs = (size_t) ((std::min(ix[0].total(), ix[1].total()) + 2) / 4);
    std::vector<uint32_t> scratch;
    const size_t n_items = (n_pairs + per_item - 1) / per_item;
    std::vector<std::vector<long long>> rows(n_pairs);
    for (size_t item = n_items; item-- > 0;) {
        const size_t r0 = item * per_item, r1 = std::min(n_pairs, r0 + per_item);
        for (int m = 0; m < 2; m++) {
            LineCursor cur;
            cur.init(ix[m].base, ix[m].size, ix[m].line_start((int64_t) (4 * r0), scratch));
            for (size_t r = r0; r < r1; r++) {
                const int64_t a = cur.next(), b = cur.next();
                cur.next();
                cur.next();
                rows[r].push_back(a + 1);
                rows[r].push_back(b - a - 1);
            }
        }
    }
    for (auto &r : rows) printf("%lld %lld %lld %lld\n", r[0], r[1], r[2], r[3]);
    return 0;
}

int main(int argc, char **argv) {
    if (argc >= 6 && std::string(argv[1]) == "pair") return pair_mode(argv);
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
