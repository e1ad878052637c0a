// bgzf_cat -- writes the decompressed content of a BGZF file to stdout through BgzfReader
// (test helper for tests/test_bgzf_cpu.py; also handy to check a file before a long run).
#include <cstdio>
#include <cstdlib>

#include "bgzf_reader.hpp"

int main(int argc, char **argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: bgzf_cat file.gz [threads] [read_size]\n");
        return 2;
    }
    if (!trew_host::BgzfReader::sniff(argv[1])) {
        fprintf(stderr, "not a BGZF file\n");
        return 3;
    }
    const int threads = argc > 2 ? atoi(argv[2]) : 4;
    const int chunk = argc > 3 ? atoi(argv[3]) : (1 << 20);
    trew_host::BgzfReader r(argv[1], threads);
    if (!r.ok()) return 4;
    std::vector<char> buf((size_t) chunk);
    for (;;) {
        const int n = r.read(buf.data(), chunk);
        if (n < 0) {
            fprintf(stderr, "error: %s\n", r.error());
            return 1;
        }
        if (n == 0) break;
        fwrite(buf.data(), 1, (size_t) n, stdout);
    }
    return r.eof() ? 0 : 5;
}
