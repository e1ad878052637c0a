// Test harness (CPU only): feeds rows "table k word_hi word_lo count" from stdin to the product's
// process_output / final_process_output (trew_amd/csrc/host/output.cpp) and prints what the CLI would print.
// A line "file NAME" starts a new file section.  Built and driven by tests/test_output_cpu.py.
#include <cinttypes>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../trew_amd/csrc/host/trew_host.hpp"

using namespace trew_host;

int main(int argc, char **argv) {
    const int min_mer = argc > 1 ? atoi(argv[1]) : 5;
    FinalFastqData total_high, total_low;
    ResultMapData cur;
    std::string name;
    bool have = false;
    auto finish = [&]() {
        if (!have) return;
        FinalFastqOutput fo = process_output(name.c_str(), cur, min_mer, stdout);
        for (const auto &kv : fo.high) {
            FinalData &d = total_high[kv.first];
            d.forward += kv.second.forward, d.backward += kv.second.backward, d.both += kv.second.both;
        }
        for (const auto &kv : fo.low) {
            FinalData &d = total_low[kv.first];
            d.forward += kv.second.forward, d.backward += kv.second.backward, d.both += kv.second.both;
        }
        cur = ResultMapData();
    };
    char line[512];
    while (fgets(line, sizeof line, stdin)) {
        if (!strncmp(line, "file ", 5)) {
            finish();
            name = std::string(line + 5);
            while (!name.empty() && (name.back() == '\n' || name.back() == '\r')) name.pop_back();
            have = true;
            continue;
        }
        int table, k;
        uint64_t hi, lo, count;
        if (sscanf(line, "%d %d %" SCNu64 " %" SCNu64 " %" SCNu64, &table, &k, &hi, &lo, &count) != 5) continue;
        cur.table[table][KmerSeq{k, ((uint128_t) hi << 64) | lo}] += count;
    }
    finish();
    final_process_output(total_high, total_low, stdout);
    return 0;
}
