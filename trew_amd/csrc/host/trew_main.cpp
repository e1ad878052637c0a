// trew_main.cpp -- the `trew short|long MIN_MER MAX_MER` command line of the MI355X-native build.
//
// Same sub-commands, positional arguments, options, limits and messages as the reference CLI
// (trew.cpp:22-478): stdout carries the CSV sections, stderr errors/usage, exit code 1 on error.
// Additions (stderr only, stdout stays CSV-identical): --devices LIST, --stats, --table_log2_slots N,
// --serial_reader, --batch_mib N, --host_pack, --compat_g1.
#include <climits>
#include <cstdlib>
#include <cstring>
#include <string>
#include <sys/stat.h>

#include "trew_host.hpp"

using namespace trew_host;

static const char *VERSION = "0.5.0";  // trew.cpp:23

static void usage(const char *mode) {
    if (mode && !strcmp(mode, "long")) {
        fprintf(stderr,
                "Usage: long [--help] [--version] [--thread THREAD] [--table_max_mer TABLE_MAX_MER] [--low_baseline LOW_BASELINE]\n"
                "            [--high_baseline HIGH_BASELINE] [--slice_length SLICE_LENGTH] [--queue_size QUEUE_SIZE]\n"
                "            [--devices LIST] [--stats] [--table_log2_slots N] [--serial_reader] [--batch_mib N] [--host_pack] MIN_MER MAX_MER LONG_FASTQ...\n\n"
                "Estimate TRM from long-read sequencing data.\n");
    } else if (mode && !strcmp(mode, "short")) {
        fprintf(stderr,
                "Usage: short [--help] [--version] [--thread THREAD] [--paired_end] [--fq1 FASTQ_FRONT...] [--fq2 FASTQ_REVERSE...]\n"
                "             [--table_max_mer TABLE_MAX_MER] [--low_baseline LOW_BASELINE] [--high_baseline HIGH_BASELINE]\n"
                "             [--queue_size QUEUE_SIZE] [--devices LIST] [--stats] [--table_log2_slots N] [--serial_reader]\n"
                "             [--batch_mib N] [--host_pack] [--compat_g1] MIN_MER MAX_MER [SHORT_FASTQ]...\n\n"
                "Estimate TRM from short-read sequencing data.\n");
    } else {
        fprintf(stderr, "Usage: trew [--help] [--version] {long,short}\n\nSubcommands:\n  long          Estimate TRM from long-read sequencing data.\n"
                        "  short         Estimate TRM from short-read sequencing data.\n");
    }
}

static bool is_regular_file(const std::string &p) {
    struct stat st;
    return stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode);
}

static bool parse_int(const char *s, int *out) {
    char *end = nullptr;
    long v = strtol(s, &end, 10);
    if (!s[0] || *end || v < INT_MIN || v > INT_MAX) return false;
    *out = (int) v;
    return true;
}
static bool parse_double(const char *s, double *out) {
    char *end = nullptr;
    double v = strtod(s, &end);
    if (!s[0] || *end) return false;
    *out = v;
    return true;
}

static bool has_gz_ext(const std::string &p) {  // trew.cpp:407,422-433
    const size_t dot = p.find_last_of('.');
    const size_t slash = p.find_last_of('/');
    if (dot == std::string::npos || (slash != std::string::npos && dot < slash)) return false;
    const std::string ext = p.substr(dot);
    return ext == ".gz" || ext == ".bgz";
}

static std::string canonical(const std::string &p) {  // std::filesystem::canonical, trew.cpp:439-451
    char buf[PATH_MAX];
    if (realpath(p.c_str(), buf)) return std::string(buf);
    return p;
}

int main(int argc, char **argv) {
    if (argc < 2) {
        usage(nullptr);
        return 1;
    }
    const std::string mode = argv[1];
    if (mode == "--version" || mode == "-v") {
        printf("%s\n", VERSION);
        return 0;
    }
    if (mode == "--help" || mode == "-h") {
        usage(nullptr);
        return 0;
    }
    if (mode != "short" && mode != "long") {
        usage(nullptr);
        return 1;
    }
    const bool IS_SHORT = mode == "short";
    Config cfg;
    bool IS_PAIRED_END = false, fq1_used = false, fq2_used = false;
    std::vector<std::string> positional, fq1, fq2;
    std::vector<std::string> *multi = nullptr;  // option currently collecting values (--fq1 / --fq2, nargs at least one)
    for (int i = 2; i < argc; i++) {
        const std::string a = argv[i];
        auto need = [&](const char *name) -> const char * {
            if (i + 1 >= argc) {
                fprintf(stderr, "%s: expected 1 argument(s). 0 provided.\n", name);
                usage(mode.c_str());
                exit(1);
            }
            return argv[++i];
        };
        bool ok = true;
        if (a == "-h" || a == "--help") {
            usage(mode.c_str());
            return 0;
        } else if (a == "--version") {
            printf("%s\n", VERSION);
            return 0;
        } else if (a == "-t" || a == "--thread") {
            multi = nullptr;
            ok = parse_int(need("--thread"), &cfg.NUM_THREAD);
        } else if (a == "-m" || a == "--table_max_mer") {
            multi = nullptr;
            ok = parse_int(need("--table_max_mer"), &cfg.TABLE_MAX_MER);
        } else if (a == "-L" || a == "--low_baseline") {
            multi = nullptr;
            ok = parse_double(need("--low_baseline"), &cfg.LOW_BASELINE);
        } else if (a == "-H" || a == "--high_baseline") {
            multi = nullptr;
            ok = parse_double(need("--high_baseline"), &cfg.HIGH_BASELINE);
        } else if (a == "-q" || a == "--queue_size") {
            multi = nullptr;
            ok = parse_int(need("--queue_size"), &cfg.QUEUE_SIZE);
        } else if (!IS_SHORT && (a == "-s" || a == "--slice_length")) {
            multi = nullptr;
            ok = parse_int(need("--slice_length"), &cfg.SLICE_LENGTH);
        } else if (IS_SHORT && a == "--paired_end") {
            multi = nullptr;
            IS_PAIRED_END = true;
        } else if (IS_SHORT && a == "--fq1") {
            fq1_used = true;
            multi = &fq1;
        } else if (IS_SHORT && a == "--fq2") {
            fq2_used = true;
            multi = &fq2;
        } else if (a == "--stats") {
            multi = nullptr;
            cfg.stats = true;
        } else if (a == "--serial_reader") {
            multi = nullptr;
            cfg.serial_reader = true;
        } else if (a == "--host_pack") {
            cfg.host_pack = true;
        } else if (IS_SHORT && a == "--compat_g1") {
            multi = nullptr;
            cfg.compat_g1 = true;
        } else if (a == "--batch_mib") {
            multi = nullptr;
            ok = parse_int(need("--batch_mib"), &cfg.batch_mib) && cfg.batch_mib >= 1 && cfg.batch_mib <= 32;
        } else if (a == "--table_log2_slots") {
            multi = nullptr;
            ok = parse_int(need("--table_log2_slots"), &cfg.table_log2_slots);
        } else if (a == "--devices") {
            multi = nullptr;
            std::string list = need("--devices");
            cfg.devices.clear();
            size_t pos = 0;
            while (pos <= list.size()) {
                size_t comma = list.find(',', pos);
                if (comma == std::string::npos) comma = list.size();
                int d;
                if (!parse_int(list.substr(pos, comma - pos).c_str(), &d) || d < 0) ok = false;
                else cfg.devices.push_back(d);
                pos = comma + 1;
            }
            if (cfg.devices.empty()) ok = false;
        } else if (a.size() > 1 && a[0] == '-' && !(a[1] >= '0' && a[1] <= '9')) {
            fprintf(stderr, "Unknown argument: %s\n", a.c_str());
            ok = false;
        } else if (multi) {
            multi->push_back(a);
        } else {
            positional.push_back(a);
        }
        if (!ok) {
            usage(mode.c_str());
            return 1;
        }
    }
    if (positional.size() < 2 || !parse_int(positional[0].c_str(), &cfg.MIN_MER) || !parse_int(positional[1].c_str(), &cfg.MAX_MER)) {
        usage(mode.c_str());
        return 1;
    }
    std::vector<std::string> files(positional.begin() + 2, positional.end());

    // argument checks, same order and text as trew.cpp:175-228 / 256-304
    auto bad = [&](const char *msg) {
        fprintf(stderr, "%s\n", msg);
        usage(mode.c_str());
        return 1;
    };
    if (cfg.MIN_MER > cfg.MAX_MER) return bad("MIN_MER must not be greater than MAX_MER.");
    if (cfg.MIN_MER < 3) return bad("MIN_MER must be greater than or equal to 3.");
    if (cfg.MAX_MER > 64) return bad("MAX_MER must be less than or equal to 64.");
    if (cfg.TABLE_MAX_MER > 15) return bad("TABLE_MAX_MER must be less than or equal to 15.");
    if (!IS_SHORT && cfg.SLICE_LENGTH < 2 * cfg.MAX_MER) return bad("SLICE_LENGTH must be greater than or equal to twice of MAX_MER.");
    if (cfg.QUEUE_SIZE != -1 && cfg.QUEUE_SIZE < 4) return bad("QUEUE_SIZE must be -1 (unlimited) or greater than or equal to 4.");
    if (cfg.TABLE_MAX_MER <= 0) return bad("TABLE_MAX_MER must be positive.");
    if (cfg.NUM_THREAD <= 0) return bad("number of threads must be positive.");
    if (!(0 < cfg.LOW_BASELINE && cfg.LOW_BASELINE <= 1) || !(0 < cfg.HIGH_BASELINE && cfg.HIGH_BASELINE <= 1)) return bad("Baseline must be in range 0 to 1.");
    if (cfg.LOW_BASELINE > cfg.HIGH_BASELINE) return bad("Low baseline must be smaller than high baseline.");
    if (cfg.NUM_THREAD < 2) return bad("You must use at least two threads.");
    if (cfg.table_log2_slots < 12 || cfg.table_log2_slots > 30) return bad("table_log2_slots must be in range 12 to 30.");
    if (cfg.NUM_THREAD - 1 > 16 * (int) cfg.devices.size()) cfg.NUM_THREAD = 16 * (int) cfg.devices.size() + 1;  // 16 batch slots per device

    std::vector<std::string> fastq_path_list;
    if (!IS_SHORT) {
        if (files.empty()) {
            usage("long");
            return 1;
        }
        for (const auto &f : files) {
            if (!is_regular_file(f)) {
                fprintf(stderr, "%s : file not found\n", f.c_str());
                return 1;
            }
            fastq_path_list.push_back(f);
        }
    } else if (IS_PAIRED_END) {
        if (!files.empty()) return bad("SHORT_FASTQ must not be provided when --IS_PAIRED_END is used.");
        if (!fq1_used || !fq2_used) return bad("--fq1 and --fq2 are required in paired-end mode.");
        if (fq1.size() != fq2.size() || fq1.empty()) return bad("--fq1 and --fq2 must have the same number of files.");
        for (size_t i = 0; i < fq1.size(); i++) {
            if (!is_regular_file(fq1[i])) {
                fprintf(stderr, "%s : file not found\n", fq1[i].c_str());
                usage("short");
                return 1;
            }
            if (!is_regular_file(fq2[i])) {
                fprintf(stderr, "%s : file not found\n", fq2[i].c_str());
                usage("short");
                return 1;
            }
            fastq_path_list.push_back(fq1[i]);
            fastq_path_list.push_back(fq2[i]);
        }
    } else {
        if (files.empty()) return bad("SHORT_FASTQ is required in single-end mode.");
        if (fq1_used || fq2_used) return bad("--fq1 and --fq2 should not be used in single-end mode.");
        for (const auto &f : files) {
            if (!is_regular_file(f)) {
                fprintf(stderr, "%s : file not found\n", f.c_str());
                usage("short");
                return 1;
            }
            fastq_path_list.push_back(f);
        }
    }

    const bool is_pair = IS_SHORT && IS_PAIRED_END;
    const int dev_mode = !IS_SHORT ? TREW_MODE_LONG : (is_pair ? TREW_MODE_PAIR : TREW_MODE_SHORT);
    Scanner *scanner = scanner_create(cfg, dev_mode);
    FinalFastqData total_result_low, total_result_high;
    for (size_t i = 0; i < fastq_path_list.size() / (is_pair ? 2 : 1); ++i) {  // trew.cpp:413-471
        FinalFastqOutput fo;
        if (is_pair) {
            const std::string a = fastq_path_list[2 * i], b = fastq_path_list[2 * i + 1];
            fo = process_kmer_pair(scanner, cfg, canonical(a).c_str(), canonical(b).c_str(), has_gz_ext(a), has_gz_ext(b));
        } else if (IS_SHORT) {
            const std::string a = fastq_path_list[i];
            fo = process_kmer(scanner, cfg, canonical(a).c_str(), has_gz_ext(a));
        } else {
            const std::string a = fastq_path_list[i];
            fo = process_kmer_long(scanner, cfg, canonical(a).c_str(), has_gz_ext(a));
        }
        for (const auto &kv : fo.high) {  // add_data, kmer.cpp:76-78; trew.cpp:454-467
            FinalData &d = total_result_high[kv.first];
            d.forward += kv.second.forward;
            d.backward += kv.second.backward;
            d.both += kv.second.both;
        }
        for (const auto &kv : fo.low) {
            FinalData &d = total_result_low[kv.first];
            d.forward += kv.second.forward;
            d.backward += kv.second.backward;
            d.both += kv.second.both;
        }
    }
    scanner_destroy(scanner);
    final_process_output(total_result_high, total_result_low, stdout);  // trew.cpp:476
    return 0;
}
