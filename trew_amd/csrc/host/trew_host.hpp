// trew_host.hpp -- C++ host side of the MI355X-native `trew` binary.
//
// Mirrors the interface of the reference's kmer.h (Chemical118/TREW) for the parts that stay
// on the CPU -- FASTQ/.gz chunk reading, per-file aggregation, CSV output, Putative_TRM ranking --
// and drives the device hot path through the C ABI of include/trew_hip.h.  Same function names
// and argument meaning as the reference where a counterpart exists:
//   process_kmer / process_kmer_pair / process_kmer_long   kmer.h:218-228, kmer.cpp:1266-1476
//   process_output                                          kmer.cpp:1478-1634
//   check_ans_seq                                           kmer.cpp:2549-2569
//   final_process_output / get_score_map                    kmer.cpp:2571-2761
// Errors follow the reference's convention: message on stderr, exit(EXIT_FAILURE).
#pragma once
#include <cstdint>
#include <cstdio>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "../../../include/trew_hip.h"

namespace trew_host {

typedef unsigned __int128 uint128_t;

// the configuration globals of kmer.h:55-63
struct Config {
    int MIN_MER = 5;
    int MAX_MER = 32;
    int NUM_THREAD = 2;
    int SLICE_LENGTH = 150;
    int QUEUE_SIZE = -1;
    double LOW_BASELINE = 0.5;
    double HIGH_BASELINE = 0.8;
    int TABLE_MAX_MER = 12;  // accepted for CLI compatibility; the device path has no tables
    std::vector<int> devices = {0};
    bool stats = false;
    int table_log2_slots = 24;   // device count table: 2^24 slots (256 MiB); emptied into host memory whenever half full
    bool serial_reader = false;  // --serial_reader: plain FASTQ through the reference-shaped single reader as well
    int batch_mib = 32;          // --batch_mib: text per device batch of the block-parallel reader (1..32 MiB)
    bool compat_g1 = false;      // --compat_g1 (short --paired_end): the reference's un-cleared temp_result_left, one consumer, file order
    bool host_pack = false;      // --host_pack: the block-parallel reader packs bases on the CPU (trew_pack_reads) instead of shipping text to the pack kernel
};

// KmerSeq (kmer.h:77) with a total order so that output is deterministic
struct KmerSeq {
    int k;
    uint128_t seq;
    bool operator<(const KmerSeq &o) const { return k != o.k ? k < o.k : seq < o.seq; }
    bool operator==(const KmerSeq &o) const { return k == o.k && seq == o.seq; }
};

// FinalData<int64_t> (kmer.h:65-70)
struct FinalData {
    int64_t forward = 0, backward = 0, both = 0;
};

typedef std::map<KmerSeq, FinalData> FinalFastqData;                  // kmer.h:89 (ordered => deterministic)
typedef std::vector<std::pair<KmerSeq, FinalData>> FinalFastqVector;  // kmer.h:90
struct FinalFastqOutput {                                             // kmer.h:106-109
    FinalFastqVector high, low;
};

typedef std::map<KmerSeq, uint64_t> ResultMap;  // kmer.h:79
struct ResultMapData {                          // kmer.h:81: {forward, backward, both} x {high(first), low(second)}
    ResultMap table[TREW_NUM_TABLES];
};

// ---- primitives (kmer.cpp:39-70, 1815-1892) ----
uint128_t get_rot_seq_128(uint128_t seq, int k);
uint128_t reverse_complement_k(uint128_t seq, int k);  // reverse_complement_128(x) >> 2*(64-k)
int get_dna_count(uint128_t seq, int k);
void int_to_four(char *buffer, uint128_t seq, int n);
bool check_ans_seq(const KmerSeq &seq, int min_mer);

// ---- per-file pipelines: reader on the caller's thread, NUM_THREAD-1 packer threads each
// owning one device slot, then process_output.  file names are printed as given (the CLI passes
// canonical absolute paths, trew.cpp:439-451). ----
struct Scanner;  // device contexts, one per GPU
Scanner *scanner_create(const Config &cfg, int mode);
void scanner_destroy(Scanner *s);

FinalFastqOutput process_kmer(Scanner *s, const Config &cfg, const char *file_name, bool is_gz);
FinalFastqOutput process_kmer_pair(Scanner *s, const Config &cfg, const char *file_name1, const char *file_name2,
                                   bool is_gz1, bool is_gz2);
FinalFastqOutput process_kmer_long(Scanner *s, const Config &cfg, const char *file_name, bool is_gz);

// prints the >H: / >L: sections and returns the folded rows (kmer.cpp:1478-1634)
FinalFastqOutput process_output(const char *file_name, const ResultMapData &result, int min_mer, FILE *out);
// prints >Putative_TRM (kmer.cpp:2571-2691)
void final_process_output(FinalFastqData &total_result_high, FinalFastqData &total_result_low, FILE *out);
std::map<KmerSeq, uint32_t> get_score_map(const FinalFastqData &total_result);

struct RunStats {
    uint64_t reads = 0, bases = 0;
    double seconds = 0;
};
const RunStats &last_stats(const Scanner *s);

}  // namespace trew_host
