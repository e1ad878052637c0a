// output.cpp -- per-file aggregation, CSV sections and Putative_TRM ranking of the `trew` host.
//
// Restates process_output (kmer.cpp:1478-1634), check_ans_seq (2549-2569),
// final_process_output (2571-2691) and get_score_map (2693-2761).  The reference sorts
// hash-map iteration order with std::sort, so its row order among ties -- and top-4
// membership at a tie boundary -- changes from run to run.  Here every tie
// is broken by (k ascending, sequence ascending): output is deterministic, and equal to the
// reference's as a sorted row set.
#include <algorithm>
#include <cinttypes>
#include <set>

#include "trew_host.hpp"

namespace trew_host {

static const char trans_arr[4] = {'T', 'G', 'C', 'A'};  // kmer.cpp:7

uint128_t get_rot_seq_128(uint128_t seq, int k) {  // kmer.cpp:1825-1833
    uint128_t tmp = seq, ans = seq;
    for (int i = 0; i < k - 1; i++) {
        tmp = ((tmp & 0x3) << (2 * (k - 1))) + (tmp >> 2);
        if (tmp < ans) ans = tmp;
    }
    return ans;
}

uint128_t reverse_complement_k(uint128_t seq, int k) {  // kmer.cpp:62-70 followed by >> 2*(64-k)
    uint128_t r = 0;
    for (int i = 0; i < k; i++) {
        r = (r << 2) | (uint128_t) (3 - (int) (seq & 3));
        seq >>= 2;
    }
    return r;
}

int get_dna_count(uint128_t seq, int k) {  // kmer.cpp:1869-1884
    int seen[4] = {0, 0, 0, 0};
    for (int i = 0; i < k; i++, seq >>= 2) seen[(int) (seq & 3)] = 1;
    return seen[0] + seen[1] + seen[2] + seen[3];
}

void int_to_four(char *buffer, uint128_t seq, int n) {  // kmer.cpp:1886-1892
    for (int i = 0; i < n; i++) {
        buffer[n - 1 - i] = trans_arr[(int) (seq & 0x3)];
        seq >>= 2;
    }
    buffer[n] = '\0';
}

// kmer.cpp:2549-2569: reject a k-mer whose length-j windows (3 <= j < MIN_MER) all fall into
// one rotation class, i.e. a pure repetition of a unit shorter than MIN_MER
bool check_ans_seq(const KmerSeq &seq, int min_mer) {
    for (int j = 3; j < min_mer; j++) {
        uint128_t num = seq.seq, bef = 0;
        const uint128_t mask = (((uint128_t) 1) << (2 * j)) - 1;
        int i;
        for (i = 0; i < seq.k - j + 1; i++) {
            const uint128_t t = get_rot_seq_128(num & mask, j);
            if (i > 0 && t != bef) break;
            bef = t;
            num >>= 2;
        }
        if (i == seq.k - j + 1) return false;
    }
    return true;
}

static inline KmerSeq rot_rc(const KmerSeq &s) { return KmerSeq{s.k, get_rot_seq_128(reverse_complement_k(s.seq, s.k), s.k)}; }

// The reference's maps hold uint32_t counts (ResultMap, kmer.h:79): the thread merge and the
// backward -> forward fold add modulo 2^32 before the values are widened to FinalData<int64_t>
// (kmer.cpp:1486-1549).  The device accumulates in 64 bits; addition modulo 2^32 is associative, so
// truncating once here, at the per-file merge point, gives the reference's value even past a wrap.
static inline int64_t wrap32(uint64_t v) { return (int64_t) (uint32_t) v; }

// one baseline of process_output, kmer.cpp:1518-1579
static FinalFastqData fold(const ResultMap &forward_in, const ResultMap &backward, const ResultMap &both) {
    ResultMap forward = forward_in;
    for (const auto &kv : backward) forward[rot_rc(kv.first)] += kv.second;  // kmer.cpp:1518-1523
    FinalFastqData final_result;
    for (const auto &kv : forward) {  // kmer.cpp:1526-1540
        const KmerSeq t = rot_rc(kv.first);
        const KmerSeq kseq = t.seq < kv.first.seq ? t : kv.first;
        auto it = final_result.find(kseq);
        if (it == final_result.end()) {
            FinalData d;
            d.backward = (t.seq == kv.first.seq) ? -1 : 0;  // palindromic class
            it = final_result.emplace(kseq, d).first;
        }
        if (kseq.seq == kv.first.seq)
            it->second.forward = wrap32(kv.second);
        else
            it->second.backward = wrap32(kv.second);
    }
    for (const auto &kv : both) {  // kmer.cpp:1541-1549
        auto it = final_result.find(kv.first);
        if (it != final_result.end()) {
            it->second.both = wrap32(kv.second);
        } else {
            const KmerSeq t = rot_rc(kv.first);
            FinalData d;
            d.backward = (t.seq == kv.first.seq) ? -1 : 0;
            d.both = wrap32(kv.second);
            final_result.emplace(kv.first, d);
        }
    }
    return final_result;
}

static FinalFastqVector filter_sort(const FinalFastqData &data, int min_mer) {
    FinalFastqVector v;
    for (const auto &kv : data)
        if (check_ans_seq(kv.first, min_mer)) v.emplace_back(kv.first, kv.second);  // kmer.cpp:1585-1590
    std::sort(v.begin(), v.end(), [](const auto &a, const auto &b) {  // kmer.cpp:1592-1598 + total order
        if (a.second.forward != b.second.forward) return a.second.forward > b.second.forward;
        if (a.second.both != b.second.both) return a.second.both > b.second.both;
        return a.first < b.first;
    });
    return v;
}

static void print_rows(const FinalFastqVector &v, FILE *out) {
    char buffer[65];
    for (const auto &kv : v) {
        const FinalData &d = kv.second;
        if (d.forward + d.backward + d.both >= 10) {  // ABS_MIN_PRINT_COUNT, kmer.cpp:1617
            int_to_four(buffer, kv.first.seq, kv.first.k);
            fprintf(out, "%d,%s,%" PRId64 ",%" PRId64 ",%" PRId64 ",%c\n", kv.first.k, buffer,
                    std::max(d.forward, d.backward), std::min(d.forward, d.backward), d.both,
                    d.forward > d.backward ? '+' : (d.forward < d.backward ? '-' : '?'));
        }
    }
}

FinalFastqOutput process_output(const char *file_name, const ResultMapData &r, int min_mer, FILE *out) {
    FinalFastqOutput o;
    o.high = filter_sort(fold(r.table[TREW_TABLE_FORWARD_HIGH], r.table[TREW_TABLE_BACKWARD_HIGH], r.table[TREW_TABLE_BOTH_HIGH]), min_mer);
    o.low = filter_sort(fold(r.table[TREW_TABLE_FORWARD_LOW], r.table[TREW_TABLE_BACKWARD_LOW], r.table[TREW_TABLE_BOTH_LOW]), min_mer);
    fprintf(out, ">H:%s\n", file_name);  // kmer.cpp:1615-1622
    print_rows(o.high, out);
    fprintf(out, ">L:%s\n", file_name);  // kmer.cpp:1624-1631
    print_rows(o.low, out);
    return o;
}

// get_score_map, kmer.cpp:2693-2761
std::map<KmerSeq, uint32_t> get_score_map(const FinalFastqData &total_result) {
    FinalFastqVector vec;
    for (const auto &kv : total_result) {
        const FinalData &v = kv.second;
        if (v.forward + v.backward + v.both >= 10) {
            if (v.backward > v.forward) {
                FinalData s;
                s.forward = v.backward;
                s.backward = v.forward;
                s.both = v.both;
                vec.emplace_back(kv.first, s);
            } else {
                vec.emplace_back(kv.first, v);
            }
        }
    }
    std::map<KmerSeq, FinalData> ratio_result;
    std::map<KmerSeq, uint32_t> score;
    std::sort(vec.begin(), vec.end(), [](const auto &a, const auto &b) {
        if (a.second.forward != b.second.forward) return a.second.forward > b.second.forward;
        return a.first < b.first;
    });
    int cnt = 0;
    for (const auto &kv : vec) {
        if (kv.second.forward == 0 || cnt >= 20) break;  // NUM_RAT_CAND
        if (kv.second.backward >= 0) {
            cnt += 1;
            ratio_result[kv.first] = kv.second;
        }
    }
    for (size_t i = 0; i < std::min<size_t>(4, vec.size()); i++) {  // NUM_FOR_MAX_COUNT
        if (vec[i].second.forward == 0) break;
        score[vec[i].first] += 1;
    }
    std::sort(vec.begin(), vec.end(), [](const auto &a, const auto &b) {
        const int64_t ta = a.second.forward + a.second.backward + a.second.both;
        const int64_t tb = b.second.forward + b.second.backward + b.second.both;
        if (ta != tb) return ta > tb;
        return a.first < b.first;
    });
    cnt = 0;
    for (const auto &kv : vec) {
        if (cnt >= 20) break;
        if (kv.second.forward > 0 && kv.second.backward >= 0) {
            cnt += 1;
            ratio_result[kv.first] = kv.second;
        }
    }
    for (size_t i = 0; i < std::min<size_t>(4, vec.size()); i++) score[vec[i].first] += 1;  // NUM_TOT_MAX_COUNT
    FinalFastqVector rvec(ratio_result.begin(), ratio_result.end());
    std::sort(rvec.begin(), rvec.end(), [](const auto &a, const auto &b) {
        const double ra = (double) a.second.backward / (double) a.second.forward;
        const double rb = (double) b.second.backward / (double) b.second.forward;
        if (ra != rb) return ra < rb;
        return a.first < b.first;
    });
    for (size_t i = 0; i < std::min<size_t>(4, rvec.size()); i++) score[rvec[i].first] += 1;  // NUM_RAT_MAX_COUNT
    return score;
}

static inline int sign_of(int64_t v) { return v > 0 ? 1 : (v < 0 ? -1 : 0); }

enum class StrandRule { none, low, high, agree, conflict };
// [sign in the low table + 1][sign in the high table + 1]
static const StrandRule kStrandRule[3][3] = {
    /* low says -  */ {StrandRule::agree, StrandRule::low, StrandRule::conflict},
    /* low says ?  */ {StrandRule::high, StrandRule::none, StrandRule::high},
    /* low says +  */ {StrandRule::conflict, StrandRule::low, StrandRule::agree},
};

// Two tables disagree about the strand: trust the one whose minority/majority ratio is smaller (compared by cross
// multiplication, as the reference does); equal ratios go to the table with more reads, then to the high table.
static bool purer_table_is_low(const FinalData &low, const FinalData &high) {
    const int64_t l_major = std::max(low.forward, low.backward), l_minor = std::min(low.forward, low.backward);
    const int64_t h_major = std::max(high.forward, high.backward), h_minor = std::min(high.forward, high.backward);
    const int64_t lhs = l_minor * h_major, rhs = h_minor * l_major;
    if (lhs != rhs) return lhs < rhs;
    return l_major + l_minor > h_major + h_minor;
}

void final_process_output(FinalFastqData &total_high, FinalFastqData &total_low, FILE *out) {
    bool max_cnt_check = false;
    for (const auto &kv : total_high)
        if (kv.second.forward + kv.second.backward + kv.second.both >= 20) max_cnt_check = true;  // ABS_MIN_ANS_COUNT
    for (const auto &kv : total_low)
        if (kv.second.forward + kv.second.backward + kv.second.both >= 20) max_cnt_check = true;
    fprintf(out, ">Putative_TRM\n");
    if (!max_cnt_check) {
        fprintf(out, "NO_PUTATIVE_TRM,-1\n");
        return;
    }
    std::map<KmerSeq, uint32_t> score = get_score_map(total_low);
    for (const auto &kv : get_score_map(total_high)) score[kv.first] += kv.second;
    struct Row {
        KmerSeq k;
        uint32_t score;
        int dna;
        int dir;
    };
    std::vector<Row> rows;
    for (const auto &kv : score) {
        FinalData low_result, high_result;
        auto il = total_low.find(kv.first);
        if (il != total_low.end()) low_result = il->second;
        auto ih = total_high.find(kv.first);
        if (ih != total_high.end()) high_result = ih->second;
        // Strand of the motif from the two baselines' tables (kmer.cpp:2603-2650), as a decision table over the
        // sign of forward - backward in each: agreement earns a bonus point, a one-sided verdict is taken as it
        // is, and two opposite verdicts go to the table whose minority strand is relatively smaller.
        const int low_dir = sign_of(low_result.forward - low_result.backward), high_dir = sign_of(high_result.forward - high_result.backward);
        int bonus = 0, final_dir = 0;
        switch (kStrandRule[low_dir + 1][high_dir + 1]) {
        case StrandRule::none: break;
        case StrandRule::low: final_dir = low_dir; break;
        case StrandRule::high: final_dir = high_dir; break;
        case StrandRule::agree:
            final_dir = low_dir;
            bonus += 1;
            break;
        case StrandRule::conflict: final_dir = purer_table_is_low(low_result, high_result) ? low_dir : high_dir; break;
        }
        const int dna_cnt = get_dna_count(kv.first.seq, kv.first.k);
        if (dna_cnt > 2) bonus += 1;
        rows.push_back(Row{kv.first, kv.second + (uint32_t) bonus, dna_cnt, final_dir});
    }
    std::sort(rows.begin(), rows.end(), [](const Row &a, const Row &b) {  // kmer.cpp:2665-2673 (+ sequence)
        if (a.score != b.score) return a.score > b.score;
        if (a.dna != b.dna) return a.dna > b.dna;
        return a.k < b.k;
    });
    char buffer[65];
    for (size_t i = 0; i < std::min<size_t>(10, rows.size()); i++) {  // ABS_MAX_ANS_NUM
        int_to_four(buffer, rows[i].k.seq, rows[i].k.k);
        fprintf(out, "%d,%s,%" PRIu32 ",%c\n", rows[i].k.k, buffer, rows[i].score, rows[i].dir == 1 ? '+' : (rows[i].dir == -1 ? '-' : '?'));
    }
}

}  // namespace trew_host
