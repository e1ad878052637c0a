"""Oracle for the per-file aggregation / CSV / Putative_TRM stage -- TEST INFRASTRUCTURE ONLY.

Pure-Python restatement (tables are small) of
  process_output        kmer.cpp:1478-1634
  check_ans_seq         kmer.cpp:2549-2569  (via the C oracle)
  final_process_output  kmer.cpp:2571-2691
  get_score_map         kmer.cpp:2693-2761

The reference sorts hash-map iteration order with std::sort, so its row order
among ties (and top-4 membership at a tie boundary) is nondeterministic
(SURVEY G2, G3).  This oracle -- and the product -- break every tie by
(k ascending, word ascending); compare against the reference as sorted row
sets only.
"""
from __future__ import annotations

from .pyoracle import check_ans_seq, int_to_four, revcomp, rot_seq

ABS_MIN_PRINT_COUNT = 10  # kmer.h:15
ABS_MIN_ANS_COUNT = 20  # kmer.h:16
ABS_MAX_ANS_NUM = 10  # kmer.h:12
NUM_FOR_MAX_COUNT = 4  # kmer.h:18-21
NUM_TOT_MAX_COUNT = 4
NUM_RAT_MAX_COUNT = 4
NUM_RAT_CAND = 20


def _rot_rc(k, w):
    return rot_seq(revcomp(w, k), k)


_U32 = 0xFFFFFFFF  # ResultMap values are uint32_t (kmer.h:79): every += up to here wraps modulo 2^32


def _fold_one(forward, backward, both, min_mer):
    """One baseline of process_output: kmer.cpp:1518-1579 + filter 1585-1605.
    Returns {(k, word): [forward, backward, both]} (backward == -1 marks a
    palindromic class, kmer.cpp:1531)."""
    fwd = {key: cnt & _U32 for key, cnt in forward.items()}
    both = {key: cnt & _U32 for key, cnt in both.items()}
    for (k, w), cnt in backward.items():  # kmer.cpp:1518-1523
        key = (k, _rot_rc(k, w))
        fwd[key] = (fwd.get(key, 0) + cnt) & _U32
    final = {}
    for (k, w), cnt in fwd.items():  # kmer.cpp:1526-1540
        t = _rot_rc(k, w)
        kseq = min(t, w)
        if (k, kseq) not in final:
            final[(k, kseq)] = [0, -1 if t == w else 0, 0]
        if kseq == w:
            final[(k, kseq)][0] = cnt
        else:
            final[(k, kseq)][1] = cnt
    for (k, w), cnt in both.items():  # kmer.cpp:1541-1549
        t = _rot_rc(k, w)
        if (k, w) in final:
            final[(k, w)][2] = cnt
        else:
            final[(k, w)] = [0, -1 if t == w else 0, cnt]
    return {key: v for key, v in final.items() if check_ans_seq(key[1], key[0], min_mer)}


def fold_tables(tables, min_mer):
    """process_output without printing: six tables -> (high, low) folded maps."""
    high = _fold_one(tables["forward_high"], tables["backward_high"], tables["both_high"], min_mer)
    low = _fold_one(tables["forward_low"], tables["backward_low"], tables["both_low"], min_mer)
    return high, low


def _sorted_rows(final):
    # kmer.cpp:1592-1613: forward desc, then both desc; ties by (k, word) asc (documented total order)
    return sorted(final.items(), key=lambda kv: (-kv[1][0], -kv[1][2], kv[0][0], kv[0][1]))


def format_sections(file_name, high, low):
    """The >H: / >L: sections, kmer.cpp:1615-1631."""
    out = [">H:%s" % file_name]
    for name, final in ((None, high), (">L:%s" % file_name, low)):
        if name:
            out.append(name)
        for (k, w), (f, b, bo) in _sorted_rows(final):
            if f + b + bo >= ABS_MIN_PRINT_COUNT:
                sign = "+" if f > b else ("-" if f < b else "?")
                out.append("%d,%s,%d,%d,%d,%s" % (k, int_to_four(w, k), max(f, b), min(f, b), bo, sign))
    return out


def _score_map(total):
    """get_score_map, kmer.cpp:2693-2761."""
    vec = []
    for key, (f, b, bo) in total.items():
        if f + b + bo >= ABS_MIN_PRINT_COUNT:
            vec.append((key, (b, f, bo) if b > f else (f, b, bo)))
    tie = lambda kv: (kv[0][0], kv[0][1])  # noqa: E731
    ratio = {}
    score = {}
    vec.sort(key=lambda kv: (-kv[1][0],) + tie(kv))
    cnt = 0
    for key, v in vec:
        if v[0] == 0 or cnt >= NUM_RAT_CAND:
            break
        if v[1] >= 0:
            cnt += 1
            ratio[key] = v
    for i in range(min(NUM_FOR_MAX_COUNT, len(vec))):
        if vec[i][1][0] == 0:
            break
        score[vec[i][0]] = score.get(vec[i][0], 0) + 1
    vec.sort(key=lambda kv: (-(kv[1][0] + kv[1][1] + kv[1][2]),) + tie(kv))
    cnt = 0
    for key, v in vec:
        if cnt >= NUM_RAT_CAND:
            break
        if v[0] > 0 and v[1] >= 0:
            cnt += 1
            ratio[key] = v
    for i in range(min(NUM_TOT_MAX_COUNT, len(vec))):
        score[vec[i][0]] = score.get(vec[i][0], 0) + 1
    rvec = sorted(ratio.items(), key=lambda kv: (float(kv[1][1]) / float(kv[1][0]),) + tie(kv))
    for i in range(min(NUM_RAT_MAX_COUNT, len(rvec))):
        score[rvec[i][0]] = score.get(rvec[i][0], 0) + 1
    return score


def _dna_count(w, k):
    return len({(w >> (2 * i)) & 3 for i in range(k)})


def putative_trm(total_high, total_low):
    """final_process_output, kmer.cpp:2571-2691.  total_* are the cross-file sums
    (trew.cpp:454-467) of the folded maps.  Returns the printed lines."""
    out = [">Putative_TRM"]
    chk = any(f + b + bo >= ABS_MIN_ANS_COUNT for f, b, bo in total_high.values()) or any(
        f + b + bo >= ABS_MIN_ANS_COUNT for f, b, bo in total_low.values()
    )
    if not chk:
        out.append("NO_PUTATIVE_TRM,-1")
        return out
    score = _score_map(total_low)
    for key, v in _score_map(total_high).items():
        score[key] = score.get(key, 0) + v
    rows = []
    for key, v in score.items():
        lf, lb, _ = total_low.get(key, (0, 0, 0))
        hf, hb, _ = total_high.get(key, (0, 0, 0))
        bonus = 0
        high_dir = 1 if hf > hb else (-1 if hf < hb else 0)
        low_dir = 1 if lf > lb else (-1 if lf < lb else 0)
        if low_dir != 0 and low_dir == high_dir:
            bonus += 1
            final_dir = low_dir
        elif low_dir == 0 and high_dir != 0:
            final_dir = high_dir
        elif low_dir != 0 and high_dir == 0:
            final_dir = low_dir
        elif low_dir != high_dir and (lf > 0 or lb > 0 or hf > 0 or hb > 0):
            if lf < lb:
                lf, lb = lb, lf
            if hf < hb:
                hf, hb = hb, hf
            if lb * hf == hb * lf:
                final_dir = low_dir if lf + lb > hf + hb else high_dir
            elif lb * hf < hb * lf:
                final_dir = low_dir
            else:
                final_dir = high_dir
        else:
            final_dir = 0
        dna = _dna_count(key[1], key[0])
        if dna > 2:
            bonus += 1
        rows.append((key, v + bonus, dna, final_dir))
    rows.sort(key=lambda r: (-r[1], -r[2], r[0][0], r[0][1]))  # kmer.cpp:2665-2673 (+ word asc)
    for key, sc, _, d in rows[:ABS_MAX_ANS_NUM]:
        sign = "+" if d == 1 else ("-" if d == -1 else "?")
        out.append("%d,%s,%d,%s" % (key[0], int_to_four(key[1], key[0]), sc, sign))
    return out


def add_totals(total, final):
    """Cross-file accumulation, trew.cpp:454-467 (add_data, kmer.cpp:76-78)."""
    for key, v in final.items():
        if key in total:
            t = total[key]
            total[key] = [t[0] + v[0], t[1] + v[1], t[2] + v[2]]
        else:
            total[key] = list(v)
    return total
