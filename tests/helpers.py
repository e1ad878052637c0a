"""Shared generators of test reads (deterministic)."""
import random


def mutate(s, rnd, p_sub=0.0, p_n=0.0):
    out = []
    for c in s:
        x = rnd.random()
        if x < p_n:
            out.append("N")
        elif x < p_n + p_sub:
            out.append(rnd.choice("ACGT"))
        else:
            out.append(c)
    return "".join(out)


def periodic(unit, n, phase=0):
    return (unit * (n // len(unit) + 3))[phase:phase + n]


def mixed_segments(seed, count, lengths):
    """random / periodic / noisy / with-N segments (SURVEY section 7's probe mix)."""
    rnd = random.Random(seed)
    out = []
    for i in range(count):
        n = rnd.choice(lengths)
        kind = rnd.random()
        if kind < 0.25:
            s = "".join(rnd.choice("ACGT") for _ in range(n))
        elif kind < 0.5:
            unit = "".join(rnd.choice("ACGT") for _ in range(rnd.randint(1, 36)))
            s = periodic(unit, n, rnd.randint(0, 7))
        elif kind < 0.8:
            unit = "".join(rnd.choice("ACGT") for _ in range(rnd.randint(2, 33)))
            s = mutate(periodic(unit, n, rnd.randint(0, 7)), rnd, p_sub=rnd.choice([0.01, 0.03, 0.08, 0.15]))
        elif kind < 0.9:
            unit = "".join(rnd.choice("ACGT") for _ in range(rnd.randint(2, 20)))
            s = mutate(periodic(unit, n), rnd, p_sub=0.02, p_n=rnd.choice([0.01, 0.05]))
        else:
            # two different repeats glued together, low-complexity 2-letter words
            u1 = "".join(rnd.choice("AT") for _ in range(rnd.randint(2, 9)))
            u2 = "".join(rnd.choice("ACGT") for _ in range(rnd.randint(2, 12)))
            cut = rnd.randint(0, n)
            s = (periodic(u1, cut) + periodic(u2, n - cut))[:n]
        if rnd.random() < 0.1:
            s = s.lower()
        out.append(s.encode())
    return out


EDGE_LENGTHS = [9, 10, 11, 19, 20, 21, 39, 40, 63, 64, 65, 96, 127, 128, 129, 150, 151, 190, 191, 192, 246, 250, 300, 999, 1000]


def edge_reads(seed=3):
    """Reads exercising the geometry thresholds of buffer_task (kmer.cpp:115-171)."""
    rnd = random.Random(seed)
    out = []
    units = ["TTAGGG", "CCCTAA", "A", "AT", "ACG", "TTAGG", "TTTAGGG", "TTAGGGTTAGGC", "ACGTACGTAC",
             "TTGCATCACACCCTCGCCG", "TTTTGCCCTCATCACACCCTCGCCTCCTTCGC", "AATT", "GATC", "AACCGGTT"]
    for n in EDGE_LENGTHS:
        for u in units:
            out.append(periodic(u, n, rnd.randint(0, 5)).encode())
            out.append(mutate(periodic(u, n), rnd, p_sub=0.02).encode())
        # half repeat / half random (junction), both orientations
        for u in units[:6]:
            h = n // 2
            rand = "".join(rnd.choice("ACGT") for _ in range(n - h))
            out.append((periodic(u, h) + rand).encode())
            out.append((rand + periodic(u, h)).encode())
            # different repeats in each half
            out.append((periodic(u, h) + periodic("GGGTTA", n - h)).encode())
        out.append(("N" * n).encode())
        out.append(mutate(periodic("TTAGGG", n), rnd, p_n=0.1).encode())
        out.append(("acgtn" * n)[:n].encode())
    out.append(b"")
    out.append(b"ACGT")
    return out
