"""tools/check_big.py N [first_read]: one N-read device batch twice (idempotence) and as five sub-batches; prints table digests.
Run with TREW_HIP_LIB=... to compare A/B builds: equal digests <=> equal tables."""
import hashlib
import sys

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import trew_amd as T  # noqa: E402

n = int(sys.argv[1])
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
L, seed = 150, 20250218
stride = 3 * ((L + 31) // 32)


def digest(tabs):
    h = hashlib.sha256()
    for name in sorted(tabs):
        for key in sorted(tabs[name]):
            h.update(("%s %d %d %d;" % (name, key[0], key[1], tabs[name][key])).encode())
    return h.hexdigest()[:16], sum(sum(v.values()) for v in tabs.values())


with T.TrewHip(mode=T.MODE_SHORT, n_slots=2, max_batch_reads=n, max_batch_words=16, table_log2_slots=22) as t:
    d = t.malloc(n * stride * 4 + 64)
    t.synth_short_device(seed, first, n, L, d)
    for rep in range(3):
        t.reset_tables()
        t.submit(t.device_uniform_batch(d, n, L), rep & 1)
        t.wait(rep & 1)
        print("whole pass", rep, digest(t.collect()), "flagged", t.last_timing(rep & 1)[2])
    t.reset_tables()
    for rep in range(3):  # no reset in between: counts accumulate (what the full-size tests do)
        t.submit(t.device_uniform_batch(d, n, L), rep & 1)
        t.wait(rep & 1)
        print("accumulated", rep + 1, digest(t.collect()), "flagged", t.last_timing(rep & 1)[2], t.debug_counters())
    t.reset_tables()
    q = n // 5
    for i in range(5):
        t.submit(t.device_uniform_batch(d + i * q * stride * 4, q, L), i & 1)
    t.wait(0)
    t.wait(1)
    print("five parts", digest(t.collect()))
