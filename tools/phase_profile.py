"""Where the exact kernel's wave time goes: run the bench workload on a library built with
-DTREW_PHASE_PROFILE (make -C trew_amd/csrc OUT=../../tools/proflib BIN=../../tools/proflib CXXFLAGS='-O3 -std=c++17 -fPIC -DTREW_PHASE_PROFILE')
and print the s_memtime totals per phase.  Usage:
  TREW_HIP_LIB=tools/proflib/libtrew_hip.so python tools/phase_profile.py [short|pair|long]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import trew_amd as T
from trew_amd import capi

mode = sys.argv[1] if len(sys.argv) > 1 else "short"
lib = capi.load()
n, L = 10_000_000, 150
to_free = []
if mode == "long":
    n = 1_000_000
    t = T.TrewHip(mode=T.MODE_LONG, n_slots=1, max_batch_words=16, max_batch_reads=n, table_log2_slots=20)
    b, to_free, _ = t.synth_long_device(20250218, 0, n)
elif mode == "pair":
    t = T.TrewHip(mode=T.MODE_PAIR, n_slots=1, max_batch_words=16, max_batch_reads=2 * n, table_log2_slots=20)
    d = t.malloc(2 * n * 60 + 64)
    t.synth_pair_device(20250218, 0, n, L, d)
    b = t.device_uniform_batch(d, 2 * n, L)
else:
    t = T.TrewHip(mode=T.MODE_SHORT, n_slots=1, max_batch_words=16, max_batch_reads=n, table_log2_slots=20)
    d = t.malloc(n * 60 + 64)
    t.synth_short_device(20250218, 0, n, L, d)
    b = t.device_uniform_batch(d, n, L)
out = (C.c_ulonglong * 32)()
for rnd in range(3):
    t.submit(b, 0)
    t.wait(0)
    lib.trew_debug_phases(out, 1)
names = ["total", "stage", "loadseg", "bounds", "decide", "runs", "windows", "record_eval", "emit", "flush", "evalk_A", "pair_stage", "pair_flush", "pair_fwd", "pair_bwd", "pair_whole"]
tot = out[0]
a, e, fl = t.last_timing(0)
print("filter %.4f ms exact %.4f ms flagged %d" % (a, e, fl))
for i, nm in enumerate(names):
    print("%-12s %14d  %5.1f %%" % (nm, out[i], 100.0 * out[i] / tot))
if mode == "long":
    print("long walks: max steps per read %d; reads with <=4 / <=16 / <=64 / <=128 / <=256 / more steps: %s; steps in all %d" % (out[24], [int(out[i]) for i in range(25, 31)], out[31]))
cn = ["reads", "runs_calls", "windows_calls", "records", "runs_total", "k5_calls"]
for i, nm in enumerate(cn):
    print("%-14s %10d" % (nm, out[16 + i]))
