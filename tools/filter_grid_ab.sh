#!/bin/bash
# A/B of the prefilter's block size and grid size: tools/proflib/f64 and f256 are builds with
# -DTREW_FILTER_THREADS=64 / 256; TREW_FILTER_BLOCKS_PER_CU overrides the occupancy-sized grid.
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in f64 f256; do
  for bpc in 0 16 20 24 28 32 4 5 6 7 8 10 12; do
    if [ "$v" = f256 ] && [ $bpc -gt 12 ]; then continue; fi
    if [ "$v" = f64 ] && [ $bpc -gt 0 ] && [ $bpc -lt 12 ]; then continue; fi
    if [ $bpc -eq 0 ]; then unset TREW_FILTER_BLOCKS_PER_CU; else export TREW_FILTER_BLOCKS_PER_CU=$bpc; fi
    for st in 1 2; do
      out=$(TREW_HIP_LIB=$R/tools/proflib/$v/libtrew_hip.so python3 $R/bench.py --steps 30 --warmup 3 --no-cpu --no-other-configs --streams $st 2>/dev/null | tail -1)
      echo "$v bpc=$bpc streams=$st $(echo "$out" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("value", d["value"], "ms/step", d["ms_per_step"], "serial", d["roofline"]["serial_launch_ms"], "timed", d["roofline"]["avg_launch_ms"])')"
    done
  done
done
