#!/bin/bash
# A/B of the exact kernel's occupancy target: tools/proflib/w5, w4 are builds with __launch_bounds__(64, 5) / (64, 4)
# (sed on trew_kernels.hip, see profiles/r02/README.md); the in-tree library is what the Makefile builds.
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in tree w5 w4; do
  lib=$R/tools/proflib/$v/libtrew_hip.so
  [ "$v" = tree ] && lib=$R/trew_amd/lib/libtrew_hip.so
  for st in 1 2; do
    for rep in 1 2; do
      out=$(TREW_HIP_LIB=$lib python3 $R/bench.py --steps 40 --warmup 5 --no-cpu --no-other-configs --streams $st 2>/dev/null | tail -1)
      echo "$v streams=$st $(echo "$out" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("value", d["value"], "ms/step", d["ms_per_step"], "timed", d["roofline"]["avg_launch_ms"])')"
    done
  done
done
for v in tree w5; do
  lib=$R/tools/proflib/$v/libtrew_hip.so
  [ "$v" = tree ] && lib=$R/trew_amd/lib/libtrew_hip.so
  for mode in pair long; do
    out=$(TREW_HIP_LIB=$lib python3 $R/bench.py --mode $mode --steps 10 --warmup 2 --no-cpu 2>/dev/null | tail -1)
    echo "$v mode=$mode $(echo "$out" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("value", d["value"], "ms/step", d["ms_per_step"], "timed", d["roofline"]["avg_launch_ms"], "serial", d["roofline"]["serial_launch_ms"])')"
  done
done
