#!/bin/bash
# tools/ab_libs.sh "name1 name2 ..." [bench args]  -- bench.py (short, no CPU leg) once per A/B build in tools/proflib/<name>/
# ("main" = trew_amd/lib), one stream and two; prints value, ms per step and the kernels' own durations.
R=${GRAFT_REPO_ROOT:-/root/repo}
NAMES=$1; shift
for v in $NAMES; do
  for st in 1 2; do
    if [ "$v" = main ]; then unset TREW_HIP_LIB; else export TREW_HIP_LIB=$R/tools/proflib/$v/libtrew_hip.so; fi
    out=$(python3 $R/bench.py --steps 40 --warmup 5 --no-cpu --no-other-configs --streams $st "$@" 2>/dev/null | tail -1)
    echo "$v streams=$st $(echo "$out" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("value", d["value"], "ms/step", d["ms_per_step"], "serial", d["roofline"]["serial_launch_ms"], "timed", d["roofline"]["avg_launch_ms"], "flagged", d["flagged_reads_per_step"])')"
  done
done
