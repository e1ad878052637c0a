#!/bin/bash
# tools/pair_ab.sh "build ..." : bench.py --mode pair once per A/B build (tools/proflib/<build>/, "main" = trew_amd/lib), one stream and two
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in $1; do
  for st in 1 2; do
    if [ "$v" = main ]; then unset TREW_HIP_LIB; else export TREW_HIP_LIB=$R/tools/proflib/$v/libtrew_hip.so; fi
    out=$(python3 $R/bench.py --mode pair --steps 10 --warmup 2 --no-cpu --no-e2e --streams $st 2>/dev/null | tail -1)
    echo "$v streams=$st $(echo "$out" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms/step", d["ms_per_step"], "serial", d["roofline"]["serial_launch_ms"], "timed", d["roofline"]["avg_launch_ms"])')"
  done
done
