#!/bin/bash
# tools/ab_coresident.sh [bench args]: the step with the two kernels' grids limited so that both are resident at once
# (TREW_FILTER_BLOCKS_PER_CU x TREW_EXACT_WAVES_PER_CU, experiments only), two streams.
R=${GRAFT_REPO_ROOT:-/root/repo}
run() {
  out=$(python3 $R/bench.py --steps 40 --warmup 5 --no-cpu --no-other-configs --no-e2e --streams 2 "$@" 2>/dev/null | tail -1)
  echo "F=$TREW_FILTER_BLOCKS_PER_CU E=$TREW_EXACT_WAVES_PER_CU $(echo "$out" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("value", d["value"], "ms/step", d["ms_per_step"], "timed", d["roofline"]["avg_launch_ms"])')"
}
unset TREW_FILTER_BLOCKS_PER_CU TREW_EXACT_WAVES_PER_CU; run "$@"
for f in $FS; do for e in $ES; do export TREW_FILTER_BLOCKS_PER_CU=$f TREW_EXACT_WAVES_PER_CU=$e; run "$@"; done; done
