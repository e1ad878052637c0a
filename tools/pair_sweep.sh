R=${GRAFT_REPO_ROOT:-/root/repo}
run() { out=$(python3 $R/bench.py --mode pair --steps 10 --warmup 2 --no-cpu --no-e2e "$@" 2>/dev/null | tail -1); echo "F=$TREW_FILTER_BLOCKS_PER_CU E=$TREW_EXACT_WAVES_PER_CU $* $(echo "$out" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms/step", d["ms_per_step"], "timed", d["roofline"]["avg_launch_ms"], "serial", d["roofline"]["serial_launch_ms"])')"; }
run --streams 1
run --streams 2
for f in 3 4 6; do for e in 8 12 20; do export TREW_FILTER_BLOCKS_PER_CU=$f TREW_EXACT_WAVES_PER_CU=$e; run --streams 2; done; done
