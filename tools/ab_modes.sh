R=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do for v in old main; do
  if [ "$v" = main ]; then unset TREW_HIP_LIB; else export TREW_HIP_LIB=$R/tools/proflib/$v/libtrew_hip.so; fi
  for m in short pair long; do
    out=$(python3 $R/bench.py --mode $m --steps 20 --warmup 3 --no-cpu --no-e2e --no-other-configs --streams 1 2>/dev/null | tail -1)
    echo "$v $m $(echo "$out" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["serial_launch_ms"])')"
  done
done; done
