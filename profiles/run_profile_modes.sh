#!/bin/bash
# rocprofv3 kernel statistics for the non-headline configs (pair, long); run through gpurun:
#   profiles/run_profile_modes.sh <tag>
set -e
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_modes_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pair -- python3 $R/bench.py --mode pair --steps 10 --warmup 2 --no-cpu --no-other-configs --no-e2e > $OUT/pair.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/long -- python3 $R/bench.py --mode long --steps 10 --warmup 2 --no-cpu --no-other-configs --no-e2e > $OUT/long.log 2>&1
find $OUT -name "*_kernel_stats.csv"
