#!/bin/bash
# tools/ab_coresident_builds.sh "build ..." "F ..." "E ..." [bench args]: tools/ab_coresident.sh's sweep (prefilter blocks per CU x
# exact waves per CU, two streams) for several A/B builds of the library in tools/proflib/<build>/ ("main" = trew_amd/lib).
R=${GRAFT_REPO_ROOT:-/root/repo}
BUILDS=$1; export FS=$2; export ES=$3; shift 3
for b in $BUILDS; do
  if [ "$b" = main ]; then unset TREW_HIP_LIB; else export TREW_HIP_LIB=$R/tools/proflib/$b/libtrew_hip.so; fi
  echo "== build $b"
  bash $R/tools/ab_coresident.sh "$@"
done
