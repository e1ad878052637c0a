#!/bin/bash
# tools/attribute_insts.sh "name1 name2 ..." [bench flags]  -- SQ instruction counters of the exact kernel for A/B builds in
# tools/proflib/<name>/ ("main" = trew_amd/lib): one rocprofv3 --pmc pass each (PMC alone, one stream).  Used for the
# instruction attribution of profiles/r04 (builds with -DTREW_AB_SKIP=n leave out phases of the short driver).
R=${GRAFT_REPO_ROOT:-/root/repo}
NAMES=$1; shift
cd /tmp && export TMPDIR=/tmp
for v in $NAMES; do
  if [ "$v" = main ]; then unset TREW_HIP_LIB; else export TREW_HIP_LIB=$R/tools/proflib/$v/libtrew_hip.so; fi
  OUT=$R/gpurun_out/attr_$v
  rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES --output-format csv -d $OUT -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu --no-other-configs --no-e2e --streams 1 "$@" > $OUT/log.txt 2>&1
  python3 - "$v" $OUT/*/*_counter_collection.csv <<'PY'
import sys, csv, collections
name, path = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(path)):
    if "exact_kernel" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(name, " ".join("%s=%.1fM" % (k.replace("SQ_", ""), sum(v) / len(v) / 1e6) for k, v in sorted(agg.items())))
PY
done
