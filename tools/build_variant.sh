#!/bin/bash
# tools/build_variant.sh NAME [extra hipcc flags...]  -- A/B builds of libtrew_hip.so into tools/proflib/NAME/ (git-ignored; travels
# with gpurun).  Run a variant with TREW_HIP_LIB=tools/proflib/NAME/libtrew_hip.so python3 bench.py ...
set -e
NAME=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/tools/proflib/$NAME
mkdir -p $OUT/obj
cd $R/trew_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-pass-failed $*"
/opt/rocm/bin/hipcc $FLAGS -c trew_kernels.hip -o $OUT/obj/trew_kernels.o &
/opt/rocm/bin/hipcc $FLAGS -x hip -c trew_capi.cpp -o $OUT/obj/trew_capi.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libtrew_hip.so $OUT/obj/trew_kernels.o $OUT/obj/trew_capi.o
echo "built $OUT/libtrew_hip.so"
