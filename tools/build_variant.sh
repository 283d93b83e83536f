#!/bin/bash
# Build a variant of libkinectpx.so with ONE translation unit (default kpx_icp; KPX_VARIANT_TU=kpx_knn ... for another) compiled with extra
# -D flags (for same-box A/B runs through KPX_LIBRARY):   tools/build_variant.sh NAME -DKPX_ICP_WPE=3 -DKPX_MUL_BATCH=16   ->  kinectpy_amd/libkinectpx_NAME.so
set -e
cd "$(dirname "$0")/../kinectpy_amd/csrc"
name=$1; shift
tu=${KPX_VARIANT_TU:-kpx_icp}
make -s >/dev/null
mkdir -p build/var_$name
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fvisibility=hidden -Wall -Wno-unused-function -Wno-unused-result \
    "$@" -c $tu.hip -o build/var_$name/$tu.o
objs=$(ls build/*.o | grep -v $tu.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libkinectpx_$name.so $objs build/var_$name/$tu.o
echo "built kinectpy_amd/libkinectpx_$name.so ($*)"
