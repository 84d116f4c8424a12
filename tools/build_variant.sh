#!/bin/bash
# A/B build of the library with extra compiler flags: bash tools/build_variant.sh <suffix> <flags...>  ->  treegp_amd/csrc/libtgp_<suffix>.so
set -e
suffix=$1; shift
cd "$(dirname "$0")/../treegp_amd/csrc"
d=.variant_$suffix; mkdir -p $d
for f in api handoff kbuild chol trsv trsv_big predict kk kk_boot cov dist knn binstat vcorr; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 "$@" -Wno-unused-function -Wno-unused-variable -c $f.hip -o $d/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libtgp_$suffix.so $d/*.o
rm -rf $d
echo built treegp_amd/csrc/libtgp_$suffix.so
