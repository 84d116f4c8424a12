#!/bin/bash
# Development build of the library with in-kernel time stamps in potrf128 (tools/potrf_stamps.py): libtgp_stamps.so
set -e
cd "$(dirname "$0")/../treegp_amd/csrc"
mkdir -p .stamps
for f in api handoff kbuild chol trsv trsv_big predict kk kk_boot cov dist knn binstat vcorr; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DTGP_POTRF_STAMPS -Wno-unused-function -Wno-unused-variable -c $f.hip -o .stamps/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libtgp_stamps.so .stamps/*.o
rm -rf .stamps
echo built treegp_amd/csrc/libtgp_stamps.so
