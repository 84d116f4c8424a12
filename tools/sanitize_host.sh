#!/bin/bash
# CPU-side sanitizer run of the native HOST logic of libtgp.so ("race detection / sanitizers"): every translation
# unit is rebuilt with AddressSanitizer + UBSan on the host side only (-fno-gpu-sanitize: GPU ASan is not available on this
# pool; the device code is compiled as usual), linked with tests/sanitize/host_logic_driver.cpp and run WITHOUT a GPU: Morton
# keys, counting sorts of bootstrap rows, packed-layout helpers, the XCD-aware tile maps, the no-device error path.
# usage: tools/sanitize_host.sh [build dir]      exit code 0 = clean
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
B=${1:-/tmp/tgp_sanitize}
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
mkdir -p "$B"
FLAGS="--offload-arch=gfx950 -std=c++17 -O1 -g -fPIC -fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer -w"
OBJS=""
N=0
for f in api handoff kk chol kbuild trsv trsv_big predict kk_boot cov dist knn binstat vcorr; do
  $HIPCC $FLAGS -c "$ROOT/treegp_amd/csrc/$f.hip" -o "$B/$f.o" &
  OBJS="$OBJS $B/$f.o"
  N=$((N + 1))
  if [ $((N % 4)) -eq 0 ]; then wait; fi
done
wait
$HIPCC $FLAGS -I "$ROOT/include" -x hip -c "$ROOT/tests/sanitize/host_logic_driver.cpp" -o "$B/drv.o"
$HIPCC --offload-arch=gfx950 -fsanitize=address,undefined -fno-gpu-sanitize -o "$B/drv" "$B/drv.o" $OBJS
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 "$B/drv"
