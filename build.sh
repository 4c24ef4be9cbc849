#!/bin/bash
# Build the C-ABI shared library (gfx950 only) in-tree. hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
PKG="gif-synthesis-with-discrete-diffusion_amd"
SRC="$PKG/csrc"
OUT="$PKG/libgsdd.so"
mkdir -p build
OBJS=""
PIDS=""
for f in $SRC/*.hip; do
  o="build/$(basename "${f%.hip}").o"
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ "$SRC/common.hpp" -nt "$o" ] || [ include/gsdd.h -nt "$o" ]; then
    echo "hipcc $f"
    rm -f "$o"                      # a failed compile must not leave a stale object behind for the link
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -Wall -Wno-unused-function \
        -c "$f" -o "$o" &
    PIDS="$PIDS $!"
  fi
  OBJS="$OBJS $o"
done
for p in $PIDS; do
  wait "$p" || { echo "build failed" >&2; exit 1; }
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" $OBJS
echo "built $OUT"
