#!/bin/bash
# Build a tuning / ablation variant of libdepthcore.so OUTSIDE the package: build/variants/<name>/libdepthcore.so
# (git-ignored; travels to the GPU box with gpurun).  Only the listed sources are recompiled with the extra flags, the
# other objects are the product's.  Used through DEPTHCORE_LIB by the scripts under tools/ only -- bench.py and the tests
# refuse the override.
# usage: tools/build_variant.sh <name> "<extra hipcc flags>" file.hip [file.hip ...]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/self-supervised-depth-estimation_amd/csrc
NAME=$1; EXTRA=$2; shift 2
OUT=$ROOT/build/variants/$NAME
mkdir -p "$OUT"
make -s -C "$CS" -j4
OBJS=""
for o in "$CS"/*.o; do
  b=$(basename "$o" .o); skip=0
  for f in "$@"; do [ "$b.hip" = "$f" ] && skip=1; done
  [ $skip = 0 ] && OBJS="$OBJS $o"
done
for f in "$@"; do
  b=$(basename "$f" .hip)
  FL="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -I$ROOT/include -Wall -Wno-unused-function"
  [ "$b" = data ] && FL="$FL -ffp-contract=off"
  /opt/rocm/bin/hipcc $FL $EXTRA -c "$CS/$f" -o "$OUT/$b.o"
  OBJS="$OBJS $OUT/$b.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o "$OUT/libdepthcore.so"
echo "$OUT/libdepthcore.so"
