#!/bin/bash
# GPU box: build a -DWINO_DIAG copy of the library and print the per-phase cycle breakdown.
set -e
cd $GRAFT_REPO_ROOT/self-supervised-depth-estimation_amd/csrc
mkdir -p /tmp/diagb && cp *.hip *.h Makefile /tmp/diagb/ && cd /tmp/diagb
for f in *.hip; do /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -I$GRAFT_REPO_ROOT/include -DWINO_DIAG=${WINO_DIAG:-1} -c $f -o ${f%.hip}.o & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC *.o -o /tmp/diagb/libdepthcore_diag.so
DEPTHCORE_LIB=/tmp/diagb/libdepthcore_diag.so python3 $GRAFT_REPO_ROOT/tools/diag_wino.py "$@"
