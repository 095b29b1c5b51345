#!/bin/bash
# A/B of the photometric kernel variants under rocprofv3 (GPU box).  usage: bash tools/prof_photo_ab.sh
R=$GRAFT_REPO_ROOT
V=$R/build/variants
run() { tag=$1; shift; ( export "$@" DUMMY=1; bash $R/tools/prof_photo.sh $tag ) 2>&1 | tail -1; }
run old DC_PHOTO_OLD=1
run new
for g in 16 24 32 48; do run g$g DC_PHOTO_ROWS_G=$g; done
for p in 4 8 12 16 24 32; do run p$p DC_PHOTO_ROWS_P=$p; done
[ -f $V/fwdg3/libdepthcore.so ] && run fwdg3 DEPTHCORE_LIB=$V/fwdg3/libdepthcore.so
