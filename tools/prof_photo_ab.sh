#!/bin/bash
# A/B of photometric kernel variants under rocprofv3 (GPU box): the product library, then every build/variants/<name>/ given.
# usage: bash tools/prof_photo_ab.sh [variant ...]      (variants: tools/build_variant.sh <name> "<flags>" photo.hip)
R=$GRAFT_REPO_ROOT
( bash $R/tools/prof_photo.sh product ) 2>&1 | tail -1
for v in "$@"; do
  ( export DEPTHCORE_LIB=$R/build/variants/$v/libdepthcore.so; bash $R/tools/prof_photo.sh $v ) 2>&1 | tail -1
done
