#!/bin/bash
# forward time of single layers through ablated builds of c3b_conv_kernel (csrc/conv_bf16.hip, -DC3B_ABL=bits; built OUTSIDE the package by
# `tools/build_variant.sh abl<bits> -DC3B_ABL=<bits> conv_bf16.hip` -> build/variants/abl<bits>/libdepthcore.so): 1 no MFMA, 2 no LDS commit, 4 no stores, 8 no patch loads.  GPU box.
for A in 0 1 2 3 4 8 7; do
  L=$GRAFT_REPO_ROOT/build/variants/abl$A/libdepthcore.so
  [ $A = 0 ] && L=$GRAFT_REPO_ROOT/self-supervised-depth-estimation_amd/depthcore/libdepthcore.so
  echo "== ABL $A"
  DEPTHCORE_LIB=$L python3 $GRAFT_REPO_ROOT/tools/bench_bf16.py --batch 36 --prec bf16 --layer upconv_1_1,upconv_0_0,layer1,layer4 2>&1 | grep -v amdgpu.ids | cut -c1-70
done
