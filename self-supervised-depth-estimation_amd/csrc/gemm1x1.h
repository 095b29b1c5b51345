// Tiled 1x1-convolution GEMMs (gemm1x1.hip) behind the dc_conv1x1_* entry points of pointwise.hip.
#pragma once
#include <stddef.h>

#include "depthcore.h"

extern "C" {
/* per pass: 16-byte staging possible and the reduction extent a multiple of the 32-wide chunk */
int dc_gemm1x1_fwd_ok(int B, int Ci, int Co, int Hi, int Wi, int stride);
int dc_gemm1x1_dgrad_ok(int B, int Ci, int Co, int Hi, int Wi, int stride);
int dc_gemm1x1_wgrad_ok(int B, int Ci, int Co, int Hi, int Wi, int stride);
/* bn (nullable): BatchNorm folded into the pass, see dc_bn_fold in depthcore.h */
int dc_gemm1x1_stat_parts(int B, int Ci, int Co, int Hi, int Wi, int stride, int groups, int* ppg);
int dc_gemm1x1_bwd_parts(int B, int Ci, int Co, int Hi, int Wi, int stride, int groups, int* ppg);
int dc_gemm1x1_fwd(const float* x, const float* weight, const float* bias, float* y, int B, int Ci, int Co, int Hi, int Wi, int stride,
                   int act, const dc_bn_fold* bn, void* stream);
int dc_gemm1x1_dgrad(const float* gy, const float* weight, float* dx, const float* addend, const float* addend2, int B, int Ci, int Co,
                     int Hi, int Wi, int stride, const dc_bn_fold* bn, void* stream);      /* addends (nullable): added to dx in the store epilogue */
size_t dc_gemm1x1_wgrad_workspace(int B, int Ci, int Co, int Hi, int Wi, int stride);
int dc_gemm1x1_wgrad(const float* x, const float* gy, float* dweight, void* ws, int B, int Ci, int Co, int Hi, int Wi, int stride,
                     const dc_bn_fold* bn, void* stream);
}
