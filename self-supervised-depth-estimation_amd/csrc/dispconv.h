// Internal interface of the disparity-head kernels (dispconv.hip) used by the fused conv block (conv3x3.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace dc {

bool dispconv_eligible(int C0, int C1, int up0, int Co, int H, int W);
int dispconv_fwd(const float* x, const float* w, const float* bias, float* y, int B, int C, int H, int W, int act, int pad,
                 hipStream_t st);
bool dispconv_wgrad_eligible(int C0, int C1, int up0, int Co, int H, int W);      // thin heads: 16 or 32 input channels
size_t dispconv_wgrad_scratch(int B, int C, int H, int W);
int dispconv_wgrad(const float* x, const float* y, const float* gy, float* dweight, float* dbias, float* scratch, int B, int C, int H,
                   int W, int act, int pad, hipStream_t st);
int dispconv_dx(const float* w, const float* y, const float* gy, float* dx, const float* addend, int B, int C, int H, int W, int act,
                int pad, hipStream_t st);      /* addend (nullable): added to dx */

}  // namespace dc
