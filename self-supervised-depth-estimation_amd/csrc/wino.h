// Internal interface of the Winograd kernels (wino.hip, wino_wgrad.hip) used by the fused conv block (conv3x3.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace dc {

// forward / full-correlation data gradient of the fused block  y = act(conv3x3(pad1(cat(up2?(x0), x1))) + bias)
bool wino_conv_eligible(int C0, int C1, int H, int W);
size_t wino_conv_ws_bytes(int B, int Ci, int Co, int H, int W);
int wino_conv_fused_fwd(const float* x0, int C0, int up0, const float* x1, int C1, const float* weight, const float* bias, float* y,
                        void* ws, int B, int Co, int H, int W, int act, int pad, hipStream_t st);
int wino_conv_full_dgrad(const float* gp, const float* weight, float* dxpad, void* ws, int B, int Ci, int Co, int H, int W,
                         hipStream_t st);

// the same data gradient written where it belongs (no padded-domain scratch, no fold pass): dx0 (B,C0,H>>up0,W>>up0) and dx1
// (B,C1,H,W) from the interior of the correlation, addends added on the way out (nullable; may alias their output); the terms
// ReflectionPad folds back from the padded ring are NOT included (conv3x3.hip: conv_ring_kernel adds them).  dx0 / dx1 nullable.
bool wino_dgrad_split_ok(int B, int C0, int C1, int up0, int Co, int H, int W);
int wino_conv_dgrad_split(const float* gp, const float* weight, float* dx0, float* dx1, const float* add0, const float* add1, void* ws,
                          int B, int C0, int C1, int up0, int Co, int H, int W, hipStream_t st);

// weight gradient of the fused block from gp = gy * act'(y): dweight (Co, C0+C1, 3, 3)
size_t wino_wgrad_ws_bytes(int B, int Ci, int Co, int H, int W);
int wino_wgrad_fused(const float* x0, int C0, int up0, const float* x1, int C1, int pad, const float* gp, float* dweight, void* ws,
                     int B, int Co, int H, int W, hipStream_t st);

// measurement hook (dc_conv_profile_*): hipEvent pair around the main kernel of a Winograd launch.
// kind 0: wino_ps_kernel (forward / data gradient), 1: wino_wgrad_kernel, 2: c3b_conv_kernel (bf16 forward / data gradient),
// 3: c3b_wgrad_kernel (bf16), 4: the 1x1 GEMM family (g1_*).  Returns the end event or nullptr.
hipEvent_t conv_prof_begin(int kind, double algorithmic_flops, double executed_flops, double algorithmic_bytes, hipStream_t st);
void conv_prof_end(hipEvent_t e, hipStream_t st);

}  // namespace dc
