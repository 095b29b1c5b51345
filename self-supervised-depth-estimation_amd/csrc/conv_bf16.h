// Internal interface of the bf16 matrix-core convolutions (conv_bf16.hip) used by the conv entry points
// (conv3x3.hip, wino.hip, wino_wgrad.hip, convgemm.hip) when dc_set_matrix_precision(DC_PREC_BF16) is in effect.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace dc {

int matrix_precision();       // the calling thread's DC_PREC_* (dc_set_matrix_precision)

// shapes the bf16 kernels take (16-byte staging): stride 1: W % 4 == 0 (W % 8 == 0 when x0 is upsampled / dilated), concat
// boundary on a 32-channel chunk; stride 2: W % 8 == 0, single source.  Anything else keeps the fp32 kernels.
bool c3b_eligible(int C0, int C1, int up0, int H, int W, int stride);
size_t c3b_weights_bytes(int Ci, int Co);                              // prepared bf16 weights of either pass
int c3b_wgrad_split(int B, int OH, int OW, int Co, int Cin, int stride);
int c3b_dpad_pitch(int W);

// out = act(conv3x3(pad1(cat(up2?(x0), x1))) + bias) (dgrad = 0), or the same convolution with the rotated, transposed
// filter (dgrad = 1: `x0` holds g', C0 = Co, out has Cin channels); dpad = 1: output over the padded domain: (H+2) rows of
// c3b_dpad_pitch(W) floats, the first W+2 of each row valid (16-byte row alignment for the kernel's vector stores).
// up0: 0 plain, 1 nearest-x2 upsampled x0, 3 DILATED x0 (full[2y][2x] = x0[y][x], zeros between: with dgrad = 1 and zero
// padding that is the data gradient of the stride-2 convolution).
int c3b_conv(const float* x0, int C0, int up0, const float* x1, int C1, const float* weight, int Co, int Cin, int dgrad, int dpad,
             const float* bias, float* out, void* ws, int B, int H, int W, int act, int pad, int stride, hipStream_t st);
// part[split][Co][Cin*9] from x = cat(up2?(x0), x1) and g' (B,Co,H/stride,W/stride)
int c3b_wgrad(const float* x0, int C0, int up0, const float* x1, int C1, const float* gp, float* part, int split, int B, int Co, int H,
              int W, int pad, int stride, hipStream_t st);

// fixed-order reduction of split slabs (conv3x3.hip): dw[i] = sum_s part[s][i] (i < nW), db[c] = sum_s pbias[s][c]
int conv_wreduce(const float* part, const float* pbias, float* dw, float* db, int split, int nW, int Co, hipStream_t st);

}  // namespace dc
