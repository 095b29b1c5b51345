// Internal interface of the bf16 matrix-core convolutions (conv_bf16.hip) used by the conv entry points
// (conv3x3.hip, wino.hip, wino_wgrad.hip, convgemm.hip) when dc_set_matrix_precision(DC_PREC_BF16) is in effect.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace dc {

int matrix_precision();       // the calling thread's DC_PREC_* (dc_set_matrix_precision)

constexpr int C3B_BC = 32;    // reduction channels per chunk = the K of one v_mfma_f32_16x16x32_bf16

#ifdef __HIPCC__
__device__ __forceinline__ unsigned c3b_pack_bf16(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 c3b_bf2;
    const c3b_bf2 p = {(__bf16)lo, (__bf16)hi};          // v_cvt_pk_bf16_f32 (round to nearest even, NaN stays NaN)
    return __builtin_bit_cast(unsigned, p);
}
// One 16-byte item of the prepared bf16 weights [m-block][chunk][tap][k-group 4][m MT][8] (conv_bf16.hip: c3b_wprep_kernel, and the
// weight cache's batched refresh in wino.hip): forward M = Co, K = Cin, value w[m][k][tap]; data gradient M = Cin, K = Co, value
// w[k][m][8 - tap] (rotated, transposed filter).
__device__ __forceinline__ void c3b_wprep_item(const float* __restrict__ w, uint4* __restrict__ wb, int idx, int Co, int Cin, int dgrad,
                                               int MT, int nmblk, int nchunks) {
    if (idx >= nmblk * nchunks * 36 * MT) return;
    const int m = idx % MT, cg = (idx / MT) & 3, tap = (idx / (4 * MT)) % 9;
    const int chunk = (idx / (36 * MT)) % nchunks, mblk = idx / (36 * MT * nchunks);
    const int M = dgrad ? Cin : Co, K = dgrad ? Co : Cin;
    const int mm = mblk * MT + m;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = chunk * C3B_BC + cg * 8 + j;
        float t = 0.f;
        if (mm < M && k < K) t = dgrad ? w[((size_t)k * Cin + mm) * 9 + 8 - tap] : w[((size_t)mm * Cin + k) * 9 + tap];
        v[j] = t;
    }
    wb[idx] = make_uint4(c3b_pack_bf16(v[0], v[1]), c3b_pack_bf16(v[2], v[3]), c3b_pack_bf16(v[4], v[5]), c3b_pack_bf16(v[6], v[7]));
}
#endif

// the prepared bf16 weights of (weight, pass, MT) from the per-step weight cache (wino.hip: dc_wino_cache_*), or nullptr: then the
// caller packs them into its workspace as before.  Same rules as the Winograd variants (registered weights, fresh after a refresh).
const void* wc_lookup_c3b(const float* w, int Ci, int Co, int dgrad, int MT, int nmblk, int nchunks, hipStream_t st);

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
             const float* bias, float* out, void* ws, int B, int H, int W, int act, int pad, int stride, hipStream_t st,
             const float* addend = nullptr);       // addend (dpad = 0 only): same shape as out, added in the store epilogue
// part[split][Co][Cin*9] from x = cat(up2?(x0), x1) and g' (B,Co,H/stride,W/stride)
int c3b_wgrad(const float* x0, int C0, int up0, const float* x1, int C1, const float* gp, float* part, int split, int B, int Co, int H,
              int W, int pad, int stride, hipStream_t st);

// fixed-order reduction of split slabs (conv3x3.hip): dw[i] = sum_s part[s][i] (i < nW), db[c] = sum_s pbias[s][c]
int conv_wreduce(const float* part, const float* pbias, float* dw, float* db, int split, int nW, int Co, hipStream_t st);

}  // namespace dc
