// Split-operand 1x1 GEMMs (gemm1x1_x3.hip): what the per-step weight cache (wino.hip: dc_wino_cache_*) needs of them.
#pragma once
#include "dc_common.h"

namespace dc {

#ifdef __HIPCC__
typedef __attribute__((ext_vector_type(2))) __bf16 x3h_bf2;
typedef __attribute__((ext_vector_type(2))) unsigned x3h_u2;
__device__ __forceinline__ unsigned x3h_pack(float lo, float hi) {
    const x3h_bf2 p = {(__bf16)lo, (__bf16)hi};          // v_cvt_pk_bf16_f32 (round to nearest even)
    return __builtin_bit_cast(unsigned, p);
}
// (a, b) -> the packed bf16 pairs of their three pieces: x = p0 + p1 + p2 up to 2^-25 |x|; each residual is exact in fp32
__device__ __forceinline__ void x3h_split2(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) {
    p0 = x3h_pack(a, b);
    const float ra = a - __builtin_bit_cast(float, p0 << 16), rb = b - __builtin_bit_cast(float, p0 & 0xffff0000u);
    p1 = x3h_pack(ra, rb);
    const float sa = ra - __builtin_bit_cast(float, p1 << 16), sb = rb - __builtin_bit_cast(float, p1 & 0xffff0000u);
    p2 = x3h_pack(sa, sb);
}
// One 8-byte item of the split weights [piece][Mp][K] (bf16): row m, four consecutive POSITIONS of the reduction, which is permuted
// inside every chunk of 32 -- position 8 kg + e <-> k = e < 4 ? 4 kg + e : 16 + 4 kg + e - 4 (the transposed LDS read of the B
// operand delivers its rows in that order, gemm1x1_x3.hip).  forward (tr = 0): A[m][k] = w[m][k] (M = Co, K = Ci); data gradient
// (tr = 1): A[m][k] = w[k][m] (M = Ci, K = Co).  Rows m >= M are zero.
// (the source's row length is K for tr = 0 -- also a (Co, Ci, 3, 3) weight seen as (Co, 9 Ci) -- and M for tr = 1)
__device__ __forceinline__ void g1x3_prep_item(const float* __restrict__ w, unsigned short* __restrict__ wa, int idx, int tr,
                                               int M, int Mp, int K) {
    const int Ci = tr ? M : K;
    const int q = K >> 2;
    if (idx >= Mp * q) return;
    const int m = idx / q, pos4 = (idx - m * q) * 4;
    const int c = pos4 >> 5, pl = pos4 & 31, kg = pl >> 3, e = pl & 7;
    const int k0 = c * 32 + (e < 4 ? 4 * kg : 16 + 4 * kg);
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = m < M ? (tr ? w[(size_t)(k0 + j) * Ci + m] : w[(size_t)m * Ci + k0 + j]) : 0.f;
    unsigned p[3][2];
    x3h_split2(v[0], v[1], p[0][0], p[1][0], p[2][0]);
    x3h_split2(v[2], v[3], p[0][1], p[1][1], p[2][1]);
#pragma unroll
    for (int s = 0; s < 3; ++s)
        *reinterpret_cast<x3h_u2*>(wa + ((size_t)s * Mp + m) * K + pos4) = x3h_u2{p[s][0], p[s][1]};
}
#endif

// the 3 x 3 / 2 trunk convolutions on the split-operand kernels (forward, weight gradient): eligibility, workspace bytes, launches
bool g1x3_conv3s2_ok(int B, int Ci, int Co, int Hi, int Wi);
bool g1x3_conv3s2_wgrad_ok(int B, int Ci, int Co, int Hi, int Wi);
size_t g1x3_conv3s2_fwd_ws(int Ci, int Co);
size_t g1x3_conv3s2_wgrad_ws(int B, int Ci, int Co, int Hi, int Wi);
int g1x3_conv3s2_fwd(const float* x, const float* weight, float* y, void* ws, int B, int Ci, int Co, int Hi, int Wi, hipStream_t st);
int g1x3_conv3s2_wgrad(const float* x, const float* gy, float* dweight, void* ws, int B, int Ci, int Co, int Hi, int Wi, hipStream_t st);

// the split weights of (weight, direction) from the per-step weight cache, or nullptr (then the launch prepares them into its workspace)
const void* wc_lookup_x3(const float* w, int Ci, int Co, int tr, int Mp, int K, hipStream_t st);

}  // namespace dc
