// Internal interface of the Winograd F(4x4,3x3) kernel (wino4.hip), used by wino.hip's dispatch and weight cache.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace dc {

constexpr int W4_MT = 116;     // tile-layout code of the F(4x4,3x3) transformed weights in the weight cache (wino.hip: WcVariant.MT)

bool wino4_eligible(int B, int K, int M, int H, int W);
double wino4_utilisation(int H, int W);                 // useful / covered output pixels of the tile-group cover of an H x W map
size_t wino4_uhat_bytes(int Ci, int Co);
void wino4_dims(int Ci, int Co, bool dgrad, int& Mp, int& Kp);
int wino4_launch(const float* x, const float* weight, const float* cached_uhat, float* y, const float* addend, void* ws, float* slabs,
                 int B, int Ci, int Co, int H, int W, bool dgrad, hipStream_t st);
// wino.hip: y = sum of `ksplit` slabs (fixed order) [+ addend]
int wino_ysum_launch(const float* slabs, float* y, size_t n4, int ksplit, const float* addend, hipStream_t st);

// ---- U = G g G^T (6x6) for one (m, k), written in the staging order of wino4_kernel:
//      uhat[m-block][step][kk = k % 4][p4 = p / 4][m % 16][p % 4],  p = 6 a + b.   One thread per (m, k); 256 consecutive
//      threads = a 16 (m) x 16 (k) tile with m fastest (the 16-byte stores of 16 consecutive m are one 256-byte run).
template <bool DGRAD>
__device__ __forceinline__ void wino4_weight_one(const float* __restrict__ w, float* __restrict__ uhat, int idx, int Co, int Ci,
                                                 int Mp, int Kp) {
    const int tiles_k = (Kp + 15) >> 4;
    const int tile = idx >> 8, within = idx & 255;
    const int m = (tile / tiles_k) * 16 + (within & 15), k = (tile % tiles_k) * 16 + (within >> 4);
    if (m >= Mp || k >= Kp) return;
    const int M = DGRAD ? Ci : Co, K = DGRAD ? Co : Ci;
    float g[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float v = 0.f;
            if (m < M && k < K)
                v = DGRAD ? w[((size_t)k * Ci + m) * 9 + (2 - i) * 3 + (2 - j)] : w[((size_t)m * Ci + k) * 9 + i * 3 + j];
            g[i][j] = v;
        }
    // G = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]   (Lavin & Gray 2015, F(4x4,3x3))
    float t[6][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float g0 = g[0][j], g1 = g[1][j], g2 = g[2][j];
        t[0][j] = 0.25f * g0;
        t[1][j] = -(g0 + g1 + g2) * (1.f / 6.f);
        t[2][j] = -(g0 - g1 + g2) * (1.f / 6.f);
        t[3][j] = g0 * (1.f / 24.f) + g1 * (1.f / 12.f) + g2 * (1.f / 6.f);
        t[4][j] = g0 * (1.f / 24.f) - g1 * (1.f / 12.f) + g2 * (1.f / 6.f);
        t[5][j] = g2;
    }
    float u[36];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const float g0 = t[i][0], g1 = t[i][1], g2 = t[i][2];
        u[i * 6 + 0] = 0.25f * g0;
        u[i * 6 + 1] = -(g0 + g1 + g2) * (1.f / 6.f);
        u[i * 6 + 2] = -(g0 - g1 + g2) * (1.f / 6.f);
        u[i * 6 + 3] = g0 * (1.f / 24.f) + g1 * (1.f / 12.f) + g2 * (1.f / 6.f);
        u[i * 6 + 4] = g0 * (1.f / 24.f) - g1 * (1.f / 12.f) + g2 * (1.f / 6.f);
        u[i * 6 + 5] = g2;
    }
    const int nsteps = Kp >> 2;
    float* dst = uhat + ((((size_t)(m >> 4) * nsteps + (k >> 2)) * 4 + (k & 3)) * 9 * 16 + (m & 15)) * 4;
#pragma unroll
    for (int p4 = 0; p4 < 9; ++p4)
        *reinterpret_cast<float4*>(dst + (size_t)p4 * 64) = make_float4(u[p4 * 4], u[p4 * 4 + 1], u[p4 * 4 + 2], u[p4 * 4 + 3]);
}

}  // namespace dc
