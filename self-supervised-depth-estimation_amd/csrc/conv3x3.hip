// Fused 3x3 convolution blocks of the depth decoder (reference layers.py:106-136,196-199 and
// networks/depth_decoder.py:50-66) as fp32 MFMA implicit GEMMs for gfx950:
//
//     y = act( conv3x3( pad1( cat( nearest_x2?(x0), x1 ) ) ) + bias )
//
// forward : M = Co, N = pixels, K = Cin*9.  The nearest x2 upsample, the channel concat and the
//           ReflectionPad (or ZeroPad) are index arithmetic in the global->LDS staging of the input patch;
//           bias + ELU / sigmoid are the epilogue.  None of those tensors ever exists in HBM.
// dgrad   : the same GEMM on g' = gy * act'(y) with flipped / transposed weights over the PADDED output
//           domain (H+2 x W+2); `conv_fold_kernel` then folds the reflected border back, sums the 2x2
//           blocks of the upsampled half and splits the concat -- deterministic, no atomics.
// wgrad   : M = Co, N = Cin*9, K = pixels, split-K over pixel tiles into fp32 partial slabs + fixed-order
//           reduce (also produces dbias).
// Exact fp32: v_mfma_f32_16x16x4_f32 (k-ordered fmaf chain), so parity with the fp32 reference holds to
// summation order.
#include "dc_common.h"
#include "conv_bf16.h"
#include "dispconv.h"
#include "wino.h"

#include <algorithm>
#include <cstdlib>

namespace dc {

typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int CT = 16;                 // pixel tile: CT x CT outputs per block
constexpr int CK = 8;                  // input channels per K chunk
constexpr int PW_ = CT + 2;            // patch width / height
constexpr int PS_ = 336;               // patch plane stride in floats (>= 18*18, == 16 mod 32)

struct ConvArgs {
    const float* x0; int C0; int up0;
    const float* x1; int C1;
    const float* wt;       // prepared weights: fwd [9][Cin][Co]; dgrad [9][Co][Cin] (flipped taps)
    const float* bias;
    const float* y;        // dgrad / wgrad: activated output (for act')
    const float* gy;
    const float* gp;       // dgrad v2: precomputed g' = gy * act'(y), dense (B,Co,H,W)
    float* out;            // fwd: y (B,Co,H,W); dgrad: dxpad (B,Cin,H+2,W+2)
    int B, Co, H, W, act, pad;
    int tiles_x, tiles_y;
};

__device__ __forceinline__ int pad_index(int i, int n, int pad, bool& ok) {
    ok = true;
    if (i >= 0 && i < n) return i;
    if (pad == PAD_REFLECT) {
        i = i < 0 ? -i : 2 * n - 2 - i;
        return min(max(i, 0), n - 1);
    }
    ok = false;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// weight preparation: W (Co,Cin,3,3) -> wf [9][Cin][Co] and wd [9][Co][Cin] with flipped taps
// ------------------------------------------------------------------------------------------------
__global__ void conv_wprep_kernel(const float* w, float* wf, float* wd, int Co, int Cin) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = Co * Cin * 9;
    if (i >= n) return;
    const int t = i % 9, ci = (i / 9) % Cin, co = i / (9 * Cin);
    const float v = w[i];
    if (wf) wf[((size_t)t * Cin + ci) * Co + co] = v;
    if (wd) wd[((size_t)(8 - t) * Co + co) * Cin + ci] = v;
}

// ------------------------------------------------------------------------------------------------
// forward / dgrad GEMM.  grid (tiles_x*tiles_y, ceil(M/(16*MR)), B), block 256 (4 waves, wave w = rows 4w..4w+3)
// ------------------------------------------------------------------------------------------------
template <int MR, bool DGRAD>
__global__ __launch_bounds__(256) void conv_gemm_kernel(ConvArgs a) {
    constexpr int MT = 16 * MR;
    constexpr int WS = MT + 16 + (MR == 1 ? 16 : 0);      // weight row stride, == 16 mod 32
    __shared__ float patch[CK * PS_];
    __shared__ float wl[9 * CK * WS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x, b = blockIdx.z;
    const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    const int oy0 = ty * CT, ox0 = tx * CT;               // tile origin in the OUTPUT domain
    const int m0 = blockIdx.y * MT;
    const int H = a.H, W = a.W;
    const int Cin = a.C0 + a.C1;
    // GEMM dims of this mode
    const int Mtot = DGRAD ? Cin : a.Co;
    const int Ktot = DGRAD ? a.Co : Cin;
    const int OH = DGRAD ? H + 2 : H, OW = DGRAD ? W + 2 : W;
    const int h0 = H >> a.up0, w0 = W >> a.up0;

    f4 acc[MR][4];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

    for (int k0 = 0; k0 < Ktot; k0 += CK) {
        // ---- stage the input patch of CK channels (fused upsample / concat / padding, or g' for dgrad)
        for (int e = tid; e < CK * PW_ * PW_; e += 256) {
            const int kc = e / (PW_ * PW_), rem = e - kc * (PW_ * PW_);
            const int r = rem / PW_, c = rem - r * PW_;
            const int ch = k0 + kc;
            float v = 0.f;
            if (ch < Ktot) {
                if (!DGRAD) {
                    bool oky, okx;
                    const int yy = pad_index(oy0 + r - 1, H, a.pad, oky), xx = pad_index(ox0 + c - 1, W, a.pad, okx);
                    if (oky && okx) {
                        v = (ch < a.C0) ? a.x0[(((size_t)b * a.C0 + ch) * h0 + (yy >> a.up0)) * w0 + (xx >> a.up0)]
                                        : a.x1[(((size_t)b * a.C1 + (ch - a.C0)) * H + yy) * W + xx];
                    }
                } else {
                    const int yy = oy0 + r - 2, xx = ox0 + c - 2;
                    if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
                        const size_t o = (((size_t)b * a.Co + ch) * H + yy) * W + xx;
                        v = a.gy[o] * act_bwd(a.y[o], a.act);
                    }
                }
            }
            patch[kc * PS_ + r * PW_ + c] = v;
        }
        // ---- stage the weights of the chunk: wl[t][kc][m] = wt[t][k0+kc][m0+m]
        for (int e = tid; e < 9 * CK * MT; e += 256) {
            const int m = e % MT, kc = (e / MT) % CK, t = e / (MT * CK);
            float v = 0.f;
            if (k0 + kc < Ktot && m0 + m < Mtot) v = a.wt[((size_t)t * Ktot + k0 + kc) * Mtot + m0 + m];
            wl[(t * CK + kc) * WS + m] = v;
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int ky = t / 3, kx = t - ky * 3;
#pragma unroll
            for (int kk = 0; kk < CK / 4; ++kk) {
                const int kc = kk * 4 + (lane >> 4);
                float af[MR], bf[4];
#pragma unroll
                for (int i = 0; i < MR; ++i) af[i] = wl[(t * CK + kc) * WS + i * 16 + (lane & 15)];
#pragma unroll
                for (int j = 0; j < 4; ++j) bf[j] = patch[kc * PS_ + (wave * 4 + j + ky) * PW_ + (lane & 15) + kx];
#pragma unroll
                for (int i = 0; i < MR; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // ---- epilogue: C/D layout col (pixel x) = lane&15, row (m) = (lane>>4)*4 + reg
    const int px = ox0 + (lane & 15);
#pragma unroll
    for (int i = 0; i < MR; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int py = oy0 + wave * 4 + j;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + i * 16 + (lane >> 4) * 4 + r;
                if (m < Mtot && py < OH && px < OW) {
                    float v = acc[i][j][r];
                    if (!DGRAD) v = act_fwd(v + (a.bias ? a.bias[m] : 0.f), a.act);
                    a.out[(((size_t)b * Mtot + m) * OH + py) * OW + px] = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// v2 of the forward / dgrad GEMM for W % 16 == 0 (every decoder level >= 48x160): the same MFMA core, but
//   * staging moves 16-byte vectors: the 16 interior columns of a patch row are 64-byte aligned in NCHW, so a
//     row is 4 float4 (2 for the nearest-upsampled half, each value written twice) + 2 edge dwords instead of
//     18 dword gathers, and weight rows ([tap][k][m], m contiguous) are float4 as well;
//   * the next chunk's global loads are issued before the MFMA phase of the current one and land in registers
//     (software double buffering), so HBM/L2 latency overlaps the matrix work inside a block;
//   * dgrad reads the precomputed g' = gy*act'(y) (one elementwise pass shared with wgrad) instead of gy and y.
// ------------------------------------------------------------------------------------------------
__global__ void conv_gprime_kernel(const float* gy, const float* y, float* gp, size_t n4, int act) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n4; i += stride) {
        float4 g = reinterpret_cast<const float4*>(gy)[i];
        const float4 v = reinterpret_cast<const float4*>(y)[i];
        g.x *= act_bwd(v.x, act); g.y *= act_bwd(v.y, act); g.z *= act_bwd(v.z, act); g.w *= act_bwd(v.w, act);
        reinterpret_cast<float4*>(gp)[i] = g;
    }
}

// g' = gy * act'(y) AND the bias-gradient partials in the same pass (the separate per-channel sum read g' again at
// 0.6 TB/s: 128 blocks cannot fill the chip).  grid (Co, split): block (co, sp) takes slice sp of channel co's B * HW / 4
// float4 items, writes their g' (not when act == NONE: g' is gy itself) and one partial sum pbias[sp][co] (block tree,
// fixed order); conv_wreduce_kernel adds the `split` partials in order.  HW % 4 == 0.
__global__ __launch_bounds__(256) void conv_gprime_dbias_kernel(const float* __restrict__ gy, const float* __restrict__ y,
                                                               float* __restrict__ gp, float* __restrict__ pbias, int B, int Co,
                                                               int HW4, int act, int split) {
    __shared__ float sm[256];
    const int co = blockIdx.x, sp = blockIdx.y;
    const int total = B * HW4, per = (total + split - 1) / split;
    const int i1 = min(total, (sp + 1) * per);
    float acc = 0.f;
    for (int i = sp * per + threadIdx.x; i < i1; i += 256) {
        const int b = i / HW4, q = i - b * HW4;
        const size_t o = ((size_t)b * Co + co) * HW4 + q;
        float4 g = reinterpret_cast<const float4*>(gy)[o];
        if (act != ACT_NONE) {
            const float4 v = reinterpret_cast<const float4*>(y)[o];
            g.x *= act_bwd(v.x, act); g.y *= act_bwd(v.y, act); g.z *= act_bwd(v.z, act); g.w *= act_bwd(v.w, act);
            reinterpret_cast<float4*>(gp)[o] = g;
        }
        acc += (g.x + g.y) + (g.z + g.w);
    }
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0 && pbias) pbias[(size_t)sp * Co + co] = sm[0];
}
static inline int gpd_split(int Co) { return std::max(8, std::min(256, 2048 / std::max(Co, 1))); }

template <int MR, bool DGRAD>
__global__ __launch_bounds__(256) void conv_gemm_v2_kernel(ConvArgs a) {
    constexpr int MT = 16 * MR;
    constexpr int WS = MT + 16 + (MR == 1 ? 16 : 0);      // weight row stride, == 16 mod 32
    constexpr int NVEC = 3, NEDGE = 2, NWV = (9 * CK * (MT / 4) + 255) / 256;
    __shared__ __attribute__((aligned(16))) float patch[CK * PS_];
    __shared__ __attribute__((aligned(16))) float wl[9 * CK * WS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x, b = blockIdx.z;
    const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    const int oy0 = ty * CT, ox0 = tx * CT;
    const int m0 = blockIdx.y * MT;
    const int H = a.H, W = a.W;
    const int Cin = a.C0 + a.C1;
    const int Mtot = DGRAD ? Cin : a.Co;
    const int Ktot = DGRAD ? a.Co : Cin;
    const int OH = DGRAD ? H + 2 : H, OW = DGRAD ? W + 2 : W;
    const int h0 = H >> a.up0, w0 = W >> a.up0;

    f4 acc[MR][4];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

    float4 rv[NVEC];      // vector items of the patch
    float re[NEDGE];      // edge items
    float4 rw[NWV];       // weight items

    // ---- issue the global loads of the chunk starting at channel k0 (values land in rv / re / rw)
    auto prefetch = [&](int k0) {
        const bool upmode = !DGRAD && a.up0 && k0 < a.C0;          // chunk comes from the half-resolution x0
        const int quads = upmode ? 2 : 4;
#pragma unroll
        for (int j = 0; j < NVEC; ++j) {
            const int it = tid + j * 256;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (it < CK * PW_ * quads) {
                const int kc = it / (PW_ * quads), rem = it - kc * (PW_ * quads);
                const int r = rem / quads, q = rem - r * quads;
                const int ch = k0 + kc;
                if (ch < Ktot) {
                    if (DGRAD) {
                        const int yy = oy0 + r - 2, xx = ox0 + 4 * q;
                        if (yy >= 0 && yy < H && xx < W)
                            v = *reinterpret_cast<const float4*>(a.gp + (((size_t)b * a.Co + ch) * H + yy) * W + xx);
                    } else {
                        bool oky;
                        const int yy = pad_index(oy0 + r - 1, H, a.pad, oky);
                        if (oky) {
                            if (upmode)
                                v = *reinterpret_cast<const float4*>(a.x0 + (((size_t)b * a.C0 + ch) * h0 + (yy >> 1)) * w0 + (ox0 >> 1) + 4 * q);
                            else if (ch < a.C0)
                                v = *reinterpret_cast<const float4*>(a.x0 + (((size_t)b * a.C0 + ch) * H + yy) * W + ox0 + 4 * q);
                            else
                                v = *reinterpret_cast<const float4*>(a.x1 + (((size_t)b * a.C1 + (ch - a.C0)) * H + yy) * W + ox0 + 4 * q);
                        }
                    }
                }
            }
            rv[j] = v;
        }
#pragma unroll
        for (int j = 0; j < NEDGE; ++j) {
            const int it = tid + j * 256;
            float v = 0.f;
            if (it < CK * PW_ * 2) {
                const int kc = it / (PW_ * 2), rem = it - kc * (PW_ * 2);
                const int r = rem >> 1, side = rem & 1;
                const int ch = k0 + kc;
                if (ch < Ktot) {
                    if (DGRAD) {
                        const int yy = oy0 + r - 2, xx = ox0 - 2 + side;       // patch columns 0, 1
                        if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = a.gp[(((size_t)b * a.Co + ch) * H + yy) * W + xx];
                    } else {
                        bool oky, okx;
                        const int yy = pad_index(oy0 + r - 1, H, a.pad, oky);
                        const int xx = pad_index(side ? ox0 + CT : ox0 - 1, W, a.pad, okx);   // patch columns 0, 17
                        if (oky && okx) {
                            v = (ch < a.C0) ? a.x0[(((size_t)b * a.C0 + ch) * h0 + (yy >> a.up0)) * w0 + (xx >> a.up0)]
                                            : a.x1[(((size_t)b * a.C1 + (ch - a.C0)) * H + yy) * W + xx];
                        }
                    }
                }
            }
            re[j] = v;
        }
#pragma unroll
        for (int j = 0; j < NWV; ++j) {
            const int it = tid + j * 256;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (it < 9 * CK * (MT / 4)) {
                const int m4 = it % (MT / 4), kc = (it / (MT / 4)) % CK, t = it / ((MT / 4) * CK);
                if (k0 + kc < Ktot && m0 + 4 * m4 < Mtot) {
                    const float* wp = a.wt + ((size_t)t * Ktot + k0 + kc) * Mtot + m0 + 4 * m4;
                    if ((Mtot & 3) == 0) {
                        v = *reinterpret_cast<const float4*>(wp);
                    } else {                      // e.g. the Co = 1 dispconv: rows are not 16-byte aligned
                        const int left = Mtot - (m0 + 4 * m4);
                        v.x = wp[0];
                        if (left > 1) v.y = wp[1];
                        if (left > 2) v.z = wp[2];
                        if (left > 3) v.w = wp[3];
                    }
                }
            }
            rw[j] = v;
        }
    };
    // ---- registers -> LDS
    auto commit = [&](int k0) {
        const bool upmode = !DGRAD && a.up0 && k0 < a.C0;
        const int quads = upmode ? 2 : 4;
#pragma unroll
        for (int j = 0; j < NVEC; ++j) {
            const int it = tid + j * 256;
            if (it < CK * PW_ * quads) {
                const int kc = it / (PW_ * quads), rem = it - kc * (PW_ * quads);
                const int r = rem / quads, q = rem - r * quads;
                float* dst = patch + kc * PS_ + r * PW_;
                const float4 v = rv[j];
                if (DGRAD) {
                    dst += 2 + 4 * q;
                    dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
                } else if (upmode) {
                    dst += 1 + 8 * q;
                    dst[0] = v.x; dst[1] = v.x; dst[2] = v.y; dst[3] = v.y; dst[4] = v.z; dst[5] = v.z; dst[6] = v.w; dst[7] = v.w;
                } else {
                    dst += 1 + 4 * q;
                    dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NEDGE; ++j) {
            const int it = tid + j * 256;
            if (it < CK * PW_ * 2) {
                const int kc = it / (PW_ * 2), rem = it - kc * (PW_ * 2);
                const int r = rem >> 1, side = rem & 1;
                patch[kc * PS_ + r * PW_ + (DGRAD ? side : side * (PW_ - 1))] = re[j];
            }
        }
#pragma unroll
        for (int j = 0; j < NWV; ++j) {
            const int it = tid + j * 256;
            if (it < 9 * CK * (MT / 4)) {
                const int m4 = it % (MT / 4), kc = (it / (MT / 4)) % CK, t = it / ((MT / 4) * CK);
                *reinterpret_cast<float4*>(wl + (t * CK + kc) * WS + 4 * m4) = rw[j];
            }
        }
    };

    prefetch(0);
    for (int k0 = 0; k0 < Ktot; k0 += CK) {
        __syncthreads();              // the previous chunk's MFMAs are done with the LDS tiles
        commit(k0);
        __syncthreads();
        if (k0 + CK < Ktot) prefetch(k0 + CK);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int ky = t / 3, kx = t - ky * 3;
#pragma unroll
            for (int kk = 0; kk < CK / 4; ++kk) {
                const int kc = kk * 4 + (lane >> 4);
                float af[MR], bf[4];
#pragma unroll
                for (int i = 0; i < MR; ++i) af[i] = wl[(t * CK + kc) * WS + i * 16 + (lane & 15)];
#pragma unroll
                for (int j = 0; j < 4; ++j) bf[j] = patch[kc * PS_ + (wave * 4 + j + ky) * PW_ + (lane & 15) + kx];
#pragma unroll
                for (int i = 0; i < MR; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
        }
    }
    const int px = ox0 + (lane & 15);
#pragma unroll
    for (int i = 0; i < MR; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int py = oy0 + wave * 4 + j;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + i * 16 + (lane >> 4) * 4 + r;
                if (m < Mtot && py < OH && px < OW) {
                    float v = acc[i][j][r];
                    if (!DGRAD) v = act_fwd(v + (a.bias ? a.bias[m] : 0.f), a.act);
                    a.out[(((size_t)b * Mtot + m) * OH + py) * OW + px] = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// fold: dxpad (B,Cin,H+2,W+2) -> dx0 (B,C0,H>>up,W>>up) [2x2 sum when up], dx1 (B,C1,H,W)
// reflect:  d x[r] = dxpad[r] + (r==1 ? dxpad[-1] : 0) + (r==H-2 ? dxpad[H] : 0), same along x.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float fold_at(const float* p, int r, int c, int H, int W, int pad, int PWd) {
    // padded coordinates that map onto (r, c): itself, and with ReflectionPad the mirror row (-1 onto 1, H onto H-2) and the
    // mirror column.
    if (H >= 4 && W >= 4) {
        // Branch-free: the mirror terms are always loaded (from the pixel itself when there is none) and weighted 0 / 1 --
        // data-dependent branches around the loads made every load wait for the previous one.
        const bool refl = pad == PAD_REFLECT;
        const bool fr = refl && (r == 1 || r == H - 2), fc = refl && (c == 1 || c == W - 2);
        const int mr = fr ? (r == 1 ? -1 : H) : r, mc = fc ? (c == 1 ? -1 : W) : c;
        const float* row0 = p + (size_t)(r + 1) * PWd + 1;
        const float* row1 = p + (size_t)(mr + 1) * PWd + 1;
        const float v00 = row0[c], v01 = row0[mc], v10 = row1[c], v11 = row1[mc];
        return v00 + (fc ? v01 : 0.f) + (fr ? v10 : 0.f) + ((fr && fc) ? v11 : 0.f);
    }
    // tiny maps (H or W of 2 or 3): row 1 can be the mirror target of both borders -- up to three sources per axis
    int rs[3], cs[3], nr = 1, nc = 1;
    rs[0] = r; cs[0] = c;
    if (pad == PAD_REFLECT) {
        if (r == 1) rs[nr++] = -1;
        if (r == H - 2) rs[nr++] = H;
        if (c == 1) cs[nc++] = -1;
        if (c == W - 2) cs[nc++] = W;
    }
    float v = 0.f;
    for (int i = 0; i < nr; ++i)
        for (int j = 0; j < nc; ++j) v += p[(size_t)(rs[i] + 1) * PWd + cs[j] + 1];
    return v;
}

// `pitch`: row pitch of dxpad in floats (W + 2, or c3b_dpad_pitch(W) behind the bf16 kernels)
// grid (ceil(npix / (256 FOLD_PPT)), Cin, B): a thread folds FOLD_PPT pixels, 256 apart.
// Only the pixels of rows 1 / H-2 and columns 1 / W-2 receive mirror terms (ReflectionPad2d), and only under reflection
// padding: every thread issues its own 1 (4: upsampled source) loads per pixel for all its pixels first, and the few
// threads that touch a mirrored row or column redo that pixel through fold_at afterwards.
constexpr int FOLD_PPT = 1;      // (4 pixels per thread measured slower: 352 vs 307 us per decoder step -- it is not the block count)
// add0 / add1 (nullable, shapes of dx0 / dx1): another consumer's gradient of the same input, added on the way out (the sum
// autograd would otherwise form in a pass of its own)
__global__ __launch_bounds__(256) void conv_fold_kernel(const float* __restrict__ dxpad, float* dx0, float* dx1,
                                                        int B, int C0, int C1, int up0, int H, int W, int pad, int pitch,
                                                        const float* add0, const float* add1) {      // (an addend may BE its output: in-place sum)
    const int Cin = C0 + C1;
    const int b = blockIdx.z, ch = blockIdx.y;
    const float* p = dxpad + ((size_t)b * Cin + ch) * (H + 2) * pitch;
    const bool refl = pad == PAD_REFLECT && H >= 4 && W >= 4;
    const bool generic = !(H >= 4 && W >= 4);          // tiny maps: fold_at's multi-source path
    const bool first = ch < C0;
    float* dst = first ? dx0 : dx1;
    if (!dst) return;
    const int upx = first ? up0 : 0;
    const int h0 = H >> upx, w0 = W >> upx, np = h0 * w0;
    const size_t plane_off = ((size_t)b * (first ? C0 : C1) + (first ? ch : ch - C0)) * np;
    dst += plane_off;
    const float* add = first ? add0 : add1;
    float av[FOLD_PPT];
    float v[FOLD_PPT];
    int yy[FOLD_PPT], xx[FOLD_PPT];
    bool in[FOLD_PPT];
#pragma unroll
    for (int k = 0; k < FOLD_PPT; ++k) {
        const int i = (blockIdx.x * FOLD_PPT + k) * 256 + threadIdx.x;
        in[k] = i < np;
        const int ii = in[k] ? i : 0;
        yy[k] = ii / w0; xx[k] = ii - yy[k] * w0;
        av[k] = add ? add[plane_off + ii] : 0.f;
        if (upx) {
            const float* r0 = p + (size_t)(2 * yy[k] + 1) * pitch + 1 + 2 * xx[k];
            v[k] = (r0[0] + r0[1]) + (r0[pitch] + r0[pitch + 1]);
        } else {
            v[k] = p[(size_t)(yy[k] + 1) * pitch + 1 + xx[k]];
        }
    }
#pragma unroll
    for (int k = 0; k < FOLD_PPT; ++k) {
        const int y = yy[k], x = xx[k];
        if (upx) {
            const bool edge = generic || (refl && (y == 0 || 2 * y + 1 == H - 2 || 2 * y == H - 2 || x == 0 || 2 * x + 1 == W - 2 || 2 * x == W - 2));
            if (edge)
                v[k] = (fold_at(p, 2 * y, 2 * x, H, W, pad, pitch) + fold_at(p, 2 * y, 2 * x + 1, H, W, pad, pitch)) +
                       (fold_at(p, 2 * y + 1, 2 * x, H, W, pad, pitch) + fold_at(p, 2 * y + 1, 2 * x + 1, H, W, pad, pitch));
        } else if (generic || (refl && (y == 1 || y == H - 2 || x == 1 || x == W - 2))) {
            v[k] = fold_at(p, y, x, H, W, pad, pitch);
        }
        if (in[k]) dst[(blockIdx.x * FOLD_PPT + k) * 256 + threadIdx.x] = v[k] + av[k];
    }
}

// Four consecutive pixels of a row per thread (maps of >= 4 x 4 with whole quads per row): the padded-domain reads sit one float
// off a 16-byte boundary -- dword-aligned 16-byte loads (the hardware takes them) -- and the gradient leaves as one aligned
// 16-byte store: 4 x the bytes in flight per thread of conv_fold_kernel (which ran at ~2 TB/s; 0.40 ms of a C2 step).  Same sums
// in the same order: results are bitwise those of conv_fold_kernel.  grid (ceil(max quads per plane / 256), Cin, B).
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
__global__ __launch_bounds__(256) void conv_fold4_kernel(const float* __restrict__ dxpad, float* dx0, float* dx1,
                                                         int B, int C0, int C1, int up0, int H, int W, int pad, int pitch,
                                                         const float* add0, const float* add1) {
    const int Cin = C0 + C1;
    const int b = blockIdx.z, ch = blockIdx.y;
    const float* p = dxpad + ((size_t)b * Cin + ch) * (H + 2) * pitch;
    const bool refl = pad == PAD_REFLECT;
    const bool first = ch < C0;
    float* dst = first ? dx0 : dx1;
    if (!dst) return;
    const int upx = first ? up0 : 0;
    const int h0 = H >> upx, w0 = W >> upx, wq = w0 >> 2;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= h0 * wq) return;
    const int y = i / wq, x = (i - y * wq) * 4;
    const size_t o = ((size_t)b * (first ? C0 : C1) + (first ? ch : ch - C0)) * h0 * w0 + (size_t)y * w0 + x;
    const float* add = first ? add0 : add1;
    float4 av = make_float4(0.f, 0.f, 0.f, 0.f);
    if (add) av = *reinterpret_cast<const float4*>(add + o);
    float v[4];
    if (upx) {
        const float* r0 = p + (size_t)(2 * y + 1) * pitch + 1 + 2 * x;
        const f4u a0 = *reinterpret_cast<const f4u*>(r0), a1 = *reinterpret_cast<const f4u*>(r0 + 4);
        const f4u b0 = *reinterpret_cast<const f4u*>(r0 + pitch), b1 = *reinterpret_cast<const f4u*>(r0 + pitch + 4);
        v[0] = (a0.x + a0.y) + (b0.x + b0.y); v[1] = (a0.z + a0.w) + (b0.z + b0.w);
        v[2] = (a1.x + a1.y) + (b1.x + b1.y); v[3] = (a1.z + a1.w) + (b1.z + b1.w);
        if (refl) {
            const bool er = y == 0 || 2 * y + 1 == H - 2 || 2 * y == H - 2;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int xk = x + k;
                if (er || xk == 0 || 2 * xk + 1 == W - 2 || 2 * xk == W - 2)
                    v[k] = (fold_at(p, 2 * y, 2 * xk, H, W, pad, pitch) + fold_at(p, 2 * y, 2 * xk + 1, H, W, pad, pitch)) +
                           (fold_at(p, 2 * y + 1, 2 * xk, H, W, pad, pitch) + fold_at(p, 2 * y + 1, 2 * xk + 1, H, W, pad, pitch));
            }
        }
    } else {
        const f4u a = *reinterpret_cast<const f4u*>(p + (size_t)(y + 1) * pitch + 1 + x);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
        if (refl) {
            const bool er = y == 1 || y == H - 2;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int xk = x + k;
                if (er || xk == 1 || xk == W - 2) v[k] = fold_at(p, y, xk, H, W, pad, pitch);
            }
        }
    }
    *reinterpret_cast<float4*>(dst + o) = make_float4(v[0] + av.x, v[1] + av.y, v[2] + av.z, v[3] + av.w);
}

static int conv_fold(const float* dxpad, float* dx0, float* dx1, int B, int C0, int C1, int up0, int H, int W, int pad, int pitch,
                     const float* add0, const float* add1, hipStream_t st) {
    const int Cin = C0 + C1;
    static const bool fold4 = !(std::getenv("DC_FOLD4") && std::getenv("DC_FOLD4")[0] == '0');      // (0: the one-pixel kernel, for A/Bs)
    const bool quads = fold4 && H >= 4 && W >= 4 && W % 4 == 0 && (W >> up0) % 4 == 0 && ((H >> up0) >= 1) &&
                       !(((size_t)dx0 | (size_t)dx1 | (size_t)add0 | (size_t)add1) & 15);
    if (quads) {
        const int nq = std::max(C1 > 0 ? H * (W / 4) : 0, C0 > 0 ? (H >> up0) * ((W >> up0) / 4) : 0);
        hipLaunchKernelGGL(conv_fold4_kernel, dim3(ceil_div(nq, 256), Cin, B), dim3(256), 0, st, dxpad, dx0, dx1, B, C0, C1, up0, H, W, pad,
                           pitch, add0, add1);
    } else {
        const int npix = std::max(H * W, (H >> up0) * (W >> up0));
        hipLaunchKernelGGL(conv_fold_kernel, dim3(ceil_div(npix, 256 * FOLD_PPT), Cin, B), dim3(256), 0, st, dxpad, dx0, dx1, B, C0, C1, up0,
                           H, W, pad, pitch, add0, add1);
    }
    DC_CHECK_LAUNCH();
    return DC_OK;
}

// ---- the padded ring of the full correlation, for the data gradient that is written without a padded scratch (wino.h:
// wino_conv_dgrad_split).  ReflectionPad2d makes padded row -1 a copy of row 1, row H of row H-2, and the same for columns: the
// correlation's values on that ring belong to rows 1 / H-2 and columns 1 / W-2 of the gradient.  A ring value has only the taps
// that reach inside the image: one kernel row (or column) of g' row 0 / H-1 (column 0 / W-1).  One thread per AFFECTED gradient
// element (2 W + 2 (H - 2) per plane; of the half-resolution plane for an upsampled x0: the elements whose 2 x 2 block touches an
// affected pixel) sums its ring terms over the output channels in a fixed order and adds them to what the split store wrote.
// element t of the border set of an h x w plane whose rows r0, r1 and columns c0, c1 are affected: the two rows first, then the two
// columns without the rows' elements
__device__ __forceinline__ bool ring_element(int t, int h, int w, int r0, int r1, int c0, int c1, int& y, int& x) {
    if (t < w) { y = r0; x = t; return true; }
    if (t < 2 * w) { y = r1; x = t - w; return r1 != r0; }
    t -= 2 * w;
    const int nrest = h - (r1 != r0 ? 2 : 1);
    if (t >= 2 * nrest) return false;
    const int j = t % nrest;
    x = t < nrest ? c0 : c1;
    if (t >= nrest && c1 == c0) return false;
    y = j < r0 ? j : (j + 1 < r1 || r1 == r0 ? j + 1 : j + 2);          // the j-th row that is neither r0 nor r1
    return true;
}
// Phase 1 -- the ring itself, four strips per (image, input channel) into a small scratch R[b][c][strip][LP]:
//   strip 0 / 1 (padded row -1 / H):    R[pos] = sum_co sum_kx g'[co][0 | H-1][pos - 1 + 1 - kx] w[co][c][0 | 2][kx],  pos - 1 = x in [-1, W]
//   strip 2 / 3 (padded column -1 / W): R[pos] = sum_co sum_ky g'[co][pos - 1 + 1 - ky][0 | W-1] w[co][c][ky][0 | 2],  pos - 1 = y in [-1, H]
// (x = -1 and x = W are the corners of the padded domain).  A small dense product per strip -- M = Cin, N = strip length, K = 3 Co --
// as plain FMAs: a block takes 256 positions x RING_CB channels, the strip's g' line and the weights of RING_CO output channels at a
// time through LDS.  grid (ceil((max(H, W) + 2) / 256), 4 * ceil(Cin / RING_CB), B)
// A block stages the strip's g' line for ALL output channels once (Co <= RING_MAXCO: the wide levels) and walks its share of the
// input channels in chunks of RING_CB; 128 positions x two channel halves per block.  (The first version re-staged the line for every
// channel chunk: the column strips -- one cache line per element -- made it 25 us per launch.)
// grid (ceil((max(H, W) + 2) / 128), 4 * cgroups, B), dynamic LDS (Co * 132 + Co * RING_CB * 3) floats
constexpr int RING_CB = 8, RING_MAXCO = 64, RING_SEG = 128;
__global__ __launch_bounds__(256) void conv_ring_strips_kernel(const float* __restrict__ gp, const float* __restrict__ w, float* __restrict__ R,
                                                               int Cin, int Co, int H, int W, int LP, int cgroups) {
    extern __shared__ float ring_lds[];
    float* line = ring_lds;                                   // [Co][RING_SEG + 4]
    float* wl = ring_lds + Co * (RING_SEG + 4);               // [Co][RING_CB][3]
    const int t = threadIdx.x, tp = t & (RING_SEG - 1), half = t >> 7;
    const int strip = blockIdx.y & 3, grp = blockIdx.y >> 2, b = blockIdx.z;
    const int L = strip < 2 ? W : H, p0 = blockIdx.x * RING_SEG;
    if (p0 >= L + 2) return;
    const size_t HW = (size_t)H * W;
    const float* gpb = gp + (size_t)b * Co * HW;
    const int fixed = (strip & 1) ? (strip < 2 ? H - 1 : W - 1) : 0;            // the row (strips 0, 1) / column (2, 3) of g' the strip reads
    for (int e = t; e < Co * (RING_SEG + 2); e += 256) {
        const int co = e / (RING_SEG + 2), j = e - co * (RING_SEG + 2), i = p0 - 2 + j;
        float v = 0.f;
        if (i >= 0 && i < L) v = gpb[(size_t)co * HW + (strip < 2 ? (size_t)fixed * W + i : (size_t)i * W + fixed)];
        line[co * (RING_SEG + 4) + j] = v;
    }
    const int nchunk = (Cin + RING_CB - 1) / RING_CB, per = (nchunk + cgroups - 1) / cgroups;
    const int pos = p0 + tp;
    for (int ch = grp * per; ch < min(nchunk, (grp + 1) * per); ++ch) {
        const int c0 = ch * RING_CB;
        __syncthreads();                                      // the line (first round) / the previous chunk's weights are free
        for (int e = t; e < Co * RING_CB * 3; e += 256) {
            const int co = e / (RING_CB * 3), rem = e - co * (RING_CB * 3), cc = rem / 3, k = rem - cc * 3;
            float v = 0.f;
            if (c0 + cc < Cin) {
                const float* wk = w + ((size_t)co * Cin + c0 + cc) * 9;
                v = strip < 2 ? wk[((strip & 1) ? 6 : 0) + k] : wk[k * 3 + ((strip & 1) ? 2 : 0)];
            }
            wl[e] = v;
        }
        __syncthreads();
        float acc[RING_CB / 2];
#pragma unroll
        for (int i = 0; i < RING_CB / 2; ++i) acc[i] = 0.f;
#pragma unroll 4
        for (int co = 0; co < Co; ++co) {
            const float* ln = line + co * (RING_SEG + 4) + tp;
            const float l0 = ln[2], l1 = ln[1], l2 = ln[0];                    // taps k = 0, 1, 2: position pos - k
            const float* wq = wl + (co * RING_CB + half * (RING_CB / 2)) * 3;
#pragma unroll
            for (int cc = 0; cc < RING_CB / 2; ++cc) acc[cc] = fmaf(l2, wq[cc * 3 + 2], fmaf(l1, wq[cc * 3 + 1], fmaf(l0, wq[cc * 3], acc[cc])));
        }
        if (pos < L + 2)
#pragma unroll
            for (int cc = 0; cc < RING_CB / 2; ++cc) {
                const int c = c0 + half * (RING_CB / 2) + cc;
                if (c < Cin) R[(((size_t)b * Cin + c) * 4 + strip) * LP + pos] = acc[cc];
            }
    }
}
// Phase 2 -- one thread per AFFECTED gradient element adds what the ring folds onto it.  An element covers full-resolution pixels
// [ya, yb] x [xa, xb] (one pixel; a 2 x 2 block of the half-resolution plane of an upsampled x0): the row strip's values of its
// columns if it contains row 1 / H-2, the column strip's values of its rows if it contains column 1 / W-2, the corner if both.
// grid (ceil((2 W + 2 H) / 256), Cin, B)
__global__ __launch_bounds__(256) void conv_ring_kernel(const float* __restrict__ R, float* dx0, float* dx1, int C0, int C1, int up0, int H,
                                                        int W, int LP) {
    const int Cin = C0 + C1, c = blockIdx.y, b = blockIdx.z;
    const bool first = c < C0;
    float* dst = first ? dx0 : dx1;
    if (!dst) return;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const bool up = first && up0;
    int ey, ex;
    const int eh = up ? H >> 1 : H, ew = up ? W >> 1 : W;
    if (!(up ? ring_element(t, eh, ew, 0, eh - 1, 0, ew - 1, ey, ex) : ring_element(t, H, W, 1, H - 2, 1, W - 2, ey, ex))) return;
    const int ya = up ? 2 * ey : ey, yb = up ? 2 * ey + 1 : ey, xa = up ? 2 * ex : ex, xb = up ? 2 * ex + 1 : ex;
    const bool top = ya <= 1 && 1 <= yb, bot = ya <= H - 2 && H - 2 <= yb;          // (H >= 4: never both)
    const bool lef = xa <= 1 && 1 <= xb, rig = xa <= W - 2 && W - 2 <= xb;
    const float* Rc = R + ((size_t)b * Cin + c) * 4 * LP;
    float r = 0.f;
    if (top || bot) {
        const float* Rr = Rc + (top ? 0 : 1) * LP;
        r += Rr[xa + 1];
        if (xb > xa) r += Rr[xb + 1];
        if (lef || rig) r += Rr[lef ? 0 : W + 1];
    }
    if (lef || rig) {
        const float* Rl = Rc + (lef ? 2 : 3) * LP;
        r += Rl[ya + 1];
        if (yb > ya) r += Rl[yb + 1];
    }
    const size_t plane = first ? (size_t)b * C0 + c : (size_t)b * C1 + (c - C0);
    dst[(plane * eh + ey) * ew + ex] += r;
}

// ------------------------------------------------------------------------------------------------
// wgrad: dW[co][ci][t] = sum_{b,y,x} g'[b,co,y,x] * xpad[b,ci,y+ky-1,x+kx-1]
// GEMM M = Co (16*MR per block), N = 16 (ci) per n-subtile x 9 taps, K = pixels (4 per MFMA).
// grid (split, ceil(Co/(16*MR)), ceil(Cin/CW)), block 256; block loops over its share of the pixel tiles.
// Each wave takes a quarter of the tile's pixel rows and all (ci, tap) columns of the chunk; the four
// waves' accumulators are summed through LDS in fixed order, then written as one partial slab.
// ------------------------------------------------------------------------------------------------
constexpr int CW = 16;                 // input channels per wgrad block
constexpr int GS_ = 273;               // g' plane stride (16*16 + 17), odd
constexpr int XS_ = 325;               // x patch plane stride (18*18 + 1), odd

struct WgradArgs {
    const float* x0; int C0; int up0;
    const float* x1; int C1;
    const float* y; const float* gy;
    const float* gp;       // v2: precomputed g' = gy * act'(y)
    float* part;           // [split][Co][Cin*9]
    float* pbias;          // [split][Co]  (written by the blocks with blockIdx.z == 0)
    int B, Co, H, W, act, pad;
    int tiles_x, tiles_y, split;
};

template <int MR>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs a) {
    constexpr int MT = 16 * MR;
    __shared__ float gl[MT * GS_];
    __shared__ float xl[CW * XS_];
    __shared__ float red[4][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * MT, c0 = blockIdx.z * CW;
    const int H = a.H, W = a.W, Cin = a.C0 + a.C1;
    const int h0 = H >> a.up0, w0 = W >> a.up0;
    const int ntiles = a.tiles_x * a.tiles_y * a.B;

    f4 acc[MR][9];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[i][t] = f4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;    // dbias partial of channel m0 + (tid % MT) over a slice of pixels (tid / MT)

    for (int tile = blockIdx.x; tile < ntiles; tile += a.split) {
        const int b = tile / (a.tiles_x * a.tiles_y), tt = tile - b * (a.tiles_x * a.tiles_y);
        const int ty = tt / a.tiles_x, tx = tt - ty * a.tiles_x;
        const int oy0 = ty * CT, ox0 = tx * CT;
        // g' tile [MT][16x16]
        for (int e = tid; e < MT * CT * CT; e += 256) {
            const int m = e / (CT * CT), rem = e - m * (CT * CT);
            const int r = rem / CT, c = rem - r * CT;
            const int yy = oy0 + r, xx = ox0 + c, co = m0 + m;
            float v = 0.f;
            if (co < a.Co && yy < H && xx < W) {
                const size_t o = (((size_t)b * a.Co + co) * H + yy) * W + xx;
                v = a.gy[o] * act_bwd(a.y[o], a.act);
            }
            gl[m * GS_ + r * CT + c] = v;
        }
        // x patch [CW][18x18] (fused upsample / concat / padding)
        for (int e = tid; e < CW * PW_ * PW_; e += 256) {
            const int kc = e / (PW_ * PW_), rem = e - kc * (PW_ * PW_);
            const int r = rem / PW_, c = rem - r * PW_;
            const int ch = c0 + kc;
            float v = 0.f;
            if (ch < Cin) {
                bool oky, okx;
                const int yy = pad_index(oy0 + r - 1, H, a.pad, oky), xx = pad_index(ox0 + c - 1, W, a.pad, okx);
                if (oky && okx && (oy0 + r - 1) <= H && (ox0 + c - 1) <= W) {
                    v = (ch < a.C0) ? a.x0[(((size_t)b * a.C0 + ch) * h0 + (yy >> a.up0)) * w0 + (xx >> a.up0)]
                                    : a.x1[(((size_t)b * a.C1 + (ch - a.C0)) * H + yy) * W + xx];
                }
            }
            xl[kc * XS_ + r * PW_ + c] = v;
        }
        __syncthreads();
        if (blockIdx.z == 0 && tid < MT * (256 / MT)) {
            const int m = tid % MT, sl = tid / MT, nsl = 256 / MT;
            for (int q = sl; q < CT * CT; q += nsl) bsum += gl[m * GS_ + q];
        }
        // K loop: the wave's 4 rows x 16 columns = 64 pixels, 4 pixels (along x) per MFMA
#pragma unroll 1
        for (int r = 0; r < 4; ++r) {
            const int row = wave * 4 + r;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = q * 4 + (lane >> 4);       // pixel of this lane's k index
                float af[MR];
#pragma unroll
                for (int i = 0; i < MR; ++i) af[i] = gl[(i * 16 + (lane & 15)) * GS_ + row * CT + col];
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int ky = t / 3, kx = t - ky * 3;
                    const float bf = xl[(lane & 15) * XS_ + (row + ky) * PW_ + col + kx];
#pragma unroll
                    for (int i = 0; i < MR; ++i)
                        acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf, acc[i][t], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    // ---- sum the four waves (fixed order) and write the slab: part[split][co][ci*9 + t]
    float* slab = a.part + (size_t)blockIdx.x * a.Co * Cin * 9;
#pragma unroll
    for (int i = 0; i < MR; ++i) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                red[wave][lane] = acc[i][t][r];
                __syncthreads();
                if (wave == 0) {
                    const float v = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
                    const int co = m0 + i * 16 + (lane >> 4) * 4 + r, ci = c0 + (lane & 15);
                    if (co < a.Co && ci < Cin) slab[((size_t)co * Cin + ci) * 9 + t] = v;
                }
                __syncthreads();
            }
        }
    }
    if (blockIdx.z == 0) {
        // dbias partial: reduce the 256/MT slices per channel through LDS (gl is free now)
        __syncthreads();
        gl[tid] = bsum;
        __syncthreads();
        if (tid < MT && m0 + tid < a.Co) {
            float v = 0.f;
            for (int sl = 0; sl < 256 / MT; ++sl) v += gl[sl * MT + tid];
            a.pbias[(size_t)blockIdx.x * a.Co + m0 + tid] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// wgrad v2 for W % 16 == 0: same MFMA core and slab protocol as conv_wgrad_kernel, but the g' tile and the x
// patch are staged with float4 loads (15 loads per thread and tile instead of 84) from the precomputed g', and
// the next tile's loads are issued before the current tile's MFMA phase.
// ------------------------------------------------------------------------------------------------
template <int MR>
__global__ __launch_bounds__(256) void conv_wgrad_v2_kernel(WgradArgs a) {
    constexpr int MT = 16 * MR;
    constexpr int NG = (MT * CT * 4 + 255) / 256;           // float4 items of the g' tile per thread
    constexpr int NXV = (CW * PW_ * 4 + 255) / 256, NXE = (CW * PW_ * 2 + 255) / 256;
    constexpr int ACC_PER_WAVE = MR * 9 * 4 * 64;                      // accumulator floats of one wave
    constexpr int SMEM = (MT * GS_ + CW * XS_) > 3 * ACC_PER_WAVE ? (MT * GS_ + CW * XS_) : 3 * ACC_PER_WAVE;
    __shared__ float smem[SMEM];
    float* gl = smem;
    float* xl = smem + MT * GS_;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * MT, c0 = blockIdx.z * CW;
    const int H = a.H, W = a.W, Cin = a.C0 + a.C1;
    const int h0 = H >> a.up0, w0 = W >> a.up0;
    const int ntiles = a.tiles_x * a.tiles_y * a.B;
    const bool upmode = a.up0 && c0 < a.C0;                  // this block's channels live in the half-res x0
    const int quads = upmode ? 2 : 4;

    f4 acc[MR][9];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[i][t] = f4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;

    float4 rg[NG], rxv[NXV];
    float rxe[NXE];
    auto prefetch = [&](int tile) {
        const int b = tile / (a.tiles_x * a.tiles_y), tt = tile - b * (a.tiles_x * a.tiles_y);
        const int ty = tt / a.tiles_x, tx = tt - ty * a.tiles_x;
        const int oy0 = ty * CT, ox0 = tx * CT;
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const int it = tid + j * 256;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (it < MT * CT * 4) {
                const int m = it / (CT * 4), rem = it - m * (CT * 4);
                const int r = rem >> 2, q = rem & 3;
                const int yy = oy0 + r, co = m0 + m;
                if (co < a.Co && yy < H) v = *reinterpret_cast<const float4*>(a.gp + (((size_t)b * a.Co + co) * H + yy) * W + ox0 + 4 * q);
            }
            rg[j] = v;
        }
#pragma unroll
        for (int j = 0; j < NXV; ++j) {
            const int it = tid + j * 256;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (it < CW * PW_ * quads) {
                const int kc = it / (PW_ * quads), rem = it - kc * (PW_ * quads);
                const int r = rem / quads, q = rem - r * quads;
                const int ch = c0 + kc;
                bool oky;
                const int yy = pad_index(oy0 + r - 1, H, a.pad, oky);
                if (ch < Cin && oky) {
                    if (upmode)
                        v = *reinterpret_cast<const float4*>(a.x0 + (((size_t)b * a.C0 + ch) * h0 + (yy >> 1)) * w0 + (ox0 >> 1) + 4 * q);
                    else if (ch < a.C0)
                        v = *reinterpret_cast<const float4*>(a.x0 + (((size_t)b * a.C0 + ch) * H + yy) * W + ox0 + 4 * q);
                    else
                        v = *reinterpret_cast<const float4*>(a.x1 + (((size_t)b * a.C1 + (ch - a.C0)) * H + yy) * W + ox0 + 4 * q);
                }
            }
            rxv[j] = v;
        }
#pragma unroll
        for (int j = 0; j < NXE; ++j) {
            const int it = tid + j * 256;
            float v = 0.f;
            if (it < CW * PW_ * 2) {
                const int kc = it / (PW_ * 2), rem = it - kc * (PW_ * 2);
                const int r = rem >> 1, side = rem & 1;
                const int ch = c0 + kc;
                bool oky, okx;
                const int yy = pad_index(oy0 + r - 1, H, a.pad, oky);
                const int xx = pad_index(side ? ox0 + CT : ox0 - 1, W, a.pad, okx);
                if (ch < Cin && oky && okx)
                    v = (ch < a.C0) ? a.x0[(((size_t)b * a.C0 + ch) * h0 + (yy >> a.up0)) * w0 + (xx >> a.up0)]
                                    : a.x1[(((size_t)b * a.C1 + (ch - a.C0)) * H + yy) * W + xx];
            }
            rxe[j] = v;
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const int it = tid + j * 256;
            if (it < MT * CT * 4) {
                const int m = it / (CT * 4), rem = it - m * (CT * 4);
                float* dst = gl + m * GS_ + (rem >> 2) * CT + (rem & 3) * 4;
                const float4 v = rg[j];
                dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
            }
        }
#pragma unroll
        for (int j = 0; j < NXV; ++j) {
            const int it = tid + j * 256;
            if (it < CW * PW_ * quads) {
                const int kc = it / (PW_ * quads), rem = it - kc * (PW_ * quads);
                const int r = rem / quads, q = rem - r * quads;
                float* dst = xl + kc * XS_ + r * PW_;
                const float4 v = rxv[j];
                if (upmode) {
                    dst += 1 + 8 * q;
                    dst[0] = v.x; dst[1] = v.x; dst[2] = v.y; dst[3] = v.y; dst[4] = v.z; dst[5] = v.z; dst[6] = v.w; dst[7] = v.w;
                } else {
                    dst += 1 + 4 * q;
                    dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NXE; ++j) {
            const int it = tid + j * 256;
            if (it < CW * PW_ * 2) {
                const int kc = it / (PW_ * 2), rem = it - kc * (PW_ * 2);
                xl[kc * XS_ + (rem >> 1) * PW_ + (rem & 1) * (PW_ - 1)] = rxe[j];
            }
        }
    };

    int tile = blockIdx.x;
    if (tile < ntiles) prefetch(tile);
    for (; tile < ntiles; tile += a.split) {
        __syncthreads();
        commit();
        __syncthreads();
        if (tile + a.split < ntiles) prefetch(tile + a.split);
        if (blockIdx.z == 0 && tid < MT * (256 / MT)) {
            const int m = tid % MT, sl = tid / MT, nsl = 256 / MT;
            for (int q = sl; q < CT * CT; q += nsl) bsum += gl[m * GS_ + q];
        }
#pragma unroll 1
        for (int r = 0; r < 4; ++r) {
            const int row = wave * 4 + r;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = q * 4 + (lane >> 4);
                float af[MR];
#pragma unroll
                for (int i = 0; i < MR; ++i) af[i] = gl[(i * 16 + (lane & 15)) * GS_ + row * CT + col];
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int ky = t / 3, kx = t - ky * 3;
                    const float bf = xl[(lane & 15) * XS_ + (row + ky) * PW_ + col + kx];
#pragma unroll
                    for (int i = 0; i < MR; ++i)
                        acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf, acc[i][t], 0, 0, 0);
                }
            }
        }
    }
    // ---- dbias partials first (they live in the g' tile area), then the four waves' accumulators are summed in
    // fixed order through LDS in one shot: waves 1-3 park theirs, wave 0 adds them to its own and writes the slab.
    __syncthreads();
    if (blockIdx.z == 0) {
        gl[tid] = bsum;
        __syncthreads();
        if (tid < MT && m0 + tid < a.Co) {
            float v = 0.f;
            for (int sl = 0; sl < 256 / MT; ++sl) v += gl[sl * MT + tid];
            a.pbias[(size_t)blockIdx.x * a.Co + m0 + tid] = v;
        }
        __syncthreads();
    }
    if (wave > 0) {
        float* dst = smem + (size_t)(wave - 1) * ACC_PER_WAVE + lane;
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[((i * 9 + t) * 4 + r) * 64] = acc[i][t][r];
    }
    __syncthreads();
    if (wave == 0) {
        float* slab = a.part + (size_t)blockIdx.x * a.Co * Cin * 9;
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = ((i * 9 + t) * 4 + r) * 64 + lane;
                    const float v = (acc[i][t][r] + smem[o]) + (smem[ACC_PER_WAVE + o] + smem[2 * ACC_PER_WAVE + o]);
                    const int co = m0 + i * 16 + (lane >> 4) * 4 + r, ci = c0 + (lane & 15);
                    if (co < a.Co && ci < Cin) slab[((size_t)co * Cin + ci) * 9 + t] = v;
                }
    }
}

// fixed-order reduction of the split-K slabs: block = 16 outputs x 16 slab groups; each thread sums its
// group's slabs in order, the 16 group sums are then added in order through LDS.
__global__ __launch_bounds__(256) void conv_wreduce_kernel(const float* part, const float* pbias, float* dw, float* db,
                                                           int split, int nW, int Co) {
    __shared__ float sm[16][17];
    const int o = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + o;
    const int per = (split + 15) / 16;
    float v = 0.f;
    if (i < nW + Co) {
        const float* src = (i < nW) ? part + i : pbias + (i - nW);
        const size_t stride = (i < nW) ? (size_t)nW : (size_t)Co;
        const int s1 = min(split, (grp + 1) * per);
        for (int s = grp * per; s < s1; ++s) v += src[(size_t)s * stride];
    }
    sm[grp][o] = v;
    __syncthreads();
    if (grp == 0 && i < nW + Co) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += sm[k][o];
        if (i < nW) { if (dw) dw[i] = t; }
        else if (db) db[i - nW] = t;
    }
}

// dbias partials for the Winograd backward path: block (co, s) sums slice s of {b, pixel} of gp[:, co]; tree in LDS
// (fixed order), the DB_SPLIT partials per channel are then added in order by conv_wreduce_kernel.
constexpr int DB_SPLIT = 8;
__global__ __launch_bounds__(256) void conv_dbias_kernel(const float* __restrict__ gp, float* __restrict__ pbias, int B, int Co, int HW) {
    __shared__ float sm[256];
    const int co = blockIdx.x, sp = blockIdx.y;
    const int total = B * HW, per = (total + DB_SPLIT - 1) / DB_SPLIT;
    const int i1 = min(total, (sp + 1) * per);
    float v = 0.f;
    for (int i = sp * per + threadIdx.x; i < i1; i += 256) {
        const int b = i / HW, q = i - b * HW;
        v += gp[((size_t)b * Co + co) * HW + q];
    }
    sm[threadIdx.x] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) pbias[(size_t)sp * Co + co] = sm[0];
}

static inline size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }
// Winograd (wino.hip, wino_wgrad.hip) or direct implicit GEMM (this file), decided per pass from measurements on the
// decoder pyramid (tools/bench_convblock.py, B=12): the forward wins at every width; the data gradient from 32 input
// channels; the weight gradient, whose blocks tile 64 x 32 channels and pay for split slabs, from 64 output channels.
// DC_CONV_WINO=0 forces the direct kernels (benchmarks only); read once.
static inline int pick_mr(int M) { return M > 32 ? 4 : (M > 16 ? 2 : 1); }
static inline int pick_mr_w(int M) { return M > 16 ? 2 : 1; }   // wgrad: LDS holds the g' tile of 16*MR channels
static inline int pick_split(int B, int H, int W, int Co, int Cin) {
    const int ntiles = ceil_div(W, CT) * ceil_div(H, CT) * B;
    const int outer = ceil_div(Co, 16 * pick_mr_w(Co)) * ceil_div(Cin, CW);
    // enough blocks to fill 256 CUs a few times over, but at least ~4 pixel tiles per block so that the
    // pipeline prologue and the slab epilogue are amortised
    int split = std::max(1, std::min(ntiles, 2048 / std::max(outer, 1)));
    split = std::min(split, std::max(1, ntiles / 4));
    return std::min(split, 512);
}

static bool wino_enabled() {
    static const bool v = [] { const char* e = getenv("DC_CONV_WINO"); return !e || atoi(e) != 0; }();
    return v;
}
// the Winograd kernels address their operands with 32-bit buffer offsets: a tensor of 2 GiB or more takes the direct kernels
static inline bool wino_fits(int B, int C0, int C1, int Co, int H, int W) {
    return (size_t)B * std::max(std::max(C0, C1), Co) * H * W * 4 < 0x7fffffffull;
}
static inline bool wino_fwd(int C0, int C1, int B, int Co, int H, int W) {
    return wino_enabled() && wino_conv_eligible(C0, C1, H, W) && wino_fits(B, C0, C1, Co, H, W);
}
static inline bool wino_gp_ok(int B, int Co, int H, int W, int act) { return act == ACT_NONE || ((size_t)B * Co * H * W) % 4 == 0; }
static inline bool wino_dx(int C0, int C1, int B, int Co, int H, int W, int act) {
    static const int min_cin = std::getenv("DC_WINO_DX_MIN") ? atoi(std::getenv("DC_WINO_DX_MIN")) : 16;      // (32 before the split store: the 16-channel level then paid the padded scratch + fold on its 192 x 640 map)
    return wino_enabled() && wino_conv_eligible(C0, C1, H, W) && C0 + C1 >= min_cin && wino_gp_ok(B, Co, H, W, act) && wino_fits(B, C0, C1, Co, H, W);
}
static inline bool wino_dw(int C0, int C1, int B, int Co, int H, int W, int act) {
    static const int min_co = std::getenv("DC_WINO_DW_MIN") ? atoi(std::getenv("DC_WINO_DW_MIN")) : 32;
    static const int min_ci = std::getenv("DC_WINO_DW_MINCI") ? atoi(std::getenv("DC_WINO_DW_MINCI")) : 32;
    return wino_enabled() && wino_conv_eligible(C0, C1, H, W) && C0 + C1 >= min_ci && Co >= min_co && wino_gp_ok(B, Co, H, W, act) &&
           wino_fits(B, C0, C1, Co, H, W);
}

// The same reduction with four consecutive weights per thread (16-byte loads: a wave reads 256 contiguous bytes of four slabs
// instead of 64; conv_wreduce_kernel ran at ~2 TB/s on the 38 MB of slabs behind every bf16 weight gradient) -- the same
// per-group order, then the 16 groups in order: bitwise the same sums.  Blocks [0, wblocks) take 64 weights each, the blocks
// after them the bias partials (16 each, the scalar form).  nW % 4 == 0, 16-byte aligned slabs.
__global__ __launch_bounds__(256) void conv_wreduce4_kernel(const float* __restrict__ part, const float* __restrict__ pbias,
                                                            float* __restrict__ dw, float* __restrict__ db, int split, int nW, int Co,
                                                            int wblocks) {
    const int o = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int per = (split + 15) / 16;
    const int s0 = grp * per, s1 = min(split, (grp + 1) * per);
    if ((int)blockIdx.x < wblocks) {
        __shared__ float4 sm4[16][17];
        const int i = (blockIdx.x * 16 + o) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < nW) {
            const float* src = part + i;
#pragma unroll 4
            for (int s = s0; s < s1; ++s) {
                const float4 t = *reinterpret_cast<const float4*>(src + (size_t)s * nW);
                v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
            }
        }
        sm4[grp][o] = v;
        __syncthreads();
        if (grp == 0 && i < nW && dw) {
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int k = 0; k < 16; ++k) { const float4 u = sm4[k][o]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
            *reinterpret_cast<float4*>(dw + i) = t;
        }
        return;
    }
    __shared__ float sm[16][17];
    const int i = ((int)blockIdx.x - wblocks) * 16 + o;
    float v = 0.f;
    if (i < Co)
        for (int s = s0; s < s1; ++s) v += pbias[(size_t)s * Co + i];
    sm[grp][o] = v;
    __syncthreads();
    if (grp == 0 && i < Co && db) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += sm[k][o];
        db[i] = t;
    }
}

int conv_wreduce(const float* part, const float* pbias, float* dw, float* db, int split, int nW, int Co, hipStream_t st) {
    if (nW > 0 && nW % 4 == 0 && part && dw && !(((size_t)part | (size_t)dw) & 15)) {
        const int wblocks = ceil_div(nW, 64), bblocks = (pbias && Co > 0) ? ceil_div(Co, 16) : 0;
        hipLaunchKernelGGL(conv_wreduce4_kernel, dim3(wblocks + bblocks), dim3(256), 0, st, part, pbias, dw, db, split, nW, Co, wblocks);
    } else {
        hipLaunchKernelGGL(conv_wreduce_kernel, dim3(ceil_div(nW + Co, 16)), dim3(256), 0, st, part, pbias, dw, db, split, nW, Co);
    }
    DC_CHECK_LAUNCH();
    return DC_OK;
}

// dc_set_dgrad_split: 1 (default; DC_DGRAD_SPLIT=0 in the environment starts with 0) = the fused blocks' Winograd data gradient is
// written without the padded-domain scratch (wino_conv_dgrad_split + conv_ring_kernel); 0 = full correlation + fold pass
int g_dgrad_split = !(std::getenv("DC_DGRAD_SPLIT") && std::getenv("DC_DGRAD_SPLIT")[0] == '0');
// Under ReflectionPad the ring costs Cin x Co x perimeter multiplies on plain FMAs while the fold pass it replaces costs Cin x H x W
// bytes: the wide shallow levels (48 x 160 and up at 192 x 640) gain, the deep 6 x 20 ... 24 x 80 levels (256 ... 512 channels) would
// pay 10-50 us of ring for a 3-8 us fold, and at batch 1 the ring's two launches cost more than a fold of < 16 MB (C1 under a
// graph 3.23 -> 3.32 ms).  Zero padding has no ring: always.
int g_dgrad_split_min_pixels = std::getenv("DC_DGRAD_SPLIT_MIN") ? atoi(std::getenv("DC_DGRAD_SPLIT_MIN")) : 6000;

// bf16 matrix-core kernels (conv_bf16.hip) instead of the fp32 ones: thread precision + shapes of their 16-byte staging
static inline bool bf16_path(int C0, int C1, int up0, int H, int W) {
    return matrix_precision() == DC_PREC_BF16 && c3b_eligible(C0, C1, up0, H, W, 1);
}

}  // namespace dc

using namespace dc;
#define ST ((hipStream_t)stream)

extern "C" size_t dc_conv3x3_fwd_workspace(int C0, int C1, int B, int Co, int H, int W) {
    if (C0 < 0 || C1 < 0 || C0 + C1 <= 0 || Co <= 0 || B <= 0 || H <= 0 || W <= 0) return 0;
    const size_t direct = std::max(al256((size_t)9 * (C0 + C1) * Co * sizeof(float)), c3b_weights_bytes(C0 + C1, Co));
    return wino_fwd(C0, C1, B, Co, H, W) ? std::max(direct, wino_conv_ws_bytes(B, C0 + C1, Co, H, W)) : direct;
}

extern "C" int dc_conv3x3_fwd(const float* x0, int C0, int up0, const float* x1, int C1, const float* weight,
                              const float* bias, float* y, void* ws, int B, int Co, int H, int W, int act,
                              int pad_mode, void* stream) {
    if (!x0 || C0 <= 0 || (C1 > 0 && !x1) || C1 < 0 || !weight || !y || !ws || B <= 0 || Co <= 0 || H < 2 || W < 2)
        return DC_EINVAL;
    if (up0 && ((H | W) & 1)) return DC_EINVAL;
    if (act < 0 || act > ACT_LAST || pad_mode < 0 || pad_mode > 1) return DC_EINVAL;
    const int Cin = C0 + C1;
    // single-channel heads: plain-FMA kernels (dispconv.hip)
    if (wino_enabled() && dispconv_eligible(C0, C1, up0 ? 1 : 0, Co, H, W))
        return dispconv_fwd(x0, weight, bias, y, B, C0, H, W, act, pad_mode, ST);
    // reduced-precision policy: direct implicit GEMM on the bf16 matrix cores (conv_bf16.hip)
    if (bf16_path(C0, C1, up0 ? 1 : 0, H, W))
        return c3b_conv(x0, C0, up0 ? 1 : 0, x1, C1, weight, Co, Cin, 0, 0, bias, y, ws, B, H, W, act, pad_mode, 1, ST);
    // even widths: fused Winograd F(2x2,3x3) (wino.hip); otherwise the direct implicit GEMM below
    if (wino_fwd(C0, C1, B, Co, H, W))
        return wino_conv_fused_fwd(x0, C0, up0 ? 1 : 0, x1, C1, weight, bias, y, ws, B, Co, H, W, act, pad_mode, ST);
    float* wf = (float*)ws;
    hipLaunchKernelGGL(conv_wprep_kernel, dim3(ceil_div(Co * Cin * 9, 256)), dim3(256), 0, ST, weight, wf,
                       (float*)nullptr, Co, Cin);
    DC_CHECK_LAUNCH();
    ConvArgs a{};
    a.x0 = x0; a.C0 = C0; a.up0 = up0 ? 1 : 0; a.x1 = x1; a.C1 = C1; a.wt = wf; a.bias = bias; a.out = y;
    a.B = B; a.Co = Co; a.H = H; a.W = W; a.act = act; a.pad = pad_mode;
    a.tiles_x = ceil_div(W, CT); a.tiles_y = ceil_div(H, CT);
    const int mr = pick_mr(Co);
    const dim3 grid(a.tiles_x * a.tiles_y, ceil_div(Co, 16 * mr), B);
    const bool fast = (W % 16 == 0) && (C1 == 0 || C0 % CK == 0);
    if (fast) {
        if (mr == 4) hipLaunchKernelGGL((conv_gemm_v2_kernel<4, false>), grid, dim3(256), 0, ST, a);
        else if (mr == 2) hipLaunchKernelGGL((conv_gemm_v2_kernel<2, false>), grid, dim3(256), 0, ST, a);
        else hipLaunchKernelGGL((conv_gemm_v2_kernel<1, false>), grid, dim3(256), 0, ST, a);
    } else {
        if (mr == 4) hipLaunchKernelGGL((conv_gemm_kernel<4, false>), grid, dim3(256), 0, ST, a);
        else if (mr == 2) hipLaunchKernelGGL((conv_gemm_kernel<2, false>), grid, dim3(256), 0, ST, a);
        else hipLaunchKernelGGL((conv_gemm_kernel<1, false>), grid, dim3(256), 0, ST, a);
    }
    DC_CHECK_LAUNCH();
    return DC_OK;
}

extern "C" size_t dc_conv3x3_bwd_workspace(int C0, int C1, int B, int Co, int H, int W) {
    if (C0 <= 0 || C1 < 0 || B <= 0 || Co <= 0 || H <= 0 || W <= 0) return 0;
    const int Cin = C0 + C1;
    const size_t nW = (size_t)Co * Cin * 9;
    const int split = std::max(pick_split(B, H, W, Co, Cin), c3b_wgrad_split(B, H, W, Co, Cin, 1));
    const size_t direct = std::max(al256(nW * 4), c3b_weights_bytes(Cin, Co)) + al256((size_t)B * Cin * (H + 2) * (W + 5) * 4) +
                          al256((size_t)split * nW * 4) + al256((size_t)std::max(split, DB_SPLIT) * Co * 4) + al256((size_t)B * Co * H * W * 4);
    // the Winograd passes keep their scratch behind the direct layout (dxpad and g' are shared)
    size_t extra = 0;
    if (wino_dx(C0, C1, B, Co, H, W, ACT_NONE)) extra += al256(wino_conv_ws_bytes(B, Cin, Co, H, W));
    if (wino_dw(C0, C1, B, Co, H, W, ACT_NONE)) extra += al256(wino_wgrad_ws_bytes(B, Cin, Co, H, W)) + al256((size_t)DB_SPLIT * Co * 4);
    return direct + extra + al256((size_t)256 * Co * 4);        // + the bias-gradient partials of conv_gprime_dbias_kernel
}

extern "C" int dc_conv3x3_bwd(const float* x0, int C0, int up0, const float* x1, int C1, const float* weight,
                              const float* y, const float* gy, float* dx0, float* dx1, float* dweight, float* dbias,
                              void* ws, int B, int Co, int H, int W, int act, int pad_mode, void* stream) {
    return dc_conv3x3_bwd_add(x0, C0, up0, x1, C1, weight, y, gy, dx0, dx1, nullptr, nullptr, dweight, dbias, ws, B, Co, H, W, act, pad_mode,
                              stream);
}

extern "C" int dc_conv3x3_bwd_add(const float* x0, int C0, int up0, const float* x1, int C1, const float* weight,
                                  const float* y, const float* gy, float* dx0, float* dx1, const float* addend0, const float* addend1,
                                  float* dweight, float* dbias, void* ws, int B, int Co, int H, int W, int act, int pad_mode,
                                  void* stream) {
    if ((addend0 && !dx0) || (addend1 && !dx1)) return DC_EINVAL;
    if (!x0 || C0 <= 0 || (C1 > 0 && !x1) || C1 < 0 || !weight || !y || !gy || !ws || B <= 0 || Co <= 0 || H < 2 || W < 2)
        return DC_EINVAL;
    if (up0 && ((H | W) & 1)) return DC_EINVAL;
    if (act < 0 || act > ACT_LAST || pad_mode < 0 || pad_mode > 1) return DC_EINVAL;
    const int Cin = C0 + C1;
    const size_t nW = (size_t)Co * Cin * 9;
    const int split = pick_split(B, H, W, Co, Cin);
    const int split_ws = std::max(split, c3b_wgrad_split(B, H, W, Co, Cin, 1));      // (the layout of dc_conv3x3_bwd_workspace)
    char* p = (char*)ws;
    float* wd = (float*)p; p += std::max(al256(nW * 4), c3b_weights_bytes(Cin, Co));
    float* dxpad = (float*)p; p += al256((size_t)B * Cin * (H + 2) * (W + 5) * 4);
    float* part = (float*)p; p += al256((size_t)split_ws * nW * 4);
    float* pbias = (float*)p; p += al256((size_t)std::max(split_ws, DB_SPLIT) * Co * 4);
    float* gpbuf = (float*)p; p += al256((size_t)B * Co * H * W * 4);
    const bool b16 = bf16_path(C0, C1, up0 ? 1 : 0, H, W) && wino_gp_ok(B, Co, H, W, act);
    // (the bf16 weight-gradient kernel tiles 64 output channels; a quarter- or half-filled tile still beats the fp32 direct
    // kernel on the thin 16 / 32-channel levels -- 879 -> 328 us for 96 -> 32 at 96 x 320, B = 36 -- DC_B16_DW_MIN restores 64 for A/Bs)
    static const int b16_dw_min = std::getenv("DC_B16_DW_MIN") ? atoi(std::getenv("DC_B16_DW_MIN")) : 16;
    const bool b16_dw = b16 && Co >= b16_dw_min;
    const bool w_dx = !b16 && (dx0 || dx1) && wino_dx(C0, C1, B, Co, H, W, act);
    const bool w_dw = !b16 && dweight && wino_dw(C0, C1, B, Co, H, W, act);
    // bias gradient from the same pass that forms g' (whole float4s per channel plane)
    const bool fused_db = dbias && (w_dw || b16_dw) && (H * W) % 4 == 0;
    void* wws = p; if (wino_dx(C0, C1, B, Co, H, W, ACT_NONE)) p += al256(wino_conv_ws_bytes(B, Cin, Co, H, W));
    void* gws = p; if (wino_dw(C0, C1, B, Co, H, W, ACT_NONE)) p += al256(wino_wgrad_ws_bytes(B, Cin, Co, H, W));
    float* pb2 = (float*)p; if (wino_dw(C0, C1, B, Co, H, W, ACT_NONE)) p += al256((size_t)DB_SPLIT * Co * 4);
    float* pbd = (float*)p;
    const int tiles_x = ceil_div(W, CT), tiles_y = ceil_div(H, CT);
    // fast path (v2 kernels): full 16-wide tiles, 16-byte aligned rows, chunks that do not straddle the concat
    const bool fast = (W % 16 == 0) && (C1 == 0 || C0 % CK == 0) && (C1 == 0 || C0 % CW == 0);
    // the thin single-channel heads have their own plain-FMA gradients (dispconv.hip), which form g' on the fly
    const bool head_dx = dx0 && wino_enabled() && dispconv_eligible(C0, C1, up0 ? 1 : 0, Co, H, W);
    const bool head_dw = (dweight || dbias) && wino_enabled() && dispconv_wgrad_eligible(C0, C1, up0 ? 1 : 0, Co, H, W) &&
                         dispconv_wgrad_scratch(B, C0, H, W) <= al256((size_t)B * Cin * (H + 2) * (W + 5) * 4);
    const bool gp_unused = (head_dx || !(dx0 || dx1)) && (head_dw || !(dweight || dbias));
    const float* gp = gy;
    if (gp_unused) {
    } else if (fused_db) {
        hipLaunchKernelGGL(conv_gprime_dbias_kernel, dim3(Co, gpd_split(Co)), dim3(256), 0, ST, gy, y, gpbuf, pbd, B, Co, H * W / 4, act,
                           gpd_split(Co));
        DC_CHECK_LAUNCH();
        if (act != ACT_NONE) gp = gpbuf;
    } else if ((fast || w_dx || w_dw || b16) && act != ACT_NONE && ((size_t)B * Co * H * W) % 4 == 0) {
        const size_t n4 = (size_t)B * Co * H * W / 4;
        hipLaunchKernelGGL(conv_gprime_kernel, dim3((unsigned)std::min<size_t>((n4 + 255) / 256, 4096)), dim3(256), 0, ST, gy, y,
                           gpbuf, n4, act);
        DC_CHECK_LAUNCH();
        gp = gpbuf;
    }
    if (head_dx) {
        // thin single-channel head: folded-window data gradient, no padded scratch / fold pass (dispconv.hip)
        const int rc = dispconv_dx(weight, y, gy, dx0, addend0, B, C0, H, W, act, pad_mode, ST);
        if (rc != DC_OK) return rc;
    } else if (b16 && (dx0 || dx1)) {
        // bf16 matrix cores: a zero-padded single-source block gets its data gradient directly (convolution of g' with the
        // rotated, transposed filter); reflection / upsample / concat go through the padded domain and the fold below
        if (pad_mode == PAD_ZERO && !up0 && C1 == 0) {
            const int rc = c3b_conv(gp, Co, 0, nullptr, 0, weight, Co, Cin, 1, 0, nullptr, dx0, wd, B, H, W, ACT_NONE, PAD_ZERO, 1, ST, addend0);
            if (rc != DC_OK) return rc;
        } else {
            const int rc = c3b_conv(gp, Co, 0, nullptr, 0, weight, Co, Cin, 1, 1, nullptr, dxpad, wd, B, H, W, ACT_NONE, PAD_ZERO, 1, ST);
            if (rc != DC_OK) return rc;
            const int rcf = conv_fold(dxpad, dx0, dx1, B, C0, C1, up0 ? 1 : 0, H, W, pad_mode, c3b_dpad_pitch(W), addend0, addend1, ST);
            if (rcf != DC_OK) return rcf;
        }
    } else if (w_dx && g_dgrad_split && wino_dgrad_split_ok(B, C0, C1, up0 ? 1 : 0, Co, H, W) &&
               (pad_mode == PAD_ZERO || (Co <= RING_MAXCO && (g_dgrad_split == 2 || (H * W >= g_dgrad_split_min_pixels &&
                                                                                  (size_t)B * Cin * H * W >= ((size_t)4 << 20)))))) {
        // the interior of the correlation written straight to dx0 / dx1 (concat split, 2 x 2 sums of the upsampled half and the
        // addends in the Winograd kernel's store epilogue), then the few ring terms ReflectionPad folds back: no padded-domain
        // scratch (B x Cin x (H+2) x (W+2) written and read again) and no fold pass
        const int rc = wino_conv_dgrad_split(gp, weight, dx0, dx1, addend0, addend1, wws, B, C0, C1, up0 ? 1 : 0, Co, H, W, ST);
        if (rc != DC_OK) return rc;
        if (pad_mode == PAD_REFLECT) {
            // (scratch: the padded-domain buffer, which this path does not use -- 4 strips of max(H, W) + 2 floats per plane)
            const int LP = std::max(H, W) + 2;
            const int segs = ceil_div(LP, RING_SEG), nchunk = ceil_div(Cin, RING_CB);
            const int cgroups = std::max(1, std::min(nchunk, ceil_div(512, segs * 4 * B)));       // ~two blocks per CU
            hipLaunchKernelGGL(conv_ring_strips_kernel, dim3(segs, 4 * cgroups, B), dim3(256),
                               (size_t)(Co * (RING_SEG + 4) + Co * RING_CB * 3) * sizeof(float), ST, gp, weight, dxpad, Cin, Co, H, W, LP, cgroups);
            DC_CHECK_LAUNCH();
            hipLaunchKernelGGL(conv_ring_kernel, dim3(ceil_div(2 * W + 2 * H, 256), Cin, B), dim3(256), 0, ST, (const float*)dxpad, dx0, dx1,
                               C0, C1, up0 ? 1 : 0, H, W, LP);
            DC_CHECK_LAUNCH();
        }
    } else if (w_dx) {
        // full correlation of g' with the rotated weights in the Winograd domain, then the same fold as below
        const int rc = wino_conv_full_dgrad(gp, weight, dxpad, wws, B, Cin, Co, H, W, ST);
        if (rc != DC_OK) return rc;
        const int rcf = conv_fold(dxpad, dx0, dx1, B, C0, C1, up0 ? 1 : 0, H, W, pad_mode, W + 2, addend0, addend1, ST);
        if (rcf != DC_OK) return rcf;
    } else if (dx0 || dx1) {
        hipLaunchKernelGGL(conv_wprep_kernel, dim3(ceil_div((int)nW, 256)), dim3(256), 0, ST, weight, (float*)nullptr, wd,
                           Co, Cin);
        DC_CHECK_LAUNCH();
        ConvArgs a{};
        a.C0 = C0; a.C1 = C1; a.up0 = up0 ? 1 : 0; a.wt = wd; a.y = y; a.gy = gy; a.gp = gp; a.out = dxpad;
        a.B = B; a.Co = Co; a.H = H; a.W = W; a.act = act; a.pad = pad_mode;
        a.tiles_x = ceil_div(W + 2, CT); a.tiles_y = ceil_div(H + 2, CT);
        const int mr = pick_mr(Cin);
        const dim3 grid(a.tiles_x * a.tiles_y, ceil_div(Cin, 16 * mr), B);
        if (fast) {
            if (mr == 4) hipLaunchKernelGGL((conv_gemm_v2_kernel<4, true>), grid, dim3(256), 0, ST, a);
            else if (mr == 2) hipLaunchKernelGGL((conv_gemm_v2_kernel<2, true>), grid, dim3(256), 0, ST, a);
            else hipLaunchKernelGGL((conv_gemm_v2_kernel<1, true>), grid, dim3(256), 0, ST, a);
        } else {
            if (mr == 4) hipLaunchKernelGGL((conv_gemm_kernel<4, true>), grid, dim3(256), 0, ST, a);
            else if (mr == 2) hipLaunchKernelGGL((conv_gemm_kernel<2, true>), grid, dim3(256), 0, ST, a);
            else hipLaunchKernelGGL((conv_gemm_kernel<1, true>), grid, dim3(256), 0, ST, a);
        }
        DC_CHECK_LAUNCH();
        const int rcf = conv_fold(dxpad, dx0, dx1, B, C0, C1, up0 ? 1 : 0, H, W, pad_mode, W + 2, addend0, addend1, ST);
        if (rcf != DC_OK) return rcf;
    }
    if (head_dw) {
        // (scratch: the padded-domain buffer -- the head's data gradient does not use it, and any other data-gradient path has
        // finished with it in stream order)
        const int rc = dispconv_wgrad(x0, y, gy, dweight, dbias, dxpad, B, C0, H, W, act, pad_mode, ST);
        if (rc != DC_OK) return rc;
    } else if (b16_dw && (dweight || dbias)) {
        if (dweight) {
            const int sp = c3b_wgrad_split(B, H, W, Co, Cin, 1);
            int rc = c3b_wgrad(x0, C0, up0 ? 1 : 0, x1, C1, gp, part, sp, B, Co, H, W, pad_mode, 1, ST);
            if (rc != DC_OK) return rc;
            rc = conv_wreduce(part, nullptr, dweight, nullptr, sp, (int)nW, 0, ST);
            if (rc != DC_OK) return rc;
        }
        if (dbias && fused_db) {
            const int rc = conv_wreduce(nullptr, pbd, nullptr, dbias, gpd_split(Co), 0, Co, ST);
            if (rc != DC_OK) return rc;
        } else if (dbias) {
            hipLaunchKernelGGL(conv_dbias_kernel, dim3(Co, DB_SPLIT), dim3(256), 0, ST, gp, pbias, B, Co, H * W);
            DC_CHECK_LAUNCH();
            const int rc = conv_wreduce(nullptr, pbias, nullptr, dbias, DB_SPLIT, 0, Co, ST);
            if (rc != DC_OK) return rc;
        }
    } else if (w_dw) {
        const int rc = wino_wgrad_fused(x0, C0, up0 ? 1 : 0, x1, C1, pad_mode, gp, dweight, gws, B, Co, H, W, ST);
        if (rc != DC_OK) return rc;
        if (dbias && fused_db) {
            const int rc2 = conv_wreduce(nullptr, pbd, nullptr, dbias, gpd_split(Co), 0, Co, ST);
            if (rc2 != DC_OK) return rc2;
        } else if (dbias) {
            hipLaunchKernelGGL(conv_dbias_kernel, dim3(Co, DB_SPLIT), dim3(256), 0, ST, gp, pb2, B, Co, H * W);
            DC_CHECK_LAUNCH();
            hipLaunchKernelGGL(conv_wreduce_kernel, dim3(ceil_div(Co, 16)), dim3(256), 0, ST, (const float*)nullptr, pb2,
                               (float*)nullptr, dbias, DB_SPLIT, 0, Co);
            DC_CHECK_LAUNCH();
        }
    } else if (dweight || dbias) {
        WgradArgs g{};
        g.x0 = x0; g.C0 = C0; g.up0 = up0 ? 1 : 0; g.x1 = x1; g.C1 = C1; g.y = y; g.gy = gy; g.gp = gp; g.part = part; g.pbias = pbias;
        g.B = B; g.Co = Co; g.H = H; g.W = W; g.act = act; g.pad = pad_mode;
        g.tiles_x = tiles_x; g.tiles_y = tiles_y; g.split = split;
        const int mr = pick_mr_w(Co);
        const dim3 grid(split, ceil_div(Co, 16 * mr), ceil_div(Cin, CW));
        if (fast) {
            if (mr == 2) hipLaunchKernelGGL((conv_wgrad_v2_kernel<2>), grid, dim3(256), 0, ST, g);
            else hipLaunchKernelGGL((conv_wgrad_v2_kernel<1>), grid, dim3(256), 0, ST, g);
        } else {
            if (mr == 2) hipLaunchKernelGGL((conv_wgrad_kernel<2>), grid, dim3(256), 0, ST, g);
            else hipLaunchKernelGGL((conv_wgrad_kernel<1>), grid, dim3(256), 0, ST, g);
        }
        DC_CHECK_LAUNCH();
        if (dweight) {
            const int rc = conv_wreduce(part, pbias, dweight, dbias, split, (int)nW, Co, ST);
            if (rc != DC_OK) return rc;
        } else {
            hipLaunchKernelGGL(conv_wreduce_kernel, dim3(ceil_div((int)nW + Co, 16)), dim3(256), 0, ST, part, pbias, dweight,
                               dbias, split, (int)nW, Co);
            DC_CHECK_LAUNCH();
        }
    }
    return DC_OK;
}

extern "C" int dc_set_dgrad_split(int mode) {
    if (mode < 0 || mode > 2) return DC_EINVAL;
    const int prev = dc::g_dgrad_split;
    dc::g_dgrad_split = mode;
    return prev;
}
