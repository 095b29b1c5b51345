// Winograd F(2x2, 3x3) convolution on the fp32 matrix cores, one fused kernel (gfx950).
//
// Replaces, for the ResNet trunks' stride-1 3x3 convolutions (reference networks/resnet_encoder.py:74-98, i.e.
// torchvision BasicBlock/Bottleneck conv3x3 with padding=1, bias=False), the library call behind nn.Conv2d:
//
//     y[b,m,Y,X] = sum_k sum_{ky,kx} w[m,k,ky,kx] * x[b,k,Y+ky-1,X+kx-1]          (zero padding)
//
// as  Y_tile = A^T [ sum_k (G g G^T) .* (B^T d B) ] A  over 2x2 output tiles (Lavin & Gray 2015):
// 16 independent GEMMs  M_p[m][tile] = sum_k U_p[m][k] V_p[k][tile],  2.25x fewer multiplies than direct.
//
// Mapping to CDNA4
//   * v_mfma_f32_16x16x4_f32: rows = 16 output channels, cols = 16 tiles, K = 4 input channels; one wave owns
//     16 tiles (a RH x RW patch of tiles, chosen per layer so that the map divides evenly) x 16*MR channels x all
//     16 Winograd positions = 64*MR accumulator registers.
//   * the input transform B^T d B is computed by the lane that owns B[k = lane>>4][tile = lane&15] straight from
//     the wave's private LDS slab of the raw input (4 rows x 2 ds_read_b64, 32 adds per 16 MFMA operands);
//     nothing transformed ever touches HBM.
//   * the transformed weights U are produced once per call by wino_weights_kernel in exactly the order the
//     block stages them: [m-block][k-chunk][k][p/4][m][p%4], so staging is a linear 128-bit copy and the A
//     operands for 4 positions arrive with one conflict-free ds_read_b128.
//   * the output transform A^T M A happens in registers (a lane's 16 positions of one (m, tile) are 16 of its
//     accumulators), then one 8-byte store per output row.
//   * global -> register -> LDS double buffering: chunk c+1's loads (1 buffer_load_b64 of x per channel and lane,
//     MR b128 of U) are in flight during chunk c's MFMAs; 2 blocks per CU.
// dgrad is the same kernel on the 180-degree-rotated, transposed weights (wino_weights_kernel<DGRAD>).
// Requires even W (8-byte row alignment); H arbitrary.
#include "dc_common.h"

#include <algorithm>
#include <cstdlib>

namespace dc {

using f4 = __attribute__((ext_vector_type(4))) float;
using f2w = __attribute__((ext_vector_type(2))) float;
using wrsrc_t = __amdgpu_buffer_rsrc_t;

constexpr int WSLAB = 176;            // floats per channel of a wave's input slab (max over the region shapes)

struct WinoArgs {
    const float* x; const float* uhat; float* y;
    int B, K, M, H, W;                // K = reduction channels, M = output channels
    int RH, RW, RS;                   // wave region in tiles, LDS row stride (floats)
    int regs_x, regs_y, nreg;         // regions per image / total
    int nchunks;
    unsigned xbytes;
    int dbg;                          // timing experiments only (DC_WINO_DBG): 1 no loads, 2 no commit/barriers, 4 no MFMA phase
};

// U = G g G^T for every (m, k), written in staging order.  grid over padded (Mp x Kp); one thread per (m, k).
template <bool DGRAD>
__global__ __launch_bounds__(256) void wino_weights_kernel(const float* __restrict__ w, float* __restrict__ uhat,
                                                           int Co, int Ci, int MT, int Mp, int Kp, int WK) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= Mp * Kp) return;
    const int m = idx / Kp, k = idx - m * Kp;
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
    float t[4][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        t[0][j] = g[0][j];
        t[1][j] = 0.5f * (g[0][j] + g[1][j] + g[2][j]);
        t[2][j] = 0.5f * (g[0][j] - g[1][j] + g[2][j]);
        t[3][j] = g[2][j];
    }
    float u[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        u[i][0] = t[i][0];
        u[i][1] = 0.5f * (t[i][0] + t[i][1] + t[i][2]);
        u[i][2] = 0.5f * (t[i][0] - t[i][1] + t[i][2]);
        u[i][3] = t[i][2];
    }
    const int mb = m / MT, mi = m - mb * MT, kc = k / WK, kin = k - kc * WK;
    const int nchunks = Kp / WK;
    float* dst = uhat + ((((size_t)mb * nchunks + kc) * WK + kin) * 4) * MT * 4 + (size_t)mi * 4;
#pragma unroll
    for (int pq = 0; pq < 4; ++pq)
        *reinterpret_cast<float4*>(dst + (size_t)pq * MT * 4) = make_float4(u[pq][0], u[pq][1], u[pq][2], u[pq][3]);
}

// MR = 16-channel output blocks per wave, WK = reduction channels per staged chunk (WK/4 MFMA k-steps)
template <int MR, int WK>
__global__ __launch_bounds__(256, 2) void wino_conv_kernel(WinoArgs a) {
    constexpr int MT = 16 * MR;
    constexpr int UF4 = WK * 4 * MT;                  // float4 items of one U chunk
    constexpr int NU = UF4 / 256;                     // per thread
    __shared__ f4 ul[2][UF4];
    __shared__ float xl[4][WK * WSLAB];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kk = lane >> 4;
    const int H = a.H, W = a.W, RH = a.RH, RW = a.RW, RS = a.RS;
    const int PS = (2 * RH + 2) * RS;                 // slab plane stride
    const int mblk = blockIdx.y;

    // ---- this wave's region
    const int region = blockIdx.x * 4 + wave;
    const bool active = region < a.nreg;
    const int per_img = a.regs_x * a.regs_y;
    const int rr = active ? region : 0;
    const int b = rr / per_img, rq = rr - b * per_img;
    const int ry = rq / a.regs_x, rx = rq - ry * a.regs_x;
    const int Y0 = ry * RH * 2, X0 = rx * RW * 2;

    // ---- staging role of this lane: one (row, column pair) of every channel plane of the slab
    const int PR = RW + 2;                            // pairs per slab row
    const int sr = lane / PR, scp = lane - sr * PR;
    const int iy = Y0 - 1 + sr, ix = X0 - 2 + 2 * scp;
    const bool sok = active && sr < 2 * RH + 2 && iy >= 0 && iy < H && ix >= 0 && ix < W;
    const bool swr = sr < 2 * RH + 2;                 // lanes past the slab do not write
    const unsigned svoff = sok ? (unsigned)(iy * W + ix) * 4u : 0x80000000u;
    const int slds0 = sr * RS + max(2 * scp - 1, 0), slds1 = sr * RS + 2 * scp;
    const wrsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), (short)0, (int)a.xbytes, 0x00020000);
    const unsigned plane = (unsigned)(H * W) * 4u;
    const unsigned img0 = (unsigned)b * (unsigned)a.K * plane;
    float* xw = xl[wave];

    // ---- compute role: tile n of the region, reduction channel kk of each k-step
    const bool tile_in = n < RH * RW;
    const int tn = tile_in ? n : 0;
    const int ty = tn / RW, tx = tn - ty * RW;
    const int rd0 = kk * PS + 2 * ty * RS + 2 * tx;

    f4 acc[MR][16];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int p = 0; p < 16; ++p) acc[i][p] = f4{0.f, 0.f, 0.f, 0.f};

    f2w px[WK];
    f4 pu[NU];
    const f4* ug = reinterpret_cast<const f4*>(a.uhat) + (size_t)mblk * a.nchunks * UF4;
    auto prefetch = [&](int c) {
#pragma unroll
        for (int j = 0; j < NU; ++j) pu[j] = ug[(size_t)c * UF4 + tid + j * 256];
#pragma unroll
        for (int k = 0; k < WK; ++k) {
            const int ch = c * WK + k;
            const unsigned vo = ch < a.K ? svoff : 0x80000000u;
            px[k] = __builtin_bit_cast(f2w, __builtin_amdgcn_raw_buffer_load_b64(xr, (int)vo, (int)(img0 + (unsigned)ch * plane), 0));
        }
    };
    auto commit = [&](f4* udst) {
#pragma unroll
        for (int j = 0; j < NU; ++j) udst[tid + j * 256] = pu[j];
        if (swr) {
#pragma unroll
            for (int k = 0; k < WK; ++k) {
                xw[k * PS + slds0] = px[k].x;          // (pair 0: the discarded column lands on, and is overwritten by, .y)
                xw[k * PS + slds1] = px[k].y;
            }
        }
    };

    // One chunk's MFMA phase, software-pipelined by hand: group g = (k-step g>>2, positions 4*(g&3)..+3); the A
    // operands of group g+1 and the raw patch of the next k-step are requested before group g's MFMAs issue.
    auto compute = [&](const f4* ucur) {
        const f4* up = ucur + (kk * 4) * MT + n;
        float d[4][4], v[16];
        f4 ua[2][MR];
        auto read_patch = [&](int ks) {
            const float* src = xw + ks * 4 * PS + rd0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f2w lo = *reinterpret_cast<const f2w*>(src + i * RS);
                const f2w hi = *reinterpret_cast<const f2w*>(src + i * RS + 2);
                d[i][0] = lo.x; d[i][1] = lo.y; d[i][2] = hi.x; d[i][3] = hi.y;
            }
        };
        auto transform = [&]() {
            float t[4][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                t[0][j] = d[0][j] - d[2][j];
                t[1][j] = d[1][j] + d[2][j];
                t[2][j] = d[2][j] - d[1][j];
                t[3][j] = d[1][j] - d[3][j];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[i * 4 + 0] = t[i][0] - t[i][2];
                v[i * 4 + 1] = t[i][1] + t[i][2];
                v[i * 4 + 2] = t[i][2] - t[i][1];
                v[i * 4 + 3] = t[i][1] - t[i][3];
            }
        };
        read_patch(0);
#pragma unroll
        for (int i = 0; i < MR; ++i) ua[0][i] = up[i * 16];
        transform();
#pragma unroll
        for (int g = 0; g < WK; ++g) {                // WK/4 k-steps x 4 position groups
            const int ks = g >> 2, pq = g & 3;
            if (g + 1 < WK) {
                const int ks1 = (g + 1) >> 2, pq1 = (g + 1) & 3;
#pragma unroll
                for (int i = 0; i < MR; ++i) ua[(g + 1) & 1][i] = up[((ks1 * 16 + pq1) * MT) + i * 16];
            }
            if (pq == 0 && ks + 1 < WK / 4) read_patch(ks + 1);        // d is dead once v exists
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MR; ++i) {
                const f4 u = ua[g & 1][i];
                acc[i][pq * 4 + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(u.x, v[pq * 4 + 0], acc[i][pq * 4 + 0], 0, 0, 0);
                acc[i][pq * 4 + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(u.y, v[pq * 4 + 1], acc[i][pq * 4 + 1], 0, 0, 0);
                acc[i][pq * 4 + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(u.z, v[pq * 4 + 2], acc[i][pq * 4 + 2], 0, 0, 0);
                acc[i][pq * 4 + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(u.w, v[pq * 4 + 3], acc[i][pq * 4 + 3], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (pq == 3 && ks + 1 < WK / 4) transform();
        }
    };

    // U is double-buffered in LDS, the x slab is wave-private: one barrier per chunk.
    prefetch(0);
    commit(ul[0]);
    if (a.nchunks > 1) prefetch(1);
    __syncthreads();
    for (int c = 0; c < a.nchunks; ++c) {
        if (!(a.dbg & 4)) compute(ul[c & 1]);
        if (c + 1 < a.nchunks) {
            if (!(a.dbg & 2)) commit(ul[(c + 1) & 1]);
            if (c + 2 < a.nchunks && !(a.dbg & 1)) prefetch(c + 2);
        }
        __syncthreads();
    }

    // ---- output transform Y = A^T M A in registers; D layout: row m = kk*4 + r, column = tile n
    if (!active || !tile_in) return;
    const int oy = Y0 + 2 * ty, ox = X0 + 2 * tx;
    if (oy >= H || ox >= W) return;
#pragma unroll
    for (int i = 0; i < MR; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = mblk * MT + i * 16 + kk * 4 + r;
            if (m >= a.M) continue;
            float s0[4], s1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s0[j] = acc[i][0 + j][r] + acc[i][4 + j][r] + acc[i][8 + j][r];
                s1[j] = acc[i][4 + j][r] - acc[i][8 + j][r] - acc[i][12 + j][r];
            }
            float* dst = a.y + (((size_t)b * a.M + m) * H + oy) * W + ox;
            *reinterpret_cast<f2w*>(dst) = f2w{s0[0] + s0[1] + s0[2], s0[1] - s0[2] - s0[3]};
            if (oy + 1 < H) *reinterpret_cast<f2w*>(dst + W) = f2w{s1[0] + s1[1] + s1[2], s1[1] - s1[2] - s1[3]};
        }
    }
}

// region shape per layer: the candidate with the least padding waste (ties: the widest, for longer store runs)
static void wino_pick_region(int TH, int TW, int& RH, int& RW, int& RS) {
    const int cand[3][3] = {{2, 8, 24}, {4, 4, 12}, {3, 5, 22}};
    double best = -1.0;
    for (auto& c : cand) {
        const double covered = (double)ceil_div(TH, c[0]) * ceil_div(TW, c[1]) * 16.0;
        const double util = (double)TH * TW / covered;
        if (util > best + 1e-9) { best = util; RH = c[0]; RW = c[1]; RS = c[2]; }
    }
}

static inline size_t wino_al256(size_t v) { return (v + 255) & ~(size_t)255; }
static inline int wino_pick_mr(int M, int nreg) {
    if (M <= 16) return 1;
    // enough blocks for 256 CUs x 2: fall back to 16-channel blocks on the small maps
    return ((long)ceil_div(nreg, 4) * ceil_div(M, 32) >= 512) ? 2 : 1;
}

static int wino_run(const float* x, const float* w, float* y, void* ws, int B, int Ci, int Co, int H, int W, bool dgrad,
                    hipStream_t st) {
    if (!x || !w || !y || !ws || B <= 0 || Ci <= 0 || Co <= 0 || H < 1 || W < 2 || (W & 1)) return DC_EINVAL;
    const int K = dgrad ? Co : Ci, M = dgrad ? Ci : Co;
    const size_t xb = (size_t)B * K * H * W * 4;
    if (xb >= 0x7fffffffull) return DC_EINVAL;
    WinoArgs a{};
    a.x = x; a.uhat = (const float*)ws; a.y = y; a.B = B; a.K = K; a.M = M; a.H = H; a.W = W;
    const int TH = ceil_div(H, 2), TW = W / 2;
    wino_pick_region(TH, TW, a.RH, a.RW, a.RS);
    a.regs_x = ceil_div(TW, a.RW); a.regs_y = ceil_div(TH, a.RH); a.nreg = a.regs_x * a.regs_y * B;
    a.xbytes = (unsigned)xb;
    { const char* e = getenv("DC_WINO_DBG"); a.dbg = e ? atoi(e) : 0; }
    int mr = wino_pick_mr(M, a.nreg);
    { const char* e = getenv("DC_WINO_MR"); if (e && M > 16) mr = atoi(e) == 2 ? 2 : 1; }
    const int MT = 16 * mr;
    const int WK = 8;
    a.nchunks = ceil_div(K, WK);
    const int Mp = ceil_div(M, MT) * MT, Kp = a.nchunks * WK;
    if (dgrad)
        hipLaunchKernelGGL((wino_weights_kernel<true>), dim3(ceil_div(Mp * Kp, 256)), dim3(256), 0, st, w, (float*)ws, Co, Ci, MT, Mp, Kp, WK);
    else
        hipLaunchKernelGGL((wino_weights_kernel<false>), dim3(ceil_div(Mp * Kp, 256)), dim3(256), 0, st, w, (float*)ws, Co, Ci, MT, Mp, Kp, WK);
    DC_CHECK_LAUNCH();
    const dim3 grid(ceil_div(a.nreg, 4), Mp / MT);
    if (mr == 2) hipLaunchKernelGGL((wino_conv_kernel<2, 8>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((wino_conv_kernel<1, 8>), grid, dim3(256), 0, st, a);
    DC_CHECK_LAUNCH();
    return DC_OK;
}

}  // namespace dc

using namespace dc;

extern "C" size_t dc_wino3x3_workspace(int Ci, int Co) {
    if (Ci <= 0 || Co <= 0) return 0;
    const size_t a = (size_t)ceil_div(Ci, 32) * 32, b = (size_t)ceil_div(Co, 32) * 32;
    return wino_al256(a * b * 16 * sizeof(float));
}

extern "C" int dc_wino3x3_fwd(const float* x, const float* weight, float* y, void* ws, int B, int Ci, int Co, int H, int W,
                              void* stream) {
    return wino_run(x, weight, y, ws, B, Ci, Co, H, W, false, (hipStream_t)stream);
}

extern "C" int dc_wino3x3_dgrad(const float* gy, const float* weight, float* gx, void* ws, int B, int Ci, int Co, int H,
                                int W, void* stream) {
    return wino_run(gy, weight, gx, ws, B, Ci, Co, H, W, true, (hipStream_t)stream);
}
