// Winograd F(4x4, 3x3) convolution on the fp32 matrix cores (gfx950): the large-map variant of wino.hip.
//
// Same op as wino_ps_kernel's plain form -- the ResNet trunks' stride-1 3x3 convolutions with zero padding (reference
// networks/resnet_encoder.py:74-98 via torchvision BasicBlock / Bottleneck), forward and, on the rotated transposed filter,
// data gradient --
//
//     y[b,m,Y,X] = sum_k sum_{ky,kx} w[m,k,ky,kx] * x[b,k,Y+ky-1,X+kx-1]
//
// as  Y_tile = A^T [ sum_k (G g G^T) .* (B^T d B) ] A  over 4x4 output tiles and 6x6 patches (Lavin & Gray 2015): 36
// independent GEMMs, 2.25 multiplies per output against F(2x2,3x3)'s 4 and the direct convolution's 9 -- 0.5625 of the
// matrix-core work of wino.hip.  The price is accuracy: the transforms' constants span 1/24 .. 8, and fp32 products are
// summed with those weights.  Measured against an fp64 direct convolution on trunk-sized layers: relative L2 error 1.1-1.4e-6
// (F(2x2,3x3) and a direct fp32 sum: 2.1-2.8e-7), worst element 5-8e-6 of the output's maximum.  The contract of the path is
// 1e-3 (BASELINE north_star); tests/test_wino_gpu.py states the bound this kernel is held to.
//
// Mapping to CDNA4 (wino4_kernel)
//   * v_mfma_f32_16x16x4_f32: rows = 16 output channels, cols = 16 tiles, K = 4 reduction channels.  A WAVE owns all 36
//     positions of a 16-channel x 16-tile block: 36 accumulator tiles (144 VGPRs), so the output transform Y = A^T M A happens
//     in registers and a lane stores finished 16-byte rows -- no exchange between waves, no row-combine epilogue.
//   * a block = 4 waves = 4 tile GROUPS (<= 16 tiles each, GH x GW chosen per map: 3x5, 2x8, 4x4 ...) of the same 16-channel
//     block; the waves share the transformed weights of a reduction step through LDS.
//   * per reduction step (4 channels = one MFMA k-step): the raw 6x6 patches of the block's groups and the step's U slice are
//     staged in LDS (double-buffered, one barrier per step, the next step's global loads in flight in registers); a lane
//     reads ITS tile's patch (6 rows x {b128, b64}), forms B^T d B in registers (row pass as the rows arrive, column pass per
//     Winograd row) and issues the 36 MFMAs.
//   * ONE block per CU (one wave per SIMD, the whole 512-entry register file: 144 accumulators + a step's patches, weights and
//     the next step's 10 in-flight 16-byte loads do not fit the 256 of a two-wave schedule): a wave has to hide its own ~170
//     vector instructions and 21 LDS reads per step behind its own 36 MFMAs (1152 cycles of matrix pipe).
//   * K-split over gridDim.z with fixed-order slab sums for launches that would not fill the chip (wino_ysum_kernel).
#include "dc_common.h"
#include "wino.h"
#include "wino4.h"

#include <algorithm>

namespace dc {

using f4 = __attribute__((ext_vector_type(4))) float;
using f2w = __attribute__((ext_vector_type(2))) float;
using wrsrc_t = __amdgpu_buffer_rsrc_t;

__global__ __launch_bounds__(256) void wino4_weights_kernel(const float* __restrict__ w, float* __restrict__ uhat, int Co, int Ci,
                                                            int Mp, int Kp, int dgrad) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (dgrad) wino4_weight_one<true>(w, uhat, idx, Co, Ci, Mp, Kp);
    else wino4_weight_one<false>(w, uhat, idx, Co, Ci, Mp, Kp);
}

// ------------------------------------------------------------------------------------------------
constexpr int W4_ITEMS = 7;           // 16-byte staging loads per thread and step: 16 (group, channel) patches x PH x (GW + 2) quads <= 7 * 256
constexpr int W4_USTEP = 4 * 36 * 16; // floats of one (m-block, step) U slice
constexpr int W4_XMAX = 6528;         // floats of one step's patches in LDS: 16 x PH x PS <= 16 x 408 (1 x 16 groups)

struct Wino4Args {
    const float* x; const float* uhat; float* y; const float* addend;
    int B, K, M, H, W;
    int GH, GW, gx, gy, ngroups;      // group shape in tiles; groups per row / column of a map; groups in the batch
    int PH, NQ, PS, PLANE;            // patch rows (4 GH + 2), quads per patch row (GW + 2), LDS row stride (4 GW + 4), floats per (group, channel)
    int nsteps, steps_per_split;
    unsigned xbytes;
    size_t slab_stride;
    int mblocks, tblocks, m_fast;
    unsigned mg_mblocks, mg_tblocks, mg_per_img, mg_gx, mg_GW, mg_NQ, mg_PHNQ;
};

__global__ __launch_bounds__(256, 1) void wino4_kernel(Wino4Args a) {
    extern __shared__ __attribute__((aligned(16))) float w4lds[];     // [2][16 PLANE] patches, then [2][W4_USTEP] weights
    float* const xl[2] = {w4lds, w4lds + 16 * a.PLANE};
    float* const ul[2] = {w4lds + 32 * a.PLANE, w4lds + 32 * a.PLANE + W4_USTEP};
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kk = lane >> 4;
    const int H = a.H, W = a.W, GW = a.GW, PS = a.PS, PLANE = a.PLANE;
    const int lbid = xcd_logical_block(blockIdx.x, gridDim.x);
    const int q_m = fdiv(lbid, a.mg_mblocks), q_t = fdiv(lbid, a.mg_tblocks);
    const int mblk = a.m_fast ? lbid - q_m * a.mblocks : q_t;
    const int tblk = a.m_fast ? q_m : lbid - q_t * a.tblocks;
    const int per_img = a.gx * a.gy;
    const int s_begin = blockIdx.z * a.steps_per_split;
    const int s_end = min(a.nsteps, s_begin + a.steps_per_split);
    const int nloc = s_end - s_begin;
    const unsigned plane = (unsigned)(H * W) * 4u;

    // ---- staging role: W4_ITEMS 16-byte loads per thread and step.  item -> (group g, channel ch, patch row, quad); the LDS
    // image of a patch has column c = image column X0 - 1 + c, so that a tile's six columns start 16-byte aligned (4 tx)
    const wrsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), (short)0, (int)a.xbytes, 0x00020000);
    unsigned svoff[W4_ITEMS];
    int slds[W4_ITEMS];               // LDS float index of the quad's LAST element (.w); < 0: nothing to write
    unsigned smask = 0;               // 3 bits per item: write .x | .y.z | .w
    unsigned schan = 0;               // 2 bits per item: channel of the step
    const int per_patch = a.PH * a.NQ;
#pragma unroll
    for (int it = 0; it < W4_ITEMS; ++it) {
        const int idx = tid + 256 * it;
        const int pc = fdiv(idx, a.mg_PHNQ), rem = idx - pc * per_patch;          // pc = g * 4 + ch
        const int row = fdiv(rem, a.mg_NQ), q = rem - row * a.NQ;
        const int g = pc >> 2, ch = pc & 3;
        const int grp = tblk * 4 + g;
        const bool act = pc < 16 && grp < a.ngroups;
        const int gq = act ? grp : 0;
        const int gb = fdiv(gq, a.mg_per_img), gr = gq - gb * per_img;
        const int gyi = fdiv(gr, a.mg_gx), gxi = gr - gyi * a.gx;
        const int iy = gyi * a.GH * 4 - 1 + row, ix = gxi * GW * 4 - 4 + 4 * q;
        const bool ok = act && iy >= 0 && iy < H && ix >= 0 && ix < W;
        svoff[it] = ok ? ((unsigned)gb * (unsigned)a.K + (unsigned)ch) * plane + (unsigned)(iy * W + ix) * 4u : 0x80000000u;
        slds[it] = pc < 16 ? pc * PLANE + row * PS + 4 * q : -1;
        // quad q covers slab columns 4q-3 .. 4q: the first quad only has its .w inside the patch, the last only its .x
        const unsigned m3 = pc < 16 ? ((q > 0 ? 1u : 0u) | (q > 0 && q < a.NQ - 1 ? 2u : 0u) | (q < a.NQ - 1 ? 4u : 0u)) : 0u;
        smask |= m3 << (3 * it);
        schan |= (unsigned)ch << (2 * it);
    }
    const wrsrc_t ur = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.uhat) + (size_t)mblk * a.nsteps * W4_USTEP, (short)0, (int)((size_t)a.nsteps * W4_USTEP * 4), 0x00020000);

    f4 px[W4_ITEMS];                  // one step's patches and weights in flight in registers (committed after the next compute)
    f4 pu[3];
    auto load_x = [&](int s, f4* dst) {
        // channel of step s: 4 s + ch (ch is inside svoff); a channel past K (K % 4 != 0) reads 0 through an invalid offset
        const unsigned soff = (unsigned)s * 4u * plane;
        const int left = a.K - 4 * s;              // channels of this step that exist
#pragma unroll
        for (int it = 0; it < W4_ITEMS; ++it) {
            const unsigned vo = (int)((schan >> (2 * it)) & 3u) < left ? svoff[it] : 0x80000000u;
            dst[it] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(xr, (int)vo, (int)soff, 0));
        }
    };
    auto load_u = [&](int s, f4* dst) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int i4 = tid + 256 * j;
            dst[j] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(ur, i4 < 576 ? i4 * 16 : (int)0x80000000u, s * (W4_USTEP * 4), 0));
        }
    };
    auto commit = [&](int buf, const f4* xs, const f4* us) {
        float* xw = xl[buf];
#pragma unroll
        for (int it = 0; it < W4_ITEMS; ++it) {
            const unsigned m3 = (smask >> (3 * it)) & 7u;
            const int o = slds[it];
            if (m3 & 1u) xw[o - 3] = xs[it].x;
            if (m3 & 2u) *reinterpret_cast<f2w*>(xw + o - 2) = f2w{xs[it].y, xs[it].z};
            if (m3 & 4u) xw[o] = xs[it].w;
        }
        float* uw = ul[buf];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int i4 = tid + 256 * j;
            if (i4 < 576) *reinterpret_cast<f4*>(uw + i4 * 4) = us[j];
        }
    };

    // ---- compute role: wave = group, lane = (tile n of the group, reduction channel kk of the step)
    const int tl = n < a.GH * GW ? n : 0;
    const int ty = fdiv(tl, a.mg_GW), tx = tl - ty * GW;
    const int xoff = (wave * 4 + kk) * PLANE + (4 * ty) * PS + 4 * tx;       // the tile's patch inside the group's plane of channel kk
    const int uoff = (kk * 9 * 16 + n) * 4;

    f4 acc[36];
#pragma unroll
    for (int p = 0; p < 36; ++p) acc[p] = f4{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int buf) {
        const float* xs = xl[buf] + xoff;
        const float* us = ul[buf] + uoff;
        // row pass as the raw rows arrive: s[i][b] = sum_j BT[b][j] d[i][j]
        //   BT = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
        float sm[6][6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const f4 lo = *reinterpret_cast<const f4*>(xs + i * PS);
            const f2w hi = *reinterpret_cast<const f2w*>(xs + i * PS + 4);
            const float d0 = lo.x, d1 = lo.y, d2 = lo.z, d3 = lo.w, d4 = hi.x, d5 = hi.y;
            const float e1 = fmaf(-4.f, d2, d4), o1 = fmaf(-4.f, d1, d3);      // rows 1 / 2: e1 +- o1
            const float e2 = d4 - d2, o2 = 2.f * (d3 - d1);                    // rows 3 / 4: e2 +- o2
            sm[i][0] = fmaf(4.f, d0, fmaf(-5.f, d2, d4));
            sm[i][1] = e1 + o1;
            sm[i][2] = e1 - o1;
            sm[i][3] = e2 + o2;
            sm[i][4] = e2 - o2;
            sm[i][5] = fmaf(4.f, d1, fmaf(-5.f, d3, d5));
        }
        // column pass per pair of Winograd rows + their 12 MFMAs; the A operands of a row pair are three 16-byte LDS reads
        auto rows = [&](int a0, int a1, auto f0, auto f1) {
            f4 ua[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) ua[j] = *reinterpret_cast<const f4*>(us + ((a0 * 6) / 4 + j) * 64);
#pragma unroll
            for (int b = 0; b < 6; ++b) {
                const float v0 = f0(b), v1 = f1(b);
                const int p0 = a0 * 6 + b, p1 = a1 * 6 + b;
                acc[p0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ua[(p0 - a0 * 6) / 4][(p0 - a0 * 6) % 4], v0, acc[p0], 0, 0, 0);
                acc[p1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ua[(p1 - a0 * 6) / 4][(p1 - a0 * 6) % 4], v1, acc[p1], 0, 0, 0);
            }
        };
        // rows 0 and 1 (positions 0..11), rows 2 and 3 (12..23), rows 4 and 5 (24..35): each pair starts on a multiple of 4
        rows(0, 1, [&](int b) { return fmaf(4.f, sm[0][b], fmaf(-5.f, sm[2][b], sm[4][b])); },
             [&](int b) { return fmaf(-4.f, sm[2][b], sm[4][b]) + fmaf(-4.f, sm[1][b], sm[3][b]); });
        rows(2, 3, [&](int b) { return fmaf(-4.f, sm[2][b], sm[4][b]) - fmaf(-4.f, sm[1][b], sm[3][b]); },
             [&](int b) { return (sm[4][b] - sm[2][b]) + 2.f * (sm[3][b] - sm[1][b]); });
        rows(4, 5, [&](int b) { return (sm[4][b] - sm[2][b]) - 2.f * (sm[3][b] - sm[1][b]); },
             [&](int b) { return fmaf(4.f, sm[1][b], fmaf(-5.f, sm[3][b], sm[5][b])); });
    };

    // ---- pipeline: LDS double-buffered (one barrier per step); while step i is multiplied the loads of step i+1 are in flight
    // in registers: they were issued after compute(i-1) and are committed to the other buffer after compute(i)
    if (nloc > 0) {
        load_x(s_begin, px); load_u(s_begin, pu);
        commit(0, px, pu);
        if (nloc > 1) { load_x(s_begin + 1, px); load_u(s_begin + 1, pu); }
    }
    __syncthreads();
#pragma unroll 1
    for (int i = 0; i < nloc; ++i) {
        compute(i & 1);
        if (i + 1 < nloc) {
            commit((i + 1) & 1, px, pu);
            if (i + 2 < nloc) { load_x(s_begin + i + 2, px); load_u(s_begin + i + 2, pu); }
        }
        __syncthreads();
    }

    // ---- output transform in registers: Y = A^T M A,  A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
    const int grp = tblk * 4 + wave;
    const bool o_act = grp < a.ngroups && n < a.GH * GW;
    const int gq = o_act ? grp : 0;
    const int ob = fdiv(gq, a.mg_per_img), orr = gq - ob * per_img;
    const int ogy = fdiv(orr, a.mg_gx), ogx = orr - ogy * a.gx;
    const int oy = (ogy * a.GH + ty) * 4, ox = (ogx * GW + tx) * 4;
    float* yout = a.y + (size_t)blockIdx.z * a.slab_stride;
    const bool has_add = a.addend && gridDim.z == 1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = mblk * 16 + kk * 4 + r;
        const bool ok = o_act && m < a.M && ox < W;
        const size_t o = (((size_t)ob * a.M + min(m, a.M - 1)) * H + min(oy, H - 1)) * W + min(ox, W - 4);
        f4 ad[4];
        if (has_add) {
#pragma unroll
            for (int i = 0; i < 4; ++i) ad[i] = *reinterpret_cast<const f4*>(a.addend + o + (size_t)(oy + i < H ? i : 0) * W);
        }
        // z[i][b] = sum_a AT[i][a] M[a][b]
        float z[4][6];
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            const float m0 = acc[0 * 6 + b][r], m1 = acc[1 * 6 + b][r], m2 = acc[2 * 6 + b][r], m3 = acc[3 * 6 + b][r],
                        m4 = acc[4 * 6 + b][r], m5 = acc[5 * 6 + b][r];
            const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
            z[0][b] = m0 + s12 + s34;
            z[1][b] = fmaf(2.f, d34, d12);
            z[2][b] = fmaf(4.f, s34, s12);
            z[3][b] = fmaf(8.f, d34, d12) + m5;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float s12 = z[i][1] + z[i][2], d12 = z[i][1] - z[i][2], s34 = z[i][3] + z[i][4], d34 = z[i][3] - z[i][4];
            f4 v;
            v.x = z[i][0] + s12 + s34;
            v.y = fmaf(2.f, d34, d12);
            v.z = fmaf(4.f, s34, s12);
            v.w = fmaf(8.f, d34, d12) + z[i][5];
            if (has_add) { v.x += ad[i].x; v.y += ad[i].y; v.z += ad[i].z; v.w += ad[i].w; }
            if (ok && oy + i < H) *reinterpret_cast<f4*>(yout + o + (size_t)i * W) = v;
        }
    }
}

// ---- group shape: GH x GW <= 16 tiles that wastes the fewest lanes on a TH x TW tile grid
static void wino4_pick_group(int TH, int TW, int& GH, int& GW) {
    static const int cand[][2] = {{3, 5}, {2, 8}, {4, 4}, {1, 16}, {2, 7}, {2, 6}, {3, 4}, {2, 5}, {1, 8}, {4, 3}, {5, 3}, {1, 4}, {2, 2}};
    double best = -1.0;
    for (auto& c : cand) {
        if ((4 * c[0] + 2) * (c[1] + 2) * 16 > W4_ITEMS * 256 || 16 * (4 * c[0] + 2) * (4 * c[1] + 4) > W4_XMAX) continue;
        const double covered = (double)ceil_div(TH, c[0]) * ceil_div(TW, c[1]) * 16.0;
        const double util = (double)TH * TW / covered;
        if (util > best + 1e-9) { best = util; GH = c[0]; GW = c[1]; }
    }
}

double wino4_utilisation(int H, int W) {
    const int TH = ceil_div(H, 4), TW = W / 4;
    int GH = 1, GW = 1;
    wino4_pick_group(TH, TW, GH, GW);
    return (double)H * W / ((double)ceil_div(TH, GH) * ceil_div(TW, GW) * 256.0);
}

bool wino4_eligible(int B, int K, int M, int H, int W) {
    if (W < 4 || (W & 3) || H < 4) return false;
    if ((size_t)B * std::max(K, M) * H * W * 4 >= 0x7fffffffull) return false;
    return true;
}

size_t wino4_uhat_bytes(int Ci, int Co) {
    const size_t a = (size_t)ceil_div(Ci, 16) * 16, b = (size_t)ceil_div(Co, 16) * 16;
    return (a * b * 36 * sizeof(float) + 255) & ~(size_t)255;
}

void wino4_dims(int Ci, int Co, bool dgrad, int& Mp, int& Kp) {
    const int M = dgrad ? Ci : Co, K = dgrad ? Co : Ci;
    Mp = ceil_div(M, 16) * 16;
    Kp = ceil_div(K, 4) * 4;
}

// y (B,M,H,W) = conv3x3(x (B,K,H,W), zero pad 1) [+ addend]; `uhat` = the transformed weights if the cache has them, else they
// are produced into `ws`; `slabs` = room for 2 partial outputs when the reduction is split
int wino4_launch(const float* x, const float* weight, const float* cached_uhat, float* y, const float* addend, void* ws, float* slabs,
                 int B, int Ci, int Co, int H, int W, bool dgrad, hipStream_t st) {
    const int K = dgrad ? Co : Ci, M = dgrad ? Ci : Co;
    Wino4Args a{};
    int Mp, Kp;
    wino4_dims(Ci, Co, dgrad, Mp, Kp);
    a.x = x; a.y = y; a.addend = addend;
    a.B = B; a.K = K; a.M = M; a.H = H; a.W = W;
    const int TH = ceil_div(H, 4), TW = W / 4;
    wino4_pick_group(TH, TW, a.GH, a.GW);
    a.gx = ceil_div(TW, a.GW); a.gy = ceil_div(TH, a.GH); a.ngroups = a.gx * a.gy * B;
    a.PH = 4 * a.GH + 2; a.NQ = a.GW + 2; a.PS = 4 * a.GW + 4; a.PLANE = a.PH * a.PS;
    if (16 * a.PLANE > W4_XMAX || 16 * a.PH * a.NQ > W4_ITEMS * 256) return DC_EINVAL;
    a.nsteps = Kp / 4;
    a.xbytes = (unsigned)((size_t)B * K * H * W * 4);
    a.tblocks = ceil_div(a.ngroups, 4); a.mblocks = Mp / 16;
    // reduction split: launches that would leave most of the 512 block slots empty split the channels in two
    const size_t nout = (size_t)B * M * H * W;
    int ksplit = 1;
    if ((long)a.tblocks * a.mblocks <= 256 && a.nsteps >= 8 && (H * W) % 4 == 0) ksplit = 2;
    if (const char* f = getenv("DC_WINO4_KSPLIT")) { const int v = atoi(f); if (v >= 1 && v <= 2 && a.nsteps >= 2 * v) ksplit = v; }
    a.steps_per_split = ceil_div(a.nsteps, ksplit);
    a.y = ksplit > 1 ? slabs : y;
    a.slab_stride = ksplit > 1 ? nout : 0;
    a.mg_mblocks = fdiv_magic(a.mblocks); a.mg_tblocks = fdiv_magic(a.tblocks); a.mg_per_img = fdiv_magic(a.gx * a.gy);
    a.mg_gx = fdiv_magic(a.gx); a.mg_GW = fdiv_magic(a.GW); a.mg_NQ = fdiv_magic(a.NQ); a.mg_PHNQ = fdiv_magic(a.PH * a.NQ);
    if ((unsigned long long)a.tblocks * a.mblocks * (unsigned)std::max(a.mblocks, a.tblocks) >= 0xffffffffull ||
        (unsigned long long)(a.ngroups + 8) * (unsigned)(a.gx * a.gy) >= 0xffffffffull) return DC_EINVAL;
    a.m_fast = (size_t)B * H * W >= (size_t)M * 16 ? 1 : 0;
    if (cached_uhat) {
        a.uhat = cached_uhat;
    } else {
        hipLaunchKernelGGL(wino4_weights_kernel, dim3((Mp / 16) * ((Kp + 15) / 16)), dim3(256), 0, st, weight, (float*)ws, Co, Ci, Mp, Kp,
                           dgrad ? 1 : 0);
        DC_CHECK_LAUNCH();
        a.uhat = (const float*)ws;
    }
    // algorithmic = 2 MAC of the direct convolution; executed = the 36 Winograd-domain GEMMs incl. tile padding
    hipEvent_t pe = conv_prof_begin(0, 2.0 * B * (double)M * K * 9.0 * H * W, 2.0 * 36.0 * (double)a.tblocks * 64.0 * (double)Mp * Kp,
                                    4.0 * ((double)B * K * H * W + (double)nout) + 36.0 * Co * Ci, st);
    const size_t lds = ((size_t)32 * a.PLANE + 2 * W4_USTEP) * sizeof(float);
    static const bool lds_ok = hipFuncSetAttribute((const void*)wino4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                   (int)(((size_t)2 * W4_XMAX + 2 * W4_USTEP) * sizeof(float))) == hipSuccess;
    if (!lds_ok) return DC_ELAUNCH;
    hipLaunchKernelGGL(wino4_kernel, dim3(a.tblocks * a.mblocks, 1, ksplit), dim3(256), lds, st, a);
    conv_prof_end(pe, st);
    DC_CHECK_LAUNCH();
    if (ksplit > 1) {
        const size_t n4 = nout / 4;
        const int rc = wino_ysum_launch(slabs, y, n4, ksplit, addend, st);
        if (rc != DC_OK) return rc;
    }
    return DC_OK;
}

}  // namespace dc
