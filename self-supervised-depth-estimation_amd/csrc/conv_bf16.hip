// 3x3 convolutions on the bf16 matrix cores (v_mfma_f32_16x16x32_bf16, fp32 accumulate) for gfx950 -- the
// "reduced-precision networks, fp32 loss" policy of BASELINE configs[4] (trainer_fusion_v3.py:311-330 run under mixed
// precision; networks/resnet_encoder.py:87-98, networks/depth_decoder.py:50-67, layers.py:106-136).
//
// Policy: tensors stay fp32 in HBM (activations, master weights, gradients, BatchNorm statistics); a convolution rounds its
// two matrix operands to bf16 on the way into LDS and accumulates in fp32.  bf16 keeps fp32's exponent, so the backward
// needs no loss scaling (an fp16 policy would).
//
// At 16x the fp32 matrix rate the multiplies are no longer what a 3x3 convolution costs, so these are DIRECT implicit GEMMs
// (no Winograd transform, no row-combine epilogue): what bounds them is reading fp32 operands from HBM.
//
//   forward / data gradient (c3b_conv_kernel): M = output channels, N = pixels, K = 9 taps x input channels.
//     A block owns an 8 x 32 (stride 2: 4 x 32) pixel tile x 16*MR output channels -- 32 pixels wide so that every row
//     segment it loads or stores is a whole 128-byte line (16-pixel tiles moved half lines: each line crossed HBM twice and
//     the kernels sat at 2.2 TB/s); per chunk of 32 input channels the
//     input patch is staged ONCE as a channel-contiguous bf16 image [pixel][32 ch (+8 pad)] -- the fp32 -> bf16 rounding,
//     the nearest-x2 upsample, the channel concat and the reflection / zero padding are all index arithmetic of the staging
//     (16-byte global loads, channel pairs packed by v_cvt_pk_bf16_f32).  One MFMA consumes one tap of all 32 channels: the B
//     operand of lane (pixel n, k-group kk) is ONE ds_read_b128 at pixel (y + ky, S*n + kx), channels 8kk..8kk+7; stride 2
//     is the same read at a different pixel.  Weights are pre-packed once per call as [m-block][chunk][tap][k-group][m][8]
//     bf16 (A operand = one ds_read_b128).  The next chunk's global loads are in flight during the MFMA phase.
//     The decoder's data gradient runs over the padded output domain and is folded back by conv_fold_kernel (conv3x3.hip);
//     zero-padded convolutions (the ResNet trunks) get it directly as a convolution with the rotated, transposed filter.
//   weight gradient (c3b_wgrad_kernel): M = 64 output channels (one 16-row tile per wave), N = 32 input channels,
//     K = pixels (32 per MFMA = two tile rows).  g' is staged pixel-contiguous (A operand: 8 consecutive pixels), the input
//     patch in the same channel-contiguous image as above and read TRANSPOSED by ds_read_b64_tr_b16 (8 consecutive pixels of
//     16 channels -> B operand); 9 accumulators per (wave, channel tile) = the 9 taps.  Split over pixel tiles into fp32
//     slabs, summed in fixed order by conv_wreduce_kernel (deterministic, no atomics).
#include "dc_common.h"
#include "conv_bf16.h"
#include "wino.h"

#include <algorithm>

namespace dc {

typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
typedef __attribute__((ext_vector_type(4))) short s4;
typedef __attribute__((ext_vector_type(8))) short s8;
typedef float f4 __attribute__((ext_vector_type(4)));

thread_local int g_matrix_prec = DC_PREC_F32;
int matrix_precision() { return g_matrix_prec; }

constexpr int BC = C3B_BC;    // reduction channels per chunk = the K of one v_mfma_f32_16x16x32_bf16
constexpr int BPX = 40;       // bf16 per pixel of the LDS image: 32 channels + 8 (80-byte pixels spread the b128 reads over the banks)

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    const bf2 p = {(__bf16)lo, (__bf16)hi};          // v_cvt_pk_bf16_f32 (round to nearest even, NaN stays NaN)
    return __builtin_bit_cast(unsigned, p);
}

__device__ __forceinline__ int pad_index_b(int i, int n, int pad, bool& ok) {
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
// weights (Co,Cin,3,3) fp32 -> bf16 [m-block][chunk][tap][k-group 4][m MT][8]
//   forward:        M = Co,  K = Cin, value = w[m][k][tap]
//   data gradient:  M = Cin, K = Co,  value = w[k][m][8 - tap]      (rotated, transposed filter)
// one thread per 16-byte item
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void c3b_wprep_kernel(const float* __restrict__ w, uint4* __restrict__ wb, int Co, int Cin, int dgrad,
                                                        int MT, int nmblk, int nchunks) {
    c3b_wprep_item(w, wb, blockIdx.x * 256 + threadIdx.x, Co, Cin, dgrad, MT, nmblk, nchunks);
}

// ------------------------------------------------------------------------------------------------
// staging of one 32-channel chunk of the input patch:  global fp32 (NCHW) -> registers -> LDS bf16 [pixel][BPX]
//   S     convolution stride (1 | 2); the tile is TH x 32 outputs, the patch ((TH-1)*S + 3) x (31*S + 3) input pixels
//   DPAD  the patch origin is the output tile origin - 2 and out-of-range pixels are zero (data gradient over the padded
//         domain); otherwise origin - 1 with reflection / zero padding, fused upsample and concat
// A thread item = (channel pair, patch row, aligned 4-column group) -> two 16-byte loads -> four packed dwords.
// ------------------------------------------------------------------------------------------------
struct PatchSrc {
    const float* x0; int C0; int up0;
    const float* x1; int C1;
    int H, W, pad;          // full-resolution input maps
    int dil;                // up0 = 1 only: x0 is DILATED instead of repeated -- full[2y][2x] = x0[y][x], zero elsewhere (the
                            // data gradient of a stride-2 convolution is a stride-1 convolution over the dilated g')
    unsigned bytes0, bytes1;    // sizes of x0 / x1 (< 2 GiB: the loads are buffer loads with 32-bit offsets)
};

// Every global load of the staging is a BUFFER load whose offset is either the element's or one past the end of the
// buffer (the hardware returns 0 for it): padding, tile overhang and ragged item counts need no branch.  A branch around a
// load makes the compiler wait for that load at the join -- the first version of this file did, and ran its 30 loads per
// chunk one after the other (13 k cycles per chunk instead of 2.3 k).
using brsrc_t = __amdgpu_buffer_rsrc_t;
constexpr unsigned OOB = 0x80000000u;
__device__ __forceinline__ brsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float4 ld128(brsrc_t r, unsigned off) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0));
}
__device__ __forceinline__ float ld32(brsrc_t r, unsigned off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0));
}

template <int S, int TH, bool DPAD, int NTHR = 256>
struct PatchStager {
    static constexpr int PH = (TH - 1) * S + 3, PW = 31 * S + 3;
    static constexpr int OFF = DPAD ? 2 : 1;
    static constexpr int NQ = 8 * S;                              // aligned quads per patch row
    static constexpr int NE = (S == 1) ? 2 : 1;                   // edge columns per patch row
    static constexpr int NIT = (16 * PH * NQ + NTHR - 1) / NTHR;
    static constexpr int NEI = (16 * PH * NE + NTHR - 1) / NTHR;
    static constexpr int LDS_ELEMS = PH * PW * BPX;               // bf16 elements of the image

    float4 rv[NIT][2];
    float re[NEI][2];

    __device__ __forceinline__ static int edge_col(int e) { return S == 1 ? (DPAD ? e : e * (PW - 1)) : 0; }

    __device__ __forceinline__ void prefetch(const PatchSrc& s, int k0, int b, int iy0, int ix0, int tid) {
        // (iy0, ix0) = input coordinates of patch pixel (0, OFF): ix0 is a multiple of 32
        const int K = s.C0 + s.C1;
        const bool upmode = !DPAD && s.up0 && k0 < s.C0;          // this chunk lives in the half-resolution x0
        const bool from1 = k0 >= s.C0;                            // ... or in x1 (chunks never straddle the concat)
        const int quads = upmode ? NQ / 2 : NQ;
        const int h0 = s.H >> s.up0, w0 = s.W >> s.up0;
        const brsrc_t rs = from1 ? make_rsrc(s.x1, s.bytes1) : make_rsrc(s.x0, s.bytes0);
        const int Csrc = from1 ? s.C1 : s.C0, cbase = from1 ? s.C0 : 0;
        const int sw = from1 ? s.W : w0;                          // row length / plane size of the source
        const unsigned plane = (unsigned)((from1 ? s.H : h0) * sw);
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            // consecutive lanes = consecutive channel pairs of one quad: their LDS dwords are consecutive (the commit is free of
            // bank conflicts; with the quad index fastest 64 lanes hit 8 banks) and 4 lanes still read 64 contiguous bytes
            const int it = tid + i * NTHR;
            const int cp = it & 15, q = (it >> 4) % quads, r = (it >> 4) / quads;
            const int ch = k0 + 2 * cp;
            bool ok = r < PH && ch < K, oky;
            int yy = iy0 + r;
            if (DPAD) oky = yy >= 0 && yy < s.H;
            else yy = pad_index_b(yy, s.H, s.pad, oky);
            const int xx = ix0 + 4 * q * (upmode ? 2 : 1);
            ok = ok && oky && xx < s.W && !(upmode && s.dil && (yy & 1));          // (odd rows of a dilated map are zero)
            const unsigned off = (((unsigned)(b * Csrc + (ch - cbase)) * plane) +
                                  (unsigned)((upmode ? (yy >> 1) : yy) * sw + (upmode ? (xx >> 1) : xx))) * 4u;
            rv[i][0] = ld128(rs, ok ? off : OOB);
            rv[i][1] = ld128(rs, (ok && ch + 1 < K) ? off + plane * 4u : OOB);
        }
#pragma unroll
        for (int i = 0; i < NEI; ++i) {
            const int it = tid + i * NTHR;
            const int cp = it & 15, e = (it >> 4) % NE, r = (it >> 4) / NE;
            const int ch = k0 + 2 * cp;
            bool ok = r < PH && ch < K, oky, okx;
            int yy = iy0 + r;
            int xx = ix0 - OFF + edge_col(e);
            if (DPAD) { oky = yy >= 0 && yy < s.H; okx = xx >= 0 && xx < s.W; }
            else { yy = pad_index_b(yy, s.H, s.pad, oky); xx = pad_index_b(xx, s.W, s.pad, okx); }
            ok = ok && oky && okx && !(upmode && s.dil && ((yy | xx) & 1));
            const unsigned off = (((unsigned)(b * Csrc + (ch - cbase)) * plane) +
                                  (unsigned)((upmode ? (yy >> 1) : yy) * sw + (upmode ? (xx >> 1) : xx))) * 4u;
            re[i][0] = ld32(rs, ok ? off : OOB);
            re[i][1] = ld32(rs, (ok && ch + 1 < K) ? off + plane * 4u : OOB);
        }
    }

    __device__ __forceinline__ void commit(const PatchSrc& s, int k0, unsigned* img, int ix0, int tid) const {
        const bool upmode = !DPAD && s.up0 && k0 < s.C0;
        const int quads = upmode ? NQ / 2 : NQ;
        const int qw = upmode ? 8 : 4;
        // A tile may hang over the right border (W % 32 != 0).  Its outside quads were loaded as zeros; the one value ever
        // read out there is ReflectionPad's column W (= column W-2): the thread holding the last inside quad stores it.
        const bool refl_over = !DPAD && s.pad == PAD_REFLECT && !s.dil && ix0 + 32 * S > s.W;
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int it = tid + i * NTHR;
            if (it < 16 * PH * quads) {
                const int cp = it & 15, q = (it >> 4) % quads, r = (it >> 4) / quads;
                const float4 a = rv[i][0], c = rv[i][1];
                const unsigned p0 = pack_bf16(a.x, c.x), p1 = pack_bf16(a.y, c.y), p2 = pack_bf16(a.z, c.z), p3 = pack_bf16(a.w, c.w);
                unsigned* dst = img + (r * PW + OFF + qw * q) * (BPX / 2) + cp;
                const int xx = ix0 + qw * q;
                const bool keep0 = !(refl_over && xx == s.W);
                if (upmode) {
                    const bool rep = !s.dil;                           // nearest x2: every value twice; dilation: value, zero
                    if (keep0) dst[0] = p0;
                    dst[BPX / 2] = rep ? p0 : 0u; dst[2 * (BPX / 2)] = p1; dst[3 * (BPX / 2)] = rep ? p1 : 0u;
                    dst[4 * (BPX / 2)] = p2; dst[5 * (BPX / 2)] = rep ? p2 : 0u; dst[6 * (BPX / 2)] = p3; dst[7 * (BPX / 2)] = rep ? p3 : 0u;
                    if (refl_over && xx + 8 == s.W) dst[8 * (BPX / 2)] = p3;
                } else {
                    if (keep0) dst[0] = p0;
                    dst[BPX / 2] = p1; dst[2 * (BPX / 2)] = p2; dst[3 * (BPX / 2)] = p3;
                    if (refl_over && xx + 4 == s.W) dst[4 * (BPX / 2)] = p2;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NEI; ++i) {
            const int it = tid + i * NTHR;
            if (it < 16 * PH * NE) {
                const int cp = it & 15, e = (it >> 4) % NE, r = (it >> 4) / NE;
                img[(r * PW + edge_col(e)) * (BPX / 2) + cp] = pack_bf16(re[i][0], re[i][1]);
            }
        }
    }
};

// ------------------------------------------------------------------------------------------------
// forward / data gradient
// ------------------------------------------------------------------------------------------------
struct C3bArgs {
    PatchSrc src;
    const uint4* wb;        // prepared weights of this pass
    const float* bias;
    const float* addend;    // !DPAD: (B, M, OH, OW) or NULL -- another consumer's gradient of the same tensor, added on the way out
    float* out;             // (B, M, OH, OW)
    int B, M, K;            // M output channels, K = C0 + C1 reduction channels
    int OH, OW, opitch;     // output maps; row pitch of `out` in floats (multiple of 4)
    int act;
    int tiles_x, tiles_y, mblocks, nchunks;
};

template <int MR, int S, bool DPAD>
__global__ __launch_bounds__(512, 2) void c3b_conv_kernel(C3bArgs a) {
    // 512 threads: eight waves share the staged images, each owns GPW of the tile's 16-pixel groups.  (With four waves the
    // staging registers of 256 threads plus 16 accumulator tiles left room for 2 waves per SIMD: load wait, LDS commit, MFMA
    // phase and stores of a block simply added up.  Half the staging items and half the accumulators per thread -> 4 waves
    // per SIMD from the same two blocks per CU.)
    constexpr int MT = 16 * MR, NTHR = 512;
    constexpr int TH = S == 1 ? 8 : 4, RWV = TH / 4;              // RWV = GPW: 16-pixel groups per wave (TH rows x two halves / 8 waves)
    using Stager = PatchStager<S, TH, DPAD, NTHR>;
    constexpr int PW = Stager::PW;
    constexpr int NWI = (36 * MT + NTHR - 1) / NTHR;              // 16-byte weight items per thread and chunk
    __shared__ __attribute__((aligned(16))) unsigned img[Stager::LDS_ELEMS / 2];
    __shared__ __attribute__((aligned(16))) uint4 wl[36 * MT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, kk = lane >> 4;
    // consecutive logical blocks (same tile, successive channel blocks) share an XCD: the patch is re-read from its L2
    const int lb = xcd_logical_block(blockIdx.x, gridDim.x);
    const int mblk = lb % a.mblocks;
    const int rest = lb / a.mblocks;
    const int ntiles = a.tiles_x * a.tiles_y;
    const int tile = rest % ntiles, b = rest / ntiles;
    const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    const int oy0 = ty * TH, ox0 = tx * 32;
    const int m0 = mblk * MT;
    const int iy0 = oy0 * S - Stager::OFF, ix0 = ox0 * S;

    f4 acc[MR][RWV];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < RWV; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

    Stager st;
    uint4 rw[NWI];
    const brsrc_t wr = make_rsrc(a.wb + (size_t)mblk * a.nchunks * 36 * MT, (unsigned)(a.nchunks * 36 * MT * 16));
    auto prefetch_w = [&](int c) {
#pragma unroll
        for (int i = 0; i < NWI; ++i) {
            const int it = tid + i * NTHR;
            rw[i] = __builtin_bit_cast(uint4, ld128(wr, it < 36 * MT ? (unsigned)((c * 36 * MT + it) * 16) : OOB));
        }
    };
    auto commit_w = [&]() {
#pragma unroll
        for (int i = 0; i < NWI; ++i) {
            const int it = tid + i * NTHR;
            if (it < 36 * MT) wl[it] = rw[i];
        }
    };

#ifndef C3B_ABL
#define C3B_ABL 0          // diagnostic builds only (tools/abl_bf16.sh): 1 no MFMA, 2 no LDS commit, 4 no stores, 8 no patch loads
#endif
    if (!(C3B_ABL & 8)) st.prefetch(a.src, 0, b, iy0, ix0, tid);
    prefetch_w(0);
    const __bf16* pim = reinterpret_cast<const __bf16*>(img);
    for (int c = 0; c < a.nchunks; ++c) {
        __syncthreads();                       // the previous chunk's MFMAs are done with the LDS images
        if (!(C3B_ABL & 2)) {
            st.commit(a.src, c * BC, img, ix0, tid);
            commit_w();
        } else {
            asm volatile("" :: "v"(st.rv[0][0].x), "v"(st.rv[Stager::NIT - 1][1].w), "v"(rw[0].x), "v"(rw[NWI - 1].w), "v"(st.re[0][0]));
        }
        __syncthreads();
        if (c + 1 < a.nchunks) {
            if (!(C3B_ABL & 8)) st.prefetch(a.src, (c + 1) * BC, b, iy0, ix0, tid);
            prefetch_w(c + 1);
        }
        if (C3B_ABL & 1) continue;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int ky = t / 3, kx = t - ky * 3;
            bf8 af[MR], bfr[RWV];
#pragma unroll
            for (int i = 0; i < MR; ++i) af[i] = __builtin_bit_cast(bf8, wl[(t * 4 + kk) * MT + i * 16 + n]);
#pragma unroll
            for (int j = 0; j < RWV; ++j) {        // group g of the tile: row g / 2, columns 16 (g & 1) .. +15
                const int g = wave * RWV + j;
                bfr[j] = *reinterpret_cast<const bf8*>(pim + ((S * (g >> 1) + ky) * PW + S * (16 * (g & 1) + n) + kx) * BPX + kk * 8);
            }
#ifndef C3B_REP
#define C3B_REP 1          // diagnostic builds only: the MFMA work of a tap issued C3B_REP times (cost model of split-operand products)
#endif
#pragma unroll
            for (int rep = 0; rep < C3B_REP; ++rep)
#pragma unroll
            for (int i = 0; i < MR; ++i)
#pragma unroll
                for (int j = 0; j < RWV; ++j)      // D[pixel][channel] = patch fragment (A) x weight fragment (B): see the epilogue
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        }
    }
    // ---- epilogue.  The pixel index is the ROW of the MFMA result (the patch fragment is the A operand), so a lane holds 4
    // consecutive pixels of one output channel: C/D layout col (channel) = lane & 15, row (pixel) = (lane >> 4) * 4 + reg.
    // One 16-byte store per lane and accumulator tile (dword stores, four planes per instruction, cost as much as all the rest
    // of the kernel on the wide thin decoder levels).  bias + activation first, stores afterwards: a store behind a pending
    // load makes the compiler wait for everything outstanding.
    float bv[MR];
#pragma unroll
    for (int i = 0; i < MR; ++i) {
        const int m = m0 + i * 16 + n;
        bv[i] = (!DPAD && a.bias && m < a.M) ? a.bias[m] : 0.f;
    }
    if (!DPAD) {
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int j = 0; j < RWV; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = act_fwd(acc[i][j][r] + bv[i], a.act);
    }
    if (!DPAD && a.addend) {          // (all the addend loads first, then the adds: the stores below stay behind no pending load)
        f4 av[MR][RWV];
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int j = 0; j < RWV; ++j) {
                const int g = wave * RWV + j;
                const int py = oy0 + (g >> 1), px = ox0 + 16 * (g & 1) + 4 * kk;
                const int m = m0 + i * 16 + n;
                const bool ok = m < a.M && py < a.OH && px < a.opitch;
                av[i][j] = ok ? *reinterpret_cast<const f4*>(a.addend + (((size_t)b * a.M + m) * a.OH + py) * a.opitch + px) : f4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int j = 0; j < RWV; ++j) acc[i][j] += av[i][j];
    }
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < RWV; ++j) {
            const int g = wave * RWV + j;
            const int py = oy0 + (g >> 1), px = ox0 + 16 * (g & 1) + 4 * kk;
            const int m = m0 + i * 16 + n;
            // (a.opitch: row pitch of `out`, a multiple of 4 -- OW, or OW rounded up for the padded domain)
            if (!(C3B_ABL & 4) && m < a.M && py < a.OH && px < a.opitch)
                *reinterpret_cast<f4*>(a.out + (((size_t)b * a.M + m) * a.OH + py) * a.opitch + px) = acc[i][j];
        }
}

// ------------------------------------------------------------------------------------------------
// weight gradient: dW[co][ci][t] = sum_{b,y,x} g'[b,co,y,x] * xpad[b,ci,S*y+ky-1,S*x+kx-1]
// grid (split, ceil(Co/64), ceil(Cin/32)), 512 threads; a block walks its share of the pixel tiles; wave (cw, nt) owns output
// channels 16cw..16cw+15 x input channels 16nt..16nt+15 of the block x 9 taps (9 accumulator tiles): no cross-wave sum.
// ------------------------------------------------------------------------------------------------
struct C3bWgArgs {
    PatchSrc src;
    const float* gp;        // g' = gy * act'(y): (B, Co, OH, OW)
    float* part;            // [split][Co][Cin*9]
    unsigned gbytes;
    int B, Co, OH, OW;
    int tiles_x, tiles_y, split;
};

template <int S>
__global__ __launch_bounds__(512, 2) void c3b_wgrad_kernel(C3bWgArgs a) {
    constexpr int TH = S == 1 ? 8 : 4;
    using Stager = PatchStager<S, TH, false, 512>;
    constexpr int PW = Stager::PW;
    constexpr int GLS = TH * 32 + 8;                               // g' row stride in bf16 (rows of 16 (2n+1) bytes: conflict-free b128 reads)
    constexpr int NG = (64 * TH * 8) / 512;                        // float4 items of the g' tile per thread
    __shared__ __attribute__((aligned(16))) unsigned img[Stager::LDS_ELEMS / 2];
    __shared__ __attribute__((aligned(16))) unsigned gl[64 * GLS / 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cw = wave & 3, nt = wave >> 2;                       // wave = (16-row tile of output channels, 16-column tile of input channels)
    const int n = lane & 15, kk = lane >> 4;
    const int m0 = blockIdx.y * 64, c0 = blockIdx.z * BC;
    const int Cin = a.src.C0 + a.src.C1;
    const int per_img = a.tiles_x * a.tiles_y;
    const int ntiles = per_img * a.B;

    f4 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = f4{0.f, 0.f, 0.f, 0.f};

    Stager st;
    float4 rg[NG];
    const brsrc_t gr = make_rsrc(a.gp, a.gbytes);
    int cur_ix0 = 0;
    auto prefetch = [&](int tile) {
        const int b = tile / per_img, tt = tile - b * per_img;
        const int ty = tt / a.tiles_x, tx = tt - ty * a.tiles_x;
        const int oy0 = ty * TH, ox0 = tx * 32;
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const int it = tid + i * 512;
            const int q = it & 7, r = (it >> 3) % TH, m = it / (8 * TH);
            const int co = m0 + m, yy = oy0 + r, xx = ox0 + 4 * q;
            const bool ok = co < a.Co && yy < a.OH && xx < a.OW;
            rg[i] = ld128(gr, ok ? (unsigned)((((b * a.Co + co) * a.OH + yy) * a.OW + xx) * 4) : OOB);
        }
        st.prefetch(a.src, c0, b, oy0 * S - 1, ox0 * S, tid);
        return ox0 * S;
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const int it = tid + i * 512;
            const int q = it & 7, r = (it >> 3) % TH, m = it / (8 * TH);
            const float4 v = rg[i];
            *reinterpret_cast<uint2*>(gl + (m * GLS + r * 32 + 4 * q) / 2) = make_uint2(pack_bf16(v.x, v.y), pack_bf16(v.z, v.w));
        }
        st.commit(a.src, c0, img, cur_ix0, tid);
    };

    const __bf16* gim = reinterpret_cast<const __bf16*>(gl);
    const short* xim = reinterpret_cast<const short*>(img);
    // transposed read of the B operand: lane 4q+p of a 16-lane group supplies the address of pixel row q, channels 4p..4p+3
    const int tq = n >> 2, tp = n & 3;
    int tile = blockIdx.x, nxt_ix0 = 0;
    if (tile < ntiles) nxt_ix0 = prefetch(tile);
    for (; tile < ntiles; tile += a.split) {
        __syncthreads();
        cur_ix0 = nxt_ix0;
        commit();
        __syncthreads();
        if (tile + a.split < ntiles) nxt_ix0 = prefetch(tile + a.split);
#pragma unroll 2
        for (int s = 0; s < TH; ++s) {                                         // one tile row (32 pixels) per MFMA
            const int trow = s, tcol = 8 * kk;                                 // first of this lane's 8 output pixels
            const bf8 af = *reinterpret_cast<const bf8*>(gim + (cw * 16 + n) * GLS + trow * 32 + tcol);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int ky = t / 3, kx = t - ky * 3;
                const int pix = (S * trow + ky) * PW + S * tcol + kx;          // patch pixel of output pixel 0 of the lane group
                const short* base = xim + (pix + S * tq) * BPX + nt * 16 + 4 * tp;
                const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(base));
                const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(base + 4 * S * BPX));
                const s8 bv = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
                for (int rep = 0; rep < C3B_REP; ++rep)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, __builtin_bit_cast(bf8, bv), acc[t], 0, 0, 0);
            }
        }
    }
    // ---- slab: part[split][co][ci*9 + t]; C/D layout col (ci) = lane & 15, row (co) = (lane >> 4) * 4 + reg
    float* slab = a.part + (size_t)blockIdx.x * a.Co * Cin * 9;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = m0 + cw * 16 + kk * 4 + r, ci = c0 + nt * 16 + n;
            if (co < a.Co && ci < Cin) slab[((size_t)co * Cin + ci) * 9 + t] = acc[t][r];
        }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static inline size_t al256b(size_t v) { return (v + 255) & ~(size_t)255; }
static inline int c3b_mr(int M) { return M > 32 ? 4 : (M > 16 ? 2 : 1); }

bool c3b_eligible(int C0, int C1, int up0, int H, int W, int stride) {
    // 16-byte staging: whole quads inside a row (W % 4; W % 8 when x0 is read at half resolution), chunks that do not
    // straddle the concat, channel pairs inside one source
    if (stride == 1) return W % 4 == 0 && W >= 4 && H >= 2 && (C1 == 0 || C0 % BC == 0) && (!up0 || (W % 8 == 0 && (H & 1) == 0));
    return stride == 2 && W % 8 == 0 && H % 2 == 0 && C1 == 0 && !up0;
}

size_t c3b_weights_bytes(int Ci, int Co) {
    // either pass: M padded to its 16*MR block, K to 32
    const int a = ceil_div(Co, 16 * c3b_mr(Co)) * 16 * c3b_mr(Co), b = ceil_div(Ci, 16 * c3b_mr(Ci)) * 16 * c3b_mr(Ci);
    const size_t fwd = (size_t)a * ceil_div(Ci, BC) * BC * 9 * 2, dg = (size_t)b * ceil_div(Co, BC) * BC * 9 * 2;
    return al256b(std::max(fwd, dg));
}

int c3b_dpad_pitch(int W) { return (W + 2 + 3) & ~3; }

int c3b_wgrad_split(int B, int OH, int OW, int Co, int Cin, int stride) {
    const int TH = stride == 1 ? 8 : 4;
    const int ntiles = ceil_div(OW, 32) * ceil_div(OH, TH) * B;
    const int outer = ceil_div(Co, 64) * ceil_div(Cin, BC);
    // one round of 512-thread blocks (two per CU): every extra split is another fp32 slab of the whole weight tensor to write
    // and to read back (at 512 splits the slabs of a 64 -> 64 layer were 150 MB of traffic, more than its activations)
    int split = std::max(1, std::min(ntiles, 512 / std::max(outer, 1)));
    return std::min(split, 256);
}

// y / dxpad / dx = conv(cat(up2?(x0), x1), weight) on the bf16 matrix cores.
//   dgrad = 0: forward (M = Co) with bias + activation;  dgrad = 1: rotated, transposed filter (M = Cin).
//   dpad = 1: input = g' (B,K,H,W), output over the padded domain (H+2, W+2) (decoder data gradient; fold afterwards).
int c3b_conv(const float* x0, int C0, int up0, const float* x1, int C1, const float* weight, int Co, int Cin, int dgrad, int dpad,
             const float* bias, float* out, void* ws, int B, int H, int W, int act, int pad, int stride, hipStream_t st,
             const float* addend) {
    const int M = dgrad ? Cin : Co, K = C0 + C1;
    if (addend && dpad) return DC_EINVAL;
    if (K != (dgrad ? Co : Cin)) return DC_EINVAL;
    const int mr = c3b_mr(M), MT = 16 * mr;
    const int mblocks = ceil_div(M, MT), nchunks = ceil_div(K, BC);
    // prepared weights: from the per-step cache when the weight is registered and fresh (one batched launch per step packs
    // every variant: 100 launches of ~8 us per C5 step otherwise), else packed here into the workspace
    const uint4* wb = (const uint4*)wc_lookup_c3b(weight, Cin, Co, dgrad, MT, mblocks, nchunks, st);
    if (!wb) {
        const int nitems = mblocks * nchunks * 36 * MT;
        hipLaunchKernelGGL(c3b_wprep_kernel, dim3(ceil_div(nitems, 256)), dim3(256), 0, st, weight, (uint4*)ws, Co, Cin, dgrad, MT, mblocks,
                           nchunks);
        DC_CHECK_LAUNCH();
        wb = (const uint4*)ws;
    }
    C3bArgs a{};
    const size_t e0 = (size_t)B * C0 * (H >> (up0 & 1)) * (W >> (up0 & 1)) * 4, e1 = (size_t)B * C1 * H * W * 4;
    if (e0 >= 0x7fffffffull || e1 >= 0x7fffffffull) return DC_EINVAL;            // 32-bit buffer offsets
    a.src = PatchSrc{x0, C0, up0 & 1, x1, C1, H, W, pad, (up0 >> 1) & 1, (unsigned)e0, (unsigned)e1};
    a.wb = wb; a.bias = bias; a.addend = addend; a.out = out; a.B = B; a.M = M; a.K = K; a.act = act;
    a.OH = dpad ? H + 2 : H / stride; a.OW = dpad ? W + 2 : W / stride;
    a.opitch = (a.OW + 3) & ~3;          // padded domain: W + 2 rounded up (c3b_dpad_pitch); otherwise OW itself (W % 4 == 0)
    const int TH = stride == 1 ? 8 : 4;
    a.tiles_x = ceil_div(a.OW, 32); a.tiles_y = ceil_div(a.OH, TH); a.mblocks = mblocks; a.nchunks = nchunks;
    const long nblk = (long)a.tiles_x * a.tiles_y * mblocks * B;
    if (nblk > 0x7fffffffL) return DC_EINVAL;
    const dim3 grid((unsigned)nblk);
    // algorithmic: 2 MAC per tap of the convolution as specified (the dilated data gradient counts the stride-2 convolution's
    // MACs, i.e. a quarter of the positions); executed: what the matrix cores are issued; bytes: each operand and the result once
    const double macs = (double)B * M * K * 9.0 * a.OH * a.OW * ((up0 & 2) ? 0.25 : 1.0);
    const double in_elems = (double)B * ((double)C0 * (H >> (up0 & 1)) * (W >> (up0 & 1)) + (double)C1 * H * W);
    hipEvent_t pe = conv_prof_begin(2, 2.0 * macs, 2.0 * (double)nblk * (stride == 1 ? 256 : 128) * MT * (double)(nchunks * BC) * 9.0,
                                    4.0 * (in_elems + (double)B * M * a.OH * a.OW) + 36.0 * Co * Cin, st);
#define C3B_LAUNCH(MRV, SV, DP) hipLaunchKernelGGL((c3b_conv_kernel<MRV, SV, DP>), grid, dim3(512), 0, st, a)
    if (stride == 2) {
        if (dpad) return DC_EINVAL;
        if (mr == 4) C3B_LAUNCH(4, 2, false); else if (mr == 2) C3B_LAUNCH(2, 2, false); else C3B_LAUNCH(1, 2, false);
    } else if (dpad) {
        if (mr == 4) C3B_LAUNCH(4, 1, true); else if (mr == 2) C3B_LAUNCH(2, 1, true); else C3B_LAUNCH(1, 1, true);
    } else {
        if (mr == 4) C3B_LAUNCH(4, 1, false); else if (mr == 2) C3B_LAUNCH(2, 1, false); else C3B_LAUNCH(1, 1, false);
    }
#undef C3B_LAUNCH
    conv_prof_end(pe, st);
    DC_CHECK_LAUNCH();
    return DC_OK;
}

// part[split][Co][Cin*9] from x = cat(up2?(x0), x1) (B,Cin,H,W) and g' (B,Co,H/stride,W/stride); the caller reduces the slabs
int c3b_wgrad(const float* x0, int C0, int up0, const float* x1, int C1, const float* gp, float* part, int split, int B, int Co, int H,
              int W, int pad, int stride, hipStream_t st) {
    C3bWgArgs a{};
    const size_t e0 = (size_t)B * C0 * (H >> up0) * (W >> up0) * 4, e1 = (size_t)B * C1 * H * W * 4;
    const size_t eg = (size_t)B * Co * (H / stride) * (W / stride) * 4;
    if (e0 >= 0x7fffffffull || e1 >= 0x7fffffffull || eg >= 0x7fffffffull) return DC_EINVAL;      // 32-bit buffer offsets
    a.src = PatchSrc{x0, C0, up0, x1, C1, H, W, pad, 0, (unsigned)e0, (unsigned)e1};
    a.gbytes = (unsigned)eg;
    a.gp = gp; a.part = part; a.B = B; a.Co = Co; a.OH = H / stride; a.OW = W / stride;
    const int TH = stride == 1 ? 8 : 4;
    a.tiles_x = ceil_div(a.OW, 32); a.tiles_y = ceil_div(a.OH, TH); a.split = split;
    const dim3 grid(split, ceil_div(Co, 64), ceil_div(C0 + C1, BC));
    const int Cin = C0 + C1;
    const double in_elems = (double)B * ((double)C0 * (H >> up0) * (W >> up0) + (double)C1 * H * W);
    hipEvent_t pe = conv_prof_begin(3, 2.0 * (double)B * Co * Cin * 9.0 * a.OH * a.OW,
                                    2.0 * (double)a.tiles_x * a.tiles_y * B * TH * 32.0 * (grid.y * 64.0) * (grid.z * 32.0) * 9.0,
                                    4.0 * (in_elems + (double)B * Co * a.OH * a.OW) + 36.0 * Co * Cin, st);
    if (stride == 1) hipLaunchKernelGGL((c3b_wgrad_kernel<1>), grid, dim3(512), 0, st, a);
    else hipLaunchKernelGGL((c3b_wgrad_kernel<2>), grid, dim3(512), 0, st, a);
    conv_prof_end(pe, st);
    DC_CHECK_LAUNCH();
    return DC_OK;
}

}  // namespace dc

extern "C" int dc_set_matrix_precision(int precision) {
    if (precision != DC_PREC_F32 && precision != DC_PREC_BF16) return DC_EINVAL;
    const int prev = dc::g_matrix_prec;
    dc::g_matrix_prec = precision;
    return prev;
}

extern "C" int dc_get_matrix_precision(void) { return dc::g_matrix_prec; }
