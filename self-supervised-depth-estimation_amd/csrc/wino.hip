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
// Mapping to CDNA4 (wino_ps_kernel)
//   * v_mfma_f32_16x16x4_f32: rows = 16 output channels, cols = 16 tiles, K = 4 reduction channels.
//   * a block = 4 waves = the 4 ROWS of the 4x4 Winograd domain; all waves share one LDS slab of the raw input.
//     Row a of V = B^T d B needs two raw patch rows per tile (2 ds_read2_b64, 4 fma + 4 add -> the wave's four B
//     operands); nothing transformed ever touches HBM.
//   * the transformed weights U are produced once per call by wino_weights_kernel as [m-block][k-chunk][k][row][m][col].
//     The four waves need DISJOINT quarters of a chunk (row = wave), so U bypasses LDS: a lane loads the 16 bytes that
//     are its A operands for the 4 positions of one k-step and 16-channel block straight into registers (256-byte rows
//     per 16 lanes, L2 hits for every tile block but the first of an XCD), one chunk ahead.
//   * a wave keeps 4*MR*NR accumulator tiles; three tile variants -- (MR,NR) = (1,2): 16 output channels x 32 tiles, 4 blocks
//     per CU; (2,2): 32 x 32, 3 per CU; (2,4): 32 x 64, 2 per CU -- and a K split, picked per launch by wino_ps_cost (a model
//     of the lock-step ROUNDS a launch runs in, fitted to per-block timelines; below).  The four rows are combined at the
//     end through LDS (Y = A^T M A is linear in the rows of M), 8-byte stores.
//   * tiles are grouped in sub-regions of <= 32 (RH x RW chosen per map so that it divides evenly: 4x8, 2x16,
//     3x10, 8x4); a block takes NR/2 consecutive sub-regions, which may lie in different images.
//   * pipeline: the slab is double-buffered in LDS (one barrier per chunk of 8 channels); while chunk c is
//     multiplied, U of c+1 and x of c+1 and c+2 are in flight in registers (x comes from HBM on first touch).
//   * block order puts the readers of the larger stream (x across channel blocks, or U across tile blocks) next to
//     each other ON ONE XCD (xcd_logical_block) so that the re-reads are L2 hits; small maps split the reduction
//     over gridDim.z and sum the partial outputs in fixed order (wino_ysum_kernel).
// Measured (-DWINO_DIAG builds, tools/diag_wino.sh): mode 1 = phase split inside the loop (an earlier revision, B=24 256->64
// 48x160: MFMA phase 61 %, LDS commit incl. load wait 15 %, load issue 14 %, barrier 4 %; the timers cost ~20 % themselves);
// mode 2 = per-block start / end (s_memrealtime) and prologue / loop / epilogue lengths -> DESIGN 4a "block timelines".
// dgrad is the same kernel on the 180-degree-rotated, transposed weights (wino_weights_kernel<DGRAD>).
// Requires even W (8-byte row alignment); H arbitrary.
#include "dc_common.h"
#include "conv_bf16.h"
#include "wino.h"
#include "wino4.h"
#include "gemm1x1_x3.h"

#include <algorithm>
#include <cstdio>
#include <mutex>
#include <vector>

namespace dc {

using f4 = __attribute__((ext_vector_type(4))) float;
using f2w = __attribute__((ext_vector_type(2))) float;
using wrsrc_t = __amdgpu_buffer_rsrc_t;

// U = G g G^T for every (m, k), written in staging order.  grid over padded (Mp x Kp); one thread per (m, k).
template <bool DGRAD>
__device__ __forceinline__ void wino_weight_one(const float* __restrict__ w, float* __restrict__ uhat, int idx, int Co, int Ci, int MT,
                                                int Mp, int Kp, int WK) {
    // 256 consecutive threads = a 16 (m) x 16 (k) tile with m fastest: the 16-byte stores of 16 consecutive m are one
    // 256-byte run (the staging order has m innermost), and the filter reads stay efficient -- 16 consecutive k of a row are
    // 576 contiguous bytes (forward), 16 consecutive m are (data gradient).  With k fastest the batched launch spent 310 us
    // on 440 MB: every store instruction scattered 16-byte pieces 256 bytes apart.
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
template <bool DGRAD>
__global__ __launch_bounds__(256) void wino_weights_kernel(const float* __restrict__ w, float* __restrict__ uhat,
                                                           int Co, int Ci, int MT, int Mp, int Kp, int WK) {
    wino_weight_one<DGRAD>(w, uhat, blockIdx.x * 256 + threadIdx.x, Co, Ci, MT, Mp, Kp, WK);
}

// Every registered weight of a model in ONE launch (dc_wino_cache_refresh): `table` holds one descriptor per (weight,
// dgrad, MT) variant with the first block of its range; a block finds its descriptor by binary search.
struct WinoWDesc {
    const float* w; float* uhat;
    int Co, Ci, MT, Mp, Kp, dgrad, block0, kind;      // kind 1: the bf16 direct kernels' prepared weights (Mp = m-blocks, Kp = chunks)
};
__global__ __launch_bounds__(256) void wino_weights_batched_kernel(const WinoWDesc* __restrict__ table, const int* __restrict__ blk2desc, int WK) {
    // (a per-block binary search over the table -- eight dependent global loads in front of every block -- made this launch
    // 310 us for 0.5 GB; the host uploads the block -> descriptor map next to the table instead)
    const WinoWDesc d = table[blk2desc[blockIdx.x]];
    const int idx = ((int)blockIdx.x - d.block0) * 256 + threadIdx.x;
    if (d.kind == 1) { c3b_wprep_item(d.w, reinterpret_cast<uint4*>(d.uhat), idx, d.Co, d.Ci, d.dgrad, d.MT, d.Mp, d.Kp); return; }
    if (d.kind == 2) {       // split-operand 1x1 GEMMs: (Mp, Kp) = (padded rows, reduction extent) of this direction
        g1x3_prep_item(d.w, reinterpret_cast<unsigned short*>(d.uhat), idx, d.dgrad, d.dgrad ? d.Ci : d.Co, d.Mp, d.Kp);
        return;
    }
    if (d.dgrad) wino_weight_one<true>(d.w, d.uhat, idx, d.Co, d.Ci, d.MT, d.Mp, d.Kp, WK);
    else wino_weight_one<false>(d.w, d.uhat, idx, d.Co, d.Ci, d.MT, d.Mp, d.Kp, WK);
}

// ------------------------------------------------------------------------------------------------
// The convolution kernel (design notes at the top of the file).  MR x NR = 16-channel x 16-tile accumulator tiles
// per wave and Winograd position; gridDim.z > 1 splits the reduction channels into slabs.
// ------------------------------------------------------------------------------------------------
constexpr int PSK = 8;                // reduction channels per staged chunk
constexpr int PSUB = 256;             // floats per (channel, sub-region) plane: >= 240 (4x8 shape); 256 makes the slab pair big enough for the row exchange

struct WinoPsArgs {
    const float* x; const float* uhat; float* y;
    int B, K, M, H, W;                // input maps are H x W; K = reduction channels, M = output channels
    int Ho, Wo;                       // output map (H x W, or (H+2) x (W+2) for the full correlation P = 2)
    // FUSED only: x = cat(up2?(x0), x1) along channels, padding mode, patch origin = output - P, bias + activation
    const float* x1; const float* bias;
    const float* addend;              // !FUSED only: same shape as y, added in the store epilogue (the other gradient of a residual fork)
    // FUSED, MODE 4 (the data gradient of a fused block written where it belongs -- no padded-domain scratch, no fold pass): the first
    // split_c0 output rows are the channels of x0 -- at half resolution when split_up (the 2 x 2 output tile of a lane IS one
    // upsampled pixel: summed in registers) -- and go to y; the rest are the channels of x1 and go to out1; add0 / add1 (nullable,
    // shapes of y / out1; may alias their output) are added on the way out
    float* out1; const float* add0; const float* add1;
    int split_c0, split_up;
    int C0, up0, pad, P, act;
    unsigned x1bytes;
    int RH, RW, RS, SUBS;             // sub-region shape in tiles, LDS row stride, plane floats (rows * RS)
    int regs_x, regs_y, nsub;         // sub-regions per image / total
    int nchunks, chunks_per_split;
    unsigned xbytes;
    size_t slab_stride;               // floats between the K-split slabs
    int mblocks, tblocks, m_fast;
    int items;                        // work items = tblocks * mblocks; a launch with FEWER blocks than items is persistent (below)
    unsigned mg_mblocks, mg_tblocks, mg_per_img, mg_regs_x, mg_PR, mg_RW;   // fdiv magics of the divisors the kernel divides by
    unsigned long long* diag;         // WINO_DIAG builds only: per block {compute, commit(+load wait), issue, barrier, loop, prologue, epilogue, start time} cycles
    // ---- BatchNorm folded into the launch (plain launches; include/depthcore.h: dc_bn_fold) ----
    const float* in_scale;            // MODE 1: the input is relu(scale[g, k] x + shift[g, k]) (zero padding stays zero), applied between
    const float* in_shift;            //   the global load and the LDS store; MODE 2: the same pair re-derives the ReLU decision
    int npg;                          // images per BatchNorm group
    float* stat_part;                 // MODE 0 / 1, nullable: (M, stat_nparts) x {sum, sum of squares} of the output, partial 2 sub + half
    int stat_nparts;
    const float* bn_x;                // MODE 2 / 3 (data gradient): raw input of the BatchNorm whose ReLU-ed output the forward read,
    const float* bn_mean;             //   its mean (groups, M), its ReLU bit mask (MODE 3) and the partials (M, bwd_nparts) x
    const unsigned long long* bn_mask;   // {sum g', sum g' (x - mean)} of the masked result g'
    float* bwd_part;
    int bwd_nparts;
};

// sum over the 16 lanes of a DPP row (lanes sharing lane >> 4), result in every lane of the row
__device__ __forceinline__ float wrow16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
    return v;
}

// MODE (plain, non-persistent launches only): 0 none (+ optional statistics epilogue), 1 BatchNorm + ReLU folded into the loader
// (+ optional statistics epilogue), 2 / 3 BatchNorm-backward epilogue of a data gradient (ReLU decision re-derived / from bits);
// 4 (FUSED launches): split store of a fused block's data gradient (WinoPsArgs::split_c0)
template <int MR, int NR, bool FUSED, bool PERSIST = false, int MODE = 0>
__global__ __launch_bounds__(256, (MR * NR >= 8 ? 2 : (MR * NR >= 4 ? 3 : 4))) void wino_ps_kernel(WinoPsArgs a) {
    static_assert(MODE == 0 || (MODE == 4 && FUSED && !PERSIST) || (MODE < 4 && !FUSED && !PERSIST),
                  "the BatchNorm fold exists for the plain trunk launches, the split store for the fused data gradient");
    constexpr int MT = 16 * MR, G = NR / 2;
    constexpr int UF4 = PSK * 4 * MT;                 // f4 items of one U chunk in global memory
    static_assert(NR % 2 == 0 && NR * 2048 <= 2 * PSK * G * PSUB, "the row exchange aliases the slab double buffer");
    __shared__ float xl[2][PSK * G * PSUB];
#ifdef WINO_DIAG
    const unsigned long long dg_start = __builtin_amdgcn_s_memtime();
    const unsigned long long dg_rstart = __builtin_amdgcn_s_memrealtime();
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kk = lane >> 4;
    const int H = a.H, W = a.W, RH = a.RH, RW = a.RW, RS = a.RS, SUBS = a.SUBS;
    const int CPS = G * SUBS;                         // channel plane stride in the slab
    // block -> (tile block, channel block): the faster index is the one whose operand is the larger stream, so that
    // its other readers find it in L2
    // Work items.  A launch with as many blocks as items gives every block one (the classic form).  A PERSISTENT launch
    // (gridDim.x < a.items; gridDim.z == 1, an even number of chunks) gives a block one item per round of gridDim.x items
    // and runs their chunks as ONE software pipeline: the first two x chunks and the first U chunk of the next item are
    // requested during the last chunks of the current one, so the next item's first HBM touch flies under this item's
    // row-exchange epilogue instead of in front of an idle matrix pipe (prologue + epilogue are 12-23 % of a
    // block and nothing overlaps them inside a lock-step round).
    const int lbid = xcd_logical_block(blockIdx.x, gridDim.x);       // neighbours in this order share an XCD (one L2)
    int item = lbid;
    auto item_blocks = [&](int it, int& mb, int& tb) {
        const int q_m = fdiv(it, a.mg_mblocks), q_t = fdiv(it, a.mg_tblocks);
        mb = a.m_fast ? it - q_m * a.mblocks : q_t;
        tb = a.m_fast ? q_m : it - q_t * a.tblocks;
    };
    int mblk, tblk;
    item_blocks(item, mblk, tblk);
    const int per_img = a.regs_x * a.regs_y;
    const int c_begin = blockIdx.z * a.chunks_per_split;
    const int c_end = min(a.nchunks, c_begin + a.chunks_per_split);
    // A operands (comment at `px` below).  The first chunk's U is requested HERE, before the ~300 instructions of staging
    // index arithmetic: it depends on the block's channel tile only
    f4 ureg[2][PSK / 4][MR];
    const wrsrc_t ur = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.uhat), (short)0, (int)((size_t)a.mblocks * a.nchunks * UF4 * 16), 0x00020000);
    const int uoff = ((kk * 4 + wave) * MT + n) * 16;
    auto load_u = [&](int mb, int c, f4 (*dst)[MR]) {
#pragma unroll
        for (int ks = 0; ks < PSK / 4; ++ks)
#pragma unroll
            for (int i = 0; i < MR; ++i)
                dst[ks][i] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(ur, uoff + (ks * 16 * MT + i * 16) * 16,
                                                                                          (mb * a.nchunks + c) * (UF4 * 16), 0));
    };
    if (c_begin < c_end) load_u(mblk, c_begin, ureg[0]);

    // ---- staging role: thread -> (sub-region, slab row, column pair) and PK channels of the chunk.  G = 2: the two thread
    // halves take the two sub-regions, all PSK channels each.  G = 1: they take the two channel halves of the one sub-region
    // (with waves 0-1 staging everything, waves 2-3 sat at the barrier for the length of 16 LDS writes and 8 load issues per chunk)
    constexpr int PK = G == 1 ? PSK / 2 : PSK;
    const int sg_ = G == 1 ? 0 : tid >> 7, sp = tid & 127;
    const int kh0 = G == 1 ? (wave >> 1) * PK : 0;       // first channel of the chunk this thread stages
    const int PR = RW + 2, SR = 2 * RH + 2;
    const int sr = fdiv(sp, a.mg_PR), scp = sp - sr * PR;
    // image position of this thread's column pair (ix even) and the source it is read from:
    //   plain: zero padding, P = 1.  FUSED: ReflectionPad2d(1) folds row -1 -> 1, H -> H-2 and redirects the two border
    //   pairs to the aligned pair that holds the mirrored column (pair (-2,-1) -> (0,1): .y lands on slab column 0;
    //   pair (W,W+1) -> (W-2,W-1): .x lands on the column of W); nearest-2x upsampling reads x0[iy>>1][ix>>1] once and
    //   duplicates it; P = 2 shifts the patch origin for the full correlation (slab column 0 = image column X0-2).
    const int P = FUSED ? a.P : 1;
    const bool swr = sg_ < G && sp < SR * PR;
    const unsigned plane = (unsigned)(H * W) * 4u;                       // full-resolution plane bytes
    const int up0 = FUSED ? a.up0 : 0;
    const int C0 = FUSED ? a.C0 : a.K;
    const unsigned plane0 = up0 ? (unsigned)((H >> 1) * (W >> 1)) * 4u : plane;
    // byte offsets of this thread's pair inside source 0 (x / x0) and source 1 (x1) for the item with tile block `tb`
    // (0x80000000: outside the image / the item list -- the buffer load returns 0)
    auto item_src = [&](int tb, bool valid, unsigned& vo0, unsigned& vo1) {
        const int ssub = tb * G + sg_;
        const bool s_act = valid && sg_ < G && ssub < a.nsub && sp < SR * PR;
        const int sq = s_act ? ssub : 0;
        const int sb = fdiv(sq, a.mg_per_img), srq = sq - sb * per_img;
        const int sry = fdiv(srq, a.mg_regs_x), srx = srq - sry * a.regs_x;
        const int iy = sry * RH * 2 - P + sr, ix = srx * RW * 2 - 2 + 2 * scp;
        int sy = iy, sx = ix;
        if (FUSED && a.pad == PAD_REFLECT) {
            sy = iy == -1 ? 1 : (iy == H ? H - 2 : iy);
            sx = ix == -2 ? 0 : (ix == W ? W - 2 : ix);
        }
        const bool sok = s_act && sy >= 0 && sy < H && sx >= 0 && sx < W;
        vo0 = sok ? (unsigned)sb * (unsigned)C0 * plane0 +
                        (up0 ? (unsigned)((sy >> 1) * (W >> 1) + (sx >> 1)) : (unsigned)(sy * W + sx)) * 4u
                  : 0x80000000u;
        vo1 = (FUSED && sok) ? (unsigned)sb * (unsigned)(a.K - C0) * plane + (unsigned)(sy * W + sx) * 4u : 0x80000000u;
    };
    unsigned svoff, svoff1;
    item_src(tblk, true, svoff, svoff1);
    // MODE 1: (BatchNorm group of this wave's sub-region) * K -- the thread halves / channel halves that stage are whole waves, so
    // the scale / shift reads below are scalar loads
    int bn_gofs = 0;
    if constexpr (MODE == 1) {
        const int ssub = min(tblk * G + (G == 1 ? 0 : (wave >> 1)), a.nsub - 1);
        bn_gofs = __builtin_amdgcn_readfirstlane((fdiv(ssub, a.mg_per_img) / a.npg) * a.K);
    }
    const bool bn_inside = svoff != 0x80000000u;
    const int shift = 2 - P;                          // slab column of the pair's first element = 2 scp - shift
    const int slds0 = sg_ * SUBS + sr * RS + max(2 * scp - shift, 0), slds1 = sg_ * SUBS + sr * RS + 2 * scp + 1 - shift;
    const wrsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), (short)0, (int)a.xbytes, 0x00020000);
    const wrsrc_t x1r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(FUSED ? a.x1 : a.x), (short)0,
                                                          (int)(FUSED ? a.x1bytes : a.xbytes), 0x00020000);

    // ---- compute role: wave = Winograd row; lane = (tile slot n + 16 j, reduction channel kk)
    const int ra = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
    const int rb = wave == 3 ? 3 : (wave == 2 ? 1 : 2);
    const float sgn = wave == 1 ? 1.f : -1.f;         // t = d[ra] + sgn * d[rb]
    int offA[2], offB[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int local = n + 16 * h;
        const int tl = local < RH * RW ? local : 0;
        const int ty = fdiv(tl, a.mg_RW), tx = tl - ty * RW;
        offA[h] = kk * CPS + (2 * ty + ra) * RS + 2 * tx;
        offB[h] = kk * CPS + (2 * ty + rb) * RS + 2 * tx;
    }

    f4 acc[MR][NR][4];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[i][j][q] = f4{0.f, 0.f, 0.f, 0.f};

    f2w px[2][PK];                                   // x of chunk c lives in px[(c - c_begin) & 1], two chunks in flight
    // A operands: the four waves need DISJOINT quarters of a U chunk (wave = Winograd row = pq), so U never goes through
    // LDS: each lane loads its own 16 bytes per k-step and 16-channel block straight into the registers the MFMAs read,
    // one chunk ahead (16 lanes x 16 B = 256-byte rows; the other tile blocks of this channel block find them in L2)
    auto load_x = [&](unsigned so0, unsigned so1, int c, f2w* dst) {
        {
            // ONE load shape for every kind of chunk -- eight 8-byte buffer loads, the source picked by scalar selects (chunks
            // never straddle the concat: C0 % PSK == 0, host-checked).  Separate code paths per source (dword loads for the
            // upsampled half, two descriptors) made the number of outstanding loads path-dependent for the compiler, which then
            // waited for ALL of them (vmcnt(0)) in front of every LDS commit: the two-chunk prefetch distance was gone in the
            // decoder's kernels.  The upsampled half reads 8 bytes at x0[iy>>1][ix>>1] and uses the first 4 (commit_x).
            const int ch0 = c * PSK;
            const bool from1 = FUSED && ch0 >= C0;
            const wrsrc_t rs = from1 ? x1r : xr;
            const unsigned pl = from1 ? plane : plane0, vbase = from1 ? so1 : so0;
            const int chb = (from1 ? ch0 - C0 : ch0) + kh0, climit = from1 ? a.K - C0 : C0;
#pragma unroll
            for (int k = 0; k < PK; ++k) {
                const int ch = chb + k;
                const unsigned vo = ch < climit ? vbase : 0x80000000u;
                dst[k] = __builtin_bit_cast(f2w, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)vo, (int)((unsigned)ch * pl), 0));
            }
        }
    };
    // MODE 1: scale / shift of the chunk that is committed next (wave-uniform addresses: scalar loads), requested one compute
    // phase ahead of their use
    float bnsc[MODE == 1 ? PK : 1], bnsh[MODE == 1 ? PK : 1];
    auto load_bn = [&](int c) {
        if constexpr (MODE == 1) {
            const int ch0 = min(c * PSK + kh0, a.K - PK);
#pragma unroll
            for (int k = 0; k < PK; ++k) { bnsc[k] = a.in_scale[bn_gofs + ch0 + k]; bnsh[k] = a.in_shift[bn_gofs + ch0 + k]; }
        }
    };
    auto commit_x = [&](int buf, const f2w* src, int c) {
        if (swr) {
            float* xw = xl[buf];
            const bool dup = FUSED && up0 && c * PSK < C0;        // chunk of the nearest-x2 upsampled source: one value, two columns
            if constexpr (MODE == 1) {
                // (threads whose pair lies outside the image -- zero padding, lanes past the item list -- zeroed their slab
                // positions once, in the prologue, and store nothing here; K % PSK == 0, host-checked)
                if (bn_inside) {
#pragma unroll
                    for (int k = 0; k < PK; ++k) {
                        const f2w v = src[k] * f2w{bnsc[k], bnsc[k]} + f2w{bnsh[k], bnsh[k]};
                        xw[(kh0 + k) * CPS + slds0] = fmaxf(v.x, 0.f);
                        xw[(kh0 + k) * CPS + slds1] = fmaxf(v.y, 0.f);
                    }
                }
            } else {
#pragma unroll
                for (int k = 0; k < PK; ++k) {
                    xw[(kh0 + k) * CPS + slds0] = src[k].x;
                    xw[(kh0 + k) * CPS + slds1] = dup ? src[k].x : src[k].y;
                }
            }
        }
    };

    auto compute = [&](int buf, const f4 (*ua)[MR]) {
        const float* xs = xl[buf];
        f2w raw[2][4];                               // [buffer][row a lo, row a hi, row b lo, row b hi]
        auto read_raw = [&](int s, f2w* dst) {       // step s = ks * NR + j
            const int ks = s / NR, j = s % NR;
            const float* base = xs + ks * 4 * CPS + (j >> 1) * SUBS;
            dst[0] = *reinterpret_cast<const f2w*>(base + offA[j & 1]);
            dst[1] = *reinterpret_cast<const f2w*>(base + offA[j & 1] + 2);
            dst[2] = *reinterpret_cast<const f2w*>(base + offB[j & 1]);
            dst[3] = *reinterpret_cast<const f2w*>(base + offB[j & 1] + 2);
        };
        constexpr int STEPS = (PSK / 4) * NR;
        read_raw(0, raw[0]);
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const int ks = s / NR, j = s % NR;
            if (s + 1 < STEPS) read_raw(s + 1, raw[(s + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            const f2w* rw = raw[s & 1];
            const float t0 = fmaf(rw[2].x, sgn, rw[0].x), t1 = fmaf(rw[2].y, sgn, rw[0].y);
            const float t2 = fmaf(rw[3].x, sgn, rw[1].x), t3 = fmaf(rw[3].y, sgn, rw[1].y);
            const float v0 = t0 - t2, v1 = t1 + t2, v2 = t2 - t1, v3 = t1 - t3;
#pragma unroll
            for (int i = 0; i < MR; ++i) {
                const f4 u = ua[ks][i];
                acc[i][j][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(u.x, v0, acc[i][j][0], 0, 0, 0);
                acc[i][j][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(u.y, v1, acc[i][j][1], 0, 0, 0);
                acc[i][j][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(u.z, v2, acc[i][j][2], 0, 0, 0);
                acc[i][j][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(u.w, v3, acc[i][j][3], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // The slab is double-buffered in LDS (one barrier per chunk), U lives in registers.  In flight while chunk c is
    // multiplied: U of c+1 (L2-resident) and x of c+1 and c+2 (first touch comes from HBM).
    const int nloc = c_end - c_begin;
    const int grid = gridDim.x;
    const bool persist = PERSIST && grid < a.items;  // (host: gridDim.z == 1, nloc even and >= 4; PERSIST instantiations only)
    int round = 0;
    if constexpr (MODE == 1) {
        if (swr && !bn_inside) {
#pragma unroll
            for (int k = 0; k < PK; ++k) {
                xl[0][(kh0 + k) * CPS + slds0] = 0.f; xl[0][(kh0 + k) * CPS + slds1] = 0.f;
                xl[1][(kh0 + k) * CPS + slds0] = 0.f; xl[1][(kh0 + k) * CPS + slds1] = 0.f;
            }
        }
        load_bn(c_begin);
    }
    if (nloc > 0) {
        load_x(svoff, svoff1, c_begin, px[0]);
        if (nloc > 1) load_x(svoff, svoff1, c_begin + 1, px[1]);
        commit_x(0, px[0], c_begin);
        if (nloc > 2) load_x(svoff, svoff1, c_begin + 2, px[0]);
    }
    __syncthreads();
#ifdef WINO_DIAG
    unsigned long long dg[5] = {0, 0, 0, 0, 0};
    const unsigned long long dg0 = __builtin_amdgcn_s_memtime();
    unsigned long long dg_loop_end = 0;
#endif
#if defined(WINO_DIAG) && WINO_DIAG == 1             // WINO_DIAG=2: only prologue / loop / epilogue (the inner timers cost ~20 % themselves)
#define DIAG_T(k, stmt) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); stmt; dg[k] += __builtin_amdgcn_s_memtime() - t_; }
#define DIAG_WAIT() __builtin_amdgcn_s_waitcnt(0xc07f)   /* lgkmcnt(0): charge the LDS writes to the commit phase */
#else
#define DIAG_T(k, stmt) { stmt; }
#define DIAG_WAIT()
#endif
  for (;;) {                                          // work items of this block (one, unless the launch is persistent)
    // the next item's sources: its loads are issued by the last steps of this item's pipeline
    // (round r hands out the items [r grid, (r+1) grid): the last, partial round goes to the blocks with the LOWEST hardware
    // ids -- round-robin over the XCDs and breadth-first over their CUs, i.e. spread over the chip -- in XCD-local order)
    const int n_next = min(grid, a.items - (round + 1) * grid);
    const bool has_next = PERSIST && persist && (int)blockIdx.x < n_next;
    const int item_n = has_next ? (round + 1) * grid + xcd_logical_block(blockIdx.x, n_next) : item;
    int mblk_n = mblk, tblk_n = tblk;
    if (has_next) item_blocks(item_n, mblk_n, tblk_n);
    unsigned svoff_n, svoff1_n;
    item_src(tblk_n, has_next, svoff_n, svoff1_n);
    auto step = [&](int i, f2w* pxn, f4 (*ucur)[MR], f4 (*unxt)[MR]) {   // pxn holds x of chunk i+1
        const bool more = i + 1 < nloc;
        if (more || has_next) load_u(more ? mblk : mblk_n, more ? c_begin + i + 1 : c_begin, unxt);
        if (more) load_bn(c_begin + i + 1);
        DIAG_T(0, compute(i & 1, ucur));
        if (more) {
            DIAG_T(1, commit_x((i + 1) & 1, pxn, c_begin + i + 1); DIAG_WAIT());
            // chunk i+3 of this item -- or, at the end of the item, chunk 0 / 1 of the next one (same registers, same parity:
            // nloc is even), whose commit waits behind the epilogue below
            const bool in_cur = i + 3 < nloc;
            DIAG_T(2, if (in_cur || has_next) load_x(in_cur ? svoff : svoff_n, in_cur ? svoff1 : svoff1_n,
                                                     in_cur ? c_begin + i + 3 : c_begin + i + 3 - nloc, pxn));
        }
        DIAG_T(3, __syncthreads());
    };
    for (int i = 0; i < nloc; i += 2) {
        step(i, px[1], ureg[0], ureg[1]);
        if (i + 1 < nloc) step(i + 1, px[0], ureg[1], ureg[0]);
    }

#ifdef WINO_DIAG
    if (item == lbid) {
        dg_loop_end = __builtin_amdgcn_s_memtime();
        dg[4] = dg_loop_end - dg0;
    }
#endif
    // ---- combine the four rows: wave a contributes z[a][jj] = sum_b M[a][b] A[b][jj]; Y[0] = z0+z1+z2, Y[1] = z1-z2-z3
    float* ex = &xl[0][0];                                       // [wave][(j*4 + r)*2 + jj][lane]  (the slabs are dead now)
    float* yout = a.y + (size_t)blockIdx.z * a.slab_stride;
    // the (sub-region, tile) this lane stores in the exchange round: slot j = wave % NR, and -- NR = 2: the four waves share
    // the two slots -- the RPW = NR of the four r (channel) values starting at (wave / NR) * NR; every wave stores
    constexpr int RPW = NR;
    const int oj = wave % NR, r0 = (wave / NR) * RPW;
    const int osub = tblk * G + (oj >> 1);
    const int olocal = n + 16 * (oj & 1);
    const bool o_act = osub < a.nsub && olocal < RH * RW;
    const int oq = o_act ? osub : 0;
    const int ob = fdiv(oq, a.mg_per_img), orq = oq - ob * per_img;
    const int ory = fdiv(orq, a.mg_regs_x), orx = orq - ory * a.regs_x;
    const int otl = o_act ? olocal : 0;
    const int oty = fdiv(otl, a.mg_RW), otx = otl - oty * RW;
    const int oy = ory * RH * 2 + 2 * oty, ox = orx * RW * 2 + 2 * otx;
    const int Ho = FUSED ? a.Ho : H, Wo = FUSED ? a.Wo : W;
    const bool o_ok = o_act && oy < Ho && ox < Wo;
    const bool finish = FUSED && gridDim.z == 1;      // bias + activation here unless wino_ysum_kernel still has to add slabs
#pragma unroll
    for (int i = 0; i < MR; ++i) {
        if (i > 0) __syncthreads();
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float m0 = acc[i][j][0][r], m1 = acc[i][j][1][r], m2 = acc[i][j][2][r], m3 = acc[i][j][3][r];
                ex[((wave * NR * 4 + j * 4 + r) * 2 + 0) * 64 + lane] = m0 + m1 + m2;
                ex[((wave * NR * 4 + j * 4 + r) * 2 + 1) * 64 + lane] = m1 - m2 - m3;
            }
        // (plain launches) the other gradient of a residual fork: loaded before the barrier, so that the loads fly while the
        // rows are exchanged, and added below
        f2w ad[RPW][2];
        const bool has_add = !FUSED && a.addend && gridDim.z == 1;
        if (has_add) {
#pragma unroll
            for (int rr = 0; rr < RPW; ++rr) {
                const int m = min(mblk * MT + i * 16 + kk * 4 + r0 + rr, a.M - 1);
                const size_t o = (((size_t)ob * a.M + m) * Ho + min(oy, Ho - 1)) * Wo + min(ox, Wo - 2);
                ad[rr][0] = *reinterpret_cast<const f2w*>(a.addend + o);
                ad[rr][1] = *reinterpret_cast<const f2w*>(a.addend + o + (oy + 1 < Ho ? Wo : 0));
            }
        }
        // BatchNorm epilogues: the lane's 2 x 2 block of the BatchNorm's raw input and the row constants, requested before the barrier too
        constexpr bool BNE = MODE == 2 || MODE == 3;
        const bool stats = MODE <= 1 && a.stat_part != nullptr && gridDim.z == 1;
        f2w bx[BNE ? RPW : 1][2];
        float bmean[BNE ? RPW : 1], bsc[BNE ? RPW : 1], bsh[BNE ? RPW : 1];
        unsigned bbits[BNE ? RPW : 1];
        if constexpr (BNE) {
            const int gtab = (ob / a.npg) * a.M;
#pragma unroll
            for (int rr = 0; rr < RPW; ++rr) {
                const int m = min(mblk * MT + i * 16 + kk * 4 + r0 + rr, a.M - 1);
                const int e0 = min(oy, Ho - 1) * Wo + min(ox, Wo - 2), e1 = e0 + (oy + 1 < Ho ? Wo : 0);
                const size_t pl = (size_t)ob * a.M + m;
                bx[rr][0] = *reinterpret_cast<const f2w*>(a.bn_x + pl * Ho * Wo + e0);
                bx[rr][1] = *reinterpret_cast<const f2w*>(a.bn_x + pl * Ho * Wo + e1);
                bmean[rr] = a.bn_mean[gtab + m];
                if constexpr (MODE == 2) { bsc[rr] = a.in_scale[gtab + m]; bsh[rr] = a.in_shift[gtab + m]; }
                if constexpr (MODE == 3) {
                    // bit l of word w of a 256-element block <-> element 4 l + w (bn_apply_kernel's ballots); e0, e1 are even
                    const unsigned* mw = reinterpret_cast<const unsigned*>(a.bn_mask + pl * ((Ho * Wo + 255) >> 8) * 4);
                    const int l0 = (e0 & 255) >> 2, l1 = (e1 & 255) >> 2;
                    const unsigned* w0 = mw + ((e0 >> 8) * 4 + (e0 & 3)) * 2 + (l0 >> 5);
                    const unsigned* w1 = mw + ((e1 >> 8) * 4 + (e1 & 3)) * 2 + (l1 >> 5);
                    bbits[rr] = ((w0[0] >> (l0 & 31)) & 1u) | (((w0[2] >> (l0 & 31)) & 1u) << 1) |
                                (((w1[0] >> (l1 & 31)) & 1u) << 2) | (((w1[2] >> (l1 & 31)) & 1u) << 3);
                }
            }
        }
        __syncthreads();
        if (BNE || stats) {
            // every lane runs this form (the 16 lanes of a DPP row hold 16 tiles of ONE output channel and are summed across)
            const bool row1 = oy + 1 < Ho;
            const int slot = tblk * NR + oj;
#pragma unroll
            for (int rr = 0; rr < RPW; ++rr) {
                const int r = r0 + rr;
                const int m = mblk * MT + i * 16 + kk * 4 + r;
                const bool mok = m < a.M;
                float z[4][2];
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    z[w][0] = ex[((w * NR * 4 + oj * 4 + r) * 2 + 0) * 64 + lane];
                    z[w][1] = ex[((w * NR * 4 + oj * 4 + r) * 2 + 1) * 64 + lane];
                }
                float y00 = z[0][0] + z[1][0] + z[2][0], y01 = z[0][1] + z[1][1] + z[2][1];
                float y10 = z[1][0] - z[2][0] - z[3][0], y11 = z[1][1] - z[2][1] - z[3][1];
                if (has_add) { y00 += ad[rr][0].x; y01 += ad[rr][0].y; y10 += ad[rr][1].x; y11 += ad[rr][1].y; }
                float sv, qv;
                if constexpr (BNE) {
                    bool k00, k01, k10, k11;
                    if constexpr (MODE == 2) {
                        k00 = fmaf(bx[rr][0].x, bsc[rr], bsh[rr]) > 0.f; k01 = fmaf(bx[rr][0].y, bsc[rr], bsh[rr]) > 0.f;
                        k10 = fmaf(bx[rr][1].x, bsc[rr], bsh[rr]) > 0.f; k11 = fmaf(bx[rr][1].y, bsc[rr], bsh[rr]) > 0.f;
                    } else {
                        k00 = bbits[rr] & 1u; k01 = bbits[rr] & 2u; k10 = bbits[rr] & 4u; k11 = bbits[rr] & 8u;
                    }
                    y00 = k00 ? y00 : 0.f; y01 = k01 ? y01 : 0.f; y10 = k10 ? y10 : 0.f; y11 = k11 ? y11 : 0.f;
                    const float mu = bmean[rr];
                    sv = (y00 + y01) + (row1 ? y10 + y11 : 0.f);
                    qv = fmaf(y00, bx[rr][0].x - mu, y01 * (bx[rr][0].y - mu)) + (row1 ? fmaf(y10, bx[rr][1].x - mu, y11 * (bx[rr][1].y - mu)) : 0.f);
                } else {
                    sv = (y00 + y01) + (row1 ? y10 + y11 : 0.f);
                    qv = fmaf(y00, y00, y01 * y01) + (row1 ? fmaf(y10, y10, y11 * y11) : 0.f);
                }
                if (!o_ok) { sv = 0.f; qv = 0.f; }
                sv = wrow16_sum(sv); qv = wrow16_sum(qv);
                if (n == 0 && mok) {
                    float* pp = BNE ? a.bwd_part + ((size_t)m * a.bwd_nparts + slot) * 2 : a.stat_part + ((size_t)m * a.stat_nparts + slot) * 2;
                    *reinterpret_cast<f2w*>(pp) = f2w{sv, qv};
                }
                if (o_ok && mok) {
                    float* dst = yout + (((size_t)ob * a.M + m) * Ho + oy) * Wo + ox;
                    *reinterpret_cast<f2w*>(dst) = f2w{y00, y01};
                    if (row1) *reinterpret_cast<f2w*>(dst + Wo) = f2w{y10, y11};
                }
            }
        } else if (o_ok) {
#pragma unroll
            for (int rr = 0; rr < RPW; ++rr) {
                const int r = r0 + rr;
                const int m = mblk * MT + i * 16 + kk * 4 + r;
                if (m >= a.M) continue;
                float z[4][2];
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    z[w][0] = ex[((w * NR * 4 + oj * 4 + r) * 2 + 0) * 64 + lane];
                    z[w][1] = ex[((w * NR * 4 + oj * 4 + r) * 2 + 1) * 64 + lane];
                }
                float y00 = z[0][0] + z[1][0] + z[2][0], y01 = z[0][1] + z[1][1] + z[2][1];
                float y10 = z[1][0] - z[2][0] - z[3][0], y11 = z[1][1] - z[2][1] - z[3][1];
                if (finish) {
                    const float bv = a.bias ? a.bias[m] : 0.f;
                    y00 = act_fwd(y00 + bv, a.act); y01 = act_fwd(y01 + bv, a.act);
                    y10 = act_fwd(y10 + bv, a.act); y11 = act_fwd(y11 + bv, a.act);
                }
                if constexpr (MODE == 4) {
                    const bool row1 = oy + 1 < Ho;
                    if (m < a.split_c0 ? a.y == nullptr : a.out1 == nullptr) continue;       // (that input needs no gradient)
                    if (m < a.split_c0 && a.split_up) {              // (Ho, Wo even: the whole 2 x 2 tile is inside)
                        const size_t o = (((size_t)ob * a.split_c0 + m) * (Ho >> 1) + (oy >> 1)) * (Wo >> 1) + (ox >> 1);
                        float v = (y00 + y01) + (y10 + y11);
                        if (a.add0) v += a.add0[o];
                        a.y[o] = v;
                    } else {
                        const bool first = m < a.split_c0;
                        const int mm = first ? m : m - a.split_c0, MM = first ? a.split_c0 : a.M - a.split_c0;
                        const size_t o = (((size_t)ob * MM + mm) * Ho + oy) * Wo + ox;
                        const float* ap = first ? a.add0 : a.add1;
                        float* dst = (first ? a.y : a.out1) + o;
                        if (ap) {
                            const f2w a0 = *reinterpret_cast<const f2w*>(ap + o);
                            y00 += a0.x; y01 += a0.y;
                            if (row1) { const f2w a1 = *reinterpret_cast<const f2w*>(ap + o + Wo); y10 += a1.x; y11 += a1.y; }
                        }
                        *reinterpret_cast<f2w*>(dst) = f2w{y00, y01};
                        if (row1) *reinterpret_cast<f2w*>(dst + Wo) = f2w{y10, y11};
                    }
                    continue;
                }
                const size_t o = (((size_t)ob * a.M + m) * Ho + oy) * Wo + ox;
                if (has_add) { y00 += ad[rr][0].x; y01 += ad[rr][0].y; y10 += ad[rr][1].x; y11 += ad[rr][1].y; }
                float* dst = yout + o;
                *reinterpret_cast<f2w*>(dst) = f2w{y00, y01};
                if (oy + 1 < Ho) *reinterpret_cast<f2w*>(dst + Wo) = f2w{y10, y11};
            }
        }
    }
    if (!has_next) break;
    // ---- on to the next item: its chunks 0 and 1 are in px[0] / px[1] (requested by the last steps above), its first U chunk
    // in ureg[0]; the exchange buffer is the slab pair, so chunk 0 is committed only now, then chunk 2 is requested -- the
    // state a block's prologue leaves
    __syncthreads();
    commit_x(0, px[0], c_begin);
    load_x(svoff_n, svoff1_n, c_begin + 2, px[0]);
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[i][j][q] = f4{0.f, 0.f, 0.f, 0.f};
    item = item_n; ++round; mblk = mblk_n; tblk = tblk_n; svoff = svoff_n; svoff1 = svoff1_n;
    __syncthreads();
  }
#ifdef WINO_DIAG
    if (a.diag && lane == 0 && wave == 0) {
        unsigned long long* o = a.diag + (size_t)(blockIdx.z * gridDim.x + blockIdx.x) * 8;
        for (int q = 0; q < 5; ++q) o[q] = dg[q];
        o[5] = dg0 - dg_start;                                  // prologue (index setup, first loads, first commit)
        o[6] = __builtin_amdgcn_s_memtime() - dg_loop_end;      // epilogue (row combine through LDS, stores issued)
        o[7] = dg_start;
#if WINO_DIAG == 2
        o[7] = dg_rstart;                                       // s_memrealtime: 100 MHz, the same counter on every CU (s_memtime is per CU)
        o[0] = __builtin_amdgcn_s_memrealtime();                // block end: (start, end) of every block -> residency timeline
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        o[1] = hwid;                                            // [11:8] CU, [15:13]... which SIMD/CU/SE the block ran on
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(hwid));
        o[2] = hwid;
#endif
    }
#endif
}

// y = act(sum of the K-split slabs (fixed order) + bias)
__global__ __launch_bounds__(256) void wino_ysum_kernel(const float* __restrict__ slabs, float* __restrict__ y, size_t n4,
                                                        size_t stride4, int ksplit, const float* __restrict__ bias, int act,
                                                        int plane4, int M, const float* __restrict__ addend) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256ull) {
        float4 v = reinterpret_cast<const float4*>(slabs)[i];
        for (int s = 1; s < ksplit; ++s) {
            const float4 t = reinterpret_cast<const float4*>(slabs)[i + (size_t)s * stride4];
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        if (bias || act != ACT_NONE) {
            const float bv = bias ? bias[(i / plane4) % M] : 0.f;
            v.x = act_fwd(v.x + bv, act); v.y = act_fwd(v.y + bv, act); v.z = act_fwd(v.z + bv, act); v.w = act_fwd(v.w + bv, act);
        }
        if (addend) {
            const float4 t = reinterpret_cast<const float4*>(addend)[i];
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        reinterpret_cast<float4*>(y)[i] = v;
    }
}

static inline size_t wino_al256(size_t v) { return (v + 255) & ~(size_t)255; }

int wino_ysum_launch(const float* slabs, float* y, size_t n4, int ksplit, const float* addend, hipStream_t st) {
    hipLaunchKernelGGL(wino_ysum_kernel, dim3((unsigned)std::min<size_t>((n4 + 255) / 256, 2048)), dim3(256), 0, st, slabs, y, n4, n4,
                       ksplit, (const float*)nullptr, (int)ACT_NONE, 1, 1, addend);
    DC_CHECK_LAUNCH();
    return DC_OK;
}

// sub-region shapes {RH, RW, RS}: RS makes the 16-lane ds_read_b64 groups of a patch row conflict-free
static void wino_ps_pick_region(int TH, int TW, int& RH, int& RW, int& RS) {
    const int cand[4][3] = {{4, 8, 24}, {2, 16, 36}, {3, 10, 26}, {8, 4, 12}};
    double best = -1.0;
    for (auto& c : cand) {
        const double covered = (double)ceil_div(TH, c[0]) * ceil_div(TW, c[1]) * 32.0;
        const double util = (double)TH * TW / covered;
        if (util > best + 1e-9) { best = util; RH = c[0]; RW = c[1]; RS = c[2]; }
    }
}

static inline int wino_wblocks(int Mp, int Kp) { return (Mp / 16) * ((Kp + 15) / 16); }   // 16 x 16 tiles of wino_weight_one
static inline size_t wino_uhat_bytes(int Ci, int Co) {
    const size_t a = (size_t)ceil_div(Ci, 32) * 32, b = (size_t)ceil_div(Co, 32) * 32;
    return wino_al256(a * b * 16 * sizeof(float));
}

// ---- transformed-weight cache ---------------------------------------------------------------------------------------
// A training step uses every convolution weight twice (forward: G g G^T, data gradient: the same of the rotated,
// transposed filter) and, in the sequence models, once per frame; the weights only change in the optimiser step.  The
// host registers the weights of a model once (dc_wino_cache_register), calls dc_wino_cache_refresh at the start of a
// step -- ONE launch that transforms every variant seen so far instead of one 8 us launch in front of every convolution
// -- and dc_wino_cache_invalidate when the step's backward is done.  Between the two, wino_launch takes U from the
// cache; a variant (dgrad, MT) it has not met yet is transformed in place as before and joins the next refresh.
struct WcVariant { int dgrad, MT, Mp, Kp; float* buf; bool fresh, in_table; int kind; };     // kind 0 Winograd U, 1 bf16 prepared weights, 2 split 1x1 weights
struct WcEntry { const float* w; int Ci, Co, owner; std::vector<WcVariant> v; };
// One descriptor table PER OWNER (= per model / Trainer).  A refresh transforms -- and a captured hipGraph replays the
// transform of -- the owner's own weights only, which the owner keeps alive; weights of another owner never enter its
// table.  (Round 3 had one table for the whole process: a graph captured, or a refresh skipped because the stream was
// capturing, while the table still named the weights of a model that had since been collected read freed memory -- a GPU
// page fault, which the HSA runtime turns into abort() of the process.  See DESIGN.md "The r3s abort".)
struct WcOwner {
    int id;
    bool valid = false, dirty = true;
    WinoWDesc* table = nullptr;
    int* b2d = nullptr;
    int table_n = 0, blocks = 0;
};
static std::mutex g_wc_mu;
static std::vector<WcEntry> g_wc;
static std::vector<WcOwner> g_wc_owners;
static int g_wc_next_owner = 1;
// Device buffers a captured hipGraph may still name in its kernel arguments (descriptor tables that were outgrown, the
// variant buffers of unregistered weights): parked here, released only by dc_wino_cache_clear().
static std::vector<void*> g_wc_retired;

static bool wc_capturing(hipStream_t st) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); return false; }
    return cs != hipStreamCaptureStatusNone;
}
static WcOwner* wc_owner(int id) {
    for (auto& o : g_wc_owners)
        if (o.id == id) return &o;
    return nullptr;
}

// -> cached U for this launch, or nullptr (then the caller transforms into its workspace).  Nothing is allocated while
// `st` is being captured (hipMalloc is illegal there): an unseen variant is then transformed per launch, as before.
static inline size_t wc_variant_bytes(int kind, int MT, int Mp, int Kp) {
    if (kind == 2) return (size_t)Mp * Kp * 3 * 2 + 256;
    return kind == 1 ? (size_t)Mp * Kp * 36 * MT * 16 : (size_t)Mp * Kp * 16 * sizeof(float);
}
static inline int wc_variant_blocks(int kind, int MT, int Mp, int Kp) {
    if (kind == 2) return ceil_div(Mp * (Kp / 4), 256);
    return kind == 1 ? ceil_div(Mp * Kp * 36 * MT, 256) : wino_wblocks(Mp, Kp);
}
static const float* wc_lookup_kind(int kind, const float* w, int Ci, int Co, bool dgrad, int MT, int Mp, int Kp, hipStream_t st) {
    std::lock_guard<std::mutex> lk(g_wc_mu);
    for (auto& e : g_wc) {
        if (e.w != w) continue;
        if (e.Ci != Ci || e.Co != Co) return nullptr;
        WcOwner* o = wc_owner(e.owner);
        if (!o) return nullptr;
        for (auto& v : e.v)
            if (v.kind == kind && v.dgrad == (int)dgrad && v.MT == MT) return (o->valid && v.fresh) ? v.buf : nullptr;
        if (wc_capturing(st)) return nullptr;
        WcVariant v{(int)dgrad, MT, Mp, Kp, nullptr, false, false, kind};
        if (hipMalloc((void**)&v.buf, wc_variant_bytes(kind, MT, Mp, Kp)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        e.v.push_back(v);
        o->dirty = true;
        return nullptr;
    }
    return nullptr;
}
static const float* wc_lookup(const float* w, int Ci, int Co, bool dgrad, int MT, int Mp, int Kp, hipStream_t st) {
    return wc_lookup_kind(0, w, Ci, Co, dgrad, MT, Mp, Kp, st);
}
const void* wc_lookup_c3b(const float* w, int Ci, int Co, int dgrad, int MT, int nmblk, int nchunks, hipStream_t st) {
    return wc_lookup_kind(1, w, Ci, Co, dgrad != 0, MT, nmblk, nchunks, st);
}
const void* wc_lookup_x3(const float* w, int Ci, int Co, int tr, int Mp, int K, hipStream_t st) {
    return wc_lookup_kind(2, w, Ci, Co, tr != 0, 0, Mp, K, st);
}

#ifdef WINO_DIAG
static unsigned long long* g_wino_diag = nullptr;
extern "C" void dc_wino_set_diag(void* p) { g_wino_diag = (unsigned long long*)p; }
unsigned long long* wino_diag_ptr() { return g_wino_diag; }
#endif

// ---- measurement hook ---------------------------------------------------------------------------------
namespace {
struct ConvProf {
    std::vector<hipEvent_t> e0, e1;
    int used = 0;
    double flops = 0.0, exec = 0.0, bytes = 0.0;
};
constexpr int NPROF = 8;       // 0 wino_ps, 1 wino_wgrad, 2 c3b_conv (bf16), 3 c3b_wgrad (bf16), 4 1x1 GEMM family, 5 cg_ (3x3 / 2), 6 stem (7x7 / 2), 7 g1x3 (split-operand 1x1 GEMMs)
ConvProf g_cprof[NPROF];
int g_cprof_cap = 0, g_cprof_every = 1;
unsigned g_cprof_seen[NPROF] = {0, 0, 0, 0, 0, 0, 0, 0};
}  // namespace

hipEvent_t conv_prof_begin(int kind, double algorithmic_flops, double executed_flops, double algorithmic_bytes, hipStream_t st) {
    ConvProf& d = g_cprof[kind];
    if (g_cprof_cap == 0 || (g_cprof_seen[kind]++ % (unsigned)g_cprof_every) != 0 || d.used >= g_cprof_cap) return nullptr;
    d.flops += algorithmic_flops; d.exec += executed_flops; d.bytes += algorithmic_bytes;
    (void)hipEventRecord(d.e0[d.used], st);
    return d.e1[d.used++];
}
void conv_prof_end(hipEvent_t e, hipStream_t st) {
    if (e) (void)hipEventRecord(e, st);
}

// Predicted duration (us) of one wino_ps_kernel launch, used to pick the tile variant v (0: 16 ch x 32 tiles, 4 blocks per
// CU; 1: 32 x 32, 3 per CU; 2: 32 x 64, 2 per CU) and the reduction split.  Per-block timelines (-DWINO_DIAG=2,
// tools/diag_wino.py) showed what the old rule ("32 x 64 while that fills 512 slots") missed: the launch runs in ROUNDS of
// 256 x blocks-per-CU blocks, the blocks of a round move in lock-step (same start, same length), and a last round that is
// 40 % full still costs 65 % of a full one (720 blocks of the 32 x 64 variant: 27 us + 18 us).  Model: a block takes
// P + chunks * (a + b * o) with o = blocks resident per CU; full rounds at o = blocks-per-CU, the remainder at its own o;
// a split adds the slab sum.  {P, a, b} fitted to the 132 measurements of tools/sweep_wino.py (22 step shapes x 6
// choices, rms error 7 %); the model's pick is within 0.3 % of the best measured choice summed over those shapes, 9 %
// below the old rule's.  Deterministic in the shape, so every rank takes the same summation order.
static double wino_ps_cost(int v, int ksplit, int nsub, int M, int nchunks, size_t nout) {
    // (round 5: the 32 x 64 variant's constants x 1.07 -- tools/sweep_wino.py again, after the epilogue changes of rounds 4-5: where the
    // model preferred it (64 -> 64 at 48 x 160 B = 24, 64 -> 128 at 48 x 160, 32 -> 96 at 96 x 320) the 32 x 32 variant measures 4-7 % faster)
    static const double P[3] = {3.28, 3.51, 4.33}, A[3] = {0.614, 0.598, 0.893}, Bc[3] = {0.213, 0.460, 0.994};
    static const int bpc[3] = {4, 3, 2}, mt[3] = {16, 32, 32}, g[3] = {1, 1, 2};
    const long blocks = (long)ceil_div(nsub, g[v]) * ceil_div(M, mt[v]) * ksplit;
    const int chunks = ceil_div(nchunks, ksplit);
    const long slots = 256L * bpc[v], full = blocks / slots, rem = blocks % slots;
    double t = (double)full * (P[v] + chunks * (A[v] + Bc[v] * bpc[v]));
    if (rem) t += P[v] + chunks * (A[v] + Bc[v] * (double)ceil_div((int)rem, 256));
    if (ksplit > 1) t += 6.5 + 0.124 * (double)nout * 4.0 * (ksplit + 1) * 1e-6;      // slab sum: reads ksplit slabs, writes one
    static const double pen2 = getenv("DC_WINO_V2_PENALTY") ? atof(getenv("DC_WINO_V2_PENALTY")) : 1.0;
    static const double pen0 = getenv("DC_WINO_V0_PENALTY") ? atof(getenv("DC_WINO_V0_PENALTY")) : 1.0;
    if (v == 2) t *= pen2;
    if (v == 0) t *= pen0;
    return t;
}

// Largest reduction split a launch may use: 2 in general; 4 or 8 for SMALL outputs (<= 2 MB: the deep, low-resolution layers
// at batch 1-3 -- 512 channels on a 6 x 20 map is ONE sub-region per image and 64 dependent chunks per block: 44 us for 3.6 us
// of matrix work; its slabs are a few hundred KB).  Workspace sizes follow the same rule (wino_slab_bytes).
static int wino_ksplit_cap(size_t nout) {
    static const size_t small = getenv("DC_WINO_KSCAP_BYTES") ? (size_t)atol(getenv("DC_WINO_KSCAP_BYTES")) : (size_t)(2u << 20);      // experiments
    return nout * 4 <= small ? 8 : 2;
}
static size_t wino_slab_bytes(size_t nout) { return (size_t)wino_ksplit_cap(nout) * nout * sizeof(float); }

static int g_wino_persist = [] { const char* f = getenv("DC_WINO_PERSIST"); return f ? atoi(f) : 0; }();      // dc_set_wino_persist

// One convolution launch: reduction over K = C0 + C1 source channels, M output channels.
struct WinoLaunch {
    const float* src0; int C0; int up0; const float* src1; int C1;     // input = cat(up2?(src0), src1), maps H x W
    const float* weight; int Co, Ci; bool dgrad;                         // nn.Conv2d weight (Co,Ci,3,3) and the transform
    const float* bias; int act, pad, P;                                  // FUSED options (P: 1 same, 2 full correlation)
    const float* addend;                                                 // plain launches: added to the result (may be null)
    // fused data gradient with the split store (wino_conv_dgrad_split): rows [0, split_c0) -> out (half resolution when split_up),
    // the rest -> out1; add0 / add1 nullable
    bool split = false; int split_c0 = 0, split_up = 0; float* out1 = nullptr; const float* add0 = nullptr; const float* add1 = nullptr;
    float* out; void* ws;
    int B, H, W, M;
    bool fused;
    const dc_bn_fold* bn;                                                // plain launches: BatchNorm folded in (may be null)
};

// sub-region shape, tile variant and reduction split of a launch: deterministic in the shape (wino_ps_cost below)
struct WinoPlan { int RH, RW, RS, regs_x, regs_y, nsub, nchunks, MT, G, ksplit; };
static WinoPlan wino_plan(int B, int K, int M, int Ho, int Wo) {
    WinoPlan p{};
    const int TH = ceil_div(Ho, 2), TW = Wo / 2;
    wino_ps_pick_region(TH, TW, p.RH, p.RW, p.RS);
    p.regs_x = ceil_div(TW, p.RW); p.regs_y = ceil_div(TH, p.RH); p.nsub = p.regs_x * p.regs_y * B;
    p.nchunks = ceil_div(K, PSK);
    // Tile variant and reduction split: the cheapest of {16x32, 32x32, 32x64} x {1, 2 slabs} under wino_ps_cost
    p.MT = 32; p.G = 2; p.ksplit = 1;
    const size_t nout = (size_t)B * M * Ho * Wo;
    const bool can_split = (Ho * Wo) % 4 == 0 && p.nchunks >= 2;
    const int ks_cap = can_split ? wino_ksplit_cap(nout) : 1;
    double best = 1e30;
    for (int v = 0; v < 3; ++v)
        for (int ks = 1; ks <= ks_cap; ks *= 2) {
            if (ks > 1 && p.nchunks < 2 * ks) break;          // at least two chunks per split: something to pipeline
            const double t = wino_ps_cost(v, ks, p.nsub, M, p.nchunks, nout);
            if (t < best) { best = t; p.MT = v == 0 ? 16 : 32; p.G = v == 2 ? 2 : 1; p.ksplit = ks; }
        }
    if (getenv("DC_WINO_DEBUG"))       // experiments: the cost model's inputs and pick (tools/sweep_wino.py -> refits of wino_ps_cost)
        fprintf(stderr, "wino_plan B=%d K=%d M=%d %dx%d nsub=%d nchunks=%d nout=%zu cap=%d pick MT=%d G=%d ks=%d\n", B, K, M, Ho, Wo, p.nsub,
                p.nchunks, nout, ks_cap, p.MT, p.G, p.ksplit);
    if (const char* f = getenv("DC_WINO_FORCE")) {       // experiments: "MR,NR,ksplit" (tools/sweep_wino.py)
        int mr = 0, nr = 0, fks = 0;
        if (sscanf(f, "%d,%d,%d", &mr, &nr, &fks) >= 2 && (mr == 1 || mr == 2) && (nr == 2 || nr == 4) && !(mr == 1 && nr == 4)) {
            p.MT = 16 * mr; p.G = nr / 2;
            if (fks > 0) p.ksplit = std::max(1, std::min({fks, ks_cap, p.nchunks / 2}));
        }
    }
    return p;
}

static int wino_launch(const WinoLaunch& d, hipStream_t st) {
    const int K = d.C0 + d.C1, M = d.M, H = d.H, W = d.W;
    const int Ho = H + 2 * d.P - 2, Wo = W + 2 * d.P - 2;
    const size_t b0 = (size_t)d.B * d.C0 * (H >> d.up0) * (W >> d.up0) * 4, b1 = (size_t)d.B * d.C1 * H * W * 4;
    if (b0 >= 0x7fffffffull || b1 >= 0x7fffffffull) return DC_EINVAL;      // 32-bit buffer offsets
    WinoPsArgs a{};
    a.x = d.src0; a.x1 = d.src1; a.uhat = (const float*)d.ws; a.bias = d.bias; a.addend = d.fused ? nullptr : d.addend;
    a.B = d.B; a.K = K; a.M = M; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo;
    a.C0 = d.C0; a.up0 = d.up0; a.pad = d.pad; a.P = d.P; a.act = d.act;
    a.xbytes = (unsigned)b0; a.x1bytes = (unsigned)b1;
    const WinoPlan pl = wino_plan(d.B, K, M, Ho, Wo);
    a.RH = pl.RH; a.RW = pl.RW; a.RS = pl.RS;
    a.SUBS = (2 * a.RH + 2) * a.RS;
    a.regs_x = pl.regs_x; a.regs_y = pl.regs_y; a.nsub = pl.nsub;
    a.nchunks = pl.nchunks;
    const int MT = pl.MT, G = pl.G, ksplit = pl.ksplit;
    const size_t nout = (size_t)d.B * M * Ho * Wo;
    // BatchNorm fold (plain launches): 1 loader, 2 / 3 data-gradient epilogue; the statistics epilogue rides on mode 0 / 1
    int mode = 0;
    if (d.bn && !d.fused) {
        const dc_bn_fold* bn = d.bn;
        if (bn->groups < 1 || bn->groups > 2 || d.B % bn->groups) return DC_EINVAL;
        a.npg = d.B / bn->groups;
        a.in_scale = bn->in_scale; a.in_shift = bn->in_shift;
        if (bn->bwd_part) {
            if (ksplit > 1 || !bn->bn_x || !bn->bn_mean || (!bn->bn_mask && (!bn->in_scale || !bn->in_shift)) || (bn->bn_mask && ((H * W) & 3)))
                return DC_EINVAL;
            a.bn_x = bn->bn_x; a.bn_mean = bn->bn_mean; a.bn_mask = (const unsigned long long*)bn->bn_mask; a.bwd_part = bn->bwd_part;
            a.bwd_nparts = 2 * ceil_div(pl.nsub, G) * G;
            mode = bn->bn_mask ? 3 : 2;
        } else if (bn->in_scale) {
            if (!bn->in_shift || K % PSK) return DC_EINVAL;
            mode = 1;
        }
        if (bn->stat_part) {
            if (ksplit > 1 || mode >= 2) return DC_EINVAL;
            a.stat_part = bn->stat_part; a.stat_nparts = 2 * ceil_div(pl.nsub, G) * G;
        }
    }
    const int Mp = ceil_div(M, MT) * MT, Kp = a.nchunks * PSK;
    a.chunks_per_split = ceil_div(a.nchunks, ksplit);
    float* slabs = (float*)((char*)d.ws + wino_uhat_bytes(d.Ci, d.Co));
    a.y = ksplit > 1 ? slabs : d.out;
    a.slab_stride = ksplit > 1 ? nout : 0;
    if (const float* cached = wc_lookup(d.weight, d.Ci, d.Co, d.dgrad, MT, Mp, Kp, st)) {
        a.uhat = cached;
    } else {
        if (d.dgrad)
            hipLaunchKernelGGL((wino_weights_kernel<true>), dim3(wino_wblocks(Mp, Kp)), dim3(256), 0, st, d.weight, (float*)d.ws, d.Co, d.Ci, MT, Mp, Kp, PSK);
        else
            hipLaunchKernelGGL((wino_weights_kernel<false>), dim3(wino_wblocks(Mp, Kp)), dim3(256), 0, st, d.weight, (float*)d.ws, d.Co, d.Ci, MT, Mp, Kp, PSK);
        DC_CHECK_LAUNCH();
    }
#ifdef WINO_DIAG
    a.diag = g_wino_diag;
#endif
    a.tblocks = ceil_div(a.nsub, G); a.mblocks = Mp / MT;
    a.mg_mblocks = fdiv_magic(a.mblocks); a.mg_tblocks = fdiv_magic(a.tblocks); a.mg_per_img = fdiv_magic(a.regs_x * a.regs_y);
    a.mg_regs_x = fdiv_magic(a.regs_x); a.mg_PR = fdiv_magic(a.RW + 2); a.mg_RW = fdiv_magic(a.RW);
    // fdiv is exact for dividend * divisor < 2^32: block index by mblocks / tblocks, sub-region index by per_img / regs_x
    if ((unsigned long long)a.tblocks * a.mblocks * (unsigned)std::max(a.mblocks, a.tblocks) >= 0xffffffffull ||
        (unsigned long long)(a.nsub + 2 * G) * (unsigned)(a.regs_x * a.regs_y) >= 0xffffffffull) return DC_EINVAL;
    a.m_fast = (size_t)d.B * H * W >= (size_t)M * 16 ? 1 : 0;     // x stream (per reduction channel) vs U stream
    a.items = a.tblocks * a.mblocks;
    // Persistent form (wino_ps_kernel: one software pipeline over a block's items): only where a launch runs in more than one
    // round of resident blocks, on an unsplit reduction with an even number (>= 4) of chunks.
    int gx = a.items;
    {
        const int pmode = g_wino_persist;
        const int bpc = MT == 16 ? 4 : (G == 1 ? 3 : 2), slots = 256 * bpc;
        if (pmode == 1 && !d.fused && ksplit == 1 && a.nchunks >= 4 && (a.nchunks & 1) == 0 && a.items > slots) gx = slots;
    }
    if (mode != 0 || a.stat_part) gx = a.items;           // (the fold lives in the one-item-per-block form)
    const dim3 grid(gx, 1, ksplit);
    // algorithmic = 2 MAC of the direct convolution; executed = the 16 Winograd-domain GEMMs incl. tile padding
    hipEvent_t pe = conv_prof_begin(0, 2.0 * d.B * (double)M * K * 9.0 * H * W,
                                    2.0 * 16.0 * (double)a.nsub * 32.0 * (double)Mp * Kp,
                                    (double)b0 + (double)b1 + 4.0 * (double)nout + 36.0 * d.Co * d.Ci, st);
    if (d.fused && d.split) {
        if (ksplit != 1) return DC_EINVAL;                 // (wino_dgrad_split_ok: the split store exists for the unsplit reduction)
        a.out1 = d.out1; a.add0 = d.add0; a.add1 = d.add1; a.split_c0 = d.split_c0; a.split_up = d.split_up;
        if (MT == 16) hipLaunchKernelGGL((wino_ps_kernel<1, 2, true, false, 4>), grid, dim3(256), 0, st, a);
        else if (G == 1) hipLaunchKernelGGL((wino_ps_kernel<2, 2, true, false, 4>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((wino_ps_kernel<2, 4, true, false, 4>), grid, dim3(256), 0, st, a);
    } else if (d.fused) {
        if (MT == 16) hipLaunchKernelGGL((wino_ps_kernel<1, 2, true>), grid, dim3(256), 0, st, a);
        else if (G == 1) hipLaunchKernelGGL((wino_ps_kernel<2, 2, true>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((wino_ps_kernel<2, 4, true>), grid, dim3(256), 0, st, a);
    } else if (gx < a.items) {
        if (MT == 16) hipLaunchKernelGGL((wino_ps_kernel<1, 2, false, true>), grid, dim3(256), 0, st, a);
        else if (G == 1) hipLaunchKernelGGL((wino_ps_kernel<2, 2, false, true>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((wino_ps_kernel<2, 4, false, true>), grid, dim3(256), 0, st, a);
    } else {
#define WINO_PLAIN(MODE)                                                                                            \
        do {                                                                                                        \
            if (MT == 16) hipLaunchKernelGGL((wino_ps_kernel<1, 2, false, false, MODE>), grid, dim3(256), 0, st, a);    \
            else if (G == 1) hipLaunchKernelGGL((wino_ps_kernel<2, 2, false, false, MODE>), grid, dim3(256), 0, st, a); \
            else hipLaunchKernelGGL((wino_ps_kernel<2, 4, false, false, MODE>), grid, dim3(256), 0, st, a);             \
        } while (0)
        if (mode == 0) WINO_PLAIN(0);
        else if (mode == 1) WINO_PLAIN(1);
        else if (mode == 2) WINO_PLAIN(2);
        else WINO_PLAIN(3);
#undef WINO_PLAIN
    }
    conv_prof_end(pe, st);
    DC_CHECK_LAUNCH();
    if (ksplit > 1) {
        const size_t n4 = nout / 4;
        hipLaunchKernelGGL(wino_ysum_kernel, dim3((unsigned)std::min<size_t>((n4 + 255) / 256, 2048)), dim3(256), 0, st, slabs,
                           d.out, n4, n4, ksplit, d.fused ? d.bias : (const float*)nullptr, d.fused ? d.act : (int)ACT_NONE,
                           Ho * Wo / 4, M, d.fused ? (const float*)nullptr : d.addend);
        DC_CHECK_LAUNCH();
    }
    return DC_OK;
}

size_t wino_conv_ws_bytes(int B, int Ci, int Co, int H, int W) {
    // transformed weights + the reduction-split slabs of the larger of {forward output, full-correlation data gradient}
    const size_t fwd = (size_t)B * Co * H * W, full = (size_t)B * Ci * (H + 2) * (W + 2);
    return wino_uhat_bytes(Ci, Co) + wino_al256(std::max(wino_slab_bytes(fwd), wino_slab_bytes(full)));
}

bool wino_conv_eligible(int C0, int C1, int H, int W) {
    return W >= 2 && !(W & 1) && H >= 2 && (C1 == 0 || C0 % PSK == 0);
}

int wino_conv_fused_fwd(const float* x0, int C0, int up0, const float* x1, int C1, const float* weight, const float* bias, float* y,
                        void* ws, int B, int Co, int H, int W, int act, int pad, hipStream_t st) {
    WinoLaunch d{};
    d.src0 = x0; d.C0 = C0; d.up0 = up0; d.src1 = x1; d.C1 = C1; d.weight = weight; d.Co = Co; d.Ci = C0 + C1; d.dgrad = false;
    d.bias = bias; d.act = act; d.pad = pad; d.P = 1; d.out = y; d.ws = ws; d.B = B; d.H = H; d.W = W; d.M = Co; d.fused = true;
    return wino_launch(d, st);
}

// dxpad (B,Ci,H+2,W+2) = full correlation of gp (B,Co,H,W) with the rotated, transposed weights
int wino_conv_full_dgrad(const float* gp, const float* weight, float* dxpad, void* ws, int B, int Ci, int Co, int H, int W,
                         hipStream_t st) {
    WinoLaunch d{};
    d.src0 = gp; d.C0 = Co; d.up0 = 0; d.src1 = nullptr; d.C1 = 0; d.weight = weight; d.Co = Co; d.Ci = Ci; d.dgrad = true;
    d.bias = nullptr; d.act = ACT_NONE; d.pad = PAD_ZERO; d.P = 2; d.out = dxpad; d.ws = ws; d.B = B; d.H = H; d.W = W; d.M = Ci;
    d.fused = true;
    return wino_launch(d, st);
}

// The data gradient of a fused block (conv3x3.hip) WITHOUT the padded-domain scratch: the interior of the full correlation is the
// zero-padded "same" correlation of g' with the rotated, transposed filter (P = 1), and its 2 x 2 output tiles coincide with the
// pixels of a nearest-x2 upsampled x0; the store epilogue splits the concat, sums the upsampled half's tiles and adds the addends.
// What ReflectionPad folds back from the padded ring (rows -1 / H, columns -1 / W) is added afterwards by conv_ring_kernel.
bool wino_dgrad_split_ok(int B, int C0, int C1, int up0, int Co, int H, int W) {
    const int Ci = C0 + C1;
    if (H < 4 || W < 4 || (H & 1) || (W & 1) || C0 <= 0) return false;
    if ((size_t)B * Co * H * W * 4 >= 0x7fffffffull) return false;
    return wino_plan(B, Co, Ci, H, W).ksplit == 1;
}
int wino_conv_dgrad_split(const float* gp, const float* weight, float* dx0, float* dx1, const float* add0, const float* add1, void* ws,
                          int B, int C0, int C1, int up0, int Co, int H, int W, hipStream_t st) {
    WinoLaunch d{};
    d.src0 = gp; d.C0 = Co; d.up0 = 0; d.src1 = nullptr; d.C1 = 0; d.weight = weight; d.Co = Co; d.Ci = C0 + C1; d.dgrad = true;
    d.bias = nullptr; d.act = ACT_NONE; d.pad = PAD_ZERO; d.P = 1; d.out = dx0; d.ws = ws; d.B = B; d.H = H; d.W = W; d.M = C0 + C1;
    d.fused = true;
    d.split = true; d.split_c0 = C0; d.split_up = up0 ? 1 : 0; d.out1 = dx1; d.add0 = add0; d.add1 = add1;
    return wino_launch(d, st);
}

static int g_wino_f4 = 0;      // dc_set_wino_f4: 0 = F(2x2,3x3) everywhere (default), 1 = F(4x4,3x3) for the plain trunk convolutions it covers

static bool wino_bn_active(const dc_bn_fold* bn) { return bn && (bn->in_scale || bn->stat_part || bn->bwd_part); }
// the plain F(2x2,3x3) fp32 launch is the one that carries the fold
static bool wino_bn_path(int B, int Ci, int Co, int H, int W, bool dgrad) {
    if (B <= 0 || Ci <= 0 || Co <= 0 || H < 1 || W < 2 || (W & 1)) return false;
    if (matrix_precision() == DC_PREC_BF16 && c3b_eligible(dgrad ? Co : Ci, 0, 0, H, W, 1)) return false;
    return true;
}

static int wino_run(const float* x, const float* w, float* y, const float* addend, void* ws, int B, int Ci, int Co, int H, int W,
                    bool dgrad, hipStream_t st, const dc_bn_fold* bn = nullptr) {
    if (!x || !w || !y || !ws || B <= 0 || Ci <= 0 || Co <= 0 || H < 1 || W < 2 || (W & 1)) return DC_EINVAL;
    if (wino_bn_active(bn)) {
        if (!wino_bn_path(B, Ci, Co, H, W, dgrad)) return DC_EINVAL;
        WinoLaunch d{};
        d.src0 = x; d.C0 = dgrad ? Co : Ci; d.weight = w; d.Co = Co; d.Ci = Ci; d.dgrad = dgrad; d.act = ACT_NONE; d.pad = PAD_ZERO;
        d.P = 1; d.out = y; d.ws = ws; d.B = B; d.H = H; d.W = W; d.M = dgrad ? Ci : Co; d.fused = false; d.addend = addend; d.bn = bn;
        return wino_launch(d, st);
    }
    // reduced-precision policy: direct implicit GEMM on the bf16 matrix cores (conv_bf16.hip; the data gradient of a
    // zero-padded convolution is the convolution with the rotated, transposed filter)
    if (matrix_precision() == DC_PREC_BF16 && c3b_eligible(dgrad ? Co : Ci, 0, 0, H, W, 1)) {
        return c3b_conv(x, dgrad ? Co : Ci, 0, nullptr, 0, w, Co, Ci, dgrad ? 1 : 0, 0, nullptr, y, ws, B, H, W, ACT_NONE, PAD_ZERO, 1, st, addend);
    }
    // F(4x4,3x3) (wino4.hip): opt-in (dc_set_wino_f4) for maps its tile groups cover well -- measured against this file's
    // F(2x2,3x3) in tests/test_wino_gpu.py and tools/bench_wino.py; DESIGN 4a has the verdict
    if (g_wino_f4 && wino4_eligible(B, dgrad ? Co : Ci, dgrad ? Ci : Co, H, W) && wino4_utilisation(H, W) >= 0.85) {
        float* slabs = (float*)((char*)ws + std::max(wino_uhat_bytes(Ci, Co), wino4_uhat_bytes(Ci, Co)));
        return wino4_launch(x, w, nullptr, y, addend, ws, slabs, B, Ci, Co, H, W, dgrad, st);
    }
    WinoLaunch d{};
    d.src0 = x; d.C0 = dgrad ? Co : Ci; d.weight = w; d.Co = Co; d.Ci = Ci; d.dgrad = dgrad; d.act = ACT_NONE; d.pad = PAD_ZERO;
    d.P = 1; d.out = y; d.ws = ws; d.B = B; d.H = H; d.W = W; d.M = dgrad ? Ci : Co; d.fused = false; d.addend = addend;
    return wino_launch(d, st);
}

}  // namespace dc

using namespace dc;

extern "C" int dc_conv_profile_enable(int max_launches, int every) {
    g_cprof_every = every > 0 ? every : 1;
    for (auto& v : g_cprof_seen) v = 0;
    for (auto& d : g_cprof) {
        for (auto e : d.e0) (void)hipEventDestroy(e);
        for (auto e : d.e1) (void)hipEventDestroy(e);
        d.e0.clear(); d.e1.clear(); d.used = 0; d.flops = d.exec = d.bytes = 0.0;
    }
    g_cprof_cap = 0;
    if (max_launches <= 0) return DC_OK;
    for (auto& d : g_cprof) {
        d.e0.resize(max_launches); d.e1.resize(max_launches);
        for (int i = 0; i < max_launches; ++i)
            if (hipEventCreate(&d.e0[i]) != hipSuccess || hipEventCreate(&d.e1[i]) != hipSuccess) return DC_ELAUNCH;
    }
    g_cprof_cap = max_launches;
    return DC_OK;
}

extern "C" int dc_conv_profile_collect(int kind, double* ms, double* algorithmic_flops, double* executed_flops,
                                       double* algorithmic_bytes, int* launches) {
    if (kind < 0 || kind >= NPROF) return DC_EINVAL;
    ConvProf& d = g_cprof[kind];
    double tot = 0.0;
    for (int i = 0; i < d.used; ++i) {
        float t = 0.f;
        if (hipEventSynchronize(d.e1[i]) != hipSuccess || hipEventElapsedTime(&t, d.e0[i], d.e1[i]) != hipSuccess) return DC_ELAUNCH;
        tot += t;
    }
    if (ms) *ms = tot;
    if (algorithmic_flops) *algorithmic_flops = d.flops;
    if (executed_flops) *executed_flops = d.exec;
    if (algorithmic_bytes) *algorithmic_bytes = d.bytes;
    if (launches) *launches = d.used;
    d.used = 0; d.flops = d.exec = d.bytes = 0.0;
    return DC_OK;
}

extern "C" int dc_wino_cache_new_owner(void) {
    std::lock_guard<std::mutex> lk(g_wc_mu);
    WcOwner o;
    o.id = g_wc_next_owner++;
    g_wc_owners.push_back(o);
    return o.id;
}

extern "C" int dc_wino_cache_register(int owner, const float* weight, int Ci, int Co) {
    if (!weight || Ci <= 0 || Co <= 0) return DC_EINVAL;
    std::lock_guard<std::mutex> lk(g_wc_mu);
    WcOwner* o = wc_owner(owner);
    if (!o) return DC_EINVAL;
    for (auto& e : g_wc)
        if (e.w == weight) return (e.Ci == Ci && e.Co == Co && e.owner == owner) ? DC_OK : DC_EINVAL;
    g_wc.push_back(WcEntry{weight, Ci, Co, owner, {}});
    o->dirty = true;
    return DC_OK;
}

// Forget an owner and every weight it registered.  Its tables and variant buffers are parked, not freed: a captured
// hipGraph of the owner may still name them.
extern "C" int dc_wino_cache_release_owner(int owner) {
    std::lock_guard<std::mutex> lk(g_wc_mu);
    for (size_t i = 0; i < g_wc.size();) {
        if (g_wc[i].owner == owner) {
            for (auto& v : g_wc[i].v)
                if (v.buf) g_wc_retired.push_back(v.buf);
            g_wc.erase(g_wc.begin() + i);
        } else {
            ++i;
        }
    }
    for (size_t i = 0; i < g_wc_owners.size(); ++i)
        if (g_wc_owners[i].id == owner) {
            if (g_wc_owners[i].table) g_wc_retired.push_back(g_wc_owners[i].table);
            if (g_wc_owners[i].b2d) g_wc_retired.push_back(g_wc_owners[i].b2d);
            g_wc_owners.erase(g_wc_owners.begin() + i);
            break;
        }
    return DC_OK;
}

extern "C" int dc_wino_cache_refresh(int owner, void* stream) {
    std::lock_guard<std::mutex> lk(g_wc_mu);
    hipStream_t st = (hipStream_t)stream;
    WcOwner* o = wc_owner(owner);
    if (!o) return DC_EINVAL;
    // The descriptor upload allocates, synchronises and copies: none of it is legal on a capturing stream.  A capture that
    // meets a dirty registry replays the owner's table as it stands (variants outside it keep transforming per launch);
    // every weight the table names belongs to this owner and lives as long as it does.
    if (o->dirty && !wc_capturing(st)) {
        std::vector<WinoWDesc> host;
        std::vector<int> b2d;
        int blocks = 0;
        for (auto& e : g_wc) {
            if (e.owner != owner) continue;
            for (auto& v : e.v) {
                const int nb = wc_variant_blocks(v.kind, v.MT, v.Mp, v.Kp);
                b2d.insert(b2d.end(), nb, (int)host.size());
                host.push_back(WinoWDesc{e.w, v.buf, e.Co, e.Ci, v.MT, v.Mp, v.Kp, v.dgrad, blocks, v.kind});
                blocks += nb;
            }
        }
        // A rebuilt table goes to fresh memory and the old one is retired, not freed or rewritten: a captured graph holds
        // the old address and block count and must keep seeing the old contents.
        if (o->b2d) g_wc_retired.push_back(o->b2d);
        if (o->table) g_wc_retired.push_back(o->table);
        o->b2d = nullptr; o->table = nullptr; o->table_n = o->blocks = 0;
        if (blocks > 0 && hipMalloc((void**)&o->b2d, sizeof(int) * blocks) != hipSuccess) { o->b2d = nullptr; return DC_ELAUNCH; }
        if (!host.empty() && hipMalloc((void**)&o->table, sizeof(WinoWDesc) * host.size()) != hipSuccess) { o->table = nullptr; return DC_ELAUNCH; }
        // synchronous upload (the descriptor list only changes while the variants of a model are still being met)
        if (!host.empty() && hipStreamSynchronize(st) != hipSuccess) return DC_ELAUNCH;
        if (!host.empty() && hipMemcpy(o->table, host.data(), sizeof(WinoWDesc) * host.size(), hipMemcpyHostToDevice) != hipSuccess) return DC_ELAUNCH;
        if (!b2d.empty() && hipMemcpy(o->b2d, b2d.data(), sizeof(int) * b2d.size(), hipMemcpyHostToDevice) != hipSuccess) return DC_ELAUNCH;
        o->table_n = (int)host.size(); o->blocks = blocks; o->dirty = false;
        for (auto& e : g_wc)
            if (e.owner == owner)
                for (auto& v : e.v) v.in_table = true;
    }
    if (o->table_n > 0 && o->blocks > 0) {
        hipLaunchKernelGGL(wino_weights_batched_kernel, dim3(o->blocks), dim3(256), 0, st, (const WinoWDesc*)o->table, (const int*)o->b2d, PSK);
        DC_CHECK_LAUNCH();
    }
    for (auto& e : g_wc)
        if (e.owner == owner)
            for (auto& v : e.v) v.fresh = v.in_table;
    o->valid = true;
    return DC_OK;
}

extern "C" int dc_wino_cache_invalidate(int owner) {
    std::lock_guard<std::mutex> lk(g_wc_mu);
    WcOwner* o = wc_owner(owner);
    if (!o) return DC_EINVAL;
    o->valid = false;
    return DC_OK;
}

extern "C" int dc_wino_cache_clear(void) {
    std::lock_guard<std::mutex> lk(g_wc_mu);
    int rc = DC_OK;
    if (hipDeviceSynchronize() != hipSuccess) rc = DC_ELAUNCH;
    for (auto& e : g_wc)
        for (auto& v : e.v)
            if (v.buf && hipFree(v.buf) != hipSuccess) rc = DC_ELAUNCH;
    g_wc.clear();
    for (void* q : g_wc_retired)
        if (hipFree(q) != hipSuccess) rc = DC_ELAUNCH;
    g_wc_retired.clear();
    for (auto& o : g_wc_owners) {
        if (o.table && hipFree(o.table) != hipSuccess) rc = DC_ELAUNCH;
        if (o.b2d && hipFree(o.b2d) != hipSuccess) rc = DC_ELAUNCH;
    }
    g_wc_owners.clear();
    return rc;
}

extern "C" int dc_wino_cache_variants(void) {
    std::lock_guard<std::mutex> lk(g_wc_mu);
    int n = 0;
    for (auto& e : g_wc) n += (int)e.v.size();
    return n;
}

extern "C" size_t dc_wino3x3_workspace(int B, int Ci, int Co, int H, int W) {
    if (B <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0) return 0;
    return std::max({wino_uhat_bytes(Ci, Co), wino4_uhat_bytes(Ci, Co), c3b_weights_bytes(Ci, Co)}) +
           wino_al256(std::max(wino_slab_bytes((size_t)B * Ci * H * W), wino_slab_bytes((size_t)B * Co * H * W)));
}

extern "C" int dc_set_wino_persist(int mode) {
    if (mode != 0 && mode != 1) return DC_EINVAL;
    const int prev = g_wino_persist;
    g_wino_persist = mode;
    return prev;
}

extern "C" int dc_set_wino_f4(int mode) {
    if (mode != 0 && mode != 1) return DC_EINVAL;
    const int prev = g_wino_f4;
    g_wino_f4 = mode;
    return prev;
}

extern "C" int dc_wino3x3_fwd(const float* x, const float* weight, float* y, void* ws, int B, int Ci, int Co, int H, int W,
                              void* stream) {
    return wino_run(x, weight, y, nullptr, ws, B, Ci, Co, H, W, false, (hipStream_t)stream);
}

extern "C" int dc_wino3x3_dgrad(const float* gy, const float* weight, float* gx, void* ws, int B, int Ci, int Co, int H,
                                int W, void* stream) {
    return wino_run(gy, weight, gx, nullptr, ws, B, Ci, Co, H, W, true, (hipStream_t)stream);
}

extern "C" int dc_wino3x3_dgrad_add(const float* gy, const float* weight, float* gx, const float* addend, void* ws, int B, int Ci,
                                    int Co, int H, int W, void* stream) {
    if (addend && (((size_t)addend | (size_t)gx) & 15)) return DC_EINVAL;
    return wino_run(gy, weight, gx, addend, ws, B, Ci, Co, H, W, true, (hipStream_t)stream);
}

// ---- the same launches with a BatchNorm folded in (dc_bn_fold) ----------------------------------------------------------------
// partials per channel of the statistics epilogue (forward, rows = Co) / the BatchNorm-backward epilogue (data gradient, rows =
// Ci): two per sub-region (its 16-tile halves), sub-regions image after image; 0 = this launch has no epilogue (a split
// reduction, more than two groups, the bf16 policy)
static int wino_parts(int B, int K, int M, int H, int W, int groups, bool dgrad, int Ci, int Co, int* ppg) {
    if (groups < 1 || groups > 2 || B % groups || !wino_bn_path(B, Ci, Co, H, W, dgrad)) return 0;
    const WinoPlan p = wino_plan(B, K, M, H, W);
    if (p.ksplit > 1) return 0;
    if (ppg) *ppg = 2 * p.regs_x * p.regs_y * (B / groups);
    return 2 * ceil_div(p.nsub, p.G) * p.G;
}
extern "C" int dc_wino3x3_stat_parts(int B, int Ci, int Co, int H, int W, int groups, int* ppg) {
    return wino_parts(B, Ci, Co, H, W, groups, false, Ci, Co, ppg);
}
extern "C" int dc_wino3x3_bwd_parts(int B, int Ci, int Co, int H, int W, int groups, int* ppg) {
    return wino_parts(B, Co, Ci, H, W, groups, true, Ci, Co, ppg);
}
extern "C" int dc_wino3x3_bn_ok(int B, int Ci, int Co, int H, int W, int groups) {
    return groups >= 1 && groups <= 2 && B % groups == 0 && Ci % PSK == 0 && wino_bn_path(B, Ci, Co, H, W, false) && wino_bn_path(B, Ci, Co, H, W, true) &&
           dc_wino3x3_bwd_parts(B, Ci, Co, H, W, groups, nullptr) > 0;
}
extern "C" int dc_wino3x3_fwd_bn(const float* x, const float* weight, float* y, void* ws, int B, int Ci, int Co, int H, int W,
                                 const dc_bn_fold* bn, void* stream) {
    return wino_run(x, weight, y, nullptr, ws, B, Ci, Co, H, W, false, (hipStream_t)stream, bn);
}
extern "C" int dc_wino3x3_dgrad_bn(const float* gy, const float* weight, float* gx, const float* addend, void* ws, int B, int Ci,
                                   int Co, int H, int W, const dc_bn_fold* bn, void* stream) {
    if (addend && (((size_t)addend | (size_t)gx) & 15)) return DC_EINVAL;
    return wino_run(gy, weight, gx, addend, ws, B, Ci, Co, H, W, true, (hipStream_t)stream, bn);
}
