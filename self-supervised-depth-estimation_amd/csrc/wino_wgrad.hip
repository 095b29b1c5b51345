// Weight gradient of the stride-1 3x3 convolution in the Winograd F(2x2,3x3) domain, fp32 matrix cores (gfx950).
//
// Forward (wino.hip): Y_tile = A^T [ sum_k U .* V ] A with U = G g G^T, V = B^T d B.  Hence
//     dU_p[m][k] = sum_tiles dM_p[m][tile] * V_p[k][tile],   dM = A dY A^T (4x4 from the 2x2 tile of gy),
//     dg[m][k]   = G^T dU G                                   (4x4 -> 3x3),
// i.e. 16 GEMMs with the reduction over ALL 2x2 tiles of the batch -- 2.25x fewer multiplies than the direct
// correlation of x with gy.  Replaces the library weight-gradient call behind the trunk's nn.Conv2d
// (reference networks/resnet_encoder.py:74-98; loss.backward() at trainer.py:236).
//
// Mapping to CDNA4 (wino_wgrad_kernel)
//   * v_mfma_f32_16x16x4_f32: rows = 16 output channels m, cols = 16 input channels k, K = 4 tiles.
//     A operand lane (m = lane&15, tile = lane>>4) holds dM_p[m][tile], B operand lane (tile, k) holds V_p[k][tile].
//   * a block = 4 waves = the 4 ROWS of the Winograd domain (as in wino_ps_kernel): row a of dM needs the tile's
//     two gy rows (2 fma + 2 add), row a of V two raw patch rows (4 fma + 4 add).  The signs of row/column 3 of
//     dM are applied once to the accumulators at the end.
//   * a wave keeps 4 x MR x KR accumulator tiles (MR,KR = 4,2: 64 m x 32 k per block, 128 VGPRs).
//   * the reduction runs over sub-regions of <= 16 tiles (2x8, 4x4 or 3x5, whichever divides the map); per chunk
//     the block stages 64 channels of gy and 32 channels of x (with halo) into double-buffered LDS slabs; the next
//     chunk's 16 buffer_load_b64 per thread are in flight during the 128 MFMAs per wave of the current chunk.
//     Channel strides are = 2 (mod 32) floats so that the 16 channels of a 16-lane group hit distinct banks.
//   * split reduction without atomics: block (split, m-block, k-block) writes q[a][.] = (dU G)[a][.] (3 values per
//     row a) to its slab with lane-contiguous stores; wino_wreduce_kernel sums the slabs in fixed order and applies
//     G^T.  Deterministic.
// Requires even W; H arbitrary.
#include "dc_common.h"
#include "conv_bf16.h"
#include "wino.h"

#include <algorithm>

namespace dc {

using f4 = __attribute__((ext_vector_type(4))) float;
using f2w = __attribute__((ext_vector_type(2))) float;
using wrsrc_t = __amdgpu_buffer_rsrc_t;

constexpr int WG_MR = 4, WG_KR = 2;                    // 16-channel tiles per wave: gy side (WG_MR; 2 for the thin 32-channel layers), x side
constexpr int WG_KT = 16 * WG_KR;                      // x-side channels per block (gy side: 16 MR)
constexpr int WG_GPS = 66;                             // gy slab channel stride (floats), = 2 mod 32, >= 64
constexpr int WG_XPS = 130;                            // x slab channel stride, = 2 mod 32, >= 120
constexpr int WG_SLAB = 4 * WG_MR * WG_KR * 4 * 3 * 64;   // floats one block writes (MR = 4; half of it for MR = 2)

struct WinoWgArgs {
    const float* x; const float* gy; float* slab;
    int B, K, M, H, W;
    // FUSED only: x = cat(up2?(x), x1) along channels (C0 + C1 = K) under ReflectionPad2d(1) or zero padding
    const float* x1;
    int C0, up0, pad;
    unsigned x1bytes;
    int RH, RW, GRS, XRS;             // sub-region shape in tiles; LDS row strides of the gy and x slabs
    int regs_x, regs_y, nsub;
    int splits, kblocks, mblocks;
    unsigned xbytes, gbytes;
    unsigned mg_nmk, mg_kblocks, mg_RW, mg_XPR, mg_per_img, mg_regs_x;      // fdiv magics (dc_common.h)
    unsigned long long* diag;         // -DWINO_DIAG builds: per block {end, hw id, xcc, -, loop, prologue, epilogue, start} (tools/diag_wino.py)
    // BNIN (plain launches): x is the RAW input of a BatchNorm + ReLU folded into the forward's loader; the same
    // relu(scale[g, k] x + shift[g, k]) is re-formed between the global load and the LDS store (dc_bn_fold)
    const float* in_scale; const float* in_shift;
    int npg;
};
unsigned long long* wino_diag_ptr();

// NG = 2: a block is TWO such 4-wave groups (512 threads, one block per CU instead of two).  They take alternate sub-regions
// of the block's split with their own LDS slabs, and at the end group 1 hands its q values to group 0 through LDS (the
// staging slabs are dead by then): one slab write per CU instead of two -- the slabs were 50 MB written and 50 MB read
// back per launch whatever the layer (512 blocks x 96 KB), ~13 % of the kernel + reduce time -- at the same 8 waves per CU.
// BNIN: 0 plain input, 1 BatchNorm + ReLU folded into the loader with one BatchNorm group, 2 with two groups
template <bool FUSED, int MR, int NG, int BNIN = 0>
__global__ __launch_bounds__(256 * NG, 2) void wino_wgrad_kernel(WinoWgArgs a) {
    static_assert(!(FUSED && BNIN), "the BatchNorm fold exists for the plain trunk launches");
    constexpr int KR = WG_KR, MT = 16 * MR, NGQ = MT / 8;       // NGQ: gy channels staged per thread
    constexpr int GL = MT * WG_GPS, XL = WG_KT * WG_XPS, GROUP_LDS = 2 * GL + 2 * XL;
    static_assert(NG == 1 || NG * GROUP_LDS >= 4 * MR * KR * 4 * 3 * 64, "the q exchange aliases the staging slabs");
    extern __shared__ float wg_smem[];
#ifdef WINO_DIAG
    const unsigned long long dg_start = __builtin_amdgcn_s_memtime(), dg_rstart = __builtin_amdgcn_s_memrealtime();
#endif
    const int lane = threadIdx.x & 63, wave_all = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = NG == 1 ? 0 : wave_all >> 2, wave = wave_all & 3, tid = threadIdx.x & 255;     // group, Winograd row, thread of the group
    float* const gl0 = wg_smem + grp * GROUP_LDS;
    float* const xl0 = gl0 + 2 * GL;
    auto gl = [&](int buf) { return gl0 + buf * GL; };
    auto xl = [&](int buf) { return xl0 + buf * XL; };
    const int cl = lane & 15, tq = lane >> 4;          // channel within a 16-block, tile within a k-step
    const int H = a.H, W = a.W, RH = a.RH, RW = a.RW, GRS = a.GRS, XRS = a.XRS;
    // flat grid, XCD-aware: the nmk = mblocks * kblocks blocks of one split (same sub-regions: gy shared across the k-blocks,
    // x across the m-blocks) get adjacent logical indices, i.e. one XCD and its L2
    const int nmk = a.mblocks * a.kblocks;
    const int lbid = xcd_logical_block(blockIdx.x, gridDim.x);
    const int split0 = fdiv(lbid, a.mg_nmk), mk = lbid - split0 * nmk;
    const int mb = fdiv(mk, a.mg_kblocks), kb = mk - mb * a.kblocks;
    const int per_img = a.regs_x * a.regs_y;
    const unsigned plane = (unsigned)(H * W) * 4u;
    const wrsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.gy), (short)0, (int)a.gbytes, 0x00020000);
    const wrsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), (short)0, (int)a.xbytes, 0x00020000);
    const wrsrc_t x1r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(FUSED ? a.x1 : a.x), (short)0,
                                                          (int)(FUSED ? a.x1bytes : a.xbytes), 0x00020000);
    const int C0 = FUSED ? a.C0 : a.K, up0 = FUSED ? a.up0 : 0;
    const unsigned plane0 = up0 ? (unsigned)((H >> 1) * (W >> 1)) * 4u : plane;

    // ---- staging roles
    // gy: 32 pair slots per channel (rows 2RH x column pairs RW), thread -> (slot, channels cg + 8q)
    const int gslot = tid & 31, gcg = tid >> 5;
    const int grow = fdiv(gslot, a.mg_RW), gcp = gslot - grow * RW;
    const bool g_in = gslot < 2 * RH * RW;
    const int glds = gcg * WG_GPS + grow * GRS + 2 * gcp;
    // x: 64 pair slots per channel (rows 2RH+2 x column pairs RW+2), thread -> (slot, channels cg + 4q)
    const int xslot = tid & 63, xcg = tid >> 6;
    const int XPR = RW + 2;
    const int xrow = fdiv(xslot, a.mg_XPR), xcp = xslot - xrow * XPR;
    const bool x_in = xslot < (2 * RH + 2) * XPR;
    const int xlds0 = xcg * WG_XPS + xrow * XRS + max(2 * xcp - 1, 0), xlds1 = xcg * WG_XPS + xrow * XRS + 2 * xcp;

    f2w pg[NGQ], px[8];
    bool px_ok = false;          // BNIN: the pair requested last lies inside the image (zero padding stays zero)
    int px_grp = 0;              //       and the BatchNorm group of its image (wave-uniform)
    // BNIN: this wave stages the same 8 channels for the whole kernel: their scale / shift for both groups, read once
    float bsc[BNIN == 2 ? 2 : 1][BNIN ? 8 : 1], bsh[BNIN == 2 ? 2 : 1][BNIN ? 8 : 1];
    if constexpr (BNIN != 0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int chc = min(kb * WG_KT + wave + 4 * q, a.K - 1);
            bsc[0][q] = a.in_scale[chc]; bsh[0][q] = a.in_shift[chc];
            if constexpr (BNIN == 2) { bsc[1][q] = a.in_scale[a.K + chc]; bsh[1][q] = a.in_shift[a.K + chc]; }
        }
    }
    auto prefetch = [&](int sub) {
        const int b = fdiv(sub, a.mg_per_img), rq = sub - b * per_img;
        if constexpr (BNIN == 2) px_grp = __builtin_amdgcn_readfirstlane(b / a.npg);
        const int ry = fdiv(rq, a.mg_regs_x), rx = rq - ry * a.regs_x;
        const int Y0 = ry * RH * 2, X0 = rx * RW * 2;
        {
            const int y = Y0 + grow, xx = X0 + 2 * gcp;
            const bool ok = g_in && y < H && xx < W;
            const unsigned vo = ok ? ((unsigned)(b * a.M + mb * MT + gcg) * plane + (unsigned)(y * W + xx) * 4u) : 0x80000000u;
#pragma unroll
            for (int q = 0; q < NGQ; ++q)
                pg[q] = __builtin_bit_cast(f2w, __builtin_amdgcn_raw_buffer_load_b64(gr, (int)vo, (int)((unsigned)(8 * q) * plane), 0));
        }
        {
            // same source mapping as the forward staging (wino.hip): reflection redirects the border rows / pairs,
            // nearest-2x upsampling reads x0[y>>1][x>>1] once and duplicates it
            const int y = Y0 - 1 + xrow, xx = X0 - 2 + 2 * xcp;
            int sy = y, sx = xx;
            if (FUSED && a.pad == PAD_REFLECT) {
                sy = y == -1 ? 1 : (y == H ? H - 2 : y);
                sx = xx == -2 ? 0 : (xx == W ? W - 2 : xx);
            }
            const bool ok = x_in && sy >= 0 && sy < H && sx >= 0 && sx < W;
            if constexpr (BNIN != 0) px_ok = ok;
            const unsigned pix0 = (up0 ? (unsigned)((sy >> 1) * (W >> 1) + (sx >> 1)) : (unsigned)(sy * W + sx)) * 4u;
            const unsigned pix1 = (unsigned)(sy * W + sx) * 4u;
            const int chb = kb * WG_KT + xcg;               // wave-uniform: xcg = wave
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                // ONE load shape for every channel: an 8-byte buffer load, source picked by scalar selects (ch is wave-uniform).
                // Per-source code paths -- and `px[q] = {v, v}`, a USE of the value, for the upsampled half -- made the compiler
                // wait for these loads at their issue / in front of every commit (see wino.hip load_x).  The upsampled half
                // reads 8 bytes at x0[y>>1][x>>1] and uses the first 4 (commit).
                const int ch = chb + 4 * q;
                const bool from1 = FUSED && ch >= C0;
                const wrsrc_t rs = from1 ? x1r : xr;
                const unsigned base = from1 ? (unsigned)(b * (a.K - C0) + (ch - C0)) * plane + pix1 : (unsigned)(b * C0 + ch) * plane0 + pix0;
                px[q] = __builtin_bit_cast(f2w, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(ok ? base : 0x80000000u), 0, 0));
            }
        }
    };
    auto commit = [&](int buf) {
        if (g_in) {
#pragma unroll
            for (int q = 0; q < NGQ; ++q) *reinterpret_cast<f2w*>(gl(buf) + glds + 8 * q * WG_GPS) = pg[q];
        }
        if (x_in) {
            if constexpr (BNIN != 0) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int ch = kb * WG_KT + wave + 4 * q;        // (xcg == wave)
                    float sc = bsc[0][q], sh = bsh[0][q];
                    if constexpr (BNIN == 2) { sc = px_grp ? bsc[1][q] : sc; sh = px_grp ? bsh[1][q] : sh; }
                    const bool ok = px_ok && ch < a.K;
                    const f2w v = px[q] * f2w{sc, sc} + f2w{sh, sh};
                    xl(buf)[xlds0 + 4 * q * WG_XPS] = ok ? fmaxf(v.x, 0.f) : 0.f;
                    xl(buf)[xlds1 + 4 * q * WG_XPS] = ok ? fmaxf(v.y, 0.f) : 0.f;
                }
            } else {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const bool dup = FUSED && up0 && (kb * WG_KT + xcg + 4 * q) < C0;      // nearest-x2 source: one value, two columns
                    xl(buf)[xlds0 + 4 * q * WG_XPS] = px[q].x;      // (pair 0: the discarded column lands on, and is overwritten by, .y)
                    xl(buf)[xlds1 + 4 * q * WG_XPS] = dup ? px[q].x : px[q].y;
                }
            }
        }
    };

    // ---- compute role: wave = Winograd row a
    // dM row a = (gy[ga] + gs * gy[gb]) expanded to [r0, r0 + r1, r0 - r1, r1]; V row a from patch rows ra, rb
    const int ga = wave == 3 ? 1 : 0, gb = 1 - ga;
    const float gs = wave == 1 ? 1.f : (wave == 2 ? -1.f : 0.f);
    const int ra = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
    const int rb = wave == 3 ? 3 : (wave == 2 ? 1 : 2);
    const float sgn = wave == 1 ? 1.f : -1.f;
    int goffA[4], goffB[4], xoffA[4], xoffB[4];
    unsigned tvalid = 0;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const int t = ks * 4 + tq;
        const bool v = t < RH * RW;
        const int tl = v ? t : 0;
        const int ty = fdiv(tl, a.mg_RW), tx = tl - ty * RW;
        tvalid |= v ? (1u << ks) : 0u;
        goffA[ks] = cl * WG_GPS + (2 * ty + ga) * GRS + 2 * tx;
        goffB[ks] = cl * WG_GPS + (2 * ty + gb) * GRS + 2 * tx;
        xoffA[ks] = cl * WG_XPS + (2 * ty + ra) * XRS + 2 * tx;
        xoffB[ks] = cl * WG_XPS + (2 * ty + rb) * XRS + 2 * tx;
    }

    f4 acc[MR][KR][4];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < KR; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[i][j][q] = f4{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int buf) {
        const float* gs_ = gl(buf);
        const float* xs_ = xl(buf);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            // the two B sets (x side) of this k-step
            float v[KR][4];
#pragma unroll
            for (int j = 0; j < KR; ++j) {
                const float* pa = xs_ + j * 16 * WG_XPS + xoffA[ks];
                const float* pb = xs_ + j * 16 * WG_XPS + xoffB[ks];
                const f2w a0 = *reinterpret_cast<const f2w*>(pa), a1 = *reinterpret_cast<const f2w*>(pa + 2);
                const f2w b0 = *reinterpret_cast<const f2w*>(pb), b1 = *reinterpret_cast<const f2w*>(pb + 2);
                const float t0 = fmaf(b0.x, sgn, a0.x), t1 = fmaf(b0.y, sgn, a0.y);
                const float t2 = fmaf(b1.x, sgn, a1.x), t3 = fmaf(b1.y, sgn, a1.y);
                v[j][0] = t0 - t2; v[j][1] = t1 + t2; v[j][2] = t2 - t1; v[j][3] = t1 - t3;
            }
            const bool tv = (tvalid >> ks) & 1u;
            f2w graw[2][2];
            graw[0][0] = *reinterpret_cast<const f2w*>(gs_ + goffA[ks]);
            graw[0][1] = *reinterpret_cast<const f2w*>(gs_ + goffB[ks]);
#pragma unroll
            for (int i = 0; i < MR; ++i) {
                if (i + 1 < MR) {
                    graw[(i + 1) & 1][0] = *reinterpret_cast<const f2w*>(gs_ + (i + 1) * 16 * WG_GPS + goffA[ks]);
                    graw[(i + 1) & 1][1] = *reinterpret_cast<const f2w*>(gs_ + (i + 1) * 16 * WG_GPS + goffB[ks]);
                }
                __builtin_amdgcn_sched_barrier(0);
                float r0 = fmaf(graw[i & 1][1].x, gs, graw[i & 1][0].x);
                float r1 = fmaf(graw[i & 1][1].y, gs, graw[i & 1][0].y);
                r0 = tv ? r0 : 0.f;
                r1 = tv ? r1 : 0.f;
                const float d1 = r0 + r1, d2 = r0 - r1;
#pragma unroll
                for (int j = 0; j < KR; ++j) {
                    acc[i][j][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(r0, v[j][0], acc[i][j][0], 0, 0, 0);
                    acc[i][j][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(d1, v[j][1], acc[i][j][1], 0, 0, 0);
                    acc[i][j][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(d2, v[j][2], acc[i][j][2], 0, 0, 0);
                    acc[i][j][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(r1, v[j][3], acc[i][j][3], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    // ---- reduction over this block's sub-regions: split, split + splits, ... (NG = 2: alternately to the two groups; the
    // barriers are the block's, so a group that runs out of work keeps pace without computing)
    const int stride = NG * a.splits;
    int sub = split0 + grp * a.splits;
    if (sub < a.nsub) {
        prefetch(sub);
        commit(0);
        if (sub + stride < a.nsub) prefetch(sub + stride);
    }
    __syncthreads();
#ifdef WINO_DIAG
    const unsigned long long dg0 = __builtin_amdgcn_s_memtime();
#endif
    for (int it = 0, lead = split0; lead < a.nsub; lead += stride, sub += stride, ++it) {     // `lead`: group 0's sub-region (block-uniform trip count)
        if (sub < a.nsub) {
            compute(it & 1);
            if (sub + stride < a.nsub) {
                commit((it + 1) & 1);
                if (sub + 2 * stride < a.nsub) prefetch(sub + 2 * stride);
            }
        }
        __syncthreads();
    }

#ifdef WINO_DIAG
    const unsigned long long dg1 = __builtin_amdgcn_s_memtime();
#endif
    // ---- q[a][.] = (sigma dU)[a][.] G and the slab write (lane-contiguous):
    // slab[((((blk*4 + a)*MR + i)*KR + j)*4 + r)*3 + qq][lane],  blk = split * nmk + mk
    const float sa = wave == 3 ? -1.f : 1.f;
    float* dst = a.slab + ((size_t)(split0 * nmk + mk) * 4 + wave) * (MR * KR * 4 * 3 * 64) + lane;
    float* ex = wg_smem + (size_t)wave * (MR * KR * 4 * 3 * 64) + lane;       // NG = 2: group 1 -> group 0, same layout as the slab
    if (NG == 2 && grp == 1) {
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int j = 0; j < KR; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float u0 = sa * acc[i][j][0][r], u1 = sa * acc[i][j][1][r], u2 = sa * acc[i][j][2][r];
                    const float u3 = -sa * acc[i][j][3][r];
                    const float h = 0.5f * (u1 + u2);
                    float* d = ex + (((i * KR + j) * 4 + r) * 3) * 64;
                    d[0] = u0 + h;
                    d[64] = 0.5f * (u1 - u2);
                    d[128] = h + u3;
                }
    }
    if (NG == 2) __syncthreads();
    if (grp == 0) {
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int j = 0; j < KR; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float u0 = sa * acc[i][j][0][r], u1 = sa * acc[i][j][1][r], u2 = sa * acc[i][j][2][r];
                    const float u3 = -sa * acc[i][j][3][r];
                    const float h = 0.5f * (u1 + u2);
                    const int e = (((i * KR + j) * 4 + r) * 3) * 64;
                    float q0 = u0 + h, q1 = 0.5f * (u1 - u2), q2 = h + u3;
                    if (NG == 2) { q0 += ex[e]; q1 += ex[e + 64]; q2 += ex[e + 128]; }     // fixed order: group 0 + group 1
                    dst[e] = q0;
                    dst[e + 64] = q1;
                    dst[e + 128] = q2;
                }
    }
#ifdef WINO_DIAG
    if (a.diag && lane == 0 && wave_all == 0) {
        __builtin_amdgcn_s_waitcnt(0x0f70);                       // vmcnt(0): the slab stores have left the wave
        unsigned long long* o = a.diag + (size_t)blockIdx.x * 8;
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        o[1] = hwid; o[2] = xcc; o[3] = 0; o[4] = dg1 - dg0; o[5] = dg0 - dg_start;
        o[6] = __builtin_amdgcn_s_memtime() - dg1; o[7] = dg_rstart;
        o[0] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

// dw[m][k][3x3] = G^T (sum over splits of q), fixed order.  One block per (m-block, k-block, i, j, r): 64 lanes x 16
// split groups; a group sums its contiguous range of splits, the 16 partial sums are then added in order through LDS.
constexpr int WR_GROUPS = 16;
__global__ __launch_bounds__(64 * WR_GROUPS) void wino_wreduce_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                                      int splits, int nmk, int kblocks, int M, int K, int MR) {
    constexpr int KR = WG_KR;
    const size_t WAVE_SLAB = (size_t)MR * KR * 4 * 3 * 64;           // floats one wave (row a) of one block wrote
    __shared__ float part[WR_GROUPS][12][64];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    int e = blockIdx.x;                                 // ((mk*MR + i)*KR + j)*4 + r
    const int r = e & 3; e >>= 2;
    const int j = e % KR; e /= KR;
    const int i = e % MR; e /= MR;
    const int mk = e;
    float q[4][3];
#pragma unroll
    for (int w = 0; w < 4; ++w)
#pragma unroll
        for (int c = 0; c < 3; ++c) q[w][c] = 0.f;
    const int groups = blockDim.x >> 6;                 // <= WR_GROUPS, chosen by the host from the split count
    const int per = (splits + groups - 1) / groups;
    const int s1 = min(splits, (grp + 1) * per);
    for (int s = grp * per; s < s1; ++s) {
        const float* src = slab + ((size_t)(s * nmk + mk) * 4) * WAVE_SLAB + (size_t)(((i * KR + j) * 4 + r) * 3) * 64 + lane;
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
            for (int c = 0; c < 3; ++c) q[w][c] += src[(size_t)w * WAVE_SLAB + c * 64];
    }
#pragma unroll
    for (int w = 0; w < 4; ++w)
#pragma unroll
        for (int c = 0; c < 3; ++c) part[grp][w * 3 + c][lane] = q[w][c];
    __syncthreads();
    if (grp != 0) return;
#pragma unroll
    for (int w = 0; w < 4; ++w)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float t = part[0][w * 3 + c][lane];
            for (int g = 1; g < groups; ++g) t += part[g][w * 3 + c][lane];
            q[w][c] = t;
        }
    const int mbk = mk / kblocks, kbk = mk - mbk * kblocks;
    const int m = mbk * (16 * MR) + i * 16 + (lane >> 4) * 4 + r, k = kbk * WG_KT + j * 16 + (lane & 15);
    if (m >= M || k >= K) return;
    float* out = dw + ((size_t)m * K + k) * 9;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float h = 0.5f * (q[1][c] + q[2][c]);
        out[0 + c] = q[0][c] + h;
        out[3 + c] = 0.5f * (q[1][c] - q[2][c]);
        out[6 + c] = h + q[3][c];
    }
}

// The same reduction with one block per (m-block, k-block, i, j, r, COLUMN c of the 3x3) -- three times the blocks, for layers
// whose tile grid is small (a 64 -> 64 layer: 64 blocks of the kernel above, a quarter of the chip reading 25 MB of slabs;
// wgrad + reduce 54.9 -> 50.6 us at B = 12, 23.2 -> 19.9 at B = 1; beyond 1,024 base blocks the extra blocks cost more than
// they bring: 512 channels +2 us).  Same per-output summation order as the kernel above: bitwise the same result.
// One block per (.., r, c):
// 64 lanes x 16 split groups; a group sums its contiguous range of splits, the 16 partial sums are then added in order
// through LDS.  (One block per (.., r) with all three columns was 64 blocks for a 64 -> 64 layer: a quarter of the chip
// reading 25 MB of slabs.)
__global__ __launch_bounds__(64 * WR_GROUPS) void wino_wreduce_col_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                                      int splits, int nmk, int kblocks, int M, int K, int MR) {
    constexpr int KR = WG_KR;
    const size_t WAVE_SLAB = (size_t)MR * KR * 4 * 3 * 64;           // floats one wave (row a) of one block wrote
    __shared__ float part[WR_GROUPS][4][64];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    int e = blockIdx.x;                                 // (((mk*MR + i)*KR + j)*4 + r)*3 + c
    const int c = e % 3; e /= 3;
    const int r = e & 3; e >>= 2;
    const int j = e % KR; e /= KR;
    const int i = e % MR; e /= MR;
    const int mk = e;
    float q[4] = {0.f, 0.f, 0.f, 0.f};
    const int groups = blockDim.x >> 6;                 // <= WR_GROUPS, chosen by the host from the split count
    const int per = (splits + groups - 1) / groups;
    const int s1 = min(splits, (grp + 1) * per);
    for (int s = grp * per; s < s1; ++s) {
        const float* src = slab + ((size_t)(s * nmk + mk) * 4) * WAVE_SLAB + (size_t)(((i * KR + j) * 4 + r) * 3 + c) * 64 + lane;
#pragma unroll
        for (int w = 0; w < 4; ++w) q[w] += src[(size_t)w * WAVE_SLAB];
    }
#pragma unroll
    for (int w = 0; w < 4; ++w) part[grp][w][lane] = q[w];
    __syncthreads();
    if (grp != 0) return;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        float t = part[0][w][lane];
        for (int g = 1; g < groups; ++g) t += part[g][w][lane];
        q[w] = t;
    }
    const int mbk = mk / kblocks, kbk = mk - mbk * kblocks;
    const int m = mbk * (16 * MR) + i * 16 + (lane >> 4) * 4 + r, k = kbk * WG_KT + j * 16 + (lane & 15);
    if (m >= M || k >= K) return;
    float* out = dw + ((size_t)m * K + k) * 9;
    const float h = 0.5f * (q[1] + q[2]);
    out[0 + c] = q[0] + h;
    out[3 + c] = 0.5f * (q[1] - q[2]);
    out[6 + c] = h + q[3];
}

static inline size_t wg_lds(int mr, int ng) { return (size_t)ng * (2 * 16 * mr * WG_GPS + 2 * WG_KT * WG_XPS) * sizeof(float); }
template <typename K>
static bool wg_set_lds(K kernel, size_t bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess;
}

struct WgPlan { int RH, RW, GRS, XRS, regs_x, regs_y, nsub, mblocks, kblocks, splits, mr, ng; };

static WgPlan wg_plan(int B, int Ci, int Co, int H, int W) {
    WgPlan p{};
    const int TH = ceil_div(H, 2), TW = W / 2;
    // RH, RW, GRS = 2 RW, XRS = 2 RW + 4 (the last column pair spills one column past the 2 RW + 2 that are read)
    const int cand[3][4] = {{2, 8, 16, 20}, {4, 4, 8, 12}, {3, 5, 10, 14}};
    double best = -1.0;
    for (auto& c : cand) {
        const double util = (double)TH * TW / ((double)ceil_div(TH, c[0]) * ceil_div(TW, c[1]) * 16.0);
        if (util > best + 1e-9) { best = util; p.RH = c[0]; p.RW = c[1]; p.GRS = c[2]; p.XRS = c[3]; }
    }
    p.regs_x = ceil_div(TW, p.RW); p.regs_y = ceil_div(TH, p.RH); p.nsub = p.regs_x * p.regs_y * B;
    // 64 output channels per block; 32 (MR = 2) where the last 64-block would be at most half used --
    // the 32-channel decoder levels ran on the direct kernel at 65 TFLOP/s before (96 -> 32 at 96 x 320: 311 us)
    p.mr = (Co % 64 != 0 && Co % 64 <= 32) ? 2 : 4;
    if (const char* f = getenv("DC_WGRAD_MR")) { const int v = atoi(f); if (v == 2 || v == 4) p.mr = v; }      // experiments
    p.mblocks = ceil_div(Co, 16 * p.mr); p.kblocks = ceil_div(Ci, WG_KT);
    const int nmk = p.mblocks * p.kblocks;
    // 256 CUs x 2 blocks; at least two chunks per block so that the pipeline has something to overlap
    // 256 CUs x 8 waves: 512 one-group blocks, or -- 64-channel tiles with at least two sub-regions for each of the 512
    // groups -- 256 two-group blocks (half the slab traffic); at least two chunks per group so that the pipeline has
    // something to overlap
    p.ng = (p.mr == 4 && p.nsub >= 4 * ceil_div(256, nmk)) ? 2 : 1;
    if (const char* f = getenv("DC_WGRAD_NG")) { const int v = atoi(f); if (v == 1 || (v == 2 && p.mr == 4)) p.ng = v; }      // experiments
    // (32-channel tiles on a big map -- 96 -> 32 at 96 x 320 -- : three blocks per CU, 768 in all, measured 119 us against 158 us at 512;
    // the 64-channel tiles and the small maps are best at 512 / 256: tools/sweep_wgrad.py, round 5)
    const int target = (p.mr == 2 && p.nsub >= 4096) ? 768 : 512 / p.ng;
    p.splits = std::max(1, std::min(std::max(1, p.nsub / (2 * p.ng)), ceil_div(target, nmk)));
    if (const char* f = getenv("DC_WGRAD_BLOCKS")) {         // experiments (tools/sweep_wgrad.py): target block count
        const int tb = atoi(f);
        if (tb > 0) p.splits = std::max(1, std::min(std::max(1, p.nsub / (2 * p.ng)), ceil_div(tb, nmk)));
    }
    return p;
}

static int wg_launch(const float* x0, int C0, int up0, const float* x1, int C1, int pad, bool fused, const float* gy, float* dweight,
                     void* ws, int B, int Co, int H, int W, hipStream_t st, const dc_bn_fold* bn = nullptr) {
    const int Ci = C0 + C1;
    const size_t b0 = (size_t)B * C0 * (H >> up0) * (W >> up0) * 4, b1 = (size_t)B * C1 * H * W * 4, gb = (size_t)B * Co * H * W * 4;
    if (b0 >= 0x7fffffffull || b1 >= 0x7fffffffull || gb >= 0x7fffffffull) return DC_EINVAL;      // 32-bit buffer offsets
    const WgPlan p = wg_plan(B, Ci, Co, H, W);
    WinoWgArgs a{};
    a.x = x0; a.x1 = x1; a.gy = gy; a.slab = (float*)ws; a.B = B; a.K = Ci; a.M = Co; a.H = H; a.W = W;
    a.C0 = C0; a.up0 = up0; a.pad = pad;
    a.RH = p.RH; a.RW = p.RW; a.GRS = p.GRS; a.XRS = p.XRS;
    a.regs_x = p.regs_x; a.regs_y = p.regs_y; a.nsub = p.nsub; a.splits = p.splits; a.kblocks = p.kblocks; a.mblocks = p.mblocks;
    a.xbytes = (unsigned)b0; a.x1bytes = (unsigned)b1; a.gbytes = (unsigned)gb;
    {
        const int nmk_ = p.mblocks * p.kblocks, per_img = p.regs_x * p.regs_y;
        a.mg_nmk = fdiv_magic(nmk_); a.mg_kblocks = fdiv_magic(p.kblocks); a.mg_RW = fdiv_magic(p.RW); a.mg_XPR = fdiv_magic(p.RW + 2);
        a.mg_per_img = fdiv_magic(per_img); a.mg_regs_x = fdiv_magic(p.regs_x);
        // exact while dividend * divisor < 2^32: block index by nmk, sub-region index by per_img
        if ((unsigned long long)p.splits * nmk_ * (unsigned)nmk_ >= 0xffffffffull || (unsigned long long)p.nsub * (unsigned)per_img >= 0xffffffffull)
            return DC_EINVAL;
    }
#ifdef WINO_DIAG
    a.diag = wino_diag_ptr();
#endif
    const bool bnin = bn && bn->in_scale;
    if (bnin) {
        if (fused || !bn->in_shift || bn->groups < 1 || bn->groups > 2 || B % bn->groups) return DC_EINVAL;
        a.in_scale = bn->in_scale; a.in_shift = bn->in_shift; a.npg = B / bn->groups;
    }
    const int nmk = p.mblocks * p.kblocks;
    hipEvent_t pe = conv_prof_begin(1, 2.0 * B * (double)Co * Ci * 9.0 * H * W,
                                    2.0 * 16.0 * (double)p.nsub * 16.0 * (double)(p.mblocks * 16 * p.mr) * (p.kblocks * WG_KT),
                                    (double)b0 + (double)b1 + (double)gb + 36.0 * Co * Ci, st);
    static const bool attr = wg_set_lds(wino_wgrad_kernel<true, 4, 1>, wg_lds(4, 1)) && wg_set_lds(wino_wgrad_kernel<false, 4, 1>, wg_lds(4, 1)) &&
                             wg_set_lds(wino_wgrad_kernel<true, 4, 2>, wg_lds(4, 2)) && wg_set_lds(wino_wgrad_kernel<false, 4, 2>, wg_lds(4, 2)) &&
                             wg_set_lds(wino_wgrad_kernel<true, 2, 1>, wg_lds(2, 1)) && wg_set_lds(wino_wgrad_kernel<false, 2, 1>, wg_lds(2, 1)) &&
                             wg_set_lds(wino_wgrad_kernel<false, 4, 1, 1>, wg_lds(4, 1)) && wg_set_lds(wino_wgrad_kernel<false, 4, 2, 1>, wg_lds(4, 2)) &&
                             wg_set_lds(wino_wgrad_kernel<false, 2, 1, 1>, wg_lds(2, 1)) &&
                             wg_set_lds(wino_wgrad_kernel<false, 4, 1, 2>, wg_lds(4, 1)) && wg_set_lds(wino_wgrad_kernel<false, 4, 2, 2>, wg_lds(4, 2)) &&
                             wg_set_lds(wino_wgrad_kernel<false, 2, 1, 2>, wg_lds(2, 1));
    if (!attr) return DC_ELAUNCH;
    const dim3 grid(p.splits * nmk);
    const size_t lds = wg_lds(p.mr, p.ng);
    if (bnin && bn->groups == 1) {
        if (p.mr == 4 && p.ng == 2) hipLaunchKernelGGL((wino_wgrad_kernel<false, 4, 2, 1>), grid, dim3(512), lds, st, a);
        else if (p.mr == 4) hipLaunchKernelGGL((wino_wgrad_kernel<false, 4, 1, 1>), grid, dim3(256), lds, st, a);
        else hipLaunchKernelGGL((wino_wgrad_kernel<false, 2, 1, 1>), grid, dim3(256), lds, st, a);
    } else if (bnin) {
        if (p.mr == 4 && p.ng == 2) hipLaunchKernelGGL((wino_wgrad_kernel<false, 4, 2, 2>), grid, dim3(512), lds, st, a);
        else if (p.mr == 4) hipLaunchKernelGGL((wino_wgrad_kernel<false, 4, 1, 2>), grid, dim3(256), lds, st, a);
        else hipLaunchKernelGGL((wino_wgrad_kernel<false, 2, 1, 2>), grid, dim3(256), lds, st, a);
    } else if (p.mr == 4 && p.ng == 2) {
        if (fused) hipLaunchKernelGGL((wino_wgrad_kernel<true, 4, 2>), grid, dim3(512), lds, st, a);
        else hipLaunchKernelGGL((wino_wgrad_kernel<false, 4, 2>), grid, dim3(512), lds, st, a);
    } else if (p.mr == 4) {
        if (fused) hipLaunchKernelGGL((wino_wgrad_kernel<true, 4, 1>), grid, dim3(256), lds, st, a);
        else hipLaunchKernelGGL((wino_wgrad_kernel<false, 4, 1>), grid, dim3(256), lds, st, a);
    } else {
        if (fused) hipLaunchKernelGGL((wino_wgrad_kernel<true, 2, 1>), grid, dim3(256), lds, st, a);
        else hipLaunchKernelGGL((wino_wgrad_kernel<false, 2, 1>), grid, dim3(256), lds, st, a);
    }
    conv_prof_end(pe, st);
    DC_CHECK_LAUNCH();
    const int rblocks = nmk * p.mr * WG_KR * 4;
    const dim3 rthreads(64 * std::min(WR_GROUPS, std::max(1, p.splits / 2)));
    static const int colmode = [] { const char* f = getenv("DC_WREDUCE_COL"); return f ? atoi(f) : -1; }();      // experiments: 0 / 1 force
    if (colmode < 0 ? rblocks <= 1024 : colmode == 1)
        hipLaunchKernelGGL(wino_wreduce_col_kernel, dim3(rblocks * 3), rthreads, 0, st, (const float*)ws, dweight, p.splits, nmk, p.kblocks, Co, Ci, p.mr);
    else
        hipLaunchKernelGGL(wino_wreduce_kernel, dim3(rblocks), rthreads, 0, st, (const float*)ws, dweight, p.splits, nmk, p.kblocks, Co, Ci, p.mr);
    DC_CHECK_LAUNCH();
    return DC_OK;
}

size_t wino_wgrad_ws_bytes(int B, int Ci, int Co, int H, int W) {
    const WgPlan p = wg_plan(B, Ci, Co, H, W);
    return (size_t)p.splits * p.mblocks * p.kblocks * (WG_SLAB / WG_MR * p.mr) * sizeof(float);
}

int wino_wgrad_fused(const float* x0, int C0, int up0, const float* x1, int C1, int pad, const float* gp, float* dweight, void* ws,
                     int B, int Co, int H, int W, hipStream_t st) {
    return wg_launch(x0, C0, up0, x1, C1, pad, true, gp, dweight, ws, B, Co, H, W, st);
}

}  // namespace dc

using namespace dc;

extern "C" size_t dc_wino3x3_wgrad_workspace(int B, int Ci, int Co, int H, int W) {
    if (B <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W < 2 || (W & 1)) return 0;
    const size_t b16 = (size_t)c3b_wgrad_split(B, H, W, Co, Ci, 1) * Co * Ci * 9 * sizeof(float);
    return std::max(wino_wgrad_ws_bytes(B, Ci, Co, H, W), b16);
}

extern "C" int dc_wino3x3_wgrad(const float* x, const float* gy, float* dweight, void* ws, int B, int Ci, int Co, int H, int W,
                                void* stream) {
    if (!x || !gy || !dweight || !ws || B <= 0 || Ci <= 0 || Co <= 0 || H < 1 || W < 2 || (W & 1)) return DC_EINVAL;
    if (matrix_precision() == DC_PREC_BF16 && c3b_eligible(Ci, 0, 0, H, W, 1)) {      // bf16 matrix cores (conv_bf16.hip)
        const int sp = c3b_wgrad_split(B, H, W, Co, Ci, 1);
        const int rc = c3b_wgrad(x, Ci, 0, nullptr, 0, gy, (float*)ws, sp, B, Co, H, W, PAD_ZERO, 1, (hipStream_t)stream);
        return rc != DC_OK ? rc : conv_wreduce((const float*)ws, nullptr, dweight, nullptr, sp, Co * Ci * 9, 0, (hipStream_t)stream);
    }
    return wg_launch(x, Ci, 0, nullptr, 0, PAD_ZERO, false, gy, dweight, ws, B, Co, H, W, (hipStream_t)stream);
}

/* the weight gradient with the input's BatchNorm + ReLU re-formed in the loader (dc_bn_fold.in_scale / in_shift; fp32 policy) */
extern "C" int dc_wino3x3_wgrad_bn(const float* x, const float* gy, float* dweight, void* ws, int B, int Ci, int Co, int H, int W,
                                   const dc_bn_fold* bn, void* stream) {
    if (!bn || !bn->in_scale) return dc_wino3x3_wgrad(x, gy, dweight, ws, B, Ci, Co, H, W, stream);
    if (!x || !gy || !dweight || !ws || B <= 0 || Ci <= 0 || Co <= 0 || H < 1 || W < 2 || (W & 1)) return DC_EINVAL;
    if (matrix_precision() == DC_PREC_BF16 && c3b_eligible(Ci, 0, 0, H, W, 1)) return DC_EINVAL;
    return wg_launch(x, Ci, 0, nullptr, 0, PAD_ZERO, false, gy, dweight, ws, B, Co, H, W, (hipStream_t)stream, bn);
}
