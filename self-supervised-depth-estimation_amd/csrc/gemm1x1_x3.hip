// 1x1 convolutions as fp32-accurate GEMMs on the bf16 matrix cores ("split operands"): every fp32 operand is written as the sum
// of three bf16 pieces, a = a0 + a1 + a2 (round to nearest each: 24 significant bits in all, i.e. the fp32 value itself up to
// ~2^-25 relative), and the six partial products down to 2^-16 |a||b| -- a0b0, a0b1, a1b0, a0b2, a1b1, a2b0 -- are issued as
// v_mfma_f32_16x16x32_bf16 with fp32 accumulation.  bf16 x bf16 products are exact in fp32; what is dropped (a1b2 + a2b1 + a2b2)
// is <= 2^-23 of |a||b| per product with random sign: measured BELOW the rounding error of an fp32 multiply-add chain of the
// same length (tests/test_conv1x1_gpu.py: error against an fp64 GEMM, beside the fp32-MFMA kernels' own).  The bf16 instruction
// has 16x the rate of v_mfma_f32_16x16x4_f32, so six of them cost 6/16 of the fp32 matrix time for the same sum.
//
// Why here and not in the Winograd kernels: a GEMM reuses each operand element across a whole tile row /
// column, so the split (about 5.5 vector instructions per element, done once while staging into LDS) is amortised over 128 rows
// or columns and the weights' pieces are prepared once per launch; the Winograd-domain GEMM would have to split its
// transformed input in the inner loop (used by only 2-4 matrix instructions) and move 1.5x the transformed weights per block.
//
// reference: torchvision Bottleneck conv1 / conv3 / downsample behind networks/resnet_encoder.py:70-98 (resnet50+ trunks).
//
//   forward        y[b,m,p]  = act( sum_k w[m,k] x[b,k,S*py,S*px] + bias[m] )      A = w            B = x
//   data gradient  dx[b,k,p] = sum_m w[m,k] gy[b,m,p]  (+ addends)                  A = w^T          B = gy   (stride 1)
//
// One kernel for both: D[128 x 128] += A[128 x 32] B[32 x 128] per reduction chunk, 4 waves (2 x 2) of 64 x 64.
//   * A (weights): g1x3_prep_kernel writes the three bf16 pieces once per launch as [piece][row (padded to 128)][reduction], the
//     reduction index permuted inside every chunk of 32 so that MFMA k-group kg holds k = 4kg..4kg+3 and 16+4kg..16+4kg+3 (see B);
//     staged to LDS as [piece][row][32 + 8 pad] -- one ds_read_b128 per (16-row tile, piece).
//   * B (activations, pixels contiguous in memory): a thread loads 16 consecutive pixels of one reduction row (4 x 16 B), splits
//     them, and stores them pixel-permuted -- position 16 nt + i of a wave's 64 columns holds pixel 4 i + nt -- so that (1) the
//     four ds_write_b64 per piece are contiguous runs, (2) a lane's four accumulator tiles are four CONSECUTIVE pixels: float4
//     stores, 256 contiguous bytes per 16 lanes and output row.  The operand needs 8 reduction elements per lane for its pixel:
//     ds_read_b64_tr_b16 (MI355X guide T10) transposes 4 rows x 16 columns per 16-lane group; rows 4kg..4kg+3 and
//     16+4kg..16+4kg+3 for k-group kg make the eight rows a 32-lane half touches consecutive, and with 288-byte rows (72 dwords =
//     8 mod 64) its 64 dwords tile the 64 banks.
//   * single LDS image (58 KB: two blocks per CU), the next chunk's global loads in flight in registers during the MFMAs,
//     two barriers per chunk (the co-resident block computes meanwhile).
#include "dc_common.h"
#include "gemm1x1.h"
#include "gemm1x1_x3.h"
#include "wino.h"

#include <algorithm>
#include <stdio.h>
#include <stdlib.h>

namespace dc {

typedef __attribute__((ext_vector_type(8))) __bf16 x3bf8;
typedef __attribute__((ext_vector_type(2))) __bf16 x3bf2;
typedef __attribute__((ext_vector_type(4))) short x3s4;
typedef __attribute__((ext_vector_type(8))) short x3s8;
typedef __attribute__((ext_vector_type(4))) float x3f4;
typedef __attribute__((ext_vector_type(4))) unsigned x3u4;
typedef __attribute__((ext_vector_type(2))) unsigned x3u2;

constexpr int X3_KC = 32;
constexpr int X3_AST = 40;                 // bf16 per reduction-contiguous row in LDS (32 + 8): 80-byte rows spread the b128 reads over the banks
// tile = (32 MT) x (32 NT) per block, 4 waves (2 x 2) of (16 MT) x (16 NT); (MT, NT) in {(4,4), (2,4), (4,2), (2,2)}
template <int NT> struct X3B { static constexpr int ST = 32 * NT + 16; };     // bf16 per B row (pixels + 16): 72 / 40 dwords = 8 mod 16
template <int MT, int NT> struct X3T {
    static constexpr int BM = 32 * MT, BN = 32 * NT;
    static constexpr int APIECE = BM * X3_AST, BPIECE = X3_KC * X3B<NT>::ST;
    static constexpr size_t LDS = (size_t)(3 * APIECE + 3 * BPIECE) * 2;       // (4,4): 58,368 B; (2,2): 30,720 B
    static constexpr size_t LDSW = (size_t)(3 * APIECE + 3 * BN * X3_AST) * 2; // weight gradient: both operands reduction-contiguous
    static constexpr int PER_CU = MT * NT >= 16 ? 2 : (MT * NT >= 8 ? 3 : 4);
};

struct G1x3Args {
    const unsigned short* wa;   // split weights [3][Mp][K] (g1x3_prep_kernel)
    const float* x;             // B operand source (B, K, Hi, Wi): x (forward) or gy (data gradient)
    float* out;                 // (B, M, Ho, Wo)
    const float* bias;          // (M) or null
    const float* addend;        // same shape as out or null (the other gradient of a residual fork)
    const float* addend2;
    int act;
    int B, M, Mp, K, Hi, Wi, Ho, Wo;
    int mtiles, ntiles;
    // ---- BatchNorm folded into the launch (include/depthcore.h: dc_bn_fold): same contract as csrc/gemm1x1.hip
    const float* in_scale;      // MODE 1 (forward): the B operand is relu(scale[g, k] x + shift[g, k]), applied between load and split;
    const float* in_shift;      // MODE 2 (data gradient): the same pair re-derives the ReLU decision of the epilogue
    int npg;                    // images per BatchNorm group
    float* stat_part;           // forward (MODE 0 / 1), nullable: (M, stat_nparts) x {sum, sum of squares} of the output per wave column
    int stat_nparts;
    const float* bn_x;          // data-gradient epilogues (MODE 2 / 3): raw input of the BatchNorm whose ReLU-ed output the forward read,
    const float* bn_mean;       // its mean (groups, M), its ReLU bit mask (MODE 3) and the partials (M, bwd_nparts) x
    const unsigned long long* bn_mask;   // {sum g', sum g' (x - mean)} of the masked result g'
    float* bwd_part;
    int bwd_nparts;
};

// sum over the 16 lanes of a DPP row (lanes sharing lane >> 4), result in every lane of the row
__device__ __forceinline__ float x3_row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
    return v;
}

__device__ __forceinline__ void x3_split2(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) { x3h_split2(a, b, p0, p1, p2); }

// ---- weights -> [piece][Mp][K] bf16 (g1x3_prep_item, gemm1x1_x3.h): once per launch, or once per step by the weight cache's refresh
__global__ __launch_bounds__(256) void g1x3_prep_kernel(const float* __restrict__ w, unsigned short* __restrict__ wa, int tr, int M, int Mp, int K) {
    g1x3_prep_item(w, wa, blockIdx.x * 256 + threadIdx.x, tr, M, Mp, K);
}

// element offset of (b, channel 0, first pixel) of a 4-pixel group of the flattened (b, p) dimension, clamped to the last group
__device__ __forceinline__ size_t x3_pix_off(int n, int N, int P, int Wo, int C, int Hi, int Wi, int s) {
    const int nn = min(n, N - 4);
    const int b = nn / P, p = nn - b * P;
    if (s == 1) return (size_t)b * C * P + p;
    const int py = p / Wo, px = p - py * Wo;
    return (size_t)b * C * Hi * Wi + (size_t)(py * s) * Wi + px * s;
}
template <int S>
__device__ __forceinline__ x3f4 x3_load_pix4(const float* p) {
    if constexpr (S == 1) {
        return *reinterpret_cast<const x3f4*>(p);
    } else {
        const x3f4 u = *reinterpret_cast<const x3f4*>(p), v = *reinterpret_cast<const x3f4*>(p + 4);
        return x3f4{u.x, u.z, v.x, v.z};
    }
}

// ---- S = 3: the 3 x 3 / stride 2 / padding 1 convolution as the same GEMM (reference: torchvision BasicBlock / Bottleneck conv with stride 2
// behind networks/resnet_encoder.py:74-98).  Reduction index k = ci * 9 + ky * 3 + kx -- the weight tensor (Co, Ci, 3, 3) IS the row-major
// A matrix -- and the B element (k, output pixel (oy, ox)) is x[b, ci, 2 oy - 1 + ky, 2 ox - 1 + kx], zero outside the image.  A 4-pixel group
// of one output row reads input columns 2 ox0 - 1 + kx + {0, 2, 4, 6}: four dword loads from clamped addresses, masked afterwards (one load
// shape whatever the tap; only the top row with ky = 0 and the left column with kx = 0 ever fall outside: 2 oy + 1 <= Hi - 1, 2 ox + 1 <= Wi - 1).
struct X3Grp { long base; int top, left; };          // float offset of x[b, 0, 2 oy - 1, 2 ox0 - 1] (may be negative), border flags
__device__ __forceinline__ X3Grp x3_group3(int n, int N, int P, int Wo, int C, int Hi, int Wi) {
    const int nn = min(n, N - 4);
    const int b = nn / P, p = nn - b * P;
    const int oy = p / Wo, ox0 = p - oy * Wo;
    return X3Grp{((long)b * C * Hi + (2 * oy - 1)) * Wi + (2 * ox0 - 1), oy == 0, ox0 == 0};
}
__device__ __forceinline__ x3f4 x3_gather3(const float* __restrict__ x, const X3Grp& g, int k, int Hi, int Wi) {
    const int ci = k / 9, tap = k - 9 * ci, ky = tap / 3, kx = tap - 3 * ky;
    const long o = g.base + ((long)ci * Hi + ky) * Wi + kx;
    const bool rok = !(g.top && ky == 0), e0 = rok && !(g.left && kx == 0);
    const float v0 = x[o > 0 ? o : 0], v1 = x[o + 2 > 0 ? o + 2 : 0], v2 = x[o + 4 > 0 ? o + 4 : 0], v3 = x[o + 6 > 0 ? o + 6 : 0];
    return x3f4{e0 ? v0 : 0.f, rok ? v1 : 0.f, rok ? v2 : 0.f, rok ? v3 : 0.f};
}

extern __shared__ unsigned short g1x3_smem[];

// One reduction chunk (32 k) of a 16 x 16 tile: the six partial products, smallest first, summed from ZERO into a chunk total that is then
// added to the running accumulator by one fp32 add.  Chained through the accumulator itself, every one of the six would round at the
// accumulator's magnitude (6 roundings of ulp(acc) / 2 per chunk, however small the term); this way the five small terms round at the chunk
// total's magnitude and the accumulator takes ONE rounding per 32 k -- against one per 4 k in the fp32-MFMA kernels (v_mfma_f32_16x16x4_f32).
// Same six matrix instructions; the extra cost is four v_fma_f32 per tile and chunk.  (X3_BLOCKED_SUM=0: the chained form, for the A/B.)
// `sgn`: the matrix instruction does not round its fp32 result to nearest -- measured (tools/diag_x3_bias.py): with operands that are not
// exact in bf16 the error of a long reduction has a NEGATIVE MEAN of the size of its rms (truncation toward -inf), which grows linearly
// with the number of chunks (weight gradients: thousands).  So one of the two operands is staged with the sign sgn = +-1 of the chunk's
// parity and the chunk total enters as sgn * t: odd chunks are truncated upward, even chunks downward, and the bias cancels pairwise.
#ifndef X3_BLOCKED_SUM
#define X3_BLOCKED_SUM 1
#endif
__device__ __forceinline__ x3f4 x3_chunk(const x3bf8 (&av)[3], const x3bf8 (&bv)[3], x3f4 c, float sgn) {
#if X3_BLOCKED_SUM
    x3f4 t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[2], bv[0], x3f4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#else
    x3f4 t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[2], bv[0], c, 0, 0, 0);
#endif
    t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[1], bv[1], t, 0, 0, 0);
    t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[0], bv[2], t, 0, 0, 0);
    t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[1], bv[0], t, 0, 0, 0);
    t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[0], bv[1], t, 0, 0, 0);
    t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[0], bv[0], t, 0, 0, 0);
#if X3_BLOCKED_SUM
    return x3f4{fmaf(t.x, sgn, c.x), fmaf(t.y, sgn, c.y), fmaf(t.z, sgn, c.z), fmaf(t.w, sgn, c.w)};
#else
    return t;
#endif
}

// MODE 0: plain (EPI: bias + activation + addends); 1: forward with the BatchNorm + ReLU of its input in the loader; 2 / 3: data gradient
// with the BatchNorm-backward epilogue (ReLU decision re-derived from the raw input / read from the forward's bit mask, + addend).
// MODE 0 / 1 take the statistics epilogue when a.stat_part is set.  MODE 4: data gradient of a STRIDE-2 convolution -- the GEMM over the
// output pixels is the stride-1 one; the store scatters each value to (2 py, 2 px) of the input map and writes the zeros of the cells
// the stride skips, plus the addends of the input's other consumers over the whole cell block (every cell written exactly once).
template <int MT, int NT, int S, bool EPI, int MODE = 0>
__global__ __launch_bounds__(256, (X3T<MT, NT>::PER_CU)) void g1x3_kernel(G1x3Args a) {
    static_assert(MODE == 0 || (S == 1 && !EPI), "the BatchNorm fold / the stride-2 scatter run the stride-1 GEMM without bias / activation");
    using T = X3T<MT, NT>;
    constexpr int BST = X3B<NT>::ST;
    constexpr int NA = 3 * MT / 2;             // 16-byte A items per thread per chunk (3 pieces x BM rows x 4)
    constexpr int NG = NT;                     // float4 groups of a thread's pixel run (run = 4 NT pixels)
    unsigned short* const As = g1x3_smem;                     // [3][BM][40]
    unsigned short* const Bs = g1x3_smem + 3 * T::APIECE;     // [3][32][BST]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int i16 = lane & 15, kg = lane >> 4;
    const int lb = xcd_logical_block(blockIdx.x, gridDim.x);
    const int m0 = (lb % a.mtiles) * T::BM, n0 = (lb / a.mtiles) * T::BN;
    const int P = a.Ho * a.Wo, N = a.B * P;
    const size_t plane = MODE == 4 ? (size_t)P : (size_t)a.Hi * a.Wi;      // channel stride of the B operand's source (MODE 4: a.Hi x a.Wi is the scatter target)
    const int nch = a.K / X3_KC;

    // ---- staging roles
    const unsigned short* asrc[NA];
    int adst[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int idx = tid + j * 256, piece = idx / (4 * T::BM), rem = idx - piece * (4 * T::BM), row = rem >> 2, ch = rem & 3;
        asrc[j] = a.wa + ((size_t)piece * a.Mp + m0 + row) * a.K + ch * 8;
        adst[j] = piece * T::APIECE + row * X3_AST + ch * 8;
    }
    // B: reduction row k = tid / 8, pixel run = tid % 8: 4 NT consecutive pixels = lanes' i = 4 (run % 4) .. + 3, all NT tiles
    const int bk = tid >> 3, run = tid & 7;
    const float* bsrc[NG];
    X3Grp bgrp[S == 3 ? NG : 1];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if constexpr (S == 3) bgrp[g] = x3_group3(n0 + run * 4 * NT + 4 * g, N, P, a.Wo, a.K / 9, a.Hi, a.Wi);
        else bsrc[g] = a.x + x3_pix_off(n0 + run * 4 * NT + 4 * g, N, P, a.Wo, a.K, a.Hi, a.Wi, S) + (size_t)bk * plane;
    }
    // LDS position of pixel NT i + nt of wave column block wn': wn' * 16 NT + nt * 16 + i
    const int bdst = bk * BST + (run >> 2) * 16 * NT + (run & 3) * 4;

    x3u4 ra[NA];
    x3f4 rb[NG];
    // MODE 1: (BatchNorm group of this thread's pixel run) * K + its reduction row: the run lies in one image (P % 16 == 0)
    int bn_tab = 0;
    float bsc = 1.f, bsh = 0.f;
    if constexpr (MODE == 1) bn_tab = (min(n0 + run * 4 * NT, N - 4) / P / a.npg) * a.K + bk;
    auto load = [&](int c) {
#pragma unroll
        for (int j = 0; j < NA; ++j) ra[j] = *reinterpret_cast<const x3u4*>(asrc[j] + c * X3_KC);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if constexpr (S == 3) rb[g] = x3_gather3(a.x, bgrp[g], c * X3_KC + bk, a.Hi, a.Wi);
            else rb[g] = x3_load_pix4<S>(bsrc[g] + (size_t)c * X3_KC * plane);
        }
        if constexpr (MODE == 1) { bsc = a.in_scale[bn_tab + c * X3_KC]; bsh = a.in_shift[bn_tab + c * X3_KC]; }
    };
    auto commit = [&](int c) {
        const float sgn = X3_BLOCKED_SUM && (c & 1) ? -1.f : 1.f;         // (x3_chunk: the chunk's sign rides on the B operand)
#pragma unroll
        for (int j = 0; j < NA; ++j) *reinterpret_cast<x3u4*>(As + adst[j]) = ra[j];
        // the run's pixel j = NT il + nt (il = 0..3: consecutive lanes i): for a tile nt the four values are consecutive positions
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            float v[4];
#pragma unroll
            for (int il = 0; il < 4; ++il) {
                const int j = NT * il + nt;
                v[il] = rb[j >> 2][j & 3];
                if constexpr (MODE == 1) v[il] = fmaxf(fmaf(v[il], bsc, bsh), 0.f);
                v[il] *= sgn;
            }
            unsigned p0, p1, p2, q0, q1, q2;
            x3_split2(v[0], v[1], p0, p1, p2);
            x3_split2(v[2], v[3], q0, q1, q2);
            *reinterpret_cast<x3u2*>(Bs + bdst + nt * 16) = x3u2{p0, q0};
            *reinterpret_cast<x3u2*>(Bs + T::BPIECE + bdst + nt * 16) = x3u2{p1, q1};
            *reinterpret_cast<x3u2*>(Bs + 2 * T::BPIECE + bdst + nt * 16) = x3u2{p2, q2};
        }
    };

    x3f4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = x3f4{0.f, 0.f, 0.f, 0.f};

    // operand addresses (bf16 element offsets)
    const int aoff = (wm * 16 * MT + i16) * X3_AST + kg * 8;                              // + mt * 16 * X3_AST, + piece * APIECE
    const int boff = (4 * kg + (i16 >> 2)) * BST + wn * 16 * NT + 4 * (i16 & 3);          // + nt * 16, + 16 rows for the second half
    auto compute = [&](int c) {
        const float sgn = X3_BLOCKED_SUM && (c & 1) ? -1.f : 1.f;
        x3bf8 b[NT][3];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const unsigned short* base = Bs + s * T::BPIECE + boff + nt * 16;
                const x3s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) x3s4*)(base));
                const x3s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) x3s4*)(base + 16 * BST));
                const x3s8 v = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                b[nt][s] = __builtin_bit_cast(x3bf8, v);
            }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            x3bf8 av[3];
#pragma unroll
            for (int s = 0; s < 3; ++s)
                av[s] = *reinterpret_cast<const x3bf8*>(As + s * T::APIECE + aoff + mt * 16 * X3_AST);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                acc[mt][nt] = x3_chunk(av, b[nt], acc[mt][nt], sgn);
            }
        }
    };

    load(0);
    commit(0);
    __syncthreads();
    for (int c = 0; c < nch; ++c) {
        if (c + 1 < nch) load(c + 1);
        compute(c);
        __syncthreads();
        if (c + 1 < nch) commit(c + 1);
        __syncthreads();
    }

    // ---- epilogue: lane (i16, kg) holds rows m = 16 mt + 4 kg + r, pixels n = NT i16 + nt (nt = 0..NT-1) of its wave's tile
    const int nb = n0 + wn * 16 * NT + NT * i16;
    const bool nok = nb < N;                      // (N % 4 == 0 and the run is NT <= 4 aligned pixels: whole)
    const int nn = nok ? nb : 0;
    const int b = nn / P, p = nn - b * P;
    const int slot = (lb / a.mtiles) * 2 + wn;    // partial number of the statistics / BatchNorm-backward epilogues: (pixel tile, wave column)
    if constexpr (MODE == 2 || MODE == 3) {
        // BatchNorm-backward epilogue (csrc/gemm1x1.hip g1_dgrad_kernel<.., BNE>): the data gradient [+ the skip's gradient] is
        // masked with the ReLU decision of the activation the forward read, stored as g', and the wave column's partial
        // {sum g', sum g' (x - mean)} per channel goes to a.bwd_part
        const int grp = b / a.npg;
        const int pblk = (P + 255) >> 8;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            float xv[4][NT], ad[4][NT], rmean[4], rsc[4], rsh[4];
            unsigned bits[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ci = min(m0 + wm * 16 * MT + mt * 16 + kg * 4 + r, a.M - 1);
                const size_t o = ((size_t)b * a.M + ci) * P + p;
                const int tab = grp * a.M + ci;
                rmean[r] = a.bn_mean[tab];
                if constexpr (MODE == 2) { rsc[r] = a.in_scale[tab]; rsh[r] = a.in_shift[tab]; }
                if constexpr (NT == 4) {
                    const x3f4 t = *reinterpret_cast<const x3f4*>(a.bn_x + o);
                    xv[r][0] = t.x; xv[r][1] = t.y; xv[r][2] = t.z; xv[r][3] = t.w;
                } else {
                    const float2 t = *reinterpret_cast<const float2*>(a.bn_x + o);
                    xv[r][0] = t.x; xv[r][1] = t.y;
                }
                if (MODE == 3 && a.addend) {
                    if constexpr (NT == 4) {
                        const x3f4 t = *reinterpret_cast<const x3f4*>(a.addend + o);
                        ad[r][0] = t.x; ad[r][1] = t.y; ad[r][2] = t.z; ad[r][3] = t.w;
                    } else {
                        const float2 t = *reinterpret_cast<const float2*>(a.addend + o);
                        ad[r][0] = t.x; ad[r][1] = t.y;
                    }
                } else {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) ad[r][nt] = 0.f;
                }
                if constexpr (MODE == 3) {
                    // bit l of word w of a 256-element block <-> element 4 l + w (bn_apply_kernel's wave ballots)
                    const int l = (p & 255) >> 2, w0 = p & 3;
                    const unsigned* mw = reinterpret_cast<const unsigned*>(a.bn_mask + (((size_t)b * a.M + ci) * pblk + (p >> 8)) * 4) + (l >> 5);
                    unsigned bt = 0;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) bt |= ((mw[2 * (w0 + nt)] >> (l & 31)) & 1u) << nt;
                    bits[r] = bt;
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ci = m0 + wm * 16 * MT + mt * 16 + kg * 4 + r;
                float g[NT], sv = 0.f, qv = 0.f;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const float x = xv[r][nt];
                    bool keep;
                    if constexpr (MODE == 2) keep = fmaf(x, rsc[r], rsh[r]) > 0.f;
                    else keep = (bits[r] >> nt) & 1u;
                    const float gv = acc[mt][nt][r] + ad[r][nt];
                    g[nt] = keep ? gv : 0.f;
                    sv += g[nt]; qv = fmaf(g[nt], x - rmean[r], qv);
                }
                if (!nok) { sv = 0.f; qv = 0.f; }
                sv = x3_row16_sum(sv); qv = x3_row16_sum(qv);
                if (ci < a.M) {
                    if (i16 == 0) *reinterpret_cast<float2*>(a.bwd_part + ((size_t)ci * a.bwd_nparts + slot) * 2) = make_float2(sv, qv);
                    if (nok) {
                        float* dst = a.out + ((size_t)b * a.M + ci) * P + p;
                        if constexpr (NT == 4) *reinterpret_cast<x3f4*>(dst) = x3f4{g[0], g[1], g[2], g[3]};
                        else *reinterpret_cast<float2*>(dst) = make_float2(g[0], g[1]);
                    }
                }
            }
        }
        return;
    }
    if constexpr (MODE == 4) {
        // a.Hi x a.Wi = the INPUT map (2 Ho x 2 Wo); this lane's NT output pixels lie in one output row (Wo % 4 == 0)
        if (!nok) return;
        const int Wo = a.Wi >> 1, Po = (a.Hi >> 1) * Wo;
        const int bo = nb / Po, po = nb - bo * Po;
        const int py = po / Wo, px = po - py * Wo;
        const size_t plane = (size_t)a.Hi * a.Wi;
        const size_t pix = (size_t)(2 * py) * a.Wi + 2 * px;
        constexpr int CV = NT / 2;                // 16-byte vectors per cell row (2 NT floats)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            x3f4 c0[4][CV], c1[4][CV];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int v = 0; v < CV; ++v) { c0[r][v] = x3f4{0.f, 0.f, 0.f, 0.f}; c1[r][v] = x3f4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll 1
            for (int which = 0; which < 2; ++which) {
                const float* addp = which == 0 ? a.addend : a.addend2;
                if (!addp) continue;
                x3f4 t0[4][CV], t1[4][CV];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ci = min(m0 + wm * 16 * MT + mt * 16 + kg * 4 + r, a.M - 1);
                    const float* src = addp + ((size_t)bo * a.M + ci) * plane + pix;
#pragma unroll
                    for (int v = 0; v < CV; ++v) {
                        t0[r][v] = *reinterpret_cast<const x3f4*>(src + 4 * v);
                        t1[r][v] = *reinterpret_cast<const x3f4*>(src + a.Wi + 4 * v);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int v = 0; v < CV; ++v) { c0[r][v] += t0[r][v]; c1[r][v] += t1[r][v]; }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ci = m0 + wm * 16 * MT + mt * 16 + kg * 4 + r;
                if (ci >= a.M) continue;
                float* dst = a.out + ((size_t)bo * a.M + ci) * plane + pix;
#pragma unroll
                for (int v = 0; v < CV; ++v) {
                    x3f4 o = c0[r][v];
                    o.x += acc[mt][2 * v][r]; o.z += acc[mt][2 * v + 1][r];
                    *reinterpret_cast<x3f4*>(dst + 4 * v) = o;
                    *reinterpret_cast<x3f4*>(dst + a.Wi + 4 * v) = c1[r][v];
                }
            }
        }
        return;
    }
    const bool stats = MODE <= 1 && a.stat_part != nullptr;       // (wave-uniform)
    if constexpr (EPI) {
        if (a.bias || a.act != ACT_NONE) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + wm * 16 * MT + mt * 16 + kg * 4 + r;
                    const float bv = (a.bias && m < a.M) ? a.bias[m] : 0.f;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[mt][nt][r] = act_fwd(acc[mt][nt][r] + bv, a.act);
                }
        }
        // the other gradient(s) of a residual fork: ALL of this lane's addend values are requested (16-byte loads) before the first is
        // used -- a load next to its store would wait for itself 4 MT times over (csrc/gemm1x1.hip g1_dgrad_kernel)
#pragma unroll 1
        for (int which = 0; which < 2; ++which) {
            const float* addp = which == 0 ? a.addend : a.addend2;
            if (!addp || !nok) continue;
            float adv[MT][4][NT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = min(m0 + wm * 16 * MT + mt * 16 + kg * 4 + r, a.M - 1);
                    const float* src = addp + ((size_t)b * a.M + m) * P + p;
                    if constexpr (NT == 4) {
                        const x3f4 t = *reinterpret_cast<const x3f4*>(src);
                        adv[mt][r][0] = t.x; adv[mt][r][1] = t.y; adv[mt][r][2] = t.z; adv[mt][r][3] = t.w;
                    } else {
                        const float2 t = *reinterpret_cast<const float2*>(src);
                        adv[mt][r][0] = t.x; adv[mt][r][1] = t.y;
                    }
                }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[mt][nt][r] += adv[mt][r][nt];
        }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + wm * 16 * MT + mt * 16 + kg * 4 + r;
            const bool mok = m < a.M;
            const size_t o = ((size_t)b * a.M + min(m, a.M - 1)) * P + p;
            float v[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) v[nt] = acc[mt][nt][r];
            if (nok && mok) {
                if constexpr (NT == 4) *reinterpret_cast<x3f4*>(a.out + o) = x3f4{v[0], v[1], v[2], v[3]};
                else *reinterpret_cast<float2*>(a.out + o) = make_float2(v[0], v[1]);
            }
            if (stats) {
                float sv = 0.f, qv = 0.f;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) { sv += v[nt]; qv = fmaf(v[nt], v[nt], qv); }
                if (!nok) { sv = 0.f; qv = 0.f; }
                sv = x3_row16_sum(sv); qv = x3_row16_sum(qv);
                if (i16 == 0 && mok) *reinterpret_cast<float2*>(a.stat_part + ((size_t)m * a.stat_nparts + slot) * 2) = make_float2(sv, qv);
            }
        }
}

// =====================================================================================================================
// weight gradient  dw[m,k] = sum_n gy[m,n] x[k,n]  (n = flattened (b, pixel)): both operands are REDUCTION-contiguous in memory
// (8 consecutive pixels per lane: plain ds_read_b128, no transposed read), both are split while staging (about twice the
// vector work per chunk of the forward: 32 values per thread), 128 (co) x 128 (ci) per block, the reduction split over
// blockIdx.y into fp32 slabs summed in fixed order by slab_reduce16_kernel (as the fp32 kernel: deterministic, no atomics).
// =====================================================================================================================
struct G1x3WArgs {
    const float* gy;            // (B, Co, Ho, Wo)
    const float* x;             // (B, Ci, Hi, Wi)
    float* out;                 // slabs [split][Co][Ci], or dw itself when there is one split
    int B, Co, Ci, Hi, Wi, Ho, Wo;
    int mtiles, ntiles, splits, chunks;
    const float* in_scale;      // BNIN: x is the raw input of a BatchNorm + ReLU folded into the forward's loader: the weight gradient
    const float* in_shift;      // needs the same relu(scale[g, ci] x + shift[g, ci]), re-formed between the global load and the split
    int npg;
};

template <int MT, int NT, int S, bool BNIN = false>
__global__ __launch_bounds__(256, (X3T<MT, NT>::PER_CU)) void g1x3_wgrad_kernel(G1x3WArgs a) {
    static_assert(!BNIN || S == 1, "the BatchNorm fold exists at stride 1");
    // S = 3: the columns of dw are k = ci * 9 + tap of a 3 x 3 / 2 convolution (a.Ci = 9 * input channels), B[k][n] gathered as in the forward
    using T = X3T<MT, NT>;
    constexpr int BPIECE = T::BN * X3_AST;
    unsigned short* const As = g1x3_smem;                     // [3][BM co][40]
    unsigned short* const Bs = g1x3_smem + 3 * T::APIECE;     // [3][BN ci][40]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int i16 = lane & 15, kg = lane >> 4;
    // all tiles of one split read the same pixel range: adjacent logical indices = one XCD (g1_wgrad_kernel)
    const int tiles = gridDim.x, lb = xcd_logical_block(blockIdx.y * tiles + blockIdx.x, tiles * gridDim.y);
    const int split = lb / tiles, tile = lb - split * tiles;
    const int m0 = (tile % a.mtiles) * T::BM, c0 = (tile / a.mtiles) * T::BN;
    const int P = a.Ho * a.Wo;
    const size_t plane = (size_t)a.Hi * a.Wi;
    const int kq = tid & 7, row0 = tid >> 3;                  // 4-pixel group of the chunk, first of the thread's rows (+32 each)
    size_t arow[MT], brow[NT];
#pragma unroll
    for (int j = 0; j < MT; ++j) arow[j] = (size_t)min(m0 + row0 + 32 * j, a.Co - 1) * P;
#pragma unroll
    for (int j = 0; j < NT; ++j) brow[j] = (size_t)min(c0 + row0 + 32 * j, a.Ci - 1) * plane;
    x3f4 ra[MT], rb[NT];
    float wsc[BNIN ? NT : 1], wsh[BNIN ? NT : 1];
    auto load = [&](int ch) {
        const int n = ch * X3_KC + kq * 4;
        const int b = n / P, p = n - b * P;
        if constexpr (BNIN) {
            const int tab = (b / a.npg) * a.Ci;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int cch = min(c0 + row0 + 32 * j, a.Ci - 1);
                wsc[j] = a.in_scale[tab + cch]; wsh[j] = a.in_shift[tab + cch];
            }
        }
        size_t pix = 0;
        if constexpr (S == 1) {
            pix = p;
        } else if constexpr (S == 2) {
            const int py = p / a.Wo, px = p - py * a.Wo;
            pix = (size_t)(py * 2) * a.Wi + px * 2;
        }
        const float* ga = a.gy + (size_t)b * a.Co * P + p;
#pragma unroll
        for (int j = 0; j < MT; ++j) ra[j] = *reinterpret_cast<const x3f4*>(ga + arow[j]);
        if constexpr (S == 3) {
            const int oy = p / a.Wo, ox0 = p - oy * a.Wo;
            const X3Grp g{((long)b * (a.Ci / 9) * a.Hi + (2 * oy - 1)) * a.Wi + (2 * ox0 - 1), oy == 0, ox0 == 0};
#pragma unroll
            for (int j = 0; j < NT; ++j) rb[j] = x3_gather3(a.x, g, min(c0 + row0 + 32 * j, a.Ci - 1), a.Hi, a.Wi);
        } else {
            const float* xb = a.x + (size_t)b * a.Ci * plane + pix;
#pragma unroll
            for (int j = 0; j < NT; ++j) rb[j] = x3_load_pix4<S>(xb + brow[j]);
        }
    };
    auto put = [&](unsigned short* base, int piece_stride, int d, x3f4 v) {
        unsigned p0, p1, p2, q0, q1, q2;
        x3_split2(v.x, v.y, p0, p1, p2);
        x3_split2(v.z, v.w, q0, q1, q2);
        *reinterpret_cast<x3u2*>(base + d) = x3u2{p0, q0};
        *reinterpret_cast<x3u2*>(base + piece_stride + d) = x3u2{p1, q1};
        *reinterpret_cast<x3u2*>(base + 2 * piece_stride + d) = x3u2{p2, q2};
    };
    auto commit = [&](int ch) {
        const float sgn = X3_BLOCKED_SUM && (ch & 1) ? -1.f : 1.f;        // (x3_chunk: the chunk's sign rides on the gy operand)
#pragma unroll
        for (int j = 0; j < MT; ++j) put(As, T::APIECE, (row0 + 32 * j) * X3_AST + kq * 4, ra[j] * sgn);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            if constexpr (BNIN)
                rb[j] = x3f4{fmaxf(fmaf(rb[j].x, wsc[j], wsh[j]), 0.f), fmaxf(fmaf(rb[j].y, wsc[j], wsh[j]), 0.f),
                             fmaxf(fmaf(rb[j].z, wsc[j], wsh[j]), 0.f), fmaxf(fmaf(rb[j].w, wsc[j], wsh[j]), 0.f)};
            put(Bs, BPIECE, (row0 + 32 * j) * X3_AST + kq * 4, rb[j]);
        }
    };
    x3f4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = x3f4{0.f, 0.f, 0.f, 0.f};
    const int aoff = (wm * 16 * MT + i16) * X3_AST + kg * 8, boff = (wn * 16 * NT + i16) * X3_AST + kg * 8;
    auto compute = [&](int ch) {
        const float sgn = X3_BLOCKED_SUM && (ch & 1) ? -1.f : 1.f;
        x3bf8 b[NT][3];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int s = 0; s < 3; ++s) b[nt][s] = *reinterpret_cast<const x3bf8*>(Bs + s * BPIECE + boff + nt * 16 * X3_AST);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            x3bf8 av[3];
#pragma unroll
            for (int s = 0; s < 3; ++s) av[s] = *reinterpret_cast<const x3bf8*>(As + s * T::APIECE + aoff + mt * 16 * X3_AST);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                acc[mt][nt] = x3_chunk(av, b[nt], acc[mt][nt], sgn);
            }
        }
    };
    const int per = (a.chunks + a.splits - 1) / a.splits;
    const int ch0 = split * per, ch1 = min(ch0 + per, a.chunks);
    if (ch0 < ch1) {
        load(ch0);
        commit(ch0);
    }
    __syncthreads();
    for (int ch = ch0; ch < ch1; ++ch) {
        const bool more = ch + 1 < ch1;
        if (more) load(ch + 1);
        compute(ch);
        __syncthreads();
        if (more) commit(ch + 1);
        __syncthreads();
    }
    float* slab = a.out + (size_t)split * a.Co * a.Ci;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * 16 * MT + mt * 16 + kg * 4 + r;
                const int ci = c0 + wn * 16 * NT + nt * 16 + i16;
                if (m < a.Co && ci < a.Ci) slab[(size_t)m * a.Ci + ci] = acc[mt][nt][r];
            }
}

// fixed-order slab sum (the float4 form of gemm_tiles.h's slab_reduce16_kernel)
__global__ __launch_bounds__(256) void g1x3_slabsum_kernel(const x3f4* __restrict__ slab, x3f4* __restrict__ out, int splits, int n4) {
    __shared__ x3f4 sm[256];
    const int g = threadIdx.x >> 4, l = threadIdx.x & 15;
    const int i = blockIdx.x * 16 + l;
    const int per = (splits + 15) / 16, s0 = g * per, s1 = min(s0 + per, splits);
    x3f4 t = {0.f, 0.f, 0.f, 0.f};
    if (i < n4)
        for (int s = s0; s < s1; ++s) t += slab[(size_t)s * n4 + i];
    sm[threadIdx.x] = t;
    __syncthreads();
    if (g == 0 && i < n4) {
        x3f4 r = sm[l];
#pragma unroll
        for (int k = 1; k < 16; ++k) r += sm[k * 16 + l];
        out[i] = r;
    }
}

// default ON (env DC_G1_X3=0 / dc_set_gemm_split(0): the fp32-MFMA kernels of gemm1x1.hip, the A/B): the gate of VERDICT round 5 item 1
// -- error vs fp64 no worse than the fp32-MFMA kernels', every fp64-calibrated test green with unchanged thresholds -- is met
// (tests/test_conv1x1_gpu.py, tests/test_encoder_gpu.py, tests/test_train_gpu.py under this default)
static int g_g1x3 = [] { const char* f = getenv("DC_G1_X3"); return f ? atoi(f) : 1; }();

template <typename K>
static bool x3_set_lds(K kernel, size_t bytes) {
    return hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess;
}

// ---- tile choice: the largest tile whose grid still fills the chip's resident slots (a 128 x 128 tile amortises the operand split and
// the LDS traffic best, but 80 blocks of it on 256 CUs lose to 320 blocks of 64 x 64: the layer4 shapes of BASELINE configs[2])
struct X3Tile { int mt, nt; };
static X3Tile x3_pick(int M, int N, int stride) {
    if (const char* f = getenv("DC_G1X3_TILE")) {                 // experiments: "MT,NT"
        int mt = 0, nt = 0;
        if (sscanf(f, "%d,%d", &mt, &nt) == 2 && (mt == 2 || mt == 4) && (nt == 2 || nt == 4)) return {mt, nt};
    }
    const X3Tile cand[4] = {{4, 4}, {4, 2}, {2, 4}, {2, 2}};
    const int per_cu[4] = {2, 3, 3, 4};
    // relative efficiency of a block's inner loop (operand traffic, split amortisation), from tools/bench_g1x3.py sweeps of the resnet50
    // shapes (profiles/round6_g1x3_per_shape.txt); the stride-2 gather (two 16-byte loads per 4 pixels) favours the 64-pixel tiles
    const double eff1[4] = {1.0, 0.97, 0.9, 0.85}, eff2[4] = {0.8, 1.0, 0.75, 0.85};
    const double* eff = stride == 2 ? eff2 : eff1;
    double best = -1.0;
    X3Tile pick = cand[3];
    for (int i = 0; i < 4; ++i) {
        if (cand[i].mt == 4 && M <= 64) continue;              // (half of a 128-row tile would be padding)
        const long blocks = (long)ceil_div(M, 32 * cand[i].mt) * ceil_div(N, 32 * cand[i].nt);
        const long slots = 256L * per_cu[i];
        const long rounds = (blocks + slots - 1) / slots;
        const double fill = (double)blocks / (double)(rounds * slots);
        const double pad = (double)M / (double)(ceil_div(M, 32 * cand[i].mt) * 32 * cand[i].mt);
        const double score = eff[i] * fill * pad;
        if (score > best) { best = score; pick = cand[i]; }
    }
    return pick;
}

}  // namespace dc

using namespace dc;

extern "C" int dc_set_gemm_split(int mode) {
    if (mode != 0 && mode != 1 && mode != 3) return DC_EINVAL;      // (3 = 1 + the 3 x 3 / 2 trunk convolutions: an experiment, see depthcore.h)
    const int prev = g_g1x3;
    g_g1x3 = mode;
    return prev;
}
extern "C" int dc_get_gemm_split(void) { return g_g1x3; }

// shapes the split kernels take: pixel runs of 16 never straddle an image, reduction chunks of 32, 16-byte alignment everywhere
static bool g1x3_common(int B, int Ci, int Co, int Hi, int Wi, int stride) {
    if (B <= 0 || Ci <= 0 || Co <= 0 || Hi <= 0 || Wi <= 0 || (stride != 1 && stride != 2)) return false;
    if (stride == 2 && ((Hi & 1) || (Wi & 1) || ((Wi / 2) & 3))) return false;
    const int P = (Hi / stride) * (Wi / stride);
    if (P % 16) return false;
    if ((size_t)B * std::max(Ci, Co) * Hi * Wi >= (1ull << 31)) return false;
    return true;
}
extern "C" int dc_gemm1x1x3_fwd_ok(int B, int Ci, int Co, int Hi, int Wi, int stride) {
    return g1x3_common(B, Ci, Co, Hi, Wi, stride) && Ci % 32 == 0 && Co >= 32;
}
extern "C" int dc_gemm1x1x3_dgrad_ok(int B, int Ci, int Co, int Hi, int Wi, int stride) {
    return g1x3_common(B, Ci, Co, Hi, Wi, stride) && Co % 32 == 0 && Ci >= 32;
}
// bytes of the split weights of one launch (either direction): 3 bf16 pieces, rows padded to 128
extern "C" size_t dc_gemm1x1x3_workspace(int Ci, int Co) {
    const size_t a = (size_t)ceil_div(Co, 128) * 128 * Ci, b = (size_t)ceil_div(Ci, 128) * 128 * Co;
    return std::max(a, b) * 3 * 2 + 256;
}

// partials per channel of the forward's statistics epilogue / the data gradient's BatchNorm epilogue (0: not on this shape): one per
// wave column of 16 NT pixels; a group boundary must not fall inside one (csrc/gemm1x1.hip g1_parts, with THIS file's tile choice)
static int g1x3_parts(int rows, int B, int P, int stride, int groups, int* ppg) {
    if (groups < 1 || B % groups) return 0;
    const int N = B * P;
    const X3Tile t = x3_pick(rows, N, stride);
    const int cover = 16 * t.nt;
    if (groups > 1 && (N / groups) % cover) return 0;
    if (ppg) *ppg = (N / groups) / cover;
    return ceil_div(N, 32 * t.nt) * 2;
}
extern "C" int dc_gemm1x1x3_stat_parts(int B, int Ci, int Co, int Hi, int Wi, int stride, int groups, int* ppg) {
    if (!dc_gemm1x1x3_fwd_ok(B, Ci, Co, Hi, Wi, stride)) return 0;
    return g1x3_parts(Co, B, (Hi / stride) * (Wi / stride), stride, groups, ppg);
}
extern "C" int dc_gemm1x1x3_bwd_parts(int B, int Ci, int Co, int Hi, int Wi, int groups, int* ppg) {
    if (!dc_gemm1x1x3_dgrad_ok(B, Ci, Co, Hi, Wi, 1)) return 0;
    return g1x3_parts(Ci, B, Hi * Wi, 1, groups, ppg);
}

template <int MT, int NT>
static int g1x3_go(const G1x3Args& a, dim3 grid, int stride, bool epi, int mode, hipStream_t st) {
    using T = X3T<MT, NT>;
    static const bool attr = x3_set_lds(g1x3_kernel<MT, NT, 1, false>, T::LDS) && x3_set_lds(g1x3_kernel<MT, NT, 1, true>, T::LDS) &&
                             x3_set_lds(g1x3_kernel<MT, NT, 2, false>, T::LDS) && x3_set_lds(g1x3_kernel<MT, NT, 2, true>, T::LDS) &&
                             x3_set_lds(g1x3_kernel<MT, NT, 1, false, 1>, T::LDS) && x3_set_lds(g1x3_kernel<MT, NT, 1, false, 2>, T::LDS) &&
                             x3_set_lds(g1x3_kernel<MT, NT, 1, false, 3>, T::LDS) && x3_set_lds(g1x3_kernel<MT, NT, 1, false, 4>, T::LDS) &&
                             x3_set_lds(g1x3_kernel<MT, NT, 3, false>, T::LDS);
    if (!attr) return DC_ELAUNCH;
    if (stride == 3) hipLaunchKernelGGL((g1x3_kernel<MT, NT, 3, false>), grid, dim3(256), T::LDS, st, a);
    else if (mode == 4) hipLaunchKernelGGL((g1x3_kernel<MT, NT, 1, false, 4>), grid, dim3(256), T::LDS, st, a);
    else if (mode == 1) hipLaunchKernelGGL((g1x3_kernel<MT, NT, 1, false, 1>), grid, dim3(256), T::LDS, st, a);
    else if (mode == 2) hipLaunchKernelGGL((g1x3_kernel<MT, NT, 1, false, 2>), grid, dim3(256), T::LDS, st, a);
    else if (mode == 3) hipLaunchKernelGGL((g1x3_kernel<MT, NT, 1, false, 3>), grid, dim3(256), T::LDS, st, a);
    else if (stride == 1) {
        if (epi) hipLaunchKernelGGL((g1x3_kernel<MT, NT, 1, true>), grid, dim3(256), T::LDS, st, a);
        else hipLaunchKernelGGL((g1x3_kernel<MT, NT, 1, false>), grid, dim3(256), T::LDS, st, a);
    } else {
        if (epi) hipLaunchKernelGGL((g1x3_kernel<MT, NT, 2, true>), grid, dim3(256), T::LDS, st, a);
        else hipLaunchKernelGGL((g1x3_kernel<MT, NT, 2, false>), grid, dim3(256), T::LDS, st, a);
    }
    return DC_OK;
}

// dgrad: M = Ci rows, K = Co, src = gy; forward: M = Co, K = Ci, src = x.  bn (nullable): the fold, validated by the entry points
static int g1x3_launch(const float* src, const float* weight, float* out, void* ws, const float* bias, int act, const float* addend,
                       const float* addend2, int B, int M, int K, int Co, int Ci, int tr, int Hi, int Wi, int stride, const dc_bn_fold* bn,
                       hipStream_t st) {
    if (((size_t)ws & 15) || ((size_t)src & 15) || ((size_t)out & 15)) return DC_EINVAL;
    G1x3Args a{};
    const bool scatter = stride == -2;            // data gradient of a stride-2 convolution: (Hi, Wi) is the OUTPUT map here
    if (scatter) stride = 1;
    a.Mp = ceil_div(M, 128) * 128;
    a.wa = (unsigned short*)ws; a.x = src; a.out = out; a.bias = bias; a.addend = addend; a.addend2 = addend2; a.act = act;
    a.B = B; a.M = M; a.K = K; a.Hi = Hi; a.Wi = Wi; a.Ho = Hi / stride; a.Wo = Wi / stride;
    const int N = B * a.Ho * a.Wo;
    const X3Tile t = x3_pick(M, N, stride);
    a.mtiles = ceil_div(M, 32 * t.mt); a.ntiles = ceil_div(N, 32 * t.nt);
    int mode = scatter ? 4 : 0;
    if (bn) {
        if (bn->groups < 1 || B % bn->groups) return DC_EINVAL;
        a.npg = B / bn->groups;
        if (!tr) {                                              // forward: loader and / or statistics epilogue
            if (bn->in_scale) {
                if (!bn->in_shift || stride != 1 || bias || act != ACT_NONE) return DC_EINVAL;
                a.in_scale = bn->in_scale; a.in_shift = bn->in_shift;
                mode = 1;
            }
            if (bn->stat_part) {
                if (bias || act != ACT_NONE) return DC_EINVAL;
                a.stat_nparts = g1x3_parts(M, B, a.Ho * a.Wo, stride, bn->groups, nullptr);
                if (!a.stat_nparts) return DC_EINVAL;
                a.stat_part = bn->stat_part;
            }
        } else if (bn->bwd_part) {                              // data gradient: BatchNorm-backward epilogue
            a.bwd_nparts = g1x3_parts(M, B, a.Ho * a.Wo, 1, bn->groups, nullptr);
            if (!a.bwd_nparts || addend2 || !bn->bn_x || !bn->bn_mean || (!bn->bn_mask && (!bn->in_scale || !bn->in_shift || addend))) return DC_EINVAL;
            if (bn->bn_mask && ((Hi * Wi) & 3)) return DC_EINVAL;
            a.bn_x = bn->bn_x; a.bn_mean = bn->bn_mean; a.bn_mask = (const unsigned long long*)bn->bn_mask; a.bwd_part = bn->bwd_part;
            a.in_scale = bn->in_scale; a.in_shift = bn->in_shift;
            mode = bn->bn_mask ? 3 : 2;
        }
    }
    // split weights: from the per-step cache (registered weights, after its refresh) or prepared here
    if (const void* cached = wc_lookup_x3(weight, Ci, Co, tr, a.Mp, K, st)) {
        a.wa = (const unsigned short*)cached;
    } else {
        hipLaunchKernelGGL(g1x3_prep_kernel, dim3(ceil_div(a.Mp * (K / 4), 256)), dim3(256), 0, st, weight, (unsigned short*)ws, tr, M, a.Mp, K);
        DC_CHECK_LAUNCH();
    }
    const dim3 grid(a.mtiles * a.ntiles);
    if (scatter) { a.Hi = 2 * Hi; a.Wi = 2 * Wi; }          // (the loads use Ho x Wo = the gy map; the scatter epilogue the input map)
    const bool epi = mode == 0 && (bias || act != ACT_NONE || addend || addend2);
    // family 7: algorithmic = 2 MAC of the GEMM; executed = the six bf16 products of every padded tile (bf16 matrix FLOPs)
    hipEvent_t pe = conv_prof_begin(7, 2.0 * (double)N * M * K, 12.0 * (double)grid.x * (32.0 * t.mt) * (32.0 * t.nt) * K,
                                    4.0 * ((double)N * K + (double)N * M + (double)M * K), st);
    int rc;
    if (t.mt == 4 && t.nt == 4) rc = g1x3_go<4, 4>(a, grid, stride, epi, mode, st);
    else if (t.mt == 2 && t.nt == 4) rc = g1x3_go<2, 4>(a, grid, stride, epi, mode, st);
    else if (t.mt == 4) rc = g1x3_go<4, 2>(a, grid, stride, epi, mode, st);
    else rc = g1x3_go<2, 2>(a, grid, stride, epi, mode, st);
    conv_prof_end(pe, st);
    if (rc != DC_OK) return rc;
    DC_CHECK_LAUNCH();
    return DC_OK;
}

extern "C" int dc_gemm1x1x3_fwd_bn(const float* x, const float* weight, const float* bias, float* y, void* ws, int B, int Ci, int Co, int Hi,
                                   int Wi, int stride, int act, const dc_bn_fold* bn, void* stream) {
    if (!x || !weight || !y || !ws || !dc_gemm1x1x3_fwd_ok(B, Ci, Co, Hi, Wi, stride) || act < 0 || act > ACT_LAST) return DC_EINVAL;
    return g1x3_launch(x, weight, y, ws, bias, act, nullptr, nullptr, B, Co, Ci, Co, Ci, 0, Hi, Wi, stride, bn, (hipStream_t)stream);
}
extern "C" int dc_gemm1x1x3_fwd(const float* x, const float* weight, const float* bias, float* y, void* ws, int B, int Ci, int Co, int Hi,
                                int Wi, int stride, int act, void* stream) {
    return dc_gemm1x1x3_fwd_bn(x, weight, bias, y, ws, B, Ci, Co, Hi, Wi, stride, act, nullptr, stream);
}
extern "C" int dc_gemm1x1x3_dgrad_bn(const float* gy, const float* weight, float* dx, void* ws, const float* addend, const float* addend2,
                                     int B, int Ci, int Co, int Hi, int Wi, int stride, const dc_bn_fold* bn, void* stream) {
    if (!gy || !weight || !dx || !ws || !dc_gemm1x1x3_dgrad_ok(B, Ci, Co, Hi, Wi, stride)) return DC_EINVAL;
    if (stride == 2) {
        // the GEMM runs over the OUTPUT pixels (Hi/2 x Wi/2, stride 1); the scatter epilogue knows the input map through args_s2
        if (bn && (bn->bwd_part || bn->in_scale || bn->stat_part)) return DC_EINVAL;
        return g1x3_launch(gy, weight, dx, ws, nullptr, ACT_NONE, addend, addend2, B, Ci, Co, Co, Ci, 1, Hi / 2, Wi / 2, -2, nullptr, (hipStream_t)stream);
    }
    return g1x3_launch(gy, weight, dx, ws, nullptr, ACT_NONE, addend, addend2, B, Ci, Co, Co, Ci, 1, Hi, Wi, 1, bn, (hipStream_t)stream);
}
extern "C" int dc_gemm1x1x3_dgrad(const float* gy, const float* weight, float* dx, void* ws, const float* addend, const float* addend2, int B,
                                  int Ci, int Co, int Hi, int Wi, int stride, void* stream) {
    return dc_gemm1x1x3_dgrad_bn(gy, weight, dx, ws, addend, addend2, B, Ci, Co, Hi, Wi, stride, nullptr, stream);
}

// ---- weight gradient
static X3Tile x3_wpick(int Co, int Ci) { return {Co >= 128 ? 4 : 2, Ci >= 128 ? 4 : 2}; }
static int g1x3_wsplits(int tiles, int chunks, X3Tile t) {
    const int target = 256 * (t.mt * t.nt >= 16 ? 2 : (t.mt * t.nt >= 8 ? 3 : 4));     // the resident slots of the tile's kernel
    int s = std::max(1, std::min({chunks, ceil_div(target, tiles), 512}));
    return ceil_div(chunks, ceil_div(chunks, s));                            // no empty split: every slab gets written
}
extern "C" int dc_gemm1x1x3_wgrad_ok(int B, int Ci, int Co, int Hi, int Wi, int stride) {
    return g1x3_common(B, Ci, Co, Hi, Wi, stride) && Ci % 4 == 0 && Co >= 32 && Ci >= 32 && ((size_t)B * (Hi / stride) * (Wi / stride)) % 32 == 0;
}
extern "C" size_t dc_gemm1x1x3_wgrad_workspace(int B, int Ci, int Co, int Hi, int Wi, int stride) {
    if (!dc_gemm1x1x3_wgrad_ok(B, Ci, Co, Hi, Wi, stride)) return 0;
    const int chunks = B * (Hi / stride) * (Wi / stride) / X3_KC;
    const X3Tile t = x3_wpick(Co, Ci);
    const int splits = g1x3_wsplits(ceil_div(Co, 32 * t.mt) * ceil_div(Ci, 32 * t.nt), chunks, t);
    return splits > 1 ? (size_t)splits * Co * Ci * sizeof(float) : 16;
}
template <int MT, int NT>
static int g1x3_wgo(const G1x3WArgs& a, dim3 grid, int stride, bool bnin, hipStream_t st) {
    using T = X3T<MT, NT>;
    static const bool attr = x3_set_lds(g1x3_wgrad_kernel<MT, NT, 1>, T::LDSW) && x3_set_lds(g1x3_wgrad_kernel<MT, NT, 2>, T::LDSW) &&
                             x3_set_lds(g1x3_wgrad_kernel<MT, NT, 1, true>, T::LDSW) && x3_set_lds(g1x3_wgrad_kernel<MT, NT, 3>, T::LDSW);
    if (!attr) return DC_ELAUNCH;
    if (stride == 3) hipLaunchKernelGGL((g1x3_wgrad_kernel<MT, NT, 3>), grid, dim3(256), T::LDSW, st, a);
    else if (bnin) hipLaunchKernelGGL((g1x3_wgrad_kernel<MT, NT, 1, true>), grid, dim3(256), T::LDSW, st, a);
    else if (stride == 1) hipLaunchKernelGGL((g1x3_wgrad_kernel<MT, NT, 1>), grid, dim3(256), T::LDSW, st, a);
    else hipLaunchKernelGGL((g1x3_wgrad_kernel<MT, NT, 2>), grid, dim3(256), T::LDSW, st, a);
    return DC_OK;
}
extern "C" int dc_gemm1x1x3_wgrad_bn(const float* x, const float* gy, float* dweight, void* ws, int B, int Ci, int Co, int Hi, int Wi,
                                     int stride, const dc_bn_fold* bn, void* stream) {
    if (!x || !gy || !dweight || !ws || !dc_gemm1x1x3_wgrad_ok(B, Ci, Co, Hi, Wi, stride)) return DC_EINVAL;
    if (((size_t)x & 15) || ((size_t)gy & 15) || ((size_t)ws & 15) || ((size_t)dweight & 15)) return DC_EINVAL;
    G1x3WArgs a{};
    a.gy = gy; a.x = x; a.B = B; a.Co = Co; a.Ci = Ci; a.Hi = Hi; a.Wi = Wi; a.Ho = Hi / stride; a.Wo = Wi / stride;
    const bool bnin = bn && bn->in_scale;
    if (bnin) {
        if (!bn->in_shift || stride != 1 || bn->groups < 1 || B % bn->groups) return DC_EINVAL;
        a.npg = B / bn->groups; a.in_scale = bn->in_scale; a.in_shift = bn->in_shift;
    }
    a.chunks = B * a.Ho * a.Wo / X3_KC;
    const X3Tile t = x3_wpick(Co, Ci);
    a.mtiles = ceil_div(Co, 32 * t.mt); a.ntiles = ceil_div(Ci, 32 * t.nt);
    a.splits = g1x3_wsplits(a.mtiles * a.ntiles, a.chunks, t);
    a.out = a.splits > 1 ? (float*)ws : dweight;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(a.mtiles * a.ntiles, a.splits);
    const double npx = (double)B * a.Ho * a.Wo;
    hipEvent_t pe = conv_prof_begin(7, 2.0 * npx * Co * Ci, 12.0 * npx * (double)a.mtiles * (32.0 * t.mt) * (double)a.ntiles * (32.0 * t.nt),
                                    4.0 * (npx * Ci + npx * Co + (double)Co * Ci), st);
    int rc;
    if (t.mt == 4 && t.nt == 4) rc = g1x3_wgo<4, 4>(a, grid, stride, bnin, st);
    else if (t.mt == 2 && t.nt == 4) rc = g1x3_wgo<2, 4>(a, grid, stride, bnin, st);
    else if (t.mt == 4) rc = g1x3_wgo<4, 2>(a, grid, stride, bnin, st);
    else rc = g1x3_wgo<2, 2>(a, grid, stride, bnin, st);
    conv_prof_end(pe, st);
    if (rc != DC_OK) return rc;
    DC_CHECK_LAUNCH();
    if (a.splits > 1) {
        const int n4 = Co * Ci / 4;
        hipLaunchKernelGGL(g1x3_slabsum_kernel, dim3(ceil_div(n4, 16)), dim3(256), 0, st, (const x3f4*)ws, (x3f4*)dweight, a.splits, n4);
        DC_CHECK_LAUNCH();
    }
    return DC_OK;
}
extern "C" int dc_gemm1x1x3_wgrad(const float* x, const float* gy, float* dweight, void* ws, int B, int Ci, int Co, int Hi, int Wi,
                                  int stride, void* stream) {
    return dc_gemm1x1x3_wgrad_bn(x, gy, dweight, ws, B, Ci, Co, Hi, Wi, stride, nullptr, stream);
}
/* all three passes of a stride-1 shape take the fold on the split kernels */
extern "C" int dc_gemm1x1x3_bn_ok(int B, int Ci, int Co, int Hi, int Wi) {
    return dc_gemm1x1x3_fwd_ok(B, Ci, Co, Hi, Wi, 1) && dc_gemm1x1x3_dgrad_ok(B, Ci, Co, Hi, Wi, 1) && dc_gemm1x1x3_wgrad_ok(B, Ci, Co, Hi, Wi, 1);
}

// ---- the 3 x 3 / stride 2 / padding 1 convolutions of the trunks on the same kernels (S = 3 loaders): forward and weight gradient.  Called by
// dc_convs2_fwd / dc_convs2_wgrad (convgemm.hip) under dc_set_gemm_split(3) ONLY: more accurate than cg_fwd3 / cg_wgrad3 (about half their
// error against fp64, tests/test_convs2_gpu.py) but, with four dword gathers per pixel group in the loader, slower on five of the six step
// shapes (tools/time_convs2.py, MI355X: forward 0.50 - 1.14x, weight gradient 0.74 - 1.25x of the fp32-MFMA kernels' speed; a step: no
// change) -- so the default (mode 1) keeps the fp32-MFMA kernels for this family.  The data gradient has no split form.
namespace dc {
bool g1x3_conv3s2_ok(int B, int Ci, int Co, int Hi, int Wi) {
    if (!g_g1x3 || B <= 0 || Ci % 32 || Co < 32 || (Hi & 1) || (Wi & 1)) return false;
    const int Ho = Hi / 2, Wo = Wi / 2;
    if ((Wo & 3) || (Ho * Wo) % 16) return false;
    if ((size_t)B * std::max(Ci, Co) * Hi * Wi >= (1ull << 31)) return false;
    return true;
}
size_t g1x3_conv3s2_fwd_ws(int Ci, int Co) { return (size_t)ceil_div(Co, 128) * 128 * 9 * Ci * 6 + 512; }
int g1x3_conv3s2_fwd(const float* x, const float* weight, float* y, void* ws, int B, int Ci, int Co, int Hi, int Wi, hipStream_t st) {
    if (((size_t)ws & 15) || ((size_t)x & 15) || ((size_t)y & 15)) return DC_EINVAL;
    G1x3Args a{};
    const int M = Co, K = 9 * Ci;
    a.Mp = ceil_div(M, 128) * 128;
    a.wa = (unsigned short*)ws; a.x = x; a.out = y; a.act = ACT_NONE;
    a.B = B; a.M = M; a.K = K; a.Hi = Hi; a.Wi = Wi; a.Ho = Hi / 2; a.Wo = Wi / 2;
    const int N = B * a.Ho * a.Wo;
    const X3Tile t = x3_pick(M, N, 2);
    a.mtiles = ceil_div(M, 32 * t.mt); a.ntiles = ceil_div(N, 32 * t.nt);
    if (const void* cached = wc_lookup_x3(weight, Ci, Co, 0, a.Mp, K, st)) {
        a.wa = (const unsigned short*)cached;
    } else {
        hipLaunchKernelGGL(g1x3_prep_kernel, dim3(ceil_div(a.Mp * (K / 4), 256)), dim3(256), 0, st, weight, (unsigned short*)ws, 0, M, a.Mp, K);
        DC_CHECK_LAUNCH();
    }
    const dim3 grid(a.mtiles * a.ntiles);
    hipEvent_t pe = conv_prof_begin(7, 2.0 * (double)N * M * K, 12.0 * (double)grid.x * (32.0 * t.mt) * (32.0 * t.nt) * K,
                                    4.0 * ((double)B * Ci * Hi * Wi + (double)N * M + (double)M * K), st);
    int rc;
    if (t.mt == 4 && t.nt == 4) rc = g1x3_go<4, 4>(a, grid, 3, false, 0, st);
    else if (t.mt == 2 && t.nt == 4) rc = g1x3_go<2, 4>(a, grid, 3, false, 0, st);
    else if (t.mt == 4) rc = g1x3_go<4, 2>(a, grid, 3, false, 0, st);
    else rc = g1x3_go<2, 2>(a, grid, 3, false, 0, st);
    conv_prof_end(pe, st);
    if (rc != DC_OK) return rc;
    DC_CHECK_LAUNCH();
    return DC_OK;
}
static void g1x3_conv3s2_wplan(int B, int Ci, int Co, int Hi, int Wi, X3Tile& t, int& chunks, int& mtiles, int& ntiles, int& splits) {
    chunks = B * (Hi / 2) * (Wi / 2) / X3_KC;
    t = x3_wpick(Co, 9 * Ci);
    mtiles = ceil_div(Co, 32 * t.mt); ntiles = ceil_div(9 * Ci, 32 * t.nt);
    splits = g1x3_wsplits(mtiles * ntiles, chunks, t);
}
size_t g1x3_conv3s2_wgrad_ws(int B, int Ci, int Co, int Hi, int Wi) {
    X3Tile t; int chunks, mt, nt, splits;
    g1x3_conv3s2_wplan(B, Ci, Co, Hi, Wi, t, chunks, mt, nt, splits);
    return splits > 1 ? (size_t)splits * Co * 9 * Ci * sizeof(float) : 16;
}
bool g1x3_conv3s2_wgrad_ok(int B, int Ci, int Co, int Hi, int Wi) {
    return g1x3_conv3s2_ok(B, Ci, Co, Hi, Wi) && ((size_t)B * (Hi / 2) * (Wi / 2)) % 32 == 0;
}
int g1x3_conv3s2_wgrad(const float* x, const float* gy, float* dweight, void* ws, int B, int Ci, int Co, int Hi, int Wi, hipStream_t st) {
    if (((size_t)x & 15) || ((size_t)gy & 15) || ((size_t)ws & 15) || ((size_t)dweight & 15)) return DC_EINVAL;
    G1x3WArgs a{};
    a.gy = gy; a.x = x; a.B = B; a.Co = Co; a.Ci = 9 * Ci; a.Hi = Hi; a.Wi = Wi; a.Ho = Hi / 2; a.Wo = Wi / 2;
    X3Tile t;
    g1x3_conv3s2_wplan(B, Ci, Co, Hi, Wi, t, a.chunks, a.mtiles, a.ntiles, a.splits);
    a.out = a.splits > 1 ? (float*)ws : dweight;
    const dim3 grid(a.mtiles * a.ntiles, a.splits);
    const double npx = (double)B * a.Ho * a.Wo;
    hipEvent_t pe = conv_prof_begin(7, 2.0 * npx * Co * 9.0 * Ci, 12.0 * npx * (double)a.mtiles * (32.0 * t.mt) * (double)a.ntiles * (32.0 * t.nt),
                                    4.0 * ((double)B * Ci * Hi * Wi + npx * Co + 9.0 * Co * Ci), st);
    int rc;
    if (t.mt == 4 && t.nt == 4) rc = g1x3_wgo<4, 4>(a, grid, 3, false, st);
    else if (t.mt == 2 && t.nt == 4) rc = g1x3_wgo<2, 4>(a, grid, 3, false, st);
    else if (t.mt == 4) rc = g1x3_wgo<4, 2>(a, grid, 3, false, st);
    else rc = g1x3_wgo<2, 2>(a, grid, 3, false, st);
    conv_prof_end(pe, st);
    if (rc != DC_OK) return rc;
    DC_CHECK_LAUNCH();
    if (a.splits > 1) {
        const int n4 = Co * 9 * Ci / 4;
        hipLaunchKernelGGL(g1x3_slabsum_kernel, dim3(ceil_div(n4, 16)), dim3(256), 0, st, (const x3f4*)ws, (x3f4*)dweight, a.splits, n4);
        DC_CHECK_LAUNCH();
    }
    return DC_OK;
}
}  // namespace dc
