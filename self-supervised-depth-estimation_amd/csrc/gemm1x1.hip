// 1x1 convolutions as tiled fp32-MFMA GEMMs straight on the NCHW tensors: the Bottleneck conv1 / conv3 and `downsample`
// convolutions of the ResNet-50+ trunks (reference networks/resnet_encoder.py:70-98 via torchvision: about half of the
// trunk's multiplies at BASELINE configs[2]) and the 1x1 convolutions of the pose decoder (networks/pose_decoder.py:25,30).
//
//   forward        y[b,m,p]          = act( sum_k w[m,k] * x[b,k,s*py,s*px] + bias[m] )
//   data gradient  dx[b,k,s*py,s*px] = sum_m w[m,k] * gy[b,m,p]        (0 at the positions the stride skips)
//   weight grad    dw[m,k]           = sum_{b,p} gy[b,m,p] * x[b,k,s*py,s*px]    split over blocks, fixed-order reduce
//
// One block = 4 waves (2 x 2) = (32 MT) x (32 NT) outputs, MT, NT in {2, 4}; a wave keeps MT x NT accumulator tiles of
// v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulation).  The pixel dimension is flattened over the batch
// (n = b * P + p), so small maps (10 x 32 at layer4) still fill 128-wide tiles.
//
// No operand is ever transposed, neither in HBM nor on the way into LDS: both operands are copied global -> registers
// -> LDS as 16-byte vectors in the layout they have in memory, and the *reading* pattern adapts:
//   * an operand whose GEMM row/column index is contiguous in memory (x and gy as B operand: pixels; w as A operand of the
//     data gradient: input channels) lives in LDS as [reduction][index]; a lane reads MT (NT) consecutive floats with one
//     ds_read_b128 / b64 and uses them for its MT (NT) tiles -- tile t of lane i covers index i * MT + t;
//   * an operand whose *reduction* index is contiguous in memory (w as A operand of the forward: input channels; gy and x
//     in the weight gradient: pixels) lives in LDS as [index][reduction (+4 pad)]; a lane reads two consecutive reduction
//     elements with one ds_read_b64 and uses them in two successive MFMA steps.
// Both patterns are bank-conflict free (strides chosen per MI355X_MICROARCH.md's LDS table: b128 rows = 0 mod 64 dwords;
// index-major rows = 2 mod 32 dwords, which serves both ds_read_b64 and the ds_read2_b64 pairs the compiler forms from
// them -- 16 lanes x 2 dwords tile the 32 banks; rows of 4 * odd dwords measured 60 % conflict cycles under read2).
// The reduction order inside a chunk of 8 is permuted identically for A and B
// (element 2 k' + s of the chunk goes to MFMA step s, k-lane k'), which changes nothing in the sum.
//
// Main loop: reduction chunks of 32 (128 MFMAs per wave between barriers at 128 x 128), LDS double-buffered with ONE
// barrier per chunk; the global loads of chunk c+1 are issued before the MFMAs of chunk c and committed to LDS after
// them.  The loop has no bounds branches: the reduction extent is a multiple of the chunk (checked on the host, other
// shapes take the general kernels of pointwise.hip), and rows / columns past the edge of the matrix are CLAMPED to the
// last valid row / pixel group -- they compute garbage that is never stored.
// Block order: m-tiles of one pixel tile are adjacent and, through xcd_logical_block, on the same XCD: the activation
// tile is fetched from HBM once per XCD pass and re-served by that XCD's L2; the weights are L2-resident.
#include "dc_common.h"
#include "gemm1x1.h"
#include "wino.h"
#include "gemm_tiles.h"

#include <algorithm>

namespace dc {

struct G1Args {
    const float* w;       // (Co, Ci)
    const float* x;       // (B, Ci, Hi, Wi)
    const float* gy;      // (B, Co, Ho, Wo)           (backward)
    const float* bias;    // (Co) or null               (forward)
    const float* addend;  // (B, Ci, Hi, Wi) or null: added to dx in the epilogue (stride 1; the other gradient of a residual fork)
    const float* addend2; // a second one (a feature map's third consumer: the decoder's skip connection), or null
    float* out;           // y / dx / slab-or-dw
    int B, Co, Ci, Hi, Wi, Ho, Wo, s;
    int act;              // forward epilogue
    int mtiles, ntiles;   // tile grid
    int splits, chunks;   // weight gradient: reduction chunks in total, blocks along the reduction
    int xcd;              // weight gradient: tiles of one split on one XCD (see g1_wgrad_kernel)
    // ---- BatchNorm folded into the convolution (include/depthcore.h: dc_bn_fold) ----
    const float* in_scale;   // (groups, Ci): the B operand is relu(scale x + shift), applied on the way into LDS (forward, weight
    const float* in_shift;   //  gradient); in the data gradient the same pair re-derives the ReLU decision of the epilogue
    int npg;                 // images per BatchNorm group
    float* stat_part;        // forward epilogue: (Co, stat_nparts) x {sum, sum of squares} of the output, one partial per wave column
    int stat_nparts;
    const float* bn_x;       // data-gradient epilogue: the raw input of the BatchNorm whose (ReLU-ed) output this convolution read,
    const float* bn_mean;    //  its batch mean (groups, Ci), its ReLU decisions as bits (or null: scale x + shift > 0), and the
    const unsigned long long* bn_mask;   // partials (Ci, bwd_nparts) x {sum g', sum g' (x - mean)} of the MASKED result g'
    float* bwd_part;
    int bwd_nparts;
};

// sum over the 16 lanes of a DPP row (lanes sharing lane >> 4), result in every lane of the row: four row_ror steps
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
    return v;
}
__device__ __forceinline__ gf4 bn_relu4(gf4 v, float sc, float sh) {
    return gf4{fmaxf(fmaf(v.x, sc, sh), 0.f), fmaxf(fmaf(v.y, sc, sh), 0.f), fmaxf(fmaf(v.z, sc, sh), 0.f), fmaxf(fmaf(v.w, sc, sh), 0.f)};
}

// ---- pixel addressing: element offset of (b, channel 0, first pixel) of a 4-pixel group of the flattened (b, p) dimension.
// n is clamped to the last valid group (P % 4 == 0; stride 2: Wo % 4 == 0, so a group lies in one row).
__device__ __forceinline__ size_t pix_group_off(int n, int N, int P, int Wo, int C, int Hi, int Wi, int s) {
    const int nn = min(n, N - 4);
    const int b = nn / P, p = nn - b * P;
    if (s == 1) return (size_t)b * C * P + p;
    const int py = p / Wo, px = p - py * Wo;
    return (size_t)b * C * Hi * Wi + (size_t)(py * s) * Wi + px * s;
}
// 4 output pixels of one channel plane (stride-2: every other element of 8 consecutive ones)
// (the stride is a TEMPLATE parameter: as a run-time `if` it put a branch around every B-operand load, and the compiler
// waited for all outstanding loads -- vmcnt(0) -- right behind each, i.e. before the MFMAs the loads were meant to overlap)
template <int S>
__device__ __forceinline__ gf4 load_pix4(const float* plane_ptr) {
    if constexpr (S == 1) {
        return *reinterpret_cast<const gf4*>(plane_ptr);
    } else {
        const gf4 u = *reinterpret_cast<const gf4*>(plane_ptr), v = *reinterpret_cast<const gf4*>(plane_ptr + 4);
        return gf4{u.x, u.z, v.x, v.z};
    }
}

// =====================================================================================================================
// forward.  A = w [co][ci] (reduction-contiguous), B = x [ci][n] (index-contiguous).  Ci % KC == 0.
// =====================================================================================================================
template <int MT, int NT, bool EPI, int S, bool BNIN = false>
__global__ __launch_bounds__(256) void g1_fwd_kernel(G1Args a) {
    constexpr int BM = 32 * MT, BN = 32 * NT, KC = GKC, SB = IdxStride<NT, BN>::v;
    constexpr int NA = BM * KC / 1024, NB = KC * BN / 1024;          // float4 per thread per chunk
    constexpr int ASZ = BM * (KC + RP), BSZ = KC * SB;
    float* const As = g1_smem;                 // [2][ASZ]
    float* const Bs = g1_smem + 2 * ASZ;       // [2][BSZ]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int lb = xcd_logical_block(blockIdx.x, gridDim.x);
    const int m0 = (lb % a.mtiles) * BM, n0 = (lb / a.mtiles) * BN;
    const int P = a.Ho * a.Wo, N = a.B * P;
    const size_t plane = (size_t)a.Hi * a.Wi;

    // staging roles: global source pointers (advance by the chunk) and LDS destinations, fixed for the whole loop
    const float* asrc[NA];
    int adst[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int idx = tid + j * 256, row = idx / (KC / 4), kq = idx % (KC / 4);
        asrc[j] = a.w + (size_t)min(m0 + row, a.Co - 1) * a.Ci + kq * 4;
        adst[j] = row * (KC + RP) + kq * 4;
    }
    const float* bsrc[NB];
    int bdst[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int idx = tid + j * 256, k = idx / (BN / 4), c4 = idx % (BN / 4);
        bsrc[j] = a.x + pix_group_off(n0 + c4 * 4, N, P, a.Wo, a.Ci, a.Hi, a.Wi, S) + (size_t)k * plane;
        bdst[j] = k * SB + c4 * 4;
    }
    // BNIN: the input is relu(scale[g, ci] x + shift[g, ci]) -- the BatchNorm + ReLU in front of this convolution, applied to
    // the 16-byte pieces between their global load and their LDS store; (group, channel) table offset of each piece:
    int btab[BNIN ? NB : 1];
    float bsc[BNIN ? NB : 1], bsh[BNIN ? NB : 1];
    if constexpr (BNIN) {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int idx = tid + j * 256, k = idx / (BN / 4), c4 = idx % (BN / 4);
            btab[j] = (min(n0 + c4 * 4, N - 4) / P / a.npg) * a.Ci + k;
        }
    }
    gf4 ra[NA], rb[NB];
    auto gload = [&](int k0) {
#pragma unroll
        for (int j = 0; j < NA; ++j) ra[j] = *reinterpret_cast<const gf4*>(asrc[j] + k0);
#pragma unroll
        for (int j = 0; j < NB; ++j) rb[j] = load_pix4<S>(bsrc[j] + (size_t)k0 * plane);
        if constexpr (BNIN) {
#pragma unroll
            for (int j = 0; j < NB; ++j) { bsc[j] = a.in_scale[btab[j] + k0]; bsh[j] = a.in_shift[btab[j] + k0]; }
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NA; ++j) store_red4(As + buf * ASZ + adst[j], ra[j]);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            if constexpr (BNIN) rb[j] = bn_relu4(rb[j], bsc[j], bsh[j]);
            *reinterpret_cast<gf4*>(Bs + buf * BSZ + bdst[j]) = rb[j];
        }
    };

    gf4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = gf4{0, 0, 0, 0};

    const int nchunk = a.Ci / KC;
    gload(0);
    commit(0);
    __syncthreads();
    for (int c = 0; c < nchunk; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunk) gload((c + 1) * KC);
        mma_chunk32<MT, NT>([&](int q, float (&v)[4][2]) { read_red<MT, KC>(As + buf * ASZ, wm * 16 * MT, q, lane, v); },
                            [&](int q, float (&v)[4][2]) { read_idx<NT, SB>(Bs + buf * BSZ, wn * 16 * NT, q, lane, v); }, acc);
        if (c + 1 < nchunk) commit(buf ^ 1);
        __syncthreads();
    }
    // D: row r' = (lane >> 4) * 4 + r of tile mt <-> m = m0 + wm*16MT + mt*16 + r';  col j = lane & 15 of tile nt <-> n = .. + j*NT + nt
    const int j = lane & 15;
    const int n = n0 + wn * 16 * NT + j * NT;
    const bool nok = n < N;
    const int nn = nok ? n : 0;
    const int b = nn / P, p = nn - b * P;
    // statistics epilogue (a.stat_part, wave-uniform): {sum, sum of squares} of this wave column's 16 NT pixels per output
    // channel -- the 16 lanes of a DPP row hold one channel's pixels -- as partial number (pixel tile, wn)
    const bool stats = a.stat_part != nullptr;
    const int slot = (lb / a.mtiles) * 2 + wn;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + wm * 16 * MT + mt * 16 + (lane >> 4) * 4 + r;
            const bool mok = m < a.Co;
            float v[4];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) v[nt] = acc[mt][nt][r];
            if constexpr (EPI) {
                const float bsv = (a.bias && mok) ? a.bias[m] : 0.f;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) v[nt] = act_fwd(v[nt] + bsv, a.act);
            }
            if (nok && mok) {
                float* dst = a.out + ((size_t)b * a.Co + m) * P + p;
                if constexpr (NT == 4) *reinterpret_cast<gf4*>(dst) = gf4{v[0], v[1], v[2], v[3]};
                else *reinterpret_cast<gf2*>(dst) = gf2{v[0], v[1]};
            }
            if (stats) {
                float sv = 0.f, qv = 0.f;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) { sv += v[nt]; qv = fmaf(v[nt], v[nt], qv); }
                if (!nok) { sv = 0.f; qv = 0.f; }
                sv = row16_sum(sv); qv = row16_sum(qv);
                if (j == 0 && mok) *reinterpret_cast<gf2*>(a.stat_part + ((size_t)m * a.stat_nparts + slot) * 2) = gf2{sv, qv};
            }
        }
}

// =====================================================================================================================
// data gradient.  rows = input channels ci, reduction = co.  A = w [co][ci] (index-contiguous), B = gy [co][n] (index-cont.)
// Co % KC == 0.
// =====================================================================================================================
// SD: 1 = stride 1; 2 = stride 2; 3 = stride 2 with addends (its cell-block staging costs registers: its own instantiation)
template <int MT, int NT, int BNE = 0, int SD = 1>
__global__ __launch_bounds__(256) void g1_dgrad_kernel(G1Args a) {
    static_assert(BNE == 0 || SD == 1, "the BatchNorm epilogues exist at stride 1");
    constexpr int BM = 32 * MT, BN = 32 * NT, KC = GKC, SA = IdxStride<MT, BM>::v, SB = IdxStride<NT, BN>::v;
    constexpr int NA = KC * BM / 1024, NB = KC * BN / 1024;
    constexpr int ASZ = KC * SA, BSZ = KC * SB;
    float* const As = g1_smem;
    float* const Bs = g1_smem + 2 * ASZ;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int lb = xcd_logical_block(blockIdx.x, gridDim.x);
    const int m0 = (lb % a.mtiles) * BM, n0 = (lb / a.mtiles) * BN;
    const int P = a.Ho * a.Wo, N = a.B * P;

    const float* asrc[NA];
    const float* bsrc[NB];
    int adst[NA], bdst[NB];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int idx = tid + j * 256, k = idx / (BM / 4), c4 = idx % (BM / 4);
        asrc[j] = a.w + (size_t)k * a.Ci + min(m0 + c4 * 4, a.Ci - 4);
        adst[j] = k * SA + c4 * 4;
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int idx = tid + j * 256, k = idx / (BN / 4), c4 = idx % (BN / 4);
        bsrc[j] = a.gy + pix_group_off(n0 + c4 * 4, N, P, a.Wo, a.Co, a.Ho, a.Wo, 1) + (size_t)k * P;
        bdst[j] = k * SB + c4 * 4;
    }
    gf4 ra[NA], rb[NB];
    auto gload = [&](int k0) {
#pragma unroll
        for (int j = 0; j < NA; ++j) ra[j] = *reinterpret_cast<const gf4*>(asrc[j] + (size_t)k0 * a.Ci);
#pragma unroll
        for (int j = 0; j < NB; ++j) rb[j] = *reinterpret_cast<const gf4*>(bsrc[j] + (size_t)k0 * P);
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NA; ++j) *reinterpret_cast<gf4*>(As + buf * ASZ + adst[j]) = ra[j];
#pragma unroll
        for (int j = 0; j < NB; ++j) *reinterpret_cast<gf4*>(Bs + buf * BSZ + bdst[j]) = rb[j];
    };
    gf4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = gf4{0, 0, 0, 0};
    const int nchunk = a.Co / KC;
    gload(0);
    commit(0);
    __syncthreads();
    for (int c = 0; c < nchunk; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunk) gload((c + 1) * KC);
        mma_chunk32<MT, NT>([&](int q, float (&v)[4][2]) { read_idx<MT, SA>(As + buf * ASZ, wm * 16 * MT, q, lane, v); },
                            [&](int q, float (&v)[4][2]) { read_idx<NT, SB>(Bs + buf * BSZ, wn * 16 * NT, q, lane, v); }, acc);
        if (c + 1 < nchunk) commit(buf ^ 1);
        __syncthreads();
    }
    // row r' of tile mt <-> ci = m0 + wm*16MT + r'*MT + mt;  col j of tile nt <-> n = .. + j*NT + nt
    const int j = lane & 15;
    const int n = n0 + wn * 16 * NT + j * NT;
    if constexpr (BNE != 0) {
        // BatchNorm-backward epilogue (stride 1).  This convolution read a = relu(bn(x) [+ res]); its data gradient
        // dL/da [+ the skip's gradient] is masked with the ReLU decision HERE -- BNE 1: scale x + shift > 0 re-derived from the
        // BatchNorm's raw input x; BNE 2: the forward's bit mask (decisions that involved a residual) -- stored as g', and the
        // wave column's partial {sum g', sum g' (x - mean)} per channel goes to a.bwd_part.  What used to be a streaming
        // statistics pass over g and x reads x once, in tiles that are still warm in this XCD's L2 from the forward's order.
        const bool nok = n < N;
        const int nn = nok ? n : 0;
        const int b = nn / P, p = nn - b * P;
        const size_t plane = (size_t)a.Hi * a.Wi;
        const int grp = b / a.npg;
        const int slot = (lb / a.mtiles) * 2 + wn;
        const int pblk = (P + 255) >> 8;
        constexpr bool ADD = BNE == 2;      // (BNE 1 = a folded BatchNorm + ReLU: single consumer, never a residual fork's conv1)
        gf4 xv[2][4], ad[ADD ? 2 : 1][4];
        float rmean[2][4], rsc[2][4], rsh[2][4];
        unsigned bits[2][4];
        auto eload = [&](int mt, int h) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ci = min(m0 + wm * 16 * MT + ((lane >> 4) * 4 + r) * MT + mt, a.Ci - 1);
                const size_t o = ((size_t)b * a.Ci + ci) * plane + p;
                const int tab = grp * a.Ci + ci;
                rmean[h][r] = a.bn_mean[tab];
                if constexpr (BNE == 1) { rsc[h][r] = a.in_scale[tab]; rsh[h][r] = a.in_shift[tab]; }
                if constexpr (NT == 4) {
                    xv[h][r] = *reinterpret_cast<const gf4*>(a.bn_x + o);
                    if constexpr (ADD) ad[h][r] = a.addend ? *reinterpret_cast<const gf4*>(a.addend + o) : gf4{0.f, 0.f, 0.f, 0.f};
                } else {
                    const gf2 t = *reinterpret_cast<const gf2*>(a.bn_x + o);
                    xv[h][r] = gf4{t.x, t.y, 0.f, 0.f};
                    if constexpr (ADD) {
                        const gf2 u = a.addend ? *reinterpret_cast<const gf2*>(a.addend + o) : gf2{0.f, 0.f};
                        ad[h][r] = gf4{u.x, u.y, 0.f, 0.f};
                    }
                }
                if constexpr (BNE == 2) {
                    // bit l of word w of a 256-element block <-> element 4 l + w (bn_apply_kernel's wave ballots)
                    // (only the 32-bit half that holds bit l of each word is fetched)
                    const int l = (p & 255) >> 2, w0 = p & 3;
                    const unsigned* mw = reinterpret_cast<const unsigned*>(a.bn_mask + (((size_t)b * a.Ci + ci) * pblk + (p >> 8)) * 4) + (l >> 5);
                    unsigned bt = 0;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) bt |= ((mw[2 * (w0 + nt)] >> (l & 31)) & 1u) << nt;
                    bits[h][r] = bt;
                }
            }
        };
        eload(0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int h = mt & 1;
            if (mt + 1 < MT) eload(mt + 1, h ^ 1);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ci = m0 + wm * 16 * MT + ((lane >> 4) * 4 + r) * MT + mt;
                float g[4], sv = 0.f, qv = 0.f;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const float x = xv[h][r][nt];
                    bool keep;
                    if constexpr (BNE == 1) keep = fmaf(x, rsc[h][r], rsh[h][r]) > 0.f;
                    else keep = (bits[h][r] >> nt) & 1u;
                    float gv = acc[mt][nt][r];
                    if constexpr (ADD) gv += ad[h][r][nt];
                    g[nt] = keep ? gv : 0.f;
                    sv += g[nt]; qv = fmaf(g[nt], x - rmean[h][r], qv);
                }
                if (!nok) { sv = 0.f; qv = 0.f; }
                sv = row16_sum(sv); qv = row16_sum(qv);
                if (ci < a.Ci) {
                    if (j == 0) *reinterpret_cast<gf2*>(a.bwd_part + ((size_t)ci * a.bwd_nparts + slot) * 2) = gf2{sv, qv};
                    if (nok) {
                        float* dst = a.out + ((size_t)b * a.Ci + ci) * plane + p;
                        if constexpr (NT == 4) *reinterpret_cast<gf4*>(dst) = gf4{g[0], g[1], g[2], g[3]};
                        else *reinterpret_cast<gf2*>(dst) = gf2{g[0], g[1]};
                    }
                }
            }
        }
        return;
    }
    if (n >= N) return;
    const int b = n / P, p = n - b * P;
    const size_t plane = (size_t)a.Hi * a.Wi;
    size_t pix;
    if (a.s == 1) {
        pix = p;
    } else {
        const int py = p / a.Wo, px = p - py * a.Wo;
        pix = (size_t)(py * 2) * a.Wi + px * 2;
    }
    if constexpr (SD == 1) {
        // the other gradient(s) of a residual fork: ALL of this lane's addend values are loaded before the first store -- a load
        // next to its store in the loop below waited for itself 4 MT times over (121 -> 167 us at 128 x 128)
#pragma unroll 1
        for (int which = 0; which < 2; ++which) {
            const float* addp = which == 0 ? a.addend : a.addend2;
            if (!addp) continue;
            gf4 ad[MT][4];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ci = min(m0 + wm * 16 * MT + ((lane >> 4) * 4 + r) * MT + mt, a.Ci - 1);
                    const float* src = addp + ((size_t)b * a.Ci + ci) * plane + pix;
                    if constexpr (NT == 4) ad[mt][r] = *reinterpret_cast<const gf4*>(src);
                    else { const gf2 t = *reinterpret_cast<const gf2*>(src); ad[mt][r] = gf4{t.x, t.y, 0.f, 0.f}; }
                }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc[mt][0][r] += ad[mt][r].x; acc[mt][1][r] += ad[mt][r].y;
                    if constexpr (NT == 4) { acc[mt][2][r] += ad[mt][r].z; acc[mt][3][r] += ad[mt][r].w; }
                }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ci = m0 + wm * 16 * MT + ((lane >> 4) * 4 + r) * MT + mt;
                if (ci >= a.Ci) continue;
                float* dst = a.out + ((size_t)b * a.Ci + ci) * plane + pix;
                if constexpr (NT == 4)
                    *reinterpret_cast<gf4*>(dst) = gf4{acc[mt][0][r], acc[mt][1][r], acc[mt][2][r], acc[mt][3][r]};
                else
                    *reinterpret_cast<gf2*>(dst) = gf2{acc[mt][0][r], acc[mt][1][r]};
            }
        return;
    } else {
    // stride 2: the 2 x (2 NT) cell block of these NT output pixels -- the values and the zeros the stride skips -- plus, where the
    // input has other consumers (the 3x3 / 2 convolution next to a `downsample` branch, the decoder's skip connection), their
    // gradients: every cell of the block is written exactly once, so the sum autograd would form in passes of its own rides here.
    // One 16-channel tile (four rows per lane) at a time: its addend cells are all requested before the first store.
    constexpr int CV = NT / 2;                // 16-byte vectors per cell row
    constexpr bool has_add = SD == 3;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        gf4 c0[4][CV], c1[4][CV];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int v = 0; v < CV; ++v) { c0[r][v] = gf4{0, 0, 0, 0}; c1[r][v] = gf4{0, 0, 0, 0}; }
        if constexpr (has_add) {
#pragma unroll 1
            for (int which = 0; which < 2; ++which) {
                const float* addp = which == 0 ? a.addend : a.addend2;
                if (!addp) continue;
                gf4 t0[4][CV], t1[4][CV];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ci = min(m0 + wm * 16 * MT + ((lane >> 4) * 4 + r) * MT + mt, a.Ci - 1);
                    const float* src = addp + ((size_t)b * a.Ci + ci) * plane + pix;
#pragma unroll
                    for (int v = 0; v < CV; ++v) {
                        t0[r][v] = *reinterpret_cast<const gf4*>(src + 4 * v);
                        t1[r][v] = *reinterpret_cast<const gf4*>(src + a.Wi + 4 * v);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int v = 0; v < CV; ++v) { c0[r][v] += t0[r][v]; c1[r][v] += t1[r][v]; }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ci = m0 + wm * 16 * MT + ((lane >> 4) * 4 + r) * MT + mt;
            if (ci >= a.Ci) continue;
            float* dst = a.out + ((size_t)b * a.Ci + ci) * plane + pix;
#pragma unroll
            for (int v = 0; v < CV; ++v) {
                gf4 o = c0[r][v];
                o.x += acc[mt][2 * v][r]; o.z += acc[mt][2 * v + 1][r];
                *reinterpret_cast<gf4*>(dst + 4 * v) = o;
                *reinterpret_cast<gf4*>(dst + a.Wi + 4 * v) = c1[r][v];
            }
        }
    }
    }
}

// =====================================================================================================================
// weight gradient.  rows = co, cols = ci, reduction = flattened pixels n (N % KC == 0).  A = gy [co][n], B = x [ci][n]: both
// reduction-contiguous.  blockIdx.y = split: a contiguous range of chunks (its successive 128-byte row segments stay in one
// L2); writes slab[split][co][ci] (or dw itself when there is one split).
// =====================================================================================================================
template <int MT, int NT, int S, bool BNIN = false>
__global__ __launch_bounds__(256) void g1_wgrad_kernel(G1Args a) {
    constexpr int BM = 32 * MT, BN = 32 * NT, KC = GKC;
    constexpr int NA = BM * KC / 1024, NB = BN * KC / 1024;
    constexpr int ASZ = BM * (KC + RP), BSZ = BN * (KC + RP);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    // All tiles of one split read the same pixel range of gy and x: give them adjacent logical indices, i.e. the same XCD
    // and neighbouring dispatch slots, so that range comes from HBM once and from that XCD's L2 afterwards (the hardware
    // order deals the tiles of a split round-robin over the 8 XCDs: every L2 fetched every range; 26 resnet50 shapes,
    // tools/sweep_g1wgrad.py: -2 % in total, -1.5 ... -4 % on 22 of them).
    int tile = blockIdx.x, split = blockIdx.y;
    if (a.xcd) {
        const int tiles = gridDim.x, lb = xcd_logical_block(blockIdx.y * tiles + blockIdx.x, tiles * gridDim.y);
        split = lb / tiles; tile = lb - split * tiles;
    }
    const int m0 = (tile % a.mtiles) * BM, c0 = (tile / a.mtiles) * BN;
    const int P = a.Ho * a.Wo;
    const size_t plane = (size_t)a.Hi * a.Wi;
    const int kq = tid % (KC / 4), row0 = tid / (KC / 4);    // the same 4-pixel group for all of a thread's loads
    // rows past the matrix edge are clamped (their products are never stored)
    size_t arow[NA], brow[NB];
#pragma unroll
    for (int j = 0; j < NA; ++j) arow[j] = (size_t)min(m0 + row0 + j * (1024 / KC), a.Co - 1) * P;
#pragma unroll
    for (int j = 0; j < NB; ++j) brow[j] = (size_t)min(c0 + row0 + j * (1024 / KC), a.Ci - 1) * plane;
    // BNIN: x is the raw input of a BatchNorm + ReLU that was folded into this convolution's loader in the forward; the
    // weight gradient needs the same relu(scale x + shift), re-formed between the global load and the LDS store
    float bsc[BNIN ? NB : 1], bsh[BNIN ? NB : 1];
    gf4 ra[NA], rb[NB];
    auto gload = [&](int ch) {           // chunk ch = KC consecutive flattened pixels; a 4-pixel group lies in one image row
        const int n = ch * KC + kq * 4;
        const int b = n / P, p = n - b * P;
        if constexpr (BNIN) {
            const int tab = (b / a.npg) * a.Ci;
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int cch = min(c0 + row0 + j * (1024 / KC), a.Ci - 1);
                bsc[j] = a.in_scale[tab + cch]; bsh[j] = a.in_shift[tab + cch];
            }
        }
        size_t pix;
        if constexpr (S == 1) {
            pix = p;
        } else {
            const int py = p / a.Wo, px = p - py * a.Wo;
            pix = (size_t)(py * 2) * a.Wi + px * 2;
        }
        const float* ga = a.gy + (size_t)b * a.Co * P + p;
        const float* xb = a.x + (size_t)b * a.Ci * plane + pix;
#pragma unroll
        for (int j = 0; j < NA; ++j) ra[j] = *reinterpret_cast<const gf4*>(ga + arow[j]);
#pragma unroll
        for (int j = 0; j < NB; ++j) rb[j] = load_pix4<S>(xb + brow[j]);
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NA; ++j)
            store_red4(g1_smem + buf * ASZ + (row0 + j * (1024 / KC)) * (KC + RP) + kq * 4, ra[j]);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            if constexpr (BNIN) rb[j] = bn_relu4(rb[j], bsc[j], bsh[j]);
            store_red4(g1_smem + 2 * ASZ + buf * BSZ + (row0 + j * (1024 / KC)) * (KC + RP) + kq * 4, rb[j]);
        }
    };
    gf4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = gf4{0, 0, 0, 0};
    const int per = (a.chunks + a.splits - 1) / a.splits;
    const int ch0 = split * per, ch1 = min(ch0 + per, a.chunks);
    if (ch0 < ch1) {
        gload(ch0);
        commit(0);
    }
    __syncthreads();
    for (int ch = ch0; ch < ch1; ++ch) {
        const int buf = (ch - ch0) & 1;
        const bool more = ch + 1 < ch1;
        if (more) gload(ch + 1);
        mma_chunk32<MT, NT>([&](int q, float (&v)[4][2]) { read_red<MT, KC>(g1_smem + buf * ASZ, wm * 16 * MT, q, lane, v); },
                            [&](int q, float (&v)[4][2]) { read_red<NT, KC>(g1_smem + 2 * ASZ + buf * BSZ, wn * 16 * NT, q, lane, v); }, acc);
        if (more) commit(buf ^ 1);
        __syncthreads();
    }
    float* slab = a.out + (size_t)split * a.Co * a.Ci;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * 16 * MT + mt * 16 + (lane >> 4) * 4 + r;
                const int ci = c0 + wn * 16 * NT + nt * 16 + (lane & 15);
                if (m < a.Co && ci < a.Ci) slab[(size_t)m * a.Ci + ci] = acc[mt][nt][r];
            }
}

// ---- bias + activation backward of the 1x1 convolutions that have them (pose decoder): g' = gy * act'(y), dbias = sum g'
// one block per channel; deterministic (fixed-order tree)
__global__ __launch_bounds__(256) void g1_bias_act_bwd_kernel(const float* __restrict__ y, const float* __restrict__ gy,
                                                             float* __restrict__ gpre, float* __restrict__ dbias, int B, int C, int P,
                                                             int act) {
    __shared__ float sm[256];
    const int c = blockIdx.x;
    float s = 0.f;
    for (int i = threadIdx.x; i < B * P; i += 256) {
        const int b = i / P, p = i - b * P;
        const size_t o = ((size_t)b * C + c) * P + p;
        const float g = gy[o] * act_bwd(y[o], act);
        if (gpre) gpre[o] = g;
        s += g;
    }
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) sm[threadIdx.x] += sm[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0 && dbias) dbias[c] = sm[0];
}

// ---- tile choice ---------------------------------------------------------------------------------------------------
// Estimated time of a tile shape = rounds the grid needs on the chip x cost of one block, where the cost per block grows
// with its MFMA count and the smaller tiles pay more operand traffic per MFMA.  Blocks resident per CU: LDS-limited.
struct G1Tile { int mt, nt; };
static size_t g1_lds_fwd(G1Tile t) { return (size_t)2 * (32 * t.mt * (GKC + RP) + GKC * (t.nt == 4 ? 128 : 80)) * sizeof(float); }
static size_t g1_lds_dgrad(G1Tile t) { return (size_t)2 * GKC * ((t.mt == 4 ? 128 : 80) + (t.nt == 4 ? 128 : 80)) * sizeof(float); }
static size_t g1_lds_wgrad(G1Tile t) { return (size_t)2 * 32 * (t.mt + t.nt) * (GKC + RP) * sizeof(float); }
static G1Tile g1_pick(int M, int N, size_t (*lds)(G1Tile)) {
    if (const char* f = getenv("DC_G1_TILE")) {                 // experiments (tools/sweep_g1.py): "MT,NT"
        int mt = 0, nt = 0;
        if (sscanf(f, "%d,%d", &mt, &nt) == 2 && ((mt == 4 && nt == 4 && M > 64) || (mt == 2 && (nt == 4 || nt == 2)))) return {mt, nt};
    }
    const G1Tile cand[3] = {{4, 4}, {2, 4}, {2, 2}};
    const double penalty[3] = {1.0, 1.12, 1.3};          // relative cost per MFMA (operand re-reads, shorter MFMA runs)
    double best = 1e300;
    G1Tile pick = cand[2];
    for (int i = 0; i < 3; ++i) {
        if (cand[i].mt == 4 && M <= 64) continue;
        const long blocks = (long)ceil_div(M, 32 * cand[i].mt) * ceil_div(N, 32 * cand[i].nt);
        const int per_cu = std::max(1, std::min(4, (int)((size_t)(150 << 10) / lds(cand[i]))));
        const long slots = 256L * per_cu;
        const long rounds = (blocks + slots - 1) / slots;
        // a chip filled to less than half hides no latency: charge that more than proportionally
        const double fill = (double)blocks / (double)(rounds * slots);
        const double t = (double)rounds * per_cu * cand[i].mt * cand[i].nt * penalty[i] * (fill < 0.5 && rounds == 1 ? 0.5 + fill : 1.0);
        if (t < best) { best = t; pick = cand[i]; }
    }
    return pick;
}
// (128 x 128 tiles for every shape with both extents >= 128 -- half the operand traffic of the HBM-heavy 64 x 64 launches --
// were measured: 60 -> 80 us per launch in isolation, no change of the C3 step; the small tiles keep more blocks in flight.)
static G1Tile g1_wpick(int M, int K) {
    if (const char* f = getenv("DC_G1_WTILE")) { const int v = atoi(f); if (v == 4) return {4, 4}; if (v == 2) return {2, 2}; }      // experiments
    // 128 x 128 tiles wherever both dimensions fill one (half the operand re-reads per MAC), 64 x 64 otherwise; split
    // targets 512 / 1024 blocks (2 / 4 resident per CU).  Measured on the 26 resnet50 shapes of C3 (tools/sweep_g1wgrad.py):
    // 13 % less time in total than the round-2 rule (128 x 128 only from 32 tiles on, 768 blocks for both); in the two-stream
    // step that is +0.6 % at C3 and within noise at C2.
    if (M >= 128 && K >= 128) return {4, 4};
    return {2, 2};
}
static int g1_wsplits(int M, int K, int chunks, G1Tile t) {
    const int tiles = ceil_div(M, 32 * t.mt) * ceil_div(K, 32 * t.nt);
    int target = t.mt == 4 ? 512 : 1024;
    if (const char* f = getenv("DC_G1_WBLOCKS")) { const int v = atoi(f); if (v > 0) target = v; }      // experiments
    int s = std::max(1, std::min({chunks, ceil_div(target, tiles), 512}));   // (a floor on chunks per block measured slower, C2)
    return ceil_div(chunks, ceil_div(chunks, s));        // no empty split: every slab gets written
}

template <typename K>
static bool g1_set_lds(K kernel, size_t bytes) {
    return hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess;
}

}  // namespace dc

using namespace dc;

// ---- which shapes the tiled kernels take (the entry points of pointwise.hip ask per pass) ---------------------------------
static bool g1_common(int B, int Ci, int Co, int Hi, int Wi, int stride) {
    if (B <= 0 || Ci <= 0 || Co <= 0 || Hi <= 0 || Wi <= 0) return false;
    if (stride != 1 && stride != 2) return false;
    if (stride == 2 && ((Hi & 1) || (Wi & 1))) return false;
    const int Ho = Hi / stride, Wo = Wi / stride;
    if (stride == 2 && (Wo & 3)) return false;
    if ((Ho * Wo) & 3) return false;                    // 16-byte pixel groups never straddle images
    if (Ci & 3) return false;                           // 16-byte groups along the channels of w
    if ((size_t)B * std::max(Ci, Co) * Hi * Wi >= (1ull << 31)) return false;
    return true;
}
extern "C" int dc_gemm1x1_fwd_ok(int B, int Ci, int Co, int Hi, int Wi, int stride) {
    return g1_common(B, Ci, Co, Hi, Wi, stride) && Ci % GKC == 0;
}
extern "C" int dc_gemm1x1_dgrad_ok(int B, int Ci, int Co, int Hi, int Wi, int stride) {
    return g1_common(B, Ci, Co, Hi, Wi, stride) && Co % GKC == 0;
}
extern "C" int dc_gemm1x1_wgrad_ok(int B, int Ci, int Co, int Hi, int Wi, int stride) {
    return g1_common(B, Ci, Co, Hi, Wi, stride) && ((size_t)B * (Hi / stride) * (Wi / stride)) % GKC == 0;
}

static void g1_fill(G1Args& a, int B, int Ci, int Co, int Hi, int Wi, int stride) {
    a.B = B; a.Ci = Ci; a.Co = Co; a.Hi = Hi; a.Wi = Wi; a.Ho = Hi / stride; a.Wo = Wi / stride; a.s = stride;
}

// partials per channel of the forward's statistics epilogue / the data gradient's BatchNorm epilogue (0: not on this shape):
// one per wave column of 16 NT pixels; a group boundary must not fall inside one
static int g1_parts(int rows, int B, int P, int groups, size_t (*lds)(G1Tile), int* ppg) {
    if (groups < 1 || B % groups) return 0;
    const int N = B * P;
    const G1Tile t = g1_pick(rows, N, lds);
    const int cover = 16 * t.nt;
    if (groups > 1 && (N / groups) % cover) return 0;
    if (ppg) *ppg = (N / groups) / cover;
    return ceil_div(N, 32 * t.nt) * 2;
}
extern "C" int dc_gemm1x1_stat_parts(int B, int Ci, int Co, int Hi, int Wi, int stride, int groups, int* ppg) {
    if (!dc_gemm1x1_fwd_ok(B, Ci, Co, Hi, Wi, stride)) return 0;
    return g1_parts(Co, B, (Hi / stride) * (Wi / stride), groups, g1_lds_fwd, ppg);
}
extern "C" int dc_gemm1x1_bwd_parts(int B, int Ci, int Co, int Hi, int Wi, int stride, int groups, int* ppg) {
    if (stride != 1 || !dc_gemm1x1_dgrad_ok(B, Ci, Co, Hi, Wi, stride)) return 0;
    return g1_parts(Ci, B, Hi * Wi, groups, g1_lds_dgrad, ppg);
}

extern "C" int dc_gemm1x1_fwd(const float* x, const float* weight, const float* bias, float* y, int B, int Ci, int Co, int Hi, int Wi,
                              int stride, int act, const dc_bn_fold* bn, void* stream) {
    if (!x || !weight || !y || !dc_gemm1x1_fwd_ok(B, Ci, Co, Hi, Wi, stride) || act < 0 || act > ACT_LAST) return DC_EINVAL;
    G1Args a{};
    g1_fill(a, B, Ci, Co, Hi, Wi, stride);
    a.w = weight; a.x = x; a.bias = bias; a.out = y; a.act = act;
    const int N = B * a.Ho * a.Wo;
    const G1Tile t = g1_pick(Co, N, g1_lds_fwd);
    const bool bnin = bn && bn->in_scale;
    if (bn) {
        if (bn->groups < 1 || B % bn->groups) return DC_EINVAL;
        a.npg = B / bn->groups;
        if (bnin && (!bn->in_shift || stride != 1 || bias || act != ACT_NONE)) return DC_EINVAL;
        a.in_scale = bn->in_scale; a.in_shift = bn->in_shift;
        if (bn->stat_part) {
            a.stat_nparts = dc_gemm1x1_stat_parts(B, Ci, Co, Hi, Wi, stride, bn->groups, nullptr);
            if (!a.stat_nparts) return DC_EINVAL;
            a.stat_part = bn->stat_part;
        }
    }
    a.mtiles = ceil_div(Co, 32 * t.mt); a.ntiles = ceil_div(N, 32 * t.nt);
    const dim3 grid(a.mtiles * a.ntiles);
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = g1_lds_fwd(t);
    const bool epi = bias || act != ACT_NONE;
    static const bool attr = g1_set_lds(g1_fwd_kernel<4, 4, false, 1>, g1_lds_fwd({4, 4})) &&
                             g1_set_lds(g1_fwd_kernel<4, 4, true, 1>, g1_lds_fwd({4, 4})) &&
                             g1_set_lds(g1_fwd_kernel<4, 4, false, 2>, g1_lds_fwd({4, 4})) &&
                             g1_set_lds(g1_fwd_kernel<4, 4, true, 2>, g1_lds_fwd({4, 4})) &&
                             g1_set_lds(g1_fwd_kernel<4, 4, false, 1, true>, g1_lds_fwd({4, 4}));
    if (!attr) return DC_ELAUNCH;
    hipEvent_t pe = conv_prof_begin(4, 2.0 * (double)B * Co * Ci * a.Ho * a.Wo, 2.0 * (double)grid.x * (32.0 * t.mt) * (32.0 * t.nt) * Ci, 4.0 * ((double)B * Ci * a.Ho * a.Wo + (double)B * Co * a.Ho * a.Wo + (double)Co * Ci), st);
#define G1_FWD(MT, NT)                                                                            \
    do {                                                                                          \
        if (bnin) {                                                                                   \
            hipLaunchKernelGGL((g1_fwd_kernel<MT, NT, false, 1, true>), grid, dim3(256), lds, st, a);     \
        } else if (stride == 1) {                                                                     \
            if (epi) hipLaunchKernelGGL((g1_fwd_kernel<MT, NT, true, 1>), grid, dim3(256), lds, st, a);   \
            else hipLaunchKernelGGL((g1_fwd_kernel<MT, NT, false, 1>), grid, dim3(256), lds, st, a);      \
        } else {                                                                                      \
            if (epi) hipLaunchKernelGGL((g1_fwd_kernel<MT, NT, true, 2>), grid, dim3(256), lds, st, a);   \
            else hipLaunchKernelGGL((g1_fwd_kernel<MT, NT, false, 2>), grid, dim3(256), lds, st, a);      \
        }                                                                                             \
    } while (0)
    if (t.mt == 4) G1_FWD(4, 4);
    else if (t.nt == 4) G1_FWD(2, 4);
    else G1_FWD(2, 2);
#undef G1_FWD
    conv_prof_end(pe, st);
    DC_CHECK_LAUNCH();
    return DC_OK;
}

extern "C" int dc_gemm1x1_dgrad(const float* gy, const float* weight, float* dx, const float* addend, const float* addend2, int B, int Ci,
                                int Co, int Hi, int Wi, int stride, const dc_bn_fold* bn, void* stream) {
    if (!gy || !weight || !dx || !dc_gemm1x1_dgrad_ok(B, Ci, Co, Hi, Wi, stride)) return DC_EINVAL;
    if (addend2 && bn && bn->bwd_part) return DC_EINVAL;          // (the BatchNorm epilogues take one addend)
    G1Args a{};
    g1_fill(a, B, Ci, Co, Hi, Wi, stride);
    a.w = weight; a.gy = gy; a.out = dx; a.addend = addend; a.addend2 = addend2;
    int bne = 0;
    if (bn && bn->bwd_part) {
        a.bwd_nparts = dc_gemm1x1_bwd_parts(B, Ci, Co, Hi, Wi, stride, bn->groups, nullptr);
        if (!a.bwd_nparts || !bn->bn_x || !bn->bn_mean || (!bn->bn_mask && (!bn->in_scale || !bn->in_shift || addend))) return DC_EINVAL;
        if (bn->bn_mask && ((Hi * Wi) & 3)) return DC_EINVAL;
        a.npg = B / bn->groups;
        a.bn_x = bn->bn_x; a.bn_mean = bn->bn_mean; a.bn_mask = (const unsigned long long*)bn->bn_mask; a.bwd_part = bn->bwd_part;
        a.in_scale = bn->in_scale; a.in_shift = bn->in_shift;
        bne = bn->bn_mask ? 2 : 1;
    }
    const int N = B * a.Ho * a.Wo;
    const G1Tile t = g1_pick(Ci, N, g1_lds_dgrad);
    a.mtiles = ceil_div(Ci, 32 * t.mt); a.ntiles = ceil_div(N, 32 * t.nt);
    const dim3 grid(a.mtiles * a.ntiles);
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = g1_lds_dgrad(t);
    static const bool attr = g1_set_lds(g1_dgrad_kernel<4, 4>, g1_lds_dgrad({4, 4})) && g1_set_lds(g1_dgrad_kernel<4, 4, 1>, g1_lds_dgrad({4, 4})) &&
                             g1_set_lds(g1_dgrad_kernel<4, 4, 2>, g1_lds_dgrad({4, 4})) && g1_set_lds(g1_dgrad_kernel<4, 4, 0, 2>, g1_lds_dgrad({4, 4})) &&
                             g1_set_lds(g1_dgrad_kernel<4, 4, 0, 3>, g1_lds_dgrad({4, 4}));
    if (!attr) return DC_ELAUNCH;
    hipEvent_t pe = conv_prof_begin(4, 2.0 * (double)B * Co * Ci * a.Ho * a.Wo, 2.0 * (double)grid.x * (32.0 * t.mt) * (32.0 * t.nt) * Co, 4.0 * ((double)B * Ci * a.Ho * a.Wo + (double)B * Co * a.Ho * a.Wo + (double)Co * Ci), st);
#define G1_DGRAD(BNE, SD)                                                                                  \
    do {                                                                                                   \
        if (t.mt == 4) hipLaunchKernelGGL((g1_dgrad_kernel<4, 4, BNE, SD>), grid, dim3(256), lds, st, a);      \
        else if (t.nt == 4) hipLaunchKernelGGL((g1_dgrad_kernel<2, 4, BNE, SD>), grid, dim3(256), lds, st, a); \
        else hipLaunchKernelGGL((g1_dgrad_kernel<2, 2, BNE, SD>), grid, dim3(256), lds, st, a);                \
    } while (0)
    if (stride == 2 && (addend || addend2)) G1_DGRAD(0, 3);
    else if (stride == 2) G1_DGRAD(0, 2);
    else if (bne == 0) G1_DGRAD(0, 1);
    else if (bne == 1) G1_DGRAD(1, 1);
    else G1_DGRAD(2, 1);
#undef G1_DGRAD
    conv_prof_end(pe, st);
    DC_CHECK_LAUNCH();
    return DC_OK;
}

extern "C" size_t dc_gemm1x1_wgrad_workspace(int B, int Ci, int Co, int Hi, int Wi, int stride) {
    if (!dc_gemm1x1_wgrad_ok(B, Ci, Co, Hi, Wi, stride)) return 0;
    const int chunks = B * (Hi / stride) * (Wi / stride) / GKC;
    const G1Tile t = g1_wpick(Co, Ci);
    const int splits = g1_wsplits(Co, Ci, chunks, t);
    return splits > 1 ? (size_t)splits * Co * Ci * sizeof(float) : 16;
}

extern "C" int dc_gemm1x1_wgrad(const float* x, const float* gy, float* dweight, void* ws, int B, int Ci, int Co, int Hi, int Wi,
                                int stride, const dc_bn_fold* bn, void* stream) {
    if (!x || !gy || !dweight || !ws || !dc_gemm1x1_wgrad_ok(B, Ci, Co, Hi, Wi, stride)) return DC_EINVAL;
    G1Args a{};
    g1_fill(a, B, Ci, Co, Hi, Wi, stride);
    a.x = x; a.gy = gy;
    const bool bnin = bn && bn->in_scale;
    if (bnin) {
        if (!bn->in_shift || stride != 1 || bn->groups < 1 || B % bn->groups) return DC_EINVAL;
        a.npg = B / bn->groups; a.in_scale = bn->in_scale; a.in_shift = bn->in_shift;
    }
    a.chunks = B * a.Ho * a.Wo / GKC;
    const G1Tile t = g1_wpick(Co, Ci);
    a.splits = g1_wsplits(Co, Ci, a.chunks, t);
    a.xcd = 1;
    if (const char* f = getenv("DC_G1_WXCD")) a.xcd = atoi(f);          // experiments (tools/sweep_g1wgrad.py): 0 = hardware block order
    a.out = a.splits > 1 ? (float*)ws : dweight;
    a.mtiles = ceil_div(Co, 32 * t.mt); a.ntiles = ceil_div(Ci, 32 * t.nt);
    const dim3 grid(a.mtiles * a.ntiles, a.splits);
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = g1_lds_wgrad(t);
    static const bool attr = g1_set_lds(g1_wgrad_kernel<4, 4, 1>, g1_lds_wgrad({4, 4})) && g1_set_lds(g1_wgrad_kernel<4, 4, 2>, g1_lds_wgrad({4, 4})) &&
                             g1_set_lds(g1_wgrad_kernel<4, 4, 1, true>, g1_lds_wgrad({4, 4}));
    if (!attr) return DC_ELAUNCH;
    hipEvent_t pe = conv_prof_begin(4, 2.0 * (double)B * Co * Ci * a.Ho * a.Wo, 2.0 * (double)grid.x * (32.0 * t.mt) * (32.0 * t.nt) * (double)a.chunks * GKC, 4.0 * ((double)B * Ci * a.Ho * a.Wo + (double)B * Co * a.Ho * a.Wo + (double)Co * Ci), st);
    if (bnin) {
        if (t.mt == 4) hipLaunchKernelGGL((g1_wgrad_kernel<4, 4, 1, true>), grid, dim3(256), lds, st, a);
        else hipLaunchKernelGGL((g1_wgrad_kernel<2, 2, 1, true>), grid, dim3(256), lds, st, a);
    } else if (stride == 1) {
        if (t.mt == 4) hipLaunchKernelGGL((g1_wgrad_kernel<4, 4, 1>), grid, dim3(256), lds, st, a);
        else hipLaunchKernelGGL((g1_wgrad_kernel<2, 2, 1>), grid, dim3(256), lds, st, a);
    } else {
        if (t.mt == 4) hipLaunchKernelGGL((g1_wgrad_kernel<4, 4, 2>), grid, dim3(256), lds, st, a);
        else hipLaunchKernelGGL((g1_wgrad_kernel<2, 2, 2>), grid, dim3(256), lds, st, a);
    }
    conv_prof_end(pe, st);
    DC_CHECK_LAUNCH();
    if (a.splits > 1) {
        const int n4 = Co * Ci / 4;
        hipLaunchKernelGGL(slab_reduce16_kernel<gf4>, dim3(ceil_div(n4, 16)), dim3(256), 0, st, (const gf4*)ws, (gf4*)dweight, a.splits, n4);
        DC_CHECK_LAUNCH();
    }
    return DC_OK;
}

extern "C" int dc_bias_act_bwd(const float* y, const float* gy, float* gpre, float* dbias, int B, int C, int P, int act, void* stream) {
    if (!y || !gy || B <= 0 || C <= 0 || P <= 0 || act < 0 || act > ACT_LAST) return DC_EINVAL;
    hipLaunchKernelGGL(g1_bias_act_bwd_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, y, gy, gpre, dbias, B, C, P, act);
    DC_CHECK_LAUNCH();
    return DC_OK;
}
