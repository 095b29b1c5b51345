// The strided convolutions of the ResNet trunks -- the 7x7 / 2 stem (3 or 6 input channels) and the 3x3 / 2 first
// convolution of layer2-4 (reference networks/resnet_encoder.py:87-98 via torchvision) -- as implicit GEMMs on the tiled
// fp32-MFMA pieces of gemm_tiles.h, straight on the NCHW tensors: no im2col tensor, no NHWC transposes, exact fp32
// products with fp32 accumulation (the library's weight gradient of the stem was measured 1 % off an fp64 reference).
//
//   forward   y[b,co,oy,ox]  = sum_{ci,ky,kx} w[co,ci,ky,kx] * x[b,ci,2oy+ky-p,2ox+kx-p]        p = KS/2, zero padding
//     GEMM: M = co, N = flattened (b,oy,ox), reduction k = (ci,ky,kx)
//   weight gradient   dw[co,(ci,ky,kx)] = sum_n gy[co,n] * x[..]: M = co, N = (ci,ky,kx), reduction = pixels, split over
//     blocks with a fixed-order slab reduce
//   data gradient (3x3 only; the stem's input is the image)   dx[b,ci,2oy+dy,2ox+dx] = sum over the taps whose parity
//     matches: (dy,dx) = (0,0): 1 tap, (0,1) / (1,0): 2 taps, (1,1): 4 taps -- 9 taps per 4 input pixels, nothing multiplied
//     by a structural zero.  One block owns an output-grid pixel tile and one row parity dy and keeps BOTH column parities
//     in registers, so it stores whole 32-byte runs of dx.  Weights are re-laid out once per call to [tap][co][ci].
//
// Three families of kernels, in the order they were written (DESIGN.md 4b has the counters that drove each step):
//   cg_fwd_kernel / cg_wgrad_kernel<KS>  general gather formulation: B-operand row k -> (ci,ky,kx), four consecutive output
//       pixels -> four stride-2 dword buffer loads.  Bound by L1 line throughput (a wave-level gather touches 16 cache lines
//       and uses an eighth of each); kept for shapes the others decline (Co != 64 stems, Ci % 32 != 0).
//   stem_fwd_kernel / stem_wgrad_kernel  the 7x7 / 2 stem with the input patch of a 2 x 64 pixel tile staged ONCE in LDS,
//       de-interleaved by column parity so that every tap is a contiguous run; operands read straight from the patch.
//   cg_fwd3_kernel / cg_wgrad3_kernel    3x3 / 2 with triple gathers: two aligned 16-byte loads + one dword give the three kx
//       taps of four pixels; 96-row super-chunks (forward), 64 x 96 tiles (weight gradient).
// All deterministic (no atomics).
#include "dc_common.h"
#include "conv_bf16.h"
#include "gemm_tiles.h"
#include "gemm1x1_x3.h"
#include "wino.h"

#include <stdlib.h>

#include <algorithm>

namespace dc {

struct CgArgs {
    const float* w;       // forward: (Co, Kp) rows; dgrad: [tap][Co][Ci]
    const float* x;       // (B, Ci, Hi, Wi)
    const float* gy;      // (B, Co, Ho, Wo)
    float* out;
    int B, Co, Ci, Hi, Wi, Ho, Wo;
    int K, Kp;            // Ci * KS * KS and its padded row length in `w`
    unsigned xbytes;      // size of x (buffer descriptor of the gathers)
    int mtiles, ntiles;
    int splits, chunks;   // weight gradient
};

// four stride-2 taps of one input row: x[ix0], x[ix0+2], x[ix0+4], x[ix0+6] with zero padding outside [0, Wi) and for a row
// outside the image.  Buffer loads with a 32-bit byte offset: an invalid tap gets an offset past num_records and the
// hardware returns 0 -- no branches, no 64-bit address arithmetic, no selects (the predicated dword loads this replaces
// cost ~75 vector instructions per four taps, 5 per MFMA at the stem, and left the matrix pipe 51 % busy).
using cgrsrc_t = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ gf4 gather4_s2(cgrsrc_t xr, unsigned rowoff, bool row_ok, int ix0, int Wi) {
    gf4 v;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int ix = ix0 + 2 * t;
        const unsigned off = (row_ok && (unsigned)ix < (unsigned)Wi) ? rowoff + 4u * (unsigned)ix : 0x80000000u;
        v[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, (int)off, 0, 0));
    }
    return v;
}

// =====================================================================================================================
// forward
// =====================================================================================================================
template <int MT, int NT, int KS>
__global__ __launch_bounds__(256) void cg_fwd_kernel(CgArgs a) {
    constexpr int BM = 32 * MT, BN = 32 * NT, KC = GKC, SB = IdxStride<NT, BN>::v, T = KS * KS, PAD = KS / 2;
    constexpr int NA = BM * KC / 1024, NB = KC * BN / 1024;
    constexpr int ASZ = BM * (KC + RP), BSZ = KC * SB;
    float* const As = g1_smem;
    float* const Bs = g1_smem + 2 * ASZ;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int lb = xcd_logical_block(blockIdx.x, gridDim.x);
    const int m0 = (lb % a.mtiles) * BM, n0 = (lb / a.mtiles) * BN;
    const int P = a.Ho * a.Wo, N = a.B * P;
    const unsigned plane = (unsigned)(a.Hi * a.Wi);

    const float* asrc[NA];
    int adst[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int idx = tid + j * 256, row = idx / (KC / 4), kq = idx % (KC / 4);
        asrc[j] = a.w + (size_t)min(m0 + row, a.Co - 1) * a.Kp + kq * 4;
        adst[j] = row * (KC + RP) + kq * 4;
    }
    // B: all of a thread's rows share one pixel group (c4 is the same for every j: 256 % (BN/4) == 0)
    const int c4 = tid % (BN / 4), krow0 = tid / (BN / 4);
    const int ng = min(n0 + c4 * 4, N - 4);
    const int b = ng / P, pp = ng - b * P, py = pp / a.Wo, px0 = pp - py * a.Wo;
    const cgrsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), (short)0, (int)a.xbytes, 0x00020000);
    const unsigned xboff = (unsigned)b * (unsigned)a.Ci * plane;           // elements; the whole tensor is < 2^29 elements (cg_ok)
    const int iy0 = 2 * py - PAD, ix0 = 2 * px0 - PAD;
    // (a second register stage -- loads of chunk c+2 in flight during the MFMAs of chunk c -- was measured: 200 VGPRs, two
    // blocks per CU instead of three, 8-28 % slower.  The gathers are bound by L1 line throughput, not by latency: a
    // stride-2 dword gather touches 16 cache lines per wave instruction and uses an eighth of each.)
    gf4 ra[NA], rb[NB];
    auto gload = [&](int k0) {
#pragma unroll
        for (int j = 0; j < NA; ++j) ra[j] = *reinterpret_cast<const gf4*>(asrc[j] + k0);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int k = k0 + krow0 + j * (1024 / BN);
            const int ci = k / T, tap = k - ci * T, ky = tap / KS, kx = tap - ky * KS;
            const int iy = iy0 + ky;
            const bool ok = k < a.K && (unsigned)iy < (unsigned)a.Hi;
            rb[j] = gather4_s2(xr, (xboff + (unsigned)ci * plane + (unsigned)(iy * a.Wi)) * 4u, ok, ix0 + kx, a.Wi);
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NA; ++j) store_red4(As + buf * ASZ + adst[j], ra[j]);
#pragma unroll
        for (int j = 0; j < NB; ++j) *reinterpret_cast<gf4*>(Bs + buf * BSZ + (krow0 + j * (1024 / BN)) * SB + c4 * 4) = rb[j];
    };
    gf4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = gf4{0, 0, 0, 0};
    const int nchunk = a.Kp / KC;
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
    const int j = lane & 15;
    const int n = n0 + wn * 16 * NT + j * NT;
    if (n < N) {
        const int bo = n / P, p = n - bo * P;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * 16 * MT + mt * 16 + (lane >> 4) * 4 + r;
                if (m >= a.Co) continue;
                float* dst = a.out + (size_t)blockIdx.y * a.B * a.Co * P + ((size_t)bo * a.Co + m) * P + p;
                if constexpr (NT == 4) *reinterpret_cast<gf4*>(dst) = gf4{acc[mt][0][r], acc[mt][1][r], acc[mt][2][r], acc[mt][3][r]};
                else *reinterpret_cast<gf2*>(dst) = gf2{acc[mt][0][r], acc[mt][1][r]};
            }
    }
}

// =====================================================================================================================
// weight gradient: rows = co, cols = (ci,ky,kx), reduction = flattened output pixels (a partial last chunk contributes
// zeros through the gy operand)
// =====================================================================================================================
template <int MT, int NT, int KS>
__global__ __launch_bounds__(256) void cg_wgrad_kernel(CgArgs a) {
    constexpr int BM = 32 * MT, BN = 32 * NT, KC = GKC, T = KS * KS, PAD = KS / 2;
    constexpr int NA = BM * KC / 1024, NB = BN * KC / 1024;
    constexpr int ASZ = BM * (KC + RP), BSZ = BN * (KC + RP);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int m0 = (blockIdx.x % a.mtiles) * BM, c0 = (blockIdx.x / a.mtiles) * BN;
    const int P = a.Ho * a.Wo;
    const unsigned plane = (unsigned)(a.Hi * a.Wi);
    const cgrsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), (short)0, (int)a.xbytes, 0x00020000);
    const int kq = tid % (KC / 4), row0 = tid / (KC / 4);
    size_t arow[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) arow[j] = (size_t)min(m0 + row0 + j * (1024 / KC), a.Co - 1) * P;
    // this thread's B rows: column r of dw <-> (ci, ky, kx), fixed for the whole loop
    unsigned bch[NB];
    int bky[NB], bkx[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int r = min(c0 + row0 + j * (1024 / KC), a.K - 1);
        const int ci = r / T, tap = r - ci * T;
        bky[j] = tap / KS - PAD;
        bkx[j] = tap - (tap / KS) * KS - PAD;
        bch[j] = (unsigned)ci * plane;
    }
    gf4 ra[NA], rb[NB];
    const int Ntot = a.B * P;
    auto gload = [&](int ch) {
        const int n_ = ch * KC + kq * 4;
        const bool okn = n_ < Ntot;
        const int n = min(n_, Ntot - 4);
        const int b = n / P, p = n - b * P, py = p / a.Wo, px0 = p - py * a.Wo;
        const float* ga = a.gy + (size_t)b * a.Co * P + p;
        const unsigned xboff = (unsigned)b * (unsigned)a.Ci * plane;
#pragma unroll
        for (int j = 0; j < NA; ++j) ra[j] = okn ? *reinterpret_cast<const gf4*>(ga + arow[j]) : gf4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int iy = 2 * py + bky[j];
            rb[j] = gather4_s2(xr, (xboff + bch[j] + (unsigned)(iy * a.Wi)) * 4u, (unsigned)iy < (unsigned)a.Hi, 2 * px0 + bkx[j], a.Wi);
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NA; ++j) store_red4(g1_smem + buf * ASZ + (row0 + j * (1024 / KC)) * (KC + RP) + kq * 4, ra[j]);
#pragma unroll
        for (int j = 0; j < NB; ++j)
            store_red4(g1_smem + 2 * ASZ + buf * BSZ + (row0 + j * (1024 / KC)) * (KC + RP) + kq * 4, rb[j]);
    };
    gf4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = gf4{0, 0, 0, 0};
    const int per = (a.chunks + a.splits - 1) / a.splits;
    const int ch0 = blockIdx.y * per, ch1 = min(ch0 + per, a.chunks);
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
    float* slab = a.out + (size_t)blockIdx.y * a.Co * a.K;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * 16 * MT + mt * 16 + (lane >> 4) * 4 + r;
                const int col = c0 + wn * 16 * NT + nt * 16 + (lane & 15);
                if (m < a.Co && col < a.K) slab[(size_t)m * a.K + col] = acc[mt][nt][r];
            }
}

// =====================================================================================================================
// data gradient of the 3x3 / 2 convolution.  rows = ci, cols = output-grid pixels n = (b,oy,ox); blockIdx.y = row parity dy.
// Reduction chunks: (tap, 32 output channels); A = wt[tap][co][ci] (index-contiguous), B = gy[co] shifted by the tap.
// =====================================================================================================================
template <int MT, int NT>
__global__ __launch_bounds__(256) void cg_dgrad3_kernel(CgArgs a) {
    constexpr int BM = 32 * MT, BN = 32 * NT, KC = GKC, SA = IdxStride<MT, BM>::v, SB = IdxStride<NT, BN>::v;
    constexpr int NA = KC * BM / 1024, NB = KC * BN / 1024;
    constexpr int ASZ = KC * SA, BSZ = KC * SB;
    float* const As = g1_smem;
    float* const Bs = g1_smem + 2 * ASZ;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int lb = xcd_logical_block(blockIdx.x, gridDim.x);
    const int m0 = (lb % a.mtiles) * BM, n0 = (lb / a.mtiles) * BN;
    const int dy = blockIdx.y;
    const int P = a.Ho * a.Wo, N = a.B * P;

    const float* asrc[NA];
    int adst[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int idx = tid + j * 256, k = idx / (BM / 4), c4 = idx % (BM / 4);
        asrc[j] = a.w + (size_t)k * a.Ci + min(m0 + c4 * 4, a.Ci - 4);
        adst[j] = k * SA + c4 * 4;
    }
    const int c4 = tid % (BN / 4), krow0 = tid / (BN / 4);
    const int ng = min(n0 + c4 * 4, N - 4);
    const int b = ng / P, pp = ng - b * P, oy = pp / a.Wo, ox0 = pp - oy * a.Wo;
    const float* gb = a.gy + (size_t)b * a.Co * P;
    // chunk list of this row parity: dy = 0 -> ky = 1; dy = 1 -> ky = 0 (reads gy row oy+1) and ky = 2 (row oy)
    const int nky = dy ? 2 : 1;
    const int cpt = a.Co / KC / (int)gridDim.z;        // chunks per tap of THIS reduction slab (blockIdx.z: a range of output channels)
    const int cz0 = blockIdx.z * cpt;
    const int nchunk = nky * 3 * cpt;
    gf4 ra[NA], rb[NB];
    auto tap_of = [&](int c, int& ky, int& kx, int& co0) {
        const int t = c / cpt;
        co0 = (cz0 + c - t * cpt) * KC;
        ky = dy ? (t / 3) * 2 : 1;
        kx = t % 3;
    };
    auto gload = [&](int c) {
        int ky, kx, co0;
        tap_of(c, ky, kx, co0);
        const float* wt = a.w + (size_t)(ky * 3 + kx) * a.Co * a.Ci + (size_t)co0 * a.Ci;
#pragma unroll
        for (int j = 0; j < NA; ++j) ra[j] = *reinterpret_cast<const gf4*>(asrc[j] + (wt - a.w));
        // input pixel (2oy+dy, 2ox+dxp) takes tap (ky,kx) from gy[(2oy+dy+1-ky)/2][(2ox+dxp+1-kx)/2]
        const int sy = (dy + 1 - ky) / 2, sx = (kx == 0) ? 1 : 0;      // kx = 1: dxp 0, shift 0; kx = 0: dxp 1, shift 1; kx = 2: dxp 1, shift 0
        const int gyr = oy + sy;
        const bool rok = gyr < a.Ho;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const float* row = gb + (size_t)(co0 + krow0 + j * (1024 / BN)) * P + (size_t)min(gyr, a.Ho - 1) * a.Wo;
            gf4 v = {0.f, 0.f, 0.f, 0.f};
            if (rok) {
                if (sx == 0) {
                    v = *reinterpret_cast<const gf4*>(row + ox0);
                } else {
                    v.x = row[ox0 + 1]; v.y = row[ox0 + 2]; v.z = row[ox0 + 3];
                    if (ox0 + 4 < a.Wo) v.w = row[ox0 + 4];
                }
            }
            rb[j] = v;
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NA; ++j) *reinterpret_cast<gf4*>(As + buf * ASZ + adst[j]) = ra[j];
#pragma unroll
        for (int j = 0; j < NB; ++j) *reinterpret_cast<gf4*>(Bs + buf * BSZ + (krow0 + j * (1024 / BN)) * SB + c4 * 4) = rb[j];
    };
    gf4 acc[2][MT][NT];          // [column parity]
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[d][mt][nt] = gf4{0, 0, 0, 0};
    gload(0);
    commit(0);
    __syncthreads();
    for (int c = 0; c < nchunk; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunk) gload(c + 1);
        int ky, kx, co0;
        tap_of(c, ky, kx, co0);
        if (kx == 1) {
            mma_chunk32<MT, NT>([&](int q, float (&v)[4][2]) { read_idx<MT, SA>(As + buf * ASZ, wm * 16 * MT, q, lane, v); },
                                [&](int q, float (&v)[4][2]) { read_idx<NT, SB>(Bs + buf * BSZ, wn * 16 * NT, q, lane, v); }, acc[0]);
        } else {
            mma_chunk32<MT, NT>([&](int q, float (&v)[4][2]) { read_idx<MT, SA>(As + buf * ASZ, wm * 16 * MT, q, lane, v); },
                                [&](int q, float (&v)[4][2]) { read_idx<NT, SB>(Bs + buf * BSZ, wn * 16 * NT, q, lane, v); }, acc[1]);
        }
        if (c + 1 < nchunk) commit(buf ^ 1);
        __syncthreads();
    }
    const int j = lane & 15;
    const int n = n0 + wn * 16 * NT + j * NT;
    if (n >= N) return;
    const int bo = n / P, p = n - bo * P, qy = p / a.Wo, qx = p - qy * a.Wo;
    const size_t plane = (size_t)a.Hi * a.Wi;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ci = m0 + wm * 16 * MT + ((lane >> 4) * 4 + r) * MT + mt;
            if (ci >= a.Ci) continue;
            float* dst = a.out + (size_t)blockIdx.z * a.B * a.Ci * plane + ((size_t)bo * a.Ci + ci) * plane + (size_t)(2 * qy + dy) * a.Wi + 2 * qx;
#pragma unroll
            for (int h = 0; h < NT / 2; ++h)
                *reinterpret_cast<gf4*>(dst + 4 * h) =
                    gf4{acc[0][mt][2 * h][r], acc[1][mt][2 * h][r], acc[0][mt][2 * h + 1][r], acc[1][mt][2 * h + 1][r]};
        }
}

// dx = slab 0 + slab 1 (+ ...), fixed order (the reduction slabs of cg_dgrad3_kernel)
__global__ __launch_bounds__(256) void cg_slabsum_kernel(const float* __restrict__ slabs, float* __restrict__ out, size_t n4, int splits) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256ull) {
        gf4 v = reinterpret_cast<const gf4*>(slabs)[i];
        for (int z = 1; z < splits; ++z) {
            const gf4 t = reinterpret_cast<const gf4*>(slabs)[i + (size_t)z * n4];
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        reinterpret_cast<gf4*>(out)[i] = v;
    }
}

// =====================================================================================================================
// 3x3 / 2 with TRIPLE gathers.  The three kx taps of one (ci, ky) and four consecutive output pixels read input columns
// 2*ox0-1 .. 2*ox0+7 of one row: two aligned 16-byte loads and one dword give all twelve values, where the per-tap gather
// issues twelve stride-2 dword loads (each touching 16 cache lines per wave and using an eighth of them).  Reduction index
// k = ci*9 + ky*3 + kx, so a (ci, ky) pair is three consecutive rows (forward: B rows; weight gradient: dw columns).
//   forward: super-chunks of 96 reduction rows = 32 triples, one LDS buffer, global loads of the next super-chunk in
//   registers during the 12 MFMA octets; the weight matrix (Co, Ci*9) as it lies in memory is the A operand.
// =====================================================================================================================
struct Trip { gf4 f0, f1; float e; };
__device__ __forceinline__ Trip trip_load(cgrsrc_t xr, unsigned rowoff, bool row_ok, int X0) {
    Trip t;
    const unsigned o = row_ok ? rowoff + 4u * (unsigned)X0 : 0x80000000u;
    t.f0 = __builtin_bit_cast(gf4, __builtin_amdgcn_raw_buffer_load_b128(xr, (int)o, 0, 0));
    t.f1 = __builtin_bit_cast(gf4, __builtin_amdgcn_raw_buffer_load_b128(xr, (int)o, 16, 0));
    t.e = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, (int)((row_ok && X0 > 0) ? o - 4u : 0x80000000u), 0, 0));
    return t;
}
__device__ __forceinline__ gf4 trip_tap(const Trip& t, int kx) {
    return kx == 0 ? gf4{t.e, t.f0.y, t.f0.w, t.f1.y} : (kx == 1 ? gf4{t.f0.x, t.f0.z, t.f1.x, t.f1.z} : gf4{t.f0.y, t.f0.w, t.f1.y, t.f1.w});
}

template <int MT, int NT>
__global__ __launch_bounds__(256) void cg_fwd3_kernel(CgArgs a) {
    constexpr int BM = 32 * MT, BN = 32 * NT, KC3 = 96, SB = IdxStride<NT, BN>::v;
    constexpr int NA = BM * KC3 / 1024, NG = 32 * (BN / 4) / 256;
    constexpr int ASZ = BM * (KC3 + RP);
    float* const As = g1_smem;
    float* const Bs = g1_smem + ASZ;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int lb = xcd_logical_block(blockIdx.x, gridDim.x);
    const int m0 = (lb % a.mtiles) * BM, n0 = (lb / a.mtiles) * BN;
    const int P = a.Ho * a.Wo, N = a.B * P;
    const unsigned plane = (unsigned)(a.Hi * a.Wi);

    const float* asrc[NA];
    int adst[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int idx = tid + j * 256, row = idx / (KC3 / 4), kq = idx % (KC3 / 4);
        asrc[j] = a.w + (size_t)min(m0 + row, a.Co - 1) * a.K + kq * 4;
        adst[j] = row * (KC3 + RP) + kq * 4;
    }
    const int c4 = tid % (BN / 4), trow0 = tid / (BN / 4);
    const int ng = min(n0 + c4 * 4, N - 4);
    const int b = ng / P, pp = ng - b * P, py = pp / a.Wo, px0 = pp - py * a.Wo;
    const cgrsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), (short)0, (int)a.xbytes, 0x00020000);
    const unsigned xboff = (unsigned)b * (unsigned)a.Ci * plane;
    gf4 ra[NA];
    Trip rt[NG];
    auto gload = [&](int sc) {
#pragma unroll
        for (int j = 0; j < NA; ++j) ra[j] = *reinterpret_cast<const gf4*>(asrc[j] + sc * KC3);
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const int t3 = sc * 32 + trow0 + j * (1024 / BN);          // = k / 3 = ci * 3 + ky
            const int ci = t3 / 3, ky = t3 - ci * 3;
            const int iy = 2 * py + ky - 1;
            rt[j] = trip_load(xr, (xboff + (unsigned)ci * plane + (unsigned)(iy * a.Wi)) * 4u, (unsigned)iy < (unsigned)a.Hi, 2 * px0);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int j = 0; j < NA; ++j) store_red4(As + adst[j], ra[j]);
#pragma unroll
        for (int j = 0; j < NG; ++j)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
                *reinterpret_cast<gf4*>(Bs + ((trow0 + j * (1024 / BN)) * 3 + kx) * SB + c4 * 4) = trip_tap(rt[j], kx);
    };
    gf4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = gf4{0, 0, 0, 0};
    // gridDim.y > 1: the reduction (super-chunks of 96 rows) is split into slabs summed in fixed order by cg_slabsum_kernel --
    // the deep layers have fewer 64 x 64 tiles than the chip has slots and 24-48 super-chunks per tile
    const int nsc_all = a.K / KC3, per = (nsc_all + (int)gridDim.y - 1) / (int)gridDim.y;
    const int sc0 = blockIdx.y * per, nsc = min(sc0 + per, nsc_all);
    if (sc0 < nsc) gload(sc0);
    for (int sc = sc0; sc < nsc; ++sc) {
        commit();
        __syncthreads();
        if (sc + 1 < nsc) gload(sc + 1);
#pragma unroll
        for (int u = 0; u < 3; ++u)
            mma_chunk32<MT, NT>([&](int q, float (&v)[4][2]) { read_red<MT, KC3>(As, wm * 16 * MT, u * 4 + q, lane, v); },
                                [&](int q, float (&v)[4][2]) { read_idx<NT, SB>(Bs, wn * 16 * NT, u * 4 + q, lane, v); }, acc);
        __syncthreads();
    }
    const int j = lane & 15;
    const int n = n0 + wn * 16 * NT + j * NT;
    if (n < N) {
        const int bo = n / P, p = n - bo * P;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * 16 * MT + mt * 16 + (lane >> 4) * 4 + r;
                if (m >= a.Co) continue;
                float* dst = a.out + (size_t)blockIdx.y * a.B * a.Co * P + ((size_t)bo * a.Co + m) * P + p;
                if constexpr (NT == 4) *reinterpret_cast<gf4*>(dst) = gf4{acc[mt][0][r], acc[mt][1][r], acc[mt][2][r], acc[mt][3][r]};
                else *reinterpret_cast<gf2*>(dst) = gf2{acc[mt][0][r], acc[mt][1][r]};
            }
    }
}

// weight gradient of the 3x3 / 2 convolution: 64 output channels x 96 columns (= 32 (ci, ky) pairs x 3 kx) per block;
// one triple gather per thread and reduction chunk of 32 pixels.  Same split / slab scheme as cg_wgrad_kernel.
__global__ __launch_bounds__(256) void cg_wgrad3_kernel(CgArgs a) {
    constexpr int MT = 2, NT = 3, BM = 64, BN = 96, KC = GKC;
    constexpr int NA = BM * KC / 1024;
    constexpr int ASZ = BM * (KC + RP), BSZ = BN * (KC + RP);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int m0 = (blockIdx.x % a.mtiles) * BM, c0 = (blockIdx.x / a.mtiles) * BN;
    const int P = a.Ho * a.Wo;
    const unsigned plane = (unsigned)(a.Hi * a.Wi);
    const cgrsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), (short)0, (int)a.xbytes, 0x00020000);
    const int kq = tid % (KC / 4), row0 = tid / (KC / 4);           // row0 = 0..31: A rows row0, row0 + 32; B triple row0
    size_t arow[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) arow[j] = (size_t)min(m0 + row0 + j * 32, a.Co - 1) * P;
    const int t3 = c0 / 3 + row0, tci = t3 / 3, tky = t3 - tci * 3;
    gf4 ra[NA];
    Trip rt;
    const int Ntot = a.B * P;
    auto gload = [&](int ch) {
        const int n_ = ch * KC + kq * 4;
        const bool okn = n_ < Ntot;
        const int n = min(n_, Ntot - 4);
        const int b = n / P, p = n - b * P, py = p / a.Wo, px0 = p - py * a.Wo;
        const float* ga = a.gy + (size_t)b * a.Co * P + p;
#pragma unroll
        for (int j = 0; j < NA; ++j) ra[j] = okn ? *reinterpret_cast<const gf4*>(ga + arow[j]) : gf4{0.f, 0.f, 0.f, 0.f};
        const int iy = 2 * py + tky - 1;
        rt = trip_load(xr, ((unsigned)(b * a.Ci + tci) * plane + (unsigned)(iy * a.Wi)) * 4u, (unsigned)iy < (unsigned)a.Hi, 2 * px0);
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NA; ++j) store_red4(g1_smem + buf * ASZ + (row0 + j * 32) * (KC + RP) + kq * 4, ra[j]);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) store_red4(g1_smem + 2 * ASZ + buf * BSZ + (row0 * 3 + kx) * (KC + RP) + kq * 4, trip_tap(rt, kx));
    };
    gf4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = gf4{0, 0, 0, 0};
    const int per = (a.chunks + a.splits - 1) / a.splits;
    const int ch0 = blockIdx.y * per, ch1 = min(ch0 + per, a.chunks);
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
    float* slab = a.out + (size_t)blockIdx.y * a.Co * a.K;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * 16 * MT + mt * 16 + (lane >> 4) * 4 + r;
                const int col = c0 + wn * 16 * NT + nt * 16 + (lane & 15);
                if (m < a.Co) slab[(size_t)m * a.K + col] = acc[mt][nt][r];
            }
}

// =====================================================================================================================
// The 7x7 / 2 stem (Ci = 3 or 6, Co = 64) with the input PATCH staged in LDS.
// In the gather formulation above every (tap, pixel) pair is its own dword load: each input element is fetched 49/4 = 12
// times, by stride-2 wave instructions that touch 16 cache lines and use an eighth of each -- the L1 path is as busy as
// the matrix pipe (SQ counters: MFMA 51 % busy, waits on vector memory, none on LDS).  Here a block owns a tile of 2 x 64
// output pixels of one image and loads the (2*2+5) x (2*64+8) x Ci input patch ONCE with aligned 16-byte loads,
// de-interleaved by column parity on the way into LDS (E[j] = x[2j], O[j] = x[2j+1]): tap kx of output column c is then
//   kx odd : E[c + (kx+1)/2]      kx even : O[c + kx/2]        (j counted from input column 2*ox0 - 4)
// i.e. CONTIGUOUS in c, so the B operand of the implicit GEMM (k = (ci,ky,kx), n = pixel) is read straight from the
// patch with ds_read_b32 at  koff[k] + r * 2 * ROW + c  -- no im2col tile, no per-tap global traffic.
// =====================================================================================================================
constexpr int ST_TH = 2, ST_TW = 64, ST_PW = 68, ST_ROWS = 2 * ST_TH + 5, ST_ROW = 2 * ST_PW;   // one patch row = [E | O]
struct StemArgs {
    const float* x; const float* w; const float* gy; float* out;
    // raw-frame mode (dc_stem_fwd / dc_stem_wgrad): nf > 0 frames (Bf,3,Hi,Wi); the network input is (x - mean) / stdv of
    // frames[0] (nf = 1, Ci = 3) or of the temporal pairs cat(f[p], f[p+1]) stacked along the batch (nf = 3: B = 2 Bf, Ci = 6)
    const float* f[3]; int nf, Bf; float mean, stdv;
    int B, Ci, Co, Hi, Wi, Ho, Wo, K, Kp;
    int tiles_x, tiles_y, ntiles;     // tiles per row, per image column, total (B * tiles_y * tiles_x)
    int nblocks;                      // weight gradient: blocks that share the tiles
    unsigned xbytes;
};

// koff of reduction index k = (ci, ky, kx): float offset of tap k for tile pixel (r = 0, c = 0)
__device__ __forceinline__ int stem_koff(int k, int K) {
    if (k >= K) return 0;                                 // padded columns: weight 0 / never stored; any valid address
    const int ci = k / 49, tap = k - ci * 49, ky = tap / 7, kx = tap - ky * 7;
    return (ci * ST_ROWS + ky) * ST_ROW + ((kx & 1) ? (kx + 1) / 2 : ST_PW + kx / 2);
}

// the patch of tile (b, ty, tx): 16-byte loads of rows 2*oy0-3 .. +8, columns 2*ox0-4 .. +135; rows / columns outside
// the image come back as zeros from the buffer descriptor
template <int NV, bool FR>
__device__ __forceinline__ void stem_patch_load(const StemArgs& a, cgrsrc_t xr, int b, int oy0, int ox0, int tid, gf4 (&v)[NV]) {
    const int nitems = a.Ci * ST_ROWS * (ST_ROW / 4);
    // raw-frame mode (FR): pair index and item inside it; channels 0-2 come from frame f[pair], 3-5 from f[pair + 1] -- the
    // temporal concat of trainer.py:398-412 is a pointer select (scalar selects on the three kernel arguments: indexing the
    // argument array with a run-time index is a dependent memory load in front of every patch load)
    const int pr_ = (FR && a.nf == 3) ? b / a.Bf : 0, item = b - pr_ * a.Bf;
    const float* const fa = pr_ ? a.f[1] : a.f[0];
    const float* const fb = pr_ ? a.f[2] : a.f[1];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int idx = tid + j * 256;
        const int row = idx / (ST_ROW / 4), m = idx - row * (ST_ROW / 4);
        const int ci = row / ST_ROWS, pr = row - ci * ST_ROWS;
        const int Y = 2 * oy0 - 3 + pr, X = 2 * ox0 - 4 + 4 * m;
        const bool ok = idx < nitems && (unsigned)Y < (unsigned)a.Hi && (unsigned)X < (unsigned)a.Wi;
        if (!FR) {
            const unsigned off = ok ? (((unsigned)(b * a.Ci + ci) * a.Hi + Y) * a.Wi + X) * 4u : 0x80000000u;
            v[j] = __builtin_bit_cast(gf4, __builtin_amdgcn_raw_buffer_load_b128(xr, (int)off, 0, 0));
        } else {
            // branch-free: an item outside the image reads the frame's first vector and is zeroed at commit
            const float* src = (ci >= 3) ? fb : fa;
            const int c3 = ci >= 3 ? ci - 3 : ci;
            const size_t off = ok ? (((size_t)(item * 3 + c3) * a.Hi + Y) * a.Wi + X) : 0;
            v[j] = *reinterpret_cast<const gf4*>(src + off);
        }
    }
}
template <int NV, bool FR>
__device__ __forceinline__ void stem_patch_store(const StemArgs& a, float* P, int tid, const gf4 (&v)[NV], int oy0, int ox0) {
    const int nitems = a.Ci * ST_ROWS * (ST_ROW / 4);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int idx = tid + j * 256;
        if (idx >= nitems) continue;
        const int row = idx / (ST_ROW / 4), m = idx - row * (ST_ROW / 4);
        gf4 w = v[j];
        if (FR) {
            // (x - mean) / stdv of networks/resnet_encoder.py:89 -- the same two IEEE operations as the tensor expression it
            // replaces -- on pixels inside the image; the zero padding of conv1 stays zero
            const int ci = row / ST_ROWS, pr = row - ci * ST_ROWS;
            const int Y = 2 * oy0 - 3 + pr, X = 2 * ox0 - 4 + 4 * m;
            const bool ok = (unsigned)Y < (unsigned)a.Hi && (unsigned)X < (unsigned)a.Wi;
            w.x = ok ? (w.x - a.mean) / a.stdv : 0.f; w.y = ok ? (w.y - a.mean) / a.stdv : 0.f;
            w.z = ok ? (w.z - a.mean) / a.stdv : 0.f; w.w = ok ? (w.w - a.mean) / a.stdv : 0.f;
        }
        float* e = P + row * ST_ROW + 2 * m;
        *reinterpret_cast<gf2*>(e) = gf2{w.x, w.z};
        *reinterpret_cast<gf2*>(e + ST_PW) = gf2{w.y, w.w};
    }
}

// ---- forward: block = tile x 64 output channels; 4 waves = (channel half wm) x (tile row wn); MT = 2, NT = 4 ------------
constexpr int ST_NVP = 8;        // ceil(6 * 9 * 34 / 256) patch vectors per thread
template <bool FR>
__global__ __launch_bounds__(256) void stem_fwd_kernel(StemArgs a) {
    constexpr int KC = GKC, MT = 2, NT = 4, ASZ = 64 * (KC + RP);
    float* const As = g1_smem;                       // [2][64][KC + RP]
    float* const P = g1_smem + 2 * ASZ;              // [Ci][ST_ROWS][E | O]
    int* const ktab = reinterpret_cast<int*>(P + a.Ci * ST_ROWS * ST_ROW);      // [Kp]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int lb = xcd_logical_block(blockIdx.x, gridDim.x);
    const int per_img = a.tiles_y * a.tiles_x;
    const int b = lb / per_img, tr = lb - b * per_img, ty = tr / a.tiles_x, tx = tr - ty * a.tiles_x;
    const int oy0 = ty * ST_TH, ox0 = tx * ST_TW;
    const int m0 = blockIdx.y * 64;
    const cgrsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), (short)0, (int)a.xbytes, 0x00020000);

    gf4 pv[ST_NVP];
    stem_patch_load<ST_NVP, FR>(a, xr, b, oy0, ox0, tid, pv);
    const float* asrc[2];
    int adst[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int idx = tid + j * 256, row = idx / (KC / 4), kq = idx % (KC / 4);
        asrc[j] = a.w + (size_t)(m0 + row) * a.Kp + kq * 4;
        adst[j] = row * (KC + RP) + kq * 4;
    }
    gf4 ra[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) ra[j] = *reinterpret_cast<const gf4*>(asrc[j]);
    for (int k = tid; k < a.Kp; k += 256) ktab[k] = stem_koff(k, a.K);
    stem_patch_store<ST_NVP, FR>(a, P, tid, pv, oy0, ox0);
#pragma unroll
    for (int j = 0; j < 2; ++j) store_red4(As + adst[j], ra[j]);
    __syncthreads();

    gf4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = gf4{0, 0, 0, 0};
    const int i = lane & 15, kp = lane >> 4;
    const float* Pl = P + wn * 2 * ST_ROW + i;            // this lane's pixel (r = wn, c = 16 t + i), tap offset added per k
    const int nchunk = a.Kp / KC;
    for (int c = 0; c < nchunk; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunk) {
#pragma unroll
            for (int j = 0; j < 2; ++j) ra[j] = *reinterpret_cast<const gf4*>(asrc[j] + (c + 1) * KC);
        }
        mma_chunk32<MT, NT>([&](int q, float (&v)[4][2]) { read_red<MT, KC>(As + buf * ASZ, wm * 16 * MT, q, lane, v); },
                            [&](int q, float (&v)[4][2]) {
                                const int2 ko = *reinterpret_cast<const int2*>(ktab + c * KC + q * 8 + 2 * kp);
#pragma unroll
                                for (int t = 0; t < NT; ++t) { v[t][0] = Pl[ko.x + 16 * t]; v[t][1] = Pl[ko.y + 16 * t]; }
                            }, acc);
        if (c + 1 < nchunk) {
#pragma unroll
            for (int j = 0; j < 2; ++j) store_red4(As + (buf ^ 1) * ASZ + adst[j], ra[j]);
        }
        __syncthreads();
    }
    const int oy = oy0 + wn;
    if (oy < a.Ho) {
        const int P2 = a.Ho * a.Wo;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * 32 + mt * 16 + kp * 4 + r;
                float* dst = a.out + ((size_t)b * a.Co + m) * P2 + (size_t)oy * a.Wo + ox0 + i;
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    if (ox0 + 16 * t + i < a.Wo) dst[16 * t] = acc[mt][t][r];
            }
    }
}

// ---- forward on the bf16 matrix cores with split fp32 operands (gemm1x1_x3.hip: three bf16 pieces per operand, six partial products,
// fp32 accumulation -- fp32 accuracy at 6/16 of the fp32 matrix time).  The fp32 kernel above runs AT the fp32 matrix peak (0.65-0.69 of
// the nominal rate incl. the K padding): the stem is the one convolution of the step that the matrix pipe bounds outright.
// Same tile (2 x 64 output pixels x 64 channels) and the same parity-de-interleaved fp32 patch in LDS; the waves split the PIXELS
// (tile row wn, 32-pixel half wh), so each keeps all 64 channels and a gathered + split B element feeds 4 x 6 matrix instructions.
// A: the weights' pieces [3][64][Kp] (g1x3_prep_item on the zero-padded weights; reduction permuted inside a chunk of 32 as there),
// double-buffered in LDS, one ds_read_b128 per (16-channel tile, piece).  B: lane (pixel i, k-group kg) gathers its 8 reduction
// elements k(kg, e) from the patch through the tap-offset table and splits them in registers.
constexpr int STX_AST = 40;                       // bf16 per A row in LDS (32 + 8)
constexpr int STX_APIECE = 64 * STX_AST;
template <bool FR>
__global__ __launch_bounds__(256, 2) void stem_fwd_x3_kernel(StemArgs a, const unsigned short* __restrict__ wa) {
    typedef __attribute__((ext_vector_type(8))) __bf16 sbf8;
    typedef __attribute__((ext_vector_type(4))) unsigned su4;
    unsigned short* const As = reinterpret_cast<unsigned short*>(g1_smem);                    // [2][3][64][40] bf16
    float* const P = g1_smem + (2 * 3 * STX_APIECE) / 2;                                      // [Ci][ST_ROWS][E | O]
    int* const ktab = reinterpret_cast<int*>(P + a.Ci * ST_ROWS * ST_ROW);                    // [Kp], in the PERMUTED order of the A pieces
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wn = wave >> 1, wh = wave & 1;
    const int lb = xcd_logical_block(blockIdx.x, gridDim.x);
    const int per_img = a.tiles_y * a.tiles_x;
    const int b = lb / per_img, tr = lb - b * per_img, ty = tr / a.tiles_x, tx = tr - ty * a.tiles_x;
    const int oy0 = ty * ST_TH, ox0 = tx * ST_TW;
    const cgrsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), (short)0, (int)a.xbytes, 0x00020000);

    gf4 pv[ST_NVP];
    stem_patch_load<ST_NVP, FR>(a, xr, b, oy0, ox0, tid, pv);
    // A staging: 3 pieces x 64 rows x 4 sixteen-byte items per chunk = 768 items, three per thread
    const unsigned short* asrc[3];
    int adst[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int idx = tid + j * 256, piece = idx >> 8, rem = idx & 255, row = rem >> 2, ch = rem & 3;
        asrc[j] = wa + ((size_t)piece * 64 + row) * a.Kp + ch * 8;
        adst[j] = piece * STX_APIECE + row * STX_AST + ch * 8;
    }
    su4 ra[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) ra[j] = *reinterpret_cast<const su4*>(asrc[j]);
    // tap offsets in the order the operand needs them: position 8 kg + e of a chunk <-> k = e < 4 ? 4 kg + e : 16 + 4 kg + e - 4
    for (int pos = tid; pos < a.Kp; pos += 256) {
        const int c = pos >> 5, pl = pos & 31, kg = pl >> 3, e = pl & 7;
        ktab[pos] = stem_koff(c * 32 + (e < 4 ? 4 * kg + e : 16 + 4 * kg + e - 4), a.K);
    }
    stem_patch_store<ST_NVP, FR>(a, P, tid, pv, oy0, ox0);
#pragma unroll
    for (int j = 0; j < 3; ++j) *reinterpret_cast<su4*>(As + adst[j]) = ra[j];
    __syncthreads();

    gf4 acc[4][2];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = gf4{0, 0, 0, 0};
    const int i = lane & 15, kg = lane >> 4;
    const float* Pl = P + wn * 2 * ST_ROW + wh * 32 + i;   // this lane's pixel (r = wn, c = 32 wh + 16 nt + i), tap offset added per k
    const int aoff = i * STX_AST + kg * 8;
    const int nchunk = a.Kp / 32;
    for (int c = 0; c < nchunk; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunk) {
#pragma unroll
            for (int j = 0; j < 3; ++j) ra[j] = *reinterpret_cast<const su4*>(asrc[j] + (c + 1) * 32);
        }
        // B: gather 8 taps per 16-pixel tile, split
        const int4 k0 = *reinterpret_cast<const int4*>(ktab + c * 32 + kg * 8), k1 = *reinterpret_cast<const int4*>(ktab + c * 32 + kg * 8 + 4);
        sbf8 bv[2][3];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const float* q = Pl + 16 * nt;
            const float v0 = q[k0.x], v1 = q[k0.y], v2 = q[k0.z], v3 = q[k0.w], v4 = q[k1.x], v5 = q[k1.y], v6 = q[k1.z], v7 = q[k1.w];
            unsigned p[3][4];
            x3h_split2(v0, v1, p[0][0], p[1][0], p[2][0]);
            x3h_split2(v2, v3, p[0][1], p[1][1], p[2][1]);
            x3h_split2(v4, v5, p[0][2], p[1][2], p[2][2]);
            x3h_split2(v6, v7, p[0][3], p[1][3], p[2][3]);
#pragma unroll
            for (int s3 = 0; s3 < 3; ++s3) bv[nt][s3] = __builtin_bit_cast(sbf8, su4{p[s3][0], p[s3][1], p[s3][2], p[s3][3]});
        }
        const unsigned short* Ab = As + buf * 3 * STX_APIECE + aoff;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            sbf8 av[3];
#pragma unroll
            for (int s3 = 0; s3 < 3; ++s3) av[s3] = *reinterpret_cast<const sbf8*>(Ab + s3 * STX_APIECE + mt * 16 * STX_AST);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                gf4 d = acc[mt][nt];
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[2], bv[nt][0], d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[1], bv[nt][1], d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[0], bv[nt][2], d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[1], bv[nt][0], d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[0], bv[nt][1], d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[0], bv[nt][0], d, 0, 0, 0);
                acc[mt][nt] = d;
            }
        }
        if (c + 1 < nchunk) {
#pragma unroll
            for (int j = 0; j < 3; ++j) *reinterpret_cast<su4*>(As + (buf ^ 1) * 3 * STX_APIECE + adst[j]) = ra[j];
        }
        __syncthreads();
    }
    const int oy = oy0 + wn;
    if (oy < a.Ho) {
        const int P2 = a.Ho * a.Wo;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = mt * 16 + kg * 4 + r;
                float* dst = a.out + ((size_t)b * a.Co + m) * P2 + (size_t)oy * a.Wo + ox0 + wh * 32 + i;
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    if (ox0 + wh * 32 + 16 * nt + i < a.Wo) dst[16 * nt] = acc[mt][nt][r];
            }
    }
}
__global__ __launch_bounds__(256) void stem_x3_prep_kernel(const float* __restrict__ wp, unsigned short* __restrict__ wa, int Kp) {
    g1x3_prep_item(wp, wa, blockIdx.x * 256 + threadIdx.x, 0, 64, 64, Kp);       // the zero-padded (64, Kp) weights as a 1x1 "forward" A
}

// ---- weight gradient: a block walks a contiguous range of tiles and keeps dw[64][Kp] in registers ------------------------
// 4 waves = (channel half wm) x (column half wn); a wave owns 32 channels x Kp/2 columns = MT 2 x NT (5 or 10) tiles.
// Reduction = the tile's 128 pixels in octets of 8 along a row: A = gy[co][pixel] (reduction-contiguous image, as the
// GEMM kernels), B[k][pixel] = patch[koff[k] + r * 2 * ROW + c] with koff fixed per lane for the whole kernel.
template <int NT, bool FR>
__global__ __launch_bounds__(256, 2) void stem_wgrad_kernel(StemArgs a) {
    constexpr int MT = 2, GS = 128 + RP, NVG = 8;
    float* const G = g1_smem;                        // [64][128 + RP]   gy of the tile, pixel = r * 64 + c
    float* const P = g1_smem + 64 * GS;              // patch
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int i = lane & 15, kp = lane >> 4;
    const cgrsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), (short)0, (int)a.xbytes, 0x00020000);
    const int per = (a.ntiles + a.nblocks - 1) / a.nblocks;
    const int t0 = blockIdx.x * per, t1 = min(t0 + per, a.ntiles);
    const int per_img = a.tiles_y * a.tiles_x, P2 = a.Ho * a.Wo;
    const cgrsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.gy), (short)0, (int)((size_t)a.B * a.Co * P2 * 4), 0x00020000);
    int koff[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) koff[t] = stem_koff(wn * 16 * NT + 16 * t + i, a.K);

    // the next tile's patch is prefetched into registers during the MFMAs; its gy rows too when the accumulators leave
    // room (NT = 5) -- with NT = 10 (80 accumulator registers) they are fetched in the commit phase instead (no spills at
    // two blocks per CU; the other block multiplies meanwhile)
    constexpr bool PREFETCH_G = NT <= 5;
    gf4 pv[ST_NVP], gv[PREFETCH_G ? NVG : 1];
    auto tile_org = [&](int tile, int& b, int& oy0, int& ox0) {
        b = tile / per_img;
        const int tr = tile - b * per_img, ty = tr / a.tiles_x;
        oy0 = ty * ST_TH; ox0 = (tr - ty * a.tiles_x) * ST_TW;
    };
    auto gy_vec = [&](int b, int oy0, int ox0, int j) {          // gy: 64 channels x 2 rows x 16 float4
        const int idx = tid + j * 256, co = idx >> 5, rr = (idx >> 4) & 1, c4 = idx & 15;
        const int oy = oy0 + rr, ox = ox0 + c4 * 4;
        // buffer load with an out-of-range offset for the zeros: a select around the load made the compiler wait for each
        // of these eight loads before issuing the next
        const unsigned off = (oy < a.Ho && ox < a.Wo) ? (unsigned)(((b * a.Co + co) * P2 + oy * a.Wo + ox) * 4) : 0x80000000u;
        return __builtin_bit_cast(gf4, __builtin_amdgcn_raw_buffer_load_b128(gr, (int)off, 0, 0));
    };
    auto gy_store = [&](int j, gf4 v) {
        const int idx = tid + j * 256, co = idx >> 5, rr = (idx >> 4) & 1, c4 = idx & 15;
        store_red4(G + co * GS + rr * 64 + c4 * 4, v);
    };
    auto gload = [&](int tile) {
        int b, oy0, ox0;
        tile_org(tile, b, oy0, ox0);
        stem_patch_load<ST_NVP, FR>(a, xr, b, oy0, ox0, tid, pv);
        if constexpr (PREFETCH_G) {
#pragma unroll
            for (int j = 0; j < NVG; ++j) gv[j] = gy_vec(b, oy0, ox0, j);
        }
    };
    auto commit = [&](int tile) {
        {
            int b_, oy_, ox_;
            tile_org(tile, b_, oy_, ox_);
            stem_patch_store<ST_NVP, FR>(a, P, tid, pv, oy_, ox_);
        }
        if constexpr (PREFETCH_G) {
#pragma unroll
            for (int j = 0; j < NVG; ++j) gy_store(j, gv[j]);
        } else {
            int b, oy0, ox0;
            tile_org(tile, b, oy0, ox0);
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                gf4 t4[NVG / 2];
#pragma unroll
                for (int j = 0; j < NVG / 2; ++j) t4[j] = gy_vec(b, oy0, ox0, h2 * (NVG / 2) + j);
#pragma unroll
                for (int j = 0; j < NVG / 2; ++j) gy_store(h2 * (NVG / 2) + j, t4[j]);
            }
        }
    };
    gf4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = gf4{0, 0, 0, 0};
    if (t0 < t1) gload(t0);
    for (int tile = t0; tile < t1; ++tile) {
        commit(tile);
        __syncthreads();
        if (tile + 1 < t1) gload(tile + 1);
#pragma unroll 1
        for (int h = 0; h < 4; ++h) {                   // half rows of 32 pixels: r = h >> 1, c from (h & 1) * 32
            const float* Gh = G + h * 32;
            const float* Ph = P + (h >> 1) * 2 * ST_ROW + (h & 1) * 32 + 2 * kp;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float av[MT][2], bv[NT][2];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const gf2 v = *reinterpret_cast<const gf2*>(Gh + (wm * 32 + mt * 16 + i) * GS + q * 8 + 2 * kp);
                    av[mt][0] = v.x; av[mt][1] = v.y;
                }
#pragma unroll
                for (int t = 0; t < NT; ++t) { bv[t][0] = Ph[koff[t] + q * 8]; bv[t][1] = Ph[koff[t] + q * 8 + 1]; }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int t = 0; t < NT; ++t)
                            acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt][s2], bv[t][s2], acc[mt][t], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    float* slab = a.out + (size_t)blockIdx.x * 64 * a.Kp;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                slab[(size_t)(wm * 32 + mt * 16 + kp * 4 + r) * a.Kp + wn * 16 * NT + 16 * t + i] = acc[mt][t][r];
}

// dw[co][k] (k < K) = sum over blocks of slab[block][co][Kp], 16 block groups x 16 lanes (fixed order)
// (16-byte reads over the padded rows of the slabs -- Kp % 4 == 0 -- : a group reads 256 contiguous bytes per slab; the scalar form
// moved 64-byte segments and ran at ~1 TB/s on the 21 / 42 MB of the two stems.  Same order per element.)
__global__ __launch_bounds__(256) void stem_wreduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int nblocks, int K, int Kp) {
    __shared__ gf4 sm[256];
    const int g = threadIdx.x >> 4, l = threadIdx.x & 15;
    const int q = blockIdx.x * 16 + l;                  // over 64 * Kp / 4 quads of the padded rows
    const int nq = 64 * Kp / 4, kq = Kp / 4;
    const int co = min(q, nq - 1) / kq, k = (min(q, nq - 1) - co * kq) * 4;
    const int per = (nblocks + 15) / 16, s0 = g * per, s1 = min(s0 + per, nblocks);
    gf4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int s = s0; s < s1; ++s) t += *reinterpret_cast<const gf4*>(slab + ((size_t)s * 64 + co) * Kp + k);
    sm[threadIdx.x] = t;
    __syncthreads();
    if (g == 0 && q < nq) {
        gf4 r = sm[l];
#pragma unroll
        for (int j = 1; j < 16; ++j) r += sm[j * 16 + l];
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (k + e < K) dw[(size_t)co * K + k + e] = r[e];
    }
}

// ---- weight re-layouts (once per call; a few hundred KB) ---------------------------------------------------------------
__global__ __launch_bounds__(256) void cg_wpad_kernel(const float* __restrict__ w, float* __restrict__ wp, int Co, int K, int Kp) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Co * Kp) return;
    const int co = i / Kp, k = i - co * Kp;
    wp[i] = k < K ? w[(size_t)co * K + k] : 0.f;
}
__global__ __launch_bounds__(256) void cg_wt3_kernel(const float* __restrict__ w, float* __restrict__ wt, int Co, int Ci) {
    const int i = blockIdx.x * 256 + threadIdx.x;        // over [tap][co][ci]
    if (i >= 9 * Co * Ci) return;
    const int tap = i / (Co * Ci), r = i - tap * (Co * Ci), co = r / Ci, ci = r - co * Ci;
    wt[i] = w[((size_t)co * Ci + ci) * 9 + tap];
}

struct CgTile { int mt, nt; };
static size_t cg_lds_fwd(CgTile t) { return (size_t)2 * (32 * t.mt * (GKC + RP) + GKC * (t.nt == 4 ? 128 : 80)) * sizeof(float); }
static size_t cg_lds_dgrad(CgTile t) { return (size_t)2 * GKC * ((t.mt == 4 ? 128 : 80) + (t.nt == 4 ? 128 : 80)) * sizeof(float); }
static size_t cg_lds_wgrad(CgTile t) { return (size_t)2 * 32 * (t.mt + t.nt) * (GKC + RP) * sizeof(float); }
static CgTile cg_pick(int M, int N) {
    auto blocks = [&](int mt, int nt) { return (long)ceil_div(M, 32 * mt) * ceil_div(N, 32 * nt); };
    if (M > 64 && blocks(4, 4) >= 500) return {4, 4};
    if (blocks(2, 4) >= 300 || N >= 8 * M) return {2, 4};
    return {2, 2};
}
static int cg_wsplits(int tiles, int chunks) {
    const int s = std::max(1, std::min({chunks, ceil_div(768, tiles), 512}));
    return ceil_div(chunks, ceil_div(chunks, s));
}
template <typename K>
static bool cg_set_lds(K kernel, size_t bytes) {
    return hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess;
}

// the patch-staged stem kernels: 7x7 / 2, 64 output channels, 3 or 6 input channels (DC_STEM_PATCH=0: gather kernels, A/B)
static bool stem_enabled() {
    static const bool v = [] { const char* e = getenv("DC_STEM_PATCH"); return !e || atoi(e) != 0; }();
    return v;
}
// triple-gather kernels of the 3x3 / 2 convolutions: whole (ci, ky) triples per 96-row super-chunk (DC_CONV_TRIP=0: A/B)
static bool trip_ok(int Ci, int ks) {
    static const bool v = [] { const char* e = getenv("DC_CONV_TRIP"); return !e || atoi(e) != 0; }();
    return v && ks == 3 && Ci % 32 == 0;
}
static bool stem_ok(int Ci, int Co, int ks) { return stem_enabled() && ks == 7 && Co == 64 && (Ci == 3 || Ci == 6); }
static StemArgs stem_args(int B, int Ci, int Co, int Hi, int Wi) {
    StemArgs a{};
    a.B = B; a.Ci = Ci; a.Co = Co; a.Hi = Hi; a.Wi = Wi; a.Ho = Hi / 2; a.Wo = Wi / 2;
    a.K = Ci * 49; a.Kp = ceil_div(a.K, GKC) * GKC;
    a.tiles_x = ceil_div(a.Wo, ST_TW); a.tiles_y = ceil_div(a.Ho, ST_TH); a.ntiles = B * a.tiles_y * a.tiles_x;
    a.nblocks = std::min(a.ntiles, 512);
    a.xbytes = (unsigned)((size_t)B * Ci * Hi * Wi * sizeof(float));
    return a;
}
static size_t stem_lds_fwd(int Ci, int Kp) { return ((size_t)2 * 64 * (GKC + RP) + (size_t)Ci * ST_ROWS * ST_ROW + Kp) * sizeof(float); }
static size_t stem_lds_fwd_x3(int Ci, int Kp) { return (size_t)2 * 3 * STX_APIECE * 2 + ((size_t)Ci * ST_ROWS * ST_ROW + Kp) * sizeof(float); }
// the split-operand forward (dc_set_gemm_split, default on; DC_STEM_X3=0 keeps the fp32-MFMA kernel alone)
static bool stem_x3_enabled() {
    static const bool v = [] { const char* e = getenv("DC_STEM_X3"); return !e || atoi(e) != 0; }();
    return v && dc_get_gemm_split() != 0;
}
// pads are in ws (fp32 (64, Kp)); their pieces go behind them
template <bool FR>
static int stem_fwd_x3_launch(StemArgs& sa, void* ws, hipStream_t st) {
    const size_t off = ((size_t)64 * sa.Kp * sizeof(float) + 255) & ~(size_t)255;
    unsigned short* wa = reinterpret_cast<unsigned short*>((char*)ws + off);
    hipLaunchKernelGGL(stem_x3_prep_kernel, dim3(ceil_div(64 * (sa.Kp / 4), 256)), dim3(256), 0, st, (const float*)ws, wa, sa.Kp);
    DC_CHECK_LAUNCH();
    const size_t lds = stem_lds_fwd_x3(sa.Ci, sa.Kp);
    static const bool attr = cg_set_lds(stem_fwd_x3_kernel<FR>, stem_lds_fwd_x3(6, 320));
    if (!attr) return DC_ELAUNCH;
    hipLaunchKernelGGL(stem_fwd_x3_kernel<FR>, dim3(sa.ntiles, 1), dim3(256), lds, st, sa, (const unsigned short*)wa);
    return DC_OK;
}
static size_t stem_lds_wgrad(int Ci) { return ((size_t)64 * (128 + RP) + (size_t)Ci * ST_ROWS * ST_ROW) * sizeof(float); }

}  // namespace dc

using namespace dc;

static bool cg_ok(int B, int Ci, int Co, int Hi, int Wi, int ks) {
    if (B <= 0 || Ci <= 0 || Co <= 0 || Hi <= 0 || Wi <= 0 || (ks != 3 && ks != 7)) return false;
    if ((Hi & 1) || (Wi & 1)) return false;
    const int Wo = Wi / 2;
    if (Wo & 3) return false;                           // 16-byte pixel groups inside one output row
    if ((size_t)B * std::max(Ci, Co) * Hi * Wi >= (1ull << 29)) return false;      // 32-bit byte offsets of the buffer gathers
    return true;
}
static int cg_kp(int Ci, int ks) { return ceil_div(Ci * ks * ks, GKC) * GKC; }
// 3 x 3 / 2 forward and weight gradient on the split-operand kernels of gemm1x1_x3.hip: only under dc_set_gemm_split(3) (measured slower
// than cg_fwd3 / cg_wgrad3 on most step shapes: see g1x3_conv3s2_fwd).  The workspace sizes always cover both paths: the mode may change
// between the query and the launch.
static bool cg_x3_enabled() { return (dc_get_gemm_split() & 2) != 0; }

extern "C" int dc_convs2_supported(int B, int Ci, int Co, int Hi, int Wi, int ksize) { return cg_ok(B, Ci, Co, Hi, Wi, ksize) ? 1 : 0; }

// 3x3 / 2 forward (triple-gather kernel): tile and reduction split.  Returns the number of slabs (1: straight into y).
static int cg_fwd3_plan(int B, int Ci, int Co, int Hi, int Wi, CgTile& t) {
    const int N = B * (Hi / 2) * (Wi / 2);
    t = ((long)ceil_div(Co, 64) * ceil_div(N, 128) >= 300 || N >= 8 * Co) ? CgTile{2, 4} : CgTile{2, 2};
    const long blocks = (long)ceil_div(Co, 64) * ceil_div(N, 32 * t.nt);
    const int nsc = Ci * 9 / 96;
    int s = 1;
    while (s < 4 && blocks * s * 2 <= 768 && nsc / (s * 2) >= 3) s *= 2;
    if (const char* f = getenv("DC_FWD3_SPLIT")) { const int v = atoi(f); if (v == 1 || v == 2 || v == 4) s = std::min(v, std::max(1, nsc)); }   // experiments
    return s;
}

extern "C" size_t dc_convs2_fwd_workspace(int B, int Ci, int Co, int Hi, int Wi, int ksize) {
    if (!cg_ok(B, Ci, Co, Hi, Wi, ksize)) return 0;
    const int K = Ci * ksize * ksize, Kp = cg_kp(Ci, ksize);
    const size_t b16 = ksize == 3 ? c3b_weights_bytes(Ci, Co) : 0;
    size_t slabs = 0;
    if (trip_ok(Ci, ksize)) {
        CgTile t;
        const int sp = cg_fwd3_plan(B, Ci, Co, Hi, Wi, t);
        if (sp > 1) slabs = (size_t)sp * B * Co * (Hi / 2) * (Wi / 2) * sizeof(float);
    }
    // (7x7 stem: the zero-padded fp32 weights + their three bf16 pieces for the split-operand forward)
    const size_t wpad = Kp == K ? (size_t)16 : (size_t)Co * Kp * sizeof(float) + (ksize == 7 ? (size_t)Co * Kp * 6 + 512 : 0);
    const size_t x3 = ksize == 3 && g1x3_conv3s2_ok(B, Ci, Co, Hi, Wi) ? g1x3_conv3s2_fwd_ws(Ci, Co) : 0;
    return std::max(std::max(std::max(wpad, b16), slabs), x3);
}

extern "C" int dc_convs2_fwd(const float* x, const float* weight, float* y, void* ws, int B, int Ci, int Co, int Hi, int Wi, int ksize,
                             void* stream) {
    if (!x || !weight || !y || !ws || !cg_ok(B, Ci, Co, Hi, Wi, ksize)) return DC_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (ksize == 3 && matrix_precision() == DC_PREC_BF16 && c3b_eligible(Ci, 0, 0, Hi, Wi, 2))      // bf16 matrix cores (conv_bf16.hip)
        return c3b_conv(x, Ci, 0, nullptr, 0, weight, Co, Ci, 0, 0, nullptr, y, ws, B, Hi, Wi, ACT_NONE, PAD_ZERO, 2, st);
    if (ksize == 3 && cg_x3_enabled() && g1x3_conv3s2_ok(B, Ci, Co, Hi, Wi)) return g1x3_conv3s2_fwd(x, weight, y, ws, B, Ci, Co, Hi, Wi, st);
    CgArgs a{};
    a.x = x; a.out = y; a.B = B; a.Ci = Ci; a.Co = Co; a.Hi = Hi; a.Wi = Wi; a.Ho = Hi / 2; a.Wo = Wi / 2;
    a.K = Ci * ksize * ksize; a.Kp = cg_kp(Ci, ksize);
    a.xbytes = (unsigned)((size_t)B * Ci * Hi * Wi * sizeof(float));
    if (a.Kp != a.K) {       // stem: rows of 147 / 294 floats are neither 16-byte aligned nor a multiple of the chunk
        hipLaunchKernelGGL(cg_wpad_kernel, dim3(ceil_div(Co * a.Kp, 256)), dim3(256), 0, st, weight, (float*)ws, Co, a.K, a.Kp);
        DC_CHECK_LAUNCH();
        a.w = (const float*)ws;
    } else {
        a.w = weight;
    }
    if (stem_ok(Ci, Co, ksize)) {
        StemArgs sa = stem_args(B, Ci, Co, Hi, Wi);
        sa.x = x; sa.w = a.w; sa.out = y;
        hipEvent_t pe = conv_prof_begin(6, 2.0 * B * (double)Co * a.K * a.Ho * a.Wo, 2.0 * B * (double)Co * sa.Kp * a.Ho * a.Wo,
                                        4.0 * ((double)B * Ci * Hi * Wi + (double)B * Co * a.Ho * a.Wo + (double)Co * a.K), st);
        if (stem_x3_enabled() && a.Kp != a.K) {
            const int rc = stem_fwd_x3_launch<false>(sa, ws, st);
            if (rc != DC_OK) return rc;
        } else {
            hipLaunchKernelGGL(stem_fwd_kernel<false>, dim3(sa.ntiles, Co / 64), dim3(256), stem_lds_fwd(Ci, sa.Kp), st, sa);
        }
        conv_prof_end(pe, st);
        DC_CHECK_LAUNCH();
        return DC_OK;
    }
    const int N = B * a.Ho * a.Wo;
    if (trip_ok(Ci, ksize)) {
        CgTile t;
        const int sp = cg_fwd3_plan(B, Ci, Co, Hi, Wi, t);
        if (sp > 1) a.out = (float*)ws;            // (Kp == K for the 3x3: the workspace holds nothing else on this path)
        a.mtiles = ceil_div(Co, 64); a.ntiles = ceil_div(N, 32 * t.nt);
        const size_t lds3 = ((size_t)64 * (96 + RP) + (size_t)96 * (t.nt == 4 ? 128 : 80)) * sizeof(float);
        static const bool attr3 = cg_set_lds(cg_fwd3_kernel<2, 4>, ((size_t)64 * (96 + RP) + (size_t)96 * 128) * sizeof(float));
        if (!attr3) return DC_ELAUNCH;
        const dim3 grid(a.mtiles * a.ntiles, sp);
        // 2 MAC of the convolution; executed = the padded 64 x (32 nt) tiles over the whole reduction
        hipEvent_t pe = conv_prof_begin(5, 2.0 * (double)N * Co * a.K, 2.0 * (double)a.mtiles * 64.0 * (double)a.ntiles * 32.0 * t.nt * a.K,
                                        4.0 * ((double)B * Ci * Hi * Wi + (double)N * Co + (double)Co * a.K), st);
        if (t.nt == 4) hipLaunchKernelGGL((cg_fwd3_kernel<2, 4>), grid, dim3(256), lds3, st, a);
        else hipLaunchKernelGGL((cg_fwd3_kernel<2, 2>), grid, dim3(256), lds3, st, a);
        conv_prof_end(pe, st);
        DC_CHECK_LAUNCH();
        if (sp > 1) {
            const size_t n4 = (size_t)B * Co * a.Ho * a.Wo / 4;
            hipLaunchKernelGGL(cg_slabsum_kernel, dim3((unsigned)std::min<size_t>((n4 + 255) / 256, 4096)), dim3(256), 0, st, (const float*)ws, y, n4, sp);
            DC_CHECK_LAUNCH();
        }
        return DC_OK;
    }
    const CgTile t = cg_pick(Co, N);
    a.mtiles = ceil_div(Co, 32 * t.mt); a.ntiles = ceil_div(N, 32 * t.nt);
    const dim3 grid(a.mtiles * a.ntiles);
    const size_t lds = cg_lds_fwd(t);
    static const bool attr = cg_set_lds(cg_fwd_kernel<4, 4, 3>, cg_lds_fwd({4, 4})) && cg_set_lds(cg_fwd_kernel<4, 4, 7>, cg_lds_fwd({4, 4}));
    if (!attr) return DC_ELAUNCH;
#define CG_FWD(KS)                                                                                  \
    do {                                                                                            \
        if (t.mt == 4) hipLaunchKernelGGL((cg_fwd_kernel<4, 4, KS>), grid, dim3(256), lds, st, a);  \
        else if (t.nt == 4) hipLaunchKernelGGL((cg_fwd_kernel<2, 4, KS>), grid, dim3(256), lds, st, a); \
        else hipLaunchKernelGGL((cg_fwd_kernel<2, 2, KS>), grid, dim3(256), lds, st, a);            \
    } while (0)
    if (ksize == 3) CG_FWD(3); else CG_FWD(7);
#undef CG_FWD
    DC_CHECK_LAUNCH();
    return DC_OK;
}

extern "C" size_t dc_convs2_wgrad_workspace(int B, int Ci, int Co, int Hi, int Wi, int ksize) {
    if (!cg_ok(B, Ci, Co, Hi, Wi, ksize)) return 0;
    const int K = Ci * ksize * ksize;
    if (stem_ok(Ci, Co, ksize)) {
        const StemArgs sa = stem_args(B, Ci, Co, Hi, Wi);
        return (size_t)sa.nblocks * 64 * sa.Kp * sizeof(float);
    }
    const int tiles = ceil_div(Co, 64) * ceil_div(K, trip_ok(Ci, ksize) ? 96 : 64);
    const int splits = cg_wsplits(tiles, ceil_div(B * (Hi / 2) * (Wi / 2), GKC));
    const size_t b16 = ksize == 3 ? (size_t)c3b_wgrad_split(B, Hi / 2, Wi / 2, Co, Ci, 2) * Co * K * sizeof(float) : 0;
    const size_t x3 = ksize == 3 && g1x3_conv3s2_wgrad_ok(B, Ci, Co, Hi, Wi) ? g1x3_conv3s2_wgrad_ws(B, Ci, Co, Hi, Wi) : 0;
    return std::max(std::max(splits > 1 ? (size_t)splits * Co * K * sizeof(float) : (size_t)16, b16), x3);
}

extern "C" int dc_convs2_wgrad(const float* x, const float* gy, float* dweight, void* ws, int B, int Ci, int Co, int Hi, int Wi,
                               int ksize, void* stream) {
    if (!x || !gy || !dweight || !ws || !cg_ok(B, Ci, Co, Hi, Wi, ksize)) return DC_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (ksize == 3 && matrix_precision() == DC_PREC_BF16 && c3b_eligible(Ci, 0, 0, Hi, Wi, 2)) {    // bf16 matrix cores
        const int sp = c3b_wgrad_split(B, Hi / 2, Wi / 2, Co, Ci, 2);
        const int rc = c3b_wgrad(x, Ci, 0, nullptr, 0, gy, (float*)ws, sp, B, Co, Hi, Wi, PAD_ZERO, 2, st);
        return rc != DC_OK ? rc : conv_wreduce((const float*)ws, nullptr, dweight, nullptr, sp, Co * Ci * 9, 0, st);
    }
    if (ksize == 3 && cg_x3_enabled() && g1x3_conv3s2_wgrad_ok(B, Ci, Co, Hi, Wi))
        return g1x3_conv3s2_wgrad(x, gy, dweight, ws, B, Ci, Co, Hi, Wi, st);
    if (stem_ok(Ci, Co, ksize)) {
        StemArgs sa = stem_args(B, Ci, Co, Hi, Wi);
        sa.x = x; sa.gy = gy; sa.out = (float*)ws;
        const size_t lds = stem_lds_wgrad(Ci);
        hipEvent_t pe = conv_prof_begin(6, 2.0 * B * (double)Co * sa.K * (Hi / 2) * (Wi / 2), 2.0 * B * (double)Co * sa.Kp * (Hi / 2) * (Wi / 2),
                                        4.0 * ((double)B * Ci * Hi * Wi + (double)B * Co * (Hi / 2) * (Wi / 2) + (double)Co * sa.K), st);
        if (Ci == 3) hipLaunchKernelGGL((stem_wgrad_kernel<5, false>), dim3(sa.nblocks), dim3(256), lds, st, sa);
        else hipLaunchKernelGGL((stem_wgrad_kernel<10, false>), dim3(sa.nblocks), dim3(256), lds, st, sa);
        conv_prof_end(pe, st);
        DC_CHECK_LAUNCH();
        hipLaunchKernelGGL(stem_wreduce_kernel, dim3(ceil_div(64 * sa.Kp / 4, 16)), dim3(256), 0, st, (const float*)ws, dweight, sa.nblocks, sa.K, sa.Kp);
        DC_CHECK_LAUNCH();
        return DC_OK;
    }
    CgArgs a{};
    a.x = x; a.gy = gy; a.B = B; a.Ci = Ci; a.Co = Co; a.Hi = Hi; a.Wi = Wi; a.Ho = Hi / 2; a.Wo = Wi / 2;
    a.K = Ci * ksize * ksize; a.Kp = a.K;
    a.xbytes = (unsigned)((size_t)B * Ci * Hi * Wi * sizeof(float));
    a.chunks = ceil_div(B * a.Ho * a.Wo, GKC);
    const bool trip = trip_ok(Ci, ksize);
    a.mtiles = ceil_div(Co, 64); a.ntiles = ceil_div(a.K, trip ? 96 : 64);
    a.splits = cg_wsplits(a.mtiles * a.ntiles, a.chunks);
    a.out = a.splits > 1 ? (float*)ws : dweight;
    const dim3 grid(a.mtiles * a.ntiles, a.splits);
    const size_t lds = trip ? (size_t)2 * (64 + 96) * (GKC + RP) * sizeof(float) : cg_lds_wgrad({2, 2});
    hipEvent_t pe = ksize == 3 ? conv_prof_begin(5, 2.0 * (double)B * a.Ho * a.Wo * Co * a.K,
                                                 2.0 * (double)a.chunks * GKC * (double)a.mtiles * 64.0 * (double)a.ntiles * (trip ? 96.0 : 64.0),
                                                 4.0 * ((double)B * Ci * Hi * Wi + (double)B * a.Ho * a.Wo * Co + (double)Co * a.K), st) : nullptr;
    if (trip) hipLaunchKernelGGL(cg_wgrad3_kernel, grid, dim3(256), lds, st, a);
    else if (ksize == 3) hipLaunchKernelGGL((cg_wgrad_kernel<2, 2, 3>), grid, dim3(256), lds, st, a);
    else hipLaunchKernelGGL((cg_wgrad_kernel<2, 2, 7>), grid, dim3(256), lds, st, a);
    conv_prof_end(pe, st);
    DC_CHECK_LAUNCH();
    if (a.splits > 1) {
        const int n = Co * a.K;
        if (n % 4 == 0 && !(((size_t)ws | (size_t)dweight) & 15))      // 16-byte form: same order per element, 4 x the bytes per thread
            hipLaunchKernelGGL(slab_reduce16_kernel<gf4>, dim3(ceil_div(n / 4, 16)), dim3(256), 0, st, (const gf4*)ws, (gf4*)dweight, a.splits, n / 4);
        else
            hipLaunchKernelGGL(slab_reduce16_kernel<float>, dim3(ceil_div(n, 16)), dim3(256), 0, st, (const float*)ws, dweight, a.splits, n);
        DC_CHECK_LAUNCH();
    }
    return DC_OK;
}

// ---- the stem on the RAW frames: normalisation and temporal pair concat folded into the patch loader -------------------------
static int stem_frames(StemArgs& sa, const float* const* frames, int nf, float mean, float stdv, int Bf, int Hi, int Wi, int Co) {
    if (!frames || (nf != 1 && nf != 3) || !(stdv > 0.f) || Bf <= 0) return DC_EINVAL;
    const int B = nf == 3 ? 2 * Bf : Bf, Ci = nf == 3 ? 6 : 3;
    if (!cg_ok(B, Ci, Co, Hi, Wi, 7) || !stem_ok(Ci, Co, 7)) return DC_EINVAL;
    for (int i = 0; i < nf; ++i)
        if (!frames[i]) return DC_EINVAL;
    sa = stem_args(B, Ci, Co, Hi, Wi);
    sa.nf = nf; sa.Bf = Bf; sa.mean = mean; sa.stdv = stdv;
    for (int i = 0; i < 3; ++i) sa.f[i] = frames[i < nf ? i : 0];
    return DC_OK;
}

extern "C" int dc_stem_supported(int nf, int Bf, int Co, int Hi, int Wi) {
    if ((nf != 1 && nf != 3) || Bf <= 0) return 0;
    const int B = nf == 3 ? 2 * Bf : Bf, Ci = nf == 3 ? 6 : 3;
    return (cg_ok(B, Ci, Co, Hi, Wi, 7) && stem_ok(Ci, Co, 7)) ? 1 : 0;
}

extern "C" int dc_stem_fwd(const float* const* frames, int nf, float mean, float stdv, const float* weight, float* y, void* ws, int Bf,
                           int Hi, int Wi, int Co, void* stream) {
    StemArgs sa{};
    if (!weight || !y || !ws || stem_frames(sa, frames, nf, mean, stdv, Bf, Hi, Wi, Co) != DC_OK) return DC_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(cg_wpad_kernel, dim3(ceil_div(Co * sa.Kp, 256)), dim3(256), 0, st, weight, (float*)ws, Co, sa.K, sa.Kp);
    DC_CHECK_LAUNCH();
    sa.w = (const float*)ws; sa.out = y;
    const double npix = (double)sa.B * (Hi / 2) * (Wi / 2);
    hipEvent_t pe = conv_prof_begin(6, 2.0 * npix * Co * sa.K, 2.0 * npix * Co * sa.Kp,
                                    4.0 * ((double)sa.B * sa.Ci * Hi * Wi + npix * Co + (double)Co * sa.K), st);
    if (stem_x3_enabled()) {
        const int rc = stem_fwd_x3_launch<true>(sa, ws, st);
        if (rc != DC_OK) return rc;
    } else {
        hipLaunchKernelGGL(stem_fwd_kernel<true>, dim3(sa.ntiles, Co / 64), dim3(256), stem_lds_fwd(sa.Ci, sa.Kp), st, sa);
    }
    conv_prof_end(pe, st);
    DC_CHECK_LAUNCH();
    return DC_OK;
}

extern "C" int dc_stem_wgrad(const float* const* frames, int nf, float mean, float stdv, const float* gy, float* dweight, void* ws,
                             int Bf, int Hi, int Wi, int Co, void* stream) {
    StemArgs sa{};
    if (!gy || !dweight || !ws || stem_frames(sa, frames, nf, mean, stdv, Bf, Hi, Wi, Co) != DC_OK) return DC_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    sa.gy = gy; sa.out = (float*)ws;
    const size_t lds = stem_lds_wgrad(sa.Ci);
    const double npix = (double)sa.B * (Hi / 2) * (Wi / 2);
    hipEvent_t pe = conv_prof_begin(6, 2.0 * npix * Co * sa.K, 2.0 * npix * Co * sa.Kp,
                                    4.0 * ((double)sa.B * sa.Ci * Hi * Wi + npix * Co + (double)Co * sa.K), st);
    if (sa.Ci == 3) hipLaunchKernelGGL((stem_wgrad_kernel<5, true>), dim3(sa.nblocks), dim3(256), lds, st, sa);
    else hipLaunchKernelGGL((stem_wgrad_kernel<10, true>), dim3(sa.nblocks), dim3(256), lds, st, sa);
    conv_prof_end(pe, st);
    DC_CHECK_LAUNCH();
    hipLaunchKernelGGL(stem_wreduce_kernel, dim3(ceil_div(64 * sa.Kp / 4, 16)), dim3(256), 0, st, (const float*)ws, dweight, sa.nblocks, sa.K, sa.Kp);
    DC_CHECK_LAUNCH();
    return DC_OK;
}

// The data gradient's grid: 64 x 64 tiles (both column parities in registers: 2 x (MT x NT) accumulator tiles) x 2 row
// parities, and -- on the deep layers, where that is fewer blocks than the chip has slots and each block walks up to 96
// chunks -- the output channels split into 2 or 4 reduction slabs summed in fixed order.  (64 x 128 tiles were measured
// slower on every trunk shape: the row-parity-1 blocks carry 6 taps against 3, so fewer, larger blocks only lengthen the
// critical path; tools/bench_convs2.py.)
static int cg_dgrad3_splits(int B, int Ci, int Co, int Hi, int Wi) {
    const long blocks = (long)ceil_div(Ci, 64) * ceil_div(B * (Hi / 2) * (Wi / 2), 64) * 2;
    int s = 1;
    while (s < 4 && blocks * s * 2 <= 768 && (Co / GKC) % (s * 2) == 0 && (Co / GKC) / (s * 2) >= 2) s *= 2;
    if (const char* f = getenv("DC_DGRAD3_SPLIT")) { const int v = atoi(f); if ((v == 1 || v == 2 || v == 4) && (Co / GKC) % v == 0) s = v; }   // experiments
    return s;
}

extern "C" size_t dc_convs2_dgrad_workspace(int B, int Ci, int Co, int Hi, int Wi, int ksize) {
    if (!cg_ok(B, Ci, Co, Hi, Wi, ksize) || ksize != 3 || (Ci & 3) || Co % GKC) return 0;
    const int sp = cg_dgrad3_splits(B, Ci, Co, Hi, Wi);
    const size_t wt = ((size_t)9 * Co * Ci * sizeof(float) + 255) & ~(size_t)255;
    return std::max(wt + (sp > 1 ? (size_t)sp * B * Ci * Hi * Wi * sizeof(float) : 0), c3b_weights_bytes(Ci, Co));
}

extern "C" int dc_convs2_dgrad(const float* gy, const float* weight, float* dx, void* ws, int B, int Ci, int Co, int Hi, int Wi,
                               int ksize, void* stream) {
    if (!gy || !weight || !dx || !ws || !dc_convs2_dgrad_workspace(B, Ci, Co, Hi, Wi, ksize)) return DC_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    // bf16 matrix cores: stride-1 convolution of the DILATED g' with the rotated, transposed filter (conv_bf16.hip).  Three
    // quarters of its multiplies meet a structural zero -- at 16x the fp32 matrix rate the kernel is bound by its HBM reads
    if (matrix_precision() == DC_PREC_BF16 && c3b_eligible(Co, 0, 1, Hi, Wi, 1))
        return c3b_conv(gy, Co, 3, nullptr, 0, weight, Co, Ci, 1, 0, nullptr, dx, ws, B, Hi, Wi, ACT_NONE, PAD_ZERO, 1, st);
    hipLaunchKernelGGL(cg_wt3_kernel, dim3(ceil_div(9 * Co * Ci, 256)), dim3(256), 0, st, weight, (float*)ws, Co, Ci);
    DC_CHECK_LAUNCH();
    CgArgs a{};
    a.w = (const float*)ws; a.gy = gy; a.out = dx; a.B = B; a.Ci = Ci; a.Co = Co; a.Hi = Hi; a.Wi = Wi; a.Ho = Hi / 2; a.Wo = Wi / 2;
    const int N = B * a.Ho * a.Wo;
    const int sp = cg_dgrad3_splits(B, Ci, Co, Hi, Wi);
    float* slabs = (float*)((char*)ws + (((size_t)9 * Co * Ci * sizeof(float) + 255) & ~(size_t)255));
    if (sp > 1) a.out = slabs;
    const CgTile t{2, 2};
    a.mtiles = ceil_div(Ci, 32 * t.mt); a.ntiles = ceil_div(N, 32 * t.nt);
    const dim3 grid(a.mtiles * a.ntiles, 2, sp);
    const size_t lds = cg_lds_dgrad(t);
    // (split by output parity: no multiply meets a structural zero -- executed = the padded tiles of the same 2 MAC count)
    hipEvent_t pe = conv_prof_begin(5, 2.0 * (double)N * Co * Ci * 9.0, 2.0 * (double)a.mtiles * 64.0 * (double)a.ntiles * 64.0 * Co * 9.0,
                                    4.0 * ((double)B * Ci * Hi * Wi + (double)N * Co + 9.0 * Co * Ci), st);
    hipLaunchKernelGGL((cg_dgrad3_kernel<2, 2>), grid, dim3(256), lds, st, a);
    conv_prof_end(pe, st);
    if (sp > 1) {
        DC_CHECK_LAUNCH();
        const size_t n4 = (size_t)B * Ci * Hi * Wi / 4;
        hipLaunchKernelGGL(cg_slabsum_kernel, dim3((unsigned)std::min<size_t>((n4 + 255) / 256, 4096)), dim3(256), 0, st, (const float*)slabs, dx, n4, sp);
    }
    DC_CHECK_LAUNCH();
    return DC_OK;
}
