// 1x1 convolutions of the ResNet trunks (stride 1 or 2, no bias: the `downsample` branches and the Bottleneck
// conv1 / conv3, reference networks/resnet_encoder.py:74-98 via torchvision) as fp32-MFMA GEMMs straight on the NCHW
// tensors.  The library runs these small problems (0.2-0.4 GMAC) through NHWC implicit-GEMM kernels wrapped in
// NCHW<->NHWC transposes and zero-fills that cost more than the GEMM; here the stride-2 gather, the scatter of the data
// gradient (with its zeros) and the split reduction of the weight gradient are part of the kernels.
//   forward        y[b,m,p]          = sum_k w[m,k] * x[b,k,s*py,s*px]
//   data gradient  dx[b,k,s*py,s*px] = sum_m w[m,k] * gy[b,m,p]      (0 at the positions the stride skips)
//   weight grad    dw[m,k]           = sum_{b,p} gy[b,m,p] * x[b,k,s*py,s*px]   split over blocks, fixed-order reduce
// v_mfma_f32_16x16x4_f32, block = 4 waves = 64 outputs x 64 columns, reduction staged through LDS in chunks of 16.
//
// These are the GENERAL kernels (any shape, scalar staging).  Shapes whose pixel count and channel count allow 16-byte
// staging -- every Bottleneck / downsample / pose-decoder convolution at the BASELINE sizes -- are dispatched to the tiled
// kernels of gemm1x1.hip by the entry points at the bottom of this file.
#include "dc_common.h"
#include "gemm1x1.h"

#include <algorithm>

namespace dc {

using pf4 = __attribute__((ext_vector_type(4))) float;

constexpr int PT = 64;                 // block tile (both GEMM output dimensions)
constexpr int PK = 16;                 // reduction chunk
constexpr int PLS = PT + 4;            // LDS row stride (floats)

struct PwArgs {
    const float* a;    // forward: w (M x K);  dgrad: w (red x out = M x K);  wgrad: gy
    const float* b;    // forward: x;          dgrad: gy;                      wgrad: x
    float* out;
    int B, M, K;       // conv channels: M = Co, K = Ci
    int Hi, Wi, Ho, Wo, s;
    int splits;        // wgrad
    const float* bias; // forward epilogue: y = act(y + bias[m])   (null: none)
    int act;
};

// ---- shared MFMA core: acc[nt] += A(16 rows of this wave x PK) * B(PK x 64 columns)
// LDS images: At[PK][PLS] (reduction-major, output row fastest), Bt[PK][PLS] (reduction-major, column fastest)
__device__ __forceinline__ void pw_mma(const float (*At)[PLS], const float (*Bt)[PLS], int wave, int lane, pf4 acc[4]) {
#pragma unroll
    for (int ks = 0; ks < PK / 4; ++ks) {
        const float av = At[ks * 4 + (lane >> 4)][wave * 16 + (lane & 15)];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const float bv = Bt[ks * 4 + (lane >> 4)][nt * 16 + (lane & 15)];
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[nt], 0, 0, 0);
        }
    }
}

// forward: grid (ceil(P/64), ceil(M/64), B);  rows = output channels m, columns = output pixels p, reduction = k
__global__ __launch_bounds__(256) void pw_fwd_kernel(PwArgs a) {
    __shared__ float At[PK][PLS], Bt[PK][PLS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p0 = blockIdx.x * PT, m0 = blockIdx.y * PT, b = blockIdx.z;
    const int P = a.Ho * a.Wo;
    pf4 acc[4] = {pf4{0, 0, 0, 0}, pf4{0, 0, 0, 0}, pf4{0, 0, 0, 0}, pf4{0, 0, 0, 0}};
    // staging roles
    const int am = tid >> 2, akq = tid & 3;                 // A: w[m0+am][k0 + 4 akq .. +3]
    const int bk = tid >> 4, bpq = tid & 15;                // B: x[k0+bk][pixels p0 + 4 bpq .. +3]
    size_t boff[4]; bool bok[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int p = p0 + 4 * bpq + j;
        bok[j] = p < P;
        const int py = bok[j] ? p / a.Wo : 0, px = bok[j] ? p - py * a.Wo : 0;
        boff[j] = (size_t)(py * a.s) * a.Wi + (size_t)px * a.s;
    }
    const size_t plane = (size_t)a.Hi * a.Wi;
    for (int k0 = 0; k0 < a.K; k0 += PK) {
        float av[4] = {0.f, 0.f, 0.f, 0.f}, bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (m0 + am < a.M) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (k0 + 4 * akq + j < a.K) av[j] = a.a[(size_t)(m0 + am) * a.K + k0 + 4 * akq + j];
        }
        if (k0 + bk < a.K) {
            const float* src = a.b + ((size_t)b * a.K + k0 + bk) * plane;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (bok[j]) bv[j] = src[boff[j]];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            At[4 * akq + j][am] = av[j];
            Bt[bk][4 * bpq + j] = bv[j];
        }
        __syncthreads();
        pw_mma(At, Bt, wave, lane, acc);
    }
    // D layout: row = (lane>>4)*4 + r, column = lane&15
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + wave * 16 + (lane >> 4) * 4 + r, p = p0 + nt * 16 + (lane & 15);
            if (m < a.M && p < P)
                a.out[((size_t)b * a.M + m) * P + p] = act_fwd(acc[nt][r] + (a.bias ? a.bias[m] : 0.f), a.act);
        }
}

// data gradient: grid (ceil(P/64), ceil(K/64), B);  rows = input channels k, columns = pixels p of gy, reduction = m.
// Writes every element of dx: the value at (s*py, s*px), zeros at the positions the stride skips (Hi, Wi multiples of s).
__global__ __launch_bounds__(256) void pw_dgrad_kernel(PwArgs a) {
    __shared__ float At[PK][PLS], Bt[PK][PLS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p0 = blockIdx.x * PT, c0 = blockIdx.y * PT, b = blockIdx.z;
    const int P = a.Ho * a.Wo;
    pf4 acc[4] = {pf4{0, 0, 0, 0}, pf4{0, 0, 0, 0}, pf4{0, 0, 0, 0}, pf4{0, 0, 0, 0}};
    const int ar = tid >> 4, aoq = tid & 15;                // A: w[m0+ar][c0 + 4 aoq .. +3]  (reduction m, output k)
    const int br = tid >> 4, bpq = tid & 15;                // B: gy[m0+br][p0 + 4 bpq .. +3]
    for (int r0 = 0; r0 < a.M; r0 += PK) {
        float av[4] = {0.f, 0.f, 0.f, 0.f}, bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (r0 + ar < a.M) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (c0 + 4 * aoq + j < a.K) av[j] = a.a[(size_t)(r0 + ar) * a.K + c0 + 4 * aoq + j];
                if (p0 + 4 * bpq + j < P) bv[j] = a.b[((size_t)b * a.M + r0 + br) * P + p0 + 4 * bpq + j];
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            At[ar][4 * aoq + j] = av[j];
            Bt[br][4 * bpq + j] = bv[j];
        }
        __syncthreads();
        pw_mma(At, Bt, wave, lane, acc);
    }
    const size_t plane = (size_t)a.Hi * a.Wi;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = c0 + wave * 16 + (lane >> 4) * 4 + r, p = p0 + nt * 16 + (lane & 15);
            if (c >= a.K || p >= P) continue;
            const int py = p / a.Wo, px = p - py * a.Wo;
            float* dst = a.out + ((size_t)b * a.K + c) * plane + (size_t)(py * a.s) * a.Wi + (size_t)px * a.s;
            if (a.s == 1) {
                dst[0] = acc[nt][r];
            } else {                                            // s == 2: the 2x2 cell of this output pixel
                dst[0] = acc[nt][r]; dst[1] = 0.f;
                dst[a.Wi] = 0.f; dst[a.Wi + 1] = 0.f;
            }
        }
}

// weight gradient: grid (splits, ceil(M/64), ceil(K/64));  rows = m, columns = k, reduction = (b, p) in chunks of 16
// pixels of one image; block `split` takes chunks split, split + splits, ... and writes its partial to slab[split][M][K].
__global__ __launch_bounds__(256) void pw_wgrad_kernel(PwArgs a) {
    __shared__ float At[PK][PLS], Bt[PK][PLS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * PT, c0 = blockIdx.z * PT;
    const int P = a.Ho * a.Wo;
    const int cpi = (P + PK - 1) / PK, nchunks = a.B * cpi;      // chunks per image / total
    pf4 acc[4] = {pf4{0, 0, 0, 0}, pf4{0, 0, 0, 0}, pf4{0, 0, 0, 0}, pf4{0, 0, 0, 0}};
    const int rr = tid >> 2, pq = tid & 3;                  // A: gy[m0+rr][pc + 4 pq .. +3];  B: x[c0+rr][same pixels]
    const size_t plane = (size_t)a.Hi * a.Wi;
    for (int ch = blockIdx.x; ch < nchunks; ch += a.splits) {
        const int b = ch / cpi, pc = (ch - b * cpi) * PK;
        float av[4] = {0.f, 0.f, 0.f, 0.f}, bv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int p = pc + 4 * pq + j;
            if (p < P) {
                if (m0 + rr < a.M) av[j] = a.a[((size_t)b * a.M + m0 + rr) * P + p];
                if (c0 + rr < a.K) {
                    const int py = p / a.Wo, px = p - py * a.Wo;
                    bv[j] = a.b[((size_t)b * a.K + c0 + rr) * plane + (size_t)(py * a.s) * a.Wi + (size_t)px * a.s];
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            At[4 * pq + j][rr] = av[j];
            Bt[4 * pq + j][rr] = bv[j];
        }
        __syncthreads();
        pw_mma(At, Bt, wave, lane, acc);
    }
    float* slab = a.out + (size_t)blockIdx.x * a.M * a.K;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + wave * 16 + (lane >> 4) * 4 + r, c = c0 + nt * 16 + (lane & 15);
            if (m < a.M && c < a.K) slab[(size_t)m * a.K + c] = acc[nt][r];
        }
}

// dw = sum of the slabs, fixed order: 16 split groups x 16 outputs per block, groups combined in order through LDS
__global__ __launch_bounds__(256) void pw_wreduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int splits, int n) {
    __shared__ float sm[16][17];
    const int o = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + o;
    const int per = (splits + 15) / 16;
    float v = 0.f;
    if (i < n) {
        const int s1 = min(splits, (grp + 1) * per);
        for (int s = grp * per; s < s1; ++s) v += slab[(size_t)s * n + i];
    }
    sm[grp][o] = v;
    __syncthreads();
    if (grp == 0 && i < n) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += sm[k][o];
        dw[i] = t;
    }
}

static int pw_splits(int B, int Ho, int Wo, int M, int K) {
    const int nchunks = B * ceil_div(Ho * Wo, PK);
    const int outer = ceil_div(M, PT) * ceil_div(K, PT);
    return std::max(1, std::min({nchunks / 4 + 1, ceil_div(1024, outer), 256}));
}

}  // namespace dc

using namespace dc;

static bool pw_shape_ok(int B, int Ci, int Co, int Hi, int Wi, int s) {
    return B > 0 && Ci > 0 && Co > 0 && Hi > 0 && Wi > 0 && (s == 1 || (s == 2 && !(Hi & 1) && !(Wi & 1)));
}

extern "C" size_t dc_conv1x1_wgrad_workspace(int B, int Ci, int Co, int Hi, int Wi, int stride) {
    if (!pw_shape_ok(B, Ci, Co, Hi, Wi, stride)) return 0;
    if (dc_gemm1x1_wgrad_ok(B, Ci, Co, Hi, Wi, stride)) return dc_gemm1x1_wgrad_workspace(B, Ci, Co, Hi, Wi, stride);
    return (size_t)pw_splits(B, Hi / stride, Wi / stride, Co, Ci) * Co * Ci * sizeof(float);
}

extern "C" int dc_conv1x1_bias_act_fwd(const float* x, const float* weight, const float* bias, float* y, int B, int Ci, int Co, int Hi,
                                       int Wi, int stride, int act, void* stream) {
    if (!x || !weight || !y || !pw_shape_ok(B, Ci, Co, Hi, Wi, stride) || act < 0 || act > ACT_LAST) return DC_EINVAL;
    if (dc_gemm1x1_fwd_ok(B, Ci, Co, Hi, Wi, stride))
        return dc_gemm1x1_fwd(x, weight, bias, y, B, Ci, Co, Hi, Wi, stride, act, nullptr, stream);
    PwArgs a{};
    a.bias = bias; a.act = act;
    a.a = weight; a.b = x; a.out = y; a.B = B; a.M = Co; a.K = Ci; a.Hi = Hi; a.Wi = Wi; a.Ho = Hi / stride; a.Wo = Wi / stride; a.s = stride;
    hipLaunchKernelGGL(pw_fwd_kernel, dim3(ceil_div(a.Ho * a.Wo, PT), ceil_div(Co, PT), B), dim3(256), 0, (hipStream_t)stream, a);
    DC_CHECK_LAUNCH();
    return DC_OK;
}

extern "C" int dc_conv1x1_fwd(const float* x, const float* weight, float* y, int B, int Ci, int Co, int Hi, int Wi, int stride,
                              void* stream) {
    return dc_conv1x1_bias_act_fwd(x, weight, nullptr, y, B, Ci, Co, Hi, Wi, stride, ACT_NONE, stream);
}

namespace dc {
__global__ __launch_bounds__(256) void add_inplace_kernel(float* __restrict__ y, const float* __restrict__ a, size_t n) {
    const size_t n4 = n / 4;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256ull) {
        float4 v = reinterpret_cast<float4*>(y)[i];
        const float4 t = reinterpret_cast<const float4*>(a)[i];
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        reinterpret_cast<float4*>(y)[i] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) y[n4 * 4 + threadIdx.x] += a[n4 * 4 + threadIdx.x];
}
int add_inplace(float* y, const float* a, size_t n, hipStream_t st) {
    if ((((size_t)y | (size_t)a) & 15) != 0) return DC_EINVAL;
    hipLaunchKernelGGL(add_inplace_kernel, dim3((unsigned)std::min<size_t>((n / 4 + 255) / 256 + 1, 4096)), dim3(256), 0, st, y, a, n);
    DC_CHECK_LAUNCH();
    return DC_OK;
}
}  // namespace dc

extern "C" int dc_conv1x1_dgrad_add(const float* gy, const float* weight, float* dx, const float* addend, int B, int Ci, int Co, int Hi,
                                    int Wi, int stride, void* stream) {
    if (!gy || !weight || !dx || !pw_shape_ok(B, Ci, Co, Hi, Wi, stride)) return DC_EINVAL;
    if (dc_gemm1x1_dgrad_ok(B, Ci, Co, Hi, Wi, stride)) {
        // the tiled GEMM adds it in its store epilogue (every element of dx is written once, densely, at either stride)
        return dc_gemm1x1_dgrad(gy, weight, dx, addend, nullptr, B, Ci, Co, Hi, Wi, stride, nullptr, stream);
    }
    PwArgs a{};
    a.a = weight; a.b = gy; a.out = dx; a.B = B; a.M = Co; a.K = Ci; a.Hi = Hi; a.Wi = Wi; a.Ho = Hi / stride; a.Wo = Wi / stride; a.s = stride;
    hipLaunchKernelGGL(pw_dgrad_kernel, dim3(ceil_div(a.Ho * a.Wo, PT), ceil_div(Ci, PT), B), dim3(256), 0, (hipStream_t)stream, a);
    DC_CHECK_LAUNCH();
    return addend ? add_inplace(dx, addend, (size_t)B * Ci * Hi * Wi, (hipStream_t)stream) : DC_OK;
}

extern "C" int dc_conv1x1_dgrad(const float* gy, const float* weight, float* dx, int B, int Ci, int Co, int Hi, int Wi, int stride,
                                void* stream) {
    return dc_conv1x1_dgrad_add(gy, weight, dx, nullptr, B, Ci, Co, Hi, Wi, stride, stream);
}

extern "C" int dc_conv1x1_wgrad(const float* x, const float* gy, float* dweight, void* ws, int B, int Ci, int Co, int Hi, int Wi,
                                int stride, void* stream) {
    if (!x || !gy || !dweight || !ws || !pw_shape_ok(B, Ci, Co, Hi, Wi, stride)) return DC_EINVAL;
    if (dc_gemm1x1_wgrad_ok(B, Ci, Co, Hi, Wi, stride))
        return dc_gemm1x1_wgrad(x, gy, dweight, ws, B, Ci, Co, Hi, Wi, stride, nullptr, stream);
    PwArgs a{};
    a.a = gy; a.b = x; a.out = (float*)ws; a.B = B; a.M = Co; a.K = Ci; a.Hi = Hi; a.Wi = Wi; a.Ho = Hi / stride; a.Wo = Wi / stride; a.s = stride;
    a.splits = pw_splits(B, a.Ho, a.Wo, Co, Ci);
    hipLaunchKernelGGL(pw_wgrad_kernel, dim3(a.splits, ceil_div(Co, PT), ceil_div(Ci, PT)), dim3(256), 0, (hipStream_t)stream, a);
    DC_CHECK_LAUNCH();
    hipLaunchKernelGGL(pw_wreduce_kernel, dim3(ceil_div(Co * Ci, 16)), dim3(256), 0, (hipStream_t)stream, (const float*)ws, dweight,
                       a.splits, Co * Ci);
    DC_CHECK_LAUNCH();
    return DC_OK;
}

// ---- the 1x1 convolutions with a BatchNorm folded in (dc_bn_fold): tiled kernels only ----------------------------------------
static bool bn_active(const dc_bn_fold* bn) { return bn && (bn->in_scale || bn->stat_part || bn->bwd_part); }
extern "C" int dc_conv1x1_bn_ok(int B, int Ci, int Co, int Hi, int Wi) {
    return dc_gemm1x1_fwd_ok(B, Ci, Co, Hi, Wi, 1) && dc_gemm1x1_dgrad_ok(B, Ci, Co, Hi, Wi, 1) && dc_gemm1x1_wgrad_ok(B, Ci, Co, Hi, Wi, 1);
}
extern "C" int dc_conv1x1_stat_parts(int B, int Ci, int Co, int Hi, int Wi, int stride, int groups, int* ppg) {
    return dc_gemm1x1_stat_parts(B, Ci, Co, Hi, Wi, stride, groups, ppg);
}
extern "C" int dc_conv1x1_bwd_parts(int B, int Ci, int Co, int Hi, int Wi, int groups, int* ppg) {
    return dc_gemm1x1_bwd_parts(B, Ci, Co, Hi, Wi, 1, groups, ppg);
}
extern "C" int dc_conv1x1_fwd_bn(const float* x, const float* weight, float* y, int B, int Ci, int Co, int Hi, int Wi, int stride,
                                 const dc_bn_fold* bn, void* stream) {
    if (!bn_active(bn)) return dc_conv1x1_fwd(x, weight, y, B, Ci, Co, Hi, Wi, stride, stream);
    if (!dc_gemm1x1_fwd_ok(B, Ci, Co, Hi, Wi, stride)) return DC_EINVAL;
    return dc_gemm1x1_fwd(x, weight, nullptr, y, B, Ci, Co, Hi, Wi, stride, ACT_NONE, bn, stream);
}
extern "C" int dc_conv1x1_dgrad_bn(const float* gy, const float* weight, float* dx, const float* addend, int B, int Ci, int Co, int Hi,
                                   int Wi, int stride, const dc_bn_fold* bn, void* stream) {
    if (!bn || !bn->bwd_part) return dc_conv1x1_dgrad_add(gy, weight, dx, addend, B, Ci, Co, Hi, Wi, stride, stream);
    if (stride != 1 || !dc_gemm1x1_dgrad_ok(B, Ci, Co, Hi, Wi, stride)) return DC_EINVAL;
    return dc_gemm1x1_dgrad(gy, weight, dx, addend, nullptr, B, Ci, Co, Hi, Wi, stride, bn, stream);
}

// dx = data gradient + addend + addend2 (both nullable): an input with up to three consumers, summed in one store epilogue
extern "C" int dc_conv1x1_dgrad_add2(const float* gy, const float* weight, float* dx, const float* addend, const float* addend2, int B,
                                     int Ci, int Co, int Hi, int Wi, int stride, void* stream) {
    if (!addend2) return dc_conv1x1_dgrad_add(gy, weight, dx, addend, B, Ci, Co, Hi, Wi, stride, stream);
    if (!gy || !weight || !dx || !pw_shape_ok(B, Ci, Co, Hi, Wi, stride)) return DC_EINVAL;
    if (dc_gemm1x1_dgrad_ok(B, Ci, Co, Hi, Wi, stride))
        return dc_gemm1x1_dgrad(gy, weight, dx, addend, addend2, B, Ci, Co, Hi, Wi, stride, nullptr, stream);
    const int rc = dc_conv1x1_dgrad_add(gy, weight, dx, addend, B, Ci, Co, Hi, Wi, stride, stream);
    return rc != DC_OK ? rc : add_inplace(dx, addend2, (size_t)B * Ci * Hi * Wi, (hipStream_t)stream);
}
extern "C" int dc_conv1x1_wgrad_bn(const float* x, const float* gy, float* dweight, void* ws, int B, int Ci, int Co, int Hi, int Wi,
                                   int stride, const dc_bn_fold* bn, void* stream) {
    if (!bn || !bn->in_scale) return dc_conv1x1_wgrad(x, gy, dweight, ws, B, Ci, Co, Hi, Wi, stride, stream);
    if (!dc_gemm1x1_wgrad_ok(B, Ci, Co, Hi, Wi, stride)) return DC_EINVAL;
    return dc_gemm1x1_wgrad(x, gy, dweight, ws, B, Ci, Co, Hi, Wi, stride, bn, stream);
}
