// Direct convolution for the shapes NONE of the tiled kernels take (odd or tiny maps, a 1x1 / 2 on an odd map, a 7x7 stem whose
// input needs a gradient): nn.Conv2d semantics -- square kernel, symmetric stride and zero padding, no groups, no dilation
// (every convolution of reference networks/resnet_encoder.py:74-98 via torchvision has this form) -- so that no shape of the
// trunk ever reaches the framework's convolution on the GPU.  Not a fast path: one thread per output element, plain fp32
// FMAs, scalar weight loads; these shapes occur in tests and odd-sized crops, never in a BASELINE configuration.
// Deterministic (fixed summation order, no atomics).
#include "dc_common.h"

namespace dc {

struct CdArgs {
    const float* x; const float* w; const float* gy; const float* bias;
    float* out; float* dbias;
    int B, Ci, Co, Hi, Wi, Ho, Wo, k, s, p;
};

// y[b,co,oy,ox] = bias[co] + sum_{ci,ky,kx} w[co,ci,ky,kx] * x[b,ci,oy*s-p+ky,ox*s-p+kx].   grid (pixel blocks, Co, B)
__global__ __launch_bounds__(256) void cd_fwd_kernel(CdArgs a) {
    const int o = blockIdx.x * 256 + threadIdx.x, co = blockIdx.y, b = blockIdx.z;
    if (o >= a.Ho * a.Wo) return;
    const int oy = o / a.Wo, ox = o - oy * a.Wo;
    float acc = a.bias ? a.bias[co] : 0.f;
    const float* wp = a.w + (size_t)co * a.Ci * a.k * a.k;
    for (int ci = 0; ci < a.Ci; ++ci) {
        const float* xp = a.x + ((size_t)b * a.Ci + ci) * a.Hi * a.Wi;
        for (int ky = 0; ky < a.k; ++ky) {
            const int iy = oy * a.s - a.p + ky;
            if (iy < 0 || iy >= a.Hi) continue;
            for (int kx = 0; kx < a.k; ++kx) {
                const int ix = ox * a.s - a.p + kx;
                if (ix >= 0 && ix < a.Wi) acc = fmaf(wp[(ci * a.k + ky) * a.k + kx], xp[(size_t)iy * a.Wi + ix], acc);
            }
        }
    }
    a.out[((size_t)b * a.Co + co) * a.Ho * a.Wo + o] = acc;
}

// dx[b,ci,iy,ix] = sum_{co,ky,kx : (iy+p-ky) % s == 0, (ix+p-kx) % s == 0} w[co,ci,ky,kx] * gy[b,co,(iy+p-ky)/s,(ix+p-kx)/s]
// grid (pixel blocks, Ci, B)
__global__ __launch_bounds__(256) void cd_dgrad_kernel(CdArgs a) {
    const int i = blockIdx.x * 256 + threadIdx.x, ci = blockIdx.y, b = blockIdx.z;
    if (i >= a.Hi * a.Wi) return;
    const int iy = i / a.Wi, ix = i - iy * a.Wi;
    float acc = 0.f;
    for (int co = 0; co < a.Co; ++co) {
        const float* wp = a.w + ((size_t)co * a.Ci + ci) * a.k * a.k;
        const float* gp = a.gy + ((size_t)b * a.Co + co) * a.Ho * a.Wo;
        for (int ky = 0; ky < a.k; ++ky) {
            const int ty = iy + a.p - ky;
            if (ty < 0 || ty % a.s) continue;
            const int oy = ty / a.s;
            if (oy >= a.Ho) continue;
            for (int kx = 0; kx < a.k; ++kx) {
                const int tx = ix + a.p - kx;
                if (tx < 0 || tx % a.s) continue;
                const int ox = tx / a.s;
                if (ox < a.Wo) acc = fmaf(wp[ky * a.k + kx], gp[(size_t)oy * a.Wo + ox], acc);
            }
        }
    }
    a.out[((size_t)b * a.Ci + ci) * a.Hi * a.Wi + i] = acc;
}

// dw[co,ci,ky,kx] = sum_{b,oy,ox} gy[b,co,oy,ox] * x[b,ci,oy*s-p+ky,ox*s-p+kx].   grid (k*k, Ci, Co): one block per weight,
// thread-strided partial sums, then a fixed-order tree
__global__ __launch_bounds__(256) void cd_wgrad_kernel(CdArgs a) {
    __shared__ float sm[256];
    const int tap = blockIdx.x, ci = blockIdx.y, co = blockIdx.z;
    const int ky = tap / a.k, kx = tap - ky * a.k;
    const int P = a.Ho * a.Wo;
    float acc = 0.f;
    for (int n = threadIdx.x; n < a.B * P; n += 256) {
        const int b = n / P, o = n - b * P;
        const int oy = o / a.Wo, ox = o - oy * a.Wo;
        const int iy = oy * a.s - a.p + ky, ix = ox * a.s - a.p + kx;
        if (iy >= 0 && iy < a.Hi && ix >= 0 && ix < a.Wi)
            acc = fmaf(a.gy[((size_t)b * a.Co + co) * P + o], a.x[(((size_t)b * a.Ci + ci) * a.Hi + iy) * a.Wi + ix], acc);
    }
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) sm[threadIdx.x] += sm[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) a.out[(((size_t)co * a.Ci + ci) * a.k + ky) * a.k + kx] = sm[0];
}

// dbias[co] = sum_{b,o} gy[b,co,o].   grid (Co)
__global__ __launch_bounds__(256) void cd_dbias_kernel(CdArgs a) {
    __shared__ float sm[256];
    const int co = blockIdx.x, P = a.Ho * a.Wo;
    float acc = 0.f;
    for (int n = threadIdx.x; n < a.B * P; n += 256) {
        const int b = n / P, o = n - b * P;
        acc += a.gy[((size_t)b * a.Co + co) * P + o];
    }
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) sm[threadIdx.x] += sm[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) a.dbias[co] = sm[0];
}

static bool cd_fill(CdArgs& a, int B, int Ci, int Co, int Hi, int Wi, int k, int s, int p) {
    if (B <= 0 || Ci <= 0 || Co <= 0 || Hi <= 0 || Wi <= 0 || k < 1 || k > 11 || s < 1 || s > 4 || p < 0 || p >= k) return false;
    if (Hi + 2 * p < k || Wi + 2 * p < k) return false;
    a.B = B; a.Ci = Ci; a.Co = Co; a.Hi = Hi; a.Wi = Wi; a.k = k; a.s = s; a.p = p;
    a.Ho = (Hi + 2 * p - k) / s + 1; a.Wo = (Wi + 2 * p - k) / s + 1;
    // grid.y / grid.z limits and 32-bit pixel indices
    if (Co > 65535 || Ci > 65535 || B > 65535 || (size_t)Hi * Wi >= (1ull << 31) || (size_t)B * a.Ho * a.Wo >= (1ull << 31)) return false;
    return true;
}

}  // namespace dc

using namespace dc;

extern "C" int dc_conv2d_direct_fwd(const float* x, const float* weight, const float* bias, float* y, int B, int Ci, int Co, int Hi,
                                    int Wi, int ksize, int stride, int pad, void* stream) {
    CdArgs a{};
    if (!x || !weight || !y || !cd_fill(a, B, Ci, Co, Hi, Wi, ksize, stride, pad)) return DC_EINVAL;
    a.x = x; a.w = weight; a.bias = bias; a.out = y;
    hipLaunchKernelGGL(cd_fwd_kernel, dim3(ceil_div(a.Ho * a.Wo, 256), Co, B), dim3(256), 0, (hipStream_t)stream, a);
    DC_CHECK_LAUNCH();
    return DC_OK;
}

extern "C" int dc_conv2d_direct_dgrad(const float* gy, const float* weight, float* dx, int B, int Ci, int Co, int Hi, int Wi, int ksize,
                                      int stride, int pad, void* stream) {
    CdArgs a{};
    if (!gy || !weight || !dx || !cd_fill(a, B, Ci, Co, Hi, Wi, ksize, stride, pad)) return DC_EINVAL;
    a.gy = gy; a.w = weight; a.out = dx;
    hipLaunchKernelGGL(cd_dgrad_kernel, dim3(ceil_div(Hi * Wi, 256), Ci, B), dim3(256), 0, (hipStream_t)stream, a);
    DC_CHECK_LAUNCH();
    return DC_OK;
}

extern "C" int dc_conv2d_direct_wgrad(const float* x, const float* gy, float* dweight, float* dbias, int B, int Ci, int Co, int Hi,
                                      int Wi, int ksize, int stride, int pad, void* stream) {
    CdArgs a{};
    if (!x || !gy || !dweight || !cd_fill(a, B, Ci, Co, Hi, Wi, ksize, stride, pad)) return DC_EINVAL;
    a.x = x; a.gy = gy; a.out = dweight; a.dbias = dbias;
    hipLaunchKernelGGL(cd_wgrad_kernel, dim3(ksize * ksize, Ci, Co), dim3(256), 0, (hipStream_t)stream, a);
    DC_CHECK_LAUNCH();
    if (dbias) {
        hipLaunchKernelGGL(cd_dbias_kernel, dim3(Co), dim3(256), 0, (hipStream_t)stream, a);
        DC_CHECK_LAUNCH();
    }
    return DC_OK;
}
