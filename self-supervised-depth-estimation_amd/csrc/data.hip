// Per-item data step on the device (reference datasets/mono_dataset.py:92-118 `preprocess`, :139-211
// `__getitem__`): horizontal flip, the Lanczos resize pyramid, ColorJitter and ToTensor for a whole batch of decoded frames
// that already sit in HBM as uint8 HWC.  The arithmetic is Pillow's (libImaging Resample.c / Blend.c / Convert.c) and
// torchvision's functional_pil glue, restated so that every byte equals what the reference's CPU workers produce:
//   * resize: two 8-bit passes (horizontal, then vertical), 22-bit fixed-point coefficients computed on the host in double
//     (dc_resample_table), int32 accumulation with rounding, clip to 8 bits after each pass;
//   * ColorJitter: brightness / contrast / saturation = Image.blend(degenerate, image, factor) in single precision with
//     truncation (interpolation) or clip + truncation (extrapolation); contrast needs the integer mean of the L image
//     (exact 64-bit sum); hue = uint8 shift of H between Pillow's float/double RGB<->HSV conversions;
//   * ToTensor: CHW float32, true division by 255.
// HBM-bound byte work: no LDS staging needed beyond what L2 gives the overlapping filter taps; one thread per output pixel.
#include "dc_common.h"

#include <math.h>

#include <algorithm>

#pragma clang fp contract(off)   // Pillow's C is compiled without fused multiply-adds; every rounding below is load-bearing

namespace dc {

constexpr int kPrecisionBits = 32 - 8 - 2;   // Resample.c PRECISION_BITS

__device__ __forceinline__ uint8_t clip8(int acc) {
    const int v = acc >> kPrecisionBits;
    return (uint8_t)min(max(v, 0), 255);
}

// src (n, H, Wi, 3) -> dst (n, H, Wo, 3); flip[img] != 0 reads the row mirrored (FLIP_LEFT_RIGHT before the resize)
__global__ __launch_bounds__(256) void data_resize_x_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int H, int Wi, int Wo,
                                                           const int* __restrict__ bounds, const int* __restrict__ kk, int ksize,
                                                           const uint8_t* __restrict__ flip) {
    const int img = blockIdx.z, y = blockIdx.y, xo = blockIdx.x * 256 + threadIdx.x;
    if (xo >= Wo) return;
    const int x0 = bounds[2 * xo], n = bounds[2 * xo + 1];
    const int* k = kk + (size_t)xo * ksize;
    const uint8_t* row = src + ((size_t)img * H + y) * Wi * 3;
    const bool mirror = flip && flip[img];
    int a0 = 1 << (kPrecisionBits - 1), a1 = a0, a2 = a0;
    for (int j = 0; j < n; ++j) {
        const int sx = mirror ? Wi - 1 - (x0 + j) : x0 + j;
        const uint8_t* p = row + (size_t)sx * 3;
        const int c = k[j];
        a0 += c * p[0];
        a1 += c * p[1];
        a2 += c * p[2];
    }
    uint8_t* o = dst + (((size_t)img * H + y) * Wo + xo) * 3;
    o[0] = clip8(a0);
    o[1] = clip8(a1);
    o[2] = clip8(a2);
}

// src (n, Hi, row) -> dst (n, Ho, row), row = W * 3 bytes; one thread per output byte (coalesced along the row)
__global__ __launch_bounds__(256) void data_resize_y_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int Hi, int Ho, int row,
                                                           const int* __restrict__ bounds, const int* __restrict__ kk, int ksize) {
    const int img = blockIdx.z, yo = blockIdx.y, xb = blockIdx.x * 256 + threadIdx.x;
    if (xb >= row) return;
    const int y0 = bounds[2 * yo], n = bounds[2 * yo + 1];
    const int* k = kk + (size_t)yo * ksize;
    const uint8_t* p = src + ((size_t)img * Hi + y0) * row + xb;
    int a = 1 << (kPrecisionBits - 1);
    for (int j = 0; j < n; ++j) a += k[j] * p[(size_t)j * row];
    dst[((size_t)img * Ho + yo) * row + xb] = clip8(a);
}

// same as above when the horizontal pass is skipped but the image is flipped: plain mirrored copy
__global__ __launch_bounds__(256) void data_flip_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int H, int W,
                                                       const uint8_t* __restrict__ flip) {
    const int img = blockIdx.z, y = blockIdx.y, x = blockIdx.x * 256 + threadIdx.x;
    if (x >= W) return;
    const int sx = (flip && flip[img]) ? W - 1 - x : x;
    const uint8_t* p = src + (((size_t)img * H + y) * W + sx) * 3;
    uint8_t* o = dst + (((size_t)img * H + y) * W + x) * 3;
    o[0] = p[0];
    o[1] = p[1];
    o[2] = p[2];
}

// ------------------------------------------------------------------------------------------------------------- ColorJitter
__device__ __forceinline__ int luma(int r, int g, int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }   // Convert.c L24

// Blend.c: (UINT8)(in1 + alpha * (in2 - in1)), in1 = degenerate; outside [0, 1]: clip, then truncate
__device__ __forceinline__ uint8_t blend(int d, int x, float a, bool interp) {
    float m = a * (float)(x - d);
    asm volatile("" : "+v"(m));   // the product is rounded before the add: HIP's __fmul_rn / __fadd_rn are plain operators that
    const float t = (float)d + m;  // the compiler fuses, and a fused multiply-add differs whenever a * (x - d) is near an integer
    if (interp) return (uint8_t)(int)t;
    return t <= 0.f ? 0 : (t >= 255.f ? 255 : (uint8_t)(int)t);
}

// Convert.c rgb2hsv_row: float variables, double literals (see oracle/data_ref.py)
__device__ __forceinline__ void rgb2hsv(int r, int g, int b, int& uh, int& us, int& uv) {
    const int maxc = max(r, max(g, b)), minc = min(r, min(g, b));
    uv = maxc;
    if (minc == maxc) {
        uh = 0;
        us = 0;
        return;
    }
    const float cr = (float)(maxc - minc);
    const float s = cr / (float)maxc;
    const float rc = (float)(maxc - r) / cr, gc = (float)(maxc - g) / cr, bc = (float)(maxc - b) / cr;
    float h;
    if (r == maxc)
        h = bc - gc;
    else if (g == maxc)
        h = (float)(2.0 + (double)rc - (double)bc);
    else
        h = (float)(4.0 + (double)gc - (double)rc);
    double hd = (double)h / 6.0 + 1.0;          // in [5/6, 11/6]
    hd = hd >= 1.0 ? hd - 1.0 : hd;             // fmod(., 1.0), exact
    h = (float)hd;
    uh = min(max((int)((double)h * 255.0), 0), 255);
    us = min(max((int)((double)s * 255.0), 0), 255);
}

// Convert.c hsv2rgb
__device__ __forceinline__ void hsv2rgb(int h, int s, int v, int& r, int& g, int& b) {
    if (s == 0) {
        r = g = b = v;
        return;
    }
    const double fh = (double)(float)h * 6.0 / 255.0;
    const int i = (int)floor(fh);
    const double f = (double)(float)(fh - (double)(float)i);
    const double fs = (double)(float)((double)(float)s / 255.0);
    const double vf = (double)(float)v;
    const int p = min(max((int)round(vf * (1.0 - fs)), 0), 255);
    const int q = min(max((int)round(vf * (1.0 - fs * f)), 0), 255);
    const int t = min(max((int)round(vf * (1.0 - fs * (1.0 - f))), 0), 255);
    switch (i % 6) {
        case 0: r = v, g = t, b = p; break;
        case 1: r = q, g = v, b = p; break;
        case 2: r = p, g = v, b = t; break;
        case 3: r = p, g = q, b = v; break;
        case 4: r = t, g = p, b = v; break;
        default: r = v, g = p, b = q; break;
    }
}

// exact sum of the L image, only for images whose op at this step is CONTRAST
__global__ __launch_bounds__(256) void data_luma_sum_kernel(const uint8_t* __restrict__ img, int npix, const int* __restrict__ steps, int step,
                                                           unsigned long long* __restrict__ sums) {
    const int im = blockIdx.y;
    if (steps[im * 4 + step] != DC_JITTER_CONTRAST) return;
    const uint8_t* p = img + (size_t)im * npix * 3;
    unsigned int acc = 0;   // <= 255 * ceil(npix / threads) per thread: npix < 2^31 / 255 * threads holds for any image here
    for (int i = blockIdx.x * 256 + threadIdx.x; i < npix; i += gridDim.x * 256) acc += luma(p[3 * i], p[3 * i + 1], p[3 * i + 2]);
    unsigned long long a = acc;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(&sums[im], a);
}

// one ColorJitter operation on one pixel (the arithmetic of the in-place kernel below and of the fused chain kernels)
__device__ __forceinline__ void jitter_op(int& r, int& g, int& b, int op, float a, int mean) {
    if (op == DC_JITTER_HUE) {
        int h, s, v;
        rgb2hsv(r, g, b, h, s, v);
        h = (h + (int)a) & 0xFF;
        hsv2rgb(h, s, v, r, g, b);
    } else if (op >= 0) {
        const bool interp = a >= 0.f && a <= 1.f;
        int d0 = 0;
        if (op == DC_JITTER_CONTRAST) d0 = mean;
        else if (op == DC_JITTER_SATURATION) d0 = luma(r, g, b);
        r = blend(d0, r, a, interp);
        g = blend(d0, g, a, interp);
        b = blend(d0, b, a, interp);
    }
}
__device__ __forceinline__ int luma_mean(unsigned long long sum, int npix) { return (int)((double)sum / (double)npix + 0.5); }   // int(ImageStat mean + 0.5)

__global__ __launch_bounds__(256) void data_jitter_kernel(uint8_t* __restrict__ img, int npix, const int* __restrict__ steps,
                                                         const float* __restrict__ params, int step,
                                                         const unsigned long long* __restrict__ sums) {
    const int im = blockIdx.y;
    const int op = steps[im * 4 + step];
    if (op < 0) return;
    const float a = params[im * 4 + step];
    const int mean = op == DC_JITTER_CONTRAST ? luma_mean(sums[im], npix) : 0;
    uint8_t* base = img + (size_t)im * npix * 3;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < npix; i += gridDim.x * 256) {
        uint8_t* p = base + (size_t)3 * i;
        int r = p[0], g = p[1], b = p[2];
        jitter_op(r, g, b, op, a, mean);
        p[0] = (uint8_t)r, p[1] = (uint8_t)g, p[2] = (uint8_t)b;
    }
}

// ---- the whole chain of an item in two passes over the uint8 image, nothing written in between ----------------------------
// ColorJitter is four per-pixel operations; only contrast needs something global (the mean of the L image AT THAT POINT of
// the chain).  Pass 1 re-applies the operations in front of the contrast step and sums L (exact integers; skipped for an
// image without a contrast step); pass 2 recomputes the whole chain per pixel and writes BOTH float tensors of ToTensor --
// ("color", s) from the untouched pixel and ("color_aug", s) from the jittered one.  Per scale: 2 launches instead of 14.
struct JChain { int op[4]; float a[4]; int c; };
__device__ __forceinline__ JChain jchain_load(const int* steps, const float* params, int im) {
    JChain ch;
    ch.c = -1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        ch.op[k] = steps ? steps[im * 4 + k] : DC_JITTER_NONE;
        ch.a[k] = steps ? params[im * 4 + k] : 0.f;
        if (ch.op[k] == DC_JITTER_CONTRAST && ch.c < 0) ch.c = k;
    }
    return ch;
}
__global__ __launch_bounds__(256) void data_chain_sum_kernel(const uint8_t* __restrict__ img, int npix, const int* __restrict__ steps,
                                                            const float* __restrict__ params, unsigned long long* __restrict__ sums) {
    const int im = blockIdx.y;
    const JChain ch = jchain_load(steps, params, im);
    if (ch.c < 0) return;
    const uint8_t* p = img + (size_t)im * npix * 3;
    unsigned int acc = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < npix; i += gridDim.x * 256) {
        int r = p[3 * i], g = p[3 * i + 1], b = p[3 * i + 2];
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (k < ch.c) jitter_op(r, g, b, ch.op[k], ch.a[k], 0);
        acc += luma(r, g, b);
    }
    unsigned long long a = acc;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(&sums[im], a);
}
__global__ __launch_bounds__(256) void data_chain_tensor_kernel(const uint8_t* __restrict__ img, float* __restrict__ color,
                                                               float* __restrict__ color_aug, int npix, const int* __restrict__ steps,
                                                               const float* __restrict__ params, const unsigned long long* __restrict__ sums) {
    const int im = blockIdx.y;
    const JChain ch = jchain_load(steps, params, im);
    const int mean = ch.c >= 0 ? luma_mean(sums[im], npix) : 0;
    const uint8_t* p = img + (size_t)im * npix * 3;
    float* o = color + (size_t)im * npix * 3;
    float* oa = color_aug ? color_aug + (size_t)im * npix * 3 : nullptr;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < npix; i += gridDim.x * 256) {
        int r = p[3 * i], g = p[3 * i + 1], b = p[3 * i + 2];
        o[i] = (float)r / 255.0f;
        o[(size_t)npix + i] = (float)g / 255.0f;
        o[(size_t)2 * npix + i] = (float)b / 255.0f;
        if (oa) {
#pragma unroll
            for (int k = 0; k < 4; ++k) jitter_op(r, g, b, ch.op[k], ch.a[k], mean);
            oa[i] = (float)r / 255.0f;
            oa[(size_t)npix + i] = (float)g / 255.0f;
            oa[(size_t)2 * npix + i] = (float)b / 255.0f;
        }
    }
}

// (n, H*W, 3) uint8 -> (n, 3, H*W) float32 / 255
__global__ __launch_bounds__(256) void data_to_tensor_kernel(const uint8_t* __restrict__ img, float* __restrict__ out, int npix) {
    const int im = blockIdx.y;
    const uint8_t* p = img + (size_t)im * npix * 3;
    float* o = out + (size_t)im * npix * 3;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < npix; i += gridDim.x * 256) {
        o[i] = (float)p[3 * i] / 255.0f;
        o[(size_t)npix + i] = (float)p[3 * i + 1] / 255.0f;
        o[(size_t)2 * npix + i] = (float)p[3 * i + 2] / 255.0f;
    }
}

// (n, H*W, 3) uint8 -> (n, H*W, 4) float32 / 255, pixel-interleaved RGBx (x = 0): the layout the photometric kernels gather from
__global__ __launch_bounds__(256) void data_to_rgbx_kernel(const uint8_t* __restrict__ img, float4* __restrict__ out, int npix) {
    const int im = blockIdx.y;
    const uint8_t* p = img + (size_t)im * npix * 3;
    float4* o = out + (size_t)im * npix;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < npix; i += gridDim.x * 256)
        o[i] = make_float4((float)p[3 * i] / 255.0f, (float)p[3 * i + 1] / 255.0f, (float)p[3 * i + 2] / 255.0f, 0.f);
}
// (n, 3, H*W) float32 -> (n, H*W, 4) float32 RGBx
__global__ __launch_bounds__(256) void pack_rgbx_kernel(const float* __restrict__ x, float4* __restrict__ out, int npix) {
    const int im = blockIdx.y;
    const float* p = x + (size_t)im * npix * 3;
    float4* o = out + (size_t)im * npix;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < npix; i += gridDim.x * 256)
        o[i] = make_float4(p[i], p[(size_t)npix + i], p[(size_t)2 * npix + i], 0.f);
}

// ---- host: Resample.c precompute_coeffs + normalize_coeffs_8bpc, Lanczos (support 3) over the whole axis ----------------
static inline double sinc_filter(double x) {
    if (x == 0.0) return 1.0;
    x = x * M_PI;
    return sin(x) / x;
}
static inline double lanczos_filter(double x) { return (-3.0 <= x && x < 3.0) ? sinc_filter(x) * sinc_filter(x / 3) : 0.0; }

}  // namespace dc

using namespace dc;

extern "C" int dc_resample_ksize(int in_size, int out_size) {
    if (in_size <= 0 || out_size <= 0) return 0;
    const double scale = (double)in_size / out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    return (int)ceil(3.0 * filterscale) * 2 + 1;
}

extern "C" int dc_resample_table(int in_size, int out_size, int* bounds, int* kk) {
    if (in_size <= 0 || out_size <= 0 || !bounds || !kk) return DC_EINVAL;
    const double scale = (double)in_size / out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 3.0 * filterscale;
    const int ksize = (int)ceil(support) * 2 + 1;
    const double ss = 1.0 / filterscale;
    double* w = new double[ksize];
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = (xx + 0.5) * scale;
        double ww = 0.0;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        for (int x = 0; x < xmax; ++x) {
            w[x] = lanczos_filter((x + xmin - center + 0.5) * ss);
            ww += w[x];
        }
        int* k = kk + (size_t)xx * ksize;
        for (int x = 0; x < ksize; ++x) {
            if (x >= xmax) {
                k[x] = 0;
                continue;
            }
            const double v = ww != 0.0 ? w[x] / ww : w[x];
            k[x] = v < 0 ? (int)(-0.5 + v * (1 << kPrecisionBits)) : (int)(0.5 + v * (1 << kPrecisionBits));
        }
        bounds[2 * xx] = xmin;
        bounds[2 * xx + 1] = xmax;
    }
    delete[] w;
    return DC_OK;
}

static bool img_ok(int n, int H, int W) { return n > 0 && n <= 65535 && H > 0 && H <= 65535 && W > 0 && (size_t)n * H * W * 3 < (1ull << 40); }

extern "C" int dc_data_resize_axis(const uint8_t* src, uint8_t* dst, int n_img, int Hi, int Wi, int out_size, int axis, const int* bounds,
                                   const int* kk, int ksize, const uint8_t* flip, void* stream) {
    if (!src || !dst || !bounds || !kk || !img_ok(n_img, Hi, Wi) || out_size <= 0 || out_size > 65535 || (axis != 0 && axis != 1))
        return DC_EINVAL;
    if (ksize != dc_resample_ksize(axis ? Wi : Hi, out_size)) return DC_EINVAL;
    if (axis == 1) {
        hipLaunchKernelGGL(data_resize_x_kernel, dim3((out_size + 255) / 256, Hi, n_img), dim3(256), 0, (hipStream_t)stream, src, dst, Hi, Wi,
                           out_size, bounds, kk, ksize, flip);
    } else {
        if (flip) return DC_EINVAL;   // the mirror belongs to the horizontal pass (or dc_data_flip)
        const int row = Wi * 3;
        hipLaunchKernelGGL(data_resize_y_kernel, dim3((row + 255) / 256, out_size, n_img), dim3(256), 0, (hipStream_t)stream, src, dst, Hi,
                           out_size, row, bounds, kk, ksize);
    }
    DC_CHECK_LAUNCH();
    return DC_OK;
}

extern "C" int dc_data_flip(const uint8_t* src, uint8_t* dst, int n_img, int H, int W, const uint8_t* flip, void* stream) {
    if (!src || !dst || src == dst || !img_ok(n_img, H, W)) return DC_EINVAL;
    hipLaunchKernelGGL(data_flip_kernel, dim3((W + 255) / 256, H, n_img), dim3(256), 0, (hipStream_t)stream, src, dst, H, W, flip);
    DC_CHECK_LAUNCH();
    return DC_OK;
}

extern "C" int dc_data_jitter(uint8_t* img, int n_img, int npix, const int* steps, const float* params, unsigned long long* sums,
                              void* stream) {
    if (!img || !steps || !params || !sums || n_img <= 0 || n_img > 65535 || npix <= 0 || npix > (1 << 28)) return DC_EINVAL;
    const int bx = std::min((npix + 255) / 256, 1024);
    for (int step = 0; step < 4; ++step) {
        if (hipMemsetAsync(sums, 0, sizeof(unsigned long long) * n_img, (hipStream_t)stream) != hipSuccess) return DC_ELAUNCH;
        hipLaunchKernelGGL(data_luma_sum_kernel, dim3(bx, n_img), dim3(256), 0, (hipStream_t)stream, img, npix, steps, step, sums);
        hipLaunchKernelGGL(data_jitter_kernel, dim3(bx, n_img), dim3(256), 0, (hipStream_t)stream, img, npix, steps, params, step, sums);
    }
    DC_CHECK_LAUNCH();
    return DC_OK;
}

extern "C" int dc_data_jitter_to_tensor(const uint8_t* img, float* color, float* color_aug, int n_img, int npix, const int* steps,
                                        const float* params, unsigned long long* sums, void* stream) {
    if (!img || !color || n_img <= 0 || n_img > 65535 || npix <= 0 || npix > (1 << 28)) return DC_EINVAL;
    if (color_aug && (!steps || !params || !sums)) return DC_EINVAL;
    const int bx = std::min((npix + 255) / 256, 1024);
    if (color_aug) {
        if (hipMemsetAsync(sums, 0, sizeof(unsigned long long) * n_img, (hipStream_t)stream) != hipSuccess) return DC_ELAUNCH;
        hipLaunchKernelGGL(data_chain_sum_kernel, dim3(bx, n_img), dim3(256), 0, (hipStream_t)stream, img, npix, steps, params, sums);
    }
    hipLaunchKernelGGL(data_chain_tensor_kernel, dim3(bx, n_img), dim3(256), 0, (hipStream_t)stream, img, color, color_aug, npix,
                       color_aug ? steps : (const int*)nullptr, params, (const unsigned long long*)sums);
    DC_CHECK_LAUNCH();
    return DC_OK;
}

extern "C" int dc_data_to_tensor(const uint8_t* img, float* out, int n_img, int npix, void* stream) {
    if (!img || !out || n_img <= 0 || n_img > 65535 || npix <= 0 || npix > (1 << 28)) return DC_EINVAL;
    hipLaunchKernelGGL(data_to_tensor_kernel, dim3(std::min((npix + 255) / 256, 1024), n_img), dim3(256), 0, (hipStream_t)stream, img, out,
                       npix);
    DC_CHECK_LAUNCH();
    return DC_OK;
}

extern "C" int dc_data_to_rgbx(const uint8_t* img, float* out, int n_img, int npix, void* stream) {
    if (!img || !out || ((size_t)out & 15) || n_img <= 0 || n_img > 65535 || npix <= 0 || npix > (1 << 28)) return DC_EINVAL;
    hipLaunchKernelGGL(data_to_rgbx_kernel, dim3(std::min((npix + 255) / 256, 1024), n_img), dim3(256), 0, (hipStream_t)stream, img,
                       (float4*)out, npix);
    DC_CHECK_LAUNCH();
    return DC_OK;
}

extern "C" int dc_pack_rgbx(const float* x, float* out, int n_img, int npix, void* stream) {
    if (!x || !out || ((size_t)out & 15) || n_img <= 0 || n_img > 65535 || npix <= 0 || npix > (1 << 28)) return DC_EINVAL;
    hipLaunchKernelGGL(pack_rgbx_kernel, dim3(std::min((npix + 255) / 256, 1024), n_img), dim3(256), 0, (hipStream_t)stream, x,
                       (float4*)out, npix);
    DC_CHECK_LAUNCH();
    return DC_OK;
}
