// Unfused drop-in kernels behind the reference's `layers.py` module API.
// The fast path is the fused photometric pipeline in photo.hip; these exist so that code calling
// `layers.BackprojectDepth`, `Project3D`, `SSIM`, `get_smooth_loss`, ... one by one (as the reference
// trainer does) also runs on hand-written gfx950 kernels.  All are HBM-bound streaming kernels:
// coalesced along W, one pass, no intermediates.
#include "dc_common.h"

namespace dc {

// ------------------------------------------------------------------ a6 disp_to_depth
__global__ void d2d_fwd_kernel(const float* disp, float* scaled, float* depth, size_t n, float lo, float rng) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        float s = lo + rng * disp[i];
        if (scaled) scaled[i] = s;
        if (depth) depth[i] = 1.0f / s;
    }
}
__global__ void d2d_bwd_kernel(const float* disp, const float* gs, const float* gd, float* dd, size_t n, float lo,
                               float rng) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        float s = lo + rng * disp[i];
        float g = gs ? gs[i] : 0.f;
        if (gd) { float dep = 1.0f / s; g -= gd[i] * dep * dep; }
        dd[i] = g * rng;
    }
}

// ------------------------------------------------------------------ a7 BackprojectDepth
__global__ void pix_coords_kernel(float* pc, int B, int H, int W) {
    const int n = H * W;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (i >= n) return;
    float* o = pc + (size_t)b * 3 * n;
    o[i] = (float)(i % W);         // exact small integers (layers.py:150-161)
    o[n + i] = (float)(i / W);
    o[2 * n + i] = 1.0f;
}
__global__ void backproject_fwd_kernel(const float* depth, const float* invK, float* cam, int H, int W) {
    const int n = H * W;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (i >= n) return;
    const float* k = invK + b * 16;
    const float x = (float)(i % W), y = (float)(i / W);
    const float d = depth[(size_t)b * n + i];
    float* o = cam + (size_t)b * 4 * n;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        float ray = k[r * 4 + 0] * x;
        ray = fmaf(k[r * 4 + 1], y, ray);
        ray = ray + k[r * 4 + 2];
        o[(size_t)r * n + i] = d * ray;
    }
    o[(size_t)3 * n + i] = 1.0f;
}
__global__ void backproject_bwd_kernel(const float* gcam, const float* invK, float* dd, int H, int W) {
    const int n = H * W;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (i >= n) return;
    const float* k = invK + b * 16;
    const float x = (float)(i % W), y = (float)(i / W);
    const float* g = gcam + (size_t)b * 4 * n;
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        float ray = k[r * 4 + 0] * x;
        ray = fmaf(k[r * 4 + 1], y, ray);
        ray = ray + k[r * 4 + 2];
        acc += g[(size_t)r * n + i] * ray;
    }
    dd[(size_t)b * n + i] = acc;
}

// ------------------------------------------------------------------ a8 Project3D
__device__ __forceinline__ void load_P(float P[12], const float* K, const float* T, int b) {
    const float* k = K + b * 16;
    const float* t = T + b * 16;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float a = k[i * 4 + 0] * t[0 * 4 + j];
            a = fmaf(k[i * 4 + 1], t[1 * 4 + j], a);
            a = fmaf(k[i * 4 + 2], t[2 * 4 + j], a);
            a = fmaf(k[i * 4 + 3], t[3 * 4 + j], a);
            P[i * 4 + j] = a;
        }
}
__global__ void project3d_fwd_kernel(const float* pts, const float* K, const float* T, float* grid, int H, int W,
                                     float eps) {
    const int n = H * W;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (i >= n) return;
    float P[12];
    load_P(P, K, T, b);
    const float* p = pts + (size_t)b * 4 * n;
    const float p0 = p[i], p1 = p[n + i], p2 = p[2 * n + i], p3 = p[3 * (size_t)n + i];
    float q[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        float a = P[r * 4 + 0] * p0;
        a = fmaf(P[r * 4 + 1], p1, a);
        a = fmaf(P[r * 4 + 2], p2, a);
        q[r] = fmaf(P[r * 4 + 3], p3, a);
    }
    const float z = q[2] + eps;
    float u = (q[0] / z) / (float)(W - 1);
    float v = (q[1] / z) / (float)(H - 1);
    float2 o = make_float2((u - 0.5f) * 2.f, (v - 0.5f) * 2.f);
    reinterpret_cast<float2*>(grid)[(size_t)b * n + i] = o;
}
constexpr int PJ_PIX = 1024;   // pixels per block in the backward
__global__ __launch_bounds__(256) void project3d_bwd_kernel(const float* pts, const float* K, const float* T,
                                                            const float* ggrid, float* dpts, float* part, int H,
                                                            int W, float eps) {
    __shared__ float red[4][12];
    const int n = H * W;
    const int b = blockIdx.y;
    float P[12];
    load_P(P, K, T, b);
    const float* p = pts + (size_t)b * 4 * n;
    float dP[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) dP[k] = 0.f;
    for (int i = blockIdx.x * PJ_PIX + threadIdx.x; i < min((blockIdx.x + 1) * PJ_PIX, n); i += 256) {
        const float ph[4] = {p[i], p[n + i], p[2 * n + i], p[3 * (size_t)n + i]};
        float q[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            float a = P[r * 4 + 0] * ph[0];
            a = fmaf(P[r * 4 + 1], ph[1], a);
            a = fmaf(P[r * 4 + 2], ph[2], a);
            q[r] = fmaf(P[r * 4 + 3], ph[3], a);
        }
        const float zi = 1.0f / (q[2] + eps);
        const float2 gg = reinterpret_cast<const float2*>(ggrid)[(size_t)b * n + i];
        const float du = gg.x * 2.f / (float)(W - 1), dv = gg.y * 2.f / (float)(H - 1);
        float dq[3] = {du * zi, dv * zi, -(du * q[0] * zi + dv * q[1] * zi) * zi};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float acc = 0.f;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                acc += dq[r] * P[r * 4 + j];
                dP[r * 4 + j] += dq[r] * ph[j];
            }
            dpts[((size_t)b * 4 + j) * n + i] = acc;
        }
    }
    if (part) {
#pragma unroll
        for (int k = 0; k < 12; ++k) dP[k] = wave_sum(dP[k]);
        if ((threadIdx.x & 63) == 0)
#pragma unroll
            for (int k = 0; k < 12; ++k) red[threadIdx.x >> 6][k] = dP[k];
        __syncthreads();
        if (threadIdx.x < 12)
            part[((size_t)b * gridDim.x + blockIdx.x) * 12 + threadIdx.x] =
                red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    }
}
__global__ __launch_bounds__(64) void project3d_dT_kernel(const float* part, const float* K, float* dT, int nblk) {
    __shared__ float dPs[12];
    const int b = blockIdx.x, lane = threadIdx.x;
    float acc[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k] = 0.f;
    for (int k = lane; k < nblk; k += 64)
#pragma unroll
        for (int j = 0; j < 12; ++j) acc[j] += part[((size_t)b * nblk + k) * 12 + j];
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k] = wave_sum(acc[k]);
    if (lane == 0)
#pragma unroll
        for (int k = 0; k < 12; ++k) dPs[k] = acc[k];
    __syncthreads();
    if (lane < 16) {
        const int r = lane >> 2, c = lane & 3;
        const float* k = K + b * 16;
        dT[b * 16 + lane] = k[0 * 4 + r] * dPs[0 * 4 + c] + k[1 * 4 + r] * dPs[1 * 4 + c] + k[2 * 4 + r] * dPs[2 * 4 + c];
    }
}

// ------------------------------------------------------------------ a9 grid_sample (bilinear, border)
__global__ void grid_sample_fwd_kernel(const float* img, const float* grid, float* out, int C, int H, int W, int Ho,
                                       int Wo, int ac) {
    const int n = Ho * Wo;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (i >= n) return;
    const float2 g = reinterpret_cast<const float2*>(grid)[(size_t)b * n + i];
    float mx, my;
    const float x = unnormalize_clip(g.x, W, ac, mx), y = unnormalize_clip(g.y, H, ac, my);
    const Bilin bl = bilin_setup(x, y, H, W);
    const float wx0 = 1.f - bl.wx1, wy0 = 1.f - bl.wy1;
    const float wnw = wx0 * wy0, wne = bl.wx1 * wy0, wsw = wx0 * bl.wy1, wse = bl.wx1 * bl.wy1;
    for (int c = 0; c < C; ++c) {
        const float* pl = img + ((size_t)b * C + c) * H * W;
        out[((size_t)b * C + c) * n + i] = pl[bl.o00] * wnw + pl[bl.o01] * wne + pl[bl.o10] * wsw + pl[bl.o11] * wse;
    }
}
__global__ void grid_sample_bwd_kernel(const float* img, const float* grid, const float* gout, float* dgrid, int C,
                                       int H, int W, int Ho, int Wo, int ac) {
    const int n = Ho * Wo;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (i >= n) return;
    const float2 g = reinterpret_cast<const float2*>(grid)[(size_t)b * n + i];
    float mx, my;
    const float x = unnormalize_clip(g.x, W, ac, mx), y = unnormalize_clip(g.y, H, ac, my);
    const Bilin bl = bilin_setup(x, y, H, W);
    const float wx0 = 1.f - bl.wx1, wy0 = 1.f - bl.wy1;
    float gix = 0.f, giy = 0.f;
    for (int c = 0; c < C; ++c) {
        const float* pl = img + ((size_t)b * C + c) * H * W;
        const float nw = pl[bl.o00], ne = pl[bl.o01], sw = pl[bl.o10], se = pl[bl.o11];
        const float go = gout[((size_t)b * C + c) * n + i];
        gix += go * ((ne - nw) * wy0 + (se - sw) * bl.wy1);
        giy += go * ((sw - nw) * wx0 + (se - ne) * bl.wx1);
    }
    reinterpret_cast<float2*>(dgrid)[(size_t)b * n + i] = make_float2(gix * mx, giy * my);
}

// ------------------------------------------------------------------ a10 bilinear upsample
__global__ void upsample_fwd_kernel(const float* x, float* out, int h, int w, int Ho, int Wo, float ry, float rx) {
    const int n = Ho * Wo;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int bc = blockIdx.y;
    if (i >= n) return;
    const float* d = x + (size_t)bc * h * w;
    const int oy = i / Wo, ox = i - oy * Wo;
    float v;
    if (h == Ho && w == Wo) {
        v = d[i];
    } else {
        LinTap ty = lin_tap(oy, ry, h), tx = lin_tap(ox, rx, w);
        const float a = d[ty.i0 * w + tx.i0], b = d[ty.i0 * w + tx.i1];
        const float c = d[ty.i1 * w + tx.i0], e = d[ty.i1 * w + tx.i1];
        const float w0 = 1.f - tx.w1, h0 = 1.f - ty.w1;
        v = h0 * (w0 * a + tx.w1 * b) + ty.w1 * (w0 * c + tx.w1 * e);
    }
    out[(size_t)bc * n + i] = v;
}
__global__ void upsample_bwd_kernel(const float* gout, float* dx, int h, int w, int Ho, int Wo, float ry, float rx) {
    const int n = h * w;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int bc = blockIdx.y;
    if (i >= n) return;
    const float* gu = gout + (size_t)bc * Ho * Wo;
    const int y = i / w, x = i - y * w;
    float acc = 0.f;
    if (h == Ho && w == Wo) {
        acc = gu[i];
    } else {
        int ylo = max((int)floorf(((float)y - 0.5f) / ry - 0.5f) - 1, 0);
        int yhi = min((int)ceilf(((float)y + 1.5f) / ry - 0.5f) + 1, Ho - 1);
        int xlo = max((int)floorf(((float)x - 0.5f) / rx - 0.5f) - 1, 0);
        int xhi = min((int)ceilf(((float)x + 1.5f) / rx - 0.5f) + 1, Wo - 1);
        for (int dy = ylo; dy <= yhi; ++dy) {
            LinTap ty = lin_tap(dy, ry, h);
            float wy = (ty.i0 == y ? 1.f - ty.w1 : 0.f) + (ty.i1 == y ? ty.w1 : 0.f);
            if (wy == 0.f) continue;
            float row = 0.f;
            for (int dxx = xlo; dxx <= xhi; ++dxx) {
                LinTap tx = lin_tap(dxx, rx, w);
                float wx = (tx.i0 == x ? 1.f - tx.w1 : 0.f) + (tx.i1 == x ? tx.w1 : 0.f);
                row += wx * gu[(size_t)dy * Wo + dxx];
            }
            acc += wy * row;
        }
    }
    dx[(size_t)bc * n + i] = acc;
}

// ------------------------------------------------------------------ a11 SSIM (LDS-staged tiles)
constexpr int ST_W = 32, ST_H = 8;   // output tile per 256-thread block
constexpr float sC1 = 0.01f * 0.01f, sC2 = 0.03f * 0.03f, s9 = 1.f / 9.f;

struct SStat {
    float mu_x, mu_y, n1, n2, d1, d2;
};
template <int LW>
__device__ __forceinline__ SStat ssim_stat(const float* sx, const float* sy, int cy, int cx) {
    // Unfused throughout, as torch evaluates layers.py:219-231.  Contraction is the compiler's choice per operation: it
    // packed sum(x^2), sum(y^2) into v_pk_fma_f32 but paired sum(xy) with sum(x) in a v_pk_add_f32 of a separately rounded
    // product, so E[x^2] and E[xy] differed by an ulp (6e-8) for x == y -- amplified by 1 / C2 to 6e-5 in SSIM(x, x).
    // With every product rounded, numerator and denominator are bitwise equal for x == y (the reference's KAT: == 0).
#pragma clang fp contract(off)
    float ax = 0.f, ay = 0.f, axx = 0.f, ayy = 0.f, axy = 0.f;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const float xv = sx[(cy + dy) * LW + cx + dx], yv = sy[(cy + dy) * LW + cx + dx];
            ax += xv; ay += yv; axx += xv * xv; ayy += yv * yv; axy += xv * yv;
        }
    SStat s;
    s.mu_x = ax * s9; s.mu_y = ay * s9;
    const float mxx = s.mu_x * s.mu_x, myy = s.mu_y * s.mu_y, mxy = s.mu_x * s.mu_y;
    const float sig_x = axx * s9 - mxx, sig_y = ayy * s9 - myy;
    const float sig_xy = axy * s9 - mxy;
    s.n1 = 2.f * mxy + sC1; s.n2 = 2.f * sig_xy + sC2;
    s.d1 = mxx + myy + sC1; s.d2 = sig_x + sig_y + sC2;
    return s;
}

__global__ __launch_bounds__(256) void ssim_fwd_kernel(const float* x, const float* y, float* out, int H, int W) {
    constexpr int LW = ST_W + 2, LH = ST_H + 2;
    __shared__ float sx[LW * LH], sy[LW * LH];
    const int bc = blockIdx.z;
    const int x0 = blockIdx.x * ST_W, y0 = blockIdx.y * ST_H;
    const float* px = x + (size_t)bc * H * W;
    const float* py = y + (size_t)bc * H * W;
    for (int k = threadIdx.x; k < LW * LH; k += 256) {
        const int ly = k / LW, lx = k - ly * LW;
        const int gy = reflect_clamp(y0 + ly - 1, H), gx = reflect_clamp(x0 + lx - 1, W);
        sx[k] = px[(size_t)gy * W + gx];
        sy[k] = py[(size_t)gy * W + gx];
    }
    __syncthreads();
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int gx = x0 + tx, gy = y0 + ty;
    if (gx < W && gy < H) {
        SStat s = ssim_stat<LW>(sx, sy, ty + 1, tx + 1);
        float v = (1.f - (s.n1 * s.n2) / (s.d1 * s.d2)) * 0.5f;
        out[(size_t)bc * H * W + (size_t)gy * W + gx] = fminf(fmaxf(v, 0.f), 1.f);
    }
}

// d_x of sum(g * SSIM(x,y)); 5x5 footprint: x,y tile with halo 2, coefficient tile with halo 1.
__global__ __launch_bounds__(256) void ssim_bwd_kernel(const float* x, const float* y, const float* g, float* dx,
                                                       int H, int W) {
    constexpr int LW = ST_W + 4, LH = ST_H + 4;   // halo 2
    constexpr int CW = ST_W + 2, CH = ST_H + 2;   // halo 1
    __shared__ float sx[LW * LH], sy[LW * LH];
    __shared__ float ca[CW * CH], cb[CW * CH], cc[CW * CH];
    const int bc = blockIdx.z;
    const int x0 = blockIdx.x * ST_W, y0 = blockIdx.y * ST_H;
    const float* px = x + (size_t)bc * H * W;
    const float* py = y + (size_t)bc * H * W;
    const float* pg = g + (size_t)bc * H * W;
    for (int k = threadIdx.x; k < LW * LH; k += 256) {
        const int ly = k / LW, lx = k - ly * LW;
        const int gy = reflect_clamp(y0 + ly - 2, H), gx = reflect_clamp(x0 + lx - 2, W);
        sx[k] = px[(size_t)gy * W + gx];
        sy[k] = py[(size_t)gy * W + gx];
    }
    __syncthreads();
    for (int k = threadIdx.x; k < CW * CH; k += 256) {
        const int ly = k / CW, lx = k - ly * CW;
        const int gy = y0 + ly - 1, gx = x0 + lx - 1;   // output pixel p owning this coefficient
        float a = 0.f, b = 0.f, c = 0.f;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
            SStat s = ssim_stat<LW>(sx, sy, ly + 1, lx + 1);
            const float n = s.n1 * s.n2, d = s.d1 * s.d2, id = 1.f / d;
            const float v = (1.f - n * id) * 0.5f;
            const float G = (v >= 0.f && v <= 1.f) ? pg[(size_t)gy * W + gx] : 0.f;
            const float nid2 = n * id * id;
            a = G * (-s.mu_y * (s.n2 - s.n1) * id + nid2 * s.mu_x * (s.d2 - s.d1));
            b = G * 0.5f * nid2 * s.d1;
            c = -G * s.n1 * id;
        }
        ca[k] = a; cb[k] = b; cc[k] = c;
    }
    __syncthreads();
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int gx = x0 + tx, gy = y0 + ty;
    if (gx < W && gy < H) {
        float SA = 0.f, SB = 0.f, SC = 0.f;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
            const float fy = ((dy < 0 && gy == 1) || (dy > 0 && gy == H - 2)) ? 2.f : 1.f;
#pragma unroll
            for (int dxx = -1; dxx <= 1; ++dxx) {
                const float f = fy * (((dxx < 0 && gx == 1) || (dxx > 0 && gx == W - 2)) ? 2.f : 1.f);
                const int k = (ty + 1 + dy) * CW + tx + 1 + dxx;
                SA += f * ca[k]; SB += f * cb[k]; SC += f * cc[k];
            }
        }
        const float xv = sx[(ty + 2) * LW + tx + 2], yv = sy[(ty + 2) * LW + tx + 2];
        dx[(size_t)bc * H * W + (size_t)gy * W + gx] = (SA + 2.f * xv * SB + yv * SC) * s9;
    }
}

// ------------------------------------------------------------------ a13 get_smooth_loss
constexpr int SMO_CHUNK = 2048;
__global__ __launch_bounds__(256) void smooth_part_kernel(const float* disp, const float* img, float* part, int C,
                                                          int h, int w) {
    __shared__ float sm[2][4];
    const int b = blockIdx.y, n = h * w;
    const float* d = disp + (size_t)b * n;
    const float* im = img + (size_t)b * C * n;
    float sx = 0.f, sy = 0.f;
    const float ic = 1.f / (float)C;
    for (int i = blockIdx.x * SMO_CHUNK + threadIdx.x; i < min((blockIdx.x + 1) * SMO_CHUNK, n); i += 256) {
        const int y = i / w, x = i - y * w;
        const float dv = d[i];
        if (x < w - 1) {
            float gi = 0.f;
            for (int c = 0; c < C; ++c) gi += fabsf(im[(size_t)c * n + i] - im[(size_t)c * n + i + 1]);
            sx += fabsf(dv - d[i + 1]) * __expf(-gi * ic);
        }
        if (y < h - 1) {
            float gi = 0.f;
            for (int c = 0; c < C; ++c) gi += fabsf(im[(size_t)c * n + i] - im[(size_t)c * n + i + w]);
            sy += fabsf(dv - d[i + w]) * __expf(-gi * ic);
        }
    }
    sx = wave_sum(sx); sy = wave_sum(sy);
    if ((threadIdx.x & 63) == 0) { sm[0][threadIdx.x >> 6] = sx; sm[1][threadIdx.x >> 6] = sy; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float* o = part + ((size_t)b * gridDim.x + blockIdx.x) * 2;
        o[0] = sm[0][0] + sm[0][1] + sm[0][2] + sm[0][3];
        o[1] = sm[1][0] + sm[1][1] + sm[1][2] + sm[1][3];
    }
}
__global__ __launch_bounds__(256) void smooth_final_kernel(const float* part, float* out, int nparts, int B, int h,
                                                           int w) {
    __shared__ float sm[2][4];
    float sx = 0.f, sy = 0.f;
    for (int k = threadIdx.x; k < nparts; k += 256) { sx += part[k * 2]; sy += part[k * 2 + 1]; }
    sx = wave_sum(sx); sy = wave_sum(sy);
    if ((threadIdx.x & 63) == 0) { sm[0][threadIdx.x >> 6] = sx; sm[1][threadIdx.x >> 6] = sy; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float tx = sm[0][0] + sm[0][1] + sm[0][2] + sm[0][3], ty = sm[1][0] + sm[1][1] + sm[1][2] + sm[1][3];
        out[0] = tx / ((float)B * h * (w - 1)) + ty / ((float)B * (h - 1) * w);
    }
}
__global__ void smooth_bwd_kernel(const float* disp, const float* img, const float* g, float* dd, int B, int C, int h,
                                  int w) {
    const int n = h * w;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (i >= n) return;
    const float* d = disp + (size_t)b * n;
    const float* im = img + (size_t)b * C * n;
    const float ic = 1.f / (float)C;
    const int y = i / w, x = i - y * w;
    auto sgn = [](float v) { return (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f); };
    auto ex = [&](int i0, int i1) {
        float gi = 0.f;
        for (int c = 0; c < C; ++c) gi += fabsf(im[(size_t)c * n + i0] - im[(size_t)c * n + i1]);
        return __expf(-gi * ic);
    };
    const float dv = d[i];
    float gx = 0.f, gy = 0.f;
    if (x < w - 1) gx += sgn(dv - d[i + 1]) * ex(i, i + 1);
    if (x > 0) gx -= sgn(d[i - 1] - dv) * ex(i - 1, i);
    if (y < h - 1) gy += sgn(dv - d[i + w]) * ex(i, i + w);
    if (y > 0) gy -= sgn(d[i - w] - dv) * ex(i - w, i);
    dd[(size_t)b * n + i] = g[0] * (gx / ((float)B * h * (w - 1)) + gy / ((float)B * (h - 1) * w));
}

}  // namespace dc

using namespace dc;
#define ST ((hipStream_t)stream)

extern "C" int dc_disp_to_depth_fwd(const float* disp, float* scaled, float* depth, size_t n, float min_depth,
                                    float max_depth, void* stream) {
    if (!disp || n == 0 || !(min_depth > 0.f) || !(max_depth > min_depth)) return DC_EINVAL;
    const float lo = 1.f / max_depth, rng = 1.f / min_depth - 1.f / max_depth;
    int blocks = (int)std::min<size_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(d2d_fwd_kernel, dim3(blocks), dim3(256), 0, ST, disp, scaled, depth, n, lo, rng);
    DC_CHECK_LAUNCH();
    return DC_OK;
}
extern "C" int dc_disp_to_depth_bwd(const float* disp, const float* g_scaled, const float* g_depth, float* d_disp,
                                    size_t n, float min_depth, float max_depth, void* stream) {
    if (!disp || !d_disp || n == 0 || !(min_depth > 0.f) || !(max_depth > min_depth)) return DC_EINVAL;
    const float lo = 1.f / max_depth, rng = 1.f / min_depth - 1.f / max_depth;
    int blocks = (int)std::min<size_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(d2d_bwd_kernel, dim3(blocks), dim3(256), 0, ST, disp, g_scaled, g_depth, d_disp, n, lo, rng);
    DC_CHECK_LAUNCH();
    return DC_OK;
}
extern "C" int dc_pix_coords(float* pc, int B, int H, int W, void* stream) {
    if (!pc || B <= 0 || H <= 0 || W <= 0) return DC_EINVAL;
    hipLaunchKernelGGL(pix_coords_kernel, dim3(ceil_div(H * W, 256), B), dim3(256), 0, ST, pc, B, H, W);
    DC_CHECK_LAUNCH();
    return DC_OK;
}
extern "C" int dc_backproject_fwd(const float* depth, const float* inv_K, float* cam, int B, int H, int W,
                                  void* stream) {
    if (!depth || !inv_K || !cam || B <= 0 || H <= 0 || W <= 0) return DC_EINVAL;
    hipLaunchKernelGGL(backproject_fwd_kernel, dim3(ceil_div(H * W, 256), B), dim3(256), 0, ST, depth, inv_K, cam, H, W);
    DC_CHECK_LAUNCH();
    return DC_OK;
}
extern "C" int dc_backproject_bwd(const float* g_cam, const float* inv_K, float* d_depth, int B, int H, int W,
                                  void* stream) {
    if (!g_cam || !inv_K || !d_depth || B <= 0 || H <= 0 || W <= 0) return DC_EINVAL;
    hipLaunchKernelGGL(backproject_bwd_kernel, dim3(ceil_div(H * W, 256), B), dim3(256), 0, ST, g_cam, inv_K, d_depth, H, W);
    DC_CHECK_LAUNCH();
    return DC_OK;
}
extern "C" int dc_project3d_fwd(const float* points, const float* K, const float* T, float* grid, int B, int H, int W,
                                float eps, void* stream) {
    if (!points || !K || !T || !grid || B <= 0 || H < 2 || W < 2) return DC_EINVAL;
    hipLaunchKernelGGL(project3d_fwd_kernel, dim3(ceil_div(H * W, 256), B), dim3(256), 0, ST, points, K, T, grid, H, W, eps);
    DC_CHECK_LAUNCH();
    return DC_OK;
}
extern "C" size_t dc_project3d_bwd_workspace(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return (size_t)B * ceil_div(H * W, PJ_PIX) * 12 * sizeof(float);
}
extern "C" int dc_project3d_bwd(const float* points, const float* K, const float* T, const float* g_grid,
                                float* d_points, float* d_T, void* ws, int B, int H, int W, float eps, void* stream) {
    if (!points || !K || !T || !g_grid || !d_points || B <= 0 || H < 2 || W < 2) return DC_EINVAL;
    if (d_T && !ws) return DC_EWORKSPACE;
    const int nblk = ceil_div(H * W, PJ_PIX);
    hipLaunchKernelGGL(project3d_bwd_kernel, dim3(nblk, B), dim3(256), 0, ST, points, K, T, g_grid, d_points,
                       d_T ? (float*)ws : nullptr, H, W, eps);
    DC_CHECK_LAUNCH();
    if (d_T) {
        hipLaunchKernelGGL(project3d_dT_kernel, dim3(B), dim3(64), 0, ST, (const float*)ws, K, d_T, nblk);
        DC_CHECK_LAUNCH();
    }
    return DC_OK;
}
extern "C" int dc_grid_sample_fwd(const float* img, const float* grid, float* out, int B, int C, int H, int W, int Ho,
                                  int Wo, int align_corners, void* stream) {
    if (!img || !grid || !out || B <= 0 || C <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0) return DC_EINVAL;
    hipLaunchKernelGGL(grid_sample_fwd_kernel, dim3(ceil_div(Ho * Wo, 256), B), dim3(256), 0, ST, img, grid, out, C, H, W,
                       Ho, Wo, align_corners);
    DC_CHECK_LAUNCH();
    return DC_OK;
}
extern "C" int dc_grid_sample_bwd(const float* img, const float* grid, const float* g_out, float* d_grid, int B, int C,
                                  int H, int W, int Ho, int Wo, int align_corners, void* stream) {
    if (!img || !grid || !g_out || !d_grid || B <= 0 || C <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0) return DC_EINVAL;
    hipLaunchKernelGGL(grid_sample_bwd_kernel, dim3(ceil_div(Ho * Wo, 256), B), dim3(256), 0, ST, img, grid, g_out, d_grid,
                       C, H, W, Ho, Wo, align_corners);
    DC_CHECK_LAUNCH();
    return DC_OK;
}
extern "C" int dc_upsample_bilinear_fwd(const float* x, float* out, int BC, int h, int w, int Ho, int Wo, void* stream) {
    if (!x || !out || BC <= 0 || h <= 0 || w <= 0 || Ho <= 0 || Wo <= 0) return DC_EINVAL;
    hipLaunchKernelGGL(upsample_fwd_kernel, dim3(ceil_div(Ho * Wo, 256), BC), dim3(256), 0, ST, x, out, h, w, Ho, Wo,
                       (float)h / (float)Ho, (float)w / (float)Wo);
    DC_CHECK_LAUNCH();
    return DC_OK;
}
extern "C" int dc_upsample_bilinear_bwd(const float* g_out, float* d_x, int BC, int h, int w, int Ho, int Wo,
                                        void* stream) {
    if (!g_out || !d_x || BC <= 0 || h <= 0 || w <= 0 || Ho <= 0 || Wo <= 0) return DC_EINVAL;
    hipLaunchKernelGGL(upsample_bwd_kernel, dim3(ceil_div(h * w, 256), BC), dim3(256), 0, ST, g_out, d_x, h, w, Ho, Wo,
                       (float)h / (float)Ho, (float)w / (float)Wo);
    DC_CHECK_LAUNCH();
    return DC_OK;
}
// ---- upsample(x) = F.interpolate(x, scale_factor=2, mode="nearest")  (layers.py:196-199): out[y][x] = in[y/2][x/2];
// backward = the 2x2 block sum.  One thread per INPUT pixel: a float2 pair of each of the two output rows it owns.
__global__ __launch_bounds__(256) void nearest2x_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, int BC, int h, int w) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= h * w) return;
    const int y = i / w, xx = i - y * w;
    for (int bc = blockIdx.y; bc < BC; bc += gridDim.y) {
        const float v = x[(size_t)bc * h * w + i];
        float2* o = reinterpret_cast<float2*>(out + ((size_t)bc * 2 * h + 2 * y) * 2 * w) + xx;
        o[0] = make_float2(v, v);
        o[w] = make_float2(v, v);
    }
}
__global__ __launch_bounds__(256) void nearest2x_bwd_kernel(const float* __restrict__ g, float* __restrict__ dx, int BC, int h, int w) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= h * w) return;
    const int y = i / w, xx = i - y * w;
    for (int bc = blockIdx.y; bc < BC; bc += gridDim.y) {
        const float2* q = reinterpret_cast<const float2*>(g + ((size_t)bc * 2 * h + 2 * y) * 2 * w) + xx;
        const float2 a = q[0], b = q[w];
        dx[(size_t)bc * h * w + i] = (a.x + a.y) + (b.x + b.y);
    }
}
extern "C" int dc_upsample_nearest2x_fwd(const float* x, float* out, int BC, int h, int w, void* stream) {
    if (!x || !out || BC <= 0 || h <= 0 || w <= 0 || ((size_t)out & 7)) return DC_EINVAL;
    hipLaunchKernelGGL(nearest2x_fwd_kernel, dim3(ceil_div(h * w, 256), BC < 65535 ? BC : 65535), dim3(256), 0, ST, x, out, BC, h, w);
    DC_CHECK_LAUNCH();
    return DC_OK;
}
extern "C" int dc_upsample_nearest2x_bwd(const float* g_out, float* d_x, int BC, int h, int w, void* stream) {
    if (!g_out || !d_x || BC <= 0 || h <= 0 || w <= 0 || ((size_t)g_out & 7)) return DC_EINVAL;
    hipLaunchKernelGGL(nearest2x_bwd_kernel, dim3(ceil_div(h * w, 256), BC < 65535 ? BC : 65535), dim3(256), 0, ST, g_out, d_x, BC, h, w);
    DC_CHECK_LAUNCH();
    return DC_OK;
}
extern "C" int dc_ssim_fwd(const float* x, const float* y, float* out, int BC, int H, int W, void* stream) {
    if (!x || !y || !out || BC <= 0 || H < 2 || W < 2) return DC_EINVAL;
    hipLaunchKernelGGL(ssim_fwd_kernel, dim3(ceil_div(W, ST_W), ceil_div(H, ST_H), BC), dim3(256), 0, ST, x, y, out, H, W);
    DC_CHECK_LAUNCH();
    return DC_OK;
}
extern "C" int dc_ssim_bwd(const float* x, const float* y, const float* g_out, float* d_x, float* d_y, int BC, int H,
                           int W, void* stream) {
    if (!x || !y || !g_out || BC <= 0 || H < 2 || W < 2) return DC_EINVAL;
    const dim3 grid(ceil_div(W, ST_W), ceil_div(H, ST_H), BC);
    if (d_x) { hipLaunchKernelGGL(ssim_bwd_kernel, grid, dim3(256), 0, ST, x, y, g_out, d_x, H, W); DC_CHECK_LAUNCH(); }
    // SSIM is symmetric in its arguments: d/dy is the same kernel with the roles swapped
    if (d_y) { hipLaunchKernelGGL(ssim_bwd_kernel, grid, dim3(256), 0, ST, y, x, g_out, d_y, H, W); DC_CHECK_LAUNCH(); }
    return DC_OK;
}
extern "C" size_t dc_smooth_workspace(int B, int h, int w) {
    if (B <= 0 || h <= 0 || w <= 0) return 0;
    return (size_t)B * ceil_div(h * w, SMO_CHUNK) * 2 * sizeof(float);
}
extern "C" int dc_smooth_fwd(const float* disp, const float* img, float* out, void* ws, int B, int C, int h, int w,
                             void* stream) {
    if (!disp || !img || !out || !ws || B <= 0 || C <= 0 || h < 2 || w < 2) return DC_EINVAL;
    const int nch = ceil_div(h * w, SMO_CHUNK);
    hipLaunchKernelGGL(smooth_part_kernel, dim3(nch, B), dim3(256), 0, ST, disp, img, (float*)ws, C, h, w);
    DC_CHECK_LAUNCH();
    hipLaunchKernelGGL(smooth_final_kernel, dim3(1), dim3(256), 0, ST, (const float*)ws, out, nch * B, B, h, w);
    DC_CHECK_LAUNCH();
    return DC_OK;
}
extern "C" int dc_smooth_bwd(const float* disp, const float* img, const float* g, float* d_disp, int B, int C, int h,
                             int w, void* stream) {
    if (!disp || !img || !g || !d_disp || B <= 0 || C <= 0 || h < 2 || w < 2) return DC_EINVAL;
    hipLaunchKernelGGL(smooth_bwd_kernel, dim3(ceil_div(h * w, 256), B), dim3(256), 0, ST, disp, img, g, d_disp, B, C, h, w);
    DC_CHECK_LAUNCH();
    return DC_OK;
}
