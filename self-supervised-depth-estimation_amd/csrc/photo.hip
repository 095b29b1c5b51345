// Fused photometric loss for gfx950: generate_images_pred + compute_reprojection_loss +
// compute_losses of the reference trainer (trainer.py:465-622) as row-marching wavefront kernels.
//
// Design:
//   * one WAVE owns a 64-lane-wide column strip of one image and marches down R rows; the 3x3 SSIM
//     windows are horizontal DPP wave-shift sums + a 2-row register ring, so there is no LDS
//     staging and no barrier inside the march;
//   * wave w of a block handles (scale = w>>1, source frame = w&1) of the same strip, so the target
//     rows are fetched from HBM once and re-served by L1/L2 to the other waves;
//   * nothing of the reference's ~25 intermediate (B,3,H,W) tensors is materialised: the warped
//     image exists only in registers (it is written out only when the log tensors are requested);
//   * backward recomputes the warp (halo 2) instead of saving it, routes the min() gradient by the
//     1-byte argmin map the forward wrote, and reduces the pose gradient per wave -> per block ->
//     fixed-order final sum (no float atomics, bitwise reproducible).
#include "dc_common.h"

namespace dc {

constexpr int R_ROWS = 16;            // image rows produced per wave
constexpr float kC1 = 0.01f * 0.01f;  // layers.py:231-232
constexpr float kC2 = 0.03f * 0.03f;
constexpr float k9 = 1.f / 9.f;

struct PhotoArgs {
    int B, H, W, ns;
    unsigned flags;
    float min_disp, disp_range;   // 1/max_depth, 1/min_depth - 1/max_depth
    float inv_Wm1, inv_Hm1;
    float smoothness;
    const float* target;
    const float* src[2];
    const float* K;
    const float* invK;
    const float* T[2];
    const float* disp[DC_MAX_SCALES];
    const float* color_s[DC_MAX_SCALES];
    int hs[DC_MAX_SCALES], ws[DC_MAX_SCALES];
    float ry[DC_MAX_SCALES], rx[DC_MAX_SCALES];
    const float* noise[DC_MAX_SCALES];
    unsigned long long seed;
    float* idl;                        // identity losses (B, 2|1, H, W)
    uint8_t* argmin[DC_MAX_SCALES];
    float* depth[DC_MAX_SCALES];
    float* sample[DC_MAX_SCALES][2];
    float* color[DC_MAX_SCALES][2];
    float* idsel[DC_MAX_SCALES];
    float* losses;
    const float* g_losses;
    float* d_disp[DC_MAX_SCALES];
    float* d_T[2];
    // workspace carve
    float* part_photo;   // [ns][nblk_f][2]
    float* part_smooth;  // [ns][B][nchunk][3]
    float* stats;        // [ns][B][3]  (mean disp, Sx, Sy)
    float* gdup[DC_MAX_SCALES];   // full-res d(upsampled disp)
    float* part_dP;      // [ns][2][B][nblk_b_per_image][12]
    int nblk_f, nchunk, nblk_b_img;
};

__device__ __forceinline__ float uni(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}

// per-(batch, frame) geometry held in SGPRs
struct Geo {
    float iK[9];
    float P[12];
};

__device__ __forceinline__ void load_geo(Geo& g, const float* K, const float* invK, const float* T, int b) {
    const float* k = K + b * 16;
    const float* t = T + b * 16;
    const float* ik = invK + b * 16;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) g.iK[i * 3 + j] = uni(ik[i * 4 + j]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // torch.matmul(K, T)[:, :3, :]   (layers.py:183)
            float acc = k[i * 4 + 0] * t[0 * 4 + j];
            acc = fmaf(k[i * 4 + 1], t[1 * 4 + j], acc);
            acc = fmaf(k[i * 4 + 2], t[2 * 4 + j], acc);
            acc = fmaf(k[i * 4 + 3], t[3 * 4 + j], acc);
            g.P[i * 4 + j] = uni(acc);
        }
    }
}

// F.interpolate(disp_s, [H,W], bilinear, align_corners=False) at one pixel (trainer.py:474-475)
__device__ __forceinline__ float disp_at(const float* d, int hs, int ws, float ry, float rx, int H, int W,
                                         int x, int y) {
    if (hs == H && ws == W) return d[y * W + x];
    LinTap ty = lin_tap(y, ry, hs), tx = lin_tap(x, rx, ws);
    float a = d[ty.i0 * ws + tx.i0], b = d[ty.i0 * ws + tx.i1];
    float c = d[ty.i1 * ws + tx.i0], e = d[ty.i1 * ws + tx.i1];
    float w0 = 1.f - tx.w1, h0 = 1.f - ty.w1;
    return h0 * (w0 * a + tx.w1 * b) + ty.w1 * (w0 * c + tx.w1 * e);
}

struct Proj {
    float depth, ray[3], cam[3], u, v, zi;   // zi = 1/(z+eps)
    float ix, iy, mx, my;                    // source coords + d(ix)/d(gx) incl. clamp mask
    float gx, gy;
};

// disp -> depth -> BackprojectDepth -> Project3D -> grid_sample coordinate (layers.py:21-24,163-192)
__device__ __forceinline__ void project_pixel(Proj& p, const Geo& g, float disp, float min_disp,
                                              float disp_range, int x, int y, int H, int W, float inv_Wm1,
                                              float inv_Hm1, bool ac) {
    float scaled = min_disp + disp_range * disp;
    p.depth = 1.0f / scaled;
    float xf = (float)x, yf = (float)y;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float r = g.iK[i * 3 + 0] * xf;
        r = fmaf(g.iK[i * 3 + 1], yf, r);
        r = r + g.iK[i * 3 + 2];
        p.ray[i] = r;
        p.cam[i] = p.depth * r;
    }
    float q[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float a = g.P[i * 4 + 0] * p.cam[0];
        a = fmaf(g.P[i * 4 + 1], p.cam[1], a);
        a = fmaf(g.P[i * 4 + 2], p.cam[2], a);
        q[i] = a + g.P[i * 4 + 3];
    }
    p.zi = 1.0f / (q[2] + 1e-7f);
    p.u = q[0] * p.zi;
    p.v = q[1] * p.zi;
    p.gx = (p.u * inv_Wm1 - 0.5f) * 2.f;
    p.gy = (p.v * inv_Hm1 - 0.5f) * 2.f;
    p.ix = unnormalize_clip(p.gx, W, ac, p.mx);
    p.iy = unnormalize_clip(p.gy, H, ac, p.my);
}

struct HSum {   // horizontal 3-sums of one row, one source frame, three channels
    float hy[3], hyy[3], hx[3], hxx[3], hxy[3];
};

__device__ __forceinline__ void hsum_row(HSum& h, const float t[3], const float w[3]) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float tl = from_left(t[c]), tr = from_right(t[c]);
        float wl = from_left(w[c]), wr = from_right(w[c]);
        h.hy[c] = tl + t[c] + tr;
        h.hyy[c] = tl * tl + t[c] * t[c] + tr * tr;
        h.hx[c] = wl + w[c] + wr;
        h.hxx[c] = wl * wl + w[c] * w[c] + wr * wr;
        h.hxy[c] = wl * tl + w[c] * t[c] + wr * tr;
    }
}

struct SsimTerms {
    float mu_x, mu_y, n1, n2, d1, d2;
};
__device__ __forceinline__ SsimTerms ssim_terms(const HSum& a, const HSum& b, const HSum& c, int ch) {
    SsimTerms s;
    s.mu_x = (a.hx[ch] + b.hx[ch] + c.hx[ch]) * k9;
    s.mu_y = (a.hy[ch] + b.hy[ch] + c.hy[ch]) * k9;
    float sig_x = (a.hxx[ch] + b.hxx[ch] + c.hxx[ch]) * k9 - s.mu_x * s.mu_x;
    float sig_y = (a.hyy[ch] + b.hyy[ch] + c.hyy[ch]) * k9 - s.mu_y * s.mu_y;
    float sig_xy = (a.hxy[ch] + b.hxy[ch] + c.hxy[ch]) * k9 - s.mu_x * s.mu_y;
    s.n1 = 2.f * s.mu_x * s.mu_y + kC1;
    s.n2 = 2.f * sig_xy + kC2;
    s.d1 = s.mu_x * s.mu_x + s.mu_y * s.mu_y + kC1;
    s.d2 = sig_x + sig_y + kC2;
    return s;
}

// 0.85 * mean_c SSIM + 0.15 * mean_c |t - w|      (trainer.py:517-529)
__device__ __forceinline__ float reproj_value(const HSum& a, const HSum& b, const HSum& c, const float tc[3],
                                              const float wc[3], bool no_ssim) {
    float l1 = (fabsf(tc[0] - wc[0]) + fabsf(tc[1] - wc[1]) + fabsf(tc[2] - wc[2])) * (1.f / 3.f);
    if (no_ssim) return l1;
    float ss = 0.f;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        SsimTerms s = ssim_terms(a, b, c, ch);
        float v = (1.f - (s.n1 * s.n2) / (s.d1 * s.d2)) * 0.5f;
        ss += fminf(fmaxf(v, 0.f), 1.f);
    }
    return 0.85f * (ss * (1.f / 3.f)) + 0.15f * l1;
}

// counter-based N(0,1) for the on-device tie-break noise (used only when no noise tensor is given)
__device__ __forceinline__ float rng_normal(unsigned long long seed, unsigned idx, unsigned stream) {
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * ((unsigned long long)idx * 8ull + stream + 1ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    float u1 = ((unsigned)(z >> 40) + 1u) * (1.f / 16777217.f);
    float u2 = (unsigned)(z & 0xFFFFFFu) * (1.f / 16777216.f);
    return sqrtf(-2.f * __logf(u1)) * __cosf(6.2831853f * u2);
}

// ------------------------------------------------------------------------------------------------
// identity reprojection losses: reprojection_loss(color(f,0), color(0,0)), f = -1,+1  (trainer.py:562-568)
// one wave = one strip x R rows, both frames.   grid (strips, rowblocks, B), block 64.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void identity_kernel(PhotoArgs p) {
    const int lane = threadIdx.x;
    const int b = blockIdx.z;
    const int x = blockIdx.x * 62 - 1 + lane;
    const int y0 = blockIdx.y * R_ROWS;
    const int H = p.H, W = p.W;
    const int xr = reflect_clamp(x, W);
    const size_t plane = (size_t)H * W;
    const float* tg = p.target + (size_t)b * 3 * plane;
    const float* s0 = p.src[0] + (size_t)b * 3 * plane;
    const float* s1 = p.src[1] + (size_t)b * 3 * plane;
    const bool no_ssim = p.flags & DC_OPT_NO_SSIM;
    const bool avg = p.flags & DC_OPT_AVG_REPROJ;
    const bool lane_ok = lane >= 1 && lane <= 62 && x < W;

    HSum a0, a1, b0, b1;   // rows yy-2, yy-1 for frame 0 / 1
    float ct[3], c0[3], c1[3];
    a0 = a1 = b0 = b1 = HSum{};
    ct[0] = ct[1] = ct[2] = c0[0] = c0[1] = c0[2] = c1[0] = c1[1] = c1[2] = 0.f;
#pragma unroll 1
    for (int i = 0; i < R_ROWS + 2; ++i) {
        const int yy = y0 - 1 + i;
        const int yr = reflect_clamp(yy, H);
        float t[3], w0[3], w1[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            size_t o = c * plane + (size_t)yr * W + xr;
            t[c] = tg[o];
            w0[c] = s0[o];
            w1[c] = s1[o];
        }
        HSum h0, h1;
        hsum_row(h0, t, w0);
        hsum_row(h1, t, w1);
        const int py = yy - 1;
        if (i >= 2 && py < H && lane_ok) {
            float r0 = reproj_value(a0, b0, h0, ct, c0, no_ssim);
            float r1 = reproj_value(a1, b1, h1, ct, c1, no_ssim);
            if (avg) {
                p.idl[((size_t)b * H + py) * W + x] = (r0 + r1) * 0.5f;
            } else {
                p.idl[((size_t)(b * 2 + 0) * H + py) * W + x] = r0;
                p.idl[((size_t)(b * 2 + 1) * H + py) * W + x] = r1;
            }
        }
        a0 = b0; b0 = h0; a1 = b1; b1 = h1;
#pragma unroll
        for (int c = 0; c < 3; ++c) { ct[c] = t[c]; c0[c] = w0[c]; c1[c] = w1[c]; }
    }
}

// ------------------------------------------------------------------------------------------------
// forward: wave (scale s, frame f).  grid (strips62, rowblocks, B), block 128*ns.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void photo_fwd_kernel(PhotoArgs p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [2*ns waves][R_ROWS][64]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int s = wave >> 1, f = wave & 1;
    const int b = blockIdx.z;
    const int x = blockIdx.x * 62 - 1 + lane;
    const int y0 = blockIdx.y * R_ROWS;
    const int H = p.H, W = p.W;
    const int xr = reflect_clamp(x, W);
    const size_t plane = (size_t)H * W;
    const float* tg = p.target + (size_t)b * 3 * plane;
    const float* sp = p.src[f] + (size_t)b * 3 * plane;
    const int hs = p.hs[s], ws = p.ws[s];
    const float* dp = p.disp[s] + (size_t)b * hs * ws;
    const float ry = p.ry[s], rx = p.rx[s];
    const bool no_ssim = p.flags & DC_OPT_NO_SSIM;
    const bool ac = p.flags & DC_OPT_ALIGN_CORNERS;
    const bool lane_ok = lane >= 1 && lane <= 62 && x < W;
    Geo g;
    load_geo(g, p.K, p.invK, p.T[f], b);
    float* col = p.color[s][f];
    float* smp = p.sample[s][f];
    float* dep = (f == 0) ? p.depth[s] : nullptr;
    float* my_lds = lds + (size_t)wave * R_ROWS * 64;

    HSum ha, hb;
    ha = hb = HSum{};
    float ct[3] = {0.f, 0.f, 0.f}, cw[3] = {0.f, 0.f, 0.f};
#pragma unroll 1
    for (int i = 0; i < R_ROWS + 2; ++i) {
        const int yy = y0 - 1 + i;
        const int yr = reflect_clamp(yy, H);
        float t[3], w[3];
        float d = disp_at(dp, hs, ws, ry, rx, H, W, xr, yr);
        Proj pr;
        project_pixel(pr, g, d, p.min_disp, p.disp_range, xr, yr, H, W, p.inv_Wm1, p.inv_Hm1, ac);
        Bilin bl = bilin_setup(pr.ix, pr.iy, H, W);
        const float wx0 = 1.f - bl.wx1, wy0 = 1.f - bl.wy1;
        const float wnw = wx0 * wy0, wne = bl.wx1 * wy0, wsw = wx0 * bl.wy1, wse = bl.wx1 * bl.wy1;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float* pl = sp + c * plane;
            t[c] = tg[c * plane + (size_t)yr * W + xr];
            w[c] = pl[bl.o00] * wnw + pl[bl.o01] * wne + pl[bl.o10] * wsw + pl[bl.o11] * wse;
        }
        // optional log tensors (trainer.py:480-511), real rows / columns only
        if ((col || smp || dep) && i >= 1 && i <= R_ROWS && yy < H && lane_ok) {
            size_t o = (size_t)yy * W + x;
            if (col) {
#pragma unroll
                for (int c = 0; c < 3; ++c) col[((size_t)b * 3 + c) * plane + o] = w[c];
            }
            if (smp) {
                smp[((size_t)b * plane + o) * 2 + 0] = pr.gx;
                smp[((size_t)b * plane + o) * 2 + 1] = pr.gy;
            }
            if (dep) dep[(size_t)b * plane + o] = pr.depth;
        }
        HSum h;
        hsum_row(h, t, w);
        if (i >= 2) my_lds[(i - 2) * 64 + lane] = reproj_value(ha, hb, h, ct, cw, no_ssim);
        ha = hb; hb = h;
#pragma unroll
        for (int c = 0; c < 3; ++c) { ct[c] = t[c]; cw[c] = w[c]; }
    }
    __syncthreads();
    // ---- min over [identity(-1), identity(+1), reproj(-1), reproj(+1)]  (trainer.py:592-610)
    const float* r0p = lds + (size_t)(s * 2 + 0) * R_ROWS * 64;
    const float* r1p = lds + (size_t)(s * 2 + 1) * R_ROWS * 64;
    const bool automask = !(p.flags & DC_OPT_NO_AUTOMASK);
    const bool avg = p.flags & DC_OPT_AVG_REPROJ;
    const float* nz = p.noise[s];
    uint8_t* am = p.argmin[s];
    float* isel = p.idsel[s];
    float acc = 0.f;
    for (int r = f; r < R_ROWS; r += 2) {
        const int py = y0 + r;
        if (py < H && lane_ok) {
            float r0 = r0p[r * 64 + lane], r1 = r1p[r * 64 + lane];
            float best;
            int idx = 0;
            const size_t o = (size_t)py * W + x;
            if (avg) {
                float rr = (r0 + r1) * 0.5f;
                best = rr;
                if (automask) {
                    float n0 = nz ? nz[(size_t)b * plane + o] : rng_normal(p.seed, (unsigned)(b * plane + o), s * 2);
                    float id = __fadd_rn(p.idl[(size_t)b * plane + o], __fmul_rn(n0, 0.00001f));
                    best = id;
                    if (rr < best) { best = rr; idx = 1; }
                }
            } else {
                if (automask) {
                    float n0 = nz ? nz[((size_t)b * 2 + 0) * plane + o]
                                  : rng_normal(p.seed, (unsigned)(b * plane + o), s * 2);
                    float n1 = nz ? nz[((size_t)b * 2 + 1) * plane + o]
                                  : rng_normal(p.seed, (unsigned)(b * plane + o), s * 2 + 1);
                    float i0 = __fadd_rn(p.idl[((size_t)b * 2 + 0) * plane + o], __fmul_rn(n0, 0.00001f));
                    float i1 = __fadd_rn(p.idl[((size_t)b * 2 + 1) * plane + o], __fmul_rn(n1, 0.00001f));
                    best = i0;
                    if (i1 < best) { best = i1; idx = 1; }
                    if (r0 < best) { best = r0; idx = 2; }
                    if (r1 < best) { best = r1; idx = 3; }
                } else {
                    best = r0;
                    if (r1 < best) { best = r1; idx = 1; }
                }
            }
            acc += best;
            am[(size_t)b * plane + o] = (uint8_t)idx;
            if (isel && automask) isel[(size_t)b * plane + o] = (idx > (avg ? 0 : 1)) ? 1.f : 0.f;
        }
    }
    acc = wave_sum(acc);
    if (lane == 0) {
        int blk = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        p.part_photo[((size_t)s * p.nblk_f + blk) * 2 + f] = acc;
    }
}

// ------------------------------------------------------------------------------------------------
// smoothness forward partials (layers.py:202-215 on disp / (mean+1e-7), trainer.py:612-616)
// grid (nchunk, B, ns), block 256; each block covers SM_CHUNK pixels of image b at scale s.
// ------------------------------------------------------------------------------------------------
constexpr int SM_CHUNK = 2048;

__device__ __forceinline__ float block_sum_256(float v, float* sm) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[w] = v;
    __syncthreads();
    return sm[0] + sm[1] + sm[2] + sm[3];
}

__global__ __launch_bounds__(256) void smooth_fwd_kernel(PhotoArgs p) {
    __shared__ float sm[4];
    const int s = blockIdx.z, b = blockIdx.y;
    const int h = p.hs[s], w = p.ws[s];
    const int n = h * w;
    const int base = blockIdx.x * SM_CHUNK;
    float sd = 0.f, sx = 0.f, sy = 0.f;
    if (base < n) {
        const float* d = p.disp[s] + (size_t)b * n;
        const float* im = p.color_s[s] + (size_t)b * 3 * n;
        for (int i = base + threadIdx.x; i < min(base + SM_CHUNK, n); i += 256) {
            const int y = i / w, x = i - y * w;
            const float dv = d[i];
            sd += dv;
            if (x < w - 1) {
                float gi = (fabsf(im[i] - im[i + 1]) + fabsf(im[n + i] - im[n + i + 1]) +
                            fabsf(im[2 * n + i] - im[2 * n + i + 1])) * (1.f / 3.f);
                sx += fabsf(dv - d[i + 1]) * __expf(-gi);
            }
            if (y < h - 1) {
                float gi = (fabsf(im[i] - im[i + w]) + fabsf(im[n + i] - im[n + i + w]) +
                            fabsf(im[2 * n + i] - im[2 * n + i + w])) * (1.f / 3.f);
                sy += fabsf(dv - d[i + w]) * __expf(-gi);
            }
        }
    }
    sd = block_sum_256(sd, sm);
    sx = block_sum_256(sx, sm);
    sy = block_sum_256(sy, sm);
    if (threadIdx.x == 0) {
        float* o = p.part_smooth + (((size_t)s * p.B + b) * p.nchunk + blockIdx.x) * 3;
        o[0] = sd; o[1] = sx; o[2] = sy;
    }
}

// one block: fixed-order final sums -> losses[ns+1], stats[s][b] = (mean, Sx, Sy)
__global__ __launch_bounds__(256) void finalize_kernel(PhotoArgs p) {
    __shared__ float sm[4];
    __shared__ float smooth_s[DC_MAX_SCALES];
    const int ns = p.ns, B = p.B;
    // per (s,b) smoothness stats
    for (int sb = threadIdx.x; sb < ns * B; sb += 256) {
        const int s = sb / B;
        const int n = p.hs[s] * p.ws[s];
        const int nch = (n + SM_CHUNK - 1) / SM_CHUNK;
        const float* q = p.part_smooth + (size_t)sb * p.nchunk * 3;
        float sd = 0.f, sx = 0.f, sy = 0.f;
        for (int k = 0; k < nch; ++k) { sd += q[k * 3]; sx += q[k * 3 + 1]; sy += q[k * 3 + 2]; }
        p.stats[sb * 3 + 0] = sd / (float)n;
        p.stats[sb * 3 + 1] = sx;
        p.stats[sb * 3 + 2] = sy;
    }
    __syncthreads();
    if (threadIdx.x < ns) {
        const int s = threadIdx.x;
        const int h = p.hs[s], w = p.ws[s];
        float tx = 0.f, ty = 0.f;
        for (int b = 0; b < B; ++b) {
            float a = 1.f / (p.stats[(s * B + b) * 3] + 1e-7f);
            tx += p.stats[(s * B + b) * 3 + 1] * a;
            ty += p.stats[(s * B + b) * 3 + 2] * a;
        }
        smooth_s[s] = tx / ((float)B * h * (w - 1)) + ty / ((float)B * (h - 1) * w);
    }
    __syncthreads();
    float total = 0.f;
    for (int s = 0; s < ns; ++s) {
        float acc = 0.f;
        const float* q = p.part_photo + (size_t)s * p.nblk_f * 2;
        for (int k = threadIdx.x; k < p.nblk_f * 2; k += 256) acc += q[k];
        acc = block_sum_256(acc, sm);
        float loss = acc / ((float)B * p.H * p.W) + p.smoothness * smooth_s[s] / (float)(1 << s);
        if (threadIdx.x == 0) p.losses[s] = loss;
        total += loss;
    }
    if (threadIdx.x == 0) p.losses[ns] = total / (float)ns;
}

// ------------------------------------------------------------------------------------------------
// backward: wave (scale s, frame f), halo 2.  grid (strips60, rowblocks, B), block 128*ns.
// ------------------------------------------------------------------------------------------------
struct HAbc {
    float a[3], b[3], c[3];
};

__global__ __launch_bounds__(512) void photo_bwd_kernel(PhotoArgs p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [2*ns][R_ROWS][64]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int s = wave >> 1, f = wave & 1;
    const int b = blockIdx.z;
    const int x = blockIdx.x * 60 - 2 + lane;
    const int y0 = blockIdx.y * R_ROWS;
    const int H = p.H, W = p.W;
    const int xr = reflect_clamp(x, W);
    const size_t plane = (size_t)H * W;
    const float* tg = p.target + (size_t)b * 3 * plane;
    const float* sp = p.src[f] + (size_t)b * 3 * plane;
    const int hs = p.hs[s], ws = p.ws[s];
    const float* dp = p.disp[s] + (size_t)b * hs * ws;
    const float ry = p.ry[s], rx = p.rx[s];
    const bool no_ssim = p.flags & DC_OPT_NO_SSIM;
    const bool ac = p.flags & DC_OPT_ALIGN_CORNERS;
    const bool automask = !(p.flags & DC_OPT_NO_AUTOMASK);
    const bool avg = p.flags & DC_OPT_AVG_REPROJ;
    const bool col_ok = x >= 0 && x < W;
    const bool q_lane = lane >= 2 && lane <= 61 && col_ok;
    Geo g;
    load_geo(g, p.K, p.invK, p.T[f], b);
    const uint8_t* am = p.argmin[s] + (size_t)b * plane;
    // d loss / d to_optimise(pixel) for this scale: mean over B*H*W, total = mean over scales
    const float wgt = uni((p.g_losses[s] + p.g_losses[p.ns] / (float)p.ns) / ((float)p.B * H * W));
    const float g_ssim = no_ssim ? 0.f : 0.85f / 3.f;
    const float g_l1 = no_ssim ? 1.f / 3.f : 0.15f / 3.f;
    const int sel_idx = avg ? 1 : (automask ? 2 + f : f);
    const float sel_val = avg ? 0.5f : 1.f;
    const bool sel_all = avg && !automask;   // single channel: to_optimise = combined
    float* my_lds = lds + (size_t)wave * R_ROWS * 64;

    HSum ha, hb;
    HAbc ka, kb;
    ha = hb = HSum{};
    ka = kb = HAbc{};
    float dP[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) dP[k] = 0.f;

#pragma unroll 1
    for (int i = 0; i < R_ROWS + 4; ++i) {
        const int yy = y0 - 2 + i;
        // ---------------- stage A: target + warped values on (reflected) row yy
        HSum h;
        {
            const int yr = reflect_clamp(yy, H);
            float t[3], w[3];
            float d = disp_at(dp, hs, ws, ry, rx, H, W, xr, yr);
            Proj pr;
            project_pixel(pr, g, d, p.min_disp, p.disp_range, xr, yr, H, W, p.inv_Wm1, p.inv_Hm1, ac);
            Bilin bl = bilin_setup(pr.ix, pr.iy, H, W);
            const float wx0 = 1.f - bl.wx1, wy0 = 1.f - bl.wy1;
            const float wnw = wx0 * wy0, wne = bl.wx1 * wy0, wsw = wx0 * bl.wy1, wse = bl.wx1 * bl.wy1;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float* pl = sp + c * plane;
                t[c] = tg[c * plane + (size_t)yr * W + xr];
                w[c] = pl[bl.o00] * wnw + pl[bl.o01] * wne + pl[bl.o10] * wsw + pl[bl.o11] * wse;
            }
            hsum_row(h, t, w);
        }
        // ---------------- stage B: SSIM derivative coefficients at row p = yy-1
        HAbc k;
        {
            const int py = yy - 1;
            float gs = 0.f;
            if (i >= 2 && py >= 0 && py < H && col_ok) {
                bool sel = sel_all || (am[(size_t)py * W + x] == sel_idx);
                gs = sel ? sel_val * wgt * g_ssim : 0.f;
            }
            float a[3], bb[3], cc[3];
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                SsimTerms t = ssim_terms(ha, hb, h, ch);
                float n = t.n1 * t.n2, dd = t.d1 * t.d2;
                float id = 1.f / dd;
                float v = (1.f - n * id) * 0.5f;
                float G = (v >= 0.f && v <= 1.f) ? gs : 0.f;     // clamp(.,0,1) passes grad inclusively
                float nid2 = n * id * id;
                // d/d mu_x | d/d E[xx] | d/d E[xy]   (E[.] held fixed for mu_x)
                a[ch] = G * (-t.mu_y * (t.n2 - t.n1) * id + nid2 * t.mu_x * (t.d2 - t.d1));
                bb[ch] = G * 0.5f * nid2 * t.d1;
                cc[ch] = -G * t.n1 * id;
            }
            // transposed 3x3 (with ReflectionPad fold-back) along x
            const float fl = (x == 1) ? 2.f : 1.f, fr = (x == W - 2) ? 2.f : 1.f;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                k.a[ch] = fl * from_left(a[ch]) + a[ch] + fr * from_right(a[ch]);
                k.b[ch] = fl * from_left(bb[ch]) + bb[ch] + fr * from_right(bb[ch]);
                k.c[ch] = fl * from_left(cc[ch]) + cc[ch] + fr * from_right(cc[ch]);
            }
        }
        // ---------------- stage C: gradient at row q = yy-2
        if (i >= 4) {
            const int qy = yy - 2;
            float gd = 0.f;
            if (qy < H) {
                const bool ok = q_lane;
                const int xq = ok ? x : xr;
                float d = disp_at(dp, hs, ws, ry, rx, H, W, xq, qy);
                Proj pr;
                project_pixel(pr, g, d, p.min_disp, p.disp_range, xq, qy, H, W, p.inv_Wm1, p.inv_Hm1, ac);
                Bilin bl = bilin_setup(pr.ix, pr.iy, H, W);
                const float wx0 = 1.f - bl.wx1, wy0 = 1.f - bl.wy1;
                bool sel = sel_all || (am[(size_t)qy * W + xq] == sel_idx);
                const float gl = (sel && ok) ? sel_val * wgt * g_l1 : 0.f;
                const float fu = (qy == 1) ? 2.f : 1.f, fd = (qy == H - 2) ? 2.f : 1.f;
                float gix = 0.f, giy = 0.f;
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    const float* pl = sp + ch * plane;
                    float nw = pl[bl.o00], ne = pl[bl.o01], sw = pl[bl.o10], se = pl[bl.o11];
                    float wv = (nw * wx0 + ne * bl.wx1) * wy0 + (sw * wx0 + se * bl.wx1) * bl.wy1;
                    float tq = tg[ch * plane + (size_t)qy * W + xq];
                    float SA = fu * ka.a[ch] + kb.a[ch] + fd * k.a[ch];
                    float SB = fu * ka.b[ch] + kb.b[ch] + fd * k.b[ch];
                    float SC = fu * ka.c[ch] + kb.c[ch] + fd * k.c[ch];
                    float df = wv - tq;
                    float sg = (df > 0.f) ? 1.f : ((df < 0.f) ? -1.f : 0.f);
                    float gw = (SA + 2.f * wv * SB + tq * SC) * k9 + gl * sg;
                    gix += gw * ((ne - nw) * wy0 + (se - sw) * bl.wy1);
                    giy += gw * ((sw - nw) * wx0 + (se - ne) * bl.wx1);
                }
                if (!ok) { gix = 0.f; giy = 0.f; }
                const float du = gix * pr.mx * 2.f * p.inv_Wm1;
                const float dv = giy * pr.my * 2.f * p.inv_Hm1;
                float dq[3];
                dq[0] = du * pr.zi;
                dq[1] = dv * pr.zi;
                dq[2] = -(du * pr.u + dv * pr.v) * pr.zi;
                float dcam[3] = {0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 3; ++r) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        dP[r * 4 + j] += dq[r] * pr.cam[j];
                        dcam[j] += dq[r] * g.P[r * 4 + j];
                    }
                    dP[r * 4 + 3] += dq[r];
                }
                float ddepth = dcam[0] * pr.ray[0] + dcam[1] * pr.ray[1] + dcam[2] * pr.ray[2];
                gd = -ddepth * pr.depth * pr.depth * p.disp_range;
            }
            my_lds[(i - 4) * 64 + lane] = gd;
        }
        ha = hb; hb = h; ka = kb; kb = k;
    }
    // pose-gradient partials: per wave, fixed shuffle tree
#pragma unroll
    for (int k = 0; k < 12; ++k) dP[k] = wave_sum(dP[k]);
    if (lane == 0) {
        const int blk = blockIdx.y * gridDim.x + blockIdx.x;
        float* o = p.part_dP + ((((size_t)s * 2 + f) * p.B + b) * p.nblk_b_img + blk) * 12;
#pragma unroll
        for (int k = 0; k < 12; ++k) o[k] = dP[k];
    }
    __syncthreads();
    // d(upsampled disp) = frame(-1) + frame(+1) contributions
    const float* g0 = lds + (size_t)(s * 2 + 0) * R_ROWS * 64;
    const float* g1 = lds + (size_t)(s * 2 + 1) * R_ROWS * 64;
    float* out = p.gdup[s] + (size_t)b * plane;
    for (int r = f; r < R_ROWS; r += 2) {
        const int qy = y0 + r;
        if (qy < H && q_lane) out[(size_t)qy * W + x] = g0[r * 64 + lane] + g1[r * 64 + lane];
    }
}

// ------------------------------------------------------------------------------------------------
// d_disp[s] = bilinear-upsample^T(gdup[s]) + smoothness gradient.   grid (chunks, B, ns), block 256
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void disp_grad_kernel(PhotoArgs p) {
    const int s = blockIdx.z, b = blockIdx.y;
    const int h = p.hs[s], w = p.ws[s];
    const int n = h * w;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int H = p.H, W = p.W;
    const int y = i / w, x = i - y * w;
    const float* gu = p.gdup[s] + (size_t)b * H * W;
    float acc = 0.f;
    if (h == H && w == W) {
        acc = gu[i];
    } else {
        const float ry = p.ry[s], rx = p.rx[s];
        // destination rows/cols whose taps can touch (y,x): invert src = r*(dst+.5)-.5 with slack
        int ylo = max((int)floorf(((float)y - 1.f + 0.5f) / ry - 0.5f) - 1, 0);
        int yhi = min((int)ceilf(((float)y + 1.f + 0.5f) / ry - 0.5f) + 1, H - 1);
        int xlo = max((int)floorf(((float)x - 1.f + 0.5f) / rx - 0.5f) - 1, 0);
        int xhi = min((int)ceilf(((float)x + 1.f + 0.5f) / rx - 0.5f) + 1, W - 1);
        for (int dy = ylo; dy <= yhi; ++dy) {
            LinTap ty = lin_tap(dy, ry, h);
            float wy = (ty.i0 == y ? 1.f - ty.w1 : 0.f) + (ty.i1 == y ? ty.w1 : 0.f);
            if (wy == 0.f) continue;
            float row = 0.f;
            for (int dx = xlo; dx <= xhi; ++dx) {
                LinTap tx = lin_tap(dx, rx, w);
                float wx = (tx.i0 == x ? 1.f - tx.w1 : 0.f) + (tx.i1 == x ? tx.w1 : 0.f);
                row += wx * gu[(size_t)dy * W + dx];
            }
            acc += wy * row;
        }
    }
    // smoothness:  L = A*(cx*Sx + cy*Sy),  A = 1/(mean+eps)
    const float gsm = (p.g_losses[s] + p.g_losses[p.ns] / (float)p.ns) * p.smoothness / (float)(1 << s);
    const float* st = p.stats + ((size_t)s * p.B + b) * 3;
    const float A = 1.f / (st[0] + 1e-7f);
    const float cx = 1.f / ((float)p.B * h * (w - 1)), cy = 1.f / ((float)p.B * (h - 1) * w);
    const float* d = p.disp[s] + (size_t)b * n;
    const float* im = p.color_s[s] + (size_t)b * 3 * n;
    auto sgn = [](float v) { return (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f); };
    auto ex = [&](int i0, int i1) {
        float gi = (fabsf(im[i0] - im[i1]) + fabsf(im[n + i0] - im[n + i1]) +
                    fabsf(im[2 * n + i0] - im[2 * n + i1])) * (1.f / 3.f);
        return __expf(-gi);
    };
    const float dv = d[i];
    float gx = 0.f, gy = 0.f;
    if (x < w - 1) gx += sgn(dv - d[i + 1]) * ex(i, i + 1);
    if (x > 0) gx -= sgn(d[i - 1] - dv) * ex(i - 1, i);
    if (y < h - 1) gy += sgn(dv - d[i + w]) * ex(i, i + w);
    if (y > 0) gy -= sgn(d[i - w] - dv) * ex(i - w, i);
    float gs = A * (cx * gx + cy * gy) - A * A * (cx * st[1] + cy * st[2]) / (float)n;
    p.d_disp[s][(size_t)b * n + i] = acc + gsm * gs;
}

// d_T[f][b] = K[b][:3,:]^T @ sum_{s,blk} dP      grid (B, 2), block 64
__global__ __launch_bounds__(64) void pose_grad_kernel(PhotoArgs p) {
    __shared__ float dPs[12];
    const int b = blockIdx.x, f = blockIdx.y;
    const int lane = threadIdx.x;
    float acc[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k] = 0.f;
    for (int s = 0; s < p.ns; ++s) {
        const float* q = p.part_dP + (((size_t)s * 2 + f) * p.B + b) * p.nblk_b_img * 12;
        for (int k = lane; k < p.nblk_b_img; k += 64) {
#pragma unroll
            for (int j = 0; j < 12; ++j) acc[j] += q[k * 12 + j];
        }
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k] = wave_sum(acc[k]);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 12; ++k) dPs[k] = acc[k];
    }
    __syncthreads();
    if (lane < 16) {
        const int r = lane >> 2, c = lane & 3;   // d_T[r][c] = sum_i K[i][r] * dP[i][c]
        const float* K = p.K + b * 16;
        float v = K[0 * 4 + r] * dPs[0 * 4 + c] + K[1 * 4 + r] * dPs[1 * 4 + c] + K[2 * 4 + r] * dPs[2 * 4 + c];
        p.d_T[f][b * 16 + lane] = v;
    }
}

// ------------------------------------------------------------------------------------------------
static inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

struct Carve {
    size_t idl, part_photo, part_smooth, stats, gdup[DC_MAX_SCALES], part_dP, total;
    int nblk_f, nchunk, nblk_b_img, strips_f, strips_b, rowblocks;
};

static Carve carve(const dc_photo_desc* d) {
    Carve c;
    const size_t N = (size_t)d->B * d->H * d->W;
    c.strips_f = ceil_div(d->W, 62);
    c.strips_b = ceil_div(d->W, 60);
    c.rowblocks = ceil_div(d->H, R_ROWS);
    c.nblk_f = c.strips_f * c.rowblocks * d->B;
    c.nblk_b_img = c.strips_b * c.rowblocks;
    c.nchunk = ceil_div(d->H * d->W, SM_CHUNK);
    size_t off = 0;
    c.idl = off; off += align256(N * 2 * 4);
    c.part_photo = off; off += align256((size_t)d->num_scales * c.nblk_f * 2 * 4);
    c.part_smooth = off; off += align256((size_t)d->num_scales * d->B * c.nchunk * 3 * 4);
    c.stats = off; off += align256((size_t)d->num_scales * d->B * 3 * 4);
    for (int s = 0; s < DC_MAX_SCALES; ++s) {
        c.gdup[s] = off;
        if (s < d->num_scales) off += align256(N * 4);
    }
    c.part_dP = off; off += align256((size_t)d->num_scales * 2 * d->B * c.nblk_b_img * 12 * 4);
    c.total = off;
    return c;
}

static int fill_args(const dc_photo_desc* d, PhotoArgs& a, Carve& c, bool backward) {
    if (!d || d->B <= 0 || d->H < 4 || d->W < 4 || d->num_scales < 1 || d->num_scales > DC_MAX_SCALES)
        return DC_EINVAL;
    if (!d->target || !d->source[0] || !d->source[1] || !d->K || !d->inv_K || !d->T[0] || !d->T[1] ||
        !d->workspace)
        return DC_EINVAL;
    if (!(d->min_depth > 0.f) || !(d->max_depth > d->min_depth)) return DC_EINVAL;
    c = carve(d);
    if (d->workspace_bytes < c.total) return DC_EWORKSPACE;
    a = PhotoArgs{};
    a.B = d->B; a.H = d->H; a.W = d->W; a.ns = d->num_scales; a.flags = d->flags;
    a.min_disp = 1.f / d->max_depth;
    a.disp_range = 1.f / d->min_depth - 1.f / d->max_depth;
    a.inv_Wm1 = 1.f / (float)(d->W - 1);
    a.inv_Hm1 = 1.f / (float)(d->H - 1);
    a.smoothness = d->smoothness;
    a.target = d->target; a.src[0] = d->source[0]; a.src[1] = d->source[1];
    a.K = d->K; a.invK = d->inv_K; a.T[0] = d->T[0]; a.T[1] = d->T[1];
    a.seed = d->rng_seed;
    char* ws = (char*)d->workspace;
    for (int s = 0; s < d->num_scales; ++s) {
        if (!d->disp[s] || !d->color_s[s] || !d->argmin[s]) return DC_EINVAL;
        a.disp[s] = d->disp[s]; a.color_s[s] = d->color_s[s];
        a.hs[s] = d->H >> s; a.ws[s] = d->W >> s;
        if (a.hs[s] < 2 || a.ws[s] < 2) return DC_EINVAL;
        a.ry[s] = (float)a.hs[s] / (float)d->H;
        a.rx[s] = (float)a.ws[s] / (float)d->W;
        a.noise[s] = d->noise[s]; a.argmin[s] = d->argmin[s];
        a.depth[s] = d->depth[s]; a.idsel[s] = d->identity_selection[s];
        for (int f = 0; f < 2; ++f) { a.sample[s][f] = d->sample[s][f]; a.color[s][f] = d->color[s][f]; }
        a.gdup[s] = (float*)(ws + c.gdup[s]);
        if (backward) {
            if (!d->d_disp[s]) return DC_EINVAL;
            a.d_disp[s] = d->d_disp[s];
        }
    }
    if (backward) {
        if (!d->g_losses || !d->d_T[0] || !d->d_T[1]) return DC_EINVAL;
        a.g_losses = d->g_losses; a.d_T[0] = d->d_T[0]; a.d_T[1] = d->d_T[1];
    } else if (!d->losses) {
        return DC_EINVAL;
    }
    a.losses = d->losses;
    a.idl = (float*)(ws + c.idl);
    a.part_photo = (float*)(ws + c.part_photo);
    a.part_smooth = (float*)(ws + c.part_smooth);
    a.stats = (float*)(ws + c.stats);
    a.part_dP = (float*)(ws + c.part_dP);
    a.nblk_f = c.nblk_f; a.nchunk = c.nchunk; a.nblk_b_img = c.nblk_b_img;
    return DC_OK;
}

}  // namespace dc

using namespace dc;

extern "C" size_t dc_photo_workspace(const dc_photo_desc* d) {
    if (!d || d->B <= 0 || d->H <= 0 || d->W <= 0 || d->num_scales < 1 || d->num_scales > DC_MAX_SCALES) return 0;
    return carve(d).total;
}

extern "C" int dc_photo_fwd(const dc_photo_desc* d, void* stream) {
    PhotoArgs a;
    Carve c;
    int rc = fill_args(d, a, c, false);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = (size_t)2 * a.ns * R_ROWS * 64 * sizeof(float);
    if (!(a.flags & DC_OPT_NO_AUTOMASK)) {
        hipLaunchKernelGGL(identity_kernel, dim3(c.strips_f, c.rowblocks, a.B), dim3(64), 0, st, a);
        DC_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(smooth_fwd_kernel, dim3(c.nchunk, a.B, a.ns), dim3(256), 0, st, a);
    DC_CHECK_LAUNCH();
    hipLaunchKernelGGL(photo_fwd_kernel, dim3(c.strips_f, c.rowblocks, a.B), dim3(128 * a.ns), lds, st, a);
    DC_CHECK_LAUNCH();
    hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(256), 0, st, a);
    DC_CHECK_LAUNCH();
    return DC_OK;
}

extern "C" int dc_photo_bwd(const dc_photo_desc* d, void* stream) {
    PhotoArgs a;
    Carve c;
    int rc = fill_args(d, a, c, true);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = (size_t)2 * a.ns * R_ROWS * 64 * sizeof(float);
    hipLaunchKernelGGL(photo_bwd_kernel, dim3(c.strips_b, c.rowblocks, a.B), dim3(128 * a.ns), lds, st, a);
    DC_CHECK_LAUNCH();
    hipLaunchKernelGGL(disp_grad_kernel, dim3(ceil_div(a.H * a.W, 256), a.B, a.ns), dim3(256), 0, st, a);
    DC_CHECK_LAUNCH();
    hipLaunchKernelGGL(pose_grad_kernel, dim3(a.B, 2), dim3(64), 0, st, a);
    DC_CHECK_LAUNCH();
    return DC_OK;
}

// per scale fwd reads 36N + 16 n_s; bwd re-reads that and writes 4 n_s (fp32 bytes)
extern "C" double dc_photo_algorithmic_bytes(const dc_photo_desc* d, int backward) {
    if (!d) return 0.0;
    const double N = (double)d->B * d->H * d->W;
    double tot = 0.0;
    for (int s = 0; s < d->num_scales; ++s) {
        const double ns = (double)d->B * (d->H >> s) * (d->W >> s);
        tot += 36.0 * N + 16.0 * ns + (backward ? 4.0 * ns : 0.0);
    }
    return tot;
}
