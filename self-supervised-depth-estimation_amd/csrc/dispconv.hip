// The disparity heads: Conv3x3(C -> 1) + sigmoid of the depth decoder (reference networks/depth_decoder.py:44-45,63-64
// with layers.py:119-136).  One output channel wastes 15/16 of a matrix-core tile, and the op is bound by reading C
// planes (forward, weight gradient) or writing them (data gradient), so these are plain-FMA kernels:
//   forward        y[p]      = act(b + sum_c sum_t w[c][t] * xpad[c][p + t])                 16x16 pixel tiles, 8 channels / LDS chunk
//   data gradient  dx[c][q]  = sum_t w[c][t] * G_t[q],  G_t[q] = sum of g'[.] over the padded positions that alias to q
//                              (ReflectionPad2d folds row -1 onto 1 and H onto H-2, same for columns): the 9 folded
//                              sums are built once per pixel and reused for every channel -- no padded scratch, no fold pass
// g' = gy * act'(y) is formed on the fly.  No upsampling / concat (the heads never have them).  The weight gradient of
// the heads stays on the direct split-K kernel (conv3x3.hip).
#include "dc_common.h"
#include "dispconv.h"

#include <algorithm>

namespace dc {

constexpr int DT = 16;                 // pixel tile
constexpr int DCK = 8;                 // channels per LDS chunk
constexpr int DP = DT + 2;             // patch edge

__device__ __forceinline__ int dpad_index(int i, int n, int pad, bool& ok) {
    ok = true;
    if (i >= 0 && i < n) return i;
    if (pad == PAD_REFLECT) {
        i = i < 0 ? -i : 2 * n - 2 - i;
        return min(max(i, 0), n - 1);
    }
    ok = false;
    return 0;
}

// grid (tiles_x * tiles_y, B), block 256 = one thread per pixel of the tile
__global__ __launch_bounds__(256) void dispconv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ y, int C, int H,
                                                           int W, int act, int pad, int tiles_x) {
    __shared__ float patch[DCK][DP][DP + 1];
    __shared__ float wl[DCK][9];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int tile = blockIdx.x, b = blockIdx.y;
    const int oy0 = (tile / tiles_x) * DT, ox0 = (tile % tiles_x) * DT;
    float acc = 0.f;
    for (int c0 = 0; c0 < C; c0 += DCK) {
        __syncthreads();
        for (int e = tid; e < DCK * DP * DP; e += 256) {
            const int kc = e / (DP * DP), rem = e - kc * (DP * DP);
            const int r = rem / DP, c = rem - r * DP;
            float v = 0.f;
            if (c0 + kc < C) {
                bool oky, okx;
                const int yy = dpad_index(oy0 + r - 1, H, pad, oky), xx = dpad_index(ox0 + c - 1, W, pad, okx);
                if (oky && okx && oy0 + r - 1 <= H && ox0 + c - 1 <= W) v = x[(((size_t)b * C + c0 + kc) * H + yy) * W + xx];
            }
            patch[kc][r][c] = v;
        }
        if (tid < DCK * 9) {
            const int kc = tid / 9, t = tid - kc * 9;
            wl[kc][t] = c0 + kc < C ? w[(size_t)(c0 + kc) * 9 + t] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kc = 0; kc < DCK; ++kc)
#pragma unroll
            for (int t = 0; t < 9; ++t) acc = fmaf(wl[kc][t], patch[kc][ty + t / 3][tx + t % 3], acc);
    }
    const int oy = oy0 + ty, ox = ox0 + tx;
    if (oy < H && ox < W) y[((size_t)b * H + oy) * W + ox] = act_fwd(acc + (bias ? bias[0] : 0.f), act);
}

// folded g' window of pixel (qy, qx): G[ky][kx] = sum over padded rows r in Ry(qy), columns c in Rx(qx) of g'[r+1-ky][c+1-kx]
__device__ __forceinline__ float gprime_at(const float* gy, const float* y, int H, int W, int r, int c, int act) {
    // branch-free: load from the clamped position, select afterwards (an early return around the loads made the compiler
    // wait for each of the 18 loads of a window before issuing the next)
    const bool ok = r >= 0 && r < H && c >= 0 && c < W;
    const size_t o = (size_t)min(max(r, 0), H - 1) * W + min(max(c, 0), W - 1);
    const float v = gy[o] * act_bwd(y[o], act);
    return ok ? v : 0.f;
}

// grid (ceil(H*W / 256), B): one thread per pixel, loops over the channels (coalesced plane writes)
__global__ __launch_bounds__(256) void dispconv_dx_kernel(const float* __restrict__ gy, const float* __restrict__ y,
                                                          const float* __restrict__ w, float* __restrict__ dx, int C, int H, int W,
                                                          int act, int pad) {
    extern __shared__ float wsm[];                       // C * 9 weights
    for (int e = threadIdx.x; e < C * 9; e += 256) wsm[e] = w[e];
    __syncthreads();
    const int q = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (q >= H * W) return;
    const int qy = q / W, qx = q - qy * W;
    const float* g = gy + (size_t)b * H * W;
    const float* yy = y + (size_t)b * H * W;
    // padded rows / columns that alias to this pixel (H, W >= 4, host-checked: rows 1 and H-2 are distinct)
    int ry[2], rx[2], ny = 1, nx = 1;
    ry[0] = qy; rx[0] = qx;
    if (pad == PAD_REFLECT) {
        if (qy == 1) ry[ny++] = -1;
        if (qy == H - 2) ry[ny++] = H;
        if (qx == 1) rx[nx++] = -1;
        if (qx == W - 2) rx[nx++] = W;
    }
    float G[9];
    // the pixel's own window first, all 18 loads in flight together; the mirrored rows / columns (border pixels only) after it
#pragma unroll
    for (int t = 0; t < 9; ++t) G[t] = gprime_at(g, yy, H, W, qy + 1 - t / 3, qx + 1 - t % 3, act);
    for (int a = 0; a < ny; ++a)
        for (int c = (a == 0 ? 1 : 0); c < nx; ++c) {
#pragma unroll
            for (int t = 0; t < 9; ++t) G[t] += gprime_at(g, yy, H, W, ry[a] + 1 - t / 3, rx[c] + 1 - t % 3, act);
        }
    for (int c = 0; c < C; ++c) {
        float v = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) v = fmaf(wsm[c * 9 + t], G[t], v);
        dx[((size_t)b * C + c) * H * W + q] = v;
    }
}

// measured on the decoder pyramid (tools/bench_convblock.py): these kernels win the forward and the data gradient of the
// thin high-resolution heads (16 / 32 channels: 61 vs 88 us, 77 vs 174 us at 192x640); the wide low-resolution heads and
// every weight gradient stay on the Winograd / direct kernels
bool dispconv_eligible(int C0, int C1, int up0, int Co, int H, int W) {
    return Co == 1 && C1 == 0 && !up0 && H >= 4 && W >= 4 && C0 <= 32;
}

int dispconv_fwd(const float* x, const float* w, const float* bias, float* y, int B, int C, int H, int W, int act, int pad,
                 hipStream_t st) {
    const int tiles_x = ceil_div(W, DT), tiles_y = ceil_div(H, DT);
    hipLaunchKernelGGL(dispconv_fwd_kernel, dim3(tiles_x * tiles_y, B), dim3(256), 0, st, x, w, bias, y, C, H, W, act, pad, tiles_x);
    DC_CHECK_LAUNCH();
    return DC_OK;
}

int dispconv_dx(const float* w, const float* y, const float* gy, float* dx, int B, int C, int H, int W, int act, int pad,
                hipStream_t st) {
    hipLaunchKernelGGL(dispconv_dx_kernel, dim3(ceil_div(H * W, 256), B), dim3(256), (size_t)C * 9 * sizeof(float), st, gy, y, w, dx, C, H,
                       W, act, pad);
    DC_CHECK_LAUNCH();
    return DC_OK;
}

}  // namespace dc
