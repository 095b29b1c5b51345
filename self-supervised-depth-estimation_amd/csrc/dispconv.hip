// The disparity heads: Conv3x3(C -> 1) + sigmoid of the depth decoder (reference networks/depth_decoder.py:44-45,63-64
// with layers.py:119-136).  One output channel wastes 15/16 of a matrix-core tile, and the op is bound by reading C
// planes (forward, weight gradient) or writing them (data gradient), so these are plain-FMA kernels:
//   forward        y[p]      = act(b + sum_c sum_t w[c][t] * xpad[c][p + t])                 16x16 pixel tiles, 8 channels / LDS chunk
//   data gradient  dx[c][q]  = sum_t w[c][t] * G_t[q],  G_t[q] = sum of g'[.] over the padded positions that alias to q
//                              (ReflectionPad2d folds row -1 onto 1 and H onto H-2, same for columns): the 9 folded
//                              sums are built once per pixel and reused for every channel -- no padded scratch, no fold pass
// g' = gy * act'(y) is formed on the fly.  No upsampling / concat (the heads never have them).  The weight gradient of
// wide low-resolution heads stays on the Winograd / direct split-K kernels (conv3x3.hip); the thin heads (16 / 32 channels) have
// dispconv_wgrad_kernel below.
#include "dc_common.h"
#include "dispconv.h"
#include "conv_bf16.h"

#include <algorithm>
#include <cstdlib>

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

// Four consecutive pixels of a row per thread, no LDS patch: a block is 8 rows x 128 columns (32 x 8 threads), a thread reads,
// per channel and window row, one aligned 16-byte vector and the two columns beside it (lines its neighbours fetch anyway:
// the three row shifts and the side columns hit in L1 / L2), and 16-byte stores.  The 16 x 16 tiles of dispconv_fwd_kernel moved
// 72-byte row segments (half lines) through scalar loads into LDS: 1.9 TB/s for 16 -> 1 at 192 x 640.  Same products in the
// same order (channel, ky, kx): results are bitwise those of dispconv_fwd_kernel.  W % 4 == 0.
// grid (ceil(W / 128), ceil(H / 8), B)
__global__ __launch_bounds__(256) void dispconv_fwd4_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias, float* __restrict__ y, int C, int H,
                                                            int W, int act, int pad) {
    extern __shared__ float wsm[];                       // C * 9 weights
    for (int e = threadIdx.x; e < C * 9; e += 256) wsm[e] = w[e];
    __syncthreads();
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int ox = blockIdx.x * 128 + tx * 4, oy = blockIdx.y * 8 + ty, b = blockIdx.z;
    if (ox >= W || oy >= H) return;
    int ry[3];
    bool rok[3], lok, rrok;
    ry[0] = dpad_index(oy - 1, H, pad, rok[0]); ry[1] = oy; rok[1] = true; ry[2] = dpad_index(oy + 1, H, pad, rok[2]);
    const int xl = dpad_index(ox - 1, W, pad, lok), xr = dpad_index(ox + 4, W, pad, rrok);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const size_t HW = (size_t)H * W;
    const float* pl = x + (size_t)b * C * HW;
#pragma unroll 4
    for (int c = 0; c < C; ++c, pl += HW) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float* row = pl + (size_t)ry[r] * W;
            const float4 m = *reinterpret_cast<const float4*>(row + ox);
            const float l = row[xl], rr = row[xr];
            const bool ok = rok[r];
            const float v[6] = {(ok && lok) ? l : 0.f, ok ? m.x : 0.f, ok ? m.y : 0.f, ok ? m.z : 0.f, ok ? m.w : 0.f, (ok && rrok) ? rr : 0.f};
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float wv = wsm[c * 9 + r * 3 + kx];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = fmaf(wv, v[j + kx], acc[j]);
            }
        }
    }
    const float bv = bias ? bias[0] : 0.f;
    *reinterpret_cast<float4*>(y + ((size_t)b * H + oy) * W + ox) =
        make_float4(act_fwd(acc[0] + bv, act), act_fwd(acc[1] + bv, act), act_fwd(acc[2] + bv, act), act_fwd(acc[3] + bv, act));
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
                                                          int act, int pad, const float* __restrict__ addend) {
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
        const size_t o = ((size_t)b * C + c) * H * W + q;
        dx[o] = addend ? v + addend[o] : v;      // (the other consumer's gradient of the head's input, see conv_fold_kernel)
    }
}

// ---- weight + bias gradient:  dw[c][t] = sum_{b,q} x[b][c][q] * G_t[b][q]  (the SAME folded window as the data gradient: the
// output pixel q - t reads padded position q),  dbias = sum g'.  A head has ONE output channel: on the split-K matrix-core
// kernel of conv3x3.hip it ran as a 16-channel tile (15/16 wasted: 89 us for 16 -> 1 at 192 x 640, B = 12, where reading x
// takes ~17 us).  Here: a block walks DWP pixels of one image in steps of 256 -- thread = pixel builds G (18 L2-resident
// loads) and parks it in LDS; wave w then owns channels [w CPW, (w+1) CPW) and multiplies their x values of the 256 pixels
// (coalesced 256-byte rows) into CPW x 9 register accumulators -- one wave tree per accumulator at the end, partials
// part[block][c*9+t] / pbias[block] summed in fixed order by conv_wreduce_kernel.  Deterministic.
constexpr int DWP = 2048;              // pixels per block
template <int CPW>
__global__ __launch_bounds__(256) void dispconv_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                             const float* __restrict__ y, float* __restrict__ part,
                                                             float* __restrict__ pbias, int C, int H, int W, int act, int pad) {
    __shared__ __attribute__((aligned(16))) float Gs[9][256];
    __shared__ float bs[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int blk = blockIdx.x, b = blockIdx.y, HW = H * W;
    const float* g = gy + (size_t)b * HW;
    const float* yy = y + (size_t)b * HW;
    const float* xb = x + ((size_t)b * C + wave * CPW) * HW;
    float acc[CPW][9];
#pragma unroll
    for (int j = 0; j < CPW; ++j)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[j][t] = 0.f;
    float bsum = 0.f;
    for (int it = 0; it < DWP / 256; ++it) {
        const int q0 = blk * DWP + it * 256;
        if (q0 >= HW) break;                                   // (block-uniform)
        // this wave's x values of the 256 pixels: requested first, they fly while G is built.  A lane takes four CONSECUTIVE
        // pixels (one 16-byte load per channel, HW % 4 == 0: host-checked; dword loads of pixels 64 apart ran at 1.0-1.6 TB/s)
        float4 xv[CPW];
        {
            const int qq = q0 + 4 * lane;
            const bool in = qq < HW;
#pragma unroll
            for (int j = 0; j < CPW; ++j) {
                xv[j] = *reinterpret_cast<const float4*>(xb + (size_t)j * HW + (in ? qq : 0));
                if (!in) xv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        const int q = q0 + tid;
        float G[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) G[t] = 0.f;
        if (q < HW) {
            const int qy = q / W, qx = q - qy * W;
            int ry[2], rx[2], ny = 1, nx = 1;
            ry[0] = qy; rx[0] = qx;
            if (pad == PAD_REFLECT) {
                if (qy == 1) ry[ny++] = -1;
                if (qy == H - 2) ry[ny++] = H;
                if (qx == 1) rx[nx++] = -1;
                if (qx == W - 2) rx[nx++] = W;
            }
#pragma unroll
            for (int t = 0; t < 9; ++t) G[t] = gprime_at(g, yy, H, W, qy + 1 - t / 3, qx + 1 - t % 3, act);
            bsum += G[4];                                      // the centre tap's own term is g'[q]: every pixel once
            for (int a = 0; a < ny; ++a)
                for (int c = (a == 0 ? 1 : 0); c < nx; ++c) {
#pragma unroll
                    for (int t = 0; t < 9; ++t) G[t] += gprime_at(g, yy, H, W, ry[a] + 1 - t / 3, rx[c] + 1 - t % 3, act);
                }
        }
        __syncthreads();                                       // the previous step's readers are done with Gs
#pragma unroll
        for (int t = 0; t < 9; ++t) Gs[t][tid] = G[t];
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float4 gg = *reinterpret_cast<const float4*>(&Gs[t][4 * lane]);          // (0 for pixels past the image)
#pragma unroll
            for (int j = 0; j < CPW; ++j)
                acc[j][t] = fmaf(xv[j].w, gg.w, fmaf(xv[j].z, gg.z, fmaf(xv[j].y, gg.y, fmaf(xv[j].x, gg.x, acc[j][t]))));
        }
    }
    float* po = part + ((size_t)b * gridDim.x + blk) * (size_t)C * 9 + (size_t)wave * CPW * 9;
#pragma unroll
    for (int j = 0; j < CPW; ++j)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float v = wave_sum(acc[j][t]);
            if (lane == 0) po[j * 9 + t] = v;
        }
    bsum = wave_sum(bsum);
    if (lane == 0) bs[wave] = bsum;
    __syncthreads();
    if (tid == 0) pbias[(size_t)b * gridDim.x + blk] = (bs[0] + bs[1]) + (bs[2] + bs[3]);
}

// measured on the decoder pyramid (tools/bench_convblock.py): these kernels win the forward and the data gradient of the
// thin high-resolution heads (16 / 32 channels: 61 vs 88 us, 77 vs 174 us at 192x640); the wide low-resolution heads and
// every weight gradient stay on the Winograd / direct kernels
bool dispconv_eligible(int C0, int C1, int up0, int Co, int H, int W) {
    return Co == 1 && C1 == 0 && !up0 && H >= 4 && W >= 4 && C0 <= 32;
}

int dispconv_fwd(const float* x, const float* w, const float* bias, float* y, int B, int C, int H, int W, int act, int pad,
                 hipStream_t st) {
    static const bool quads = !(std::getenv("DC_DISP4") && std::getenv("DC_DISP4")[0] == '0');      // (0: the tile kernels, for A/Bs)
    if (quads && W % 4 == 0 && !(((size_t)x | (size_t)y) & 15)) {
        hipLaunchKernelGGL(dispconv_fwd4_kernel, dim3(ceil_div(W, 128), ceil_div(H, 8), B), dim3(256), (size_t)C * 9 * sizeof(float), st, x, w,
                           bias, y, C, H, W, act, pad);
        DC_CHECK_LAUNCH();
        return DC_OK;
    }
    const int tiles_x = ceil_div(W, DT), tiles_y = ceil_div(H, DT);
    hipLaunchKernelGGL(dispconv_fwd_kernel, dim3(tiles_x * tiles_y, B), dim3(256), 0, st, x, w, bias, y, C, H, W, act, pad, tiles_x);
    DC_CHECK_LAUNCH();
    return DC_OK;
}

int dispconv_dx(const float* w, const float* y, const float* gy, float* dx, const float* addend, int B, int C, int H, int W, int act,
                int pad, hipStream_t st) {
    // (a four-pixels-per-thread form of this kernel measured SLOWER, 38 vs 35 us per launch: the 36 accumulators of four windows
    // cost more occupancy than the wider stores gain)
    hipLaunchKernelGGL(dispconv_dx_kernel, dim3(ceil_div(H * W, 256), B), dim3(256), (size_t)C * 9 * sizeof(float), st, gy, y, w, dx, C, H,
                       W, act, pad, addend);
    DC_CHECK_LAUNCH();
    return DC_OK;
}

bool dispconv_wgrad_eligible(int C0, int C1, int up0, int Co, int H, int W) {
    return dispconv_eligible(C0, C1, up0, Co, H, W) && (C0 == 16 || C0 == 32) && (H * W) % 4 == 0;     // (16-byte loads of x)
}
size_t dispconv_wgrad_scratch(int B, int C, int H, int W) {
    return (size_t)B * ceil_div(H * W, DWP) * ((size_t)C * 9 + 1) * sizeof(float);
}

// dweight (1, C, 3, 3) and dbias (1) from x (B, C, H, W), y / gy (B, 1, H, W); scratch: dispconv_wgrad_scratch bytes
int dispconv_wgrad(const float* x, const float* y, const float* gy, float* dweight, float* dbias, float* scratch, int B, int C, int H,
                   int W, int act, int pad, hipStream_t st) {
    const int nblk = ceil_div(H * W, DWP), split = B * nblk;
    float* part = scratch;
    float* pbias = scratch + (size_t)split * C * 9;
    if (C == 16) hipLaunchKernelGGL(dispconv_wgrad_kernel<4>, dim3(nblk, B), dim3(256), 0, st, x, gy, y, part, pbias, C, H, W, act, pad);
    else if (C == 32) hipLaunchKernelGGL(dispconv_wgrad_kernel<8>, dim3(nblk, B), dim3(256), 0, st, x, gy, y, part, pbias, C, H, W, act, pad);
    else return DC_EINVAL;
    DC_CHECK_LAUNCH();
    return conv_wreduce(dweight ? part : nullptr, dbias ? pbias : nullptr, dweight, dbias, split, dweight ? C * 9 : 0, dbias ? 1 : 0, st);
}

}  // namespace dc
