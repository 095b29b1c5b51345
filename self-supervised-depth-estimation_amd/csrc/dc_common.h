// Shared device helpers for libdepthcore (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "depthcore.h"

namespace dc {

constexpr int kWave = 64;

// ---- cross-lane: whole-wave shifts by one lane (DPP wave_shr / wave_shl) --------------------
// lane i receives the value held by lane i-1 / i+1; lane 0 / 63 receive 0.
#ifndef DC_USE_BPERMUTE
__device__ __forceinline__ float from_left(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float from_right(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}
#else
__device__ __forceinline__ float from_left(float v) { return __shfl_up(v, 1); }
__device__ __forceinline__ float from_right(float v) { return __shfl_down(v, 1); }
#endif

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__device__ __forceinline__ int reflect_clamp(int i, int n) {
    // ReflectionPad2d(1) index map (-1 -> 1, n -> n-2), then clamped for the don't-care lanes.
    i = i < 0 ? -i : i;
    i = i >= n ? 2 * n - 2 - i : i;
    return min(max(i, 0), n - 1);
}

__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }

// ---- F.interpolate(bilinear, align_corners=False) source taps along one axis ------------------
struct LinTap {
    int i0, i1;
    float w1;
};
__device__ __forceinline__ LinTap lin_tap(int dst, float scale, int n_in) {
    float src = fmaxf(scale * ((float)dst + 0.5f) - 0.5f, 0.f);
    int i0 = min((int)src, n_in - 1);
    LinTap t;
    t.i0 = i0;
    t.i1 = min(i0 + 1, n_in - 1);
    t.w1 = src - (float)i0;
    return t;
}

// ---- grid_sample coordinate handling (border padding) -------------------------------------------
// returns the clamped source coordinate; `mult` = d(coord)/d(grid) including the zero of the clamp.
__device__ __forceinline__ float unnormalize_clip(float g, int size, bool align_corners, float& mult) {
    float x, m;
    if (align_corners) {
        m = 0.5f * (float)(size - 1);
        x = (g + 1.f) * m;
    } else {
        m = 0.5f * (float)size;
        x = ((g + 1.f) * (float)size - 1.f) * 0.5f;
    }
    float hi = (float)(size - 1);
    // ATen clip_coordinates_set_grad: zero gradient where x <= 0 or x >= size-1
    mult = (x > 0.f && x < hi) ? m : 0.f;
    return fminf(fmaxf(x, 0.f), hi);   // NaN -> 0 like ATen's clip (fmaxf(NaN,0)=0)
}

struct Bilin {
    int o00, o01, o10, o11;   // element offsets inside one channel plane
    float wx1, wy1;
};
__device__ __forceinline__ Bilin bilin_setup(float x, float y, int H, int W) {
    float xf = floorf(x), yf = floorf(y);
    int x0 = (int)xf, y0 = (int)yf;
    int x1 = min(x0 + 1, W - 1), y1 = min(y0 + 1, H - 1);   // the clamped tap carries weight 0
    Bilin b;
    b.wx1 = x - xf;
    b.wy1 = y - yf;
    b.o00 = y0 * W + x0;
    b.o01 = y0 * W + x1;
    b.o10 = y1 * W + x0;
    b.o11 = y1 * W + x1;
    return b;
}

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
__device__ __forceinline__ int ceil_div_dev(int a, int b) { return (a + b - 1) / b; }

// ---- XCD-aware block order ---------------------------------------------------------------------------
// The hardware hands consecutive workgroup ids to the 8 XCDs round-robin, and every XCD has its own L2.  This maps the
// flat workgroup id to a logical index such that CONSECUTIVE logical indices run on the SAME XCD (a bijection on
// [0, nblocks)): blocks that share an operand are given adjacent logical indices and so find it in their L2.
__device__ __forceinline__ int xcd_logical_block(int bid, int nblocks) {
    const int q = nblocks >> 3, r = nblocks & 7, x = bid & 7;
    return x * q + min(x, r) + (bid >> 3);
}

// n / d for 0 <= n with n * d < 2^32, from the host's magic = ceil(2^32 / d) (0 stands for d = 1; exact in that range): one
// mul_hi instead of the ~25-instruction reciprocal sequence.  The Winograd kernels' block prologue divides ten times by
// launch constants, and in a lock-step round of blocks nothing hides a prologue (same-box A/B: -2.6 % on the forward /
// data-gradient launches of a step).
__device__ __forceinline__ int fdiv(int n, unsigned magic) { return magic ? (int)__umulhi((unsigned)n, magic) : n; }
static inline unsigned fdiv_magic(int d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); }

// ---- fused conv block options (include/depthcore.h: dc_conv3x3_*) ---------------------------------
enum { ACT_NONE = 0, ACT_ELU = 1, ACT_SIGMOID = 2, ACT_RELU = 3, ACT_TANH = 4, ACT_LAST = ACT_TANH };
enum { PAD_REFLECT = 0, PAD_ZERO = 1 };

__device__ __forceinline__ float act_fwd(float v, int act) {
    if (act == ACT_ELU) return v > 0.f ? v : __expf(v) - 1.f;
    if (act == ACT_SIGMOID) return 1.f / (1.f + __expf(-v));
    if (act == ACT_RELU) return fmaxf(v, 0.f);
    if (act == ACT_TANH) return tanhf(v);
    return v;
}
// derivative expressed through the activated output y
__device__ __forceinline__ float act_bwd(float y, int act) {
    if (act == ACT_ELU) return y > 0.f ? 1.f : y + 1.f;
    if (act == ACT_SIGMOID) return y * (1.f - y);
    if (act == ACT_RELU) return y > 0.f ? 1.f : 0.f;
    if (act == ACT_TANH) return 1.f - y * y;
    return 1.f;
}

// y += a (n floats; float4 body + scalar tail), pointwise.hip.  For the data-gradient paths that cannot take the second
// gradient of a residual fork in their epilogue (dc_*_dgrad_add): stride-2 1x1, the general 1x1 kernels, the bf16 policy.
int add_inplace(float* y, const float* a, size_t n, hipStream_t st);

}  // namespace dc

#define DC_CHECK_LAUNCH()                          \
    do {                                           \
        if (hipGetLastError() != hipSuccess) return DC_ELAUNCH; \
    } while (0)
