// Gate arithmetic of the ConvGRU temporal fusion (reference networks/rnn.py:101-143 `ConvGRUCell`, :960-1028
// `ConvGRUBlocks_v5`; trainer_gru.py:595-644 `run_gru_v5`).  The cell's two 3x3 convolutions run on the fused conv block
// (dc_conv3x3_fwd with the concat of input and state as its two sources, bias and sigmoid / tanh in the epilogue); what
// remains of the cell are two elementwise steps, and one for the residual that feeds the decoder:
//   rh      = r * h                      r = gates[:, :C]  (reset gate),  gates = sigmoid(conv_gates(cat(x, h)))
//   h_next  = (1 - u) * h + u * cnm      u = gates[:, C:]  (update gate), cnm = tanh(conv_can(cat(x, r * h)))
//   out[i]  = f[i] + (H[i+1] + H[i]) / 2     H = the n+1 hidden states of a sequence of n frames (trainer_gru.py:637-639)
// Forward and backward, float4 where the plane size allows; memory-bound, one pass each.
#include "dc_common.h"

#include <algorithm>

namespace dc {

// gates: (B, 2C, P); h, out: (B, C, P)
__global__ __launch_bounds__(256) void gru_rh_fwd_kernel(const float* __restrict__ gates, const float* __restrict__ h, float* __restrict__ rh,
                                                        int B, int C, int P) {
    const size_t n = (size_t)B * C * P;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const size_t b = i / ((size_t)C * P), r = i - b * (size_t)C * P;
        rh[i] = gates[b * 2 * C * P + r] * h[i];
    }
}
// d_gates[:, :C] = g * h (the update half is written by the blend backward), d_h = g * r
__global__ __launch_bounds__(256) void gru_rh_bwd_kernel(const float* __restrict__ gates, const float* __restrict__ h, const float* __restrict__ g,
                                                        float* __restrict__ d_gates, float* __restrict__ d_h, int B, int C, int P) {
    const size_t n = (size_t)B * C * P;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const size_t b = i / ((size_t)C * P), r = i - b * (size_t)C * P;
        const size_t gi = b * 2 * C * P + r;
        const float gv = g[i];
        d_gates[gi] = gv * h[i];
        d_gates[gi + (size_t)C * P] = 0.f;
        d_h[i] = gv * gates[gi];
    }
}
// The accumulating forms of the sequence backward (dc_gru_*_bwd_acc; ops._GruLevel walks the frames of a sequence backwards and
// every step's gradient of h_i joins what the later steps and the residual already left in d_H[i]):
//   rh:    d_gates[:, :C] = g * h   (ONLY the reset half is written -- the blend backward of the same step wrote the update half),
//          d_h += g * r
__global__ __launch_bounds__(256) void gru_rh_bwd_acc_kernel(const float* __restrict__ gates, const float* __restrict__ h, const float* __restrict__ g,
                                                            float* __restrict__ d_gates, float* __restrict__ d_h, int B, int C, int P) {
    const size_t n = (size_t)B * C * P;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const size_t b = i / ((size_t)C * P), r = i - b * (size_t)C * P;
        const size_t gi = b * 2 * C * P + r;
        const float gv = g[i];
        d_gates[gi] = gv * h[i];
        d_h[i] += gv * gates[gi];
    }
}
//   blend: d_gates = [0 | g * (cnm - h)], d_h += g * (1 - u), d_cnm = g * u
__global__ __launch_bounds__(256) void gru_blend_bwd_acc_kernel(const float* __restrict__ gates, const float* __restrict__ h,
                                                               const float* __restrict__ cnm, const float* __restrict__ g,
                                                               float* __restrict__ d_gates, float* __restrict__ d_h, float* __restrict__ d_cnm,
                                                               int B, int C, int P) {
    const size_t n = (size_t)B * C * P;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const size_t b = i / ((size_t)C * P), r = i - b * (size_t)C * P;
        const size_t gi = b * 2 * C * P + r;
        const float u = gates[gi + (size_t)C * P], gv = g[i];
        d_gates[gi] = 0.f;
        d_gates[gi + (size_t)C * P] = gv * (cnm[i] - h[i]);
        d_h[i] += gv * (1.f - u);
        d_cnm[i] = gv * u;
    }
}
// The sequence trainer stacks the frames of one sequence along the batch at every use (trainer_gru.py:819-821, 886-896, 943-944:
// torch.cat of ("color", f, s, j), ("K", s, j), ("inv_K", s, j) over j): 20 concatenations of 3 tensors each per step.  One launch
// copies every segment to its place (segments as kernel arguments: no table upload).
constexpr int GATHER_MAX = 96;
struct GatherArgs {
    const float* src[GATHER_MAX];
    float* dst[GATHER_MAX];
    unsigned n[GATHER_MAX];          // floats
};
__global__ __launch_bounds__(256) void gather_copy_kernel(GatherArgs a) {
    const int s = blockIdx.y;
    const unsigned n = a.n[s];
    const float* __restrict__ src = a.src[s];
    float* __restrict__ dst = a.dst[s];
    if ((((size_t)src | (size_t)dst) & 15) == 0) {
        const unsigned n4 = n >> 2;
        for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256)
            reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(src)[i];
        for (unsigned i = (n4 << 2) + blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) dst[i] = src[i];
    } else {
        for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) dst[i] = src[i];
    }
}
__global__ __launch_bounds__(256) void gru_blend_fwd_kernel(const float* __restrict__ gates, const float* __restrict__ h,
                                                           const float* __restrict__ cnm, float* __restrict__ out, int B, int C, int P) {
    const size_t n = (size_t)B * C * P;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const size_t b = i / ((size_t)C * P), r = i - b * (size_t)C * P;
        const float u = gates[b * 2 * C * P + (size_t)C * P + r];
        out[i] = (1.f - u) * h[i] + u * cnm[i];          // rnn.py:141
    }
}
// d_gates[:, C:] = g * (cnm - h) (the reset half is zero here), d_h = g * (1 - u), d_cnm = g * u
__global__ __launch_bounds__(256) void gru_blend_bwd_kernel(const float* __restrict__ gates, const float* __restrict__ h,
                                                           const float* __restrict__ cnm, const float* __restrict__ g,
                                                           float* __restrict__ d_gates, float* __restrict__ d_h, float* __restrict__ d_cnm,
                                                           int B, int C, int P) {
    const size_t n = (size_t)B * C * P;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const size_t b = i / ((size_t)C * P), r = i - b * (size_t)C * P;
        const size_t gi = b * 2 * C * P + r;
        const float u = gates[gi + (size_t)C * P], gv = g[i];
        d_gates[gi] = 0.f;
        d_gates[gi + (size_t)C * P] = gv * (cnm[i] - h[i]);
        d_h[i] = gv * (1.f - u);
        d_cnm[i] = gv * u;
    }
}
// f, out: (n, M);  H: (n+1, M)
__global__ __launch_bounds__(256) void gru_residual_fwd_kernel(const float* __restrict__ f, const float* __restrict__ H, float* __restrict__ out,
                                                              int n, size_t M) {
    const size_t tot = (size_t)n * M;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < tot; i += (size_t)gridDim.x * 256)
        out[i] = f[i] + (H[i + M] + H[i]) / 2.f;
}
// d_f = g;  d_H[j] = (g[j] (j < n) + g[j-1] (j >= 1)) / 2
__global__ __launch_bounds__(256) void gru_residual_bwd_kernel(const float* __restrict__ g, float* __restrict__ d_H, int n, size_t M) {
    const size_t tot = (size_t)(n + 1) * M;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < tot; i += (size_t)gridDim.x * 256) {
        const size_t j = i / M;
        float v = 0.f;
        if (j < (size_t)n) v += g[i];
        if (j >= 1) v += g[i - M];
        d_H[i] = v * 0.5f;
    }
}

static inline int gru_grid(size_t n) { return (int)std::min<size_t>((n + 255) / 256, 4096); }

}  // namespace dc

using namespace dc;

static bool gru_ok(int B, int C, int P) { return B > 0 && C > 0 && P > 0 && (size_t)B * 2 * C * P < (1ull << 31); }

extern "C" int dc_gru_rh_fwd(const float* gates, const float* h, float* rh, int B, int C, int P, void* stream) {
    if (!gates || !h || !rh || !gru_ok(B, C, P)) return DC_EINVAL;
    hipLaunchKernelGGL(gru_rh_fwd_kernel, dim3(gru_grid((size_t)B * C * P)), dim3(256), 0, (hipStream_t)stream, gates, h, rh, B, C, P);
    DC_CHECK_LAUNCH();
    return DC_OK;
}
extern "C" int dc_gru_rh_bwd(const float* gates, const float* h, const float* g, float* d_gates, float* d_h, int B, int C, int P,
                             void* stream) {
    if (!gates || !h || !g || !d_gates || !d_h || !gru_ok(B, C, P)) return DC_EINVAL;
    hipLaunchKernelGGL(gru_rh_bwd_kernel, dim3(gru_grid((size_t)B * C * P)), dim3(256), 0, (hipStream_t)stream, gates, h, g, d_gates, d_h,
                       B, C, P);
    DC_CHECK_LAUNCH();
    return DC_OK;
}
extern "C" int dc_gru_blend_fwd(const float* gates, const float* h, const float* cnm, float* h_next, int B, int C, int P, void* stream) {
    if (!gates || !h || !cnm || !h_next || !gru_ok(B, C, P)) return DC_EINVAL;
    hipLaunchKernelGGL(gru_blend_fwd_kernel, dim3(gru_grid((size_t)B * C * P)), dim3(256), 0, (hipStream_t)stream, gates, h, cnm, h_next,
                       B, C, P);
    DC_CHECK_LAUNCH();
    return DC_OK;
}
extern "C" int dc_gru_blend_bwd(const float* gates, const float* h, const float* cnm, const float* g, float* d_gates, float* d_h,
                                float* d_cnm, int B, int C, int P, void* stream) {
    if (!gates || !h || !cnm || !g || !d_gates || !d_h || !d_cnm || !gru_ok(B, C, P)) return DC_EINVAL;
    hipLaunchKernelGGL(gru_blend_bwd_kernel, dim3(gru_grid((size_t)B * C * P)), dim3(256), 0, (hipStream_t)stream, gates, h, cnm, g,
                       d_gates, d_h, d_cnm, B, C, P);
    DC_CHECK_LAUNCH();
    return DC_OK;
}
extern "C" int dc_gru_residual_fwd(const float* f, const float* H, float* out, int n, size_t M, void* stream) {
    if (!f || !H || !out || n <= 0 || M == 0) return DC_EINVAL;
    hipLaunchKernelGGL(gru_residual_fwd_kernel, dim3(gru_grid((size_t)n * M)), dim3(256), 0, (hipStream_t)stream, f, H, out, n, M);
    DC_CHECK_LAUNCH();
    return DC_OK;
}
extern "C" int dc_gru_residual_bwd(const float* g, float* d_H, int n, size_t M, void* stream) {
    if (!g || !d_H || n <= 0 || M == 0) return DC_EINVAL;
    hipLaunchKernelGGL(gru_residual_bwd_kernel, dim3(gru_grid((size_t)(n + 1) * M)), dim3(256), 0, (hipStream_t)stream, g, d_H, n, M);
    DC_CHECK_LAUNCH();
    return DC_OK;
}

extern "C" int dc_gru_rh_bwd_acc(const float* gates, const float* h, const float* g, float* d_gates, float* d_h, int B, int C, int P,
                                 void* stream) {
    if (!gates || !h || !g || !d_gates || !d_h || !gru_ok(B, C, P)) return DC_EINVAL;
    hipLaunchKernelGGL(gru_rh_bwd_acc_kernel, dim3(gru_grid((size_t)B * C * P)), dim3(256), 0, (hipStream_t)stream, gates, h, g, d_gates,
                       d_h, B, C, P);
    DC_CHECK_LAUNCH();
    return DC_OK;
}
extern "C" int dc_gru_blend_bwd_acc(const float* gates, const float* h, const float* cnm, const float* g, float* d_gates, float* d_h,
                                    float* d_cnm, int B, int C, int P, void* stream) {
    if (!gates || !h || !cnm || !g || !d_gates || !d_h || !d_cnm || !gru_ok(B, C, P)) return DC_EINVAL;
    hipLaunchKernelGGL(gru_blend_bwd_acc_kernel, dim3(gru_grid((size_t)B * C * P)), dim3(256), 0, (hipStream_t)stream, gates, h, cnm, g,
                       d_gates, d_h, d_cnm, B, C, P);
    DC_CHECK_LAUNCH();
    return DC_OK;
}

extern "C" int dc_gather_copy(const float* const* src, float* const* dst, const size_t* n, int nseg, void* stream) {
    if (!src || !dst || !n || nseg <= 0 || nseg > GATHER_MAX) return DC_EINVAL;
    GatherArgs a{};
    size_t most = 0;
    for (int i = 0; i < nseg; ++i) {
        if (!src[i] || !dst[i] || n[i] == 0 || n[i] > 0xffffffffull) return DC_EINVAL;
        a.src[i] = src[i]; a.dst[i] = dst[i]; a.n[i] = (unsigned)n[i];
        most = std::max(most, n[i]);
    }
    const unsigned gx = (unsigned)std::min<size_t>((most / 4 + 1023) / 1024 + 1, 256);
    hipLaunchKernelGGL(gather_copy_kernel, dim3(gx, nseg), dim3(256), 0, (hipStream_t)stream, a);
    DC_CHECK_LAUNCH();
    return DC_OK;
}
