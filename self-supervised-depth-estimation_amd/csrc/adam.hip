// Adam update of every trainable tensor of the step in one pass (gfx950).
//
// Replaces `self.model_optimizer.step()` of the reference (trainer.py:110-113 builds optim.Adam(parameters_to_train,
// learning_rate); trainer.py:236-238 zero_grad / backward / step) with the arithmetic of torch.optim.Adam
// (betas, eps; no weight decay, no amsgrad, not maximize):
//     m = m + (g - m) (1 - b1);   v = b2 v + (1 - b2) g g;   t += 1
//     p = p - (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// in fp32, the same operation order as ATen's fused kernel.  The pass is pure HBM streaming -- 4 tensors read, 3 written,
// 28 bytes per parameter -- and sits at the serial end of the step where nothing overlaps it.  ATen's multi-tensor kernel
// reaches 3.5 TB/s on this GPU at these sizes (tools/ubench/ew_bw.py) where a plain elementwise pass reaches 6-7: this
// kernel keeps 16 independent 16-byte loads in flight per thread (4 float4 of each of p, g, m, v issued before the first
// use) over 4096-element chunks, one launch for up to 384 tensors.
//
// Layout: a static device table of slots {p, m, v, t, numel} and a static chunk list {tensor, first element} (both built by
// the host once, dc_adam_* takes their device pointers); only the gradient pointers change from step to step and travel
// as kernel arguments.  A tensor without a gradient this step (pointer 0) is skipped and its step count stays, as in
// torch.  The per-tensor step counts live in device memory and are advanced by a one-block kernel in front of the update,
// so the launches can be captured in a hipGraph (no host-side step number baked into the arguments).
#include "dc_common.h"

namespace dc {

constexpr int ADAM_CHUNK = 4096;        // elements per block
constexpr int ADAM_MAXT = 384;          // gradient pointers per launch (kernel-argument budget: 384 * 8 B + the rest < 4 KiB)

struct AdamSlot { float* p; float* m; float* v; float* t; long long n; };

struct AdamArgs {
    const AdamSlot* slots;              // device, all tensors
    const int2* chunks;                 // device, {tensor, first element}, sorted by tensor
    int t0, nt;                         // tensors of this launch
    float lr, eps;
    double b1, b2;                      // (doubles: 1 - 0.999f differs from 0.001 by 1.3e-5 relative, and so would the bias corrections)
    const float* g[ADAM_MAXT];          // gradient of tensor t0 + i (0: none this step)
};

__global__ __launch_bounds__(64) void adam_tick_kernel(AdamArgs a) {
    for (int i = threadIdx.x; i < a.nt; i += 64)
        if (a.g[i]) *a.slots[a.t0 + i].t += 1.f;
}

__global__ __launch_bounds__(256) void adam_apply_kernel(AdamArgs a, int chunk0) {
    const int2 ch = a.chunks[chunk0 + blockIdx.x];
    const float* __restrict__ g = a.g[ch.x - a.t0];
    if (!g) return;
    const AdamSlot s = a.slots[ch.x];
    const float t = *s.t;
    // the scalar factors in double, once per block, as torch computes them on the host; the element arithmetic is fp32
    const double bc1 = 1.0 - pow(a.b1, (double)t), bc2 = 1.0 - pow(a.b2, (double)t);
    const float step_size = (float)((double)a.lr / bc1), bc2_sqrt = (float)sqrt(bc2);
    const float w1 = (float)(1.0 - a.b1), w2 = (float)(1.0 - a.b2), b2f = (float)a.b2;
    float* __restrict__ p = s.p; float* __restrict__ m = s.m; float* __restrict__ v = s.v;
    const long long lo = ch.y, hi = lo + ADAM_CHUNK < s.n ? lo + ADAM_CHUNK : s.n;
    auto upd = [&](float& pp, float gg, float& mm, float& vv) {
        mm = mm + (gg - mm) * w1;                         // torch.lerp(exp_avg, grad, 1 - beta1)
        vv = b2f * vv + w2 * gg * gg;
        const float denom = sqrtf(vv) / bc2_sqrt + a.eps;
        pp = pp - step_size * mm / denom;
    };
    const bool vec = (((size_t)p | (size_t)g | (size_t)m | (size_t)v) & 15) == 0 && (lo & 3) == 0;
    if (vec) {
        // 4 float4 per thread and tensor, every load issued before the first use; a group past the end re-reads the chunk's
        // first group (valid memory) and is not stored
        float4 P[4], G[4], M[4], V[4];
        long long idx[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long i = lo + (threadIdx.x + 256 * u) * 4;
            ok[u] = i + 4 <= hi;
            idx[u] = ok[u] ? i : lo;
        }
        const bool any_vec = lo + 4 <= hi;              // (a chunk of fewer than 4 elements has no group to fall back on)
        if (any_vec) {
#pragma unroll
            for (int u = 0; u < 4; ++u) P[u] = *reinterpret_cast<const float4*>(p + idx[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) G[u] = *reinterpret_cast<const float4*>(g + idx[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) M[u] = *reinterpret_cast<const float4*>(m + idx[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) V[u] = *reinterpret_cast<const float4*>(v + idx[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                upd(P[u].x, G[u].x, M[u].x, V[u].x); upd(P[u].y, G[u].y, M[u].y, V[u].y);
                upd(P[u].z, G[u].z, M[u].z, V[u].z); upd(P[u].w, G[u].w, M[u].w, V[u].w);
                if (ok[u]) {
                    *reinterpret_cast<float4*>(p + idx[u]) = P[u];
                    *reinterpret_cast<float4*>(m + idx[u]) = M[u];
                    *reinterpret_cast<float4*>(v + idx[u]) = V[u];
                }
            }
        }
        // the chunk's last 1-3 elements (only the tensor's final chunk can have them)
        const long long tail0 = lo + ((hi - lo) & ~3LL);
        const long long i = tail0 + threadIdx.x;
        if (i < hi) { float pp = p[i], mm = m[i], vv = v[i]; upd(pp, g[i], mm, vv); p[i] = pp; m[i] = mm; v[i] = vv; }
    } else {
        for (long long i = lo + threadIdx.x; i < hi; i += 256) {
            float pp = p[i], mm = m[i], vv = v[i];
            upd(pp, g[i], mm, vv);
            p[i] = pp; m[i] = mm; v[i] = vv;
        }
    }
}

}  // namespace dc

using namespace dc;

extern "C" int dc_adam_chunk(void) { return ADAM_CHUNK; }

extern "C" int dc_adam_step(const void* slots_dev, const void* chunks_dev, const int* chunk_start_host, const void* const* grads_host,
                            int ntensors, float lr, double beta1, double beta2, float eps, void* stream) {
    if (!slots_dev || !chunks_dev || !chunk_start_host || !grads_host || ntensors <= 0) return DC_EINVAL;
    if (!(lr >= 0.f) || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0) || !(eps >= 0.f)) return DC_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    for (int t0 = 0; t0 < ntensors; t0 += ADAM_MAXT) {
        AdamArgs a{};
        a.slots = (const AdamSlot*)slots_dev; a.chunks = (const int2*)chunks_dev;
        a.t0 = t0; a.nt = std::min(ADAM_MAXT, ntensors - t0);
        a.lr = lr; a.b1 = beta1; a.b2 = beta2; a.eps = eps;
        bool any = false;
        for (int i = 0; i < a.nt; ++i) { a.g[i] = (const float*)grads_host[t0 + i]; any = any || a.g[i]; }
        if (!any) continue;
        const int c0 = chunk_start_host[t0], c1 = chunk_start_host[t0 + a.nt];
        if (c1 < c0) return DC_EINVAL;
        hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(64), 0, st, a);
        DC_CHECK_LAUNCH();
        if (c1 > c0) {
            hipLaunchKernelGGL(adam_apply_kernel, dim3(c1 - c0), dim3(256), 0, st, a, c0);
            DC_CHECK_LAUNCH();
        }
    }
    return DC_OK;
}
