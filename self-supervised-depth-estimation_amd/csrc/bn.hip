// Training-mode BatchNorm2d fused with the residual add and the ReLU of the ResNet trunks
// (reference networks/resnet_encoder.py:87-98 via torchvision BasicBlock / Bottleneck:
//   out = relu(bn(x));   out = relu(bn(x) + identity)).
// All passes are HBM-bound streaming passes over NCHW fp32 tensors with float4 accesses:
//   forward  : (1) per-channel sum / sum-of-squares partials, (2) finalize mean / invstd (+ running
//              stats, momentum 0.1, unbiased variance like torch), (3) y = relu?(x_hat*gamma + beta [+ res]);
//   backward : (1) per-channel partials of sum(g), sum(g*x_hat) with g = gy * [y > 0] (ReLU mask taken from
//              the saved output), (2) dx = gamma*invstd*(g - mean(g) - x_hat*mean(g*x_hat)), dres = g.
// Deterministic: fixed-order partial reductions, no atomics.
#include "dc_common.h"

namespace dc {

constexpr int BN_CHUNK = 4096;     // elements of one (n, c) plane slice handled by one block in the stats passes

__device__ __forceinline__ float block_sum256(float v, float* sm) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

// grid (chunks_per_plane, C, N), block 256: partial[(c*N + n)*chunks + chunk] = {sum, sumsq}
// Small planes (layer3 / layer4: 480 / 120 elements): a block per (channel, sample) spends its life on the prologue -- 12 K
// blocks of 120 elements ran 9-16 us for 6-18 MB.  `ns` > 1 gives a block `ns` consecutive samples of its channel (same
// BatchNorm group; one plane per wave and pass, so the ReLU mask's per-wave ballot layout is unchanged) and one partial.
// The float4 items of one thread in its block's span, visited BN_U at a time so that every load of a batch is issued before
// the first use (one load -> use -> store per loop iteration left a single 16-byte load in flight per thread: the passes ran
// at 3.1-3.7 TB/s where a plain elementwise pass over tensors of this size reaches 6-7, tools/ubench/ew_bw.py).
//   ns == 1: the 256 threads stride the chunk [lo, hi) by 1024 elements (item k -> i = lo + 4 tid + 1024 k);
//   ns  > 1: wave w takes samples w, w + 4, ...; its lanes stride a plane by 256 elements (item k -> sample w + 4 (k / ipp),
//            i = 4 lane + 256 (k % ipp)).  Either way a wave's 64 lanes cover 256 consecutive elements of one plane, which
//            is the layout of the ReLU bit mask (4 ballots per wave item).
constexpr int BN_U = 4;
struct BnItems {
    int wave, lane, ipp, nitems, lo, hi, ns, HW;
    __device__ __forceinline__ BnItems(int ns_, int HW_, int lo_, int hi_) : lo(lo_), hi(hi_), ns(ns_), HW(HW_) {
        wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); lane = threadIdx.x & 63;
        if (ns > 1) { ipp = (HW + 255) >> 8; nitems = ((ns - wave + 3) >> 2) * ipp; }
        else { ipp = 1; nitems = (hi - lo + 1023) >> 10; }
    }
    // item k -> sample offset nn (wave-uniform) and element offset i; false past the end (nn, i then point at a valid item)
    __device__ __forceinline__ bool at(int k, int& nn, int& i) const {
        bool ok;
        if (ns > 1) {
            const int a = k / ipp, b = k - a * ipp;
            nn = wave + 4 * a; i = lane * 4 + 256 * b;
            ok = k < nitems && i < HW;
            if (!(k < nitems)) nn = wave < ns ? wave : 0;
        } else {
            nn = 0; i = lo + (int)threadIdx.x * 4 + 1024 * k;
            ok = i < hi;
        }
        if (!ok) i = 0;
        return ok;
    }
};

// both block sums behind one pair of barriers
__device__ __forceinline__ void block_sum256x2(float& a, float& b, float* sm8) {
    a = wave_sum(a); b = wave_sum(b);
    if ((threadIdx.x & 63) == 0) { sm8[threadIdx.x >> 6] = a; sm8[4 + (threadIdx.x >> 6)] = b; }
    __syncthreads();
    a = (sm8[0] + sm8[1]) + (sm8[2] + sm8[3]);
    b = (sm8[4] + sm8[5]) + (sm8[6] + sm8[7]);
}

__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, float* __restrict__ part, int C, int HW, int ns) {
    __shared__ float sm[8];
    const int c = blockIdx.y, n0 = blockIdx.z * ns;
    const int lo = blockIdx.x * BN_CHUNK, hi = min(lo + BN_CHUNK, HW);
    float s = 0.f, q = 0.f;
    if ((HW & 3) == 0) {
        const BnItems it(ns, HW, lo, hi);
        for (int k0 = 0; k0 < it.nitems; k0 += BN_U) {
            float4 v[BN_U];
            bool ok[BN_U];
#pragma unroll
            for (int u = 0; u < BN_U; ++u) {
                int nn, i;
                ok[u] = it.at(k0 + u, nn, i);
                v[u] = *reinterpret_cast<const float4*>(x + ((size_t)(n0 + nn) * C + c) * HW + i);
            }
#pragma unroll
            for (int u = 0; u < BN_U; ++u) {
                const float4 w = ok[u] ? v[u] : make_float4(0.f, 0.f, 0.f, 0.f);
                s += (w.x + w.y) + (w.z + w.w);
                q += (w.x * w.x + w.y * w.y) + (w.z * w.z + w.w * w.w);
            }
        }
    } else {
        const float* p = x + ((size_t)n0 * C + c) * HW;
        for (int i = lo + threadIdx.x; i < hi; i += 256) { const float v = p[i]; s += v; q += v * v; }
    }
    block_sum256x2(s, q, sm);
    if (threadIdx.x == 0) {
        float* o = part + (((size_t)c * gridDim.z + blockIdx.z) * gridDim.x + blockIdx.x) * 2;
        o[0] = s; o[1] = q;
    }
}

// Per-(group, channel) statistics from the partial sums, recomputed by every block that needs them (a few dozen
// L2-resident floats; cheaper than a separate finalize launch per layer).  Fixed summation order -> every block
// gets bit-identical values.
__device__ __forceinline__ void bn_reduce_partials(const float* __restrict__ part, int c, int nparts, int per, int gidx, float& s, float& q) {
    // every WAVE reduces the partials by itself (lane-strided sums, then the wave tree): no barrier in front of the streaming
    // loop, and the four waves -- like every block -- get bit-identical values
    const float2* p2 = reinterpret_cast<const float2*>(part) + (size_t)c * nparts + (size_t)gidx * per;
    float a = 0.f, b = 0.f;
    for (int i = threadIdx.x & 63; i < per; i += 64) { const float2 v = p2[i]; a += v.x; b += v.y; }
    s = wave_sum(a);
    q = wave_sum(b);
}

// grid (chunks, C, N), block 256: y = relu?((x - mean)*invstd*gamma + beta [+ res]); block (0, c, first sample of
// a group) also publishes mean / invstd for the backward, block (0, c, 0) the running statistics.
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* x, const float* res, const float* part,
                                                       float* mean, float* invstd, float* run_mean, float* run_var,
                                                       const float* gamma, const float* beta, float* y, int C, int HW,
                                                       int relu, int n_per_group, int groups, float eps, float momentum,
                                                       unsigned long long* mask, int ns) {
    const int c = blockIdx.y, n0 = blockIdx.z * ns, n = n0;
    const int nparts = gridDim.z * gridDim.x, per = nparts / groups;
    const int gidx = n / n_per_group;
    const float count = (float)n_per_group * (float)HW;
    float s, q;
    bn_reduce_partials(part, c, nparts, per, gidx, s, q);
    const float m = s / count;
    const float var = fmaxf(q / count - m * m, 0.f);          // biased (used to normalise)
    const float is = rsqrtf(var + eps);
    if (blockIdx.x == 0 && threadIdx.x == 0 && n == gidx * n_per_group) { mean[gidx * C + c] = m; invstd[gidx * C + c] = is; }
    if (blockIdx.x == 0 && n == 0 && run_mean) {
        // sequential running-stat updates, one per group, as separate module calls would do
        float rm = run_mean[c], rv = run_var[c];
        for (int g2 = 0; g2 < groups; ++g2) {
            float s2, q2;
            bn_reduce_partials(part, c, nparts, per, g2, s2, q2);
            const float m2 = s2 / count, v2 = fmaxf(q2 / count - m2 * m2, 0.f);
            rm = (1.f - momentum) * rm + momentum * m2;
            rv = (1.f - momentum) * rv + momentum * v2 * (count / fmaxf(count - 1.f, 1.f));
        }
        if (threadIdx.x == 0) { run_mean[c] = rm; run_var[c] = rv; }
    }
    const float a = is * gamma[c], b = beta[c] - m * a;
    const int lo = blockIdx.x * BN_CHUNK, hi = min(lo + BN_CHUNK, HW);
    if ((HW & 3) == 0) {
        const BnItems it(ns, HW, lo, hi);
        const float* rsrc = res ? res : x;                      // one code shape with or without the residual (the loads stay unconditional)
        for (int k0 = 0; k0 < it.nitems; k0 += BN_U) {
            float4 v[BN_U], r[BN_U];
            size_t off[BN_U];
            bool ok[BN_U];
            int iu[BN_U], nu[BN_U];
#pragma unroll
            for (int u = 0; u < BN_U; ++u) {
                ok[u] = it.at(k0 + u, nu[u], iu[u]);
                off[u] = ((size_t)(n0 + nu[u]) * C + c) * HW + iu[u];
                v[u] = *reinterpret_cast<const float4*>(x + off[u]);
            }
#pragma unroll
            for (int u = 0; u < BN_U; ++u) r[u] = *reinterpret_cast<const float4*>(rsrc + off[u]);
#pragma unroll
            for (int u = 0; u < BN_U; ++u) {
                float4 w = v[u];
                w.x = fmaf(w.x, a, b); w.y = fmaf(w.y, a, b); w.z = fmaf(w.z, a, b); w.w = fmaf(w.w, a, b);
                if (res) { w.x += r[u].x; w.y += r[u].y; w.z += r[u].z; w.w += r[u].w; }
                if (relu) {
                    if (mask) {
                        // ReLU mask for the backward as bits: the 256 elements of this wave item -> 4 ballots (one per
                        // float4 component), 32 bytes instead of the 1 KiB of y the backward kernels would re-read twice.
                        // (lanes past the end of the plane contribute 0 bits; a wave whose item is entirely past the end
                        // writes nothing)
                        const unsigned long long b0 = __ballot(ok[u] && w.x > 0.f), b1 = __ballot(ok[u] && w.y > 0.f),
                                                 b2 = __ballot(ok[u] && w.z > 0.f), b3 = __ballot(ok[u] && w.w > 0.f);
                        const bool any = __ballot(ok[u]) != 0ull;
                        if (any && (threadIdx.x & 63) == 0) {
                            // lane 0 is valid whenever any lane of the wave item is (items are lane-ascending)
                            unsigned long long* mw = mask + (((size_t)(n0 + nu[u]) * C + c) * ((HW + 255) >> 8) + (iu[u] >> 8)) * 4;
                            mw[0] = b0; mw[1] = b1; mw[2] = b2; mw[3] = b3;
                        }
                    }
                    w.x = fmaxf(w.x, 0.f); w.y = fmaxf(w.y, 0.f); w.z = fmaxf(w.z, 0.f); w.w = fmaxf(w.w, 0.f);
                }
                if (ok[u]) *reinterpret_cast<float4*>(y + off[u]) = w;
            }
        }
    } else {
        const size_t base = ((size_t)n * C + c) * HW;
        for (int i = lo + threadIdx.x; i < hi; i += 256) {
            float v = fmaf(x[base + i], a, b);
            if (res) v += res[base + i];
            y[base + i] = relu ? fmaxf(v, 0.f) : v;
        }
    }
}

// The ReLU decisions of one float4 as 4 bits (x, y, z, w): from the forward's ballots (this lane's bit of the wave item's 4
// words -- `mw` loaded by the caller together with the batch's other loads) or from y > 0.
__device__ __forceinline__ float4 bn_apply_bits(float4 g, unsigned bits) {
    g.x = (bits & 1u) ? g.x : 0.f; g.y = (bits & 2u) ? g.y : 0.f; g.z = (bits & 4u) ? g.z : 0.f; g.w = (bits & 8u) ? g.w : 0.f;
    return g;
}
struct BnMaskWords { unsigned long long w[4]; };
__device__ __forceinline__ unsigned bn_bits_of(const BnMaskWords& m) {
    const int l = threadIdx.x & 63;
    return (unsigned)((m.w[0] >> l) & 1ull) | ((unsigned)((m.w[1] >> l) & 1ull) << 1) | ((unsigned)((m.w[2] >> l) & 1ull) << 2) |
           ((unsigned)((m.w[3] >> l) & 1ull) << 3);
}
__device__ __forceinline__ unsigned bn_bits_of(float4 yv) {
    return (yv.x > 0.f ? 1u : 0u) | (yv.y > 0.f ? 2u : 0u) | (yv.z > 0.f ? 4u : 0u) | (yv.w > 0.f ? 8u : 0u);
}

// backward stats: partial {sum g, sum g*x_hat}, g = gy*[y>0] when relu.  MODE: 0 no ReLU, 1 bit mask, 2 saved output y
// (a template parameter so that each instantiation has ONE set of unconditional loads per batch)
template <int MODE>
__global__ __launch_bounds__(256) void bn_bwd_stats_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ gy,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd, float* __restrict__ part, int C,
                                                           int HW, int n_per_group, const unsigned long long* __restrict__ mask, int ns) {
    __shared__ float sm[8];
    const int c = blockIdx.y, n0 = blockIdx.z * ns;
    const int gc = (n0 / n_per_group) * C + c;
    const float m = mean[gc], is = invstd[gc];
    const int lo = blockIdx.x * BN_CHUNK, hi = min(lo + BN_CHUNK, HW);
    float s = 0.f, q = 0.f;
    if ((HW & 3) == 0) {
        const BnItems it(ns, HW, lo, hi);
        for (int k0 = 0; k0 < it.nitems; k0 += BN_U) {
            float4 xv[BN_U], g[BN_U], yv[BN_U];
            BnMaskWords mw[BN_U];
            bool ok[BN_U];
#pragma unroll
            for (int u = 0; u < BN_U; ++u) {
                int nn, i;
                ok[u] = it.at(k0 + u, nn, i);
                const size_t plane = (size_t)(n0 + nn) * C + c, off = plane * HW + i;
                xv[u] = *reinterpret_cast<const float4*>(x + off);
                g[u] = *reinterpret_cast<const float4*>(gy + off);
                if (MODE == 1) mw[u] = *reinterpret_cast<const BnMaskWords*>(mask + (plane * ((HW + 255) >> 8) + (i >> 8)) * 4);
                if (MODE == 2) yv[u] = *reinterpret_cast<const float4*>(y + off);
            }
#pragma unroll
            for (int u = 0; u < BN_U; ++u) {
                float4 gg = g[u];
                if (MODE == 1) gg = bn_apply_bits(gg, bn_bits_of(mw[u]));
                if (MODE == 2) gg = bn_apply_bits(gg, bn_bits_of(yv[u]));
                if (!ok[u]) gg = make_float4(0.f, 0.f, 0.f, 0.f);
                s += (gg.x + gg.y) + (gg.z + gg.w);
                q += (gg.x * (xv[u].x - m) + gg.y * (xv[u].y - m)) + (gg.z * (xv[u].z - m) + gg.w * (xv[u].w - m));
            }
        }
    } else {
        const size_t base = ((size_t)n0 * C + c) * HW;
        for (int i = lo + threadIdx.x; i < hi; i += 256) {
            float g = gy[base + i];
            if (MODE != 0 && !(y[base + i] > 0.f)) g = 0.f;
            s += g; q += g * (x[base + i] - m);
        }
    }
    block_sum256x2(s, q, sm);
    q *= is;
    if (threadIdx.x == 0) {
        float* o = part + (((size_t)c * gridDim.z + blockIdx.z) * gridDim.x + blockIdx.x) * 2;
        o[0] = s; o[1] = q;
    }
}

// dx = gamma*invstd*(g - mean(g) - x_hat*mean(g*x_hat)); dres = g.  The two means come from the partials (reduced
// by every wave, see bn_reduce_partials); block (0, c, 0) also writes dgamma / dbeta (sums over all groups).
template <int MODE>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ gy,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ part, float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dx,
                                                           float* __restrict__ dres, int C, int HW, int n_per_group,
                                                           int groups, const unsigned long long* __restrict__ mask, int ns) {
    const int c = blockIdx.y, n0 = blockIdx.z * ns, n = n0;
    const int nparts = gridDim.z * gridDim.x, per = nparts / groups;
    const int gidx = n / n_per_group;
    const float count = (float)n_per_group * (float)HW;
    float s, q;
    bn_reduce_partials(part, c, nparts, per, gidx, s, q);
    if (blockIdx.x == 0 && n == 0) {
        float ts = 0.f, tq = 0.f;
        for (int g2 = 0; g2 < groups; ++g2) {
            float s2, q2;
            bn_reduce_partials(part, c, nparts, per, g2, s2, q2);
            ts += s2; tq += q2;
        }
        if (threadIdx.x == 0) {
            if (dbeta) dbeta[c] = ts;
            if (dgamma) dgamma[c] = tq;
        }
    }
    const int gc = gidx * C + c;
    const float m = mean[gc], is = invstd[gc], k = gamma[c] * is, a = s / count, bq = (q / count) * is;
    const int lo = blockIdx.x * BN_CHUNK, hi = min(lo + BN_CHUNK, HW);
    if ((HW & 3) == 0) {
        const BnItems it(ns, HW, lo, hi);
        for (int k0 = 0; k0 < it.nitems; k0 += BN_U) {
            float4 xv[BN_U], g[BN_U], yv[BN_U];
            BnMaskWords mw[BN_U];
            size_t off[BN_U];
            bool ok[BN_U];
#pragma unroll
            for (int u = 0; u < BN_U; ++u) {
                int nn, i;
                ok[u] = it.at(k0 + u, nn, i);
                const size_t plane = (size_t)(n0 + nn) * C + c;
                off[u] = plane * HW + i;
                xv[u] = *reinterpret_cast<const float4*>(x + off[u]);
                g[u] = *reinterpret_cast<const float4*>(gy + off[u]);
                if (MODE == 1) mw[u] = *reinterpret_cast<const BnMaskWords*>(mask + (plane * ((HW + 255) >> 8) + (i >> 8)) * 4);
                if (MODE == 2) yv[u] = *reinterpret_cast<const float4*>(y + off[u]);
            }
#pragma unroll
            for (int u = 0; u < BN_U; ++u) {
                float4 gg = g[u];
                if (MODE == 1) gg = bn_apply_bits(gg, bn_bits_of(mw[u]));
                if (MODE == 2) gg = bn_apply_bits(gg, bn_bits_of(yv[u]));
                float4 d;
                d.x = k * (gg.x - a - (xv[u].x - m) * bq); d.y = k * (gg.y - a - (xv[u].y - m) * bq);
                d.z = k * (gg.z - a - (xv[u].z - m) * bq); d.w = k * (gg.w - a - (xv[u].w - m) * bq);
                if (ok[u]) {
                    if (dres) *reinterpret_cast<float4*>(dres + off[u]) = gg;
                    *reinterpret_cast<float4*>(dx + off[u]) = d;
                }
            }
        }
    } else {
        const size_t base = ((size_t)n * C + c) * HW;
        for (int i = lo + threadIdx.x; i < hi; i += 256) {
            float g = gy[base + i];
            if (MODE != 0 && !(y[base + i] > 0.f)) g = 0.f;
            if (dres) dres[base + i] = g;
            dx[base + i] = k * (g - a - (x[base + i] - m) * bq);
        }
    }
}

}  // namespace dc

using namespace dc;
#define ST ((hipStream_t)stream)

// samples per block: the largest divisor of the group size whose planes still fit one chunk (1 for planes of half a chunk
// or more, and on the scalar path)
static int bn_ns(int n_per_group, int HW) {
    if ((HW & 3) || HW * 2 > BN_CHUNK) return 1;
    int best = 1;
    for (int d = 2; d <= n_per_group; ++d)
        if (n_per_group % d == 0 && d * HW <= BN_CHUNK) best = d;
    return best;
}

extern "C" size_t dc_bn_workspace(int N, int C, int HW) {
    if (N <= 0 || C <= 0 || HW <= 0) return 0;
    return ((size_t)C * N * ceil_div(HW, BN_CHUNK) * 2 + 2 * (size_t)C * N) * sizeof(float);   // partials + per-group means
}

extern "C" size_t dc_bn_mask_bytes(int N, int C, int HW) {
    if (N <= 0 || C <= 0 || HW <= 0 || (HW & 3)) return 0;            // the bit mask exists for the float4 path only
    return (size_t)N * C * ((HW + 255) >> 8) * 4 * sizeof(unsigned long long);
}

extern "C" int dc_bn_relu_fwd(const float* x, const float* res, const float* gamma, const float* beta, float* y,
                              float* save_mean, float* save_invstd, float* running_mean, float* running_var, void* ws,
                              void* relu_mask, int N, int C, int HW, float eps, float momentum, int relu, int groups,
                              void* stream) {
    if (!x || !gamma || !beta || !y || !save_mean || !save_invstd || !ws || N <= 0 || C <= 0 || HW <= 0) return DC_EINVAL;
    if (groups < 1 || N % groups) return DC_EINVAL;
    const int chunks = ceil_div(HW, BN_CHUNK), ns = bn_ns(N / groups, HW);
    float* part = (float*)ws;
    hipLaunchKernelGGL(bn_stats_kernel, dim3(chunks, C, N / ns), dim3(256), 0, ST, x, part, C, HW, ns);
    DC_CHECK_LAUNCH();
    hipLaunchKernelGGL(bn_apply_kernel, dim3(chunks, C, N / ns), dim3(256), 0, ST, x, res, (const float*)part, save_mean,
                       save_invstd, running_mean, running_var, gamma, beta, y, C, HW, relu, N / groups, groups, eps, momentum,
                       (HW & 3) ? (unsigned long long*)nullptr : (unsigned long long*)relu_mask, ns);
    DC_CHECK_LAUNCH();
    return DC_OK;
}

extern "C" int dc_bn_relu_bwd(const float* x, const float* y, const float* gy, const float* gamma,
                              const float* save_mean, const float* save_invstd, float* dx, float* dres, float* dgamma,
                              float* dbeta, void* ws, const void* relu_mask, int N, int C, int HW, int relu, int groups,
                              void* stream) {
    if (!x || !gy || !gamma || !save_mean || !save_invstd || !dx || !ws || N <= 0 || C <= 0 || HW <= 0) return DC_EINVAL;
    const unsigned long long* mk = (HW & 3) ? nullptr : (const unsigned long long*)relu_mask;
    if (relu && !y && !mk) return DC_EINVAL;
    if (groups < 1 || N % groups) return DC_EINVAL;
    const int chunks = ceil_div(HW, BN_CHUNK), ns = bn_ns(N / groups, HW);
    float* part = (float*)ws;
    const dim3 grid(chunks, C, N / ns);
    const int mode = !relu ? 0 : (mk ? 1 : 2);
#define BN_BWD(MODE)                                                                                                             \
    do {                                                                                                                         \
        hipLaunchKernelGGL(bn_bwd_stats_kernel<MODE>, grid, dim3(256), 0, ST, x, y, gy, save_mean, save_invstd, part, C, HW,      \
                           N / groups, mk, ns);                                                                                  \
        DC_CHECK_LAUNCH();                                                                                                       \
        hipLaunchKernelGGL(bn_bwd_apply_kernel<MODE>, grid, dim3(256), 0, ST, x, y, gy, save_mean, save_invstd, gamma,            \
                           (const float*)part, dgamma, dbeta, dx, dres, C, HW, N / groups, groups, mk, ns);                      \
        DC_CHECK_LAUNCH();                                                                                                       \
    } while (0)
    if (mode == 0) BN_BWD(0);
    else if (mode == 1) BN_BWD(1);
    else BN_BWD(2);
#undef BN_BWD
    return DC_OK;
}

// ------------------------------------------------------------------------------------------------
// BatchNorm folded into the neighbouring convolutions (round 5).  The statistics pass is gone -- the PRODUCING
// convolution's store epilogue emits per-channel partial {sum, sum of squares} of its output -- and, where a BatchNorm + ReLU
// has ONE consumer, so is the apply pass: the consuming convolution reads relu(scale * x + shift) in its loader.  What is
// left of the layer in HBM terms is a finalize launch over the partials (a few KB per channel) and, for the block outputs
// (BatchNorm + residual + ReLU, several consumers), one apply pass.  Backward: the consuming convolution's data-gradient
// epilogue masks its result with the ReLU decision and emits partial {sum g, sum g (x - mean)}; a finalize turns them into
// per-channel coefficients and ONE pass forms dx = a g + b (x - mean) + c.
//
// Partials: float2 part[c][p], p < nparts; partial p covers pixels of ONE BatchNorm group, group(p) = min(p / ppg, groups-1).
// Fixed summation order everywhere (thread-strided sums + the block tree): deterministic, every rank the same.
// ------------------------------------------------------------------------------------------------
namespace dc {

__device__ __forceinline__ void block_sum256x2_all(float& a, float& b, float* sm8) {
    __syncthreads();
    block_sum256x2(a, b, sm8);
}

// grid (C), block 256.  mean / invstd / scale / shift: (groups, C).  Running statistics updated group after group, as
// separate module calls would.
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float2* __restrict__ part, int nparts, int ppg, float count,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* __restrict__ run_mean, float* __restrict__ run_var,
                                                          float* __restrict__ mean, float* __restrict__ invstd,
                                                          float* __restrict__ scale, float* __restrict__ shift, int C, int groups,
                                                          float eps, float momentum) {
    __shared__ float sm[8];
    const int c = blockIdx.x;
    const float2* p = part + (size_t)c * nparts;
    float rm = run_mean ? run_mean[c] : 0.f, rv = run_var ? run_var[c] : 0.f;
    const float ga = gamma[c], be = beta[c];
    for (int g = 0; g < groups; ++g) {
        const int lo = g * ppg, hi = g == groups - 1 ? nparts : (g + 1) * ppg;
        float s = 0.f, q = 0.f;
        for (int i = lo + threadIdx.x; i < hi; i += 256) { const float2 v = p[i]; s += v.x; q += v.y; }
        block_sum256x2_all(s, q, sm);
        const float m = s / count;
        const float var = fmaxf(q / count - m * m, 0.f);
        const float is = rsqrtf(var + eps);
        if (threadIdx.x == 0) {
            mean[g * C + c] = m; invstd[g * C + c] = is;
            const float a = is * ga;
            scale[g * C + c] = a; shift[g * C + c] = be - m * a;
        }
        rm = (1.f - momentum) * rm + momentum * m;
        rv = (1.f - momentum) * rv + momentum * var * (count / fmaxf(count - 1.f, 1.f));
    }
    if (threadIdx.x == 0 && run_mean) { run_mean[c] = rm; run_var[c] = rv; }
}

// y = relu?(scale * x + shift [+ res]) from finished per-(group, channel) scale / shift; same item walk, same ReLU bit mask
// as bn_apply_kernel.  grid (chunks, C, N / ns).
__global__ __launch_bounds__(256) void bn_apply2_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                                        float* __restrict__ y, int C, int HW, int relu, int n_per_group,
                                                        unsigned long long* __restrict__ mask, int ns) {
    const int c = blockIdx.y, n0 = blockIdx.z * ns;
    const int gc = (n0 / n_per_group) * C + c;
    const float a = scale[gc], b = shift[gc];
    const int lo = blockIdx.x * BN_CHUNK, hi = min(lo + BN_CHUNK, HW);
    const BnItems it(ns, HW, lo, hi);
    const float* rsrc = res ? res : x;
    for (int k0 = 0; k0 < it.nitems; k0 += BN_U) {
        float4 v[BN_U], r[BN_U];
        size_t off[BN_U];
        bool ok[BN_U];
        int iu[BN_U], nu[BN_U];
#pragma unroll
        for (int u = 0; u < BN_U; ++u) {
            ok[u] = it.at(k0 + u, nu[u], iu[u]);
            off[u] = ((size_t)(n0 + nu[u]) * C + c) * HW + iu[u];
            v[u] = *reinterpret_cast<const float4*>(x + off[u]);
        }
#pragma unroll
        for (int u = 0; u < BN_U; ++u) r[u] = *reinterpret_cast<const float4*>(rsrc + off[u]);
#pragma unroll
        for (int u = 0; u < BN_U; ++u) {
            float4 w = v[u];
            w.x = fmaf(w.x, a, b); w.y = fmaf(w.y, a, b); w.z = fmaf(w.z, a, b); w.w = fmaf(w.w, a, b);
            if (res) { w.x += r[u].x; w.y += r[u].y; w.z += r[u].z; w.w += r[u].w; }
            if (relu) {
                if (mask) {
                    const unsigned long long b0 = __ballot(ok[u] && w.x > 0.f), b1 = __ballot(ok[u] && w.y > 0.f),
                                             b2 = __ballot(ok[u] && w.z > 0.f), b3 = __ballot(ok[u] && w.w > 0.f);
                    const bool any = __ballot(ok[u]) != 0ull;
                    if (any && (threadIdx.x & 63) == 0) {
                        unsigned long long* mw = mask + (((size_t)(n0 + nu[u]) * C + c) * ((HW + 255) >> 8) + (iu[u] >> 8)) * 4;
                        mw[0] = b0; mw[1] = b1; mw[2] = b2; mw[3] = b3;
                    }
                }
                w.x = fmaxf(w.x, 0.f); w.y = fmaxf(w.y, 0.f); w.z = fmaxf(w.z, 0.f); w.w = fmaxf(w.w, 0.f);
            }
            if (ok[u]) *reinterpret_cast<float4*>(y + off[u]) = w;
        }
    }
}

// Backward partials {sum g', sum g' (x - mean)} -> coef[(g, c)] = {a, b, c0, mean} with dx = a g' + b (x - mean) + c0,
// a = gamma invstd, b = -a invstd^2 mean(g' (x - mean)), c0 = -a mean(g'); dgamma = sum_g invstd sum g' (x - mean), dbeta = sum g'.
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float2* __restrict__ part, int nparts, int ppg, float count,
                                                              const float* __restrict__ gamma, const float* __restrict__ mean,
                                                              const float* __restrict__ invstd, float4* __restrict__ coef,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta, int C, int groups) {
    __shared__ float sm[8];
    const int c = blockIdx.x;
    const float2* p = part + (size_t)c * nparts;
    const float ga = gamma[c];
    float tg = 0.f, tb = 0.f;
    for (int g = 0; g < groups; ++g) {
        const int lo = g * ppg, hi = g == groups - 1 ? nparts : (g + 1) * ppg;
        float s = 0.f, q = 0.f;
        for (int i = lo + threadIdx.x; i < hi; i += 256) { const float2 v = p[i]; s += v.x; q += v.y; }
        block_sum256x2_all(s, q, sm);
        const float m = mean[g * C + c], is = invstd[g * C + c];
        q *= is;                                             // sum g' x_hat
        const float k = ga * is;
        if (threadIdx.x == 0) coef[g * C + c] = make_float4(k, -k * is * (q / count), -k * (s / count), m);
        tg += q; tb += s;
    }
    if (threadIdx.x == 0) {
        if (dgamma) dgamma[c] = tg;
        if (dbeta) dbeta[c] = tb;
    }
}

// dx = a g' + b (x - mean) + c0.  g' is the ALREADY MASKED upstream gradient (the producing data-gradient kernel's epilogue
// applied the ReLU decision), which is also the gradient of the residual input: no dres store.  grid (chunks, C, N / ns).
__global__ __launch_bounds__(256) void bn_bwd_apply2_kernel(const float* __restrict__ x, const float* __restrict__ gp,
                                                            const float4* __restrict__ coef, float* __restrict__ dx, int C, int HW,
                                                            int n_per_group, int ns) {
    const int c = blockIdx.y, n0 = blockIdx.z * ns;
    const float4 k = coef[(n0 / n_per_group) * C + c];
    const int lo = blockIdx.x * BN_CHUNK, hi = min(lo + BN_CHUNK, HW);
    const BnItems it(ns, HW, lo, hi);
    for (int k0 = 0; k0 < it.nitems; k0 += BN_U) {
        float4 xv[BN_U], g[BN_U];
        size_t off[BN_U];
        bool ok[BN_U];
#pragma unroll
        for (int u = 0; u < BN_U; ++u) {
            int nn, i;
            ok[u] = it.at(k0 + u, nn, i);
            off[u] = ((size_t)(n0 + nn) * C + c) * HW + i;
            xv[u] = *reinterpret_cast<const float4*>(x + off[u]);
            g[u] = *reinterpret_cast<const float4*>(gp + off[u]);
        }
#pragma unroll
        for (int u = 0; u < BN_U; ++u) {
            float4 d;
            d.x = fmaf(k.x, g[u].x, fmaf(k.y, xv[u].x - k.w, k.z)); d.y = fmaf(k.x, g[u].y, fmaf(k.y, xv[u].y - k.w, k.z));
            d.z = fmaf(k.x, g[u].z, fmaf(k.y, xv[u].z - k.w, k.z)); d.w = fmaf(k.x, g[u].w, fmaf(k.y, xv[u].w - k.w, k.z));
            if (ok[u]) *reinterpret_cast<float4*>(dx + off[u]) = d;
        }
    }
}

}  // namespace dc

// stand-alone statistics pass in the partial layout (for producers without a statistics epilogue): the partials of
// bn_stats_kernel, (N / ns) * chunks per channel, equal shares per group
extern "C" int dc_bn_stat_parts(int N, int C, int HW, int groups, int* ppg) {
    if (N <= 0 || C <= 0 || HW <= 0 || groups < 1 || N % groups || (HW & 3)) return 0;
    const int chunks = ceil_div(HW, BN_CHUNK), ns = bn_ns(N / groups, HW);
    if (ppg) *ppg = (N / groups / ns) * chunks;
    return (N / ns) * chunks;
}

extern "C" int dc_bn_stats(const float* x, float* part, int N, int C, int HW, int groups, void* stream) {
    if (!x || !part || !dc_bn_stat_parts(N, C, HW, groups, nullptr)) return DC_EINVAL;
    const int chunks = ceil_div(HW, BN_CHUNK), ns = bn_ns(N / groups, HW);
    hipLaunchKernelGGL(bn_stats_kernel, dim3(chunks, C, N / ns), dim3(256), 0, ST, x, part, C, HW, ns);
    DC_CHECK_LAUNCH();
    return DC_OK;
}

extern "C" int dc_bn_finalize(const float* part, int nparts, int ppg, double count, const float* gamma, const float* beta,
                              float* running_mean, float* running_var, float* mean, float* invstd, float* scale, float* shift,
                              int C, int groups, float eps, float momentum, void* stream) {
    if (!part || !gamma || !beta || !mean || !invstd || !scale || !shift || C <= 0 || groups < 1 || nparts < groups || ppg < 1 ||
        (groups - 1) * ppg >= nparts || count < 1.0)
        return DC_EINVAL;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(256), 0, ST, (const float2*)part, nparts, ppg, (float)count, gamma, beta,
                       running_mean, running_var, mean, invstd, scale, shift, C, groups, eps, momentum);
    DC_CHECK_LAUNCH();
    return DC_OK;
}

extern "C" int dc_bn_apply(const float* x, const float* res, const float* scale, const float* shift, float* y, void* relu_mask,
                           int N, int C, int HW, int relu, int groups, void* stream) {
    if (!x || !scale || !shift || !y || N <= 0 || C <= 0 || HW <= 0 || (HW & 3) || groups < 1 || N % groups) return DC_EINVAL;
    const int chunks = ceil_div(HW, BN_CHUNK), ns = bn_ns(N / groups, HW);
    hipLaunchKernelGGL(bn_apply2_kernel, dim3(chunks, C, N / ns), dim3(256), 0, ST, x, res, scale, shift, y, C, HW, relu, N / groups,
                       relu ? (unsigned long long*)relu_mask : (unsigned long long*)nullptr, ns);
    DC_CHECK_LAUNCH();
    return DC_OK;
}

extern "C" int dc_bn_bwd_finalize(const float* part, int nparts, int ppg, double count, const float* gamma, const float* mean,
                                  const float* invstd, float* coef, float* dgamma, float* dbeta, int C, int groups, void* stream) {
    if (!part || !gamma || !mean || !invstd || !coef || C <= 0 || groups < 1 || nparts < groups || ppg < 1 ||
        (groups - 1) * ppg >= nparts || count < 1.0)
        return DC_EINVAL;
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, ST, (const float2*)part, nparts, ppg, (float)count, gamma, mean,
                       invstd, (float4*)coef, dgamma, dbeta, C, groups);
    DC_CHECK_LAUNCH();
    return DC_OK;
}

extern "C" int dc_bn_bwd_apply(const float* x, const float* gp, const float* coef, float* dx, int N, int C, int HW, int groups,
                               void* stream) {
    if (!x || !gp || !coef || !dx || N <= 0 || C <= 0 || HW <= 0 || (HW & 3) || groups < 1 || N % groups) return DC_EINVAL;
    const int chunks = ceil_div(HW, BN_CHUNK), ns = bn_ns(N / groups, HW);
    hipLaunchKernelGGL(bn_bwd_apply2_kernel, dim3(chunks, C, N / ns), dim3(256), 0, ST, x, gp, (const float4*)coef, dx, C, HW,
                       N / groups, ns);
    DC_CHECK_LAUNCH();
    return DC_OK;
}

// ------------------------------------------------------------------------------------------------
// MaxPool2d(kernel 3, stride 2, padding 1) of the ResNet stem (torchvision ResNet.maxpool, reached from
// networks/resnet_encoder.py:93).  Forward stores the window position of the maximum (first maximum in
// row-major scan order, like ATen) as one byte per output; backward is a gather over the <= 4 windows that
// cover an input pixel -- no atomics.
// ------------------------------------------------------------------------------------------------
namespace dc {

__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* x, float* y, uint8_t* code, int H, int W, int Ho,
                                                          int Wo) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const size_t plane = blockIdx.y;
    if (i >= Ho * Wo) return;
    const int oy = i / Wo, ox = i - oy * Wo;
    const float* p = x + plane * H * W;
    float best = -INFINITY;
    int bc = 0;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int yy = oy * 2 - 1 + ky;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int xx = ox * 2 - 1 + kx;
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
                const float v = p[(size_t)yy * W + xx];
                if (v > best || v != v) { best = v; bc = ky * 3 + kx; }
            }
        }
    }
    y[plane * Ho * Wo + i] = best;
    code[plane * Ho * Wo + i] = (uint8_t)bc;
}

// two adjacent outputs per thread (ox0 even, W % 4 == 0): a window row is one aligned 16-byte load + one dword instead of
// six stride-2 dword loads; same scan order and tie-break as the scalar kernel
__global__ __launch_bounds__(256) void maxpool_fwd2_kernel(const float* __restrict__ x, float* __restrict__ y, uint8_t* __restrict__ code,
                                                           int H, int W, int Ho, int Wo) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const size_t plane = blockIdx.y;
    const int Wo2 = Wo >> 1;
    if (i >= Ho * Wo2) return;
    const int oy = i / Wo2, ox0 = (i - oy * Wo2) * 2;
    const float* p = x + plane * H * W;
    float ba = -INFINITY, bb = -INFINITY;
    int ca = 0, cb = 0;
    // the three window rows are fetched together from clamped rows (a `continue` / `if` around the loads made each row wait
    // for the previous one) and rows / the left column outside the image are skipped in the comparisons
    float4 f[3];
    float e[3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int yy = min(max(oy * 2 - 1 + ky, 0), H - 1);
        const float* row = p + (size_t)yy * W + 2 * ox0;
        f[ky] = *reinterpret_cast<const float4*>(row);
        e[ky] = row[ox0 > 0 ? -1 : 0];
    }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int yy = oy * 2 - 1 + ky;
        if (yy < 0 || yy >= H) continue;
        if (ox0 > 0) { if (e[ky] > ba || e[ky] != e[ky]) { ba = e[ky]; ca = ky * 3; } }
        if (f[ky].x > ba || f[ky].x != f[ky].x) { ba = f[ky].x; ca = ky * 3 + 1; }
        if (f[ky].y > ba || f[ky].y != f[ky].y) { ba = f[ky].y; ca = ky * 3 + 2; }
        if (f[ky].y > bb || f[ky].y != f[ky].y) { bb = f[ky].y; cb = ky * 3; }
        if (f[ky].z > bb || f[ky].z != f[ky].z) { bb = f[ky].z; cb = ky * 3 + 1; }
        if (f[ky].w > bb || f[ky].w != f[ky].w) { bb = f[ky].w; cb = ky * 3 + 2; }
    }
    const size_t o = plane * Ho * Wo + (size_t)oy * Wo + ox0;
    *reinterpret_cast<float2*>(y + o) = make_float2(ba, bb);
    *reinterpret_cast<uchar2*>(code + o) = make_uchar2((uint8_t)ca, (uint8_t)cb);
}

__device__ __forceinline__ float maxpool_gather(const float* g, const uint8_t* cd, int yy, int xx, int Ho, int Wo) {
    float acc = 0.f;
    // windows oy with oy*2-1 <= yy <= oy*2+1
    const int oy0 = max(yy >> 1, 0), oy1 = min((yy + 1) >> 1, Ho - 1);
    const int ox0 = max(xx >> 1, 0), ox1 = min((xx + 1) >> 1, Wo - 1);
    for (int oy = oy0; oy <= oy1; ++oy)
        for (int ox = ox0; ox <= ox1; ++ox) {
            const int want = (yy - (oy * 2 - 1)) * 3 + (xx - (ox * 2 - 1));
            if (cd[oy * Wo + ox] == want) acc += g[oy * Wo + ox];
        }
    return acc;
}

// VEC4: one thread = 4 consecutive input pixels (x0 % 4 == 0) of one row.  They are covered by the three
// windows ox = x0/2 .. x0/2+2 of one or two window rows: 3-6 (code, gy) pairs are fetched once and routed to
// the four pixels, instead of 16 byte loads + 16 float loads.
template <bool VEC4>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* gy, const uint8_t* code, float* dx, int H, int W,
                                                          int Ho, int Wo) {
    const size_t plane = blockIdx.y;
    const float* g = gy + plane * Ho * Wo;
    const uint8_t* cd = code + plane * Ho * Wo;
    if (!VEC4) {
        const int i = blockIdx.x * 256 + threadIdx.x;
        if (i >= H * W) return;
        const int yy = i / W, xx = i - yy * W;
        dx[plane * H * W + i] = maxpool_gather(g, cd, yy, xx, Ho, Wo);
        return;
    }
    const int W4 = W >> 2;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= H * W4) return;
    const int yy = i / W4, x0 = (i - yy * W4) * 4;
    float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
    const int oy0 = yy >> 1, oy1 = min((yy + 1) >> 1, Ho - 1);
    const int oxb = x0 >> 1;
    for (int oy = oy0; oy <= oy1; ++oy) {
        const int ky = yy - (oy * 2 - 1);             // row of the pixel inside that window
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int ox = oxb + j;
            if (ox < Wo) {
                const int cdv = cd[oy * Wo + ox];
                if (cdv / 3 == ky) {
                    const float gv = g[oy * Wo + ox];
                    const int px = ox * 2 - 1 + (cdv - ky * 3) - x0;    // pixel inside the group hit by this window
                    if (px == 0) out.x += gv;
                    else if (px == 1) out.y += gv;
                    else if (px == 2) out.z += gv;
                    else if (px == 3) out.w += gv;
                }
            }
        }
    }
    *reinterpret_cast<float4*>(dx + plane * H * W + (size_t)yy * W + x0) = out;
}

// eight consecutive input pixels per thread (x0 % 8 == 0, W % 8 == 0): the five windows ox = x0/2 .. x0/2+4 of a window row
// are one aligned 32-bit code load + one byte and one 16-byte gy load + one dword, routed to two float4 stores
__global__ __launch_bounds__(256) void maxpool_bwd8_kernel(const float* __restrict__ gy, const uint8_t* __restrict__ code,
                                                           float* __restrict__ dx, int H, int W, int Ho, int Wo,
                                                           const float* __restrict__ addend) {
    const size_t plane = blockIdx.y;
    const float* g = gy + plane * Ho * Wo;
    const uint8_t* cd = code + plane * Ho * Wo;
    const int W8 = W >> 3;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= H * W8) return;
    const int yy = i / W8, x0 = (i - yy * W8) * 8;
    float out[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (addend) {       // the pooled tensor's OTHER consumer's gradient (the decoder's skip connection), added on the way out
        const float* ap = addend + plane * H * W + (size_t)yy * W + x0;
        const float4 a0 = *reinterpret_cast<const float4*>(ap), a1 = *reinterpret_cast<const float4*>(ap + 4);
        out[0] = a0.x; out[1] = a0.y; out[2] = a0.z; out[3] = a0.w; out[4] = a1.x; out[5] = a1.y; out[6] = a1.z; out[7] = a1.w;
    }
    const int oy0 = yy >> 1, oy1 = min((yy + 1) >> 1, Ho - 1);
    const int oxb = x0 >> 1;                                  // multiple of 4
    for (int oy = oy0; oy <= oy1; ++oy) {
        const int ky = yy - (oy * 2 - 1);
        const unsigned c4 = *reinterpret_cast<const unsigned*>(cd + (size_t)oy * Wo + oxb);
        const float4 g4 = *reinterpret_cast<const float4*>(g + (size_t)oy * Wo + oxb);
        const bool has5 = oxb + 4 < Wo;
        const int c5 = has5 ? cd[(size_t)oy * Wo + oxb + 4] : 255;
        const float g5 = has5 ? g[(size_t)oy * Wo + oxb + 4] : 0.f;
        const int cds[5] = {(int)(c4 & 255u), (int)((c4 >> 8) & 255u), (int)((c4 >> 16) & 255u), (int)(c4 >> 24), c5};
        const float gs[5] = {g4.x, g4.y, g4.z, g4.w, g5};
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            if (cds[j] / 3 == ky) {
                const int px = (oxb + j) * 2 - 1 + (cds[j] - ky * 3) - x0;    // pixel of the group hit by this window (-1 .. 9)
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (px == q) out[q] += gs[j];
            }
        }
    }
    float* o = dx + plane * H * W + (size_t)yy * W + x0;
    *reinterpret_cast<float4*>(o) = make_float4(out[0], out[1], out[2], out[3]);
    *reinterpret_cast<float4*>(o + 4) = make_float4(out[4], out[5], out[6], out[7]);
}

}  // namespace dc

extern "C" int dc_maxpool3x3s2_fwd(const float* x, float* y, uint8_t* code, int NC, int H, int W, void* stream) {
    if (!x || !y || !code || NC <= 0 || H < 2 || W < 2 || NC > 65535) return DC_EINVAL;
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    if ((W & 3) == 0 && (Wo & 1) == 0)
        hipLaunchKernelGGL(dc::maxpool_fwd2_kernel, dim3(dc::ceil_div(Ho * (Wo >> 1), 256), NC), dim3(256), 0, (hipStream_t)stream, x, y,
                           code, H, W, Ho, Wo);
    else
        hipLaunchKernelGGL(dc::maxpool_fwd_kernel, dim3(dc::ceil_div(Ho * Wo, 256), NC), dim3(256), 0, (hipStream_t)stream, x, y,
                           code, H, W, Ho, Wo);
    DC_CHECK_LAUNCH();
    return DC_OK;
}

extern "C" int dc_maxpool3x3s2_bwd(const float* gy, const uint8_t* code, float* dx, int NC, int H, int W, void* stream) {
    return dc_maxpool3x3s2_bwd_add(gy, code, dx, nullptr, NC, H, W, stream);
}

extern "C" int dc_maxpool3x3s2_bwd_add(const float* gy, const uint8_t* code, float* dx, const float* addend, int NC, int H, int W,
                                       void* stream) {
    if (!gy || !code || !dx || NC <= 0 || H < 2 || W < 2 || NC > 65535) return DC_EINVAL;
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    if (addend && (W & 7)) {          // (the vectorised kernel adds on the way out; the others in a pass of their own)
        const int rc = dc_maxpool3x3s2_bwd_add(gy, code, dx, nullptr, NC, H, W, stream);
        return rc != DC_OK ? rc : dc::add_inplace(dx, addend, (size_t)NC * H * W, (hipStream_t)stream);
    }
    if ((W & 7) == 0)
        hipLaunchKernelGGL(dc::maxpool_bwd8_kernel, dim3(dc::ceil_div(H * (W >> 3), 256), NC), dim3(256), 0, (hipStream_t)stream, gy, code,
                           dx, H, W, Ho, Wo, addend);
    else if ((W & 3) == 0)
        hipLaunchKernelGGL(dc::maxpool_bwd_kernel<true>, dim3(dc::ceil_div(H * (W >> 2), 256), NC), dim3(256), 0,
                           (hipStream_t)stream, gy, code, dx, H, W, Ho, Wo);
    else
        hipLaunchKernelGGL(dc::maxpool_bwd_kernel<false>, dim3(dc::ceil_div(H * W, 256), NC), dim3(256), 0,
                           (hipStream_t)stream, gy, code, dx, H, W, Ho, Wo);
    DC_CHECK_LAUNCH();
    return DC_OK;
}
