// Shared device pieces of the tiled fp32-MFMA GEMM kernels (gemm1x1.hip: 1x1 convolutions; convgemm.hip: the strided
// 3x3 / 7x7 convolutions of the ResNet trunks as implicit GEMMs): LDS image strides, operand reads for
// v_mfma_f32_16x16x4_f32 and the MFMA step.  See gemm1x1.hip's header for the layout discussion.
#pragma once
#include "dc_common.h"

namespace dc {

using gf4 = __attribute__((ext_vector_type(4))) float;
using gf2 = __attribute__((ext_vector_type(2))) float;

constexpr int GKC = 32;                // reduction chunk

// ---- LDS strides -------------------------------------------------------------------------------------
// index-contiguous image [KC][W]: row stride so that the MT-wide reads of a 16-lane k-group are conflict-free
// (a k-group pair of one 32-lane read group sits two reduction rows apart: element 2 k' + s)
template <int T, int W>
struct IdxStride { static constexpr int v = (T == 4) ? W : W + 16; };

// ---- operand reads: fill a[t][s] for the two MFMA steps s of reduction octet q ---------------------------------
// index-contiguous image: S[(red)][stride], tile t of lane i <-> index base + i * T + t
template <int T, int STRIDE>
__device__ __forceinline__ void read_idx(const float* S, int base, int q, int lane, float (&a)[4][2]) {
    const int i = lane & 15, kp = lane >> 4;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const float* p = S + (q * 8 + 2 * kp + s) * STRIDE + base + i * T;
        if constexpr (T == 4) {
            const gf4 v = *reinterpret_cast<const gf4*>(p);
            a[0][s] = v.x; a[1][s] = v.y; a[2][s] = v.z; a[3][s] = v.w;
        } else {
            const gf2 v = *reinterpret_cast<const gf2*>(p);
            a[0][s] = v.x; a[1][s] = v.y;
        }
    }
}
// reduction-contiguous image: S[(index)][KC + RP], tile t of lane i <-> index base + t * 16 + i
constexpr int RP = 2;                  // row padding of the reduction-contiguous images (rows are 8-byte aligned)
template <int T, int KC>
__device__ __forceinline__ void read_red(const float* S, int base, int q, int lane, float (&a)[4][2]) {
    const int i = lane & 15, kp = lane >> 4;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const gf2 v = *reinterpret_cast<const gf2*>(S + (base + t * 16 + i) * (KC + RP) + q * 8 + 2 * kp);
        a[t][0] = v.x; a[t][1] = v.y;
    }
}
// 16 bytes into a reduction-contiguous image (its rows are only 8-byte aligned: two ds_write_b64)
__device__ __forceinline__ void store_red4(float* p, gf4 v) {
    *reinterpret_cast<gf2*>(p) = gf2{v.x, v.y};
    *reinterpret_cast<gf2*>(p + 2) = gf2{v.z, v.w};
}

template <int MT, int NT>
__device__ __forceinline__ void mma_octet(const float (&a)[4][2], const float (&b)[4][2], gf4 (&acc)[MT][NT]) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][s], b[nt][s], acc[mt][nt], 0, 0, 0);
}


// One reduction chunk of 32 = four octets, software-pipelined in registers: the operands of octet q+1 are read from LDS
// while octet q multiplies.  (With one operand set the compiler issues each octet's ds_reads right before the s_waitcnt
// in front of its 16 MFMAs -- the LDS latency is exposed four times per chunk; a second set costs 16 VGPRs.)
template <int MT, int NT, class RA, class RB>
__device__ __forceinline__ void mma_chunk32(RA&& ra, RB&& rb, gf4 (&acc)[MT][NT]) {
    float a0[4][2], b0[4][2], a1[4][2], b1[4][2];
    // sched_barrier(0): the machine scheduler otherwise sinks every read group back in front of its consumer
    ra(0, a0); rb(0, b0);
    ra(1, a1); rb(1, b1);
    __builtin_amdgcn_sched_barrier(0);
    mma_octet<MT, NT>(a0, b0, acc);
    __builtin_amdgcn_sched_barrier(0);
    ra(2, a0); rb(2, b0);
    __builtin_amdgcn_sched_barrier(0);
    mma_octet<MT, NT>(a1, b1, acc);
    __builtin_amdgcn_sched_barrier(0);
    ra(3, a1); rb(3, b1);
    __builtin_amdgcn_sched_barrier(0);
    mma_octet<MT, NT>(a0, b0, acc);
    mma_octet<MT, NT>(a1, b1, acc);
}

extern __shared__ float g1_smem[];

// Sum of `splits` slabs of n elements in a fixed order, 16 split groups x 16 lanes per block: a group adds a contiguous
// range of slabs for 16 consecutive outputs (64- / 256-byte segments), the 16 partials are combined through LDS in group
// order.  (One thread per output looping over 150-250 slabs measured 38-79 us on the stem's 9.4 K outputs.)
template <typename V>
__global__ __launch_bounds__(256) void slab_reduce16_kernel(const V* __restrict__ slab, V* __restrict__ out, int splits, int n) {
    __shared__ V sm[256];
    const int g = threadIdx.x >> 4, l = threadIdx.x & 15;
    const int i = blockIdx.x * 16 + l;
    const int per = (splits + 15) / 16, s0 = g * per, s1 = min(s0 + per, splits);
    V t = V{};
    if (i < n)
        for (int s = s0; s < s1; ++s) t += slab[(size_t)s * n + i];
    sm[threadIdx.x] = t;
    __syncthreads();
    if (g == 0 && i < n) {
        V r = sm[l];
#pragma unroll
        for (int k = 1; k < 16; ++k) r += sm[k * 16 + l];
        out[i] = r;
    }
}

}  // namespace dc
