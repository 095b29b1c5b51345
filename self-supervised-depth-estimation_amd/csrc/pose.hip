// a5: transformation_from_parameters (reference layers.py:28-103), forward and backward.
// The 6-parameter -> 4x4 map is tiny (one thread per batch element), so the backward is done
// with forward-mode dual numbers carried through the SAME templated forward code: one
// statement of the formula, derivative correct by construction.
#include "dc_common.h"

namespace dc {

template <int N>
struct Dual {
    float v;
    float d[N];
};

struct F1 {   // plain scalar with the same interface
    float v;
};

template <int N> __device__ __forceinline__ Dual<N> mk(float c) { Dual<N> r; r.v = c; for (int i = 0; i < N; ++i) r.d[i] = 0.f; return r; }
template <int N> __device__ __forceinline__ Dual<N> operator+(Dual<N> a, Dual<N> b) { for (int i = 0; i < N; ++i) a.d[i] += b.d[i]; a.v += b.v; return a; }
template <int N> __device__ __forceinline__ Dual<N> operator-(Dual<N> a, Dual<N> b) { for (int i = 0; i < N; ++i) a.d[i] -= b.d[i]; a.v -= b.v; return a; }
template <int N> __device__ __forceinline__ Dual<N> operator-(Dual<N> a) { for (int i = 0; i < N; ++i) a.d[i] = -a.d[i]; a.v = -a.v; return a; }
template <int N> __device__ __forceinline__ Dual<N> operator*(Dual<N> a, Dual<N> b) { Dual<N> r; r.v = a.v * b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
template <int N> __device__ __forceinline__ Dual<N> operator/(Dual<N> a, Dual<N> b) { Dual<N> r; float ib = 1.f / b.v; r.v = a.v * ib; for (int i = 0; i < N; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * ib; return r; }
template <int N> __device__ __forceinline__ Dual<N> dsqrt(Dual<N> a) { Dual<N> r; r.v = sqrtf(a.v); float k = (a.v > 0.f) ? 0.5f / r.v : 0.f; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * k; return r; }
template <int N> __device__ __forceinline__ Dual<N> dsin(Dual<N> a) { Dual<N> r; r.v = sinf(a.v); float k = cosf(a.v); for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * k; return r; }
template <int N> __device__ __forceinline__ Dual<N> dcos(Dual<N> a) { Dual<N> r; r.v = cosf(a.v); float k = -sinf(a.v); for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * k; return r; }

__device__ __forceinline__ F1 operator+(F1 a, F1 b) { return F1{a.v + b.v}; }
__device__ __forceinline__ F1 operator-(F1 a, F1 b) { return F1{a.v - b.v}; }
__device__ __forceinline__ F1 operator-(F1 a) { return F1{-a.v}; }
__device__ __forceinline__ F1 operator*(F1 a, F1 b) { return F1{a.v * b.v}; }
__device__ __forceinline__ F1 operator/(F1 a, F1 b) { return F1{a.v / b.v}; }
__device__ __forceinline__ F1 dsqrt(F1 a) { return F1{sqrtf(a.v)}; }
__device__ __forceinline__ F1 dsin(F1 a) { return F1{sinf(a.v)}; }
__device__ __forceinline__ F1 dcos(F1 a) { return F1{cosf(a.v)}; }

// S = F1 or Dual<6>.  `cst(c)` lifts a constant.
template <class S, class Cst>
__device__ __forceinline__ void pose_matrix(const S aa[3], const S tr[3], bool invert, S M[16], Cst cst) {
    // rot_from_axisangle (layers.py:64-103)
    S angle = dsqrt(aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2]);
    S den = angle + cst(1e-7f);
    S x = aa[0] / den, y = aa[1] / den, z = aa[2] / den;
    S ca = dcos(angle), sa = dsin(angle);
    S C = cst(1.f) - ca;
    S xs = x * sa, ys = y * sa, zs = z * sa;
    S xC = x * C, yC = y * C, zC = z * C;
    S xyC = x * yC, yzC = y * zC, zxC = z * xC;
    S R[9];
    R[0] = x * xC + ca; R[1] = xyC - zs;    R[2] = zxC + ys;
    R[3] = xyC + zs;    R[4] = y * yC + ca; R[5] = yzC - xs;
    R[6] = zxC - ys;    R[7] = yzC + xs;    R[8] = z * zC + ca;
    S t[3] = {tr[0], tr[1], tr[2]};
    if (invert) {   // R^T and -t, then M = R @ T  (layers.py:34-43)
        S tmp;
        tmp = R[1]; R[1] = R[3]; R[3] = tmp;
        tmp = R[2]; R[2] = R[6]; R[6] = tmp;
        tmp = R[5]; R[5] = R[7]; R[7] = tmp;
        for (int i = 0; i < 3; ++i) t[i] = -t[i];
    }
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) M[i * 4 + j] = R[i * 3 + j];
        // T @ R: last column is t;  R @ T: last column is R t
        M[i * 4 + 3] = invert ? (R[i * 3 + 0] * t[0] + R[i * 3 + 1] * t[1] + R[i * 3 + 2] * t[2]) : t[i];
    }
    M[12] = cst(0.f); M[13] = cst(0.f); M[14] = cst(0.f); M[15] = cst(1.f);
}

__global__ void pose_fwd_kernel(const float* aa, const float* tr, int invert, float* M, int B) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    F1 a[3], t[3], m[16];
    for (int i = 0; i < 3; ++i) { a[i].v = aa[b * 3 + i]; t[i].v = tr[b * 3 + i]; }
    pose_matrix(a, t, invert != 0, m, [](float c) { return F1{c}; });
    for (int i = 0; i < 16; ++i) M[b * 16 + i] = m[i].v;
}

__global__ void pose_bwd_kernel(const float* aa, const float* tr, int invert, const float* dM, float* daa,
                                float* dtr, int B) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    Dual<6> a[3], t[3], m[16];
    for (int i = 0; i < 3; ++i) {
        a[i] = mk<6>(aa[b * 3 + i]); a[i].d[i] = 1.f;
        t[i] = mk<6>(tr[b * 3 + i]); t[i].d[3 + i] = 1.f;
    }
    pose_matrix(a, t, invert != 0, m, [](float c) { return mk<6>(c); });
    float g[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 16; ++i) {
        float u = dM[b * 16 + i];
        for (int k = 0; k < 6; ++k) g[k] += u * m[i].d[k];
    }
    for (int k = 0; k < 3; ++k) { daa[b * 3 + k] = g[k]; dtr[b * 3 + k] = g[3 + k]; }
}

}  // namespace dc

using namespace dc;

extern "C" int dc_pose_matrix_fwd(const float* axisangle, const float* translation, int invert, float* M, int B,
                                  void* stream) {
    if (!axisangle || !translation || !M || B <= 0) return DC_EINVAL;
    hipLaunchKernelGGL(pose_fwd_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, (hipStream_t)stream, axisangle,
                       translation, invert, M, B);
    DC_CHECK_LAUNCH();
    return DC_OK;
}

extern "C" int dc_pose_matrix_bwd(const float* axisangle, const float* translation, int invert, const float* dM,
                                  float* d_axisangle, float* d_translation, int B, void* stream) {
    if (!axisangle || !translation || !dM || !d_axisangle || !d_translation || B <= 0) return DC_EINVAL;
    hipLaunchKernelGGL(pose_bwd_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, (hipStream_t)stream, axisangle,
                       translation, invert, dM, d_axisangle, d_translation, B);
    DC_CHECK_LAUNCH();
    return DC_OK;
}
