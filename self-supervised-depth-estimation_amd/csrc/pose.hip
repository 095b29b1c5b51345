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

// ---- the pose networks' tail as one launch each way (PoseDecoder / PoseCNN: networks/pose_decoder.py:50-54, pose_cnn.py:48-52, and
// the cam_T_cam of trainer.py:416-419,436-440):  vec = scale * mean over the P pixels of y (N, 6 nf, P), viewed (N, nf, [axisangle |
// translation]); group g takes rows [row0, row0 + rows) of frame `slot` to its own (rows, 4, 4) matrices.
constexpr int POSE_MAXG = 8, POSE_MAXC = 6 * 8;
struct PoseHead {
    int ng, C, P;
    float scale;
    int row0[POSE_MAXG], rows[POSE_MAXG], slot[POSE_MAXG], invert[POSE_MAXG];
    float* M[POSE_MAXG];
    const float* dM[POSE_MAXG];
};

__global__ __launch_bounds__(256) void pose_head_fwd_kernel(const float* __restrict__ y, float* __restrict__ vec, PoseHead h) {
    __shared__ float s_vec[POSE_MAXC];
    const int n = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int c = wv; c < h.C; c += 4) {
        const float* src = y + ((size_t)n * h.C + c) * h.P;
        float s = 0.f;
        for (int p = lane; p < h.P; p += 64) s += src[p];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) { s = s * (h.scale / (float)h.P); s_vec[c] = s; vec[(size_t)n * h.C + c] = s; }
    }
    __syncthreads();
    const int g = threadIdx.x;
    if (g < h.ng && n >= h.row0[g] && n < h.row0[g] + h.rows[g]) {
        F1 a[3], t[3], m[16];
        for (int i = 0; i < 3; ++i) { a[i].v = s_vec[h.slot[g] * 6 + i]; t[i].v = s_vec[h.slot[g] * 6 + 3 + i]; }
        pose_matrix(a, t, h.invert[g] != 0, m, [](float c) { return F1{c}; });
        float* M = h.M[g] + (size_t)(n - h.row0[g]) * 16;
        for (int i = 0; i < 16; ++i) M[i] = m[i].v;
    }
}

// d_y[n, c, p] = scale / P * d_vec[n, c]; d_vec = the groups' matrix gradients through the same dual-number statement as
// pose_bwd_kernel; channels no group reads get zeros (the reference predicts nf frames and uses one: trainer.py:416)
__global__ __launch_bounds__(256) void pose_head_bwd_kernel(const float* __restrict__ vec, float* __restrict__ d_y, PoseHead h) {
    __shared__ float s_d[POSE_MAXC];
    const int n = blockIdx.x;
    if (threadIdx.x < h.C) s_d[threadIdx.x] = 0.f;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int g = 0; g < h.ng; ++g) {
            if (!h.dM[g] || n < h.row0[g] || n >= h.row0[g] + h.rows[g]) continue;
            const float* v = vec + (size_t)n * h.C + h.slot[g] * 6;
            Dual<6> a[3], t[3], m[16];
            for (int i = 0; i < 3; ++i) {
                a[i] = mk<6>(v[i]); a[i].d[i] = 1.f;
                t[i] = mk<6>(v[3 + i]); t[i].d[3 + i] = 1.f;
            }
            pose_matrix(a, t, h.invert[g] != 0, m, [](float c) { return mk<6>(c); });
            const float* dM = h.dM[g] + (size_t)(n - h.row0[g]) * 16;
            for (int i = 0; i < 16; ++i) {
                const float u = dM[i];
                for (int k = 0; k < 6; ++k) s_d[h.slot[g] * 6 + k] += u * m[i].d[k];
            }
        }
    }
    __syncthreads();
    const float k = h.scale / (float)h.P;
    float* dst = d_y + (size_t)n * h.C * h.P;
    for (int i = threadIdx.x; i < h.C * h.P; i += 256) dst[i] = s_d[i / h.P] * k;
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

static int pose_head_pack(PoseHead& h, int N, int nf, int P, float scale, const dc_pose_group* groups, int ngroups) {
    if (N <= 0 || nf <= 0 || 6 * nf > POSE_MAXC || P <= 0 || !groups || ngroups <= 0 || ngroups > POSE_MAXG) return DC_EINVAL;
    h.ng = ngroups; h.C = 6 * nf; h.P = P; h.scale = scale;
    for (int g = 0; g < ngroups; ++g) {
        const dc_pose_group& q = groups[g];
        if (q.row0 < 0 || q.rows <= 0 || q.row0 + q.rows > N || q.slot < 0 || q.slot >= nf) return DC_EINVAL;
        h.row0[g] = q.row0; h.rows[g] = q.rows; h.slot[g] = q.slot; h.invert[g] = q.invert ? 1 : 0;
        h.M[g] = nullptr; h.dM[g] = nullptr;
    }
    return DC_OK;
}

extern "C" int dc_pose_head_fwd(const float* y, int N, int nf, int P, float scale, const dc_pose_group* groups, int ngroups, float* vec,
                                float* const* M, void* stream) {
    PoseHead h{};
    if (!y || !vec || !M) return DC_EINVAL;
    const int rc = pose_head_pack(h, N, nf, P, scale, groups, ngroups);
    if (rc != DC_OK) return rc;
    for (int g = 0; g < ngroups; ++g) {
        if (!M[g]) return DC_EINVAL;
        h.M[g] = M[g];
    }
    hipLaunchKernelGGL(pose_head_fwd_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, y, vec, h);
    DC_CHECK_LAUNCH();
    return DC_OK;
}

extern "C" int dc_pose_head_bwd(const float* vec, int N, int nf, int P, float scale, const dc_pose_group* groups, int ngroups,
                                const float* const* dM, float* d_y, void* stream) {
    PoseHead h{};
    if (!vec || !dM || !d_y) return DC_EINVAL;
    const int rc = pose_head_pack(h, N, nf, P, scale, groups, ngroups);
    if (rc != DC_OK) return rc;
    for (int g = 0; g < ngroups; ++g) h.dM[g] = dM[g];          // (NULL: that group's matrices had no gradient)
    hipLaunchKernelGGL(pose_head_bwd_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, vec, d_y, h);
    DC_CHECK_LAUNCH();
    return DC_OK;
}
