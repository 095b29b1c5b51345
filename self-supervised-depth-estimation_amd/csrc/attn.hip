// AttentionConv of the Fusion_v3 front-end (reference networks/fusion_v2.py:46-98, as used by ResidualAttentionUnit
// :101-137 with kernel 3, stride 1, padding 1, groups 1, bias=True) on 2- and 4-channel maps, forward and backward, each as
// ONE kernel: the reference materialises q, k, v, two (B,C,H,W,3,3) unfolds, the logits, the softmax and the einsum
// (~60 passes over the map per AttentionConv); here x is read once and y written once.
//
//   q = Wq r + bq            r = relu(x) if relu_in else x                (the unit's in-place ReLUs, fusion_v2.py:130-134)
//   k, v = 1x1 convolutions of the ZERO-PADDED r (an out-of-image tap carries k = bk, v = bv)
//   logit_t[c] = q[c] * (k_t[c] + rel_t[c]),  rel = rel_h[dy] for c < C/2, rel_w[dx] otherwise;  t = (dy, dx) in 3x3
//   y[c] = sum_t softmax_t(logit)[c] * v_t[c]  (+ relu(res) or res: the unit's skip connection, which adds the ReLU'd input)
//
// Forward: a block owns a 16 x 32 pixel tile; r of the tile + 1-pixel ring goes to LDS, k and v of that region are computed
// once into LDS ({k[C], v[C]} per pixel: one or two ds_read_b128 per tap), every thread then finishes two pixels.
//
// Backward (recompute, nothing saved by the forward): a block owns a 6 x 30 tile of dx.  With p' = p + o_t the position tap t
// of pixel p reads, the gradients of the k and v maps are GATHERED:
//   dV[p'] = sum_t a_t[p] g[p],     dK[p'] = sum_t a_t[p] g[p] (v[p'] - y[p]) q[p],     p = p' - o_t
// so the block evaluates attention weights for the tile + 1 ring (8 x 32 = one pixel per thread), parks dv_t = a_t g and
// dk_t = dv_t (v_t - y) q in LDS one channel at a time (channels are independent until the 1x1 convolutions), and each
// thread gathers its own pixel's 9 contributions.  dq needs no exchange.  Parameter gradients (Wq bq Wk bk Wv bv rel_h rel_w)
// are accumulated per thread in registers, reduced per block in fixed order and written as one partial row per block; a
// second tiny kernel sums the rows in fixed order.  No atomics: bitwise reproducible.
#include "dc_common.h"

namespace dc {

struct AttnArgs {
    dc_attn_map x;         // input channels (before the optional ReLU), gathered from up to C tensors
    dc_attn_map res;       // skip input (res.ptr[0] null: none)
    const float* gy;       // backward: gradient of y, (B,C,H,W)
    float* y;              // forward output, (B,C,H,W)
    dc_attn_map dx;        // backward: where the gradient of each input channel goes
    const float* dx_add;   // backward: (B,C,H,W) added to dx before it is stored (null: nothing)
    dc_attn_map dres;      // backward: gradient of res (dres.ptr[0] null: not wanted)
    float* partial;        // backward: [nblocks][NP] parameter-gradient partials
    const float *wq, *bq, *wk, *bk, *wv, *bv, *rel_h, *rel_w;
    int B, H, W;
    int relu_in, relu_res;
    int tiles_x, tiles_y;
};

// element (b, channel c, y, x) of a channel map (include/depthcore.h: dc_attn_map)
__device__ __forceinline__ size_t attn_off(const dc_attn_map& m, int c, int b, int y, int x, int H, int W) {
    const size_t base = (size_t)b * (size_t)m.batch_stride[c];
    if (m.mode[c] == DC_ATTN_PIXEL_SHUFFLE2) {
        const int h2 = H >> 1, w2 = W >> 1;
        return base + (size_t)((y & 1) * 2 + (x & 1)) * h2 * w2 + (size_t)(y >> 1) * w2 + (x >> 1);
    }
    return base + (size_t)y * W + x;
}

template <int C>
struct AttnP {           // parameters in registers (uniform across the block: the compiler keeps them in SGPRs)
    float wq[C][C], wk[C][C], wv[C][C], bq[C], bk[C], bv[C], rh[3], rw[3];
    __device__ __forceinline__ void load(const AttnArgs& a) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            bq[c] = a.bq[c]; bk[c] = a.bk[c]; bv[c] = a.bv[c];
#pragma unroll
            for (int i = 0; i < C; ++i) { wq[c][i] = a.wq[c * C + i]; wk[c][i] = a.wk[c * C + i]; wv[c][i] = a.wv[c * C + i]; }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) { rh[i] = a.rel_h[i]; rw[i] = a.rel_w[i]; }
    }
    __device__ __forceinline__ float rel(int c, int dy, int dx) const { return c < C / 2 ? rh[dy] : rw[dx]; }
};

// ---- stage r = relu?(x) of a (RH x RW) region whose top-left image coordinate is (y0, x0) into LDS [pix][C]; 0 outside
template <int C, int RH, int RW>
__device__ __forceinline__ void attn_stage_x(const AttnArgs& a, int b, int y0, int x0, float* xs) {
    // All loads of the region are issued back to back from CLAMPED coordinates (always a valid address) and zeroed at the
    // commit: the earlier `if (inside) v = ptr[c][...]` with a run-time channel index was a dependent pointer load plus a
    // branch around every element -- ten serialised round trips per block in front of the arithmetic.
    constexpr int NPX = RH * RW, IT = (NPX + 255) / 256;
    float v[C][IT];
    const int h2 = a.H >> 1, w2 = a.W >> 1;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const float* base = a.x.ptr[c] + (size_t)b * (size_t)a.x.batch_stride[c];
        const bool ps = a.x.mode[c] == DC_ATTN_PIXEL_SHUFFLE2;
#pragma unroll
        for (int k = 0; k < IT; ++k) {
            const int p = (int)threadIdx.x + k * 256;
            const int ry = p / RW, rx = p - ry * RW;
            const int gy_ = min(max(y0 + ry, 0), a.H - 1), gx_ = min(max(x0 + rx, 0), a.W - 1);
            const size_t off = ps ? (size_t)((gy_ & 1) * 2 + (gx_ & 1)) * h2 * w2 + (size_t)(gy_ >> 1) * w2 + (gx_ >> 1)
                                  : (size_t)gy_ * a.W + gx_;
            v[c][k] = base[off];
        }
    }
#pragma unroll
    for (int k = 0; k < IT; ++k) {
        const int p = (int)threadIdx.x + k * 256;
        if (p >= NPX) continue;
        const int ry = p / RW, rx = p - ry * RW;
        const int gy_ = y0 + ry, gx_ = x0 + rx;
        const bool inside = gy_ >= 0 && gy_ < a.H && gx_ >= 0 && gx_ < a.W;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float t = v[c][k];
            if (a.relu_in) t = fmaxf(t, 0.f);
            xs[p * C + c] = inside ? t : 0.f;
        }
    }
}
// ---- k, v of every pixel of the staged region -> LDS kv[pix][2C] = {k[0..C), v[0..C)}
template <int C, int NPIX>
__device__ __forceinline__ void attn_kv(const AttnP<C>& P, const float* xs, float* kv) {
    for (int p = threadIdx.x; p < NPIX; p += 256) {
        float r[C];
#pragma unroll
        for (int i = 0; i < C; ++i) r[i] = xs[p * C + i];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float k = P.bk[c], v = P.bv[c];
#pragma unroll
            for (int i = 0; i < C; ++i) { k = fmaf(P.wk[c][i], r[i], k); v = fmaf(P.wv[c][i], r[i], v); }
            kv[p * 2 * C + c] = k;
            kv[p * 2 * C + C + c] = v;
        }
    }
}

// =====================================================================================================================
// forward
// =====================================================================================================================
constexpr int AF_TH = 16, AF_TW = 32;              // output tile: 512 pixels, two per thread
template <int C>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnArgs a) {
    constexpr int RH = AF_TH + 2, RW = AF_TW + 2;
    __shared__ float xs[RH * RW * C];
    __shared__ float kv[RH * RW * 2 * C];
    AttnP<C> P;
    P.load(a);
    const int b = blockIdx.z, ty0 = blockIdx.y * AF_TH, tx0 = blockIdx.x * AF_TW;
    attn_stage_x<C, RH, RW>(a, b, ty0 - 1, tx0 - 1, xs);
    __syncthreads();
    attn_kv<C, RH * RW>(P, xs, kv);
    __syncthreads();
    const size_t plane = (size_t)a.H * a.W;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int t = threadIdx.x + h * 256;
        const int ly = t / AF_TW, lx = t - ly * AF_TW;
        const int py = ty0 + ly, px = tx0 + lx;
        if (py >= a.H || px >= a.W) continue;
        const int pc = (ly + 1) * RW + lx + 1;                  // region index of the pixel itself
        // the residual input of the pixel, fetched before the arithmetic (a load next to the stores below made each store
        // wait for the previous one)
        float rres[C];
#pragma unroll
        for (int c = 0; c < C; ++c) rres[c] = 0.f;
        if (a.res.ptr[0]) {
#pragma unroll
            for (int c = 0; c < C; ++c) rres[c] = a.res.ptr[c][attn_off(a.res, c, b, py, px, a.H, a.W)];
        }
        float r[C], q[C];
#pragma unroll
        for (int i = 0; i < C; ++i) r[i] = xs[pc * C + i];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            q[c] = P.bq[c];
#pragma unroll
            for (int i = 0; i < C; ++i) q[c] = fmaf(P.wq[c][i], r[i], q[c]);
        }
        float lg[C][9], vt[C][9];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const float* e = kv + ((ly + dy) * RW + lx + dx) * 2 * C;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    lg[c][dy * 3 + dx] = q[c] * (e[c] + P.rel(c, dy, dx));
                    vt[c][dy * 3 + dx] = e[C + c];
                }
            }
        float out[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float m = lg[c][0];
#pragma unroll
            for (int k = 1; k < 9; ++k) m = fmaxf(m, lg[c][k]);
            float s = 0.f, o = 0.f;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const float e = __expf(lg[c][k] - m);
                s += e;
                o = fmaf(e, vt[c][k], o);
            }
            o = o / s;
            if (a.res.ptr[0]) o += a.relu_res ? fmaxf(rres[c], 0.f) : rres[c];
            out[c] = o;
        }
#pragma unroll
        for (int c = 0; c < C; ++c) a.y[((size_t)b * C + c) * plane + (size_t)py * a.W + px] = out[c];
    }
}

// =====================================================================================================================
// backward
// =====================================================================================================================
constexpr int AB_TH = 6, AB_TW = 30;               // dx tile; tile + 1 ring = 8 x 32 = 256 evaluated pixels, one per thread
template <int C>
struct AttnNP { static constexpr int v = 3 * (C * C + C) + 6; };

template <int C>
__global__ __launch_bounds__(256) void attn_bwd_kernel(AttnArgs a) {
    constexpr int EW = AB_TW + 2;                       // evaluated region (tile + ring 1): 8 x 32
    constexpr int RH = AB_TH + 4, RW = AB_TW + 4;       // staged region (tile + ring 2)
    constexpr int NP = AttnNP<C>::v;
    constexpr int ADS = 19;                             // per-pixel stride of the parked tap gradients (18 + 1 pad)
    __shared__ float xs[RH * RW * C];
    __shared__ float kv[RH * RW * 2 * C];
    __shared__ float ad[256 * ADS];                     // [pix][0..9) dv_t, [9..18) dk_t   (one channel at a time)
    __shared__ float red[4][NP];
    AttnP<C> P;
    P.load(a);
    const int b = blockIdx.z, ty0 = blockIdx.y * AB_TH, tx0 = blockIdx.x * AB_TW;
    attn_stage_x<C, RH, RW>(a, b, ty0 - 2, tx0 - 2, xs);
    __syncthreads();
    attn_kv<C, RH * RW>(P, xs, kv);
    __syncthreads();
    const size_t plane = (size_t)a.H * a.W;

    float pg[NP];                                       // this thread's parameter-gradient partials
#pragma unroll
    for (int i = 0; i < NP; ++i) pg[i] = 0.f;
    // pg layout: dWq [0, C*C), dbq, dWk, dbk, dWv, dbv, drel_h[3], drel_w[3]
    constexpr int O_WQ = 0, O_BQ = C * C, O_WK = C * C + C, O_BK = 2 * C * C + C, O_WV = 2 * C * C + 2 * C,
                  O_BV = 3 * C * C + 2 * C, O_RH = 3 * C * C + 3 * C, O_RW = O_RH + 3;

    // this thread's evaluated pixel
    const int t = threadIdx.x;
    const int ey = t / EW, ex = t - ey * EW;
    const int py = ty0 - 1 + ey, px = tx0 - 1 + ex;
    const bool inimg = py >= 0 && py < a.H && px >= 0 && px < a.W;
    const bool intile = inimg && ey >= 1 && ey <= AB_TH && ex >= 1 && ex <= AB_TW;
    const int pc = (ey + 1) * RW + ex + 1;              // staged-region index of the pixel
    float dq[C], dK[C], dV[C];
#pragma unroll
    for (int c = 0; c < C; ++c) dq[c] = dK[c] = dV[c] = 0.f;
    // everything this thread reads from HBM besides the staged region, fetched up front from clamped coordinates (used only
    // where the pixel is inside): the output gradient, the gradient to add to dx, the residual input of the ReLU mask
    const int cy = min(max(py, 0), a.H - 1), cx = min(max(px, 0), a.W - 1);
    float gpix[C], dxadd[C], resv[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const size_t off = ((size_t)b * C + c) * plane + (size_t)cy * a.W + cx;
        gpix[c] = a.gy[off];
        dxadd[c] = a.dx_add ? a.dx_add[off] : 0.f;
        resv[c] = (a.dres.ptr[0] && a.relu_res) ? a.res.ptr[c][attn_off(a.res, c, b, cy, cx, a.H, a.W)] : 1.f;
    }

#pragma unroll
    for (int c = 0; c < C; ++c) {
        // ---- evaluate channel c at this thread's pixel, park dv_t / dk_t
        float dv[9], dk[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) dv[k] = dk[k] = 0.f;
        if (inimg) {
            float q = P.bq[c];
#pragma unroll
            for (int i = 0; i < C; ++i) q = fmaf(P.wq[c][i], xs[pc * C + i], q);
            float lg[9], vt[9], kr[9];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const float* e = kv + ((ey + dy) * RW + ex + dx) * 2 * C;
                    kr[dy * 3 + dx] = e[c] + P.rel(c, dy, dx);
                    lg[dy * 3 + dx] = q * kr[dy * 3 + dx];
                    vt[dy * 3 + dx] = e[C + c];
                }
            float m = lg[0];
#pragma unroll
            for (int k = 1; k < 9; ++k) m = fmaxf(m, lg[k]);
            float s = 0.f, o = 0.f;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                lg[k] = __expf(lg[k] - m);
                s += lg[k];
                o = fmaf(lg[k], vt[k], o);
            }
            const float inv = 1.f / s;
            o *= inv;
            const float g = gpix[c];
            float dqc = 0.f, sdk = 0.f, drh[3] = {0.f, 0.f, 0.f}, drw[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const float ag = lg[k] * inv * g;           // dv_t
                const float dl = ag * (vt[k] - o);          // d logit_t
                dv[k] = ag;
                dk[k] = dl * q;
                dqc = fmaf(dl, kr[k], dqc);
                sdk += dk[k];
                if (c < C / 2) drh[k / 3] += dk[k]; else drw[k % 3] += dk[k];
            }
            if (intile) {                                   // scatter-form parameter gradients at the tile pixels
                dq[c] = dqc;
                pg[O_BQ + c] += dqc;
                pg[O_BK + c] += sdk;                        // includes the padded taps (they carry the bias)
                pg[O_BV + c] += g;                          // sum_t a_t g = g
#pragma unroll
                for (int i = 0; i < C; ++i) pg[O_WQ + c * C + i] = fmaf(dqc, xs[pc * C + i], pg[O_WQ + c * C + i]);
#pragma unroll
                for (int i = 0; i < 3; ++i) { pg[O_RH + i] += drh[i]; pg[O_RW + i] += drw[i]; }
            }
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) { ad[t * ADS + k] = dv[k]; ad[t * ADS + 9 + k] = dk[k]; }
        __syncthreads();
        // ---- gather: tap t of pixel p = p' - o_t lands on p'
        if (intile) {
            float sv = 0.f, sk = 0.f;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int pe = (ey + 1 - dy) * EW + ex + 1 - dx;      // evaluated-region index of p
                    sv += ad[pe * ADS + dy * 3 + dx];
                    sk += ad[pe * ADS + 9 + dy * 3 + dx];
                }
            dV[c] = sv;
            dK[c] = sk;
        }
        __syncthreads();
    }

    // ---- dx = relu' . (Wq^T dq + Wk^T dK + Wv^T dV);  gather-form weight gradients of the k / v convolutions; dres
    if (intile) {
#pragma unroll
        for (int i = 0; i < C; ++i) {
            const float r = xs[pc * C + i];
            float d = 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                d = fmaf(P.wq[c][i], dq[c], d);
                d = fmaf(P.wk[c][i], dK[c], d);
                d = fmaf(P.wv[c][i], dV[c], d);
                pg[O_WK + c * C + i] = fmaf(dK[c], r, pg[O_WK + c * C + i]);
                pg[O_WV + c * C + i] = fmaf(dV[c], r, pg[O_WV + c * C + i]);
            }
            if (a.relu_in && !(r > 0.f)) d = 0.f;               // r = relu(x): r > 0 <=> x > 0
            d += dxadd[i];
            const_cast<float*>(a.dx.ptr[i])[attn_off(a.dx, i, b, py, px, a.H, a.W)] = d;
            if (a.dres.ptr[0]) {
                const bool dead = a.relu_res && !(resv[i] > 0.f);
                const_cast<float*>(a.dres.ptr[i])[attn_off(a.dres, i, b, py, px, a.H, a.W)] = dead ? 0.f : gpix[i];
            }
        }
    }

    // ---- block reduction of the parameter partials: wave tree, then the four waves in order
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const float v = wave_sum(pg[i]);
        if (lane == 0) red[wave][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < NP) {
        const int bid = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        a.partial[(size_t)bid * NP + threadIdx.x] =
            ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
    }
}

// sum of the per-block rows, fixed order: one block of 256 threads per parameter entry
__global__ __launch_bounds__(256) void attn_param_reduce_kernel(const float* __restrict__ partial, float* __restrict__ out, int nblocks,
                                                               int np) {
    __shared__ float sm[256];
    const int e = blockIdx.x;
    float s = 0.f;
    for (int i = threadIdx.x; i < nblocks; i += 256) s += partial[(size_t)i * np + e];
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) sm[threadIdx.x] += sm[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[e] = sm[0];
}

}  // namespace dc

using namespace dc;

static bool attn_map_ok(const dc_attn_map* m, int C, int H, int W, bool optional) {
    if (!m) return optional;
    if (!m->ptr[0]) return optional;
    for (int c = 0; c < C; ++c) {
        if (!m->ptr[c] || m->batch_stride[c] < 0) return false;
        if (m->mode[c] != DC_ATTN_PLAIN && m->mode[c] != DC_ATTN_PIXEL_SHUFFLE2) return false;
        if (m->mode[c] == DC_ATTN_PIXEL_SHUFFLE2 && ((H & 1) || (W & 1))) return false;
    }
    return true;
}
static void attn_map_copy(dc_attn_map& dst, const dc_attn_map* src) {
    if (src) dst = *src; else dst = dc_attn_map{};
}
static bool attn_ok(const dc_attn_params* p, int B, int C, int H, int W) {
    return p && p->wq && p->bq && p->wk && p->bk && p->wv && p->bv && p->rel_h && p->rel_w && B > 0 && (C == 2 || C == 4) &&
           H > 0 && W > 0 && (size_t)B * C * H * W < (1ull << 31);
}
static void attn_params(AttnArgs& a, const dc_attn_params* p) {
    a.wq = p->wq; a.bq = p->bq; a.wk = p->wk; a.bk = p->bk; a.wv = p->wv; a.bv = p->bv; a.rel_h = p->rel_h; a.rel_w = p->rel_w;
}

extern "C" int dc_attnconv_fwd(const dc_attn_map* x, const dc_attn_params* p, const dc_attn_map* res, float* y, int B, int C, int H,
                               int W, int relu_in, int relu_res, void* stream) {
    if (!y || !attn_ok(p, B, C, H, W) || !attn_map_ok(x, C, H, W, false) || !attn_map_ok(res, C, H, W, true)) return DC_EINVAL;
    AttnArgs a{};
    attn_map_copy(a.x, x); attn_map_copy(a.res, res);
    a.y = y; a.B = B; a.H = H; a.W = W; a.relu_in = relu_in; a.relu_res = relu_res;
    attn_params(a, p);
    const dim3 grid(ceil_div(W, AF_TW), ceil_div(H, AF_TH), B);
    if (C == 2) hipLaunchKernelGGL(attn_fwd_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(attn_fwd_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, a);
    DC_CHECK_LAUNCH();
    return DC_OK;
}

extern "C" int dc_attnconv_param_count(int C) { return (C == 2 || C == 4) ? 3 * (C * C + C) + 6 : 0; }

extern "C" size_t dc_attnconv_bwd_workspace(int B, int C, int H, int W) {
    if (B <= 0 || (C != 2 && C != 4) || H <= 0 || W <= 0) return 0;
    return (size_t)ceil_div(W, AB_TW) * ceil_div(H, AB_TH) * B * dc_attnconv_param_count(C) * sizeof(float);
}

extern "C" int dc_attnconv_bwd(const dc_attn_map* x, const dc_attn_params* p, const dc_attn_map* res, const float* gy,
                               const dc_attn_map* dx, const float* dx_add, const dc_attn_map* dres, float* dparams, void* ws, int B,
                               int C, int H, int W, int relu_in, int relu_res, void* stream) {
    if (!gy || !dparams || !ws || !attn_ok(p, B, C, H, W) || !attn_map_ok(x, C, H, W, false) || !attn_map_ok(dx, C, H, W, false) ||
        !attn_map_ok(res, C, H, W, true) || !attn_map_ok(dres, C, H, W, true))
        return DC_EINVAL;
    if (dres && dres->ptr[0] && relu_res && !(res && res->ptr[0])) return DC_EINVAL;
    AttnArgs a{};
    attn_map_copy(a.x, x); attn_map_copy(a.res, res); attn_map_copy(a.dx, dx); attn_map_copy(a.dres, dres);
    a.gy = gy; a.dx_add = dx_add; a.partial = (float*)ws;
    a.B = B; a.H = H; a.W = W; a.relu_in = relu_in; a.relu_res = relu_res;
    attn_params(a, p);
    const dim3 grid(ceil_div(W, AB_TW), ceil_div(H, AB_TH), B);
    const int nblocks = grid.x * grid.y * grid.z, np = dc_attnconv_param_count(C);
    hipStream_t st = (hipStream_t)stream;
    if (C == 2) hipLaunchKernelGGL(attn_bwd_kernel<2>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(attn_bwd_kernel<4>, grid, dim3(256), 0, st, a);
    DC_CHECK_LAUNCH();
    hipLaunchKernelGGL(attn_param_reduce_kernel, dim3(np), dim3(256), 0, st, (const float*)ws, dparams, nblocks, np);
    DC_CHECK_LAUNCH();
    return DC_OK;
}
