// Micro-benchmark: issue cost of v_fma_f32 / v_pk_fma_f32 / DPP wave shifts on gfx950 at 1..8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define N_IT 4096
template <int MODE>
__global__ void k(float* out, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7};
    f2 pa = {a, a}, pb = {b, b};
    for (int i = 0; i < N_IT; ++i) {
        if (MODE == 0) {   // 8 independent v_fma_f32
            x0 = fmaf(x0, a, b); x1 = fmaf(x1, a, b); x2 = fmaf(x2, a, b); x3 = fmaf(x3, a, b);
            x4 = fmaf(x4, a, b); x5 = fmaf(x5, a, b); x6 = fmaf(x6, a, b); x7 = fmaf(x7, a, b);
        } else if (MODE == 1) {   // 4 independent v_pk_fma_f32 (= 8 fma lanes-worth)
            asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa), "v"(pb));
        } else if (MODE == 2) {   // 8 DPP wave_shr moves
            asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %2, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %2, %3 wave_shl:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %0 wave_shl:1 row_mask:0xf bank_mask:0xf"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
        } else if (MODE == 3) {   // 8 row_shr DPP moves
            asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %0, %1 row_shl:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_shl:1 row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %2, %3 row_shl:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %0 row_shl:1 row_mask:0xf bank_mask:0xf"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
        } else if (MODE == 4) {   // 8 ds_bpermute
            int idx = ((threadIdx.x + 1) & 63) * 4;
            x0 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(idx, __builtin_bit_cast(int, x0)));
            x1 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(idx, __builtin_bit_cast(int, x1)));
            x2 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(idx, __builtin_bit_cast(int, x2)));
            x3 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(idx, __builtin_bit_cast(int, x3)));
            x4 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(idx, __builtin_bit_cast(int, x4)));
            x5 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(idx, __builtin_bit_cast(int, x5)));
            x6 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(idx, __builtin_bit_cast(int, x6)));
            x7 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(idx, __builtin_bit_cast(int, x7)));
        } else if (MODE == 5) {   // dependent chain of 8 v_fma
            x0 = fmaf(x0, a, b); x0 = fmaf(x0, a, b); x0 = fmaf(x0, a, b); x0 = fmaf(x0, a, b);
            x0 = fmaf(x0, a, b); x0 = fmaf(x0, a, b); x0 = fmaf(x0, a, b); x0 = fmaf(x0, a, b);
        } else if (MODE == 6) {   // 8 v_rcp_f32
            x0 = __builtin_amdgcn_rcpf(x0); x1 = __builtin_amdgcn_rcpf(x1); x2 = __builtin_amdgcn_rcpf(x2); x3 = __builtin_amdgcn_rcpf(x3);
            x4 = __builtin_amdgcn_rcpf(x4); x5 = __builtin_amdgcn_rcpf(x5); x6 = __builtin_amdgcn_rcpf(x6); x7 = __builtin_amdgcn_rcpf(x7);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}
template <int MODE>
void run(const char* name, float* d) {
    for (int wps = 1; wps <= 8; wps *= 2) {   // waves per SIMD
        int threads = 256 * wps;   // 4 waves per 256 threads -> 1 wave/SIMD per 256 threads
        if (threads > 1024) threads = 1024;
        int blocks_per_cu = (256 * wps) / threads;
        int blocks = 256 * blocks_per_cu;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        k<MODE><<<blocks, threads>>>(d, 1.0001f, 0.5f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) k<MODE><<<blocks, threads>>>(d, 1.0001f, 0.5f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double t = ms / 5 * 1e-3;
        double instr_per_simd = (double)N_IT * 8 * wps;   // wave-instructions (8 per iteration; pk counted as 4x2)
        double cyc = t * 2.4e9 / instr_per_simd;
        printf("%-22s waves/SIMD=%d  %.1f us  -> %.2f cycles@2.4GHz per wave-op (pk: per half)\n", name, wps, t * 1e6, cyc);
    }
}
int main() {
    float* d; hipMalloc(&d, 256 * 8 * 1024 * 4);
    run<0>("v_fma_f32 indep", d);
    run<1>("v_pk_fma_f32 indep", d);
    run<2>("dpp wave_shr/shl", d);
    run<3>("dpp row_shr/shl", d);
    run<4>("ds_bpermute", d);
    run<5>("v_fma_f32 dependent", d);
    run<6>("v_rcp_f32", d);
    return 0;
}
