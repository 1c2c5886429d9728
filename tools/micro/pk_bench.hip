// pk_bench.hip -- VALU issue cost on gfx950 of v_fma_f32, v_pk_fma_f32, v_fma_f64, v_cndmask, v_mov (dependent-free streams, one..eight waves per SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a2}, p5 = {a3, a4}, p6 = {a5, a6}, p7 = {a7, a0};
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;
    const float m = 1.0000001f, c = 1e-9f;
    const v2f pm = {m, m}, pc = {c, c};
    const double dm = 1.0000001, dc = 1e-9;
    const unsigned long long mask = 0x5555aaaa3333ccccull ^ (unsigned long long)iters;
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        } else if (KIND == 1) {
            asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                         "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pm), "v"(pc));
        } else if (KIND == 2) {
            asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                         "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(dm), "v"(dc));
        } else if (KIND == 3) {
            asm volatile("v_cndmask_b32 %0, %1, %0, vcc\n v_cndmask_b32 %1, %2, %1, vcc\n v_cndmask_b32 %2, %3, %2, vcc\n v_cndmask_b32 %3, %4, %3, vcc\n"
                         "v_cndmask_b32 %4, %5, %4, vcc\n v_cndmask_b32 %5, %6, %5, vcc\n v_cndmask_b32 %6, %7, %6, vcc\n v_cndmask_b32 %7, %0, %7, vcc"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "vcc");
        } else if (KIND == 5) {
            asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"
                         "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        } else if (KIND == 6) {
            asm volatile("v_cndmask_b32_e64 %0, %1, %0, %8\n v_cndmask_b32_e64 %1, %2, %1, %8\n v_cndmask_b32_e64 %2, %3, %2, %8\n v_cndmask_b32_e64 %3, %4, %3, %8\n"
                         "v_cndmask_b32_e64 %4, %5, %4, %8\n v_cndmask_b32_e64 %5, %6, %5, %8\n v_cndmask_b32_e64 %6, %7, %6, %8\n v_cndmask_b32_e64 %7, %0, %7, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(mask));
        } else if (KIND == 7) {
            asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
        } else if (KIND == 8) {
            asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %2, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %4, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %6, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
        } else if (KIND == 9) {
            a0 = __shfl_xor(a0, 32, 64); a1 = __shfl_xor(a1, 16, 64); a2 = __shfl_xor(a2, 32, 64); a3 = __shfl_xor(a3, 16, 64);
            a4 = __shfl_xor(a4, 32, 64); a5 = __shfl_xor(a5, 16, 64); a6 = __shfl_xor(a6, 32, 64); a7 = __shfl_xor(a7, 16, 64);
        } else if (KIND == 10) {
            asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                         "v_pk_add_f32 %4, %4, %9\n v_pk_add_f32 %5, %5, %9\n v_pk_add_f32 %6, %6, %9\n v_pk_add_f32 %7, %7, %9"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pm), "v"(pc));
        } else if (KIND == 4) {
            asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                         "v_add_f32 %4, %4, %9\n v_add_f32 %5, %5, %9\n v_add_f32 %6, %6, %9\n v_add_f32 %7, %7, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
}
template <int KIND> int run(const char* name, float* d, int wg_per_cu) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000, blocks = 256 * wg_per_cu;
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 100);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    // waves per SIMD = wg_per_cu (4 waves per WG over 4 SIMDs); instructions per SIMD = wg_per_cu * 8 * iters
    const double cyc = ms * 1e-3 * 2.4e9 / ((double)wg_per_cu * 8 * iters);
    printf("%-14s %d wave(s)/SIMD: %.3f ms -> %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, wg_per_cu, ms, cyc);
    return 0;
}
int main() {
    float* d; CK(hipMalloc(&d, 256 * 8 * 256 * sizeof(float)));
    for (int w : {1, 4, 8}) {
        run<0>("v_fma_f32", d, w); run<5>("v_fmac_f32", d, w); run<1>("v_pk_fma_f32", d, w); run<10>("v_pk_mul/add", d, w); run<2>("v_fma_f64", d, w);
        run<4>("v_mul/add_f32", d, w); run<3>("cndmask vcc", d, w); run<6>("cndmask sgpr", d, w); run<7>("v_mov_b32", d, w); run<8>("v_mov_dpp", d, w); run<9>("shfl_xor", d, w);
    }
    return 0;
}
