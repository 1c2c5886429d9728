// lat_bench.hip -- scratch: issue cost / latency of the instructions on the Cholesky pivot chain (one wave, gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int N = 2048;
template <int CH> __global__ void k_fma(long long* out, double* sink, double m) {
    double x[CH];
    for (int c = 0; c < CH; ++c) x[c] = threadIdx.x + c;
    const long long t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int c = 0; c < CH; ++c) x[c] = fma(x[c], m, 1e-9);
    }
    const long long t1 = clock64();
    double s = 0; for (int c = 0; c < CH; ++c) s += x[c];
    if (threadIdx.x == 0) out[0] = t1 - t0;
    if (s == 1.2345) sink[0] = s;
}
__global__ void k_rsq(long long* out, double* sink) {
    double x = 1.0 + threadIdx.x;
    const long long t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < N; ++i) x = __builtin_amdgcn_rsq(x) + 1.0;
    const long long t1 = clock64();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    if (x == 1.2345) sink[0] = x;
}
__global__ void k_readlane_chain(long long* out, double* sink) {
    double x = 1.0 + threadIdx.x;
    const long long t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < N; ++i) {
        const int l = i & 63;
        const int lo = __builtin_amdgcn_readlane(__double2loint(x), l), hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
        x = fma(x, __hiloint2double(hi, lo), 1e-9);
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    if (x == 1.2345) sink[0] = x;
}
template <int CH> __global__ void k_readlane_tp(long long* out, double* sink) {
    double x[CH];
    for (int c = 0; c < CH; ++c) x[c] = threadIdx.x + c;
    double s = 1.0 + threadIdx.x;
    const long long t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int l = (i + c) & 63;
            const int lo = __builtin_amdgcn_readlane(__double2loint(s), l), hi = __builtin_amdgcn_readlane(__double2hiint(s), l);
            x[c] = fma(s, __hiloint2double(hi, lo), x[c]);
        }
    }
    const long long t1 = clock64();
    double r = 0; for (int c = 0; c < CH; ++c) r += x[c];
    if (threadIdx.x == 0) out[0] = t1 - t0;
    if (r == 1.2345) sink[0] = r;
}
template <int L> __device__ __forceinline__ double bcast16(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x150 + L, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x150 + L, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int L> __device__ __forceinline__ double bcast16_b64(double v) {
    double r;
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(L));
    return r;
}
template <bool B64> __global__ void k_dpp_chain(long long* out, double* sink) {
    double x = 1.0 + threadIdx.x;
    const long long t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < N; ++i) x = fma(x, B64 ? bcast16_b64<3>(x) : bcast16<3>(x), 1e-9);
    const long long t1 = clock64();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    if (x == 1.2345) sink[0] = x;
}
template <bool B64> __global__ void k_dpp_tp(long long* out, double* sink) {
    double x[8];
    for (int c = 0; c < 8; ++c) x[c] = threadIdx.x + c;
    double s = 1.0 + threadIdx.x;
    const long long t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < N; ++i) {
        x[0] = fma(s, B64 ? bcast16_b64<0>(s) : bcast16<0>(s), x[0]); x[1] = fma(s, B64 ? bcast16_b64<1>(s) : bcast16<1>(s), x[1]);
        x[2] = fma(s, B64 ? bcast16_b64<2>(s) : bcast16<2>(s), x[2]); x[3] = fma(s, B64 ? bcast16_b64<3>(s) : bcast16<3>(s), x[3]);
        x[4] = fma(s, B64 ? bcast16_b64<4>(s) : bcast16<4>(s), x[4]); x[5] = fma(s, B64 ? bcast16_b64<5>(s) : bcast16<5>(s), x[5]);
        x[6] = fma(s, B64 ? bcast16_b64<6>(s) : bcast16<6>(s), x[6]); x[7] = fma(s, B64 ? bcast16_b64<7>(s) : bcast16<7>(s), x[7]);
        s += 1e-12;
    }
    const long long t1 = clock64();
    double r = 0; for (int c = 0; c < 8; ++c) r += x[c];
    if (threadIdx.x == 0) out[0] = t1 - t0;
    if (r == 1.2345) sink[0] = r;
}
__global__ void k_rsq_acc(double* out) {
    // worst relative error of the hardware rsq estimate, after one and after two Newton steps, over a sweep of arguments
    double w0 = 0, w1 = 0, w2 = 0;
    for (int i = 0; i < 20000; ++i) {
        const double x = (1.0 + (threadIdx.x * 20000 + i) * (3.0 / (64 * 20000))) * (i % 3 == 0 ? 1e-7 : i % 3 == 1 ? 1.0 : 1e9);
        const double ex = 1.0 / sqrt(x);
        double y = __builtin_amdgcn_rsq(x);
        w0 = fmax(w0, fabs(y - ex) / ex);
        double e = fma(-x * y, y, 1.0); y = fma(0.5 * y, e, y);
        w1 = fmax(w1, fabs(y - ex) / ex);
        e = fma(-x * y, y, 1.0); y = fma(0.5 * y, e, y);
        w2 = fmax(w2, fabs(y - ex) / ex);
    }
    out[threadIdx.x * 3 + 0] = w0; out[threadIdx.x * 3 + 1] = w1; out[threadIdx.x * 3 + 2] = w2;
}
int main() {
    long long* d; double* s; CK(hipMalloc(&d, 64)); CK(hipMalloc(&s, 64));
    long long h;
#define RUN(name, call, per) do { call; call;  CK(hipDeviceSynchronize()); CK(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost)); printf("%-44s %7.1f cycles per %s\n", name, (double)h / N / (per), "op"); } while (0)
    RUN("dependent DFMA (1 chain)", hipLaunchKernelGGL(k_fma<1>, dim3(1), dim3(64), 0, 0, d, s, 1.0000001), 1);
    RUN("DFMA, 2 chains", hipLaunchKernelGGL(k_fma<2>, dim3(1), dim3(64), 0, 0, d, s, 1.0000001), 2);
    RUN("DFMA, 4 chains", hipLaunchKernelGGL(k_fma<4>, dim3(1), dim3(64), 0, 0, d, s, 1.0000001), 4);
    RUN("DFMA, 8 chains", hipLaunchKernelGGL(k_fma<8>, dim3(1), dim3(64), 0, 0, d, s, 1.0000001), 8);
    RUN("DFMA, 16 chains", hipLaunchKernelGGL(k_fma<16>, dim3(1), dim3(64), 0, 0, d, s, 1.0000001), 16);
    RUN("dependent rsq_f64 + add", hipLaunchKernelGGL(k_rsq, dim3(1), dim3(64), 0, 0, d, s), 1);
    RUN("dependent readlane x2 + DFMA", hipLaunchKernelGGL(k_readlane_chain, dim3(1), dim3(64), 0, 0, d, s), 1);
    RUN("readlane x2 + DFMA, 8 independent", hipLaunchKernelGGL(k_readlane_tp<8>, dim3(1), dim3(64), 0, 0, d, s), 8);
    RUN("readlane x2 + DFMA, 16 independent", hipLaunchKernelGGL(k_readlane_tp<16>, dim3(1), dim3(64), 0, 0, d, s), 16);
    RUN("dependent 2x v_mov_b32_dpp + DFMA", hipLaunchKernelGGL(k_dpp_chain<false>, dim3(1), dim3(64), 0, 0, d, s), 1);
    RUN("dependent v_mov_b64_dpp + DFMA", hipLaunchKernelGGL(k_dpp_chain<true>, dim3(1), dim3(64), 0, 0, d, s), 1);
    RUN("2x v_mov_b32_dpp + DFMA, 8 independent", hipLaunchKernelGGL(k_dpp_tp<false>, dim3(1), dim3(64), 0, 0, d, s), 8);
    RUN("v_mov_b64_dpp + DFMA, 8 independent", hipLaunchKernelGGL(k_dpp_tp<true>, dim3(1), dim3(64), 0, 0, d, s), 8);
    {
        double* acc; CK(hipMalloc(&acc, 64 * 3 * 8)); double ha[192];
        hipLaunchKernelGGL(k_rsq_acc, dim3(1), dim3(64), 0, 0, acc); CK(hipMemcpy(ha, acc, sizeof(ha), hipMemcpyDeviceToHost));
        double w[3] = { 0, 0, 0 }; for (int i = 0; i < 64; ++i) for (int k = 0; k < 3; ++k) w[k] = ha[i * 3 + k] > w[k] ? ha[i * 3 + k] : w[k];
        printf("v_rsq_f64 relative error: estimate %.2e, one Newton step %.2e, two %.2e\n", w[0], w[1], w[2]);
    }
    return 0;
}
