// chol_tile_bench.hip -- scratch: correctness and timing of the one-workgroup 64x64 factor + inverse (csrc/chol_tile.h).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I sfm-toy-library_amd/csrc tools/micro/chol_tile_bench.hip -o /tmp/chol_tile_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <random>
__device__ long long g_dbg[4];
#define CT_DEBUG_CLK g_dbg
#include "chol_tile.h"
using namespace sfmba;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ __launch_bounds__(256) void k_factor(const double* __restrict__ A, double* __restrict__ L, double* __restrict__ Einv, long long* clk, int* info) {
    extern __shared__ double sm[];
    double* U = sm; double* X = U + CT_NB * CT_LDT; double* ED = X + CT_NB * CT_LDP; double* V = ED + (CT_NB / CT_PB + 1) * CT_PB * CT_LDP;
    const long long t0 = clock64();
    const double* Ab = A + (size_t)blockIdx.x * CT_NB * CT_NB;
    for (int idx = threadIdx.x; idx < CT_NB * CT_NB; idx += 256) { const int r = idx % CT_NB, c = idx / CT_NB; U[r * CT_LDT + c] = Ab[r + c * CT_NB]; }
    __syncthreads();
    const long long t1 = clock64();
    const int bad = chol_tile_factor(U, X, ED, V, 0, CT_NB);
    const long long t2 = clock64();
    for (int idx = threadIdx.x; idx < CT_NB * CT_NB; idx += 256) {
        const int r = idx % CT_NB, c = idx / CT_NB, q = r >> 4, cb = c >> 4;
        L[(size_t)blockIdx.x * CT_NB * CT_NB + r + c * CT_NB] = r >= c ? X[c] * U[r * CT_LDT + c] : 0.0;
        Einv[(size_t)blockIdx.x * CT_NB * CT_NB + r + c * CT_NB] = X[c] * (q < cb ? U[r * CT_LDT + c] : q == cb ? ED[(q * CT_PB + (r & 15)) * CT_LDP + (c & 15)] : 0.0);
    }
    const long long t3 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = t2 - t1; clk[2] = t3 - t2; if (bad) *info = bad; }
}

int main() {
    const int n = CT_NB, nt = 256;
    std::mt19937_64 rng(1);
    std::normal_distribution<double> g;
    std::vector<double> A((size_t)nt * n * n), B(n * n);
    for (int t = 0; t < nt; ++t) {
        for (auto& v : B) v = g(rng);
        for (int r = 0; r < n; ++r) for (int c = 0; c < n; ++c) { double s = r == c ? 1e-3 * n : 0.0; for (int k = 0; k < n; ++k) s += B[r * n + k] * B[c * n + k]; A[(size_t)t * n * n + r + c * n] = s; }
    }
    double *dA, *dL, *dE; long long* dclk; int* dinfo;
    CK(hipMalloc(&dA, A.size() * 8)); CK(hipMalloc(&dL, A.size() * 8)); CK(hipMalloc(&dE, A.size() * 8)); CK(hipMalloc(&dclk, 64)); CK(hipMalloc(&dinfo, 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice)); CK(hipMemset(dinfo, 0, 4));
    const size_t lds = sizeof(double) * CT_LDS_DOUBLES;
    CK(hipFuncSetAttribute((const void*)k_factor, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_factor, dim3(nt), dim3(256), lds, 0, dA, dL, dE, dclk, dinfo);
    CK(hipDeviceSynchronize());
    std::vector<double> L(A.size()), E(A.size());
    CK(hipMemcpy(L.data(), dL, L.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(E.data(), dE, E.size() * 8, hipMemcpyDeviceToHost));
    double worstA = 0, worstI = 0, worstLow = 0;
    for (int t = 0; t < nt; t += 37) {
        const double* a = &A[(size_t)t * n * n]; const double* l = &L[(size_t)t * n * n]; const double* e = &E[(size_t)t * n * n];
        double amax = 0; for (int i = 0; i < n * n; ++i) amax = std::max(amax, std::fabs(a[i]));
        for (int r = 0; r < n; ++r) for (int c = 0; c <= r; ++c) {
            double s = 0; for (int k = 0; k <= c; ++k) s += l[r + k * n] * l[c + k * n];
            worstA = std::max(worstA, std::fabs(s - a[r + c * n]) / amax);
        }
        // E = L^-T  <=>  E^T L^T... check L^T E^T? use  sum_k E[k][r]... (L^-T)(r,c): L^T E = I -> sum_k L[k][r] E[k][c] = delta
        for (int r = 0; r < n; ++r) for (int c = 0; c < n; ++c) {
            double s = 0; for (int k = 0; k < n; ++k) s += l[k + r * n] * e[k + c * n];
            worstI = std::max(worstI, std::fabs(s - (r == c ? 1.0 : 0.0)));
            if (r > c) worstLow = std::max(worstLow, std::fabs(e[r + c * n]));
        }
    }
    int info; CK(hipMemcpy(&info, dinfo, 4, hipMemcpyDeviceToHost));
    printf("info %d  |L L^T - A|/|A| %.2e   |L^T E - I| %.2e   |E below diagonal| %.2e\n", info, worstA, worstI, worstLow);
    long long clk[3]; CK(hipMemcpy(clk, dclk, 24, hipMemcpyDeviceToHost));
    printf("cycles (wg 0 of %d): load %lld  factor %lld  store %lld   (100 MHz wall? no: clock64 = shader cycles)\n", nt, clk[0], clk[1], clk[2]);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int wgs : { 1, 16, 256 }) {
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k_factor, dim3(wgs), dim3(256), lds, 0, dA, dL, dE, dclk, dinfo);
        CK(hipEventRecord(e0, 0));
        const int reps = 200;
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_factor, dim3(wgs), dim3(256), lds, 0, dA, dL, dE, dclk, dinfo);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(clk, dclk, 24, hipMemcpyDeviceToHost));
        { long long hd[4]; CK(hipMemcpyFromSymbol(hd, HIP_SYMBOL(g_dbg), 32)); printf("    inside factor (warm): panel loads %lld  sweeps %lld  stores+sync %lld  trailing updates %lld cycles\n", hd[0], hd[1], hd[3], hd[2]); }
        printf("%3d workgroups: %.2f us per launch (back to back), in-kernel cycles load %lld factor %lld store %lld\n", wgs, 1e3 * ms / reps, clk[0], clk[1], clk[2]);
    }
    return 0;
}
