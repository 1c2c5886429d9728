// microbench.hip -- scratch measurements that inform the kernel design (not part of the product).
//   1. shader clock vs wall clock for short kernels (DVFS state during a latency-bound workload)
//   2. dependent global-load latency (pointer chase) for L2-resident / MALL-resident / HBM footprints
//   3. fp64 atomic-add throughput: LDS (ds_add_f64) and global, random addresses
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void k_clock(long long* out, int iters) {
    const long long w0 = wall_clock64(), c0 = clock64();
    double x = threadIdx.x;
    for (int i = 0; i < iters; ++i) x = fma(x, 1.0000001, 1e-9);
    const long long w1 = wall_clock64(), c1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = w1 - w0; out[1] = c1 - c0; out[2] = (long long)x; }
}

__global__ void k_chase(const int* __restrict__ next, int steps, long long* out, int stride_threads, int n) {
    int p = (int)(((long long)(blockIdx.x * blockDim.x + threadIdx.x) * stride_threads) % n) & ~31;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < steps; ++i) p = next[p];
    const long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = p; }
    if (p == -1) out[3] = p;
}

__global__ void k_lds_atomic(double* sink, const int* __restrict__ idx, int n, int reps) {
    extern __shared__ double acc[];
    for (int e = threadIdx.x; e < n; e += blockDim.x) acc[e] = 0.0;
    __syncthreads();
    int j = idx[blockIdx.x * blockDim.x + threadIdx.x];
    for (int r = 0; r < reps; ++r) { atomicAdd(&acc[j], 1.0); j = (j * 1103515245 + 12345) & 0x7fffffff; j %= n; }
    __syncthreads();
    if (threadIdx.x == 0) sink[blockIdx.x] = acc[0];
}

__global__ void k_glb_atomic(double* target, int n, int reps) {
    int j = (blockIdx.x * blockDim.x + threadIdx.x) * 7919;
    for (int r = 0; r < reps; ++r) { j = (j * 1103515245 + 12345) & 0x7fffffff; atomicAdd(&target[j % n], 1.0); }
}

__global__ void k_burn(double* out, int iters) {
    double x = threadIdx.x;
    for (int i = 0; i < iters; ++i) x = fma(x, 1.0000001, 1e-9);
    if (x == 12345.678) out[0] = x;
}

int main() {
    long long* d_out; CK(hipMalloc(&d_out, 64));
    long long h[8];
    double* d_sink; CK(hipMalloc(&d_sink, 8 * 65536));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // ---- 1. clocks: cold short kernel, then after a 20 ms burn
    for (int phase = 0; phase < 3; ++phase) {
        if (phase == 1) { hipLaunchKernelGGL(k_burn, dim3(4096), dim3(256), 0, 0, d_sink, 400000); CK(hipDeviceSynchronize()); }
        if (phase == 2) { CK(hipDeviceSynchronize()); struct timespec ts = {0, 50000000}; nanosleep(&ts, nullptr); }
        hipLaunchKernelGGL(k_clock, dim3(1), dim3(64), 0, 0, d_out, 20000);
        CK(hipMemcpy(h, d_out, 24, hipMemcpyDeviceToHost));
        printf("clock phase %d (0 cold, 1 right after burn, 2 after 50 ms idle): wall ticks %lld (100 MHz) shader cycles %lld -> %.0f MHz, %.1f cyc per dependent DFMA\n",
               phase, h[0], h[1], 100.0 * h[1] / h[0], (double)h[1] / 20000);
    }
    // ---- 2. pointer chase
    for (size_t bytes : { (size_t)1 << 20, (size_t)16 << 20, (size_t)128 << 20, (size_t)1024 << 20 }) {
        const size_t n = bytes / 4;
        std::vector<int> perm(n / 32);    // one hop per 128-byte line
        std::iota(perm.begin(), perm.end(), 0);
        std::mt19937 rng(1);
        std::shuffle(perm.begin(), perm.end(), rng);
        std::vector<int> next(n, 0);
        for (size_t i = 0; i < perm.size(); ++i) next[(size_t)perm[i] * 32] = perm[(i + 1) % perm.size()] * 32;
        int* d_next; CK(hipMalloc(&d_next, bytes));
        CK(hipMemcpy(d_next, next.data(), bytes, hipMemcpyHostToDevice));
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k_chase, dim3(1), dim3(1), 0, 0, d_next, 2000, d_out, 0, (int)n);
            CK(hipMemcpy(h, d_out, 24, hipMemcpyDeviceToHost));
        }
        printf("chase footprint %5zu MB: %.0f shader cycles / %.1f ns per dependent load (1 lane)\n", bytes >> 20, (double)h[0] / 2000, 10.0 * h[1] / 2000);
        // loaded machine: every CU chasing
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_chase, dim3(2048), dim3(256), 0, 0, d_next, 200, d_out, 32 * 7, (int)n);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(h, d_out, 24, hipMemcpyDeviceToHost));
        printf("      loaded (2048x256 lanes chasing): %.1f ns per dependent load seen by lane 0, kernel %.1f us, %.1f G loads/s\n",
               10.0 * h[1] / 200, 1e3 * ms, 2048.0 * 256 * 200 / (ms * 1e6));
        CK(hipFree(d_next));
    }
    // ---- 3. atomics
    {
        const int n = 7200;   // doubles (57.6 KB) like a 6 x 1200 row block
        std::vector<int> idx(1024 * 256);
        for (size_t i = 0; i < idx.size(); ++i) idx[i] = rand() % n;
        int* d_idx; CK(hipMalloc(&d_idx, idx.size() * 4)); CK(hipMemcpy(d_idx, idx.data(), idx.size() * 4, hipMemcpyHostToDevice));
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_lds_atomic, dim3(1024), dim3(256), n * 8, 0, d_sink, d_idx, n, 1000);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("LDS ds_add_f64 random addresses: %.1f G atomics/s chip (%.2f per clk per CU at 2.1 GHz)\n", 1024.0 * 256 * 1000 / (ms * 1e6), 1024.0 * 256 * 1000 / (ms * 1e-3) / 256 / 2.1e9);
        }
        double* d_t; const int ng = 1442401; CK(hipMalloc(&d_t, (size_t)ng * 8)); CK(hipMemset(d_t, 0, (size_t)ng * 8));
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_glb_atomic, dim3(2048), dim3(256), 0, 0, d_t, ng, 200);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("global atomicAdd(double) random over 11.5 MB: %.1f G atomics/s\n", 2048.0 * 256 * 200 / (ms * 1e6));
        }
    }
    // ---- 4. launch boundary: 200 trivial kernels back to back
    {
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_clock, dim3(1), dim3(64), 0, 0, d_out, 1);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("200 trivial dependent launches: %.2f us per launch (device timeline)\n", 1e3 * ms / 200);
    }
    return 0;
}
