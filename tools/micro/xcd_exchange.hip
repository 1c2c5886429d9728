// xcd_exchange.hip -- what one CG iteration's exchange costs when the team of workgroups sits on ONE XCD (or two, four, eight).
// Team = the workgroups with blockIdx % STRIDE == 0 of a 256-workgroup launch (dispatch order: workgroup i -> XCD i % 8), the others
// return at once.  Every team member publishes its slice of a d-vector as 8-byte {tag, half} granules (agent-scope relaxed stores) and
// gathers the whole vector (agent-scope relaxed loads) until every tag carries the iteration -- the exchange of k_pcg_persistent
// (dense_solver.hip).  Also: load time of the team's rows of a d x d fp64 matrix into registers (once per solve).
//   hipcc --offload-arch=gfx950 -O3 -o xcd_exchange xcd_exchange.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int D = 1201, LD = 1280;

template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_exchange(unsigned long long* gran, int stride, int iters, unsigned epoch0, double* sink, int* xcc_out, long long* clk_out) {
    if (blockIdx.x % stride != 0) return;
    const int team = (int)gridDim.x / stride, rank = (int)blockIdx.x / stride;
    const int rows = (D + team - 1) / team, row0 = rank * rows, row1 = min(D, row0 + rows);
    const int tid = threadIdx.x;
    if (tid == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc_out[rank] = (int)(xcc & 0xf);
    }
    double acc = 0.0;
    const long long t0 = clock64();
    for (int it = 1; it <= iters; ++it) {
        const unsigned tag = epoch0 + (unsigned)it;
        unsigned long long* gb = gran + (size_t)(it & 1) * 2 * LD;
        for (int g = 2 * row0 + tid; g < 2 * row1; g += THREADS)
            __hip_atomic_store(gb + g, ((unsigned long long)tag << 32) | (unsigned)(g + it), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        for (;;) {
            bool ok = true;
            unsigned v = 0;
            for (int g = tid; g < 2 * D; g += THREADS) {
                const unsigned long long x = __hip_atomic_load(gb + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                v += (unsigned)x;
                ok &= (unsigned)(x >> 32) == tag;
            }
            if (__syncthreads_and(ok)) { acc += (double)v; break; }
            if (++spins > (1u << 20)) { acc = -1.0; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (acc < 0.0) break;
    }
    const long long t1 = clock64();
    if (tid == 0) { sink[rank] = acc; clk_out[rank] = t1 - t0; }
}

template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_load_rows(const double* __restrict__ F, int stride, double* sink) {
    if (blockIdx.x % stride != 0) return;
    const int team = (int)gridDim.x / stride, rank = (int)blockIdx.x / stride;
    const int rows = (D + team - 1) / team, row0 = rank * rows, row1 = min(D, row0 + rows);
    double s = 0.0;
    const double2* F2 = reinterpret_cast<const double2*>(F);
    for (size_t e = (size_t)row0 * LD / 2 + threadIdx.x; e < (size_t)row1 * LD / 2; e += THREADS) { const double2 v = F2[e]; s += v.x + v.y; }
    if (s == 12345.678) sink[rank] = s;
}

int main() {
    unsigned long long* gran; double *sink, *F; int* xcc; long long* clk;
    CK(hipMalloc(&gran, sizeof(unsigned long long) * 4 * LD));
    CK(hipMemset(gran, 0, sizeof(unsigned long long) * 4 * LD));
    CK(hipMalloc(&sink, sizeof(double) * 256)); CK(hipMalloc(&xcc, sizeof(int) * 256)); CK(hipMalloc(&clk, sizeof(long long) * 256));
    CK(hipMalloc(&F, sizeof(double) * (size_t)LD * LD));
    CK(hipMemset(F, 0, sizeof(double) * (size_t)LD * LD));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    unsigned epoch = 1000;
    const int iters = 200;
    for (int stride : { 8, 4, 2, 1 }) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_exchange<512>, dim3(256), dim3(512), 0, 0, gran, stride, iters, epoch, sink, xcc, clk);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            epoch += iters + 10;
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<int> hx(256); std::vector<double> hs(256);
            CK(hipMemcpy(hx.data(), xcc, sizeof(int) * 256, hipMemcpyDeviceToHost)); CK(hipMemcpy(hs.data(), sink, sizeof(double) * 256, hipMemcpyDeviceToHost));
            int team = 256 / stride, on0 = 0; bool bad = false;
            for (int r = 0; r < team; ++r) { on0 += hx[r] == hx[0]; bad |= hs[r] < 0.0; }
            if (rep == 2) printf("team %3d workgroups (every %d-th), %d of them on the XCD of member 0: %.2f us per exchange%s\n", team, stride, on0, 1e3 * ms / iters, bad ? "  TIMEOUT" : "");
        }
    }
    for (int stride : { 8, 1 }) {
        for (int rep = 0; rep < 3; ++rep) {
            // evict: touch another buffer? the matrix was just written by a memset (MALL / HBM resident like after the pair pass)
            CK(hipMemsetAsync(F, 0, sizeof(double) * (size_t)LD * LD));
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_load_rows<512>, dim3(256), dim3(512), 0, 0, F, stride, sink);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 2) printf("loading the %d x %d fp64 matrix (%.1f MB) by a team of %d workgroups: %.1f us (launch included)\n", D, LD, 8.0 * D * LD / 1e6, 256 / stride, 1e3 * ms);
        }
    }
    return 0;
}
