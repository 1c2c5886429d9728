// graph_floor.hip -- what a launch boundary costs on this box, in a stream and as nodes of a captured hipGraph.
//   hipcc --offload-arch=gfx950 -O3 -o graph_floor graph_floor.hip && ./graph_floor
// N dependent launches of (a) an empty kernel, (b) a kernel of 256 workgroups that reads and writes one cache line each (the shape of a CG
// iteration of the fast path without its arithmetic); each timed by HIP events around the whole chain, best of 5.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_empty() {}
__global__ __launch_bounds__(256) void k_touch(double* buf, int it) {
    const int i = blockIdx.x * 8;
    if (threadIdx.x == 0) buf[i + (it & 1)] = buf[i + ((it & 1) ^ 1)] + 1.0;
}

int main() {
    const int N = 1000;
    hipStream_t s; CK(hipStreamCreate(&s));
    double* buf; CK(hipMalloc(&buf, 256 * 8 * sizeof(double))); CK(hipMemset(buf, 0, 256 * 8 * sizeof(double)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int kind = 0; kind < 2; ++kind) {
        auto chain = [&]() { for (int i = 0; i < N; ++i) { if (kind == 0) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s); else hipLaunchKernelGGL(k_touch, dim3(256), dim3(256), 0, s, buf, i); } };
        float best_stream = 1e30f, best_graph = 1e30f;
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipEventRecord(e0, s)); chain(); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep) best_stream = std::min(best_stream, ms);
        }
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal)); chain(); CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep) best_graph = std::min(best_graph, ms);
        }
        printf("%-44s stream %.3f us per launch, graph %.3f us per node\n", kind == 0 ? "empty kernel (1 wave):" : "256 workgroups, one cache line each:", 1e3f * best_stream / N, 1e3f * best_graph / N);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
