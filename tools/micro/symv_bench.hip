// symv_bench.hip -- the symmetric product of the CG on the UPPER triangle of the dense reduced matrix (round 6), stand-alone:
//   q = (U + U^T - diag(U)) p   with U the row-major upper triangle (diagonal included) of a d x d fp32 matrix, p and q fp64
// against the full-matrix product the streaming CG does today (both triangles read).  Prices the two things the design hinges on:
//   (1) the tile kernel: one 16-byte load per four entries, BOTH uses of an entry from that load (row sums reduced across the wave by a halving
//       butterfly, column sums kept per lane and reduced across the four waves in LDS);
//   (2) what the cross-workgroup sums cost as fp64 device-scope atomics on q (R + C per tile) -- measured with and without them.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o symv_bench symv_bench.hip && ./symv_bench [d]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include <algorithm>
#include "../../sfm-toy-library_amd/csrc/sfmba_device.h"

using namespace sfmba;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int TC = 256;      // columns of a tile: 64 lanes x one 16-byte load

// sums of 8 per-lane values over the wave: afterwards lane l holds the total of value (l >> 3) & 7 ... in every lane of its group of 8 lanes
__device__ __forceinline__ double rows8_reduce(double (&v)[8], int lane) {
    double a[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] = xlane_pairsum<32>(v[k], v[4 + k]);          // lanes < 32: values 0..3, lanes >= 32: values 4..7
    double b[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) b[k] = xlane_pairsum<16>(a[k], a[2 + k]);          // bit 4 clear: first half of those, set: second half
    const bool up = (lane & 8) != 0;
    const double send = up ? b[0] : b[1], keep = up ? b[1] : b[0];
    double c = keep + xlane_get<8>(send);
    c = xlane_add<4>(c); c = xlane_add<2>(c); c = xlane_add<1>(c);
    return c;        // value index: (lane >> 5) * 4 + ((lane >> 4) & 1) * 2 + ((lane >> 3) & 1)
}

// tile = R rows x 256 columns; tiles[t] = {first row, first column (multiple of 256), rows, flags: 1 = overlaps the diagonal}
template <int RPW, bool ATOMICS>
__global__ __launch_bounds__(256) void k_symv(int d, int ld, const float* __restrict__ F, const double* __restrict__ p, double* __restrict__ q,
                                              const int4* __restrict__ tiles, double* __restrict__ sink) {
    __shared__ double colsh[4][TC];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int4 tl = tiles[blockIdx.x];
    const int r0 = tl.x, c0 = tl.y, nrows = tl.z;
    const bool diag = tl.w != 0;
    const int j0 = c0 + 4 * lane;
    double pj[4], colacc[4] = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
    for (int e = 0; e < 4; ++e) pj[e] = j0 + e < d ? p[j0 + e] : 0.0;
    const int rw0 = r0 + w * RPW;
    double mine = 0.0;           // row total this lane will publish
    int mine_row = -1;
#pragma unroll
    for (int b0 = 0; b0 < RPW; b0 += 8) {
        float4 f[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int row = rw0 + b0 + u;
            const int rc = row < r0 + nrows ? row : r0;
            f[u] = *reinterpret_cast<const float4*>(F + (size_t)rc * ld + j0);
        }
        double racc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int row = __builtin_amdgcn_readfirstlane(rw0 + b0 + u);
            const bool live = row < r0 + nrows;
            const double pi = live ? p[row] : 0.0;
            double x[4] = { (double)f[u].x, (double)f[u].y, (double)f[u].z, (double)f[u].w };
            double s = 0.0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = j0 + e;
                double xe = x[e];
                if (diag) { if (j < row) xe = 0.0; }
                if (j >= d || !live) xe = 0.0;
                s = fma(xe, pj[e], s);
                const double xc = (diag && j == row) ? 0.0 : xe;        // the diagonal entry is used once
                colacc[e] = fma(xc, pi, colacc[e]);
            }
            racc[u] = s;
        }
        const double tot = rows8_reduce(racc, lane);
        const int vi = (lane >> 5) * 4 + ((lane >> 4) & 1) * 2 + ((lane >> 3) & 1);
        if ((lane & 7) == (b0 >> 3)) { mine = tot; mine_row = rw0 + b0 + vi; }      // batch b publishes through lanes with (lane & 7) == b
    }
    if (mine_row >= 0 && mine_row < r0 + nrows) {
        if (ATOMICS) atomicAdd(q + mine_row, mine); else sink[(size_t)blockIdx.x * 512 + 256 + (mine_row - r0)] = mine;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) colsh[w][4 * lane + e] = colacc[e];
    __syncthreads();
    const double cs = colsh[0][tid] + colsh[1][tid] + colsh[2][tid] + colsh[3][tid];
    if (c0 + tid < d) {
        if (ATOMICS) atomicAdd(q + c0 + tid, cs); else sink[(size_t)blockIdx.x * 512 + tid] = cs;
    }
}

// the same tiles walked by FEWER workgroups (nwg of them, tile t = blockIdx.x + k nwg), as the CG's iteration kernel has to (its vector phase is per
// workgroup): MODE 0 plain loop, 1 = the next tile's eight loads issued row by row while the current tile is consumed (rolling prefetch),
// 2 = MODE 1 + every workgroup first reads two whole vectors (the |r - alpha q|^2 pass of the CG: 96 KB from L2 per workgroup)
template <int MODE>
__global__ __launch_bounds__(256) void k_symv_loop(int d, int ld, const float* __restrict__ F, const double* __restrict__ p, double* __restrict__ q,
                                                   const int4* __restrict__ tiles, int ntiles, const double* __restrict__ r2, double* __restrict__ sink) {
    __shared__ double colsh[4][TC];
    __shared__ double red4[4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nwg = gridDim.x;
    double alpha = 1.0;
    int t = blockIdx.x;
    int4 tl = tiles[t < ntiles ? t : 0];
    float4 f[8];
    if (t < ntiles) {        // the first tile's loads BEFORE the vector phase
        const int j0 = tl.y + 4 * lane, rw0 = tl.x + 8 * w;
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int row = rw0 + u; f[u] = *reinterpret_cast<const float4*>(F + (size_t)(row < tl.x + tl.z ? row : tl.x) * ld + j0); }
    }
    if (MODE == 2) {
        double rr = 0.0;
        const double2* p2 = reinterpret_cast<const double2*>(p); const double2* r22 = reinterpret_cast<const double2*>(r2);
        const int n2 = d >> 1;
        for (int e0 = tid; e0 < n2; e0 += 256 * 6) {
            double2 a8[6], b8[6];
#pragma unroll
            for (int u = 0; u < 6; ++u) { const int e = e0 + 256 * u, ec = e < n2 ? e : n2 - 1; a8[u] = p2[ec]; b8[u] = r22[ec]; }
#pragma unroll
            for (int u = 0; u < 6; ++u) { const int e = e0 + 256 * u; if (e < n2) { const double v = a8[u].x - 0.5 * b8[u].x, v2 = a8[u].y - 0.5 * b8[u].y; rr += v * v + v2 * v2; } }
        }
        rr = wave_allsum(rr);
        if (lane == 0) red4[w] = rr;
        __syncthreads();
        alpha = (red4[0] + red4[1] + red4[2] + red4[3]) > 1e300 ? 2.0 : 1.0;      // (keeps the pass alive; always 1)
    }
    for (; t < ntiles; t += nwg) {
        const bool has_next = t + nwg < ntiles;
        const int4 tnx = tiles[has_next ? t + nwg : t];
        const int r0 = tl.x, c0 = tl.y, rend = tl.x + tl.z;
        const bool diag = tl.w != 0;
        const int j0 = c0 + 4 * lane, rw0 = r0 + 8 * w;
        double pj[4], colacc[4] = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
        for (int e = 0; e < 4; ++e) pj[e] = j0 + e < d ? alpha * p[j0 + e] : 0.0;
        const int myrow = rw0 + (lane >> 3);
        const double prow = myrow < rend ? alpha * p[myrow] : 0.0;
        double racc[8];
        const int nj0 = tnx.y + 4 * lane, nrw0 = tnx.x + 8 * w;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int row = rw0 + u;
            const bool live = row < rend;
            const double pi = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(prow), 8 * u), __builtin_amdgcn_readlane(__double2loint(prow), 8 * u));
            const double x[4] = { (double)f[u].x, (double)f[u].y, (double)f[u].z, (double)f[u].w };
            if (MODE >= 1 && has_next) { const int nrow = nrw0 + u; f[u] = *reinterpret_cast<const float4*>(F + (size_t)(nrow < tnx.x + tnx.z ? nrow : tnx.x) * ld + nj0); }
            double s = 0.0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = j0 + e;
                double xe = x[e];
                if (!live || j >= d || (diag && j < row)) xe = 0.0;
                s = fma(xe, pj[e], s);
                colacc[e] = fma((diag && j == row) ? 0.0 : xe, pi, colacc[e]);
            }
            racc[u] = s;
        }
        if (MODE == 0 && has_next) {
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int nrow = nrw0 + u; f[u] = *reinterpret_cast<const float4*>(F + (size_t)(nrow < tnx.x + tnx.z ? nrow : tnx.x) * ld + nj0); }
        }
        const double tot = rows8_reduce(racc, lane);
        if ((lane & 7) == 0 && myrow < rend) atomicAdd(q + myrow, tot);
#pragma unroll
        for (int e = 0; e < 4; ++e) colsh[w][4 * lane + e] = colacc[e];
        __syncthreads();
        if (c0 + tid < d) atomicAdd(q + c0 + tid, (colsh[0][tid] + colsh[1][tid]) + (colsh[2][tid] + colsh[3][tid]));
        __syncthreads();
        tl = tnx;
    }
}

// Wave-independent form: the triangle as UNITS of 8 rows x 256 columns (one batch of eight 16-byte loads per lane), ordered chunk by chunk (a chunk =
// 256 columns) with ascending rows inside a chunk; every wave takes a contiguous run of units, keeps the column sums of its run in registers and flushes
// them with atomics when the chunk changes or the run ends -- no LDS exchange, no barrier in the product.  DEPTH units' loads are in flight per wave
// (rolling), the first DEPTH issued BEFORE the (emulated) vector phase.  One workgroup of WPB waves per CU: the 96 KB vector pass once per CU.
template <int DEPTH, bool VEC, int WPB>
__global__ __launch_bounds__(64 * WPB) void k_symv_units(int d, int ld, const float* __restrict__ F, const double* __restrict__ p, double* __restrict__ q,
                                                     int nunits, int nrg, const double* __restrict__ r2) {
    __shared__ double red4[16];
    __shared__ double tsh[WPB][256];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wpb = blockDim.x >> 6;
    const int nwaves = gridDim.x * wpb, gw = blockIdx.x * wpb + w;
    // column sums of a run leave through LDS so that every atomic instruction covers 512 contiguous bytes (4 lines, not 16)
    auto flush = [&](int cc, const double (&acc)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 4; ++e) tsh[w][4 * lane + e] = acc[e];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int j = 256 * cc + 64 * k + lane; if (j < d) atomicAdd(q + j, tsh[w][64 * k + lane]); }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
    };
    const int u0 = (int)((long long)nunits * gw / nwaves), u1 = (int)((long long)nunits * (gw + 1) / nwaves);
    int c = 0, rg = 0;
    { int base = 0; for (;;) { const int n = min(nrg, 32 * (c + 1)); if (u0 < base + n || c >= 1000) break; base += n; ++c; } rg = u0 - base; }
    int pc = c, prg = rg;               // position of the next unit to prefetch
#define ISSUE(dst) do { const int j0_ = 256 * pc + 4 * lane; _Pragma("unroll") for (int u_ = 0; u_ < 8; ++u_) { const int row_ = 8 * prg + u_; \
                        dst[u_] = *reinterpret_cast<const float4*>(F + (size_t)(row_ < d ? row_ : 0) * ld + j0_); } \
                        if (++prg >= min(nrg, 32 * (pc + 1))) { prg = 0; ++pc; } } while (0)
    float4 fA[8], fB[8];
    if (u0 < u1) ISSUE(fA);
    if (DEPTH == 2 && u0 + 1 < u1) ISSUE(fB);
    double alpha = 1.0;
    if (VEC) {
        double rr = 0.0;
        for (int e0 = tid; e0 < d; e0 += blockDim.x * 4) {
            double a8[4], b8[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int e = e0 + blockDim.x * u, ec = e < d ? e : d - 1; a8[u] = p[ec]; b8[u] = r2[ec]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int e = e0 + blockDim.x * u; if (e < d) { const double v = a8[u] - 0.5 * b8[u]; rr += v * v; } }
        }
        rr = wave_allsum(rr);
        if (lane == 0) red4[w] = rr;
        __syncthreads();
        double t = 0.0; for (int i = 0; i < wpb; ++i) t += red4[i];
        alpha = t > 1e300 ? 2.0 : 1.0;
    }
    double pj[4] = { 0, 0, 0, 0 }, colacc[4] = { 0, 0, 0, 0 };
    int curc = -1;
    auto consume = [&](const float4 (&f)[8]) __attribute__((always_inline)) {
        if (c != curc) {
            if (curc >= 0) flush(curc, colacc);
            curc = c;
#pragma unroll
            for (int e = 0; e < 4; ++e) { const int j = 256 * c + 4 * lane + e; pj[e] = j < d ? alpha * p[j] : 0.0; colacc[e] = 0.0; }
        }
        const int j0 = 256 * c + 4 * lane, rw0 = 8 * rg;
        const bool diag = rw0 + 7 >= 256 * c;
        const int myrow = rw0 + (lane >> 3);
        const double prow = myrow < d ? alpha * p[myrow] : 0.0;
        double racc[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int row = rw0 + r;
            const double pi = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(prow), 8 * r), __builtin_amdgcn_readlane(__double2loint(prow), 8 * r));
            const double x[4] = { (double)f[r].x, (double)f[r].y, (double)f[r].z, (double)f[r].w };
            double sacc = 0.0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = j0 + e;
                double xe = x[e];
                if (row >= d || j >= d || (diag && j < row)) xe = 0.0;
                sacc = fma(xe, pj[e], sacc);
                colacc[e] = fma((diag && j == row) ? 0.0 : xe, pi, colacc[e]);
            }
            racc[r] = sacc;
        }
        const double tot = rows8_reduce(racc, lane);
        if ((lane & 7) == 0 && myrow < d) atomicAdd(q + myrow, tot);
        if (++rg >= min(nrg, 32 * (c + 1))) { rg = 0; ++c; }
    };
    for (int u = u0; u < u1; u += DEPTH) {
        consume(fA);
        if (u + DEPTH < u1) ISSUE(fA);
        if (DEPTH == 2 && u + 1 < u1) {
            consume(fB);
            if (u + 3 < u1) ISSUE(fB);
        }
    }
#undef ISSUE
    if (curc >= 0) flush(curc, colacc);
}

// the product as the streaming CG does it today: both triangles read, a wave per two rows, 16-byte loads four deep (k_pcg_iter, dense_solver.hip)
__global__ __launch_bounds__(256) void k_full(int d, int ld, const float* __restrict__ F, const double* __restrict__ p, double* __restrict__ q, int rows_per_wg) {
    extern __shared__ double pl[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int e = tid; e < ld; e += 256) pl[e] = e < d ? p[e] : 0.0;
    __syncthreads();
    const int row0 = blockIdx.x * rows_per_wg, row1 = min(d, row0 + rows_per_wg), nd4 = ld >> 2;
    for (int row = row0 + w; row < row1; row += 8) {
        const int rowb = row + 4 < row1 ? row + 4 : row;
        const float4* Fa = reinterpret_cast<const float4*>(F + (size_t)row * ld);
        const float4* Fb = reinterpret_cast<const float4*>(F + (size_t)rowb * ld);
        double sa = 0.0, sb = 0.0;
        for (int c = lane; c < nd4; c += 256) {
            float4 a[4], b[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) { const int cc = c + 64 * m < nd4 ? c + 64 * m : lane; a[m] = Fa[cc]; b[m] = Fb[cc]; }
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                if (c + 64 * m >= nd4) continue;
                const double2 p0 = reinterpret_cast<const double2*>(pl)[2 * (c + 64 * m)], p1 = reinterpret_cast<const double2*>(pl)[2 * (c + 64 * m) + 1];
                sa += (double)a[m].x * p0.x + (double)a[m].y * p0.y + (double)a[m].z * p1.x + (double)a[m].w * p1.y;
                sb += (double)b[m].x * p0.x + (double)b[m].y * p0.y + (double)b[m].z * p1.x + (double)b[m].w * p1.y;
            }
        }
        sa = wave_allsum(sa); sb = wave_allsum(sb);
        if (lane == 0) { q[row] = sa; if (rowb != row) q[rowb] = sb; }
    }
}

// only the atomics of the tile kernel (R + 256 per tile), nothing else: what the cross-workgroup sums cost on their own
__global__ __launch_bounds__(256) void k_atomics_only(int d, double* __restrict__ q, const int4* __restrict__ tiles, int rpt) {
    const int4 tl = tiles[blockIdx.x];
    if (tl.y + (int)threadIdx.x < d) atomicAdd(q + tl.y + threadIdx.x, 1.0);
    if ((int)threadIdx.x < rpt && tl.x + (int)threadIdx.x < d) atomicAdd(q + tl.x + threadIdx.x, 1.0);
}

template <typename Fn> static float time_us(hipStream_t s, Fn fn, int reps = 20) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) fn();
    hipEventRecord(e0, s);
    for (int i = 0; i < reps; ++i) fn();
    hipEventRecord(e1, s); hipStreamSynchronize(s);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return 1e3f * ms / reps;
}

template <int RPW> static int run(int d, int ld, const float* dF, const double* dp, double* dq, double* dsink, const std::vector<double>& want, hipStream_t s) {
    const int R = 4 * RPW;
    std::vector<int4> tiles;
    for (int r0 = 0; r0 < d; r0 += R) {
        const int nrows = std::min(R, d - r0);
        for (int c0 = (r0 / TC) * TC; c0 < d; c0 += TC) tiles.push_back(make_int4(r0, c0, nrows, c0 < r0 + nrows ? 1 : 0));
    }
    int4* dt; CK(hipMalloc(&dt, tiles.size() * sizeof(int4))); CK(hipMemcpy(dt, tiles.data(), tiles.size() * sizeof(int4), hipMemcpyHostToDevice));
    const int nt = (int)tiles.size();
    // correctness
    CK(hipMemsetAsync(dq, 0, sizeof(double) * ld, s));
    hipLaunchKernelGGL((k_symv<RPW, true>), dim3(nt), dim3(256), 0, s, d, ld, dF, dp, dq, dt, dsink);
    std::vector<double> got(d);
    CK(hipMemcpyAsync(got.data(), dq, sizeof(double) * d, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
    double err = 0.0, scale = 0.0;
    for (int i = 0; i < d; ++i) { err = std::max(err, fabs(got[i] - want[i])); scale = std::max(scale, fabs(want[i])); }
    const float t_at = time_us(s, [&]() { hipMemsetAsync(dq, 0, sizeof(double) * ld, s); hipLaunchKernelGGL((k_symv<RPW, true>), dim3(nt), dim3(256), 0, s, d, ld, dF, dp, dq, dt, dsink); });
    const float t_ms = time_us(s, [&]() { hipMemsetAsync(dq, 0, sizeof(double) * ld, s); });
    const float t_no = time_us(s, [&]() { hipLaunchKernelGGL((k_symv<RPW, false>), dim3(nt), dim3(256), 0, s, d, ld, dF, dp, dq, dt, dsink); });
    const float t_ao = time_us(s, [&]() { hipLaunchKernelGGL(k_atomics_only, dim3(nt), dim3(256), 0, s, d, dq, dt, R); });
    const double mb = 2.0 * d * (double)d / 1e6;      // bytes of the upper triangle in fp32 (d^2 / 2 entries x 4 B), MB
    printf("tile %3d x %d: %5d tiles, %7d atomics | symv + atomics %6.2f us (%.2f TB/s of the %.1f MB upper triangle; memset alone %.2f us) | partial stores instead %6.2f us | atomics only %6.2f us | max err %.2e of %.2e\n",
           R, TC, nt, nt * (R + TC), t_at - t_ms, mb / (t_at - t_ms), mb, t_ms, t_no, t_ao, err, scale);
    hipFree(dt);
    return 0;
}

int main(int argc, char** argv) {
    const int d = argc > 1 ? atoi(argv[1]) : 6001;
    const int ld = (d + 63) / 64 * 64;
    std::vector<float> F((size_t)d * ld);
    std::vector<double> p(d), want(d, 0.0);
    srand(7);
    for (int i = 0; i < d; ++i) p[i] = rand() / (double)RAND_MAX - 0.5;
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < ld; ++j) F[(size_t)i * ld + j] = j >= d ? NAN : (j >= i ? (float)(rand() / (double)RAND_MAX - 0.5) : 1e30f);   // lower triangle / padding: poison
    for (int i = 0; i < d; ++i)
        for (int j = i; j < d; ++j) { const double u = F[(size_t)i * ld + j]; want[i] += u * p[j]; if (j > i) want[j] += u * p[i]; }
    hipStream_t s; CK(hipStreamCreate(&s));
    float* dF; double *dp, *dq, *dsink;
    CK(hipMalloc(&dF, F.size() * sizeof(float))); CK(hipMalloc(&dp, sizeof(double) * ld)); CK(hipMalloc(&dq, sizeof(double) * ld)); CK(hipMalloc(&dsink, sizeof(double) * 512 * 8192));
    CK(hipMemcpy(dF, F.data(), F.size() * sizeof(float), hipMemcpyHostToDevice)); CK(hipMemcpy(dp, p.data(), sizeof(double) * d, hipMemcpyHostToDevice));
    printf("d = %d (ld %d): fp32 matrix %.1f MB, upper triangle %.1f MB\n", d, ld, 4.0 * d * ld / 1e6, 2.0 * d * d / 1e6);
    // the full-matrix product (symmetric completion of the same matrix would be needed for equal results; timing only)
    for (int rows_per_wg : { 6, 8 }) {
        const int nwg = (d + rows_per_wg - 1) / rows_per_wg;
        std::vector<float> Fz((size_t)d * ld, 0.25f);
        float* dFz; CK(hipMalloc(&dFz, Fz.size() * sizeof(float))); CK(hipMemcpy(dFz, Fz.data(), Fz.size() * sizeof(float), hipMemcpyHostToDevice));
        const float t = time_us(s, [&]() { hipLaunchKernelGGL(k_full, dim3(nwg), dim3(256), sizeof(double) * ld, s, d, ld, dFz, dp, dq, rows_per_wg); });
        printf("full matrix, %d rows per workgroup (%d workgroups): %6.2f us (%.2f TB/s of %.1f MB)\n", rows_per_wg, nwg, t, 4.0 * d * ld / 1e6 / t, 4.0 * d * ld / 1e6);
        hipFree(dFz);
    }
    {   // the loop forms on the 32 x 256 tiles
        std::vector<int4> tiles;
        for (int r0 = 0; r0 < d; r0 += 32) { const int nrows = std::min(32, d - r0); for (int c0 = (r0 / TC) * TC; c0 < d; c0 += TC) tiles.push_back(make_int4(r0, c0, nrows, c0 < r0 + nrows ? 1 : 0)); }
        int4* dt; CK(hipMalloc(&dt, tiles.size() * sizeof(int4))); CK(hipMemcpy(dt, tiles.data(), tiles.size() * sizeof(int4), hipMemcpyHostToDevice));
        const int nt = (int)tiles.size();
        const float t_ms = time_us(s, [&]() { hipMemsetAsync(dq, 0, sizeof(double) * ld, s); });
        for (int nwg : { 512, 640, 768, 1024, 1200, 1536, 2048, nt }) {
            float tm[3]; double err = 0.0;
            for (int mode = 0; mode < 3; ++mode) {
                auto go = [&]() { hipMemsetAsync(dq, 0, sizeof(double) * ld, s);
                                  if (mode == 0) hipLaunchKernelGGL(k_symv_loop<0>, dim3(nwg), dim3(256), 0, s, d, ld, dF, dp, dq, dt, nt, dp, dsink);
                                  else if (mode == 1) hipLaunchKernelGGL(k_symv_loop<1>, dim3(nwg), dim3(256), 0, s, d, ld, dF, dp, dq, dt, nt, dp, dsink);
                                  else hipLaunchKernelGGL(k_symv_loop<2>, dim3(nwg), dim3(256), 0, s, d, ld, dF, dp, dq, dt, nt, dp, dsink); };
                tm[mode] = time_us(s, go) - t_ms;
                go();
                std::vector<double> got(d);
                CK(hipMemcpyAsync(got.data(), dq, sizeof(double) * d, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
                for (int i = 0; i < d; ++i) err = std::max(err, fabs(got[i] - want[i]));
            }
            printf("loop over 32 x 256 tiles, %4d workgroups: plain %6.2f us | rolling prefetch %6.2f us | + the 96 KB vector pass per workgroup %6.2f us | max err %.2e\n", nwg, tm[0], tm[1], tm[2], err);
        }
        hipFree(dt);
    }
    {   // the wave-independent unit form
        const int nrg = (d + 7) / 8, nch = (d + 255) / 256;
        int nunits = 0; for (int c = 0; c < nch; ++c) nunits += std::min(nrg, 32 * (c + 1));
        const float t_ms = time_us(s, [&]() { hipMemsetAsync(dq, 0, sizeof(double) * ld, s); });
        auto bench = [&](auto launch, int nwg, int wpb, const char* what) {
            const float t = time_us(s, [&]() { hipMemsetAsync(dq, 0, sizeof(double) * ld, s); launch(); }) - t_ms;
            hipMemsetAsync(dq, 0, sizeof(double) * ld, s); launch();
            std::vector<double> got(d); double err = 0.0;
            hipMemcpyAsync(got.data(), dq, sizeof(double) * d, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s);
            for (int i = 0; i < d; ++i) err = std::max(err, fabs(got[i] - want[i]));
            printf("units (8 x 256), %3d workgroups of %2d waves (%.2f units per wave), %-34s %6.2f us   max err %.2e\n", nwg, wpb, (double)nunits / (nwg * wpb), what, t, err);
        };
#define UNITS(D, V, W, nwg, what) bench([&]() { hipLaunchKernelGGL((k_symv_units<D, V, W>), dim3(nwg), dim3(64 * W), 0, s, d, ld, dF, dp, dq, nunits, nrg, dp); }, nwg, W, what)
        UNITS(1, false, 4, 256, "depth 1"); UNITS(2, false, 4, 256, "depth 2"); UNITS(2, true, 4, 256, "depth 2 + vector pass");
        UNITS(1, false, 4, 512, "depth 1"); UNITS(2, false, 4, 512, "depth 2"); UNITS(2, true, 4, 512, "depth 2 + vector pass");
        UNITS(1, false, 8, 256, "depth 1"); UNITS(2, false, 8, 256, "depth 2"); UNITS(2, true, 8, 256, "depth 2 + vector pass");
        UNITS(2, false, 8, 512, "depth 2"); UNITS(2, true, 8, 512, "depth 2 + vector pass");
        UNITS(2, false, 16, 256, "depth 2"); UNITS(2, true, 16, 256, "depth 2 + vector pass");
        UNITS(2, false, 12, 256, "depth 2"); UNITS(2, true, 12, 256, "depth 2 + vector pass");
#undef UNITS
    }
    if (run<8>(d, ld, dF, dp, dq, dsink, want, s)) return 1;
    if (run<16>(d, ld, dF, dp, dq, dsink, want, s)) return 1;
    if (run<24>(d, ld, dF, dp, dq, dsink, want, s)) return 1;
    if (run<48>(d, ld, dF, dp, dq, dsink, want, s)) return 1;
    return 0;
}
