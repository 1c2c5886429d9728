// dense_solver.hip -- the reduced camera system  S z = rhs  (dim 6*Nc+1) solved on the device by preconditioned CG.
//
// Replaces what DENSE_SCHUR hands to Eigen's LLT in the reference configuration
// (SfMToyLib/SfMBundleAdjustmentUtils.cpp:172, DenseSchurComplementSolver [Ceres-upstream]); the factorisation itself
// (SFMBA_LINEAR_CHOLESKY, AUTO's fallback) is dense_cholesky.hip.  Both share the workspace of dense_solver.h and the storage:
// the kernels in ba_kernels.hip accumulate the UPPER triangle of the row-major matrix, padded to a multiple of CHOL_NB.
//
// Four CG families (DESIGN.md section 4 "CG families"): fast (d <= 1280, rows in registers), segmented fast, streaming, streaming segmented.
#include "dense_solver.h"
#include "sfmba_device.h"
#include "chol_tile.h"
#include "coarse_inverse.h"
#include <math.h>
#include <stdio.h>
#include <algorithm>
#include <type_traits>
#include <utility>
#include <chrono>
#include <cstdlib>

namespace sfmba {

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

constexpr int NB = CHOL_NB;   // padding unit of the matrix (dense_cholesky.hip)

// ------------------------------------------------------------------------------------------
// Block-Jacobi preconditioned conjugate gradients on the dense reduced system.
//
// The preconditioner is folded into the matrix once per solve: with Lb = blockdiag(chol(S_jj)) (6x6
// camera blocks + the 1x1 focal), S~ = Lb^-1 S Lb^-T has identity diagonal blocks and plain CG on
// S~ x~ = Lb^-1 rhs is exactly block-Jacobi PCG on S.  Each CG iteration is then ONE kernel launch:
// every workgroup redundantly forms alpha, r, beta and the new search direction p (length d, from L2)
// in LDS, multiplies its own rows of S~ by p and publishes its slice of x, r, p, q = S~ p plus its
// partial p.q; the next launch (stream order) finishes the dot product.  Vectors are double-buffered
// by iteration parity so no workgroup overwrites what another one is still reading.  The rows a
// workgroup owns never change, so its slab of S~ stays in its XCD's L2 across iterations.
// ------------------------------------------------------------------------------------------
#ifndef SFMBA_PCG_MAXWG
#define SFMBA_PCG_MAXWG 256
#endif
constexpr int PCG_MAXWG = SFMBA_PCG_MAXWG;        // workgroups of the fast path (one partial dot product per thread)
constexpr int PCG_MAXWG_BIG = 1024;   // workgroups of the generic path
constexpr int PCG_PART = 1024;        // stride (workgroups) of the per-iteration partial-sum buffers
enum { PF_DONE = 0, PF_ITERS = 1, PF_XBUF = 2 };
enum { PS_RR0 = 0, PS_RRF = 1 };     // threshold base of the running solve; |b~|^2 of the FIRST solve of an anchored sequence

// Stopping rule: |r|^2 <= tol^2 * base.  Plain CG: base = |b~|^2 (relative residual).  Inside one LM solve the
// tolerance is ANCHORED to the first iteration's right-hand side: base = min(max(|b~_k|^2, |b~_first|^2), cap * |b~_k|^2).
// Why: the error a truncated solve leaves in the PARAMETERS is ~ cond * |r|, absolute -- the first LM step is orders of
// magnitude larger than the later ones, so a relative tolerance spends its iterations on the small steps and leaves
// the big step's error (drift along the gauge directions, 1e-4 at tol 1e-6) in the result.  Anchored, every step is
// solved to the same absolute accuracy; cap keeps every solve at least 1e-4 relative (the accept/reject and
// function-tolerance decisions of the LM loop are insensitive well beyond that).
__device__ __forceinline__ double pcg_threshold_base(double rr, double* scal, int anchor, double cap) {
    if (anchor == 1) { scal[PS_RRF] = rr; return rr; }
    if (anchor == 2) return fmin(fmax(rr, scal[PS_RRF]), cap * rr);
    return rr;
}
// vec layout: x[2] r[2] p[2] q[2], each ld doubles; btilde after them
__device__ __forceinline__ double* pcg_vec(double* vec, int which, int buf, int ld) { return vec + (size_t)(2 * which + buf) * ld; }

// Linv of each diagonal block: row-major lower 6x6 (zeros above), focal: 1/sqrt(S_ff) at [nb6*36]
__global__ void k_pcg_blockchol(const double* __restrict__ S, int ld, int d, double* __restrict__ linv, int* info) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    const int nb6 = (d - 1) / 6;
    const int rem = d - 6 * nb6;     // trailing scalars (1 for a BA system)
    if (b >= nb6 + rem) return;
    if (b >= nb6) {
        const int e = 6 * nb6 + (b - nb6);
        const double v = S[(size_t)e * ld + e];
        if (!(v > 0.0)) atomicCAS(info, 0, e + 1);
        linv[(size_t)nb6 * 36 + (b - nb6)] = 1.0 / sqrt(v > 0.0 ? v : 1.0);
        return;
    }
    double L[6][6], Li[6][6];
    const int o = 6 * b;
    for (int r = 0; r < 6; ++r)
        for (int c = 0; c <= r; ++c) L[r][c] = S[(size_t)(o + c) * ld + o + r];
    bool ok = true;
    for (int j = 0; j < 6; ++j) {
        double dj = L[j][j];
        for (int t = 0; t < j; ++t) dj -= L[j][t] * L[j][t];
        if (!(dj > 0.0)) { ok = false; dj = 1.0; }
        const double lj = sqrt(dj);
        L[j][j] = lj;
        for (int i = j + 1; i < 6; ++i) {
            double v = L[i][j];
            for (int t = 0; t < j; ++t) v -= L[i][t] * L[j][t];
            L[i][j] = v / lj;
        }
    }
    if (!ok) atomicCAS(info, 0, o + 1);
    for (int c = 0; c < 6; ++c)
        for (int r = 0; r < 6; ++r) {
            if (r < c) { Li[r][c] = 0.0; continue; }
            double v = (r == c) ? 1.0 : 0.0;
            for (int t = c; t < r; ++t) v -= L[r][t] * Li[t][c];
            Li[r][c] = v / L[r][r];
        }
    for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 6; ++c) linv[(size_t)b * 36 + r * 6 + c] = Li[r][c];
}

// F = Lb^-1 S Lb^-T (full symmetric, d rows x ld), btilde = Lb^-1 rhs.  One thread per block pair I <= J;
// trailing 1x1 blocks are padded to 6x6 with zeros so that every loop has compile-time bounds (registers).
template <typename FT>
__global__ __launch_bounds__(64) void k_pcg_transform(const double* __restrict__ S, int ld, int d, const double* __restrict__ linv,
                                                      const double* __restrict__ rhs, FT* __restrict__ F, double* __restrict__ bt) {
    const int nb6 = (d - 1) / 6;
    const int nB = nb6 + (d - 6 * nb6);
    const int J = blockIdx.x * blockDim.x + threadIdx.x;
    const int I = blockIdx.y;
    if (J >= nB || J < I) return;
    const int ri = I < nb6 ? 6 * I : 6 * nb6 + (I - nb6), si = I < nb6 ? 6 : 1;
    const int rj = J < nb6 ? 6 * J : 6 * nb6 + (J - nb6), sj = J < nb6 ? 6 : 1;
    double A[6][6], Lj[6][6], Li[6][6];
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const double fi = (r == 0 && c == 0) ? linv[(size_t)nb6 * 36 + (I < nb6 ? 0 : I - nb6)] : 0.0;
            const double fj = (r == 0 && c == 0) ? linv[(size_t)nb6 * 36 + (J < nb6 ? 0 : J - nb6)] : 0.0;
            Li[r][c] = (I < nb6) ? linv[(size_t)I * 36 + r * 6 + c] : fi;
            Lj[r][c] = (J < nb6) ? linv[(size_t)J * 36 + r * 6 + c] : fj;
            const int gr = ri + r, gc = rj + c;
            const bool in = r < si && c < sj;
            // upper storage: element (gr, gc) with column >= row, mirrored inside diagonal blocks
            A[r][c] = in ? (gc >= gr ? S[(size_t)gr * ld + gc] : S[(size_t)gc * ld + gr]) : 0.0;
        }
    double U[6][6];
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            double v = 0.0;
#pragma unroll
            for (int t = 0; t < 6; ++t) v += Li[r][t] * A[t][c];      // Li is lower triangular (zeros above)
            U[r][c] = v;
        }
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            double v = 0.0;
#pragma unroll
            for (int t = 0; t < 6; ++t) v += U[r][t] * Lj[c][t];
            if (r < si && c < sj) {
                F[(size_t)(ri + r) * ld + rj + c] = (FT)v;
                if (I != J) F[(size_t)(rj + c) * ld + ri + r] = (FT)v;
            }
        }
    if (I == J) {
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            double v = 0.0;
#pragma unroll
            for (int t = 0; t < 6; ++t) v += (t < si) ? Li[r][t] * rhs[ri + t] : 0.0;
            if (r < si) bt[ri + r] = v;
        }
    }
}

__device__ __forceinline__ void block_sum2(double& a, double& b, double* red) {
    { a = wave_allsum(a); b = wave_allsum(b); }
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { red[2 * w] = a; red[2 * w + 1] = b; }
    __syncthreads();
    a = red[0] + red[2] + red[4] + red[6];
    b = red[1] + red[3] + red[5] + red[7];
}

// Host mailbox (pinned, host-mapped): {iterations, done}.  The host polls it instead of issuing a D2H copy + stream
// synchronise per batch; written by one lane with system-scope stores.
__device__ __forceinline__ void pcg_post(int* mailbox, int iters, int done) {
    __hip_atomic_store(mailbox + 1, done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(mailbox, iters, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---------------------------------------------------------------------------------------------------------------------
// Coarse space ("gauge deflation").  adjustBundle() holds no parameter block constant (BA.cpp:160-164), so the
// undamped problem is invariant under the 7 similarity transforms of the scene; the LM damping lifts those directions to
// eigenvalues ~1/radius of S~ while the rest of the spectrum sits in [0.2, 2] (measured, cfg 3: seven eigenvalues
// 2.3e-4 .. 2.8e-4, one at 3e-2 -- the focal/depth direction --, everything else >= 0.59).  Plain block-Jacobi CG spends
// most of its iterations on those 8 directions (48 .. 68 iterations to 1e-8) and leaves its truncation error exactly
// there (the "gauge drift" of the parameters).  With the 8 analytic vectors W~ (k_finalize writes them: world
// translation x3, world rotation x3, scale, focal/depth) as a coarse space and the additive two-level preconditioner
//      M^-1 = I + W~ E^-1 W~^T,      E = W~^T S~ W~   (8 x 8),
// the same accuracy takes 8 .. 12 iterations and the coarse components are solved exactly.
//
// The preconditioner is never applied to a full vector: the search direction is kept split, p = p_r + W~ p_mu, so that
//      q = S~ p   = S~ p_r + (S~ W~) p_mu          -- own rows of AW = S~ W~ only
//      W~^T r     carried by the recurrence c <- c - alpha (W~^T q), with W~^T q summed from per-workgroup partials
//      p . q      = p_r . q + p_mu . (W~^T q)
// i.e. one CG iteration still streams r, q, p_r and S~ once; W~ and AW are touched only in the rows a workgroup owns.
// k_pcg_coarse forms AW, E and c_0 = W~^T b~ (one extra pass over S~ per LM iteration), k_pcg_coarse_invert the scaled
// 8 x 8 inverse.  A vector whose pivot vanishes (degenerate configurations, fewer cameras than gauge freedoms) is dropped.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int PCG_NW = 8;                 // coarse vectors
constexpr int PCG_NPART = 1 + PCG_NW;     // per-workgroup partials per iteration: p_r . q, W~^T q
// scal: [PS_RR0] [PS_RRF] ... then per iteration parity PS_STATE_LEN doubles of solver state written by workgroup 0
enum { PS_STATE = 8, PS_STATE_LEN = 32, PS_RZ = 0, PS_C = 1, PS_MU = 9, PS_PMU = 17 };      // c = W~^T r, mu = E^-1 c, p_mu
constexpr int CO_TILE = 1024;             // columns of W~ staged in LDS (fp32) per pass of k_pcg_coarse
constexpr int CO_MAXROWS = 4;             // rows per wave k_pcg_coarse can hold (rows_per_wg <= 16)

__device__ __forceinline__ double* pcg_part(double* part, int parity, int v) { return part + ((size_t)parity * PCG_NPART + v) * PCG_PART; }

// four consecutive matrix entries, loaded with 16-byte loads
template <typename FT> struct Quad;
template <> struct Quad<float> {
    float4 v;
    __device__ __forceinline__ void load(const float* p) { v = *reinterpret_cast<const float4*>(p); }
    __device__ __forceinline__ double get(int i) const { return (double)(i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w); }
};
template <> struct Quad<double> {
    double2 a, b;
    __device__ __forceinline__ void load(const double* p) { a = reinterpret_cast<const double2*>(p)[0]; b = reinterpret_cast<const double2*>(p)[1]; }
    __device__ __forceinline__ double get(int i) const { return i == 0 ? a.x : i == 1 ? a.y : i == 2 ? b.x : b.y; }
};

// NR: rows per wave held at a time (2 for rows_per_wg <= 8 -- BASELINE config 5 --, else NR)
template <typename FT, int NR>
__global__ __launch_bounds__(256) void k_pcg_coarse(int d, int ld, const FT* __restrict__ F, const double* __restrict__ W, const double* __restrict__ bt,
                                                    double* __restrict__ AW, double* __restrict__ epart, int rows_per_wg) {
    __shared__ __align__(16) float wt[PCG_NW][CO_TILE];
    __shared__ double awrow[4][PCG_NW];
    __shared__ double esum[4][PCG_NW * PCG_NW + PCG_NW];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int row0 = blockIdx.x * rows_per_wg, row1 = min(d, row0 + rows_per_wg);
    double acc[NR][PCG_NW];
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int k = 0; k < PCG_NW; ++k) acc[r][k] = 0.0;
    for (int t0 = 0; t0 < d; t0 += CO_TILE) {
        // The wave's rows of this tile first, 16 bytes per load and ALL of them in flight before anything waits (a lane takes four consecutive
        // columns of each 256-column chunk): with one 4-byte load per row and chunk, as this loop used to be written, a wave of cfg 5 (two rows)
        // had 32 bytes per lane in flight and the pass ran at 1.3 TB/s (112 us for the 144 MB of the fp32 matrix).  They do not depend on the
        // staging of W~ below and overlap it.
        Quad<FT> f[NR][CO_TILE / 256];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int row = row0 + w + 4 * r;
            const FT* Fr = F + (size_t)(row < row1 ? row : row0) * ld + t0;
#pragma unroll
            for (int q = 0; q < CO_TILE / 256; ++q) {
                const int c = 256 * q + 4 * lane;
                f[r][q].load(Fr + ((row < row1 && t0 + c < d) ? c : 0));
            }
        }
        __syncthreads();
        // (eight loads in flight per thread: left as one load per loop iteration the staging was a chain of L2 round trips per tile and
        // cost more than the pass over the matrix it serves)
#pragma unroll
        for (int b = 0; b < CO_TILE / 256; ++b) {
            double wv[PCG_NW];
            const int c = tid + 256 * b;
#pragma unroll
            for (int k = 0; k < PCG_NW; ++k) wv[k] = W[(size_t)k * ld + (t0 + c < d ? t0 + c : 0)];
#pragma unroll
            for (int k = 0; k < PCG_NW; ++k) wt[k][c] = (t0 + c < d) ? (float)wv[k] : 0.0f;       // W~ holds fp32-representable values: lossless
        }
        __syncthreads();
        // a lane's W~ values of its four columns are read from LDS (and widened) ONCE per chunk and used by all of the wave's rows
#pragma unroll
        for (int q = 0; q < CO_TILE / 256; ++q) {
            const int c = 256 * q + 4 * lane;
            if (t0 + 256 * q >= d) break;                            // wave-uniform
            float4 wq[PCG_NW];
#pragma unroll
            for (int k = 0; k < PCG_NW; ++k) wq[k] = *reinterpret_cast<const float4*>(&wt[k][c]);
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                if (row0 + w + 4 * r >= row1) continue;              // wave-uniform
                // columns beyond d hold padding: never multiply garbage (W~ is zero there, 0 x NaN is not)
                const double f0 = t0 + c + 0 < d ? f[r][q].get(0) : 0.0, f1 = t0 + c + 1 < d ? f[r][q].get(1) : 0.0;
                const double f2 = t0 + c + 2 < d ? f[r][q].get(2) : 0.0, f3 = t0 + c + 3 < d ? f[r][q].get(3) : 0.0;
#pragma unroll
                for (int k = 0; k < PCG_NW; ++k)
                    acc[r][k] = fma(f0, (double)wq[k].x, fma(f1, (double)wq[k].y, fma(f2, (double)wq[k].z, fma(f3, (double)wq[k].w, acc[r][k]))));
            }
        }
    }
    double e_acc = 0.0, c_acc = 0.0;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int row = row0 + w + 4 * r;
        if (row >= row1) continue;                        // wave-uniform
#pragma unroll
        for (int k = 0; k < PCG_NW; ++k) {
            double v = acc[r][k];
            v = wave_allsum(v);
            if (lane == k) { AW[(size_t)row * PCG_NW + k] = v; awrow[w][k] = v; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        e_acc = fma(W[(size_t)(lane >> 3) * ld + row], awrow[w][lane & 7], e_acc);
        if (lane < PCG_NW) c_acc = fma(W[(size_t)lane * ld + row], bt[row], c_acc);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
    esum[w][lane] = e_acc;
    if (lane < PCG_NW) esum[w][PCG_NW * PCG_NW + lane] = c_acc;
    __syncthreads();
    if (tid < PCG_NW * PCG_NW + PCG_NW) epart[(size_t)tid * PCG_PART + blockIdx.x] = esum[0][tid] + esum[1][tid] + esum[2][tid] + esum[3][tid];
}

// E and c_0 = sums of the per-workgroup partials of k_pcg_coarse*: NPW values per wave (18: E and c_0; 20: W~^T S~ b~ behind them, fast
// path), their lane-partials reduced in lock step; tot[0 .. 4 NPW) (LDS) is complete after the caller's next __syncthreads().  All 256 threads.
template <int NPW = 18>
__device__ __forceinline__ void coarse_sum_partials(int nwg, const double* __restrict__ epart, double* tot) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double part[NPW];
#pragma unroll
    for (int m = 0; m < NPW; ++m) part[m] = 0.0;
    for (int i0 = 0; i0 < nwg; i0 += 128) {               // 2 NPW independent loads per lane and pass (clamped, branch-free)
        double t[NPW][2];
#pragma unroll
        for (int m = 0; m < NPW; ++m)
#pragma unroll
            for (int i = 0; i < 2; ++i) { const int wg = i0 + lane + 64 * i; t[m][i] = epart[(size_t)(w + 4 * m) * PCG_PART + (wg < nwg ? wg : nwg - 1)]; }
#pragma unroll
        for (int m = 0; m < NPW; ++m)
#pragma unroll
            for (int i = 0; i < 2; ++i) part[m] += (i0 + lane + 64 * i < nwg) ? t[m][i] : 0.0;
    }
#pragma unroll
    for (int m = 0; m < NPW; ++m) part[m] = wave_allsum(part[m]);
    if (lane == 0) {
#pragma unroll
        for (int m = 0; m < NPW; ++m) tot[w + 4 * m] = part[m];
    }
}

// stand-alone version (streaming CG path: up to 1024 workgroups of partials): out = [Einv 64 | c_0 8 | E 64]
__global__ __launch_bounds__(256) void k_pcg_coarse_invert(int nwg, const double* __restrict__ epart, double* __restrict__ out) {
    constexpr int N = PCG_NW, NV = N * N + N;
    __shared__ double tot[NV];
    __shared__ double sa[N * N], sb[N * N];
    coarse_sum_partials(nwg, epart, tot);
    __syncthreads();
    if (threadIdx.x >= 64) return;
    if (threadIdx.x < N) out[N * N + threadIdx.x] = tot[N * N + threadIdx.x];
    out[N * N + N + threadIdx.x] = tot[threadIdx.x];          // E itself (symmetric streaming path: W~^T q = AW^T p_r + E p_mu)
    out[threadIdx.x] = coarse_invert_wave(tot, sa, sb);
}

// Sum of the per-workgroup partials of the previous launch.  Wave w owns values w, w + 4, w + 8: `mine` holds this lane's
// share (loaded up front by the caller), the totals land in red[0 .. NV) after the caller's next __syncthreads().  The
// (up to three) wave reductions advance in lock step: a shuffle is ~50 cycles of latency, three dependent chains of six
// would sit on the critical path of every CG iteration.
template <int NV>
__device__ __forceinline__ void reduce_partials(double (&mine)[3], double* red) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < 3; ++j) if (4 * j < NV) mine[j] = wave_allsum(mine[j]);
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < 3; ++j) if (w + 4 * j < NV) red[w + 4 * j] = mine[j];
    }
}

// 8-term dot product as two chains of four (a dependent DFMA is ~32 cycles)
__device__ __forceinline__ double dot8(const double (&a)[PCG_NW], const double (&b)[PCG_NW]) {
    double s0 = a[0] * b[0], s1 = a[1] * b[1];
    s0 = fma(a[2], b[2], s0); s1 = fma(a[3], b[3], s1);
    s0 = fma(a[4], b[4], s0); s1 = fma(a[5], b[5], s1);
    s0 = fma(a[6], b[6], s0); s1 = fma(a[7], b[7], s1);
    return s0 + s1;
}
__device__ __forceinline__ double lane_bcast(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
// out = E^-1 v in every lane: lane t (mod 8) forms row t, eight lane broadcasts distribute the result inside the wave
// (no workgroup barrier).  einv_s: the 64 entries in LDS.
__device__ __forceinline__ void einv_apply(const double* einv_s, const double (&v)[PCG_NW], double (&out)[PCG_NW]) {
    const int t = threadIdx.x & 7;
    double row[PCG_NW];
#pragma unroll
    for (int j = 0; j < PCG_NW; ++j) row[j] = einv_s[t * PCG_NW + j];
    const double r = dot8(row, v);
#pragma unroll
    for (int k = 0; k < PCG_NW; ++k) out[k] = lane_bcast(r, k);
}

// LDS scratch of the CG kernels behind the search direction: [0..9) partial totals, [16..20) rrn per wave,
// [32..40) p_mu of this iteration, [40..76) end-of-kernel partials per wave (4 x 9), [80..144) E^-1; first launch of the fast
// path only: [144..224) E, c_0 and W~^T S~ b~ summed from the partials, [224..352) work space of the 8 x 8 inversion
constexpr int PCG_RED = 352;

// Generic path of one CG iteration (any d).  Vector phase as in the fast path but looped; the matvec streams two rows
// of S~ per wave with 16-byte loads, four deep, so that a wave keeps 128 B per lane in flight (the rows are HBM/MALL
// traffic: d*ld*8 bytes per iteration, 289 MB at d = 6001).  Up to PCG_MAXWG_BIG workgroups.
template <bool INIT, typename FT, bool COARSE>
__global__ __launch_bounds__(256) void k_pcg_iter(int d, int ld, const FT* __restrict__ F, double* __restrict__ vec,
                                                  const double* __restrict__ bt, double* __restrict__ part, double* scal,
                                                  int* flags, int rows_per_wg, double tol2, int in, int* info, int* mailbox, int anchor, double cap,
                                                  const double* __restrict__ W, const double* __restrict__ AW, const double* __restrict__ coarse) {
    extern __shared__ __align__(16) double sm[];
    double* pl = sm;            // [ld] new search direction (p_r)
    double* red = sm + ld;      // [PCG_RED]
    // `in` = (launch number << 1) | parity.  PF_DONE holds the first launch number that has nothing left to do: a launch must not act
    // on the flag its own workgroup 0 raises (workgroups that start late, e.g. behind another process's kernels, would skip the
    // converging iteration's x update).
    const int seq = in >> 1;
    in &= 1;
    if (!INIT) { const int dn = flags[PF_DONE]; if (dn != 0 && seq >= dn) return; }
    constexpr int NV = COARSE ? PCG_NPART : 1;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, out = in ^ 1;
    const int row0 = blockIdx.x * rows_per_wg;
    const int row1 = min(d, row0 + rows_per_wg);
    const int nwg = (int)gridDim.x;
    double* x_out = pcg_vec(vec, 0, out, ld); double* r_out = pcg_vec(vec, 1, out, ld);
    double* p_out = pcg_vec(vec, 2, out, ld); double* q_out = pcg_vec(vec, 3, out, ld);
    const double* x_in = pcg_vec(vec, 0, in, ld); const double* r_in = INIT ? bt : pcg_vec(vec, 1, in, ld);
    const double* p_in = pcg_vec(vec, 2, in, ld); const double* q_in = pcg_vec(vec, 3, in, ld);
    const double* st_in = scal + PS_STATE + PS_STATE_LEN * in;
    double* st_out = scal + PS_STATE + PS_STATE_LEN * out;
    if (COARSE && tid < PCG_NW * PCG_NW) red[80 + tid] = coarse[tid];
    // fp32 matrix: the first 16-byte batch of this wave's first two rows is requested NOW -- the vector phase below (three syncs, the
    // vectors from L2) then runs under the matrix's first memory round trip instead of in front of it
    float4 pre_a[4], pre_b[4];
    const bool pre = sizeof(FT) == 4 && row0 + w < row1 && lane + 192 < (d >> 2);
    if (sizeof(FT) == 4) {
        const int rowa = row0 + w < row1 ? row0 + w : row0, rowb = rowa + 4 < row1 ? rowa + 4 : rowa;
        const float4* Fa = reinterpret_cast<const float4*>(F + (size_t)rowa * ld);
        const float4* Fb = reinterpret_cast<const float4*>(F + (size_t)rowb * ld);
#pragma unroll
        for (int m = 0; m < 4; ++m) { pre_a[m] = Fa[pre ? lane + 64 * m : 0]; pre_b[m] = Fb[pre ? lane + 64 * m : 0]; }
    }
    double c_new[PCG_NW], mu_new[PCG_NW], pmu_new[PCG_NW], pmu_in[PCG_NW];
    double rz_new = 0.0;
    if (INIT) {
        double rr = 0.0;
        for (int e0 = tid; e0 < d; e0 += 256 * 8) {
            double bv8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int e = e0 + 256 * u; bv8[u] = bt[e < d ? e : d - 1]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int e = e0 + 256 * u; if (e < d) { pl[e] = bv8[u]; rr += bv8[u] * bv8[u]; } }
        }
        rr = wave_allsum(rr);
        if (lane == 0) red[16 + w] = rr;
        __syncthreads();
        rr = red[16] + red[17] + red[18] + red[19];
#pragma unroll
        for (int k = 0; k < PCG_NW; ++k) { c_new[k] = COARSE ? coarse[PCG_NW * PCG_NW + k] : 0.0; mu_new[k] = 0.0; pmu_in[k] = 0.0; }
        if (COARSE) einv_apply(red + 80, c_new, mu_new);
        rz_new = rr + (COARSE ? dot8(c_new, mu_new) : 0.0);
#pragma unroll
        for (int k = 0; k < PCG_NW; ++k) pmu_new[k] = mu_new[k];
        for (int e = row0 + tid; e < row1; e += 256) { x_out[e] = 0.0; r_out[e] = pl[e]; p_out[e] = pl[e]; }
        if (blockIdx.x == 0 && tid == 0) {
            scal[PS_RR0] = pcg_threshold_base(rr, scal, anchor, cap); flags[PF_DONE] = (rr == 0.0); flags[PF_ITERS] = 0; flags[PF_XBUF] = out;
            if (mailbox && rr == 0.0) pcg_post(mailbox, 0, 1);
        }
    } else {
        double mine[3] = { 0.0, 0.0, 0.0 };
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int v = w + 4 * j;
            if (4 * j < NV) {
                const double* pp = pcg_part(part, in, v < NV ? v : 0);
                for (int i0 = 0; i0 < nwg; i0 += 256) {       // four loads in flight per value, never a `+= load` chain
                    double t[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) { const int wg = i0 + lane + 64 * i; t[i] = pp[wg < nwg ? wg : nwg - 1]; }
#pragma unroll
                    for (int i = 0; i < 4; ++i) mine[j] += (i0 + lane + 64 * i < nwg && v < NV) ? t[i] : 0.0;
                }
            }
        }
        reduce_partials<NV>(mine, red);
        __syncthreads();
        double g[PCG_NW], Eg[PCG_NW], c_in[PCG_NW], mu_in[PCG_NW];
        const double rz_in = st_in[PS_RZ];
#pragma unroll
        for (int k = 0; k < PCG_NW; ++k) {
            g[k] = COARSE ? red[1 + k] : 0.0;
            pmu_in[k] = COARSE ? st_in[PS_PMU + k] : 0.0;
            c_in[k] = COARSE ? st_in[PS_C + k] : 0.0;
            mu_in[k] = COARSE ? st_in[PS_MU + k] : 0.0;
            Eg[k] = 0.0;
        }
        if (COARSE) einv_apply(red + 80, g, Eg);
        const double pq = red[0] + (COARSE ? dot8(pmu_in, g) : 0.0);
        const double alpha = rz_in / pq;
#pragma unroll
        for (int k = 0; k < PCG_NW; ++k) { c_new[k] = fma(-alpha, g[k], c_in[k]); mu_new[k] = fma(-alpha, Eg[k], mu_in[k]); }
        const double cmu = COARSE ? dot8(c_new, mu_new) : 0.0;
        double rrn = 0.0;
        // (eight elements' loads in flight: one load / use pair per loop iteration is a chain of d / 256 cache round trips)
        for (int e0 = tid; e0 < d; e0 += 256 * 8) {
            double rv8[8], qv8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int e = e0 + 256 * u, ec = e < d ? e : d - 1; rv8[u] = r_in[ec]; qv8[u] = q_in[ec]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int e = e0 + 256 * u; if (e < d) { const double v = rv8[u] - alpha * qv8[u]; pl[e] = v; rrn += v * v; } }
        }
        rrn = wave_allsum(rrn);
        if (lane == 0) red[16 + w] = rrn;
        // x += alpha p  with p = p_r + W~ p_mu, own rows
        for (int e = row0 + tid; e < row1; e += 256) {
            double pe = p_in[e];
            if (COARSE) {
#pragma unroll
                for (int k = 0; k < PCG_NW; ++k) pe = fma(W[(size_t)k * ld + e], pmu_in[k], pe);
            }
            x_out[e] = x_in[e] + alpha * pe;
        }
        __syncthreads();
        rrn = red[16] + red[17] + red[18] + red[19];
        rz_new = rrn + cmu;
        const bool broke = !(pq > 0.0) || !(rrn == rrn);
        if (rrn <= tol2 * scal[PS_RR0] || broke) {
            if (blockIdx.x == 0 && tid == 0) {
                flags[PF_DONE] = seq + 1; flags[PF_XBUF] = out; const int it = flags[PF_ITERS] + 1; flags[PF_ITERS] = it;
                if (broke) atomicCAS(info, 0, d + 1);
                if (mailbox) pcg_post(mailbox, it, 1);
            }
            return;
        }
        const double beta = rz_new / rz_in;
#pragma unroll
        for (int k = 0; k < PCG_NW; ++k) pmu_new[k] = fma(beta, pmu_in[k], mu_new[k]);
        for (int e = row0 + tid; e < row1; e += 256) r_out[e] = pl[e];      // pl holds r_new
        __syncthreads();
        for (int e0 = tid; e0 < d; e0 += 256 * 8) {
            double pv8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int e = e0 + 256 * u; pv8[u] = p_in[e < d ? e : d - 1]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int e = e0 + 256 * u; if (e < d) pl[e] = pl[e] + beta * pv8[u]; }
        }
        __syncthreads();
        for (int e = row0 + tid; e < row1; e += 256) p_out[e] = pl[e];
        if (blockIdx.x == 0 && tid == 0) { const int it = flags[PF_ITERS] + 1; flags[PF_ITERS] = it; flags[PF_XBUF] = out; if (mailbox) pcg_post(mailbox, it, 0); }
    }
    if (blockIdx.x == 0 && tid == 0) {
        st_out[PS_RZ] = rz_new;
#pragma unroll
        for (int k = 0; k < PCG_NW; ++k) { st_out[PS_C + k] = c_new[k]; st_out[PS_MU + k] = mu_new[k]; st_out[PS_PMU + k] = pmu_new[k]; }
    }
    if (COARSE && tid < PCG_NW) red[32 + tid] = pmu_new[tid];
    __syncthreads();
    // q = S~ p_r + AW p_mu for the rows this workgroup owns: each wave takes rows (row0 + w + 4k), two at a time
    double pqp = 0.0, gacc = 0.0;
    const int nd2 = d >> 1, nd4 = d >> 2;
    for (int row = row0 + w; row < row1; row += 8) {
        const int rowb = (row + 4 < row1) ? row + 4 : row;
        double sa = 0.0, sb = 0.0;
        if (COARSE && lane < PCG_NW) {
            sa = AW[(size_t)row * PCG_NW + lane] * red[32 + lane];
            sb = AW[(size_t)rowb * PCG_NW + lane] * red[32 + lane];
        }
        if (sizeof(FT) == 8) {
            const double2* pl2 = reinterpret_cast<const double2*>(pl);
            const double2* Fa = reinterpret_cast<const double2*>(F + (size_t)row * ld);
            const double2* Fb = reinterpret_cast<const double2*>(F + (size_t)rowb * ld);
            int c = lane;
            for (; c + 192 < nd2; c += 256) {
                double2 a[4], b[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) { a[m] = Fa[c + 64 * m]; b[m] = Fb[c + 64 * m]; }
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const double2 pv = pl2[c + 64 * m];
                    sa += a[m].x * pv.x + a[m].y * pv.y;
                    sb += b[m].x * pv.x + b[m].y * pv.y;
                }
            }
            for (; c < nd2; c += 64) {
                const double2 a = Fa[c], b = Fb[c], pv = pl2[c];
                sa += a.x * pv.x + a.y * pv.y;
                sb += b.x * pv.x + b.y * pv.y;
            }
            if ((d & 1) && lane == 0) {
                sa += (double)F[(size_t)row * ld + d - 1] * pl[d - 1];
                sb += (double)F[(size_t)rowb * ld + d - 1] * pl[d - 1];
            }
        } else {
            // fp32 matrix: 16-byte loads of four columns, products and sums in fp64
            const float4* Fa = reinterpret_cast<const float4*>(F + (size_t)row * ld);
            const float4* Fb = reinterpret_cast<const float4*>(F + (size_t)rowb * ld);
            int c = lane;
            for (; c + 192 < nd4; c += 256) {
                float4 a[4], b[4];
                const bool first = pre && row == row0 + w && c == lane;       // (wave-uniform) the batch requested before the vector phase
                if (first) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) { a[m] = pre_a[m]; b[m] = pre_b[m]; }
                } else {
#pragma unroll
                    for (int m = 0; m < 4; ++m) { a[m] = Fa[c + 64 * m]; b[m] = Fb[c + 64 * m]; }
                }
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const double2 p0 = reinterpret_cast<const double2*>(pl)[2 * (c + 64 * m)], p1 = reinterpret_cast<const double2*>(pl)[2 * (c + 64 * m) + 1];
                    sa += (double)a[m].x * p0.x + (double)a[m].y * p0.y + (double)a[m].z * p1.x + (double)a[m].w * p1.y;
                    sb += (double)b[m].x * p0.x + (double)b[m].y * p0.y + (double)b[m].z * p1.x + (double)b[m].w * p1.y;
                }
            }
            for (; c < nd4; c += 64) {
                const float4 a = Fa[c], b = Fb[c];
                const double2 p0 = reinterpret_cast<const double2*>(pl)[2 * c], p1 = reinterpret_cast<const double2*>(pl)[2 * c + 1];
                sa += (double)a.x * p0.x + (double)a.y * p0.y + (double)a.z * p1.x + (double)a.w * p1.y;
                sb += (double)b.x * p0.x + (double)b.y * p0.y + (double)b.z * p1.x + (double)b.w * p1.y;
            }
            if (lane == 0) {
                for (int cc = 4 * nd4; cc < d; ++cc) {
                    sa += (double)F[(size_t)row * ld + cc] * pl[cc];
                    sb += (double)F[(size_t)rowb * ld + cc] * pl[cc];
                }
            }
        }
        { sa = wave_allsum(sa); sb = wave_allsum(sb); }
        if (lane == 0) {
            q_out[row] = sa; pqp += pl[row] * sa;
            if (rowb != row) { q_out[rowb] = sb; pqp += pl[rowb] * sb; }
        }
        if (COARSE && lane >= 8 && lane < 8 + PCG_NW) {
            gacc = fma(W[(size_t)(lane - 8) * ld + row], sa, gacc);
            if (rowb != row) gacc = fma(W[(size_t)(lane - 8) * ld + rowb], sb, gacc);
        }
    }
    if (lane == 0) red[40 + 9 * w] = pqp;
    if (COARSE && lane >= 8 && lane < 8 + PCG_NW) red[40 + 9 * w + 1 + (lane - 8)] = gacc;
    __syncthreads();
    if (tid < NV) pcg_part(part, out, tid)[blockIdx.x] = red[40 + tid] + red[49 + tid] + red[58 + tid] + red[67 + tid];
}

// ---------------------------------------------------------------------------------------------------------------------
// Symmetric streaming path (round 6).  S~ is symmetric: the CG reads ONE triangle of it per iteration -- 72 MB instead of
// 144 MB at d = 6001 -- and uses every entry twice from the one 16-byte load that brought it in:
//     q_i += U_ij p_j      (row use: summed across the wave by a halving butterfly)
//     q_j += U_ij p_i      (column use, j > i: kept per lane, summed across the four waves of the tile in LDS)
// over TILES of SY_R rows x SY_C columns of the scalar upper triangle (diagonal included; the lower triangle and the padding are never multiplied),
// ONE TILE PER WORKGROUP: every load of the product is in flight as soon as the launch starts (2 396 workgroups at d = 6001).  A row of q collects
// sums from every tile of its row strip and of its column chunk: fp64 device-scope atomics on a zeroed q buffer, every atomic instruction on
// CONSECUTIVE addresses (tools/micro/symv_bench.hip: the memory side retires ~6 G line-sized atomic transactions per second whatever they carry;
// 48 k of them per product cost nothing next to the stream -- 14.7 us = 4.9 TB/s of the 72 MB under rocprofv3, against 27 - 32 us for both triangles --,
// the 2 M of a form with one flush per wave and strided lanes cost 100 us).  In the solve: 17.9 us per launch, of which 15.0 the bare product and 1.8 the
// epilogue below (profiles/r06_ab_sy_prod_bisect.txt: occupancy, the prologue's scalar round trips, an unmasked path for interior tiles do not move it).
// What does NOT survive a grid of 2 396 workgroups is the vector phase of k_pcg_iter (every workgroup forms alpha, |r - alpha q|^2, beta from the
// whole vectors: 96 KB from L2 per workgroup -- measured +5 ... +10 us at 512 ... 1 024 workgroups in the same micro-benchmark, fused forms of the
// iteration land at 20 - 26 us).  So an iteration is TWO launches here:
//   k_sy_vec   (SY_VEC_WG workgroups) the scalars of the iteration, x, r, the new search direction p = p_r (its own slice of each), the stopping test;
//              zeroes its slice of the q buffer and of the partial sums the product is about to add into
//   k_sy_prod  (one workgroup per tile) the product; the slotted partial sums p_r . q; its first workgroups also add (S~ W~) p_mu to their slice of q
//              and form the partials of W~^T q = (S~ W~)^T p_r (+ E p_mu in k_sy_vec): S~ is symmetric, so W~^T S~ p_r needs no pass over W~ per tile
// q and the partial sums are double-buffered by iteration parity like the other vectors.  Not for deterministic handles (the order the atomics
// arrive in is not fixed): those keep k_pcg_iter.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int SY_R = 32;              // rows of a tile: eight per wave, one batch of eight 16-byte loads in flight per lane
constexpr int SY_C = 256;             // columns of a tile: 64 lanes x four entries
constexpr int SY_VEC_WG = 64;         // workgroups of the vector kernel (each forms the scalars for itself and updates its slice)
constexpr int SY_SLOTS = 64;          // slots of the partial sums (one 128-byte line per value and slot: atomics on one line serialise)
constexpr int SY_SLOT_STRIDE = 16;    // doubles between two accumulators

// sums of eight per-lane values over the wave: afterwards every lane of the group lane / 8 holds the total of value lane / 8
__device__ __forceinline__ double rows8_reduce(const double (&v)[8], int lane) {
    double a[4], b[2];
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] = xlane_pairsum<32>(v[k], v[4 + k]);
#pragma unroll
    for (int k = 0; k < 2; ++k) b[k] = xlane_pairsum<16>(a[k], a[2 + k]);
    const bool up = (lane & 8) != 0;
    const double send = up ? b[0] : b[1], keep = up ? b[1] : b[0];
    double c = keep + xlane_get<8>(send);
    c = xlane_add<4>(c); c = xlane_add<2>(c); c = xlane_add<1>(c);
    return c;
}
__device__ __forceinline__ double* sy_slot(double* part, int parity, int v, int slot) { return part + ((size_t)(parity * PCG_NPART + v) * SY_SLOTS + slot) * SY_SLOT_STRIDE; }

// the vector half of an iteration (see above).  q2: [2][ld] the products by parity; part: [2][9][SY_SLOTS] accumulators, one line each
// EPT > 0: the whole of r and q in registers (EPT entries per thread, 256 EPT >= d), requested together with everything else the launch reads BEFORE
// the first reduction -- one memory round trip per launch instead of one per phase; EPT = 0: looped (any d)
template <bool INIT, bool COARSE, int EPT>
__global__ __launch_bounds__(256) void k_sy_vec(int d, int ld, double* __restrict__ vec, double* __restrict__ q2, const double* __restrict__ bt,
                                                double* __restrict__ part, double* scal, int* flags, double tol2, int in, int* info, int* mailbox,
                                                int anchor, double cap, const double* __restrict__ W, const double* __restrict__ coarse) {
    __shared__ double red[PCG_RED];
    const int seq = in >> 1;
    in &= 1;
    if (!INIT) { const int dn = flags[PF_DONE]; if (dn != 0 && seq >= dn) return; }
    constexpr int NV = COARSE ? PCG_NPART : 1;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, out = in ^ 1;
    const int nwg = (int)gridDim.x;
    const int per = (d + nwg - 1) / nwg;
    const int row0 = min(d, (int)blockIdx.x * per), row1 = min(d, row0 + per);
    double* x_out = pcg_vec(vec, 0, out, ld); double* r_out = pcg_vec(vec, 1, out, ld); double* p_out = pcg_vec(vec, 2, out, ld);
    const double* x_in = pcg_vec(vec, 0, in, ld); const double* r_in = INIT ? bt : pcg_vec(vec, 1, in, ld);
    const double* p_in = pcg_vec(vec, 2, in, ld);
    const double* q_in = q2 + (size_t)in * ld;
    double* q_next = q2 + (size_t)out * ld;         // the product of THIS iteration is added into it
    const double* st_in = scal + PS_STATE + PS_STATE_LEN * in;
    double* st_out = scal + PS_STATE + PS_STATE_LEN * out;
    if (COARSE && tid < PCG_NW * PCG_NW) { red[80 + tid] = coarse[tid]; red[144 + tid] = coarse[PCG_NW * PCG_NW + PCG_NW + tid]; }      // E^-1, E
    // the buffers the product adds into: zero (this workgroup's slice of q, its share of the accumulators)
    for (int e = row0 + tid; e < row1; e += 256) q_next[e] = 0.0;
    for (int i = blockIdx.x * 256 + tid; i < NV * SY_SLOTS; i += nwg * 256) sy_slot(part, out, 0, 0)[(size_t)i * SY_SLOT_STRIDE] = 0.0;
    double c_new[PCG_NW], mu_new[PCG_NW], pmu_new[PCG_NW], pmu_in[PCG_NW];
    double rz_new = 0.0;
    if (INIT) {
        double rr = 0.0;
        for (int e0 = tid; e0 < d; e0 += 256 * 8) {
            double bv8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int e = e0 + 256 * u; bv8[u] = bt[e < d ? e : d - 1]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int e = e0 + 256 * u; if (e < d) rr += bv8[u] * bv8[u]; }
        }
        rr = wave_allsum(rr);
        if (lane == 0) red[16 + w] = rr;
        __syncthreads();
        rr = red[16] + red[17] + red[18] + red[19];
#pragma unroll
        for (int k = 0; k < PCG_NW; ++k) { c_new[k] = COARSE ? coarse[PCG_NW * PCG_NW + k] : 0.0; mu_new[k] = 0.0; pmu_in[k] = 0.0; }
        if (COARSE) einv_apply(red + 80, c_new, mu_new);
        rz_new = rr + (COARSE ? dot8(c_new, mu_new) : 0.0);
#pragma unroll
        for (int k = 0; k < PCG_NW; ++k) pmu_new[k] = mu_new[k];
        for (int e = row0 + tid; e < row1; e += 256) { const double b = bt[e]; x_out[e] = 0.0; r_out[e] = b; p_out[e] = b; }
        if (blockIdx.x == 0 && tid == 0) {
            scal[PS_RR0] = pcg_threshold_base(rr, scal, anchor, cap); flags[PF_DONE] = (rr == 0.0); flags[PF_ITERS] = 0; flags[PF_XBUF] = out;
            if (mailbox && rr == 0.0) pcg_post(mailbox, 0, 1);
        }
    } else {
        // the slotted sums of the previous product: wave w owns values w, w + 4, w + 8; one slot per lane
        double mine[3] = { 0.0, 0.0, 0.0 };
#pragma unroll
        for (int j = 0; j < 3; ++j) { const int v = w + 4 * j; mine[j] = (4 * j < NV && v < NV) ? *sy_slot(part, in, v, lane) : 0.0; }
        // ... and every other load of the launch, before anything waits: r and q (whole), this thread's entry of the slice
        double rv[EPT > 0 ? EPT : 1], qv[EPT > 0 ? EPT : 1];
        if (EPT > 0) {
#pragma unroll
            for (int m = 0; m < EPT; ++m) { const int e = tid + 256 * m, ec = e < d ? e : d - 1; rv[m] = r_in[ec]; qv[m] = q_in[ec]; }
        }
        const int eo = row0 + tid;
        const bool own = eo < row1;                      // (per <= 256: one entry per thread; longer slices loop below)
        const int eoc = own ? eo : 0;
        const double xo = x_in[eoc], po = p_in[eoc], ro = r_in[eoc], qo = q_in[eoc];
        double wo[PCG_NW];
#pragma unroll
        for (int k = 0; k < PCG_NW; ++k) wo[k] = COARSE ? W[(size_t)k * ld + eoc] : 0.0;
        const double rz_in = st_in[PS_RZ], rr0 = scal[PS_RR0];
        double g[PCG_NW], Eg[PCG_NW], c_in[PCG_NW], mu_in[PCG_NW];
#pragma unroll
        for (int k = 0; k < PCG_NW; ++k) {
            pmu_in[k] = COARSE ? st_in[PS_PMU + k] : 0.0;
            c_in[k] = COARSE ? st_in[PS_C + k] : 0.0;
            mu_in[k] = COARSE ? st_in[PS_MU + k] : 0.0;
            Eg[k] = 0.0;
        }
        reduce_partials<NV>(mine, red);
        __syncthreads();
        // W~^T q = (S~ W~)^T p_r (the partials) + E p_mu
#pragma unroll
        for (int k = 0; k < PCG_NW; ++k) {
            double erow[PCG_NW];
#pragma unroll
            for (int j = 0; j < PCG_NW; ++j) erow[j] = COARSE ? red[144 + PCG_NW * k + j] : 0.0;
            g[k] = COARSE ? red[1 + k] + dot8(erow, pmu_in) : 0.0;
        }
        const double pq = red[0] + (COARSE ? dot8(pmu_in, g) : 0.0);
        const double alpha = rz_in * fast_rcp(pq);
        if (COARSE) einv_apply(red + 80, g, Eg);         // independent of alpha: overlaps the reciprocal
#pragma unroll
        for (int k = 0; k < PCG_NW; ++k) { c_new[k] = fma(-alpha, g[k], c_in[k]); mu_new[k] = fma(-alpha, Eg[k], mu_in[k]); }
        const double cmu = COARSE ? dot8(c_new, mu_new) : 0.0;
        double rrn = 0.0;
        if (EPT > 0) {
#pragma unroll
            for (int m = 0; m < EPT; ++m) { const double v = fma(-alpha, qv[m], rv[m]); rrn = (tid + 256 * m < d) ? fma(v, v, rrn) : rrn; }
        } else {
            for (int e0 = tid; e0 < d; e0 += 256 * 8) {
                double rv8[8], qv8[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int e = e0 + 256 * u, ec = e < d ? e : d - 1; rv8[u] = r_in[ec]; qv8[u] = q_in[ec]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int e = e0 + 256 * u; if (e < d) { const double v = fma(-alpha, qv8[u], rv8[u]); rrn = fma(v, v, rrn); } }
            }
        }
        rrn = wave_allsum(rrn);
        if (lane == 0) red[16 + w] = rrn;
        // x += alpha p  with p = p_r + W~ p_mu, own slice
        if (own) x_out[eo] = xo + alpha * (po + (COARSE ? dot8(wo, pmu_in) : 0.0));
        for (int e = eo + 256; e < row1; e += 256) {
            double pe = p_in[e];
            if (COARSE) {
#pragma unroll
                for (int k = 0; k < PCG_NW; ++k) pe = fma(W[(size_t)k * ld + e], pmu_in[k], pe);
            }
            x_out[e] = x_in[e] + alpha * pe;
        }
        __syncthreads();
        rrn = red[16] + red[17] + red[18] + red[19];
        rz_new = rrn + cmu;
        const bool broke = !(pq > 0.0) || !(rrn == rrn);
        if (rrn <= tol2 * rr0 || broke) {
            if (blockIdx.x == 0 && tid == 0) {
                flags[PF_DONE] = seq + 1; flags[PF_XBUF] = out; const int it = flags[PF_ITERS] + 1; flags[PF_ITERS] = it;
                if (broke) atomicCAS(info, 0, d + 1);
                if (mailbox) pcg_post(mailbox, it, 1);
            }
            return;
        }
        const double beta = rz_new * fast_rcp(rz_in);
#pragma unroll
        for (int k = 0; k < PCG_NW; ++k) pmu_new[k] = fma(beta, pmu_in[k], mu_new[k]);
        if (own) { const double rn = fma(-alpha, qo, ro); r_out[eo] = rn; p_out[eo] = fma(beta, po, rn); }
        for (int e = eo + 256; e < row1; e += 256) {
            const double rn = fma(-alpha, q_in[e], r_in[e]);
            r_out[e] = rn;
            p_out[e] = fma(beta, p_in[e], rn);
        }
        if (blockIdx.x == 0 && tid == 0) { const int it = flags[PF_ITERS] + 1; flags[PF_ITERS] = it; flags[PF_XBUF] = out; if (mailbox) pcg_post(mailbox, it, 0); }
    }
    if (blockIdx.x == 0 && tid == 0) {
        st_out[PS_RZ] = rz_new;
#pragma unroll
        for (int k = 0; k < PCG_NW; ++k) { st_out[PS_C + k] = c_new[k]; st_out[PS_MU + k] = mu_new[k]; st_out[PS_PMU + k] = pmu_new[k]; }
    }
}

// the product half: tiles[blockIdx.x] = {first row, first column (a multiple of SY_C), rows, 1 if the tile overlaps the diagonal}.  `in` as k_sy_vec's
// (the launch pair of one iteration shares it): p_r and the state of parity out = in ^ 1 are what k_sy_vec just wrote.
template <typename FT, bool COARSE>
__global__ __launch_bounds__(256) void k_sy_prod(int d, int ld, const FT* __restrict__ F, const double* __restrict__ vec, double* __restrict__ q2,
                                                 double* __restrict__ part, const double* __restrict__ scal, const int* __restrict__ flags,
                                                 const int4* __restrict__ tiles, int in, int nslice, const double* __restrict__ AW) {
    __shared__ __align__(16) double colsh[4 * SY_C];
    __shared__ double red[4 * PCG_NPART];
    const int seq = in >> 1;
    in &= 1;
    { const int dn = flags[PF_DONE]; if (dn != 0 && seq + 1 >= dn) return; }       // (k_sy_vec of this iteration has raised it: nothing left to multiply)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, out = in ^ 1;
    const double* p = vec + (size_t)(2 * 2 + out) * ld;        // pcg_vec(vec, 2, out, ld)
    double* q = q2 + (size_t)out * ld;
    const int4 tl = tiles[blockIdx.x];
    const int r0 = tl.x, c0 = tl.y, rend = tl.x + tl.z;
    const bool diag = tl.w != 0;
    const int j0 = c0 + 4 * lane, rw0 = r0 + 8 * w;
    Quad<FT> f[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int row = rw0 + u; f[u].load(F + (size_t)(row < rend ? row : r0) * ld + j0); }
    double pj[4];
    {
        const double2 a = reinterpret_cast<const double2*>(p + j0)[0], b = reinterpret_cast<const double2*>(p + j0)[1];      // (j0 + 3 < ld: the padding is selected away)
        pj[0] = j0 + 0 < d ? a.x : 0.0; pj[1] = j0 + 1 < d ? a.y : 0.0; pj[2] = j0 + 2 < d ? b.x : 0.0; pj[3] = j0 + 3 < d ? b.y : 0.0;
    }
    const int myrow = rw0 + (lane >> 3);              // the row whose total this lane group publishes
    const double prow = myrow < rend ? p[myrow] : 0.0;
    double pq_acc = 0.0;
    double racc[8], colacc[4] = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int row = rw0 + u;                      // wave-uniform
        const double pi = lane_bcast(prow, 8 * u);    // (0 for a row beyond the tile)
        const bool live = row < rend;
        double s = 0.0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = j0 + e;
            double x = f[u].get(e);
            if (!live || j >= d || (diag && j < row)) x = 0.0;
            s = fma(x, pj[e], s);
            colacc[e] = fma((diag && j == row) ? 0.0 : x, pi, colacc[e]);       // the diagonal entry is used once
        }
        racc[u] = s;
    }
    const double tot = rows8_reduce(racc, lane);
    if ((lane & 7) == 0 && myrow < rend) { atomicAdd(q + myrow, tot); pq_acc = prow * tot; }
#pragma unroll
    for (int e = 0; e < 4; ++e) colsh[w * SY_C + 4 * lane + e] = colacc[e];
    __syncthreads();
    if (c0 + tid < d) {
        const double cs = (colsh[tid] + colsh[SY_C + tid]) + (colsh[2 * SY_C + tid] + colsh[3 * SY_C + tid]);
        atomicAdd(q + c0 + tid, cs);
        pq_acc = fma(p[c0 + tid], cs, pq_acc);
    }
    // ---- the first nslice workgroups: q += (S~ W~) p_mu on their slice, and the partials of (S~ W~)^T p_r ----
    double gpart[PCG_NW];
#pragma unroll
    for (int k = 0; k < PCG_NW; ++k) gpart[k] = 0.0;
    const bool slice = COARSE && (int)blockIdx.x < nslice;
    if (slice) {
        const int per = (d + nslice - 1) / nslice;
        const int e0 = min(d, (int)blockIdx.x * per), e1 = min(d, e0 + per);
        const double* st = scal + PS_STATE + PS_STATE_LEN * out;
        double pmu[PCG_NW];
#pragma unroll
        for (int k = 0; k < PCG_NW; ++k) pmu[k] = st[PS_PMU + k];
        for (int e = e0 + tid; e < e1; e += 256) {
            const double pe = p[e];
            double aw[PCG_NW];
#pragma unroll
            for (int k = 0; k < PCG_NW; ++k) aw[k] = AW[(size_t)k * ld + e];        // (AWt: vector-major)
            const double a = dot8(aw, pmu);
            atomicAdd(q + e, a);
            pq_acc = fma(pe, a, pq_acc);
#pragma unroll
            for (int k = 0; k < PCG_NW; ++k) gpart[k] = fma(aw[k], pe, gpart[k]);
        }
    }
    pq_acc = wave_allsum(pq_acc);
    if (slice) {
#pragma unroll
        for (int k = 0; k < PCG_NW; ++k) gpart[k] = wave_allsum(gpart[k]);
    }
    if (lane == 0) {
        red[PCG_NPART * w] = pq_acc;
#pragma unroll
        for (int k = 0; k < PCG_NW; ++k) red[PCG_NPART * w + 1 + k] = gpart[k];
    }
    __syncthreads();
    if (tid == 0 || (slice && tid < PCG_NPART)) {
        const double v = (red[tid] + red[PCG_NPART + tid]) + (red[2 * PCG_NPART + tid] + red[3 * PCG_NPART + tid]);
        atomicAdd(sy_slot(part, out, tid, (int)(blockIdx.x % SY_SLOTS)), v);
    }
}

// The coarse set-up on the same triangle: AW = S~ W~ for the eight gauge vectors at once, every entry of the triangle used sixteen times from its
// one load.  With eight vectors the atomics are what has to be budgeted (the memory side retires ~6 G line-sized atomic transactions per second,
// see above): TALL tiles, SY_CR rows x SY_C columns -- the eight column sums of a lane's four columns stay in registers over the wave's 32 rows
// and leave through LDS (ds_add_f64 across the four waves), the row sums of a wave's rows are parked in LDS until the tile is done -- so a tile
// issues 8 x (256 + 128) atomics as 192 instructions on consecutive addresses (~115 k lines per pass at d = 6001).  The result is vector-major,
// AWt[k][i] (consecutive lanes = consecutive entries), zeroed by the linearisation (k_finalize: DeviceBuffers::pcg_zero).
#ifndef SFMBA_SY_CR
#define SFMBA_SY_CR 128
#endif
constexpr int SY_CR = SFMBA_SY_CR;
// (Measured and not kept, profiles/r06_ab_sy_coarse.txt: the batch specialised on the diagonal flag with the prefetch peeled -- four copies of the loop,
// 86 spilled registers, 86 us --; W~ of the lane's columns held as floats with the widening kept opaque -- 18 spills, 58 us; this form, one spill: 51 us.
// The pass is bound by its instruction count, ~1 800 per eight rows of 256 columns -- 512 fp64 FMAs, 330 v_readlane for W~ of the rows, ~300 for the
// eight halving reductions, the masks -- at two waves per SIMD: the 72 MB stream at 1.4 TB/s.)
template <typename FT>
__global__ __launch_bounds__(256, 2) void k_sy_coarse(int d, int ld, const FT* __restrict__ F, const double* __restrict__ W, double* __restrict__ AWt,
                                                      const int4* __restrict__ tiles) {
    __shared__ double colsh[PCG_NW * SY_C];      // [k][column]
    __shared__ double rowsh[PCG_NW * SY_CR];     // [k][row]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int4 tl = tiles[blockIdx.x];
    const int r0 = tl.x, c0 = tl.y, rend = tl.x + tl.z;
    const bool diag = tl.w != 0;
    const int j0 = c0 + 4 * lane;
    for (int i = tid; i < PCG_NW * SY_C; i += 256) colsh[i] = 0.0;
    double wj[4][PCG_NW], colacc[4][PCG_NW];
#pragma unroll
    for (int k = 0; k < PCG_NW; ++k) {
        const double2 a = reinterpret_cast<const double2*>(W + (size_t)k * ld + j0)[0], b = reinterpret_cast<const double2*>(W + (size_t)k * ld + j0)[1];
        wj[0][k] = j0 + 0 < d ? a.x : 0.0; wj[1][k] = j0 + 1 < d ? a.y : 0.0; wj[2][k] = j0 + 2 < d ? b.x : 0.0; wj[3][k] = j0 + 3 < d ? b.y : 0.0;
#pragma unroll
        for (int e = 0; e < 4; ++e) colacc[e][k] = 0.0;
    }
    constexpr int RPWV = SY_CR / 4, NBATCH = RPWV / 8;
    const int rw = r0 + RPWV * w;                     // this wave's rows, eight at a time
    Quad<FT> f[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int row = rw + u; f[u].load(F + (size_t)(row < rend ? row : r0) * ld + j0); }
#pragma unroll 1
    for (int b = 0; b < NBATCH; ++b) {
        const int rb = rw + 8 * b;
        // W~ of the batch's rows: lane l holds vector l % 8 of row l / 8; a row's eight values reach the FMAs as scalar operands (v_readlane)
        const int wr = rb + (lane >> 3);
        const double wrow = wr < rend ? W[(size_t)(lane & 7) * ld + wr] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int row = rb + u;
            const bool live = row < rend;
            double x[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { const int j = j0 + e; x[e] = (!live || j >= d || (diag && j < row)) ? 0.0 : f[u].get(e); }
            if (b < NBATCH - 1) { const int nrow = rb + 8 + u; f[u].load(F + (size_t)(nrow < rend ? nrow : r0) * ld + j0); }      // the next batch's load of this slot goes out at once
            double sacc[PCG_NW];
#pragma unroll
            for (int k = 0; k < PCG_NW; ++k) {
                const double wi = lane_bcast(wrow, 8 * u + k);
                double sk = 0.0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    sk = fma(x[e], wj[e][k], sk);
                    colacc[e][k] = fma((diag && j0 + e == row) ? 0.0 : x[e], wi, colacc[e][k]);       // the diagonal entry is used once
                }
                sacc[k] = sk;
            }
            const double tot = rows8_reduce(sacc, lane);          // lanes of group g: vector g of this row
            if ((lane & 7) == 0) rowsh[(lane >> 3) * SY_CR + RPWV * w + 8 * b + u] = tot;
        }
    }
    __syncthreads();                                  // colsh zeroed, rowsh complete
#pragma unroll
    for (int k = 0; k < PCG_NW; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(&colsh[k * SY_C + 4 * lane + e], colacc[e][k]);
    // the row sums: consecutive threads = consecutive rows of one vector
#pragma unroll
    for (int m = 0; m < PCG_NW * SY_CR / 256; ++m) {
        const int v = tid + 256 * m, k = v / SY_CR, row = r0 + (v % SY_CR);
        if (row < rend) atomicAdd(AWt + (size_t)k * ld + row, rowsh[v]);
    }
    __syncthreads();
    if (c0 + tid < d) {
#pragma unroll
        for (int k = 0; k < PCG_NW; ++k) atomicAdd(AWt + (size_t)k * ld + c0 + tid, colsh[k * SY_C + tid]);
    }
}

// E = W~^T (S~ W~) and c_0 = W~^T b~ from the finished AWt: partials per workgroup into epart (the layout k_pcg_coarse_invert sums)
__global__ __launch_bounds__(256) void k_sy_e(int d, int ld, const double* __restrict__ W, const double* __restrict__ AWt, const double* __restrict__ bt,
                                              double* __restrict__ epart) {
    __shared__ double sh[4][PCG_NW * PCG_NW + PCG_NW];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int per = (d + gridDim.x - 1) / gridDim.x, i0 = blockIdx.x * per, i1 = min(d, i0 + per);
    double e[PCG_NW][PCG_NW], c[PCG_NW];
#pragma unroll
    for (int k = 0; k < PCG_NW; ++k) { c[k] = 0.0;
#pragma unroll
        for (int l = 0; l < PCG_NW; ++l) e[k][l] = 0.0; }
    for (int i = i0 + tid; i < i1; i += 256) {
        double wv[PCG_NW], av[PCG_NW];
#pragma unroll
        for (int k = 0; k < PCG_NW; ++k) { wv[k] = W[(size_t)k * ld + i]; av[k] = AWt[(size_t)k * ld + i]; }
        const double b = bt[i];
#pragma unroll
        for (int k = 0; k < PCG_NW; ++k) { c[k] = fma(wv[k], b, c[k]);
#pragma unroll
            for (int l = 0; l < PCG_NW; ++l) e[k][l] = fma(wv[k], av[l], e[k][l]); }
    }
#pragma unroll
    for (int k = 0; k < PCG_NW; ++k) { c[k] = wave_allsum(c[k]);
#pragma unroll
        for (int l = 0; l < PCG_NW; ++l) e[k][l] = wave_allsum(e[k][l]); }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < PCG_NW; ++k) { sh[w][PCG_NW * PCG_NW + k] = c[k];
#pragma unroll
            for (int l = 0; l < PCG_NW; ++l) sh[w][k * PCG_NW + l] = e[k][l]; }
    }
    __syncthreads();
    if (tid < PCG_NW * PCG_NW + PCG_NW) epart[(size_t)tid * PCG_PART + blockIdx.x] = (sh[0][tid] + sh[1][tid]) + (sh[2][tid] + sh[3][tid]);
}

// Fast path of one CG iteration for d <= 1280 (all BASELINE single-GPU configs): every global load of
// the iteration -- the three length-d vectors, the partial sums and this wave's rows of S~ --
// is issued up front into registers, so the launch pays ONE memory round trip; the rest is LDS + ALU.
// The special roles are spread over the waves (wave 0..3: partial sums; last wave: this workgroup's own rows of x, r, p_r) and
// everything about the 8-dimensional coarse space that does not need alpha (E^-1 g) is formed while alpha is on its way.
constexpr int PCG_EPT = 5;    // vector elements per thread  (256 * 5 >= d)
constexpr int PCG_RPW = 2;    // rows of S~ per wave         (rows_per_wg <= 8)
constexpr int PCG_CPL = 20;   // columns per lane            (64 * 20 >= d)

// MODE 0: an iteration.  MODE 1: the first launch of a solve without a coarse space (x0 = 0, r0 = p0 = b~, the product).  MODE 2: the first launch
// WITH the coarse space, which is also the first ITERATION: k_pcg_coarse_fast left t = S~ b~ (in the q buffer), AW, and the partials of E, c_0 and
// W~^T t; with p_r0 = b~ and p_mu0 = E^-1 c_0 the first product is q_0 = t + AW p_mu0 and W~^T q_0 = W~^T t + E p_mu0 -- nothing of it needs a pass
// over S~, so the launch that used to do only the initialisation and that product (10.7 us) is gone and this one (E^-1 by one wave, then a regular
// iteration) takes its place.
template <int MODE, bool COARSE>
__global__ __launch_bounds__(256) void k_pcg_iter_fast(int d, int ld, const double* __restrict__ F, double* __restrict__ vec,
                                                       const double* __restrict__ bt, double* __restrict__ part, double* __restrict__ scal,
                                                       int* flags, int rows_per_wg, double tol2, int in, int* info, int* mailbox, int anchor, double cap,
                                                       const double* __restrict__ W, const double* __restrict__ AW, double* __restrict__ coarse,
                                                       const double* __restrict__ epart) {
    extern __shared__ __align__(16) double sm[];
    double* pl = sm;
    double* red = sm + ld;
    // `in` = (launch number << 1) | parity.  PF_DONE holds the first launch number that has nothing left to do: a launch must not act
    // on the flag its own workgroup 0 raises (workgroups that start late, e.g. behind another process's kernels, would skip the
    // converging iteration's x update).
    constexpr bool INIT = MODE == 1, FIRST = MODE == 2;
    static_assert(!FIRST || COARSE, "the merged first launch exists for the coarse space only");
    const int seq = in >> 1;
    in &= 1;
    if (MODE == 0) { const int dn = flags[PF_DONE]; if (dn != 0 && seq >= dn) return; }
    constexpr int NV = COARSE ? PCG_NPART : 1;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, out = in ^ 1;
    // first launch of a solve: E^-1 and c_0 from the partials k_pcg_coarse_fast left behind -- formed by EVERY workgroup for
    // itself (no launch of its own: the 8 x 8 inversion is ~3 us of one wave), published by workgroup 0 for the launches that follow
    if (INIT && COARSE) coarse_sum_partials((int)gridDim.x, epart, red + 144);
    if (FIRST) coarse_sum_partials<20>((int)gridDim.x, epart, red + 144);
    const int row0 = blockIdx.x * rows_per_wg;
    const int row1 = min(d, row0 + rows_per_wg);
    const int nwg = (int)gridDim.x;
    const double* x_in = pcg_vec(vec, 0, in, ld); const double* r_in = (INIT || FIRST) ? bt : pcg_vec(vec, 1, in, ld);
    const double* p_in = FIRST ? bt : pcg_vec(vec, 2, in, ld); const double* q_in = pcg_vec(vec, 3, in, ld);
    double* x_out = pcg_vec(vec, 0, out, ld); double* r_out = pcg_vec(vec, 1, out, ld);
    double* p_out = pcg_vec(vec, 2, out, ld); double* q_out = pcg_vec(vec, 3, out, ld);
    const double* st_in = scal + PS_STATE + PS_STATE_LEN * in;
    double* st_out = scal + PS_STATE + PS_STATE_LEN * out;

    // ---- all global loads of this iteration ----
    double rv[PCG_EPT], qv[PCG_EPT], pv[PCG_EPT];
#pragma unroll
    for (int m = 0; m < PCG_EPT; ++m) {
        const int e = tid + 256 * m;
        const bool ok = e < d;
        rv[m] = ok ? r_in[e] : 0.0; qv[m] = (ok && !INIT) ? q_in[e] : 0.0; pv[m] = FIRST ? rv[m] : (ok && !INIT) ? p_in[e] : 0.0;      // (first launch: p_r0 = r_0 = b~)
    }
    double mine[3] = { 0.0, 0.0, 0.0 };           // partial sums of the previous launch: wave w owns values w, w + 4, w + 8
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int v = w + 4 * j;
        if (MODE == 0 && 4 * j < NV) {
            // branch-free, clamped: a conditional `+= load` makes the compiler wait for every load in turn (measured: twelve
            // dependent memory round trips, +3 us per iteration)
            const double* pp = pcg_part(part, in, v < NV ? v : 0);
            double t[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { const int wg = lane + 64 * i; t[i] = pp[wg < nwg ? wg : nwg - 1]; }
#pragma unroll
            for (int i = 0; i < 4; ++i) mine[j] += (lane + 64 * i < nwg && v < NV) ? t[i] : 0.0;
        }
    }
    const double einv_mine = (COARSE && MODE == 0 && tid < PCG_NW * PCG_NW) ? coarse[tid] : 0.0;
    double rr0 = MODE == 0 ? scal[PS_RR0] : 0.0;
    double rz_in = MODE == 0 ? st_in[PS_RZ] : 0.0;
    const double rrf = (FIRST && anchor == 2) ? scal[PS_RRF] : 0.0;
    double c_in[PCG_NW], mu_in[PCG_NW], pmu_in[PCG_NW];
#pragma unroll
    for (int k = 0; k < PCG_NW; ++k) {
        c_in[k] = (COARSE && MODE == 0) ? st_in[PS_C + k] : 0.0;
        mu_in[k] = (COARSE && MODE == 0) ? st_in[PS_MU + k] : 0.0;
        pmu_in[k] = (COARSE && MODE == 0) ? st_in[PS_PMU + k] : 0.0;
    }
    // own rows (the last wave's lanes < rows): what the x update and the stores of r, p_r need
    const int eo = row0 + (tid - 192);
    const bool own = tid >= 192 && eo < row1;
    double xo = 0.0, po = 0.0, ro = 0.0, qo = 0.0, wo[PCG_NW];
    if (own) { ro = r_in[eo]; if (MODE == 0) { xo = x_in[eo]; po = p_in[eo]; qo = q_in[eo]; } if (FIRST) { po = ro; qo = q_in[eo]; } }
#pragma unroll
    for (int k = 0; k < PCG_NW; ++k) wo[k] = (COARSE && own && !INIT) ? W[(size_t)k * ld + eo] : 0.0;
    double aw5[FIRST ? PCG_EPT : 1][PCG_NW], awo[PCG_NW];     // first launch: AW[e][:] for q_0 = t + AW p_mu0
    if (FIRST) {
#pragma unroll
        for (int m = 0; m < PCG_EPT; ++m) {
            const int e = tid + 256 * m;
#pragma unroll
            for (int k = 0; k < PCG_NW; ++k) { const double v = AW[(size_t)(e < d ? e : 0) * PCG_NW + k]; aw5[m][k] = e < d ? v : 0.0; }
        }
#pragma unroll
        for (int k = 0; k < PCG_NW; ++k) awo[k] = own ? AW[(size_t)eo * PCG_NW + k] : 0.0;
    }
    double2 fv[PCG_RPW][PCG_CPL / 2];          // 16-byte loads: lane takes columns 2*(lane + 64 m), +1
    double awv[PCG_RPW];                       // lanes 0..7: AW[row][lane]; lanes 8..15: W~[lane - 8][row]
#pragma unroll
    for (int k = 0; k < PCG_RPW; ++k) {
        const int row = row0 + w + 4 * k;
        const double2* Fr = reinterpret_cast<const double2*>(F + (size_t)(row < row1 ? row : row0) * ld);
#pragma unroll
        for (int m = 0; m < PCG_CPL / 2; ++m) {
            const int c2 = lane + 64 * m;
            double2 v = make_double2(0.0, 0.0);
            if (row < row1 && 2 * c2 < d) v = Fr[c2];
            if (2 * c2 + 1 >= d) v.y = 0.0;                 // padding column: never multiply garbage
            fv[k][m] = v;
        }
        awv[k] = 0.0;
        if (COARSE && row < row1 && lane < 2 * PCG_NW) awv[k] = lane < PCG_NW ? AW[(size_t)row * PCG_NW + lane] : W[(size_t)(lane - PCG_NW) * ld + row];
    }
    if (COARSE && MODE == 0 && tid < PCG_NW * PCG_NW) red[80 + tid] = einv_mine;
    double c_new[PCG_NW], mu_new[PCG_NW], pmu_new[PCG_NW];
    double rz_new;
    if (INIT) {
        if (COARSE) {
            __syncthreads();                              // E, c_0 complete in red[144 ..)
            if (w == 0) {
                const double e = coarse_invert_wave(red + 144, red + 224, red + 288);
                red[80 + lane] = e;
                if (blockIdx.x == 0) { coarse[lane] = e; if (lane < PCG_NW) coarse[PCG_NW * PCG_NW + lane] = red[144 + PCG_NW * PCG_NW + lane]; }
            }
        }
        // x0 = 0, r0 = b~, z0 = r0 + W~ E^-1 c0, p0 = z0
        double rr = 0.0;
#pragma unroll
        for (int m = 0; m < PCG_EPT; ++m) rr += rv[m] * rv[m];
        rr = wave_allsum(rr);
        if (lane == 0) red[16 + w] = rr;
        __syncthreads();
        rr = red[16] + red[17] + red[18] + red[19];
#pragma unroll
        for (int k = 0; k < PCG_NW; ++k) { c_new[k] = COARSE ? red[144 + PCG_NW * PCG_NW + k] : 0.0; mu_new[k] = 0.0; }
        if (COARSE) einv_apply(red + 80, c_new, mu_new);
        rz_new = rr + (COARSE ? dot8(c_new, mu_new) : 0.0);
#pragma unroll
        for (int k = 0; k < PCG_NW; ++k) pmu_new[k] = mu_new[k];
#pragma unroll
        for (int m = 0; m < PCG_EPT; ++m) { const int e = tid + 256 * m; if (e < d) pl[e] = rv[m]; }
        if (own) { x_out[eo] = 0.0; r_out[eo] = ro; p_out[eo] = ro; }
        if (blockIdx.x == 0 && tid == 0) {
            scal[PS_RR0] = pcg_threshold_base(rr, scal, anchor, cap); flags[PF_DONE] = (rr == 0.0); flags[PF_ITERS] = 0; flags[PF_XBUF] = out;
            if (mailbox && rr == 0.0) pcg_post(mailbox, 0, 1);
        }
    } else {
        double g[PCG_NW], Eg[PCG_NW];
        double pq;
        if (FIRST) {
            __syncthreads();                              // E, c_0, W~^T t complete in red[144 .. 224)
            if (w == 0) {
                const double e = coarse_invert_wave(red + 144, red + 224, red + 288);
                red[80 + lane] = e;
                if (blockIdx.x == 0) { coarse[lane] = e; if (lane < PCG_NW) coarse[PCG_NW * PCG_NW + lane] = red[144 + PCG_NW * PCG_NW + lane]; }
            }
            double rr = 0.0;
#pragma unroll
            for (int m = 0; m < PCG_EPT; ++m) rr += rv[m] * rv[m];
            rr = wave_allsum(rr);
            if (lane == 0) red[16 + w] = rr;
            __syncthreads();                              // E^-1 in red[80 .. 144), |b~|^2
            rr = red[16] + red[17] + red[18] + red[19];
            if (rr == 0.0) {                              // b~ = 0: x = 0 is the solution
                if (own) x_out[eo] = 0.0;
                if (blockIdx.x == 0 && tid == 0) {
                    scal[PS_RR0] = pcg_threshold_base(rr, scal, anchor, cap); flags[PF_DONE] = seq + 1; flags[PF_ITERS] = 0; flags[PF_XBUF] = out;
                    if (mailbox) pcg_post(mailbox, 0, 1);
                }
                return;
            }
            // c_0, mu_0 = E^-1 c_0 = p_mu0; q_0 = t + AW p_mu0; W~^T q_0 = W~^T t + E p_mu0; r_0 . z_0 = |b~|^2 + c_0 . mu_0
#pragma unroll
            for (int k = 0; k < PCG_NW; ++k) { c_in[k] = red[144 + PCG_NW * PCG_NW + k]; mu_in[k] = 0.0; }
            einv_apply(red + 80, c_in, mu_in);
#pragma unroll
            for (int k = 0; k < PCG_NW; ++k) pmu_in[k] = mu_in[k];
            rz_in = rr + dot8(c_in, mu_in);
            rr0 = anchor == 2 ? fmin(fmax(rr, rrf), cap * rr) : rr;          // pcg_threshold_base, without its store
            if (blockIdx.x == 0 && tid == 0) scal[PS_RR0] = pcg_threshold_base(rr, scal, anchor, cap);
#pragma unroll
            for (int k = 0; k < PCG_NW; ++k) {
                double erow[PCG_NW];
#pragma unroll
                for (int j = 0; j < PCG_NW; ++j) erow[j] = red[144 + PCG_NW * k + j];
                g[k] = red[144 + PCG_NW * PCG_NW + PCG_NW + k] + dot8(erow, pmu_in);
                Eg[k] = 0.0;
            }
            double pqr = 0.0;
#pragma unroll
            for (int m = 0; m < PCG_EPT; ++m) { qv[m] += dot8(aw5[m], pmu_in); pqr = fma(pv[m], qv[m], pqr); }     // (elements beyond d: all zero)
            qo += dot8(awo, pmu_in);
            pqr = wave_allsum(pqr);
            if (lane == 0) red[20 + w] = pqr;
            __syncthreads();
            pq = (red[20] + red[21]) + (red[22] + red[23]) + dot8(pmu_in, g);
        } else {
            reduce_partials<NV>(mine, red);
            __syncthreads();
#pragma unroll
            for (int k = 0; k < PCG_NW; ++k) { g[k] = COARSE ? red[1 + k] : 0.0; Eg[k] = 0.0; }
            pq = red[0] + (COARSE ? dot8(pmu_in, g) : 0.0);
        }
        const double alpha = rz_in * fast_rcp(pq);       // rcp + 2 Newton steps: the generic fp64 division is a ~15-deep dependent chain on the critical path
        if (COARSE) einv_apply(red + 80, g, Eg);          // independent of alpha: overlaps the reciprocal
#pragma unroll
        for (int k = 0; k < PCG_NW; ++k) { c_new[k] = fma(-alpha, g[k], c_in[k]); mu_new[k] = fma(-alpha, Eg[k], mu_in[k]); }
        double rrn = 0.0;
#pragma unroll
        for (int m = 0; m < PCG_EPT; ++m) { rv[m] -= alpha * qv[m]; rrn += rv[m] * rv[m]; }
        rrn = wave_allsum(rrn);
        if (lane == 0) red[16 + w] = rrn;
        const double cmu = COARSE ? dot8(c_new, mu_new) : 0.0;
        if (own) {                                       // x += alpha (p_r + W~ p_mu)
            x_out[eo] = xo + alpha * (po + (COARSE ? dot8(wo, pmu_in) : 0.0));
        }
        __syncthreads();
        rrn = red[16] + red[17] + red[18] + red[19];
        rz_new = rrn + cmu;
        const bool broke = !(pq > 0.0) || !(rrn == rrn);
        const bool done = rrn <= tol2 * rr0 || broke;
        if (done) {
            if (blockIdx.x == 0 && tid == 0) {
                flags[PF_DONE] = seq + 1; flags[PF_XBUF] = out; const int it = FIRST ? 1 : flags[PF_ITERS] + 1; flags[PF_ITERS] = it;
                if (broke) atomicCAS(info, 0, d + 1);
                if (mailbox) pcg_post(mailbox, it, 1);
            }
            return;
        }
        const double beta = rz_new * fast_rcp(rz_in);
#pragma unroll
        for (int k = 0; k < PCG_NW; ++k) pmu_new[k] = fma(beta, pmu_in[k], mu_new[k]);
#pragma unroll
        for (int m = 0; m < PCG_EPT; ++m) { const int e = tid + 256 * m; if (e < d) pl[e] = rv[m] + beta * pv[m]; }
        if (own) { const double rn = ro - alpha * qo; r_out[eo] = rn; p_out[eo] = rn + beta * po; }
        if (blockIdx.x == 0 && tid == 0) {
            const int it = FIRST ? 1 : flags[PF_ITERS] + 1; flags[PF_ITERS] = it; flags[PF_XBUF] = out; if (FIRST) flags[PF_DONE] = 0;
            if (mailbox) pcg_post(mailbox, it, 0);
        }
    }
    if (blockIdx.x == 0 && tid == 64) {
        st_out[PS_RZ] = rz_new;
#pragma unroll
        for (int k = 0; k < PCG_NW; ++k) { st_out[PS_C + k] = c_new[k]; st_out[PS_MU + k] = mu_new[k]; st_out[PS_PMU + k] = pmu_new[k]; }
    }
    if (COARSE && tid < PCG_NW) red[32 + tid] = pmu_new[tid];
    __syncthreads();
    // ---- q = S~ p_r + AW p_mu for the rows of this workgroup ----
    double pqp = 0.0, gacc = 0.0;
#pragma unroll
    for (int k = 0; k < PCG_RPW; ++k) {
        const int row = row0 + w + 4 * k;
        double sacc = (COARSE && lane < PCG_NW) ? awv[k] * red[32 + lane] : 0.0, sacc2 = 0.0;       // two chains: a dependent DFMA is ~32 cycles
#pragma unroll
        for (int m = 0; m < PCG_CPL / 2; ++m) {
            const int c2 = lane + 64 * m;
            double2 pv2 = (2 * c2 < d) ? reinterpret_cast<const double2*>(pl)[c2] : make_double2(0.0, 0.0);
            if (2 * c2 + 1 >= d) pv2.y = 0.0;              // pl[d] is not written
            sacc = fma(fv[k][m].x, pv2.x, sacc);
            sacc2 = fma(fv[k][m].y, pv2.y, sacc2);
        }
        sacc += sacc2;
        sacc = wave_allsum(sacc);
        if (lane == 0 && row < row1) { q_out[row] = sacc; pqp += pl[row] * sacc; }
        if (COARSE && lane >= PCG_NW && lane < 2 * PCG_NW && row < row1) gacc = fma(awv[k], sacc, gacc);
    }
    if (lane == 0) red[40 + 9 * w] = pqp;
    if (COARSE && lane >= PCG_NW && lane < 2 * PCG_NW) red[40 + 9 * w + 1 + (lane - PCG_NW)] = gacc;
    __syncthreads();
    if (tid < NV) pcg_part(part, out, tid)[blockIdx.x] = red[40 + tid] + red[49 + tid] + red[58 + tid] + red[67 + tid];
}

// AW = S~ W~, E = W~^T AW, c_0 = W~^T b~ for d <= 1280, same workgroup geometry as k_pcg_iter_fast: the rows of S~ and this
// thread's share of W~ are loaded up front, W~ goes to LDS in fp32 (its values are fp32-representable: lossless).
__global__ __launch_bounds__(256) void k_pcg_coarse_fast(int d, int ld, const double* __restrict__ F, const double* __restrict__ W,
                                                         const double* __restrict__ bt, double* __restrict__ AW, double* __restrict__ epart, int rows_per_wg,
                                                         double* __restrict__ t_out) {
    __shared__ __align__(16) float wt[PCG_NW][64 * PCG_CPL];
    __shared__ double esum[4][PCG_NW * PCG_NW + 2 * PCG_NW];
    __shared__ __align__(16) double bl[64 * PCG_CPL];            // b~ (t = S~ b~ for the first CG launch)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int row0 = blockIdx.x * rows_per_wg, row1 = min(d, row0 + rows_per_wg);
    double wreg[PCG_NW][PCG_EPT], breg[PCG_EPT];
#pragma unroll
    for (int m = 0; m < PCG_EPT; ++m) { const int e = tid + 256 * m; breg[m] = bt[e < d ? e : 0]; }
#pragma unroll
    for (int k = 0; k < PCG_NW; ++k)
#pragma unroll
        for (int m = 0; m < PCG_EPT; ++m) { const int e = tid + 256 * m; wreg[k][m] = W[(size_t)k * ld + (e < d ? e : 0)]; }      // clamped, branch-free:
                                                                    // a conditional load whose consumer is sunk into the branch is waited for on its own
    double2 fv[PCG_RPW][PCG_CPL / 2];
    double wrow[PCG_RPW], wcol[PCG_RPW], btr[PCG_RPW];
#pragma unroll
    for (int k = 0; k < PCG_RPW; ++k) {
        const int row = row0 + w + 4 * k;
        const bool have = row < row1;
        const double2* Fr = reinterpret_cast<const double2*>(F + (size_t)(have ? row : row0) * ld);
#pragma unroll
        for (int m = 0; m < PCG_CPL / 2; ++m) {
            const int c2 = lane + 64 * m;
            double2 v = make_double2(0.0, 0.0);
            if (have && 2 * c2 < d) v = Fr[c2];
            if (2 * c2 + 1 >= d) v.y = 0.0;
            fv[k][m] = v;
        }
        wrow[k] = have ? W[(size_t)(lane >> 3) * ld + row] : 0.0;           // E[k1][k2] += W~[k1][row] AW[row][k2], lane = 8 k1 + k2
        wcol[k] = have ? W[(size_t)(lane & 7) * ld + row] : 0.0;            // c_0[k] += W~[k][row] b~[row], lanes 0..7
        btr[k] = have ? bt[row] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < PCG_NW; ++k)
#pragma unroll
        for (int m = 0; m < PCG_EPT; ++m) wt[k][tid + 256 * m] = (tid + 256 * m < d) ? (float)wreg[k][m] : 0.0f;
#pragma unroll
    for (int m = 0; m < PCG_EPT; ++m) bl[tid + 256 * m] = (tid + 256 * m < d) ? breg[m] : 0.0;
    __syncthreads();
    double acc[PCG_RPW][PCG_NW], tacc[PCG_RPW];
#pragma unroll
    for (int r = 0; r < PCG_RPW; ++r) {
        tacc[r] = 0.0;
#pragma unroll
        for (int k = 0; k < PCG_NW; ++k) acc[r][k] = 0.0;
    }
#pragma unroll
    for (int m = 0; m < PCG_CPL / 2; ++m) {
        const int c2 = lane + 64 * m;
        const double2 bv = reinterpret_cast<const double2*>(bl)[c2];
#pragma unroll
        for (int r = 0; r < PCG_RPW; ++r) tacc[r] = fma(fv[r][m].x, bv.x, fma(fv[r][m].y, bv.y, tacc[r]));
#pragma unroll
        for (int k = 0; k < PCG_NW; ++k) {
            const float2 wv = reinterpret_cast<const float2*>(&wt[k][0])[c2];
#pragma unroll
            for (int r = 0; r < PCG_RPW; ++r) acc[r][k] = fma(fv[r][m].x, (double)wv.x, fma(fv[r][m].y, (double)wv.y, acc[r][k]));
        }
    }
#pragma unroll
    for (int r = 0; r < PCG_RPW; ++r)
#pragma unroll
        for (int k = 0; k < PCG_NW; ++k) acc[r][k] = wave_allsum(acc[r][k]);
    double e_acc = 0.0, c_acc = 0.0, t_acc = 0.0;
#pragma unroll
    for (int r = 0; r < PCG_RPW; ++r) {
        const int row = row0 + w + 4 * r;
        const double tr = wave_allsum(tacc[r]);          // (S~ b~)[row]: the first CG launch's q = S~ p_r with p_r = b~
        if (lane == 0 && row < row1) t_out[row] = tr;
        t_acc = fma(wcol[r], tr, t_acc);                 // W~^T S~ b~ (lanes 0..7; rows beyond row1: wcol = 0)
        double mine = 0.0;                               // AW[row][lane & 7]
#pragma unroll
        for (int k = 0; k < PCG_NW; ++k) mine = ((lane & 7) == k) ? acc[r][k] : mine;
        if (row < row1 && lane < PCG_NW) AW[(size_t)row * PCG_NW + lane] = mine;
        e_acc = fma(wrow[r], mine, e_acc);               // rows beyond row1 contribute wrow = 0
        c_acc = fma(wcol[r], btr[r], c_acc);
    }
    esum[w][lane] = e_acc;
    if (lane < PCG_NW) { esum[w][PCG_NW * PCG_NW + lane] = c_acc; esum[w][PCG_NW * PCG_NW + PCG_NW + lane] = t_acc; }
    __syncthreads();
    if (tid < PCG_NW * PCG_NW + 2 * PCG_NW) epart[(size_t)tid * PCG_PART + blockIdx.x] = esum[0][tid] + esum[1][tid] + esum[2][tid] + esum[3][tid];
}

// ---------------------------------------------------------------------------------------------------------------------
// Segmented coarse space (d <= 1280): the gauge vectors restricted to SEGMENTS of the camera order.
//
// A reduced matrix that is sparsely filled is a camera graph of large diameter -- views along a path, tracks shared by neighbouring
// cameras (what SfM.cpp:366-469 builds).  Block-Jacobi CG then needs hundreds of iterations, and the eight GLOBAL gauge vectors remove
// only the eight smallest eigenvalues: the next few dozen are similarity transforms of PIECES of the path against each other
// (tools/coarse_space_study.py on cfg3_banded: 187 iterations with block-Jacobi alone, 117 with the 8 global vectors, 33 with the seven
// similarity vectors multiplied by eight hat functions along the cyclic camera order + the global focal/depth vector).  That coarse space,
// 7 G + 1 = 57 vectors, is what this path uses.  Nothing new is stored per vector: with workgroup j = camera j (six rows; the last
// workgroup = the focal row) and the hats a partition of unity with two non-zero hats per camera,
//      W~_(g,k) = hat_g(camera) * W~_k        (W~_k: the 8 vectors k_finalize writes)
//      W~_(g,k)^T q = sum_j hat_g(j) t_k(j),  t_k(j) = sum over camera j's rows of W~_k[row] q[row]
// so a workgroup still publishes NINE partial sums per iteration (p_r . q and t_0 .. t_7) exactly like the 8-vector path; what changes is
// how the next launch adds them up (per hat instead of over all workgroups), that the coarse state c, mu, p_mu (57 each) lives one entry
// per lane in every wave, and that E^-1 is 57 x 57: each wave forms a quarter of E^-1 g, one LDS exchange completes it.
// Set-up per linear solve: k_ml_aw (AW = S~ W~ for the 57 vectors, per-camera pieces of E and c_0), k_ml_e (E, c_0 summed per hat),
// k_ml_invert (Jacobi-scaled Gauss-Jordan in one workgroup, 57 pivot steps with one barrier each; a vanishing pivot drops its vector).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int ML_G = 8;                    // hat functions along the (cyclic) camera order
constexpr int ML_NC = 7 * ML_G + 1;        // coarse vectors: (g, k) -> 7 g + k, the global focal/depth vector last
constexpr int ML_N = 64;                   // padded: one coarse entry per lane
constexpr int ML_MIN_CAMS = 4 * ML_G;      // below this the hats have too few cameras each: the 8-vector path
constexpr int ML_LDS_TAIL = 96 + 7 * 256 + 2 * 4 * ML_N;     // doubles of LDS behind the search direction: red | tmp | gq | egq

// cameras whose LOWER hat is a: [ml_first_cam(a), ml_first_cam(a + 1)); camera j there has weight 1 - frac in hat a and frac in hat
// (a + 1) mod G, frac = (j G - a nc) / nc -- ONE formula for every place that needs a hat weight
__device__ __forceinline__ int ml_first_cam(int a, int nc) { return (a * nc + ML_G - 1) / ML_G; }
__device__ __forceinline__ double ml_frac(int j, int a, int nc, double inv_nc) { return (double)(j * ML_G - a * nc) * inv_nc; }

// sum over the cameras of hat g of weight * f(camera): the two ranges (which_range 0: cameras whose lower hat is g - 1, weight frac; 1: lower hat
// g, weight 1 - frac); `part` of `nparts` equal slices of the range.  Fixed order: deterministic.  MAXT bounds the terms of one slice: all of them
// are fetched before the first is used (clamped, branch-free -- a loop of load / use pairs pays one memory round trip per term).
template <int MAXT, typename Fn>
__device__ __forceinline__ double ml_hat_sum(int g, int nc, double inv_nc, int which_range, int part, int nparts, Fn f) {
    const int a = which_range == 0 ? (g + ML_G - 1) % ML_G : g;
    const int lo = ml_first_cam(a, nc), hi = ml_first_cam(a + 1, nc);
    const int len = hi - lo, chunk = (len + nparts - 1) / nparts;
    const int j0 = lo + part * chunk, j1 = min(hi, j0 + chunk);
    double s = 0.0;
    for (int jb = j0; jb < j1; jb += MAXT) {
        double val[MAXT];
#pragma unroll
        for (int t = 0; t < MAXT; ++t) val[t] = f(jb + t < j1 ? jb + t : j1 - 1);
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            const double fr = ml_frac(jb + t, a, nc, inv_nc);
            s = fma(jb + t < j1 ? (which_range == 0 ? fr : 1.0 - fr) : 0.0, val[t], s);
        }
    }
    return s;
}

// AW[row][64] = S~ W~ for the 57 vectors (own rows), V[wg][8][64] = sum over own rows of W~_k[row] AW[row][:], u[wg][8] = sum over own rows
// of W~_k[row] b~[row].  Workgroup = camera (last: the focal row).  The camera's six rows of S~ and W~_0..7 (fp32: lossless) go to LDS in one
// round trip; lane (g, k) of a wave then walks the cameras of hat g for the wave's two rows.
constexpr int ML_ROWLEN = 64 * PCG_CPL;
constexpr size_t ML_AW_LDS = sizeof(double) * 6 * ML_ROWLEN + sizeof(float) * PCG_NW * ML_ROWLEN + sizeof(double) * 4 * PCG_NW * ML_N;
__global__ __launch_bounds__(256) void k_ml_aw(int d, int ld, const double* __restrict__ F, const double* __restrict__ W, const double* __restrict__ bt,
                                               double* __restrict__ AW, double* __restrict__ V, double* __restrict__ U) {
    extern __shared__ __align__(16) double sm[];
    double* rows = sm;                                                   // [6][ML_ROWLEN]
    float* wt = reinterpret_cast<float*>(rows + 6 * ML_ROWLEN);          // [8][ML_ROWLEN]
    double* vbuf = reinterpret_cast<double*>(wt + PCG_NW * ML_ROWLEN);   // [4][8][64]
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nc = (d - 1) / 6;
    const double inv_nc = 1.0 / (double)nc;
    const int row0 = 6 * blockIdx.x, row1 = min(d, row0 + 6);
    {
        double2 rv[6][3];
        double wv[PCG_NW][PCG_EPT];
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                const int c2 = tid + 256 * m;
                const bool ok = row0 + r < row1 && 2 * c2 < d;
                rv[r][m] = reinterpret_cast<const double2*>(F + (size_t)(row0 + r < row1 ? row0 + r : row0) * ld)[ok ? c2 : 0];
                if (!ok) rv[r][m] = make_double2(0.0, 0.0);
                if (2 * c2 + 1 >= d) rv[r][m].y = 0.0;
            }
#pragma unroll
        for (int k = 0; k < PCG_NW; ++k)
#pragma unroll
            for (int m = 0; m < PCG_EPT; ++m) { const int e = tid + 256 * m; wv[k][m] = W[(size_t)k * ld + (e < d ? e : 0)]; }
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int m = 0; m < 3; ++m) { const int c2 = tid + 256 * m; if (2 * c2 < ML_ROWLEN) reinterpret_cast<double2*>(rows + r * ML_ROWLEN)[c2] = rv[r][m]; }
#pragma unroll
        for (int k = 0; k < PCG_NW; ++k)
#pragma unroll
            for (int m = 0; m < PCG_EPT; ++m) { const int e = tid + 256 * m; wt[k * ML_ROWLEN + e] = e < d ? (float)wv[k][m] : 0.0f; }
    }
    __syncthreads();
    const int g = lane / 7, k = lane - 7 * g;        // lanes 0..55: vector (g, k); lane 56: the global vector; beyond: nothing
    const int ra = w, rb = w + 4;                    // this wave's rows (of the camera's six)
    const bool have_a = row0 + ra < row1, have_b = row0 + rb < row1;
    double awa = 0.0, awb = 0.0;
    if (lane < 7 * ML_G) {
        const float* wk = wt + k * ML_ROWLEN;
        const double* Ra = rows + ra * ML_ROWLEN;
        const double* Rb = rows + (have_b ? rb : ra) * ML_ROWLEN;
        for (int range = 0; range < 2; ++range) {
            const int a = range == 0 ? (g + ML_G - 1) % ML_G : g;
            const int lo = ml_first_cam(a, nc), hi = ml_first_cam(a + 1, nc);
#pragma unroll 3
            for (int j = lo; j < hi; ++j) {               // (three cameras' LDS reads in flight)
                const double fr = ml_frac(j, a, nc, inv_nc);
                const double wgt = range == 0 ? fr : 1.0 - fr;
                const float2 w01 = reinterpret_cast<const float2*>(wk + 6 * j)[0], w23 = reinterpret_cast<const float2*>(wk + 6 * j)[1], w45 = reinterpret_cast<const float2*>(wk + 6 * j)[2];
                const double2 a0 = reinterpret_cast<const double2*>(Ra + 6 * j)[0], a1 = reinterpret_cast<const double2*>(Ra + 6 * j)[1], a2 = reinterpret_cast<const double2*>(Ra + 6 * j)[2];
                const double2 b0 = reinterpret_cast<const double2*>(Rb + 6 * j)[0], b1 = reinterpret_cast<const double2*>(Rb + 6 * j)[1], b2 = reinterpret_cast<const double2*>(Rb + 6 * j)[2];
                const double ta = fma(a0.x, (double)w01.x, fma(a0.y, (double)w01.y, fma(a1.x, (double)w23.x, fma(a1.y, (double)w23.y, fma(a2.x, (double)w45.x, a2.y * (double)w45.y)))));
                const double tb = fma(b0.x, (double)w01.x, fma(b0.y, (double)w01.y, fma(b1.x, (double)w23.x, fma(b1.y, (double)w23.y, fma(b2.x, (double)w45.x, b2.y * (double)w45.y)))));
                awa = fma(wgt, ta, awa);
                awb = fma(wgt, tb, awb);
            }
        }
    }
    {   // the global vector: all d columns, the lanes stride them
        double sa = 0.0, sb = 0.0;
        const float* w7 = wt + (PCG_NW - 1) * ML_ROWLEN;
        for (int c = lane; c < d; c += 64) { sa = fma(rows[ra * ML_ROWLEN + c], (double)w7[c], sa); sb = fma(rows[(have_b ? rb : ra) * ML_ROWLEN + c], (double)w7[c], sb); }
        sa = wave_allsum(sa); sb = wave_allsum(sb);
        if (lane == ML_NC - 1) { awa = sa; awb = sb; }
    }
    if (!have_a) awa = 0.0;
    if (!have_b) awb = 0.0;
    if (have_a) AW[(size_t)(row0 + ra) * ML_N + lane] = awa;
    if (have_b) AW[(size_t)(row0 + rb) * ML_N + lane] = awb;
#pragma unroll
    for (int q = 0; q < PCG_NW; ++q) {
        const double wa = have_a ? (double)wt[q * ML_ROWLEN + row0 + ra] : 0.0, wb = have_b ? (double)wt[q * ML_ROWLEN + row0 + rb] : 0.0;
        vbuf[(w * PCG_NW + q) * ML_N + lane] = fma(wa, awa, wb * awb);
    }
    __syncthreads();
    for (int e = tid; e < PCG_NW * ML_N; e += 256)
        V[(size_t)blockIdx.x * PCG_NW * ML_N + e] = (vbuf[e] + vbuf[PCG_NW * ML_N + e]) + (vbuf[2 * PCG_NW * ML_N + e] + vbuf[3 * PCG_NW * ML_N + e]);
    if (tid < PCG_NW) {
        double s = 0.0;
        for (int row = row0; row < row1; ++row) s = fma((double)wt[tid * ML_ROWLEN + row], bt[row], s);
        U[(size_t)blockIdx.x * PCG_NW + tid] = s;
    }
}

// E[i][:] and c_0[i]: row i = (g, k) is the hat-weighted sum of the cameras' pieces k; the last row the plain sum of piece 7 over all workgroups.
// One workgroup per row, lane = column, the terms split over the four waves.
__global__ __launch_bounds__(256) void k_ml_e(int d, const double* __restrict__ V, const double* __restrict__ U, double* __restrict__ E, double* __restrict__ c0) {
    __shared__ double eq[4][ML_N], cq[4];
    const int i = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nc = (d - 1) / 6;
    const double inv_nc = 1.0 / (double)nc;
    double e = 0.0, c = 0.0;
    if (i < 7 * ML_G) {
        const int g = i / 7, k = i - 7 * g;
        e = ml_hat_sum<16>(g, nc, inv_nc, w >> 1, w & 1, 2, [&](int j) { return V[((size_t)j * PCG_NW + k) * ML_N + lane]; });
        c = ml_hat_sum<16>(g, nc, inv_nc, w >> 1, w & 1, 2, [&](int j) { return U[(size_t)j * PCG_NW + k]; });
    } else {
        for (int jb = w; jb <= nc; jb += 64) {          // wave w: workgroups w, w + 4, ...; sixteen loads in flight
            double ve[16], vc[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int j = jb + 4 * t, jc = j <= nc ? j : nc;
                ve[t] = V[((size_t)jc * PCG_NW + (PCG_NW - 1)) * ML_N + lane]; vc[t] = U[(size_t)jc * PCG_NW + (PCG_NW - 1)];
            }
#pragma unroll
            for (int t = 0; t < 16; ++t) { const bool ok = jb + 4 * t <= nc; e += ok ? ve[t] : 0.0; c += ok ? vc[t] : 0.0; }
        }
    }
    eq[w][lane] = e;
    if (lane == 0) cq[w] = c;
    __syncthreads();
    if (tid < ML_N) E[(size_t)i * ML_N + tid] = (eq[0][tid] + eq[1][tid]) + (eq[2][tid] + eq[3][tid]);
    if (tid == 0) c0[i] = (cq[0] + cq[1]) + (cq[2] + cq[3]);
}

// In-place Gauss-Jordan inverse of the Jacobi-scaled N x N matrix E (symmetric positive definite, no pivot search) in the registers of ONE workgroup:
// thread (tr, tc) holds the TR x TC tile of rows TR tr .., columns TC tc .. -- per pivot it needs TR entries of the pivot column and TC of the pivot row
// from LDS (a column-per-thread layout reads a whole row slice per thread: the kernel was bound by that LDS traffic).  One barrier per pivot (row, column
// and the next diagonal entry double-buffered by pivot parity); the next pivot's reciprocal is formed during the current update; L = lcm(TR, TC)
// steps are instantiated with compile-time register indices and that body loops.  A pivot below 1e-10 of the unit diagonal: the vector depends on the
// earlier ones, its step is skipped and its row and column of the result are zero.  Rows / columns >= NC: zero.
template <typename F, int... S>
__device__ __forceinline__ void gj_steps(F& step, int m, std::integer_sequence<int, S...>) { (step(m, std::integral_constant<int, S>()), ...); }
template <int N, int TR, int TC, int L>
__device__ __forceinline__ void gj_invert_tiled(int NC, const double* __restrict__ E, double* __restrict__ einv,
                                                double* rowbuf, double* colbuf, double* sc, double* diagbuf, unsigned char* drop) {
    static_assert(N % TR == 0 && N % TC == 0 && L % TR == 0 && L % TC == 0, "tile geometry");
    constexpr int NTC = N / TC;
    const int tid = threadIdx.x, tr = tid / NTC, tc = tid % NTC;
    for (int t = tid; t <= N; t += blockDim.x) {
        const double dii = t < NC ? E[(size_t)t * N + t] : 0.0;
        const bool ok = dii > 0.0 && dii <= 1.7e308;
        if (t < N) sc[t] = ok ? 1.0 / sqrt(dii) : 0.0;
        drop[t] = ok ? 0 : 1;
    }
    __syncthreads();
    double a[TR][TC];
#pragma unroll
    for (int rr = 0; rr < TR; ++rr)
#pragma unroll
        for (int cc = 0; cc < TC; ++cc) {
            const int i = TR * tr + rr, j = TC * tc + cc;
            const bool in = i < NC && j < NC;
            const double v = in ? 0.5 * (E[(size_t)i * N + j] + E[(size_t)j * N + i]) * sc[i] * sc[j] : 0.0;
            a[rr][cc] = (i == j) ? 1.0 : v;
        }
    double piv = 1.0, ip = 1.0;                      // unit diagonal after the scaling
    auto step = [&](int m, auto sconst) __attribute__((always_inline)) {
        constexpr int sidx = decltype(sconst)::value, rr0 = sidx % TR, cc0 = sidx % TC, rr1 = (sidx + 1) % TR, cc1 = (sidx + 1) % TC;
        const int p = L * m + sidx;
        if (p >= NC) return;                         // uniform
        const int rg = p / TR, cg = p / TC, par = p & 1;
        if (tr == rg) {
#pragma unroll
            for (int cc = 0; cc < TC; ++cc) rowbuf[par * N + TC * tc + cc] = a[rr0][cc];
        }
        if (tc == cg) {
#pragma unroll
            for (int rr = 0; rr < TR; ++rr) colbuf[par * N + TR * tr + rr] = a[rr][cc0];
        }
        if (tr == (p + 1) / TR && tc == (p + 1) / TC) diagbuf[par] = a[rr1][cc1];
        __syncthreads();
        const bool ok = !drop[p] && piv > 1e-10;
        double dn = diagbuf[par];
        if (ok) {
            // a_ij -= (a_ip / piv) a_pj everywhere -- the pivot row itself with the multiplier 1 - 1/piv --, then the pivot column is set
            double fc[TR], f[TR], rv[TC];
#pragma unroll
            for (int rr = 0; rr < TR; ++rr) { fc[rr] = colbuf[par * N + TR * tr + rr] * ip; f[rr] = (TR * tr + rr == p) ? 1.0 - ip : fc[rr]; }
#pragma unroll
            for (int cc = 0; cc < TC; ++cc) rv[cc] = rowbuf[par * N + TC * tc + cc];
            const int q = p + 1 < N ? p + 1 : N - 1;
            dn = fma(-colbuf[par * N + q] * ip, rowbuf[par * N + q], dn);
#pragma unroll
            for (int rr = 0; rr < TR; ++rr)
#pragma unroll
                for (int cc = 0; cc < TC; ++cc) a[rr][cc] = fma(-f[rr], rv[cc], a[rr][cc]);
            if (tc == cg) {
#pragma unroll
                for (int rr = 0; rr < TR; ++rr) a[rr][cc0] = (TR * tr + rr == p) ? ip : -fc[rr];
            }
        } else if (tid == 0) drop[p] = 1;
        piv = dn;
        ip = fast_rcp(dn);
    };
    for (int m = 0; m * L < NC; ++m) gj_steps(step, m, std::make_integer_sequence<int, L>());
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < TR; ++rr)
#pragma unroll
        for (int cc = 0; cc < TC; ++cc) {
            const int i = TR * tr + rr, j = TC * tc + cc;
            einv[(size_t)i * N + j] = (drop[i] || drop[j]) ? 0.0 : a[rr][cc] * sc[i] * sc[j];
        }
}

// E^-1 (64 x 64, rows / columns beyond the 57 vectors and of dropped vectors zero): 4 x 4 tiles on 256 threads.  (History: the compiler REFUSED a 57-step
// `#pragma unroll` -- "loop not unrolled", dynamic register indices, 74 us; a column slice per thread with compile-time indices took 32 us.)
__global__ __launch_bounds__(256) void k_ml_invert(const double* __restrict__ E, double* __restrict__ einv, double* __restrict__ c0) {
    __shared__ double rowbuf[2 * ML_N], colbuf[2 * ML_N], sc[ML_N], diagbuf[2];
    __shared__ unsigned char drop[ML_N + 1];
    if (threadIdx.x >= ML_NC && threadIdx.x < ML_N) c0[threadIdx.x] = 0.0;
    gj_invert_tiled<ML_N, 4, 4, 4>(ML_NC, E, einv, rowbuf, colbuf, sc, diagbuf, drop);
}

// One CG iteration with the segmented coarse space: k_pcg_iter_fast's structure (every global load issued up front, one memory round trip per
// launch), workgroup = camera.  LDS behind the search direction: red[96] | tmp[7][256] (the partials t_0..t_6 of every workgroup) |
// gq[4][64] (quarter sums of W~^T q per wave) | egq[4][64] (quarter products of E^-1 g per wave).
template <bool INIT>
__global__ __launch_bounds__(256) void k_pcg_iter_ml(int d, int ld, const double* __restrict__ F, double* __restrict__ vec,
                                                     const double* __restrict__ bt, double* __restrict__ part, double* __restrict__ scal,
                                                     int* flags, double tol2, int in, int* info, int* mailbox, int anchor, double cap,
                                                     const double* __restrict__ W, const double* __restrict__ AW, const double* __restrict__ einv,
                                                     const double* __restrict__ c0, double* __restrict__ mlstate) {
    extern __shared__ __align__(16) double sm[];
    double* pl = sm;
    double* red = sm + ld;
    double* tmp = red + 96;
    double* gq = tmp + 7 * 256;
    double* egq = gq + 4 * ML_N;
    const int seq = in >> 1;
    in &= 1;
    if (!INIT) { const int dn = flags[PF_DONE]; if (dn != 0 && seq >= dn) return; }
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), out = in ^ 1;
    const int nc = (d - 1) / 6, nwg = (int)gridDim.x;
    const double inv_nc = 1.0 / (double)nc;
    const int row0 = 6 * blockIdx.x, row1 = min(d, row0 + 6);
    const double* x_in = pcg_vec(vec, 0, in, ld); const double* r_in = INIT ? bt : pcg_vec(vec, 1, in, ld);
    const double* p_in = pcg_vec(vec, 2, in, ld); const double* q_in = pcg_vec(vec, 3, in, ld);
    double* x_out = pcg_vec(vec, 0, out, ld); double* r_out = pcg_vec(vec, 1, out, ld);
    double* p_out = pcg_vec(vec, 2, out, ld); double* q_out = pcg_vec(vec, 3, out, ld);
    const double* st_in = scal + PS_STATE + PS_STATE_LEN * in;
    double* st_out = scal + PS_STATE + PS_STATE_LEN * out;
    const double* ms_in = mlstate + 3 * ML_N * in;
    double* ms_out = mlstate + 3 * ML_N * out;

    // ---- all global loads of this iteration ----
    double rv[PCG_EPT], qv[PCG_EPT], pv[PCG_EPT];
#pragma unroll
    for (int m = 0; m < PCG_EPT; ++m) {
        const int e = tid + 256 * m;
        const bool ok = e < d;
        rv[m] = ok ? r_in[e] : 0.0; qv[m] = (ok && !INIT) ? q_in[e] : 0.0; pv[m] = (ok && !INIT) ? p_in[e] : 0.0;
    }
    double pp[PCG_NPART];                          // the nine partial sums workgroup `tid` published (clamped, branch-free)
#pragma unroll
    for (int v = 0; v < PCG_NPART; ++v) {
        const double t = INIT ? 0.0 : pcg_part(part, in, v)[tid < nwg ? tid : nwg - 1];
        pp[v] = tid < nwg ? t : 0.0;
    }
    double em[16];                                 // E^-1[16 w + jj][lane] (symmetric: = row `lane`, this wave's quarter of the columns)
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) em[jj] = einv[(size_t)(16 * w + jj) * ML_N + lane];
    const double c_in = INIT ? c0[lane] : ms_in[lane];
    const double mu_in = INIT ? 0.0 : ms_in[ML_N + lane];
    const double pmu_in = INIT ? 0.0 : ms_in[2 * ML_N + lane];
    const double rr0 = INIT ? 0.0 : scal[PS_RR0];
    const double rz_in = INIT ? 0.0 : st_in[PS_RZ];
    const int eo = row0 + (tid - 192);
    const bool own = tid >= 192 && eo < row1;
    double xo = 0.0, po = 0.0, ro = 0.0, qo = 0.0, wo[PCG_NW];
    if (own) { ro = r_in[eo]; if (!INIT) { xo = x_in[eo]; po = p_in[eo]; qo = q_in[eo]; } }
#pragma unroll
    for (int k = 0; k < PCG_NW; ++k) wo[k] = (own && !INIT) ? W[(size_t)k * ld + eo] : 0.0;
    double2 fv[PCG_RPW][PCG_CPL / 2];
    double awv[PCG_RPW], wg8[PCG_RPW];             // AW[row][lane]; W~_(lane & 7)[row]
#pragma unroll
    for (int k = 0; k < PCG_RPW; ++k) {
        const int row = row0 + w + 4 * k;
        const bool have = row < row1;
        const double2* Fr = reinterpret_cast<const double2*>(F + (size_t)(have ? row : row0) * ld);
#pragma unroll
        for (int m = 0; m < PCG_CPL / 2; ++m) {
            const int c2 = lane + 64 * m;
            double2 v = make_double2(0.0, 0.0);
            if (have && 2 * c2 < d) v = Fr[c2];
            if (2 * c2 + 1 >= d) v.y = 0.0;
            fv[k][m] = v;
        }
        awv[k] = have ? AW[(size_t)row * ML_N + lane] : 0.0;
        wg8[k] = have ? W[(size_t)(lane & 7) * ld + row] : 0.0;
    }
    double c_new, mu_new, pmu_new, rz_new;
    if (INIT) {
        // x0 = 0, r0 = b~, c0 = W~^T b~ (k_ml_e), mu0 = E^-1 c0, z0 = r0 + W~ mu0, p0 = z0
        double rr = 0.0;
#pragma unroll
        for (int m = 0; m < PCG_EPT; ++m) rr += rv[m] * rv[m];
        rr = wave_allsum(rr);
        if (lane == 0) red[16 + w] = rr;
        double e = 0.0;
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) e = fma(em[jj], lane_bcast(c_in, 16 * w + jj), e);
        egq[w * ML_N + lane] = e;
        __syncthreads();
        rr = red[16] + red[17] + red[18] + red[19];
        c_new = c_in;
        mu_new = (egq[lane] + egq[ML_N + lane]) + (egq[2 * ML_N + lane] + egq[3 * ML_N + lane]);
        rz_new = rr + wave_allsum(c_new * mu_new);
        pmu_new = mu_new;
#pragma unroll
        for (int m = 0; m < PCG_EPT; ++m) { const int e2 = tid + 256 * m; if (e2 < d) pl[e2] = rv[m]; }
        if (own) { x_out[eo] = 0.0; r_out[eo] = ro; p_out[eo] = ro; }
        if (blockIdx.x == 0 && tid == 0) {
            scal[PS_RR0] = pcg_threshold_base(rr, scal, anchor, cap); flags[PF_DONE] = (rr == 0.0); flags[PF_ITERS] = 0; flags[PF_XBUF] = out;
            if (mailbox && rr == 0.0) pcg_post(mailbox, 0, 1);
        }
    } else {
        // ---- p_r . q and t_7 over all workgroups (registers), t_0..t_6 per hat (through LDS) ----
#pragma unroll
        for (int v = 1; v < PCG_NW; ++v) tmp[(v - 1) * 256 + tid] = pp[v];
        {
            const double a = wave_allsum(pp[0]), b = wave_allsum(pp[PCG_NW]);
            if (lane == 0) { red[w] = a; red[4 + w] = b; }
        }
        __syncthreads();
        {
            // wave 0, 1: the two halves of the hat's lower range (cameras whose lower hat is g - 1); wave 2, 3: of its upper range
            double s = 0.0;
            if (lane < 7 * ML_G) {
                const int g = lane / 7, k = lane - 7 * g;
                const double* tk = tmp + k * 256;
                s = ml_hat_sum<8>(g, nc, inv_nc, w >> 1, w & 1, 2, [&](int j) { return tk[j]; });
            }
            gq[w * ML_N + lane] = s;
        }
        __syncthreads();
        double g = (gq[lane] + gq[ML_N + lane]) + (gq[2 * ML_N + lane] + gq[3 * ML_N + lane]);
        if (lane == ML_NC - 1) g = (red[4] + red[5]) + (red[6] + red[7]);
        const double pq = (red[0] + red[1]) + (red[2] + red[3]) + wave_allsum(pmu_in * g);
        const double alpha = rz_in * fast_rcp(pq);
        {
            double e = 0.0;                                // this wave's quarter of E^-1 g (independent of alpha)
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) e = fma(em[jj], lane_bcast(g, 16 * w + jj), e);
            egq[w * ML_N + lane] = e;
        }
        double rrn = 0.0;
#pragma unroll
        for (int m = 0; m < PCG_EPT; ++m) { rv[m] -= alpha * qv[m]; rrn += rv[m] * rv[m]; }
        rrn = wave_allsum(rrn);
        if (lane == 0) red[16 + w] = rrn;
        if (w == 3) {                                      // x += alpha (p_r + W~ p_mu): this camera's two hats
            const int jc = min((int)blockIdx.x, nc - 1);   // (the focal row: only W~_7 is non-zero there)
            const int gl = (jc * ML_G) / nc, gh = gl + 1 == ML_G ? 0 : gl + 1;
            const double fr = ml_frac(jc, gl, nc, inv_nc);
            double wp = wo[PCG_NW - 1] * lane_bcast(pmu_in, ML_NC - 1);
#pragma unroll
            for (int k = 0; k < 7; ++k)
                wp = fma(wo[k], fma(fr, lane_bcast(pmu_in, 7 * gh + k), (1.0 - fr) * lane_bcast(pmu_in, 7 * gl + k)), wp);
            if (own) x_out[eo] = xo + alpha * (po + wp);
        }
        __syncthreads();
        const double Eg = (egq[lane] + egq[ML_N + lane]) + (egq[2 * ML_N + lane] + egq[3 * ML_N + lane]);
        c_new = fma(-alpha, g, c_in);
        mu_new = fma(-alpha, Eg, mu_in);
        rrn = red[16] + red[17] + red[18] + red[19];
        rz_new = rrn + wave_allsum(c_new * mu_new);
        const bool broke = !(pq > 0.0) || !(rrn == rrn);
        const bool done = rrn <= tol2 * rr0 || broke;
        if (done) {
            if (blockIdx.x == 0 && tid == 0) {
                flags[PF_DONE] = seq + 1; flags[PF_XBUF] = out; const int it = flags[PF_ITERS] + 1; flags[PF_ITERS] = it;
                if (broke) atomicCAS(info, 0, d + 1);
                if (mailbox) pcg_post(mailbox, it, 1);
            }
            return;
        }
        const double beta = rz_new * fast_rcp(rz_in);
        pmu_new = fma(beta, pmu_in, mu_new);
#pragma unroll
        for (int m = 0; m < PCG_EPT; ++m) { const int e = tid + 256 * m; if (e < d) pl[e] = rv[m] + beta * pv[m]; }
        if (own) { const double rn = ro - alpha * qo; r_out[eo] = rn; p_out[eo] = rn + beta * po; }
        if (blockIdx.x == 0 && tid == 0) { const int it = flags[PF_ITERS] + 1; flags[PF_ITERS] = it; flags[PF_XBUF] = out; if (mailbox) pcg_post(mailbox, it, 0); }
    }
    if (blockIdx.x == 0 && w == 1) {
        ms_out[lane] = c_new; ms_out[ML_N + lane] = mu_new; ms_out[2 * ML_N + lane] = pmu_new;
        if (lane == 0) st_out[PS_RZ] = rz_new;
    }
    __syncthreads();
    // ---- q = S~ p_r + AW p_mu for the rows of this camera ----
    double pqp = 0.0, gacc = 0.0;
#pragma unroll
    for (int k = 0; k < PCG_RPW; ++k) {
        const int row = row0 + w + 4 * k;
        double sacc = awv[k] * pmu_new, sacc2 = 0.0;
#pragma unroll
        for (int m = 0; m < PCG_CPL / 2; ++m) {
            const int c2 = lane + 64 * m;
            double2 pv2 = (2 * c2 < d) ? reinterpret_cast<const double2*>(pl)[c2] : make_double2(0.0, 0.0);
            if (2 * c2 + 1 >= d) pv2.y = 0.0;
            sacc = fma(fv[k][m].x, pv2.x, sacc);
            sacc2 = fma(fv[k][m].y, pv2.y, sacc2);
        }
        sacc += sacc2;
        sacc = wave_allsum(sacc);
        if (lane == 0 && row < row1) { q_out[row] = sacc; pqp += pl[row] * sacc; }
        if (lane >= PCG_NW && lane < 2 * PCG_NW && row < row1) gacc = fma(wg8[k], sacc, gacc);
    }
    if (lane == 0) red[40 + 9 * w] = pqp;
    if (lane >= PCG_NW && lane < 2 * PCG_NW) red[40 + 9 * w + 1 + (lane - PCG_NW)] = gacc;
    __syncthreads();
    if (tid < PCG_NPART) pcg_part(part, out, tid)[blockIdx.x] = red[40 + tid] + red[49 + tid] + red[58 + tid] + red[67 + tid];
}

// ---------------------------------------------------------------------------------------------------------------------
// Segmented coarse space on the STREAMING path (d > 1280: long camera paths -- 600 cameras of a path need 314 .. 396 CG iterations per
// linearisation with the eight global vectors, tools/large_banded_check.py).  Same coarse space as above with G = cameras / 25 hats (at most 20:
// 7 G + 1 <= 141 vectors), but a coarse operator of that size cannot ride in every workgroup of a fused launch (E^-1 is 166 KB), and the search
// direction need not be kept split: classical PCG with M^-1 = I + W~ E^-1 W~^T in THREE launches per iteration --
//   k_sg_q   q = S~ p for the rows of a workgroup (the streaming product of k_pcg_iter), per-workgroup partials of p . q
//   k_sg_u   alpha; x += alpha p, r -= alpha q on the rows of a workgroup's cameras; per camera t_k = sum of W~_k[row] r[row]; partials of |r|^2
//   k_sg_p   |r|^2 (the stopping test), c = W~^T r from the t_k per hat, mu = E^-1 c, r . z = |r|^2 + c . mu, beta, p = r + W~ mu + beta p
// (the coarse solve is formed by each of the SG_UWG workgroups of k_sg_p for itself).  Set-up per linear solve: k_sg_v (per camera: the pieces of E
// and c_0, AW is never stored), k_sg_e (E, hat sums), k_sg_invert (Gauss-Jordan in the registers of one workgroup -- what limits the hats to 20,
// see there).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int SG_MAXG = 20;
constexpr int SG_NCP = 144;                 // padded coarse dimension: 7 SG_MAXG + 1 = 141 vectors
constexpr int SG_CB = 36;                   // columns per part of the E^-1 c product (SG_NCP / 4)
constexpr int SG_UWG = 16;                  // workgroups of the vector kernels
constexpr int SG_UT = 1024;                 // ... and their threads
constexpr int SG_TC = 64;                   // cameras per column tile of k_sg_v
enum { SGS_RZ = 0, SGS_PQ = 1, SGS_LEN = 4 };     // per-parity scalars of the running solve (sg_state)

__device__ __forceinline__ int sg_first_cam(int a, int nc, int G) { return (a * nc + G - 1) / G; }
__device__ __forceinline__ double sg_frac(int j, int a, int nc, int G, double inv_nc) { return (double)(j * G - a * nc) * inv_nc; }
__host__ __device__ __forceinline__ int sg_hats(int nc) { const int g = nc / 25; return g < ML_G ? ML_G : g > SG_MAXG ? SG_MAXG : g; }

// per camera j (workgroup; the last one: the focal row): V[j][8][SG_NCP] = sum over the camera's rows of W~_k[row] (S~ W~)[row][:].  Column tiles of
// SG_TC cameras, two phases per tile through LDS: (1) T[row][camera][k] = the 6-term product of the camera's row entries with W~_k, one item per thread
// and step -- k = 7 is the global vector's share --, (2) lane + 64 pass = coarse vector (g, k) adds hat_g(camera) T over the cameras of its hat inside the
// tile.  (With the 6-term products inside phase 2, every wave ran as long as its busiest lane's hat: 187 us.)
template <typename FT>
__global__ __launch_bounds__(256) void k_sg_v(int d, int ld, int G, const FT* __restrict__ F, const double* __restrict__ W,
                                              double* __restrict__ V) {
    constexpr int TW = 6 * SG_TC;
    __shared__ __align__(16) double rows[6 * TW];               // [6][TW]
    __shared__ __align__(16) float wt[PCG_NW * TW];             // [8][TW]
    __shared__ double T[6 * SG_TC * PCG_NW];                    // [6][SG_TC][8]
    __shared__ double vbuf[4 * SG_NCP];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nc = (d - 1) / 6, NC = 7 * G + 1;
    const double inv_nc = 1.0 / (double)nc;
    const int row0 = 6 * blockIdx.x, row1 = min(d, row0 + 6);
    const int ra = w, rb = w + 4;
    const bool have_a = row0 + ra < row1, have_b = row0 + rb < row1;
    double acc[2][3], gs[2] = { 0.0, 0.0 };
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int p = 0; p < 3; ++p) acc[r][p] = 0.0;
    for (int c0 = 0; c0 < nc; c0 += SG_TC) {
        const int col0 = 6 * c0, ncol = min(6 * nc - col0, TW);
        // all loads of the tile first (9 + 12 per thread, clamped, branch-free), then the LDS stores
        FT fr_[6 * TW / 256];
        double wr_[PCG_NW * TW / 256];
#pragma unroll
        for (int u = 0; u < 6 * TW / 256; ++u) {
            const int e = tid + 256 * u, r = e / TW, c = e - TW * r;
            const bool ok = row0 + r < row1 && c < ncol;
            fr_[u] = F[(size_t)(ok ? row0 + r : row0) * ld + col0 + (ok ? c : 0)];
        }
#pragma unroll
        for (int u = 0; u < PCG_NW * TW / 256; ++u) {
            const int e = tid + 256 * u, k = e / TW, c = e - TW * k;
            wr_[u] = W[(size_t)k * ld + col0 + (c < ncol ? c : 0)];
        }
        __syncthreads();                                         // (the previous tile's phase 2 is done with T, rows, wt)
#pragma unroll
        for (int u = 0; u < 6 * TW / 256; ++u) {
            const int e = tid + 256 * u, r = e / TW, c = e - TW * r;
            rows[e] = (row0 + r < row1 && c < ncol) ? (double)fr_[u] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < PCG_NW * TW / 256; ++u) {
            const int e = tid + 256 * u, c = e % TW;
            wt[e] = c < ncol ? (float)wr_[u] : 0.0f;
        }
        __syncthreads();
        // (1) T[r][camera][k]
#pragma unroll
        for (int u = 0; u < 6 * SG_TC * PCG_NW / 256; ++u) {
            const int item = tid + 256 * u, r = item / (SG_TC * PCG_NW), rem = item - (SG_TC * PCG_NW) * r, cam = rem / PCG_NW, k = rem - PCG_NW * cam;
            const double* rp = rows + r * TW + 6 * cam;
            const float* wk = wt + k * TW + 6 * cam;
            double t = 0.0;
#pragma unroll
            for (int e = 0; e < 6; ++e) t = fma(rp[e], (double)wk[e], t);
            T[item] = t;
        }
        __syncthreads();
        // (2) hat sums
        const int c1 = min(nc, c0 + SG_TC);
        const double* Ta = T + (size_t)ra * SG_TC * PCG_NW;
        const double* Tb = T + (size_t)(have_b ? rb : ra) * SG_TC * PCG_NW;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const int v = lane + 64 * p;
            if (v < NC - 1) {
                const int g = v / 7, k = v - 7 * g;
                for (int range = 0; range < 2; ++range) {
                    const int a = range == 0 ? (g + G - 1) % G : g;
                    const int lo = max(sg_first_cam(a, nc, G), c0), hi = min(sg_first_cam(a + 1, nc, G), c1);
                    for (int j = lo; j < hi; ++j) {
                        const double fr = sg_frac(j, a, nc, G, inv_nc);
                        const double wgt = range == 0 ? fr : 1.0 - fr;
                        acc[0][p] = fma(wgt, Ta[(j - c0) * PCG_NW + k], acc[0][p]);
                        acc[1][p] = fma(wgt, Tb[(j - c0) * PCG_NW + k], acc[1][p]);
                    }
                }
            }
        }
        if (c0 + lane < c1) { gs[0] += Ta[lane * PCG_NW + (PCG_NW - 1)]; gs[1] += Tb[lane * PCG_NW + (PCG_NW - 1)]; }     // the global (focal / depth) vector
    }
    gs[0] = wave_allsum(gs[0]); gs[1] = wave_allsum(gs[1]);
    {   // the focal column
        const double wf = W[(size_t)(PCG_NW - 1) * ld + d - 1];
        if (have_a) gs[0] = fma((double)F[(size_t)(row0 + ra) * ld + d - 1], wf, gs[0]);
        if (have_b) gs[1] = fma((double)F[(size_t)(row0 + rb) * ld + d - 1], wf, gs[1]);
    }
    if (!have_a) { gs[0] = 0.0; acc[0][0] = acc[0][1] = acc[0][2] = 0.0; }
    if (!have_b) { gs[1] = 0.0; acc[1][0] = acc[1][1] = acc[1][2] = 0.0; }
    if (lane + 128 == NC - 1) { acc[0][2] = gs[0]; acc[1][2] = gs[1]; }
    if (lane + 64 == NC - 1) { acc[0][1] = gs[0]; acc[1][1] = gs[1]; }
    if (lane == NC - 1) { acc[0][0] = gs[0]; acc[1][0] = gs[1]; }
    // V[k][:] = sum over the camera's rows of W~_k[row] AW[row][:], wave partials summed through LDS, one k at a time
    double wab[PCG_NW][2];
#pragma unroll
    for (int k = 0; k < PCG_NW; ++k) { wab[k][0] = have_a ? W[(size_t)k * ld + row0 + ra] : 0.0; wab[k][1] = have_b ? W[(size_t)k * ld + row0 + rb] : 0.0; }
#pragma unroll
    for (int k = 0; k < PCG_NW; ++k) {
        const double wa = wab[k][0], wb = wab[k][1];
        __syncthreads();
#pragma unroll
        for (int p = 0; p < 3; ++p) if (lane + 64 * p < SG_NCP) vbuf[w * SG_NCP + lane + 64 * p] = fma(wa, acc[0][p], wb * acc[1][p]);
        __syncthreads();
        if (tid < SG_NCP) V[((size_t)blockIdx.x * PCG_NW + k) * SG_NCP + tid] = (vbuf[tid] + vbuf[SG_NCP + tid]) + (vbuf[2 * SG_NCP + tid] + vbuf[3 * SG_NCP + tid]);
    }
}

// E[i][:]: one workgroup per coarse vector, 3 x SG_NCP threads (a third of the terms each)
__global__ __launch_bounds__(3 * SG_NCP) void k_sg_e(int d, int G, const double* __restrict__ V, double* __restrict__ E) {
    __shared__ double eq[3][SG_NCP];
    const int i = blockIdx.x, tid = threadIdx.x, v = tid % SG_NCP, part = tid / SG_NCP;
    const int nc = (d - 1) / 6, NC = 7 * G + 1;
    const double inv_nc = 1.0 / (double)nc;
    double e = 0.0;
    if (i < NC - 1) {
        const int g = i / 7, k = i - 7 * g;
        for (int range = 0; range < 2; ++range) {
            const int a = range == 0 ? (g + G - 1) % G : g;
            const int lo = sg_first_cam(a, nc, G), hi = sg_first_cam(a + 1, nc, G);
            for (int jb = lo + part; jb < hi; jb += 3 * 8) {               // eight loads in flight (clamped, branch-free)
                double val[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) { const int j = jb + 3 * t; val[t] = V[((size_t)(j < hi ? j : lo) * PCG_NW + k) * SG_NCP + v]; }
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const int j = jb + 3 * t;
                    const double fr = sg_frac(j, a, nc, G, inv_nc);
                    e = fma(j < hi ? (range == 0 ? fr : 1.0 - fr) : 0.0, val[t], e);
                }
            }
        }
    } else {
        for (int jb = part; jb <= nc; jb += 3 * 16) {                       // the plain sum over all workgroups: sixteen loads in flight
            double val[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) { const int j = jb + 3 * t; val[t] = V[((size_t)(j <= nc ? j : nc) * PCG_NW + (PCG_NW - 1)) * SG_NCP + v]; }
#pragma unroll
            for (int t = 0; t < 16; ++t) e += (jb + 3 * t <= nc) ? val[t] : 0.0;
        }
    }
    eq[part][v] = e;
    __syncthreads();
    if (tid < SG_NCP) E[(size_t)i * SG_NCP + tid] = (eq[0][tid] + eq[1][tid]) + eq[2][tid];
}

// E^-1 (SG_NCP x SG_NCP): gj_invert_tiled with 6 x 4 tiles on 864 threads (24 entries per thread).  (History: 27 hats -- 192 rows, a slice of 48 or 40
// columns per thread -- spilled two dozen doubles per thread in every step whatever the scheduling hints: 1.3 ms; hence the limit of 20 hats.  21
// columns per thread on 1008 threads: 219 us, bound by the LDS reads of the pivot row.)
constexpr int SG_ITR = 6, SG_ITC = 4, SG_ITHREADS = (SG_NCP / SG_ITR) * (SG_NCP / SG_ITC);
__global__ __launch_bounds__(SG_ITHREADS) void k_sg_invert(int NC, const double* __restrict__ E, double* __restrict__ einv) {
    __shared__ double rowbuf[2 * SG_NCP], colbuf[2 * SG_NCP], sc[SG_NCP], diagbuf[2];
    __shared__ unsigned char drop[SG_NCP + 1];
    gj_invert_tiled<SG_NCP, SG_ITR, SG_ITC, 12>(NC, E, einv, rowbuf, colbuf, sc, diagbuf, drop);
}

// q = S~ p for the rows of this workgroup (eight; two per wave at a time), partial of p . q
template <typename FT>
__global__ __launch_bounds__(256) void k_sg_q(int d, int ld, const FT* __restrict__ F, const double* __restrict__ p, double* __restrict__ q,
                                              double* __restrict__ pqpart, const int* __restrict__ flags, int rows_per_wg) {
    extern __shared__ __align__(16) double sm[];
    double* pl = sm;
    __shared__ double red[4];
    if (flags[PF_DONE] != 0) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int row0 = blockIdx.x * rows_per_wg, row1 = min(d, row0 + rows_per_wg);
    for (int e0 = tid; e0 < ld; e0 += 256 * 8) {
        double pv8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int e = e0 + 256 * u; pv8[u] = p[e < d ? e : d - 1]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int e = e0 + 256 * u; if (e < ld) pl[e] = e < d ? pv8[u] : 0.0; }
    }
    __syncthreads();
    double pqp = 0.0;
    const int nd2 = d >> 1, nd4 = d >> 2;
    for (int row = row0 + w; row < row1; row += 8) {
        const int rowb = (row + 4 < row1) ? row + 4 : row;
        double sa = 0.0, sb = 0.0;
        if (sizeof(FT) == 8) {
            const double2* pl2 = reinterpret_cast<const double2*>(pl);
            const double2* Fa = reinterpret_cast<const double2*>(F + (size_t)row * ld);
            const double2* Fb = reinterpret_cast<const double2*>(F + (size_t)rowb * ld);
            int c = lane;
            for (; c + 192 < nd2; c += 256) {
                double2 a[4], b[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) { a[m] = Fa[c + 64 * m]; b[m] = Fb[c + 64 * m]; }
#pragma unroll
                for (int m = 0; m < 4; ++m) { const double2 pv = pl2[c + 64 * m]; sa += a[m].x * pv.x + a[m].y * pv.y; sb += b[m].x * pv.x + b[m].y * pv.y; }
            }
            for (; c < nd2; c += 64) { const double2 a = Fa[c], b = Fb[c], pv = pl2[c]; sa += a.x * pv.x + a.y * pv.y; sb += b.x * pv.x + b.y * pv.y; }
            if ((d & 1) && lane == 0) { sa += (double)F[(size_t)row * ld + d - 1] * pl[d - 1]; sb += (double)F[(size_t)rowb * ld + d - 1] * pl[d - 1]; }
        } else {
            const float4* Fa = reinterpret_cast<const float4*>(F + (size_t)row * ld);
            const float4* Fb = reinterpret_cast<const float4*>(F + (size_t)rowb * ld);
            int c = lane;
            for (; c + 192 < nd4; c += 256) {
                float4 a[4], b[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) { a[m] = Fa[c + 64 * m]; b[m] = Fb[c + 64 * m]; }
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const double2 p0 = reinterpret_cast<const double2*>(pl)[2 * (c + 64 * m)], p1 = reinterpret_cast<const double2*>(pl)[2 * (c + 64 * m) + 1];
                    sa += (double)a[m].x * p0.x + (double)a[m].y * p0.y + (double)a[m].z * p1.x + (double)a[m].w * p1.y;
                    sb += (double)b[m].x * p0.x + (double)b[m].y * p0.y + (double)b[m].z * p1.x + (double)b[m].w * p1.y;
                }
            }
            for (; c < nd4; c += 64) {
                const float4 a = Fa[c], b = Fb[c];
                const double2 p0 = reinterpret_cast<const double2*>(pl)[2 * c], p1 = reinterpret_cast<const double2*>(pl)[2 * c + 1];
                sa += (double)a.x * p0.x + (double)a.y * p0.y + (double)a.z * p1.x + (double)a.w * p1.y;
                sb += (double)b.x * p0.x + (double)b.y * p0.y + (double)b.z * p1.x + (double)b.w * p1.y;
            }
            if (lane == 0) for (int cc = 4 * nd4; cc < d; ++cc) { sa += (double)F[(size_t)row * ld + cc] * pl[cc]; sb += (double)F[(size_t)rowb * ld + cc] * pl[cc]; }
        }
        sa = wave_allsum(sa); sb = wave_allsum(sb);
        if (lane == 0) {
            q[row] = sa; pqp += pl[row] * sa;
            if (rowb != row) { q[rowb] = sb; pqp += pl[rowb] * sb; }
        }
    }
    if (lane == 0) red[w] = pqp;
    __syncthreads();
    if (tid == 0) pqpart[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// The same product for a SPARSELY FILLED S~ (a long camera path: 6 % of the blocks hold anything): workgroup = camera (six rows; the last workgroup: the
// focal row, dense), the cameras it shares a non-empty block with come from the structure build's bit mask (k_block_mask; the camera itself included:
// S~_jj = I), compacted into LDS; lanes walk (neighbour, entry) pairs -- six consecutive lanes read the 24 / 48 contiguous bytes of a block row -- and
// read p from L2.  Empty blocks of S~ are exact zeros (the pair pass writes them), so this IS the dense product.
template <typename FT>
__global__ __launch_bounds__(256) void k_sg_q_sparse(int d, int ld, const FT* __restrict__ F, const double* __restrict__ p, double* __restrict__ q,
                                                     double* __restrict__ pqpart, const int* __restrict__ flags, const unsigned* __restrict__ mask) {
    __shared__ int list[1024];
    __shared__ int wcount[4], nn_s;
    __shared__ double red[4];
    if (flags[PF_DONE] != 0) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nc = (d - 1) / 6, words = (nc + 31) / 32, ja = blockIdx.x;
    double pqp = 0.0;
    if (ja < nc) {
        // compact the set bits of mask[ja] into list (ascending): ballot + popcount per wave, wave offsets through LDS
        int base = 0;
        for (int c0 = 0; c0 < nc; c0 += 256) {
            const int c = c0 + tid;
            const bool on = c < nc && ((mask[(size_t)ja * words + (c >> 5)] >> (c & 31)) & 1u);
            const unsigned long long bal = __ballot(on);
            if (lane == 0) wcount[w] = __popcll(bal);
            __syncthreads();
            int off = base;
            for (int k = 0; k < w; ++k) off += wcount[k];
            if (on) list[off + __popcll(bal & ((1ull << lane) - 1ull))] = c;
            base += wcount[0] + wcount[1] + wcount[2] + wcount[3];
            __syncthreads();
        }
        if (tid == 0) nn_s = base;
        __syncthreads();
        const int nn = nn_s;
        const double pf = p[d - 1];
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int r = w + 4 * rr;
            if (r < 6) {
                const int row = 6 * ja + r;
                const FT* Fr = F + (size_t)row * ld;
                double s = 0.0;
                for (int idx0 = lane; idx0 < 6 * nn; idx0 += 64 * 4) {          // four gathers in flight
                    double fv[4], pv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int idx = idx0 + 64 * u, ic = idx < 6 * nn ? idx : 0;
                        const int col = 6 * list[ic / 6] + ic % 6;
                        fv[u] = (double)Fr[col]; pv[u] = p[col];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) s = fma(idx0 + 64 * u < 6 * nn ? fv[u] : 0.0, pv[u], s);
                }
                if (lane == 0) s = fma((double)Fr[d - 1], pf, s);
                s = wave_allsum(s);
                if (lane == 0) { q[row] = s; pqp += p[row] * s; }
            }
        }
    } else if (w == 0) {                             // the focal row: dense
        const FT* Fr = F + (size_t)(d - 1) * ld;
        double s = 0.0;
        for (int c = lane; c < d; c += 64) s = fma((double)Fr[c], p[c], s);
        s = wave_allsum(s);
        if (lane == 0) { q[d - 1] = s; pqp = p[d - 1] * s; }
    }
    if (lane == 0) red[w] = pqp;
    __syncthreads();
    if (tid == 0) pqpart[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// block sum over SG_UT threads (every thread gets the total); scratch: 16 doubles
__device__ __forceinline__ double sg_block_sum(double v, double* scratch) {
    v = wave_allsum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < SG_UT / 64; ++k) s += scratch[k];
    return s;
}
// cameras (and the focal pseudo-camera nc) of vector workgroup g: [sg_cam0(g), sg_cam0(g + 1)) of nc + 1
__device__ __forceinline__ int sg_cam0(int g, int nc) { return (int)(((long long)g * (nc + 1)) / SG_UWG); }

// INIT: x = 0, r = b~.  Else: alpha = r.z / p.q; x += alpha p; r -= alpha q.  Both: t_k per camera, the workgroup's share of |r|^2.
template <bool INIT>
__global__ __launch_bounds__(SG_UT) void k_sg_u(int d, int ld, int nwgq, const double* __restrict__ bt, double* __restrict__ x, double* __restrict__ r,
                                                const double* __restrict__ p, const double* __restrict__ q, const double* __restrict__ W,
                                                const double* __restrict__ pqpart, double* __restrict__ tcam, double* __restrict__ rrpart,
                                                double* __restrict__ state, const int* __restrict__ flags, int in) {
    __shared__ double scratch[16];
    __shared__ double rl[6 * 64 + 8];                 // the workgroup's rows of the new r (at most ceil(1001 / 16) = 63 cameras)
    if (!INIT && flags[PF_DONE] != 0) return;
    const int tid = threadIdx.x;
    const int nc = (d - 1) / 6;
    const int j0 = sg_cam0(blockIdx.x, nc), j1 = sg_cam0(blockIdx.x + 1, nc);
    const int r0 = 6 * j0, r1 = min(d, 6 * j1), nrows = r1 - r0;
    double alpha = 0.0;
    if (!INIT) {
        double s = 0.0;
        for (int wg = tid; wg < nwgq; wg += SG_UT) s += pqpart[wg];
        const double pq = sg_block_sum(s, scratch);
        alpha = state[SGS_LEN * in + SGS_RZ] / pq;
        if (blockIdx.x == 0 && tid == 0) state[SGS_LEN * in + SGS_PQ] = pq;
    }
    double rr = 0.0;
    if (tid < nrows) {
        const int row = r0 + tid;
        double rn;
        if (INIT) { rn = bt[row]; x[row] = 0.0; }
        else { x[row] += alpha * p[row]; rn = r[row] - alpha * q[row]; }
        r[row] = rn;
        rl[tid] = rn;
        rr = rn * rn;
    }
    rr = sg_block_sum(rr, scratch);                   // (also the barrier behind rl)
    if (tid == 0) rrpart[blockIdx.x] = rr;
    if (tid < PCG_NW * (j1 - j0)) {
        const int jl = tid / PCG_NW, k = tid - PCG_NW * jl, j = j0 + jl;
        double t = 0.0;
        if (j < nc) {
#pragma unroll
            for (int e = 0; e < 6; ++e) t = fma(W[(size_t)k * ld + 6 * j + e], rl[6 * jl + e], t);
        } else t = W[(size_t)k * ld + d - 1] * rl[6 * jl];          // the focal row
        tcam[(size_t)j * PCG_NW + k] = t;
    }
}

// |r|^2 and the stopping test; c = W~^T r (hat sums of the t_k), mu = E^-1 c, r . z = |r|^2 + c . mu, beta; p = r + W~ mu + beta p on the workgroup's rows
template <bool INIT>
__global__ __launch_bounds__(SG_UT) void k_sg_p(int d, int ld, int G, const double* __restrict__ r, double* __restrict__ p, const double* __restrict__ W,
                                                const double* __restrict__ tcam, const double* __restrict__ rrpart, const double* __restrict__ einv,
                                                double* __restrict__ state, double* __restrict__ scal, int* __restrict__ flags, int* info, int* mailbox,
                                                double tol2, int in, int anchor, double cap) {
    __shared__ double scratch[16];
    __shared__ double cl[SG_NCP], ml[SG_NCP], mp[4][SG_NCP];
    if (!INIT && flags[PF_DONE] != 0) return;
    const int tid = threadIdx.x, out = in ^ 1;
    const int nc = (d - 1) / 6, NC = 7 * G + 1;
    const double inv_nc = 1.0 / (double)nc;
    double rr = 0.0;
#pragma unroll
    for (int k = 0; k < SG_UWG; ++k) rr += rrpart[k];
    if (INIT) {
        if (blockIdx.x == 0 && tid == 0) {
            scal[PS_RR0] = pcg_threshold_base(rr, scal, anchor, cap); flags[PF_DONE] = (rr == 0.0); flags[PF_ITERS] = 0; flags[PF_XBUF] = 0;
            if (mailbox && rr == 0.0) pcg_post(mailbox, 0, 1);
        }
        if (rr == 0.0) return;
    } else {
        const double pq = state[SGS_LEN * in + SGS_PQ];
        const bool broke = !(pq > 0.0) || !(rr == rr);
        if (rr <= tol2 * scal[PS_RR0] || broke) {
            if (blockIdx.x == 0 && tid == 0) {
                flags[PF_XBUF] = 0; const int it = flags[PF_ITERS] + 1; flags[PF_ITERS] = it;
                if (broke) atomicCAS(info, 0, d + 1);
                __threadfence();
                flags[PF_DONE] = 1;
                if (mailbox) pcg_post(mailbox, it, 1);
            }
            return;
        }
    }
    // c: vector i = tid % SG_NCP, a quarter of its hat's cameras per part = tid / SG_NCP (eight loads in flight); the global vector by a block sum
    {
        double t7 = 0.0;
        for (int j = tid; j <= nc; j += SG_UT) t7 += tcam[(size_t)j * PCG_NW + (PCG_NW - 1)];
        t7 = sg_block_sum(t7, scratch);
        double c = 0.0;
        const int i = tid % SG_NCP, part = tid / SG_NCP;
        if (part < 4 && i < NC - 1) {
            const int g = i / 7, k = i - 7 * g;
            const int a = (part >> 1) == 0 ? (g + G - 1) % G : g;
            const int lo = sg_first_cam(a, nc, G), hi = sg_first_cam(a + 1, nc, G);
            const int half = (hi - lo + 1) >> 1;
            const int jb0 = (part & 1) ? lo + half : lo, jb1 = (part & 1) ? hi : min(hi, lo + half);
            for (int jb = jb0; jb < jb1; jb += 8) {
                double val[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) val[t] = tcam[(size_t)(jb + t < jb1 ? jb + t : jb1 - 1) * PCG_NW + k];
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const double fr = sg_frac(jb + t, a, nc, G, inv_nc);
                    c = fma(jb + t < jb1 ? ((part >> 1) == 0 ? fr : 1.0 - fr) : 0.0, val[t], c);
                }
            }
        }
        if (part < 4) mp[part][i] = c;
        __syncthreads();
        if (tid < SG_NCP) cl[tid] = tid < NC - 1 ? (mp[0][tid] + mp[1][tid]) + (mp[2][tid] + mp[3][tid]) : tid == NC - 1 ? t7 : 0.0;
    }
    __syncthreads();
    if (tid < 4 * SG_NCP) {
        const int i = tid % SG_NCP, part = tid / SG_NCP;
        double s = 0.0;
#pragma unroll 8
        for (int jj = 0; jj < SG_CB; ++jj) { const int j = SG_CB * part + jj; s = fma(einv[(size_t)j * SG_NCP + i], cl[j], s); }
        mp[part][i] = s;
    }
    __syncthreads();
    if (tid < SG_NCP) ml[tid] = (mp[0][tid] + mp[1][tid]) + (mp[2][tid] + mp[3][tid]);
    __syncthreads();
    const double cmu = sg_block_sum(tid < SG_NCP ? cl[tid] * ml[tid] : 0.0, scratch);
    const double rz_new = rr + cmu;
    const double beta = INIT ? 0.0 : rz_new / state[SGS_LEN * in + SGS_RZ];
    // p on the rows of this workgroup's cameras
    const int j0 = sg_cam0(blockIdx.x, nc), j1 = sg_cam0(blockIdx.x + 1, nc);
    const int r0 = 6 * j0, r1 = min(d, 6 * j1);
    if (r0 + tid < r1) {
        const int row = r0 + tid;
        const int j = min(row / 6, nc - 1);           // (the focal row: W~_0..6 are zero there)
        const int gl = (j * G) / nc, gh = gl + 1 == G ? 0 : gl + 1;
        const double fr = sg_frac(j, gl, nc, G, inv_nc);
        double z = fma(W[(size_t)(PCG_NW - 1) * ld + row], ml[NC - 1], r[row]);
#pragma unroll
        for (int k = 0; k < 7; ++k) z = fma(W[(size_t)k * ld + row], fma(fr, ml[7 * gh + k], (1.0 - fr) * ml[7 * gl + k]), z);
        p[row] = INIT ? z : fma(beta, p[row], z);
    }
    if (blockIdx.x == 0 && tid == 0) {
        state[SGS_LEN * out + SGS_RZ] = rz_new;
        if (!INIT) { const int it = flags[PF_ITERS] + 1; flags[PF_ITERS] = it; if (mailbox) pcg_post(mailbox, it, 0); }
    }
}

// solution of the original system: z = Lb^-T x~
__global__ void k_pcg_finish(int d, int ld, const double* __restrict__ vec, const double* __restrict__ linv, const int* flags,
                             double* __restrict__ z) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= d) return;
    const double* x = vec + (size_t)(flags[PF_XBUF]) * ld;   // x buffers are vec[0], vec[1]
    const int nb6 = (d - 1) / 6;
    const int b = e / 6;
    if (b >= nb6) { z[e] = linv[(size_t)nb6 * 36 + (e - 6 * nb6)] * x[e]; return; }
    const int c = e - 6 * b;
    double v = 0.0;
    for (int t = c; t < 6; ++t) v += linv[(size_t)b * 36 + t * 6 + c] * x[6 * b + t];
    z[e] = v;
}

// one CG launch of the running solve (init = the first one)
template <bool INIT>
static void launch_cg_iteration(hipStream_t s, DenseSolver* ws, int anchor, double cap) {
    const DenseSolver::CgRun& r = ws->run;
    const int d = ws->d, ld = ws->ld;
    double* bt = ws->vec + (size_t)8 * ld;
    const int in = INIT ? 0 : (r.in | ((r.launched + 1) << 1));     // launch number 1.. of this solve, see k_pcg_iter
#define CG_ARGS(Fptr) d, ld, Fptr, ws->vec, bt, ws->part, ws->scal, ws->flags, r.rows_per_wg, r.tol2, in, r.info, ws->d_mailbox, anchor, cap, ws->W, ws->AW, ws->coarse
    if (r.sg) {
        const int G = sg_hats((d - 1) / 6);
        double *x = ws->vec, *rv = ws->vec + (size_t)2 * ld, *pv = ws->vec + (size_t)4 * ld, *qv = ws->vec + (size_t)6 * ld;      // the parity-0 buffers of pcg_vec
        const int par = INIT ? 0 : r.in;
        if (INIT) {
            hipLaunchKernelGGL((k_sg_u<true>), dim3(SG_UWG), dim3(SG_UT), 0, s, d, ld, r.nwg, bt, x, rv, pv, qv, ws->W, ws->part, ws->sgT, ws->sgRR, ws->sgState, ws->flags, par);
            hipLaunchKernelGGL((k_sg_p<true>), dim3(SG_UWG), dim3(SG_UT), 0, s, d, ld, G, rv, pv, ws->W, ws->sgT, ws->sgRR, ws->sgEinv, ws->sgState, ws->scal, ws->flags, r.info, ws->d_mailbox, r.tol2, par, anchor, cap);
        } else {
            // (a reduced matrix filled below a quarter: the block-sparse product, one workgroup per camera)
            const bool sparse = ws->blk_mask != nullptr && ws->blk_fill < 0.25 && (d - 1) / 6 + 1 <= PCG_PART;
            const int nwgq = sparse ? (d - 1) / 6 + 1 : r.nwg;
            if (sparse) {
                if (r.f32) hipLaunchKernelGGL((k_sg_q_sparse<float>), dim3(nwgq), dim3(256), 0, s, d, ld, ws->Sfull32, pv, qv, ws->part, ws->flags, ws->blk_mask);
                else hipLaunchKernelGGL((k_sg_q_sparse<double>), dim3(nwgq), dim3(256), 0, s, d, ld, ws->Sfull, pv, qv, ws->part, ws->flags, ws->blk_mask);
            } else if (r.f32) hipLaunchKernelGGL((k_sg_q<float>), dim3(r.nwg), dim3(256), r.lds, s, d, ld, ws->Sfull32, pv, qv, ws->part, ws->flags, r.rows_per_wg);
            else hipLaunchKernelGGL((k_sg_q<double>), dim3(r.nwg), dim3(256), r.lds, s, d, ld, ws->Sfull, pv, qv, ws->part, ws->flags, r.rows_per_wg);
            hipLaunchKernelGGL((k_sg_u<false>), dim3(SG_UWG), dim3(SG_UT), 0, s, d, ld, nwgq, bt, x, rv, pv, qv, ws->W, ws->part, ws->sgT, ws->sgRR, ws->sgState, ws->flags, par);
            hipLaunchKernelGGL((k_sg_p<false>), dim3(SG_UWG), dim3(SG_UT), 0, s, d, ld, G, rv, pv, ws->W, ws->sgT, ws->sgRR, ws->sgEinv, ws->sgState, ws->scal, ws->flags, r.info, ws->d_mailbox, r.tol2, par, anchor, cap);
        }
    } else if (r.ml) {
        hipLaunchKernelGGL((k_pcg_iter_ml<INIT>), dim3(r.nwg), dim3(256), r.lds, s, d, ld, ws->Sfull, ws->vec, bt, ws->part, ws->scal, ws->flags, r.tol2, in, r.info,
                           ws->d_mailbox, anchor, cap, ws->W, ws->mlAW, ws->mlEinv, ws->mlC0, ws->mlState);
    } else if (r.fast) {
        // (with the coarse space the first launch is also the first iteration: MODE 2, see k_pcg_iter_fast)
        if (r.coarse) hipLaunchKernelGGL((k_pcg_iter_fast<INIT ? 2 : 0, true>), dim3(r.nwg), dim3(256), r.lds, s, CG_ARGS(ws->Sfull), ws->epart);
        else hipLaunchKernelGGL((k_pcg_iter_fast<INIT ? 1 : 0, false>), dim3(r.nwg), dim3(256), r.lds, s, CG_ARGS(ws->Sfull), ws->epart);
    } else if (r.sym) {
        // two launches per iteration: the vector half, then one workgroup per tile of the upper triangle.  (ONE launch per iteration was built and measured --
        // the Chronopoulos - Gear form of the CG, whose inner products are all sums of per-workgroup partials, so that no launch needs a pass over the vectors:
        // correct, and 27 us per iteration against 17.9 + 7.5 here: profiles/r06_ab_sy_one_launch.txt, commit "k_sy_cg".)
        const int nvec = std::max(1, std::min(SY_VEC_WG, (d + 63) / 64)), nslice = std::min(ws->sym_ntiles, 256);
#define SYV_ARGS d, ld, ws->vec, ws->q3, bt, ws->sym_part, ws->scal, ws->flags, r.tol2, in, r.info, ws->d_mailbox, anchor, cap, ws->W, ws->coarse
#define SYV(C, E) hipLaunchKernelGGL((k_sy_vec<INIT, C, E>), dim3(nvec), dim3(256), 0, s, SYV_ARGS)
        const int ept = INIT ? 0 : (d <= 256 * 8 ? 8 : d <= 256 * 16 ? 16 : d <= 256 * 24 ? 24 : d <= 256 * 32 ? 32 : 0);
        if (r.coarse) { if (ept == 8) SYV(true, 8); else if (ept == 16) SYV(true, 16); else if (ept == 24) SYV(true, 24); else if (ept == 32) SYV(true, 32); else SYV(true, 0); }
        else { if (ept == 8) SYV(false, 8); else if (ept == 16) SYV(false, 16); else if (ept == 24) SYV(false, 24); else if (ept == 32) SYV(false, 32); else SYV(false, 0); }
#undef SYV
#undef SYV_ARGS
#define SY_ARGS(Fptr) d, ld, Fptr, ws->vec, ws->q3, ws->sym_part, ws->scal, ws->flags, ws->sym_tiles, in, nslice, ws->AWt
        if (r.f32) { if (r.coarse) hipLaunchKernelGGL((k_sy_prod<float, true>), dim3(ws->sym_ntiles), dim3(256), 0, s, SY_ARGS(ws->Sfull32));
                     else hipLaunchKernelGGL((k_sy_prod<float, false>), dim3(ws->sym_ntiles), dim3(256), 0, s, SY_ARGS(ws->Sfull32)); }
        else { if (r.coarse) hipLaunchKernelGGL((k_sy_prod<double, true>), dim3(ws->sym_ntiles), dim3(256), 0, s, SY_ARGS(ws->Sfull));
               else hipLaunchKernelGGL((k_sy_prod<double, false>), dim3(ws->sym_ntiles), dim3(256), 0, s, SY_ARGS(ws->Sfull)); }
#undef SY_ARGS
    } else if (r.f32) {
        if (r.coarse) hipLaunchKernelGGL((k_pcg_iter<INIT, float, true>), dim3(r.nwg), dim3(256), r.lds, s, CG_ARGS(ws->Sfull32));
        else hipLaunchKernelGGL((k_pcg_iter<INIT, float, false>), dim3(r.nwg), dim3(256), r.lds, s, CG_ARGS(ws->Sfull32));
    } else {
        if (r.coarse) hipLaunchKernelGGL((k_pcg_iter<INIT, double, true>), dim3(r.nwg), dim3(256), r.lds, s, CG_ARGS(ws->Sfull));
        else hipLaunchKernelGGL((k_pcg_iter<INIT, double, false>), dim3(r.nwg), dim3(256), r.lds, s, CG_ARGS(ws->Sfull));
    }
#undef CG_ARGS
}

int dense_pcg_more(hipStream_t s, DenseSolver* ws, int n, Profiler* prof) {
    DenseSolver::CgRun& r = ws->run;
    n = std::min(n, r.max_iters - r.launched);
    if (n <= 0) return 0;
    ProfScope psb(prof, KID_PCG_ITER, s, n);
    for (int b = 0; b < n; ++b) {
        launch_cg_iteration<false>(s, ws, 0, 1.0);
        r.in ^= 1;
        ++r.launched;
    }
    return n;
}

void dense_pcg_note(DenseSolver* ws, int hist_key, int iters) {
    ws->last_iters = iters;
    if (hist_key >= 0) {
        if (hist_key >= (int)ws->hist.size()) ws->hist.resize((size_t)hist_key + 1, 0);
        ws->hist[hist_key] = iters;
    }
}

// cap of the anchored stopping rule: every solve at least max(tol, 1e-4) relative
static double pcg_cap(double tol) { const double t2 = tol * tol; return t2 > 0.0 ? fmax(t2, 1e-8) / t2 : 1.0; }

// the segmented coarse space needs the fast path's geometry with one workgroup per camera (d = 6 nc + 1) and enough cameras per hat
bool dense_pcg_segments_applicable(const DenseSolver* ws) {
    const int d = ws->d, nc = (d - 1) / 6;
    return d == 6 * nc + 1 && nc >= ML_MIN_CAMS && nc + 1 <= PCG_MAXWG && d <= 256 * PCG_EPT && d <= 64 * PCG_CPL;
}


bool dense_pcg_segments_streaming_applicable(const DenseSolver* ws) {
    const int d = ws->d, nc = (d - 1) / 6;
    const bool fast = d <= 256 * PCG_EPT && d <= 64 * PCG_CPL;
    return d == 6 * nc + 1 && nc >= ML_MIN_CAMS && !fast && (d + 7) / 8 <= PCG_PART && nc + 1 <= 64 * SG_UWG - SG_UWG;
}

// same path selection as dense_pcg_solve
static void pcg_geometry(const DenseSolver* ws, bool* fast, bool* f32) {
    const int d = ws->d;
    const int rows_per_wg = (d + PCG_MAXWG - 1) / PCG_MAXWG;
    *fast = d <= 256 * PCG_EPT && d <= 64 * PCG_CPL && rows_per_wg <= 4 * PCG_RPW;
    *f32 = !*fast && ws->use_f32 && ws->Sfull32 != nullptr;
}

bool dense_pcg_symmetric_applicable(const DenseSolver* ws) {
    bool fast, f32;
    pcg_geometry(ws, &fast, &f32);
    return !fast;
}

int dense_pcg_transform(hipStream_t s, DenseSolver* ws, double* S, double* rhs, int* info_dev, Profiler* prof) {
    if (dense_pcg_ensure_workspace(ws)) return -1;
    const int ld = ws->ld, d = ws->d;
    const int nb6 = (d - 1) / 6, nB = nb6 + (d - 6 * nb6);
    bool fast, f32;
    pcg_geometry(ws, &fast, &f32);
    double* bt = ws->vec + (size_t)8 * ld;
    ProfScope ps(prof, KID_PCG_SETUP, s);
    hipLaunchKernelGGL(k_pcg_blockchol, dim3((nB + 63) / 64), dim3(64), 0, s, S, ld, d, ws->binv, info_dev);
    if (f32) hipLaunchKernelGGL(k_pcg_transform<float>, dim3((nB + 63) / 64, nB), dim3(64), 0, s, S, ld, d, ws->binv, rhs, ws->Sfull32, bt);
    else hipLaunchKernelGGL(k_pcg_transform<double>, dim3((nB + 63) / 64, nB), dim3(64), 0, s, S, ld, d, ws->binv, rhs, ws->Sfull, bt);
    return 0;
}

int dense_pcg_solve(hipStream_t s, DenseSolver* ws, double* S, double* rhs, double tol, int max_iters, int* info_dev, Profiler* prof,
                    bool finish, int hist_key, bool pretransformed, int anchor, bool no_wait, bool coarse, bool segments) {
    const double cap = pcg_cap(tol);
    const int ld = ws->ld, d = ws->d;
    if (dense_pcg_ensure_workspace(ws)) return -1;
    if (max_iters <= 0) max_iters = 4 * d;
    int rows_per_wg = (d + PCG_MAXWG - 1) / PCG_MAXWG;
    const bool fast = d <= 256 * PCG_EPT && d <= 64 * PCG_CPL && rows_per_wg <= 4 * PCG_RPW;
    if (!fast) rows_per_wg = std::max(8, ((d + PCG_MAXWG_BIG - 1) / PCG_MAXWG_BIG + 7) / 8 * 8);   // two rows per wave at a time
    // segmented coarse space (7 x 8 hat-restricted gauge vectors + 1): workgroup = camera
    const bool ml = segments && coarse && fast && dense_pcg_segments_applicable(ws) && ws->W && ws->mlAW;
    if (ml) rows_per_wg = 6;
    // ... its streaming-path form (d > 1280): classical PCG, three launches per iteration
    const bool sg = segments && coarse && !fast && dense_pcg_segments_streaming_applicable(ws) && ws->W && ws->sgV;
    if (sg) rows_per_wg = 8;
    // symmetric streaming path: the caller's pair pass wrote (at least) the upper triangle; one read of it per iteration (k_pcg_iter_sym)
    const bool sym = !fast && !sg && ws->symmetric && ws->sym_tiles != nullptr;
    const int nwg = (d + rows_per_wg - 1) / rows_per_wg;
    const size_t lds = sizeof(double) * (size_t)(ld + (ml ? ML_LDS_TAIL : sg ? 0 : PCG_RED));
    double* bt = ws->vec + (size_t)8 * ld;
    // fp32 storage of S~ on the streaming path whenever the caller asked for it (dense_pcg_want_f32 allocated the buffer)
    const bool f32 = !fast && ws->use_f32 && ws->Sfull32 != nullptr;
    if (!pretransformed) dense_pcg_transform(s, ws, S, rhs, info_dev, prof);
    // coarse space: the caller's linearisation wrote W~ (ws->W); AW, E^-1 and c_0 are formed here, one pass over S~
    coarse = coarse && ws->W && ws->AW && rows_per_wg <= 4 * CO_MAXROWS;
    if (ml) { ProfScope ps(prof, KID_PCG_SETUP, s, 3);
      static bool ml_attr_set = false;
      if (!ml_attr_set) { (void)hipFuncSetAttribute((const void*)k_ml_aw, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ML_AW_LDS); ml_attr_set = true; }
      hipLaunchKernelGGL(k_ml_aw, dim3(nwg), dim3(256), ML_AW_LDS, s, d, ld, ws->Sfull, ws->W, bt, ws->mlAW, ws->mlV, ws->mlU);
      hipLaunchKernelGGL(k_ml_e, dim3(ML_NC), dim3(256), 0, s, d, ws->mlV, ws->mlU, ws->mlE, ws->mlC0);
      hipLaunchKernelGGL(k_ml_invert, dim3(1), dim3(256), 0, s, ws->mlE, ws->mlEinv, ws->mlC0);
      }
    else if (sg) { ProfScope ps(prof, KID_PCG_SETUP, s, 3);
      const int nc = (d - 1) / 6, G = sg_hats(nc), NC = 7 * G + 1;
      if (f32) hipLaunchKernelGGL((k_sg_v<float>), dim3(nc + 1), dim3(256), 0, s, d, ld, G, ws->Sfull32, ws->W, ws->sgV);
      else hipLaunchKernelGGL((k_sg_v<double>), dim3(nc + 1), dim3(256), 0, s, d, ld, G, ws->Sfull, ws->W, ws->sgV);
      hipLaunchKernelGGL(k_sg_e, dim3(NC), dim3(3 * SG_NCP), 0, s, d, G, ws->sgV, ws->sgE);
      hipLaunchKernelGGL(k_sg_invert, dim3(1), dim3(SG_ITHREADS), 0, s, NC, ws->sgE, ws->sgEinv);
      }
    else if (coarse && sym) { ProfScope ps(prof, KID_PCG_SETUP, s, 3);
      // the set-up on the one triangle as well: AWt = S~ W~ (zeroed by the linearisation), then E and c_0 from it
      if (f32) hipLaunchKernelGGL((k_sy_coarse<float>), dim3(ws->sym_nctiles), dim3(256), 0, s, d, ld, ws->Sfull32, ws->W, ws->AWt, ws->sym_ctiles);
      else hipLaunchKernelGGL((k_sy_coarse<double>), dim3(ws->sym_nctiles), dim3(256), 0, s, d, ld, ws->Sfull, ws->W, ws->AWt, ws->sym_ctiles);
      hipLaunchKernelGGL(k_sy_e, dim3(SY_VEC_WG), dim3(256), 0, s, d, ld, ws->W, ws->AWt, bt, ws->epart);
      hipLaunchKernelGGL(k_pcg_coarse_invert, dim3(1), dim3(256), 0, s, SY_VEC_WG, ws->epart, ws->coarse); }
    else if (coarse) { ProfScope ps(prof, KID_PCG_SETUP, s, fast ? 1 : 2);
      const int nwg = (d + rows_per_wg - 1) / rows_per_wg;       // (the set-up keeps the row geometry whatever the iteration kernel's grid)
      if (fast) hipLaunchKernelGGL(k_pcg_coarse_fast, dim3(nwg), dim3(256), 0, s, d, ld, ws->Sfull, ws->W, bt, ws->AW, ws->epart, rows_per_wg, ws->vec + (size_t)(2 * 3 + 0) * ld);      // t -> the q buffer of parity 0 (pcg_vec)
      else if (f32) { if (rows_per_wg <= 8) hipLaunchKernelGGL((k_pcg_coarse<float, 2>), dim3(nwg), dim3(256), 0, s, d, ld, ws->Sfull32, ws->W, bt, ws->AW, ws->epart, rows_per_wg);
                      else hipLaunchKernelGGL((k_pcg_coarse<float, CO_MAXROWS>), dim3(nwg), dim3(256), 0, s, d, ld, ws->Sfull32, ws->W, bt, ws->AW, ws->epart, rows_per_wg); }
      else { if (rows_per_wg <= 8) hipLaunchKernelGGL((k_pcg_coarse<double, 2>), dim3(nwg), dim3(256), 0, s, d, ld, ws->Sfull, ws->W, bt, ws->AW, ws->epart, rows_per_wg);
             else hipLaunchKernelGGL((k_pcg_coarse<double, CO_MAXROWS>), dim3(nwg), dim3(256), 0, s, d, ld, ws->Sfull, ws->W, bt, ws->AW, ws->epart, rows_per_wg); }
      // fast path: the first CG launch sums the partials and inverts E itself (one launch fewer per LM iteration)
      if (!fast) hipLaunchKernelGGL(k_pcg_coarse_invert, dim3(1), dim3(256), 0, s, nwg, ws->epart, ws->coarse); }
    volatile int* mb = ws->h_mailbox;
    if (mb) { mb[0] = -1; mb[1] = 0; }
    ws->run.nwg = nwg; ws->run.rows_per_wg = rows_per_wg; ws->run.lds = lds; ws->run.fast = fast; ws->run.f32 = f32; ws->run.coarse = coarse; ws->run.ml = ml; ws->run.sg = sg;
    ws->run.sym = sym;

    ws->run.tol2 = tol * tol; ws->run.in = 1; ws->run.launched = 0; ws->run.max_iters = max_iters; ws->run.info = info_dev;
    { ProfScope ps(prof, KID_PCG_ITER, s);
      launch_cg_iteration<true>(s, ws, anchor, cap); }
    if (fast && coarse && !ml) ws->run.launched = 1;       // the merged first launch IS iteration 1 (max_iters counts it)
    int batch = 24;
    constexpr int batch_extra = 1;
    // history + 1 (was + 2; +0.6 % on the headline): a solve that needs two more iterations than last time costs a host round trip, a surplus (early-exit) launch ~2 us
    // (run to 1e-12 -- AUTO -- a solve takes 13 +- 1 iterations from one call to the next, the atomics' summation order is enough: a batch one
    // launch short costs a host round trip of ~50 us, a surplus launch ~2: one more in reserve there)
    if (hist_key >= 0 && hist_key < (int)ws->hist.size() && ws->hist[hist_key] > 0) batch = ws->hist[hist_key] + batch_extra + (tol < 1e-10 ? 1 : 0) - ((fast && coarse && !ml && !sg) ? 1 : 0);     // (the merged first launch is iteration 1)
    if (no_wait) return dense_pcg_more(s, ws, batch, prof);
    bool done = false;
    while (!done) {
        if (dense_pcg_more(s, ws, batch, prof) == 0) break;
        const int it = ws->run.launched;
        if (mb) {
            // poll the mailbox until the last launch of the batch has reported (or convergence was posted)
            const double t_end = now_s() + 2.0;
            while (!(mb[1] != 0 || mb[0] >= it)) {
                if (now_s() > t_end) { (void)hipStreamSynchronize(s); break; }
            }
            __sync_synchronize();
            done = mb[1] != 0;
            ws->h_flags[PF_DONE] = done; ws->h_flags[PF_ITERS] = mb[0] >= 0 ? mb[0] : it;
        } else {
            (void)hipMemcpyAsync(ws->h_flags, ws->flags, 4 * sizeof(int), hipMemcpyDeviceToHost, s);
            (void)hipStreamSynchronize(s);
            done = ws->h_flags[PF_DONE] != 0;
        }
        batch = 8;
    }
    if (finish) { ProfScope ps(prof, KID_PCG_FINISH, s);
      hipLaunchKernelGGL(k_pcg_finish, dim3((d + 255) / 256), dim3(256), 0, s, d, ld, ws->vec, ws->binv, ws->flags, rhs); }
    ws->last_iters = ws->h_flags[PF_ITERS];
    if (hist_key >= 0) {
        if (hist_key >= (int)ws->hist.size()) ws->hist.resize((size_t)hist_key + 1, 0);
        ws->hist[hist_key] = ws->h_flags[PF_ITERS];
    }
    return ws->h_flags[PF_ITERS];
}

template <typename T> static int ws_alloc(DenseSolver* ws, T** p, size_t bytes) {
    if (ws->arena) { *p = static_cast<T*>(ws->arena->alloc(bytes)); return *p ? 0 : -1; }
    return hipMalloc(reinterpret_cast<void**>(p), bytes) == hipSuccess ? 0 : -1;
}

float* dense_pcg_want_f32(DenseSolver* ws) {
    const int d = ws->d;
    const int rows_per_wg = (d + PCG_MAXWG - 1) / PCG_MAXWG;
    const bool fast = d <= 256 * PCG_EPT && d <= 64 * PCG_CPL && rows_per_wg <= 4 * PCG_RPW;
    if (fast) return nullptr;                    // the fast path is latency-bound: fp64 there
    if (!ws->Sfull32 && ws_alloc(ws, &ws->Sfull32, sizeof(float) * (size_t)ws->d * ws->ld)) return nullptr;
    return ws->Sfull32;
}

int dense_pcg_ensure_workspace(DenseSolver* ws) {
    if (!ws->Sfull) {
        if (ws_alloc(ws, &ws->Sfull, sizeof(double) * (size_t)ws->d * ws->ld)) return -1;
    }
    if (!ws->W) {
        if (ws_alloc(ws, &ws->W, sizeof(double) * (size_t)PCG_NW * ws->ld)) return -1;
        if (ws_alloc(ws, &ws->AW, sizeof(double) * (size_t)PCG_NW * ws->ld)) return -1;
        if (ws_alloc(ws, &ws->epart, sizeof(double) * (size_t)(PCG_NW * PCG_NW + 2 * PCG_NW) * PCG_PART)) return -1;
        if (ws_alloc(ws, &ws->coarse, sizeof(double) * (size_t)(2 * PCG_NW * PCG_NW + PCG_NW))) return -1;
    }
    {
        bool fast, f32;
        pcg_geometry(ws, &fast, &f32);
        if (!fast && !ws->sym_tiles) {
            // tiles of the scalar upper triangle for k_pcg_iter_sym: SY_R rows x SY_C columns, columns aligned to SY_C
            std::vector<int4> tiles;
            for (int r0 = 0; r0 < ws->d; r0 += SY_R) {
                const int nrows = std::min(SY_R, ws->d - r0);
                for (int c0 = (r0 / SY_C) * SY_C; c0 < ws->d; c0 += SY_C) tiles.push_back(make_int4(r0, c0, nrows, c0 < r0 + nrows ? 1 : 0));
            }
            if (ws_alloc(ws, &ws->q3, sizeof(double) * 2 * (size_t)ws->ld)) return -1;
            if (ws_alloc(ws, &ws->sym_part, sizeof(double) * 2 * PCG_NPART * SY_SLOTS * SY_SLOT_STRIDE)) return -1;
            std::vector<int4> ctiles;
            for (int r0 = 0; r0 < ws->d; r0 += SY_CR) {
                const int nrows = std::min(SY_CR, ws->d - r0);
                for (int c0 = (r0 / SY_C) * SY_C; c0 < ws->d; c0 += SY_C) ctiles.push_back(make_int4(r0, c0, nrows, c0 < r0 + nrows ? 1 : 0));
            }
            if (ws_alloc(ws, &ws->sym_ctiles, sizeof(int4) * ctiles.size())) return -1;
            if (hipMemcpy(ws->sym_ctiles, ctiles.data(), sizeof(int4) * ctiles.size(), hipMemcpyHostToDevice) != hipSuccess) return -1;
            ws->sym_nctiles = (int)ctiles.size();
            // what the coarse set-up ADDS into with atomics (the caller's linearisation zeroes it: DeviceBuffers::pcg_zero = sym_zero / sym_zero_n)
            ws->sym_zero_n = (size_t)PCG_NW * ws->ld;
            if (ws_alloc(ws, &ws->sym_zero, sizeof(double) * ws->sym_zero_n)) return -1;
            if (hipMemset(ws->sym_zero, 0, sizeof(double) * ws->sym_zero_n) != hipSuccess) return -1;
            ws->AWt = ws->sym_zero;
            if (ws_alloc(ws, &ws->sym_tiles, sizeof(int4) * tiles.size())) return -1;
            if (hipMemcpy(ws->sym_tiles, tiles.data(), sizeof(int4) * tiles.size(), hipMemcpyHostToDevice) != hipSuccess) return -1;
            ws->sym_ntiles = (int)tiles.size();
        }
    }
    if (!ws->mlAW && dense_pcg_segments_applicable(ws)) {
        const size_t nwg = (size_t)(ws->d - 1) / 6 + 1;
        if (ws_alloc(ws, &ws->mlAW, sizeof(double) * (size_t)ws->ld * ML_N)) return -1;
        if (ws_alloc(ws, &ws->mlV, sizeof(double) * nwg * PCG_NW * ML_N)) return -1;
        if (ws_alloc(ws, &ws->mlU, sizeof(double) * nwg * PCG_NW)) return -1;
        if (ws_alloc(ws, &ws->mlE, sizeof(double) * ML_N * ML_N)) return -1;
        if (ws_alloc(ws, &ws->mlEinv, sizeof(double) * ML_N * ML_N)) return -1;
        if (ws_alloc(ws, &ws->mlC0, sizeof(double) * ML_N)) return -1;
        if (ws_alloc(ws, &ws->mlState, sizeof(double) * 2 * 3 * ML_N)) return -1;
    }
    if (!ws->sgV && dense_pcg_segments_streaming_applicable(ws)) {
        const size_t ncp1 = (size_t)(ws->d - 1) / 6 + 1;
        if (ws_alloc(ws, &ws->sgV, sizeof(double) * ncp1 * PCG_NW * SG_NCP)) return -1;
        if (ws_alloc(ws, &ws->sgE, sizeof(double) * SG_NCP * SG_NCP)) return -1;
        if (ws_alloc(ws, &ws->sgEinv, sizeof(double) * SG_NCP * SG_NCP)) return -1;
        if (ws_alloc(ws, &ws->sgT, sizeof(double) * ncp1 * PCG_NW)) return -1;
        if (ws_alloc(ws, &ws->sgRR, sizeof(double) * SG_UWG)) return -1;
        if (ws_alloc(ws, &ws->sgState, sizeof(double) * 2 * SGS_LEN)) return -1;
    }
    return 0;
}

int dense_solver_create(DenseSolver* ws, int d, int ld, DeviceArena* arena, char* pinned) {
    ws->d = d; ws->ld = ld; ws->arena = arena;
    const int nblk = ld / NB;
    if (ws_alloc(ws, &ws->minv, sizeof(double) * (size_t)nblk * NB * NB)) return -1;
    if (ws_alloc(ws, &ws->y, sizeof(double) * ld)) return -1;
    if (ws_alloc(ws, &ws->vec, sizeof(double) * 9 * (size_t)ld)) return -1;
    if (ws_alloc(ws, &ws->part, sizeof(double) * 2 * PCG_NPART * PCG_PART)) return -1;
    if (ws_alloc(ws, &ws->binv, sizeof(double) * 36 * (size_t)(ld / 6 + 2))) return -1;
    if (ws_alloc(ws, &ws->scal, sizeof(double) * (PS_STATE + 2 * PS_STATE_LEN))) return -1;
    if (ws_alloc(ws, &ws->flags, sizeof(int) * 4)) return -1;
    if (pinned) {
        ws->pinned_external = true;
        ws->h_flags = reinterpret_cast<int*>(pinned);
        ws->h_mailbox = reinterpret_cast<volatile int*>(pinned + 64);
    } else {
        if (hipHostMalloc(reinterpret_cast<void**>(&ws->h_flags), sizeof(int) * 4, hipHostMallocDefault) != hipSuccess) return -1;
        void* hm = nullptr;
        if (hipHostMalloc(&hm, sizeof(int) * 16, hipHostMallocMapped) != hipSuccess) return -1;
        ws->h_mailbox = static_cast<volatile int*>(hm);
    }
    if (hipHostGetDevicePointer(reinterpret_cast<void**>(&ws->d_mailbox), const_cast<int*>(ws->h_mailbox), 0) != hipSuccess) return -1;
    ws->Sfull = nullptr;
    return 0;
}

void dense_solver_destroy(DenseSolver* ws) {
    if (!ws->arena) {
        if (ws->minv) (void)hipFree(ws->minv);
        if (ws->y) (void)hipFree(ws->y);
        if (ws->vec) (void)hipFree(ws->vec);
        if (ws->part) (void)hipFree(ws->part);
        if (ws->binv) (void)hipFree(ws->binv);
        if (ws->scal) (void)hipFree(ws->scal);
        if (ws->flags) (void)hipFree(ws->flags);
        if (ws->Sfull) (void)hipFree(ws->Sfull);
        if (ws->Sfull32) (void)hipFree(ws->Sfull32);
        if (ws->W) (void)hipFree(ws->W);
        if (ws->AW) (void)hipFree(ws->AW);
        if (ws->epart) (void)hipFree(ws->epart);
        if (ws->coarse) (void)hipFree(ws->coarse);
        if (ws->q3) (void)hipFree(ws->q3);
        if (ws->sym_tiles) (void)hipFree(ws->sym_tiles);
        if (ws->sym_part) (void)hipFree(ws->sym_part);
        if (ws->sym_ctiles) (void)hipFree(ws->sym_ctiles);
        if (ws->sym_zero) (void)hipFree(ws->sym_zero);
        for (double* q : { ws->mlAW, ws->mlV, ws->mlU, ws->mlE, ws->mlEinv, ws->mlC0, ws->mlState, ws->sgV, ws->sgE, ws->sgEinv, ws->sgT, ws->sgRR, ws->sgState }) if (q) (void)hipFree(q);
    }
    if (!ws->pinned_external) {
        if (ws->h_flags) (void)hipHostFree(ws->h_flags);
        if (ws->h_mailbox) (void)hipHostFree(const_cast<int*>(ws->h_mailbox));
    }
    *ws = DenseSolver();
}

}  // namespace sfmba
