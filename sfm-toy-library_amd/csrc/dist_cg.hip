// dist_cg.hip -- distributed CG of the sharded solve (see dist_cg.h).  Everything here is small and simple on purpose: the matrix product
// streams 1/N of the matrix per rank, the rest is O(d) work in one workgroup; what a CG iteration costs at N > 1 is its collective.
#include "dist_cg.h"
#include "sfmba_device.h"
#include "coarse_inverse.h"
#include <algorithm>

namespace sfmba {

namespace {

enum { PF_DONE = 0, PF_ITERS = 1, PF_XBUF = 2 };                       // DenseSolver::flags (dense_solver.hip)
enum { DS_RZ = 0, DS_RR0 = 1, DS_RRF = 2, DS_EINV = 8, DS_NV = 80 };  // scal: rz, threshold base, |b~|^2 of the first solve of an anchored run, E^-1 (64), live vectors
constexpr int NW = 8;
constexpr int STEP_T = 1024;

__device__ __forceinline__ long long first_of_row(int I, int ncam) { return (long long)I * ncam - (long long)I * (I + 1) / 2; }   // off-diagonal upper blocks before row I

template <typename FT> __device__ __forceinline__ double ld_val(const FT* p, size_t i) { return (double)p[i]; }

// N block-wide sums (every thread returns all of them); sh: >= 16 * N doubles
template <int N>
__device__ __forceinline__ void block_sums_all(double (&v)[N], double* sh) {
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = wave_allsum(v[k]);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < N; ++k) sh[w * N + k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; ++k) { double s = 0.0; for (int i = 0; i < nw; ++i) s += sh[i * N + k]; v[k] = s; }
}

// Partial product from the OWNED blocks (rows [row0, row1) of the upper triangle, in list order behind `owned`):
//   role A, one workgroup per owned row I:      qa[6 I + r]  = sum_{J > I} sum_c B_IJ[r][c] p[6 J + c]
//   role B, one workgroup per column J > row0:  qb[6 J + c]  = sum_{I in [row0, min(row1, J))} sum_r B_IJ[r][c] p[6 I + r]
template <typename FT>
__global__ __launch_bounds__(256) void k_dcg_spmv(int ncam, int row0, int row1, const FT* __restrict__ owned, const double* __restrict__ p,
                                                  double* __restrict__ qa, double* __restrict__ qb, const int* __restrict__ flags) {
    if (flags && flags[PF_DONE]) return;
    __shared__ double sh[4 * 6];
    const int nA = row1 - row0;
    const bool roleA = (int)blockIdx.x < nA;
    const long long base = first_of_row(row0, ncam);
    double acc[6] = { 0, 0, 0, 0, 0, 0 };
    if (roleA) {
        const int I = row0 + blockIdx.x;
        const long long rowpos = first_of_row(I, ncam) - base;
        for (int J = I + 1 + threadIdx.x; J < ncam; J += blockDim.x) {
            const FT* B = owned + (size_t)(rowpos + (J - I - 1)) * 36;
            double pj[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) pj[c] = p[6 * J + c];
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int c = 0; c < 6; ++c) acc[r] = fma(ld_val(B, 6 * r + c), pj[c], acc[r]);
        }
    } else {
        const int J = row0 + 1 + ((int)blockIdx.x - nA);
        const int iend = row1 < J ? row1 : J;
        for (int I = row0 + threadIdx.x; I < iend; I += blockDim.x) {
            const FT* B = owned + (size_t)(first_of_row(I, ncam) - base + (J - I - 1)) * 36;
            double pi[6];
#pragma unroll
            for (int r = 0; r < 6; ++r) pi[r] = p[6 * I + r];
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int c = 0; c < 6; ++c) acc[c] = fma(ld_val(B, 6 * r + c), pi[r], acc[c]);
        }
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) acc[k] = wave_allsum(acc[k]);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) sh[w * 6 + k] = acc[k];
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const double s = sh[threadIdx.x] + sh[6 + threadIdx.x] + sh[12 + threadIdx.x] + sh[18 + threadIdx.x];
        if (roleA) qa[6 * (row0 + blockIdx.x) + threadIdx.x] = s;
        else qb[6 * (row0 + 1 + ((int)blockIdx.x - nA)) + threadIdx.x] = s;
    }
}

// the rank's partial product as it goes into the all-reduce: owned-block parts + (rank 0 only) what is known everywhere: the identity
// diagonal blocks and the focal row / column of S~
template <typename FF>
__global__ __launch_bounds__(STEP_T) void k_dcg_comb(int d, int ncam, int row0, int row1, int rank, const double* __restrict__ qa, const double* __restrict__ qb,
                                                      const double* __restrict__ p, const FF* __restrict__ focal_row, double* __restrict__ out,
                                                      const int* __restrict__ flags) {
    if (flags && flags[PF_DONE]) return;
    __shared__ double sh[16];
    const int fo = d - 1;
    double fdot[1] = { 0.0 };
    if (rank == 0) { for (int i = threadIdx.x; i < fo; i += blockDim.x) fdot[0] += (double)focal_row[i] * p[i]; }
    block_sums_all<1>(fdot, sh);
    const double pf = p[fo];
    for (int i = threadIdx.x; i < d; i += blockDim.x) {
        const int cam = i / 6;
        double v = 0.0;
        if (i < fo) {
            if (cam >= row0 && cam < row1) v += qa[i];
            if (cam > row0) v += qb[i];
            if (rank == 0) v += p[i] + (double)focal_row[i] * pf;
        } else if (rank == 0) {
            v = pf + fdot[0];
        }
        out[i] = v;
    }
}

// 8 x 8 symmetric E -> E^-1 by Gauss-Jordan with a pivot test (dependent gauge vectors -- fewer cameras than gauge freedoms -- drop out)
__device__ void invert8(const double* E, double* Einv) {
    double a[NW][2 * NW];
    double scale = 0.0;
    for (int i = 0; i < NW; ++i) scale = fmax(scale, fabs(E[i * NW + i]));
    for (int i = 0; i < NW; ++i)
        for (int j = 0; j < NW; ++j) { a[i][j] = E[i * NW + j]; a[i][NW + j] = i == j ? 1.0 : 0.0; }
    bool dead[NW];
    for (int k = 0; k < NW; ++k) {
        const double piv = a[k][k];
        dead[k] = !(piv > 1e-12 * scale);
        if (dead[k]) continue;
        const double ip = 1.0 / piv;
        for (int j = 0; j < 2 * NW; ++j) a[k][j] *= ip;
        for (int i = 0; i < NW; ++i) {
            if (i == k) continue;
            const double f = a[i][k];
            for (int j = 0; j < 2 * NW; ++j) a[i][j] -= f * a[k][j];
        }
    }
    for (int i = 0; i < NW; ++i)
        for (int j = 0; j < NW; ++j) Einv[i * NW + j] = (dead[i] || dead[j]) ? 0.0 : a[i][NW + j];
}

// start of a solve: E^-1 from AW (all-reduced), x = 0, r = b~, z = r + W E^-1 W^T r, p = z, thresholds, flags
__global__ __launch_bounds__(STEP_T) void k_dcg_init(int d, int ld, const double* __restrict__ bt, const double* __restrict__ W, const double* __restrict__ AW,
                                                      double* __restrict__ x, double* __restrict__ r, double* __restrict__ p, double* __restrict__ scal,
                                                      int* __restrict__ flags, int anchor, double cap) {
    __shared__ double sh[16 * 9];
    __shared__ double E[NW * NW], Einv[NW * NW];
    const int tid = threadIdx.x;
    if (W) {
        // E[k][l] = sum_i W[k][i] AW[l][i]: 64 sums, eight at a time
        for (int k = 0; k < NW; ++k) {
            double v[NW];
#pragma unroll
            for (int l = 0; l < NW; ++l) v[l] = 0.0;
            for (int i = tid; i < d; i += blockDim.x) {
                const double w = W[(size_t)k * ld + i];
#pragma unroll
                for (int l = 0; l < NW; ++l) v[l] = fma(w, AW[(size_t)l * ld + i], v[l]);
            }
            block_sums_all<NW>(v, sh);
            if (tid < NW) E[k * NW + tid] = v[tid];
        }
        __syncthreads();
        if (tid == 0) invert8(E, Einv);
        __syncthreads();
        if (tid < NW * NW) scal[DS_EINV + tid] = Einv[tid];
    }
    double v[NW + 1];
#pragma unroll
    for (int k = 0; k <= NW; ++k) v[k] = 0.0;
    for (int i = tid; i < d; i += blockDim.x) {
        const double b = bt[i];
        v[NW] = fma(b, b, v[NW]);
        if (W) {
#pragma unroll
            for (int k = 0; k < NW; ++k) v[k] = fma(W[(size_t)k * ld + i], b, v[k]);
        }
    }
    block_sums_all<NW + 1>(v, sh);
    double mu[NW];
#pragma unroll
    for (int k = 0; k < NW; ++k) { mu[k] = 0.0; if (W) { for (int l = 0; l < NW; ++l) mu[k] = fma(Einv[k * NW + l], v[l], mu[k]); } }
    const double rr = v[NW];
    double rz[1] = { 0.0 };
    for (int i = tid; i < d; i += blockDim.x) {
        const double b = bt[i];
        double z = b;
        if (W) {
#pragma unroll
            for (int k = 0; k < NW; ++k) z = fma(W[(size_t)k * ld + i], mu[k], z);
        }
        x[i] = 0.0; r[i] = b; p[i] = z;
        rz[0] = fma(b, z, rz[0]);
    }
    block_sums_all<1>(rz, sh);
    if (tid == 0) {
        double base = rr;
        if (anchor == 1) scal[DS_RRF] = rr;
        else if (anchor == 2) base = fmin(fmax(rr, scal[DS_RRF]), cap * rr);
        scal[DS_RR0] = base; scal[DS_RZ] = rz[0];
        flags[PF_DONE] = (rr == 0.0); flags[PF_ITERS] = 0; flags[PF_XBUF] = 0;
    }
}

// one CG iteration behind the all-reduced product q = S~ p
__global__ __launch_bounds__(STEP_T) void k_dcg_step(int d, int ld, const double* __restrict__ q, const double* __restrict__ W, double* __restrict__ x,
                                                      double* __restrict__ r, double* __restrict__ p, double* __restrict__ scal, int* __restrict__ flags,
                                                      double tol2, int* info) {
    if (flags[PF_DONE]) return;
    __shared__ double sh[16 * 9];
    const int tid = threadIdx.x;
    double pq[1] = { 0.0 };
    for (int i = tid; i < d; i += blockDim.x) pq[0] = fma(p[i], q[i], pq[0]);
    block_sums_all<1>(pq, sh);
    const double rz = scal[DS_RZ];
    const bool broke = !(pq[0] > 0.0);
    const double alpha = broke ? 0.0 : rz / pq[0];
    double v[NW + 1];
#pragma unroll
    for (int k = 0; k <= NW; ++k) v[k] = 0.0;
    for (int i = tid; i < d; i += blockDim.x) {
        const double rn = fma(-alpha, q[i], r[i]);
        x[i] = fma(alpha, p[i], x[i]);
        r[i] = rn;
        v[NW] = fma(rn, rn, v[NW]);
        if (W) {
#pragma unroll
            for (int k = 0; k < NW; ++k) v[k] = fma(W[(size_t)k * ld + i], rn, v[k]);
        }
    }
    block_sums_all<NW + 1>(v, sh);
    const double rr = v[NW];
    const bool done = broke || !(rr == rr) || rr <= tol2 * scal[DS_RR0];
    if (done) {
        __syncthreads();
        if (tid == 0) { flags[PF_ITERS] = flags[PF_ITERS] + 1; flags[PF_DONE] = 1; if (broke || !(rr == rr)) atomicCAS(info, 0, d + 1); }
        return;
    }
    double mu[NW];
#pragma unroll
    for (int k = 0; k < NW; ++k) { mu[k] = 0.0; if (W) { for (int l = 0; l < NW; ++l) mu[k] = fma(scal[DS_EINV + k * NW + l], v[l], mu[k]); } }
    double rzn[1] = { 0.0 };
    double zl[8];                                     // this thread's entries of z (d <= 8 * STEP_T; larger systems recompute below)
    int m = 0;
    for (int i = tid; i < d; i += blockDim.x, ++m) {
        double z = r[i];
        if (W) {
#pragma unroll
            for (int k = 0; k < NW; ++k) z = fma(W[(size_t)k * ld + i], mu[k], z);
        }
        if (m < 8) zl[m] = z;
        rzn[0] = fma(r[i], z, rzn[0]);
    }
    block_sums_all<1>(rzn, sh);
    const double beta = rzn[0] / rz;
    m = 0;
    for (int i = tid; i < d; i += blockDim.x, ++m) {
        double z;
        if (m < 8) z = zl[m];
        else {
            z = r[i];
            if (W) {
#pragma unroll
                for (int k = 0; k < NW; ++k) z = fma(W[(size_t)k * ld + i], mu[k], z);
            }
        }
        p[i] = fma(beta, p[i], z);
    }
    if (tid == 0) { scal[DS_RZ] = rzn[0]; flags[PF_ITERS] = flags[PF_ITERS] + 1; }
}


// ---------------------------------------------------------------------------------------------------------------------
// The multi-workgroup form (round 5): TWO launches and ONE all-reduce per CG iteration, none of them a single
// workgroup.  k_dcg_init / k_dcg_step / k_dcg_comb above -- one workgroup over the whole vector, two or three dependent block
// reductions each: 120 / 19 / 8 us at d = 6001 -- are what sank the distributed forms of round 4; they stay for the implicit product.
//   k_dcg_prod   a WAVE per camera c: row part (blocks (c, J), J > c, if the rank owns row c) and transposed part (blocks (I, c) of the
//                owned rows I < c) of the rank's partial product, straight into the all-reduce buffer (no qa / qb, no combine launch);
//                per workgroup the partial dot products p . out and W~^T out -- LINEAR in the partial product, so they ride through
//                the same all-reduce and arrive summed over the ranks
//   all-reduce   q (ld doubles) | the partial dots (16 doubles per product workgroup)
//   k_dcg_upd    every workgroup sums the partial dots (fixed order: identical on every rank), forms alpha, c = c - alpha W~^T q,
//                mu = E^-1 c, |r - alpha q|^2 over the WHOLE vector (96 KB from L2, redundantly: cheaper than a second launch boundary),
//                beta, and updates ITS slice of x, r, p.  The stopping test is taken by every workgroup from the same numbers.
// r is double-buffered by launch parity (a workgroup's redundant norm reads what another one's slice update writes), the scalar state
// {r.z, c} likewise.  The done flag holds the NUMBER of the first launch with nothing left to do (a launch compares it with its own
// number: it must never act on a flag its own first workgroup raises, DESIGN.md section 4 "LM loop").
// ---------------------------------------------------------------------------------------------------------------------
constexpr int CPW = 4;                 // cameras per workgroup of the product kernels (a wave each)
constexpr int LIN_STRIDE = 16;         // doubles per product workgroup in the iteration's all-reduce tail (9 used: p . out, W~^T out)
constexpr int LIN8_STRIDE = 64;        // ... in the set-up's (the workgroup's part of E = W~^T (S~ W~))
constexpr int UPD_MAXWG = 32;
enum { ST_RZ = 0, ST_C = 1, ST_LEN = 16 };     // scalar state by launch parity: r.z, c = W~^T r

__device__ __forceinline__ bool dcg_done(const int* flags, int launch_no) { const int dn = flags[PF_DONE]; return dn != 0 && dn <= launch_no; }

// block (row I, column J > I) of the owned list: 36 values, row-major
template <typename FT> __device__ __forceinline__ void load_block36(const FT* __restrict__ B, double (&b)[36]) {
#pragma unroll
    for (int e = 0; e < 36; ++e) b[e] = (double)B[e];
}

template <typename FT, typename FF>
__global__ __launch_bounds__(256) void k_dcg_prod(int d, int ld, int ncam, int row0, int row1, int rank, const FT* __restrict__ owned,
                                                  const FF* __restrict__ focal_row, const double* __restrict__ p, const double* __restrict__ W,
                                                  double* __restrict__ out, const int* __restrict__ flags, int launch_no) {
    __shared__ double sh[4][9];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int fo = d - 1;
    double* lin = out + ld + (size_t)blockIdx.x * LIN_STRIDE;
    if (dcg_done(flags, launch_no)) {
        // a surplus launch of the batch: the collective behind it is issued regardless (the host cannot know), and an in-place all-reduce of the stale,
        // already summed buffer would multiply it by the world size once per surplus launch (inf / NaN after a long forced batch at large world).
        // Hand it zeros: this workgroup's part of the buffer.
        if (blockIdx.x == gridDim.x - 1) { if (tid == 0) out[fo] = 0.0; }
        else { const int c = blockIdx.x * CPW + w; if (c < ncam && lane < 6) out[6 * c + lane] = 0.0; }
        if (tid < 9) lin[tid] = 0.0;
        return;
    }
    if (blockIdx.x == gridDim.x - 1) {
        // the focal entry: known on every rank from exchange (A) -- rank 0 contributes it
        double fd = 0.0;
        if (rank == 0) { for (int i = tid; i < fo; i += 256) fd = fma((double)focal_row[i], p[i], fd); }
        fd = wave_allsum(fd);
        if (lane == 0) sh[w][0] = fd;
        __syncthreads();
        const double v = rank == 0 ? p[fo] + (sh[0][0] + sh[1][0] + sh[2][0] + sh[3][0]) : 0.0;
        if (tid == 0) out[fo] = v;
        if (tid < 9) lin[tid] = tid == 0 ? p[fo] * v : (W ? W[(size_t)(tid - 1) * ld + fo] * v : 0.0);
        return;
    }
    const int c = blockIdx.x * CPW + w;
    double mine = 0.0;
    if (c < ncam) {
        const long long base = first_of_row(row0, ncam);
        double acc[6] = { 0, 0, 0, 0, 0, 0 };
        if (c >= row0 && c < row1) {
            const long long rowpos = first_of_row(c, ncam) - base;
            for (int J = c + 1 + lane; J < ncam; J += 64) {
                double b[36], pj[6];
                load_block36(owned + (size_t)(rowpos + (J - c - 1)) * 36, b);
#pragma unroll
                for (int e = 0; e < 6; ++e) pj[e] = p[6 * J + e];
#pragma unroll
                for (int r = 0; r < 6; ++r)
#pragma unroll
                    for (int e = 0; e < 6; ++e) acc[r] = fma(b[6 * r + e], pj[e], acc[r]);
            }
        }
        if (c > row0) {
            const int iend = row1 < c ? row1 : c;
            for (int I = row0 + lane; I < iend; I += 64) {
                double b[36], pi[6];
                load_block36(owned + (size_t)(first_of_row(I, ncam) - base + (c - I - 1)) * 36, b);
#pragma unroll
                for (int r = 0; r < 6; ++r) pi[r] = p[6 * I + r];
#pragma unroll
                for (int r = 0; r < 6; ++r)
#pragma unroll
                    for (int e = 0; e < 6; ++e) acc[e] = fma(b[6 * r + e], pi[r], acc[e]);
            }
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) acc[k] = wave_allsum(acc[k]);       // (every lane holds all six)
        if (rank == 0) {
            const double pf = p[fo];
#pragma unroll
            for (int k = 0; k < 6; ++k) acc[k] += p[6 * c + k] + (double)focal_row[6 * c + k] * pf;
        }
        if (lane < 6) {
            double v = acc[0];
#pragma unroll
            for (int k = 1; k < 6; ++k) v = lane == k ? acc[k] : v;
            out[6 * c + lane] = v;
        }
        if (lane < 9) {           // lane 0: p . out of this camera; lanes 1 .. 8: W~_k . out
            const double* u = lane == 0 ? p : (W ? W + (size_t)(lane - 1) * ld : nullptr);
            if (u) {
#pragma unroll
                for (int k = 0; k < 6; ++k) mine = fma(u[6 * c + k], acc[k], mine);
            }
        }
    }
    if (lane < 9) sh[w][lane] = mine;
    __syncthreads();
    if (tid < 9) lin[tid] = sh[0][tid] + sh[1][tid] + sh[2][tid] + sh[3][tid];
}

// the same product for the EIGHT gauge vectors at once (set-up of a solve: S~ W~, read from the blocks once), with the workgroup's part of
// E[k][l] = sum_i W~_k[i] (S~ W~_l)[i] behind it
template <typename FT, typename FF>
__global__ __launch_bounds__(256) void k_dcg_prod8(int d, int ld, int ncam, int row0, int row1, int rank, const FT* __restrict__ owned,
                                                   const FF* __restrict__ focal_row, const double* __restrict__ W, double* __restrict__ out) {
    __shared__ double sh[4][64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int fo = d - 1;
    double* lin = out + (size_t)NW * ld + (size_t)blockIdx.x * LIN8_STRIDE;
    if (blockIdx.x == gridDim.x - 1) {
        double fd[NW];
#pragma unroll
        for (int l = 0; l < NW; ++l) fd[l] = 0.0;
        if (rank == 0) {
            for (int i = tid; i < fo; i += 256) {
                const double f = (double)focal_row[i];
#pragma unroll
                for (int l = 0; l < NW; ++l) fd[l] = fma(f, W[(size_t)l * ld + i], fd[l]);
            }
        }
#pragma unroll
        for (int l = 0; l < NW; ++l) fd[l] = wave_allsum(fd[l]);
        if (lane == 0) {
#pragma unroll
            for (int l = 0; l < NW; ++l) sh[w][l] = fd[l];
        }
        __syncthreads();
        if (tid < 64) {
            const int k = tid >> 3, l = tid & 7;
            const double v = rank == 0 ? W[(size_t)l * ld + fo] + (sh[0][l] + sh[1][l] + sh[2][l] + sh[3][l]) : 0.0;
            if (k == 0) out[(size_t)l * ld + fo] = v;
            lin[tid] = W[(size_t)k * ld + fo] * v;
        }
        return;
    }
    const int c = blockIdx.x * CPW + w;
    double e_kl = 0.0;
    if (c < ncam) {
        const long long base = first_of_row(row0, ncam);
        double acc[NW][6];
#pragma unroll
        for (int l = 0; l < NW; ++l)
#pragma unroll
            for (int k = 0; k < 6; ++k) acc[l][k] = 0.0;
        if (c >= row0 && c < row1) {
            const long long rowpos = first_of_row(c, ncam) - base;
            for (int J = c + 1 + lane; J < ncam; J += 64) {
                double b[36];
                load_block36(owned + (size_t)(rowpos + (J - c - 1)) * 36, b);
#pragma unroll
                for (int l = 0; l < NW; ++l) {
                    double pj[6];
#pragma unroll
                    for (int e = 0; e < 6; ++e) pj[e] = W[(size_t)l * ld + 6 * J + e];
#pragma unroll
                    for (int r = 0; r < 6; ++r)
#pragma unroll
                        for (int e = 0; e < 6; ++e) acc[l][r] = fma(b[6 * r + e], pj[e], acc[l][r]);
                }
            }
        }
        if (c > row0) {
            const int iend = row1 < c ? row1 : c;
            for (int I = row0 + lane; I < iend; I += 64) {
                double b[36];
                load_block36(owned + (size_t)(first_of_row(I, ncam) - base + (c - I - 1)) * 36, b);
#pragma unroll
                for (int l = 0; l < NW; ++l) {
                    double pi[6];
#pragma unroll
                    for (int r = 0; r < 6; ++r) pi[r] = W[(size_t)l * ld + 6 * I + r];
#pragma unroll
                    for (int r = 0; r < 6; ++r)
#pragma unroll
                        for (int e = 0; e < 6; ++e) acc[l][e] = fma(b[6 * r + e], pi[r], acc[l][e]);
                }
            }
        }
        const int kk = lane >> 3, ll = lane & 7;
        double wk[6], mine_l[6];
#pragma unroll
        for (int e = 0; e < 6; ++e) { wk[e] = W[(size_t)kk * ld + 6 * c + e]; mine_l[e] = 0.0; }
#pragma unroll
        for (int l = 0; l < NW; ++l) {
            const double wf = rank == 0 ? W[(size_t)l * ld + fo] : 0.0;
#pragma unroll
            for (int e = 0; e < 6; ++e) {
                double v = wave_allsum(acc[l][e]);
                if (rank == 0) v += W[(size_t)l * ld + 6 * c + e] + (double)focal_row[6 * c + e] * wf;
                if (lane == l * 8 + e) out[(size_t)l * ld + 6 * c + e] = v;
                mine_l[e] = ll == l ? v : mine_l[e];
            }
        }
#pragma unroll
        for (int e = 0; e < 6; ++e) e_kl = fma(wk[e], mine_l[e], e_kl);
    }
    sh[w][lane] = e_kl;
    __syncthreads();
    if (tid < 64) lin[tid] = sh[0][tid] + sh[1][tid] + sh[2][tid] + sh[3][tid];
}

__device__ __forceinline__ double block_sum_all256(double v, double* sh4) {      // sum over a workgroup of 256 threads, in every thread
    v = wave_allsum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh4[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh4[0] + sh4[1] + sh4[2] + sh4[3];
}

// start of a solve behind the all-reduce of [S~ W~ | E partials]: E^-1, c_0 = W~^T b~ and |b~|^2 (formed by every workgroup itself: 9 d
// reads from L2), mu_0, then the workgroup's slice of x = 0, r = b~, p = z = r + W~ mu_0; thresholds, flags, scalar state (workgroup 0)
__global__ __launch_bounds__(256) void k_dcg_start(int d, int ld, int np, const double* __restrict__ buf, const double* __restrict__ bt,
                                                   const double* __restrict__ W, double* __restrict__ x, double* __restrict__ r0, double* __restrict__ p,
                                                   double* __restrict__ state, double* __restrict__ scal, int* __restrict__ flags, int anchor, double cap) {
    __shared__ double tot[NW * NW], sa[NW * NW], sb[NW * NW], s_einv[NW * NW], sh4[4];
    const int tid = threadIdx.x;
    double mu[NW], c0[NW];
#pragma unroll
    for (int k = 0; k < NW; ++k) { mu[k] = 0.0; c0[k] = 0.0; }
    double rr = 0.0;
    if (W) {
        const double* lin = buf + (size_t)NW * ld;
        {   // E = the partials of the product workgroups summed: entry tid % 64 over the rows tid / 64, tid / 64 + 4, ... (eight loads in flight per
            // thread: one load per pass was np dependent L2 round trips, 74 us at 1000 cameras), the four quarters added in a fixed order
            const int e = tid & 63, part = tid >> 6;
            double acc = 0.0;
            for (int i0 = part; i0 < np; i0 += 32) {
                double t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int i = i0 + 4 * u; t[u] = lin[(size_t)(i < np ? i : part) * LIN8_STRIDE + e]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += (i0 + 4 * u < np) ? t[u] : 0.0;
            }
            sa[tid & 63] = 0.0;      // (sa / sb are scratch of the inversion below; sa doubles as the staging of the quarters: 4 x 64 would not fit, so quarter by quarter)
            __syncthreads();
            for (int q = 0; q < 4; ++q) { if (part == q) sa[e] += acc; __syncthreads(); }
            if (tid < NW * NW) tot[tid] = sa[tid];
        }
        __syncthreads();
        if (tid < 64) s_einv[tid] = coarse_invert_wave(tot, sa, sb);
        double v[NW + 1];
#pragma unroll
        for (int k = 0; k <= NW; ++k) v[k] = 0.0;
        for (int i0 = tid; i0 < d; i0 += 1024) {          // four entries per pass: 36 loads in flight
            double b4[4], w4[4][NW];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + 256 * u, ic = i < d ? i : 0;
                b4[u] = bt[ic];
#pragma unroll
                for (int k = 0; k < NW; ++k) w4[u][k] = W[(size_t)k * ld + ic];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const double b = (i0 + 256 * u < d) ? b4[u] : 0.0;
                v[NW] = fma(b, b, v[NW]);
#pragma unroll
                for (int k = 0; k < NW; ++k) v[k] = fma(w4[u][k], b, v[k]);
            }
        }
        {   // nine block sums behind ONE pair of barriers
            __shared__ double s9[4][NW + 1];
#pragma unroll
            for (int k = 0; k <= NW; ++k) v[k] = wave_allsum(v[k]);
            __syncthreads();
            if ((tid & 63) == 0) {
#pragma unroll
                for (int k = 0; k <= NW; ++k) s9[tid >> 6][k] = v[k];
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < NW; ++k) c0[k] = s9[0][k] + s9[1][k] + s9[2][k] + s9[3][k];
            rr = s9[0][NW] + s9[1][NW] + s9[2][NW] + s9[3][NW];
        }
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            double m = 0.0;
#pragma unroll
            for (int l = 0; l < NW; ++l) m = fma(s_einv[k * NW + l], c0[l], m);
            mu[k] = m;
        }
    } else {
        double v = 0.0;
        for (int i = tid; i < d; i += 256) { const double b = bt[i]; v = fma(b, b, v); }
        rr = block_sum_all256(v, sh4);
    }
    double rz = rr;
#pragma unroll
    for (int k = 0; k < NW; ++k) rz = fma(c0[k], mu[k], rz);
    const int per = (d + gridDim.x - 1) / gridDim.x, i0 = blockIdx.x * per, i1 = min(d, i0 + per);
    for (int i = i0 + tid; i < i1; i += 256) {
        const double b = bt[i];
        double z = b;
        if (W) {
#pragma unroll
            for (int k = 0; k < NW; ++k) z = fma(W[(size_t)k * ld + i], mu[k], z);
        }
        x[i] = 0.0; r0[i] = b; p[i] = z;
    }
    if (blockIdx.x == 0) {
        if (W && tid < NW * NW) scal[DS_EINV + tid] = s_einv[tid];
        if (tid < NW) state[ST_C + tid] = c0[tid];
        if (tid == 0) {
            double base = rr;
            if (anchor == 1) scal[DS_RRF] = rr;
            else if (anchor == 2) base = fmin(fmax(rr, scal[DS_RRF]), cap * rr);
            scal[DS_RR0] = base;
            state[ST_RZ] = rz;
            flags[PF_DONE] = (rr == 0.0) ? 1 : 0; flags[PF_ITERS] = 0; flags[PF_XBUF] = 0;
        }
    }
}

// one CG iteration behind the all-reduce of [q | partial dots]
__global__ __launch_bounds__(256) void k_dcg_upd(int d, int ld, int np, const double* __restrict__ qbuf, const double* __restrict__ W,
                                                 double* __restrict__ x, const double* __restrict__ r_in, double* __restrict__ r_out, double* __restrict__ p,
                                                 double* __restrict__ state, const double* __restrict__ scal, int* __restrict__ flags, double tol2, int* info,
                                                 int launch_no) {
    if (dcg_done(flags, launch_no)) return;
    __shared__ double tot[12], s_einv[NW * NW], sh4[4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const double* q = qbuf;
    const double* lin = qbuf + ld;
    {   // the nine partial dots summed over the product workgroups: wave w takes values w, w + 4, w + 8, lane-strided (fixed order)
        double m[3] = { 0.0, 0.0, 0.0 };
        for (int i = lane; i < np; i += 64) {
#pragma unroll
            for (int j = 0; j < 3; ++j) if (w + 4 * j < 9) m[j] += lin[(size_t)i * LIN_STRIDE + w + 4 * j];
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) m[j] = wave_allsum(m[j]);
        if (lane == 0) {
#pragma unroll
            for (int j = 0; j < 3; ++j) if (w + 4 * j < 9) tot[w + 4 * j] = m[j];
        }
    }
    if (W && tid < NW * NW) s_einv[tid] = scal[DS_EINV + tid];
    const double* st_in = state + (size_t)((launch_no - 1) & 1) * ST_LEN;
    double* st_out = state + (size_t)(launch_no & 1) * ST_LEN;
    __syncthreads();
    const double pq = tot[0], rz = st_in[ST_RZ];
    const bool broke = !(pq > 0.0);
    const double alpha = broke ? 0.0 : rz / pq;
    double rr = 0.0;
    for (int i0 = tid; i0 < d; i0 += 2048) {              // eight entries per pass, sixteen loads in flight (one per pass: d / 256 dependent round trips)
        double q8[8], r8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int i = i0 + 256 * u, ic = i < d ? i : 0; q8[u] = q[ic]; r8[u] = r_in[ic]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) { const double rn = fma(-alpha, q8[u], r8[u]); rr = (i0 + 256 * u < d) ? fma(rn, rn, rr) : rr; }
    }
    rr = block_sum_all256(rr, sh4);
    double c[NW], mu[NW];
    double rzn = rr;
#pragma unroll
    for (int k = 0; k < NW; ++k) { c[k] = 0.0; mu[k] = 0.0; }
    if (W) {
#pragma unroll
        for (int k = 0; k < NW; ++k) c[k] = fma(-alpha, tot[1 + k], st_in[ST_C + k]);
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            double m = 0.0;
#pragma unroll
            for (int l = 0; l < NW; ++l) m = fma(s_einv[k * NW + l], c[l], m);
            mu[k] = m;
        }
#pragma unroll
        for (int k = 0; k < NW; ++k) rzn = fma(c[k], mu[k], rzn);
    }
    const bool done = broke || !(rr == rr) || rr <= tol2 * scal[DS_RR0];
    const double beta = rzn / rz;
    const int per = (d + gridDim.x - 1) / gridDim.x, i0 = blockIdx.x * per, i1 = min(d, i0 + per);
    for (int i = i0 + tid; i < i1; i += 256) {
        const double rn = fma(-alpha, q[i], r_in[i]);
        const double pi = p[i];
        r_out[i] = rn;
        x[i] = fma(alpha, pi, x[i]);
        if (!done) {
            double z = rn;
            if (W) {
#pragma unroll
                for (int k = 0; k < NW; ++k) z = fma(W[(size_t)k * ld + i], mu[k], z);
            }
            p[i] = fma(beta, pi, z);
        }
    }
    if (blockIdx.x == 0) {
        if (tid < NW) st_out[ST_C + tid] = c[tid];
        if (tid == 0) {
            st_out[ST_RZ] = rzn;
            flags[PF_ITERS] = launch_no;
            if (done) { flags[PF_DONE] = launch_no + 1; if (broke || !(rr == rr)) atomicCAS(info, 0, d + 1); }
        }
    }
}
}  // namespace

void dcg_partition(int ncam, int world, std::vector<int>* rows, long long* chunk_blocks) {
    const long long total = (long long)ncam * (ncam - 1) / 2;
    rows->assign((size_t)world + 1, ncam);
    (*rows)[0] = 0;
    long long acc = 0;
    int r = 1;
    for (int I = 0; I < ncam && r < world; ++I) {
        acc += ncam - 1 - I;
        // row I closes rank r - 1 once its share is reached (the remaining ranks get the remaining rows)
        while (r < world && acc * world >= total * r) { (*rows)[(size_t)r] = I + 1; ++r; }
    }
    long long big = 0;
    for (int k = 0; k < world; ++k) {
        const long long a = (long long)(*rows)[(size_t)k] * ncam - (long long)(*rows)[(size_t)k] * ((*rows)[(size_t)k] + 1) / 2;
        const long long b = (long long)(*rows)[(size_t)k + 1] * ncam - (long long)(*rows)[(size_t)k + 1] * ((*rows)[(size_t)k + 1] + 1) / 2;
        big = std::max(big, b - a);
    }
    *chunk_blocks = std::max<long long>(big, 1);
}

int dcg_create(DistCg* g, int d, int ld, int ncam, int rank, int world, DeviceArena* arena) {
    g->d = d; g->ld = ld; g->ncam = ncam; g->rank = rank; g->world = world;
    dcg_partition(ncam, world, &g->rows, &g->chunk_blocks);
    g->row0 = g->rows[(size_t)rank]; g->row1 = g->rows[(size_t)rank + 1];
    std::vector<int> shift((size_t)std::max(ncam, 1), 0);
    for (int k = 0; k < world; ++k) {
        const long long first = (long long)g->rows[(size_t)k] * ncam - (long long)g->rows[(size_t)k] * (g->rows[(size_t)k] + 1) / 2;
        for (int I = g->rows[(size_t)k]; I < g->rows[(size_t)k + 1]; ++I) shift[(size_t)I] = (int)(k * g->chunk_blocks - first);
    }
    // all or nothing: a workspace that fails half way leaves nothing behind in *g (the arena keeps the bytes until the problem goes)
    int* d_row_shift = arena->alloc_n<int>(shift.size());
    double* qa = arena->alloc_n<double>((size_t)ld); double* qb = arena->alloc_n<double>((size_t)ld);
    // all-reduce buffer: [8][ld] S~ W~ (set-up) | [ld] the iteration's product | the partial dots of the product workgroups behind either
    g->np = (ncam + CPW - 1) / CPW + 1;
    const size_t qred_len = (size_t)9 * ld + (size_t)g->np * LIN8_STRIDE;
    double* qred = arena->alloc_n<double>(qred_len);
    double* r = arena->alloc_n<double>((size_t)2 * ld); double* pp = arena->alloc_n<double>((size_t)ld);
    double* scal = arena->alloc_n<double>(128);
    double* state = arena->alloc_n<double>((size_t)2 * ST_LEN);
    if (!d_row_shift || !qa || !qb || !qred || !r || !pp || !scal || !state) return -1;
    if (hipMemcpy(d_row_shift, shift.data(), sizeof(int) * shift.size(), hipMemcpyHostToDevice) != hipSuccess) return -1;
    // the collectives carry ld doubles per vector, the kernels write d of them: the padding is zero once and for all
    if (hipMemset(qred, 0, sizeof(double) * qred_len) != hipSuccess) return -1;
    if (hipMemset(state, 0, sizeof(double) * 2 * ST_LEN) != hipSuccess) return -1;
    g->d_row_shift = d_row_shift; g->qa = qa; g->qb = qb; g->qred = qred; g->r = r; g->p = pp; g->scal = scal; g->state = state;
    g->AW = g->qred;                  // slots 0..7 of the all-reduce buffer ARE S~ W~ after the setup; slot 8 carries the iterations' products
    g->ready = true;
    return 0;
}

static void launch_product(hipStream_t s, const DistCg* g, const DcgSolveArgs& a, const double* p, double* out, const int* flags) {
    if (a.implicit) { launch_implicit_product(s, *a.implicit, p, out, flags); return; }
    const int nA = g->row1 - g->row0, nB = std::max(0, g->ncam - g->row0 - 1);
    if (nA + nB > 0) {
        if (a.owned_f32) hipLaunchKernelGGL(k_dcg_spmv<float>, dim3(nA + nB), dim3(256), 0, s, g->ncam, g->row0, g->row1, static_cast<const float*>(a.owned), p, g->qa, g->qb, flags);
        else hipLaunchKernelGGL(k_dcg_spmv<double>, dim3(nA + nB), dim3(256), 0, s, g->ncam, g->row0, g->row1, static_cast<const double*>(a.owned), p, g->qa, g->qb, flags);
    }
    if (a.focal_row32) hipLaunchKernelGGL(k_dcg_comb<float>, dim3(1), dim3(STEP_T), 0, s, g->d, g->ncam, g->row0, g->row1, g->rank, g->qa, g->qb, p, a.focal_row32, out, flags);
    else hipLaunchKernelGGL(k_dcg_comb<double>, dim3(1), dim3(STEP_T), 0, s, g->d, g->ncam, g->row0, g->row1, g->rank, g->qa, g->qb, p, a.focal_row, out, flags);
}

static int upd_grid(int d) { return std::max(1, std::min(UPD_MAXWG, (d + 255) / 256)); }

int dcg_begin(hipStream_t s, DistCg* g, const DcgSolveArgs& a, dcg_allreduce_fn ar, void* ctx) {
    g->launched = 0;
    if (a.implicit) {
        // the implicit product forms whole vectors in launches of its own: the single-workgroup update kernels of round 4
        if (a.W) {
            for (int k = 0; k < NW; ++k) launch_product(s, g, a, a.W + (size_t)k * g->ld, g->qred + (size_t)k * g->ld, nullptr);
            if (ar) { const int rc = ar(ctx, g->qred, (long long)NW * g->ld, s); if (rc) return rc; }
        }
        hipLaunchKernelGGL(k_dcg_init, dim3(1), dim3(STEP_T), 0, s, g->d, g->ld, a.bt, a.W, g->AW, g->x, g->r, g->p, g->scal, a.flags, a.anchor, a.cap);
        return 0;
    }
    if (a.W) {
        const dim3 grid(g->np);
#define SFMBA_DCG_PROD8(FT, FF, fr) hipLaunchKernelGGL((k_dcg_prod8<FT, FF>), grid, dim3(256), 0, s, g->d, g->ld, g->ncam, g->row0, g->row1, g->rank, \
                                                       static_cast<const FT*>(a.owned), fr, a.W, g->qred)
        if (a.owned_f32) { if (a.focal_row32) SFMBA_DCG_PROD8(float, float, a.focal_row32); else SFMBA_DCG_PROD8(float, double, a.focal_row); }
        else { if (a.focal_row32) SFMBA_DCG_PROD8(double, float, a.focal_row32); else SFMBA_DCG_PROD8(double, double, a.focal_row); }
#undef SFMBA_DCG_PROD8
        if (ar) { const int rc = ar(ctx, g->qred, (long long)NW * g->ld + (long long)g->np * LIN8_STRIDE, s); if (rc) return rc; }
    }
    hipLaunchKernelGGL(k_dcg_start, dim3(upd_grid(g->d)), dim3(256), 0, s, g->d, g->ld, g->np, g->qred, a.bt, a.W, g->x, g->r, g->p, g->state, g->scal,
                       a.flags, a.anchor, a.cap);
    return 0;
}

int dcg_iterate(hipStream_t s, DistCg* g, const DcgSolveArgs& a, int n, dcg_allreduce_fn ar, void* ctx) {
    double* q = g->qred + (size_t)NW * g->ld;
    if (a.implicit) {
        for (int it = 0; it < n; ++it) {
            launch_product(s, g, a, g->p, q, a.flags);
            if (ar) { const int rc = ar(ctx, q, (long long)g->ld, s); if (rc) return rc; }
            hipLaunchKernelGGL(k_dcg_step, dim3(1), dim3(STEP_T), 0, s, g->d, g->ld, q, a.W, g->x, g->r, g->p, g->scal, a.flags, a.tol * a.tol, a.info);
        }
        return 0;
    }
    const dim3 grid(g->np);
    for (int it = 0; it < n; ++it) {
        const int no = ++g->launched;
#define SFMBA_DCG_PROD(FT, FF, fr) hipLaunchKernelGGL((k_dcg_prod<FT, FF>), grid, dim3(256), 0, s, g->d, g->ld, g->ncam, g->row0, g->row1, g->rank, \
                                                      static_cast<const FT*>(a.owned), fr, g->p, a.W, q, a.flags, no)
        if (a.owned_f32) { if (a.focal_row32) SFMBA_DCG_PROD(float, float, a.focal_row32); else SFMBA_DCG_PROD(float, double, a.focal_row); }
        else { if (a.focal_row32) SFMBA_DCG_PROD(double, float, a.focal_row32); else SFMBA_DCG_PROD(double, double, a.focal_row); }
#undef SFMBA_DCG_PROD
        if (ar) { const int rc = ar(ctx, q, (long long)g->ld + (long long)g->np * LIN_STRIDE, s); if (rc) return rc; }
        double* r_in = g->r + (size_t)((no - 1) & 1) * g->ld;
        double* r_out = g->r + (size_t)(no & 1) * g->ld;
        hipLaunchKernelGGL(k_dcg_upd, dim3(upd_grid(g->d)), dim3(256), 0, s, g->d, g->ld, g->np, q, a.W, g->x, r_in, r_out, g->p, g->state, g->scal,
                           a.flags, a.tol * a.tol, a.info, no);
    }
    return 0;
}

}  // namespace sfmba
