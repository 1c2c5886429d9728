// dense_solver.hip -- the reduced camera system  S z = rhs  (dim 6*Nc+1) solved on the device.
//
// Replaces what DENSE_SCHUR hands to Eigen's LLT in the reference configuration
// (SfMToyLib/SfMBundleAdjustmentUtils.cpp:172, DenseSchurComplementSolver [Ceres-upstream]).
//
// Storage: the kernels in ba_kernels.hip accumulate the UPPER triangle of the row-major matrix,
// which is byte-for-byte the LOWER triangle of a column-major matrix A(i,j) = S[j*ld + i], i >= j.
// The matrix is padded to a multiple of CHOL_NB with an identity diagonal, and the right-hand
// side is stored as one extra ROW of A (row index d): the blocked factorisation then produces
// L(d, 0:d) = (L^-1 rhs)^T, i.e. the forward substitution comes for free with the panel updates.
//
// Cholesky = right-looking blocked LL^T, NB = 64: per block column one panel kernel (every block
// factors the 64x64 diagonal tile redundantly in LDS, then solves its own tile of the panel) and
// one trailing-update kernel; back substitution is one launch per block column (eager updates).
#include "dense_solver.h"
#include <math.h>
#include <stdio.h>

namespace sfmba {

#define NB CHOL_NB
#define AT(i, j) A[(size_t)(i) + (size_t)(j) * ld]

// ------------------------------------------------------------------------------------------
// panel: factor A_kk, M_k = L_kk^-T, L_ik = A_ik L_kk^-T
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_chol_panel(double* __restrict__ A, int ld, int k, int d,
                                                    double* __restrict__ minv, int* __restrict__ info) {
    __shared__ double Lkk[NB][NB + 1];
    __shared__ double Aik[NB][NB + 1];
    const int tid = threadIdx.x;
    const int kb = k * NB;
    for (int idx = tid; idx < NB * NB; idx += 256) {
        const int r = idx % NB, c = idx / NB;
        Lkk[r][c] = (r >= c) ? AT(kb + r, kb + c) : 0.0;
    }
    __syncthreads();
    for (int j = 0; j < NB; ++j) {
        double dj = Lkk[j][j];
        if (!(dj > 0.0) || !(dj <= 1.7e308)) {
            // not positive definite: flag it for real columns; augmented/padded columns are forced to 1
            if (kb + j < d && tid == 0 && blockIdx.x == 0) atomicCAS(info, 0, kb + j + 1);
            dj = 1.0;
        }
        const double dinv = 1.0 / sqrt(dj);
        __syncthreads();
        if (tid >= j && tid < NB) Lkk[tid][j] = (tid == j) ? dj * dinv : Lkk[tid][j] * dinv;
        __syncthreads();
        const int m = NB - j - 1;
        for (int idx = tid; idx < m * m; idx += 256) {
            const int r = j + 1 + idx % m, c = j + 1 + idx / m;
            if (r >= c) Lkk[r][c] -= Lkk[r][j] * Lkk[c][j];
        }
        __syncthreads();
    }
    const bool diag_block = (blockIdx.x == 0);
    const int ib = (k + blockIdx.x) * NB;
    if (diag_block) {
        for (int idx = tid; idx < NB * NB; idx += 256) {
            const int r = idx % NB, c = idx / NB;
            if (r >= c) AT(kb + r, kb + c) = Lkk[r][c];
            Aik[r][c] = (r == c) ? 1.0 : 0.0;          // identity -> becomes L_kk^-T
        }
    } else {
        for (int idx = tid; idx < NB * NB; idx += 256) {
            const int r = idx % NB, c = idx / NB;
            Aik[r][c] = AT(ib + r, kb + c);
        }
    }
    __syncthreads();
    // X L_kk^T = Aik  (column sweep)
    for (int j = 0; j < NB; ++j) {
        const double dinv = 1.0 / Lkk[j][j];
        if (tid < NB) Aik[tid][j] *= dinv;
        __syncthreads();
        const int m = NB - j - 1;
        for (int idx = tid; idx < NB * m; idx += 256) {
            const int r = idx % NB, c = j + 1 + idx / NB;
            Aik[r][c] -= Aik[r][j] * Lkk[c][j];
        }
        __syncthreads();
    }
    if (diag_block) {
        double* M = minv + (size_t)k * NB * NB;     // M[r + c*NB] = (L_kk^-T)(r,c)
        for (int idx = tid; idx < NB * NB; idx += 256) M[idx] = Aik[idx % NB][idx / NB];
    } else {
        for (int idx = tid; idx < NB * NB; idx += 256) {
            const int r = idx % NB, c = idx / NB;
            AT(ib + r, kb + c) = Aik[r][c];
        }
    }
}

// ------------------------------------------------------------------------------------------
// trailing update: A_ij -= L_ik L_jk^T  for k < j <= i
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_chol_update(double* __restrict__ A, int ld, int k) {
    __shared__ double Li[NB][NB + 1];
    __shared__ double Lj[NB][NB + 1];
    const int tid = threadIdx.x;
    const int t = blockIdx.x;
    int ii = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((ii + 1) * (ii + 2) / 2 <= t) ++ii;
    while (ii * (ii + 1) / 2 > t) --ii;
    const int jj = t - ii * (ii + 1) / 2;
    const int ib = (k + 1 + ii) * NB, jb = (k + 1 + jj) * NB, kb = k * NB;
    for (int idx = tid; idx < NB * NB; idx += 256) {
        const int r = idx % NB, c = idx / NB;
        Li[r][c] = AT(ib + r, kb + c);
        Lj[r][c] = AT(jb + r, kb + c);
    }
    __syncthreads();
    const int tr = tid % 16, tc = tid / 16;
    double acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
    for (int kk = 0; kk < NB; ++kk) {
        double li[4], lj[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) { li[a] = Li[tr + 16 * a][kk]; lj[a] = Lj[tc + 16 * a][kk]; }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] += li[a] * lj[b];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int r = tr + 16 * a, c = tc + 16 * b;
            if (ib + r >= jb + c) AT(ib + r, jb + c) -= acc[a][b];
        }
}

// rhs -> augmented row d (row-major column d); padded diagonal is already 1
__global__ void k_augment(double* __restrict__ A, int ld, int d, const double* __restrict__ rhs) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < d) AT(d, c) = rhs[c];
}

// y = L(d, 0:d) (forward-substituted rhs), zero in the padding
__global__ void k_extract_y(const double* __restrict__ A, int ld, int d, double* __restrict__ y) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < ld) y[c] = c < d ? AT(d, c) : 0.0;
}

// back substitution step for block column k:  x_k = M_k y_k ;  y_i -= L_ki^T x_k  (i < k)
__global__ __launch_bounds__(256) void k_chol_backstep(const double* __restrict__ A, int ld, int k,
                                                       const double* __restrict__ minv, double* __restrict__ y,
                                                       double* __restrict__ x, int d) {
    __shared__ double xk[NB];
    __shared__ double part[4][NB];
    const int tid = threadIdx.x;
    const int kb = k * NB;
    const double* M = minv + (size_t)k * NB * NB;
    // x_k[r] = sum_c M(r,c) y_k[c]   (M upper triangular), 4 partial sums per row
    {
        const int r = tid % NB, seg = tid / NB;
        double s = 0.0;
        for (int c = seg * 16; c < seg * 16 + 16; ++c) s += M[r + c * NB] * y[kb + c];
        part[seg][r] = s;
    }
    __syncthreads();
    if (tid < NB) xk[tid] = part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid];
    __syncthreads();
    const int i = blockIdx.x;            // 0..k ; block k stores x_k
    if (i == k) {
        if (tid < NB && kb + tid < d) x[kb + tid] = xk[tid];
        return;
    }
    const int ib = i * NB;
    {
        const int c = tid % NB, seg = tid / NB;
        double s = 0.0;
        for (int r = seg * 16; r < seg * 16 + 16; ++r) s += AT(kb + r, ib + c) * xk[r];
        part[seg][c] = s;
    }
    __syncthreads();
    if (tid < NB) y[ib + tid] -= part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid];
}

void dense_cholesky_solve(hipStream_t s, DenseSolver* ws, double* S, double* rhs, int* info_dev, Profiler* prof) {
    const int ld = ws->ld, d = ws->d, nblk = ld / NB;
    { ProfScope ps(prof, KID_CHOL_AUGMENT, s);
      hipLaunchKernelGGL(k_augment, dim3((d + 255) / 256), dim3(256), 0, s, S, ld, d, rhs); }
    for (int k = 0; k < nblk; ++k) {
        { ProfScope ps(prof, KID_CHOL_PANEL, s);
          hipLaunchKernelGGL(k_chol_panel, dim3(nblk - k), dim3(256), 0, s, S, ld, k, d, ws->minv, info_dev); }
        const int m = nblk - k - 1;
        if (m > 0) { ProfScope ps(prof, KID_CHOL_UPDATE, s);
          hipLaunchKernelGGL(k_chol_update, dim3(m * (m + 1) / 2), dim3(256), 0, s, S, ld, k); }
    }
    { ProfScope ps(prof, KID_CHOL_EXTRACT, s);
      hipLaunchKernelGGL(k_extract_y, dim3((ld + 255) / 256), dim3(256), 0, s, S, ld, d, ws->y); }
    for (int k = nblk - 1; k >= 0; --k) {
        ProfScope ps(prof, KID_CHOL_BACKSTEP, s);
        hipLaunchKernelGGL(k_chol_backstep, dim3(k + 1), dim3(256), 0, s, S, ld, k, ws->minv, ws->y, rhs, d);
    }
}

// ------------------------------------------------------------------------------------------
// Block-Jacobi preconditioned conjugate gradients on the dense reduced system
// ------------------------------------------------------------------------------------------
enum { SC_RZ = 0, SC_PQ = 1, SC_B2 = 2, SC_R2 = 3 };
enum { VX = 0, VR = 1, VZ = 2, VP = 3, VQ = 4 };

__global__ void k_mirror_full(const double* __restrict__ S, int ld, int d, double* __restrict__ F) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = blockIdx.y;
    if (c >= d) return;
    F[(size_t)r * ld + c] = c >= r ? S[(size_t)r * ld + c] : S[(size_t)c * ld + r];
}

// inverse of each 6x6 diagonal block (and the trailing 1x1): one thread per block
__global__ void k_block_inverse(const double* __restrict__ S, int ld, int d, double* __restrict__ binv, int* info) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    const int nb6 = (d - 1) / 6;
    if (b > nb6) return;
    if (b == nb6) {   // focal
        const double v = S[(size_t)(d - 1) * ld + d - 1];
        if (!(v > 0.0)) atomicCAS(info, 0, d);
        binv[(size_t)b * 36] = 1.0 / v;
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
        for (int c = 0; c < 6; ++c) {
            double v = 0.0;
            for (int t = (r > c ? r : c); t < 6; ++t) v += Li[t][r] * Li[t][c];
            binv[(size_t)b * 36 + r * 6 + c] = v;
        }
}

__device__ __forceinline__ double precond_apply(const double* binv, const double* r, int e, int d) {
    const int nb6 = (d - 1) / 6;
    const int b = e / 6;
    if (b >= nb6) return binv[(size_t)nb6 * 36] * r[e];
    const double* Bi = binv + (size_t)b * 36 + (e - 6 * b) * 6;
    const double* rb = r + 6 * b;
    return Bi[0] * rb[0] + Bi[1] * rb[1] + Bi[2] * rb[2] + Bi[3] * rb[3] + Bi[4] * rb[4] + Bi[5] * rb[5];
}

__device__ __forceinline__ double block_reduce_1024(double v, double* sm) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
    const int nw = blockDim.x >> 6;
    for (int i = 0; i < nw; ++i) s += sm[i];
    return s;
}

__global__ __launch_bounds__(1024) void k_pcg_init(int d, int ld, const double* __restrict__ b, double* __restrict__ vec,
                                                   const double* __restrict__ binv, double* scal, int* flags) {
    __shared__ double sm[16];
    double* x = vec + VX * ld; double* r = vec + VR * ld; double* z = vec + VZ * ld; double* p = vec + VP * ld;
    double b2 = 0.0;
    for (int e = threadIdx.x; e < d; e += blockDim.x) { x[e] = 0.0; r[e] = b[e]; b2 += b[e] * b[e]; }
    __syncthreads();
    double rz = 0.0;
    for (int e = threadIdx.x; e < d; e += blockDim.x) { const double ze = precond_apply(binv, r, e, d); z[e] = ze; p[e] = ze; rz += r[e] * ze; }
    rz = block_reduce_1024(rz, sm);
    b2 = block_reduce_1024(b2, sm);
    if (threadIdx.x == 0) { scal[SC_RZ] = rz; scal[SC_PQ] = 0.0; scal[SC_B2] = b2; scal[SC_R2] = b2; flags[0] = (b2 == 0.0); flags[1] = 0; }
}

// q = S p, one wave per row; pq += p^T q
__global__ __launch_bounds__(256) void k_pcg_matvec(int d, int ld, const double* __restrict__ F, double* __restrict__ vec,
                                                    double* scal, const int* flags) {
    if (flags[0]) return;
    const double* p = vec + VP * ld; double* q = vec + VQ * ld;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + w;
    double pq = 0.0;
    if (row < d) {
        const double* Fr = F + (size_t)row * ld;
        double s = 0.0;
        for (int c = lane; c < d; c += 64) s += Fr[c] * p[c];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        if (lane == 0) { q[row] = s; pq = p[row] * s; }
    }
    __shared__ double sm[4];
    if (lane == 0) sm[w] = pq;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&scal[SC_PQ], sm[0] + sm[1] + sm[2] + sm[3]);
}

__global__ __launch_bounds__(1024) void k_pcg_update(int d, int ld, double* __restrict__ vec, const double* __restrict__ binv,
                                                     double* scal, int* flags, double tol2) {
    __shared__ double sm[16];
    if (flags[0]) return;
    double* x = vec + VX * ld; double* r = vec + VR * ld; double* z = vec + VZ * ld; double* p = vec + VP * ld; double* q = vec + VQ * ld;
    const double rz = scal[SC_RZ];
    const double alpha = rz / scal[SC_PQ];
    double r2 = 0.0;
    for (int e = threadIdx.x; e < d; e += blockDim.x) { x[e] += alpha * p[e]; const double re = r[e] - alpha * q[e]; r[e] = re; r2 += re * re; }
    __syncthreads();
    double rzn = 0.0;
    for (int e = threadIdx.x; e < d; e += blockDim.x) { const double ze = precond_apply(binv, r, e, d); z[e] = ze; rzn += r[e] * ze; }
    rzn = block_reduce_1024(rzn, sm);
    r2 = block_reduce_1024(r2, sm);
    const double beta = rzn / rz;
    for (int e = threadIdx.x; e < d; e += blockDim.x) p[e] = z[e] + beta * p[e];
    if (threadIdx.x == 0) {
        scal[SC_RZ] = rzn; scal[SC_PQ] = 0.0; scal[SC_R2] = r2;
        flags[1] += 1;
        if (r2 <= tol2 * scal[SC_B2] || !(r2 == r2)) flags[0] = 1;
    }
}

__global__ void k_copy_vec(int d, const double* __restrict__ src, double* __restrict__ dst) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < d) dst[e] = src[e];
}

int dense_pcg_solve(hipStream_t s, DenseSolver* ws, double* S, double* rhs, double tol, int max_iters, int* info_dev, Profiler* prof) {
    const int ld = ws->ld, d = ws->d;
    if (!ws->Sfull) {
        if (hipMalloc(&ws->Sfull, sizeof(double) * (size_t)ld * ld) != hipSuccess) return -1;
    }
    if (max_iters <= 0) max_iters = 4 * d;
    { ProfScope ps(prof, KID_PCG_SETUP, s);
    hipLaunchKernelGGL(k_mirror_full, dim3((d + 255) / 256, d), dim3(256), 0, s, S, ld, d, ws->Sfull);
    hipLaunchKernelGGL(k_block_inverse, dim3(((d - 1) / 6 + 1 + 63) / 64), dim3(64), 0, s, S, ld, d, ws->binv, info_dev);
    hipLaunchKernelGGL(k_pcg_init, dim3(1), dim3(1024), 0, s, d, ld, rhs, ws->vec, ws->binv, ws->scal, ws->flags); }
    int it = 0;
    const int batch = 8;
    while (it < max_iters) {
        const int n = (max_iters - it) < batch ? (max_iters - it) : batch;
        for (int b = 0; b < n; ++b) {
            { ProfScope ps(prof, KID_PCG_MATVEC, s);
            hipLaunchKernelGGL(k_pcg_matvec, dim3((d + 3) / 4), dim3(256), 0, s, d, ld, ws->Sfull, ws->vec, ws->scal, ws->flags); }
            ProfScope ps2(prof, KID_PCG_UPDATE, s);
            hipLaunchKernelGGL(k_pcg_update, dim3(1), dim3(1024), 0, s, d, ld, ws->vec, ws->binv, ws->scal, ws->flags, tol * tol);
        }
        it += n;
        (void)hipMemcpyAsync(ws->h_flags, ws->flags, 2 * sizeof(int), hipMemcpyDeviceToHost, s);
        (void)hipStreamSynchronize(s);
        if (ws->h_flags[0]) { it = ws->h_flags[1]; break; }
    }
    hipLaunchKernelGGL(k_copy_vec, dim3((d + 255) / 256), dim3(256), 0, s, d, ws->vec + VX * ld, rhs);
    return it;
}

int dense_solver_create(DenseSolver* ws, int d, int ld) {
    ws->d = d; ws->ld = ld;
    const int nblk = ld / NB;
    if (hipMalloc(&ws->minv, sizeof(double) * (size_t)nblk * NB * NB) != hipSuccess) return -1;
    if (hipMalloc(&ws->y, sizeof(double) * ld) != hipSuccess) return -1;
    if (hipMalloc(&ws->vec, sizeof(double) * 6 * (size_t)ld) != hipSuccess) return -1;
    if (hipMalloc(&ws->binv, sizeof(double) * 36 * (size_t)(ld / 6 + 2)) != hipSuccess) return -1;
    if (hipMalloc(&ws->scal, sizeof(double) * 8) != hipSuccess) return -1;
    if (hipMalloc(&ws->flags, sizeof(int) * 4) != hipSuccess) return -1;
    if (hipHostMalloc(reinterpret_cast<void**>(&ws->h_flags), sizeof(int) * 4, hipHostMallocDefault) != hipSuccess) return -1;
    ws->Sfull = nullptr;
    return 0;
}

void dense_solver_destroy(DenseSolver* ws) {
    if (ws->minv) (void)hipFree(ws->minv);
    if (ws->y) (void)hipFree(ws->y);
    if (ws->vec) (void)hipFree(ws->vec);
    if (ws->binv) (void)hipFree(ws->binv);
    if (ws->scal) (void)hipFree(ws->scal);
    if (ws->flags) (void)hipFree(ws->flags);
    if (ws->h_flags) (void)hipHostFree(ws->h_flags);
    if (ws->Sfull) (void)hipFree(ws->Sfull);
    *ws = DenseSolver();
}

}  // namespace sfmba
