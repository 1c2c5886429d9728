// dense_cholesky.hip -- the exact solve of the reduced camera system  S z = rhs  (dim 6*Nc+1): blocked LL^T on the device.
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
// Three forms, chosen by size only (dense_cholesky_solve):
//   d <= 256 unknowns                 k_chol_small: factorisation and both substitutions in ONE launch
//   <= 40 block columns of 64         k_chol_step: one launch per block column (panel + trailing update fused, the 64x64
//                                     diagonal factor with its inverse in LDS: chol_tile.h), k_chol_backsolve: one launch
//   beyond                            k_chol_panel + k_chol_update per block column, k_chol_backstep per block column
#include "dense_solver.h"
#include "sfmba_device.h"
#include "chol_tile.h"
#include "coarse_inverse.h"
#include <math.h>
#include <algorithm>

namespace sfmba {

#define NB CHOL_NB
constexpr int CHOL_FUSED_MAX_BLOCKS = 40;
#define AT(i, j) A[(size_t)(i) + (size_t)(j) * ld]

// ------------------------------------------------------------------------------------------
// panel: factor A_kk, M_k = L_kk^-T, L_ik = A_ik L_kk^-T
//
// Wave-level, register-resident: lane r owns row r of a 64x64 tile in 64 VGPR pairs; the only
// communication is one column of L per step, published in LDS and read back as broadcasts.
//   factor (wave 0 of every workgroup, redundantly):  for j: l = a_j / sqrt(a_jj);  a_c -= l * L(c,j), c > j
//   solve  (every wave, one tile each):               for j: x = a_j / L(j,j);      a_c -= x * L(c,j), c > j
// i.e. the same rank-1 sweep with a different scalar; no __syncthreads inside the 64 steps.
// Tasks of step k: task 0 = identity tile (gives M_k = L_kk^-T for the back substitution),
// task t >= 1 = tile row k + t.  Three tasks per workgroup (waves 1..3; wave 0 factors the diagonal tile).
// ------------------------------------------------------------------------------------------

// FACTOR: a[j] /= sqrt(pivot) and the column is published in LDS; otherwise a[j] *= dinv[j].
// Then a[c] -= s * L(c, j) for c > j with L(c, j) read back from LDS as a broadcast.
template <bool FACTOR>
__device__ __forceinline__ void rank_one_sweep(double (&a)[NB], double (*Lc)[NB], const double* dinv, int lane, int kb, int d,
                                               int* info, bool report) {
    int badcol = 0;     // first non-positive pivot (1-based), reported ONCE after the sweep: a branch with an atomic inside
                        // the 64-step unrolled loop made the register allocator spill the whole tile
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        double s;
        if (FACTOR) {
            double dj = __shfl(a[j], j, 64);        // pivot = a[j] of lane j
            const bool ok = (dj > 0.0) && (dj <= 1.7e308);
            badcol = (!ok && badcol == 0 && kb + j < d) ? kb + j + 1 : badcol;
            dj = ok ? dj : 1.0;     // augmented / padded columns and failed pivots: keep going with a harmless value
            // 1/sqrt(pivot): hardware estimate + two Newton steps.  The generic 1.0 / sqrt() expansion is a ~40-deep
            // dependent fp64 chain (measured 32 cycles per dependent DFMA), i.e. most of a 64-step sequential panel.
            double di = __builtin_amdgcn_rsq(dj);
            {
                const double h = 0.5 * di;
                double e = fma(-dj * di, di, 1.0);
                di = fma(h, e, di);
                const double h2 = 0.5 * di;
                e = fma(-dj * di, di, 1.0);
                di = fma(h2, e, di);
            }
            s = (lane == j) ? dj * di : a[j] * di;
            a[j] = s;
            Lc[j][lane] = (lane >= j) ? s : 0.0;
            chol_wave_fence();
        } else {
            s = a[j] * dinv[j];
            a[j] = s;
        }
        // a[c] -= s * L(c, j) in chunks of 24 columns: the scheduler may batch the broadcast LDS reads of one chunk
        // (48 VGPRs) but not of the whole remaining row, which together with the 128 VGPRs of a[] would spill
#pragma unroll
        for (int c0 = j + 1; c0 < NB; c0 += 24) {
#pragma unroll
            for (int c = c0; c < c0 + 24 && c < NB; ++c) a[c] -= s * Lc[j][c];
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (FACTOR && report && lane == 0 && badcol != 0) atomicCAS(info, 0, badcol);
}

__global__ __launch_bounds__(256, 1) void k_chol_panel(double* __restrict__ A, int ld, int k, int d, int ntask,
                                                    double* __restrict__ minv, int* __restrict__ info) {
    __shared__ double Lc[NB][NB];        // Lc[j][r] = L(r, j): column j contiguous
    __shared__ double dinv[NB];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int kb = k * NB;
    double a[NB];
    if (w == 0) {
#pragma unroll
        for (int c = 0; c < NB; ++c) a[c] = (lane >= c) ? AT(kb + lane, kb + c) : 0.0;
        rank_one_sweep<true>(a, Lc, dinv, lane, kb, d, info, blockIdx.x == 0);
        dinv[lane] = 1.0 / Lc[lane][lane];
        if (blockIdx.x == 0) {
#pragma unroll
            for (int c = 0; c < NB; ++c) if (lane >= c) AT(kb + lane, kb + c) = a[c];
        }
    }
    __syncthreads();
    // wave 0 has just spent a whole sweep on the diagonal tile: the tiles of the panel go to waves 1..3 (three per workgroup), so
    // that a panel costs one factor sweep + one solve sweep instead of two sweeps back to back on wave 0
    if (w == 0) return;
    const int task = blockIdx.x * 3 + (w - 1);
    if (task >= ntask) return;
    if (task == 0) {
#pragma unroll
        for (int c = 0; c < NB; ++c) a[c] = (lane == c) ? 1.0 : 0.0;
    } else {
        const int ib = (k + task) * NB;
#pragma unroll
        for (int c = 0; c < NB; ++c) a[c] = AT(ib + lane, kb + c);
    }
    rank_one_sweep<false>(a, Lc, dinv, lane, kb, d, info, false);
    if (task == 0) {
        double* M = minv + (size_t)k * NB * NB;     // M[r + c*NB] = (L_kk^-T)(r,c)
#pragma unroll
        for (int c = 0; c < NB; ++c) M[lane + c * NB] = a[c];
    } else {
        const int ib = (k + task) * NB;
#pragma unroll
        for (int c = 0; c < NB; ++c) AT(ib + lane, kb + c) = a[c];
    }
}

// ------------------------------------------------------------------------------------------
// trailing update: A_ij -= L_ik L_jk^T  for k < j <= i
// ------------------------------------------------------------------------------------------
// One 64x64 tile per workgroup, one 32x32 quadrant per wave on the fp64 matrix cores:
// v_mfma_f64_16x16x4_f64, D[m][n] += sum_k Aop[m][k] Bop[k][n] with Aop[m = lane&15][k = lane>>4], Bop[k = lane>>4][n = lane&15]
// and D held as 4 doubles per lane at (m = (lane>>4) + 4*reg, n = lane&15)  (cdna_hip_programming.md, fragment layout).
// The product is formed transposed -- m runs over the tile's COLUMNS (rows of L_jk), n over its ROWS (rows of L_ik) --
// so that the 16 lanes sharing a register index write 16 consecutive rows of the column-major matrix (128 B).
typedef double mfma_d4 __attribute__((ext_vector_type(4)));

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
    const int lane = tid & 63, w = tid >> 6;
    const int rbase = 32 * (w & 1), cbase = 32 * (w >> 1);
    const int l15 = lane & 15, l4 = lane >> 4;
    mfma_d4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (mfma_d4){ 0.0, 0.0, 0.0, 0.0 };
#pragma unroll 4
    for (int kk = 0; kk < NB; kk += 4) {
        double av[2], bv[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) { av[a] = Lj[cbase + 16 * a + l15][kk + l4]; bv[a] = Li[rbase + 16 * a + l15][kk + l4]; }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bv[b], acc[a][b], 0, 0, 0);
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int c = cbase + 16 * a + l4 + 4 * v;      // m: column of the tile
                const int r = rbase + 16 * b + l15;             // n: row of the tile
                if (ib + r >= jb + c) AT(ib + r, jb + c) -= acc[a][b][v];
            }
}

// ------------------------------------------------------------------------------------------
// One launch per block column (d up to ~2500): panel solve, trailing update and the factorisation of the next diagonal tile.
//
// Step k, workgroup (i, j) with i >= j > k owns tile A_ij:
//   L_ik = A_ik M_k,  L_jk = A_jk M_k      (M_k = L_kk^-T from the previous launch: the triangular solves are GEMMs on the matrix
//                                           cores, formed redundantly by every workgroup that needs them -- 64^3 flops, ~1 us)
//   A_ij -= L_ik L_jk^T                    (written back; the panel tiles A_ik themselves stay as they are: L_ik = A_ik M_k is implied)
//   (i, j) = (k+1, k+1) only: factor the updated tile in LDS and form M_{k+1} on the way (chol_tile.h), store both.
// The factorisation of the diagonal tile is the serial chain of the whole algorithm (64 dependent pivots); everything else of a
// step hangs off it with one launch boundary instead of two, and no workgroup but that one ever executes the 64-step sweep
// (the panel kernel it replaces swept 64 steps in every workgroup: a factor sweep in wave 0, then a solve sweep per tile).
// Launch k = -1 factors tile (0, 0) alone.
// ------------------------------------------------------------------------------------------
constexpr int CS_TILE = CT_NB * CT_LDT;
constexpr int CS_LDS_DOUBLES = 3 * CS_TILE + CT_NB * CT_LDP + (CT_NB / CT_PB + 1) * CT_PB * CT_LDP + (CT_NB / CT_PB + 1) * CT_PB * CT_PB;
static_assert(CT_NB == NB, "tile size");

// acc[nb] (block row w of the product, block column nb) = sum_{t} A[r][t] E[t][c] over t < 16 (nb + 1): E is upper triangular.
// D layout: register v of a lane = element (r = 16 w + (lane >> 4) + 4 v, c = 16 nb + (lane & 15)).
__device__ __forceinline__ void tile_times_upper(const double* __restrict__ A, const double* __restrict__ E, mfma_d4 (&acc)[4], int w, int lane) {
    const int l15 = lane & 15, l4 = lane >> 4;
    double av[NB / 4];
#pragma unroll
    for (int kk = 0; kk < NB / 4; ++kk) av[kk] = A[(16 * w + l15) * CT_LDT + 4 * kk + l4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        acc[nb] = (mfma_d4){ 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
        for (int kk = 0; kk < 4 * (nb + 1); ++kk)
            acc[nb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kk], E[(4 * kk + l4) * CT_LDT + 16 * nb + l15], acc[nb], 0, 0, 0);
    }
}
__device__ __forceinline__ void tile_store_rows(double* __restrict__ A, const mfma_d4 (&acc)[4], int w, int lane) {
    const int l15 = lane & 15, l4 = lane >> 4;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int v = 0; v < 4; ++v) A[(16 * w + l4 + 4 * v) * CT_LDT + 16 * nb + l15] = acc[nb][v];
}

__global__ __launch_bounds__(256, 1) void k_chol_step(double* __restrict__ A, int ld, int k, int d, double* __restrict__ minv, int* __restrict__ info) {
    extern __shared__ __align__(16) double cs_lds[];
    double* B0 = cs_lds;                 // M_k, later the tile being factored
    double* B1 = B0 + CS_TILE;           // A_ik -> L_ik
    double* B2 = B1 + CS_TILE;           // A_jk -> L_jk
    double* X = B2 + CS_TILE;
    double* ED = X + CT_NB * CT_LDP;
    double* V = ED + (CT_NB / CT_PB + 1) * CT_PB * CT_LDP;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: the per-wave block assignments below are uniform branches
    const int l15 = lane & 15, l4 = lane >> 4;
    const int t = blockIdx.x;
    int ii = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((ii + 1) * (ii + 2) / 2 <= t) ++ii;
    while (ii * (ii + 1) / 2 > t) --ii;
    const int jj = t - ii * (ii + 1) / 2;
    const int ib = (k + 1 + ii) * NB, jb = (k + 1 + jj) * NB, kb = k * NB;
    const bool diag = ii == 0;           // jj <= ii: tile (k+1, k+1)
    if (k >= 0) {
        // the tile itself, straight into the accumulator layout of the update (m = column, n = row: 16 lanes = 16 consecutive rows)
        // (diagonal tile: the 10 blocks on and below the diagonal, dealt 3 / 3 / 2 / 2 to the waves:
        //   wave 0: (3,0) (3,1) (0,0)   wave 1: (3,2) (3,3) (1,1)   wave 2: (2,0) (2,1)   wave 3: (2,2) (1,0) )
        const int rb0 = w < 2 ? 3 : 2, cbA = w == 0 ? 0 : w == 1 ? 2 : w == 2 ? 0 : 2;
        const int rb1 = w == 0 ? 0 : 1, cb1 = w == 0 ? 0 : w == 1 ? 1 : 0;     // third block of waves 0, 1; second of wave 3
        const int rowB = w == 3 ? rb1 : rb0, colB = w == 3 ? cb1 : cbA + 1;
        mfma_d4 acc[4];
        if (!diag) {
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                for (int v = 0; v < 4; ++v) acc[cb][v] = AT(ib + 16 * w + l15, jb + 16 * cb + l4 + 4 * v);
        } else {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                acc[0][v] = AT(ib + 16 * rb0 + l15, ib + 16 * cbA + l4 + 4 * v);
                acc[1][v] = AT(ib + 16 * rowB + l15, ib + 16 * colB + l4 + 4 * v);
                acc[2][v] = w < 2 ? AT(ib + 16 * rb1 + l15, ib + 16 * cb1 + l4 + 4 * v) : 0.0;
            }
        }
        const double* M = minv + (size_t)k * NB * NB;
        {
            // all global loads of the three operand tiles in flight at once (48 per thread), then the LDS stores
            const int r = tid & (NB - 1), cq = tid >> 6;        // column c = cq + 4 m
            double mv[NB / 4], iv[NB / 4], jv[NB / 4];
#pragma unroll
            for (int m = 0; m < NB / 4; ++m) {
                const int c = cq + 4 * m;
                mv[m] = M[r + c * NB];
                iv[m] = AT(ib + r, kb + c);
                jv[m] = ii != jj ? AT(jb + r, kb + c) : 0.0;
            }
#pragma unroll
            for (int m = 0; m < NB / 4; ++m) {
                const int c = cq + 4 * m;
                B0[r * CT_LDT + c] = mv[m];
                B1[r * CT_LDT + c] = iv[m];
                if (ii != jj) B2[r * CT_LDT + c] = jv[m];
            }
        }
        __syncthreads();
        {
            mfma_d4 li[4];
            tile_times_upper(B1, B0, li, w, lane);
            tile_store_rows(B1, li, w, lane);      // wave w is the only reader of its block row of A_ik
            if (ii != jj) {
                tile_times_upper(B2, B0, li, w, lane);
                tile_store_rows(B2, li, w, lane);
            }
        }
        __syncthreads();
        const double* Lj = ii != jj ? B2 : B1;
        // (L_ik is NOT stored: the other workgroups of this launch still read the raw A_ik -- tile (i, k) belongs to every workgroup of
        // block row i and to workgroup (i', i) -- and nobody needs it later: the back substitution works on the raw panel tiles,
        // sum_i L_ij^T x_i = M_j^T sum_i A_ij^T x_i, see k_chol_backsolve)
        // A_ij -= L_ik L_jk^T, product formed transposed: D[m = column][n = row] = sum_t L_jk[c][t] L_ik[r][t]
        if (!diag) {
            double bv[NB / 4];
#pragma unroll
            for (int kk = 0; kk < NB / 4; ++kk) bv[kk] = B1[(16 * w + l15) * CT_LDT + 4 * kk + l4];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                for (int kk = 0; kk < NB / 4; ++kk)
                    acc[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(-Lj[(16 * cb + l15) * CT_LDT + 4 * kk + l4], bv[kk], acc[cb], 0, 0, 0);
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                for (int v = 0; v < 4; ++v) AT(ib + 16 * w + l15, jb + 16 * cb + l4 + 4 * v) = acc[cb][v];
            return;
        }
        {
            mfma_d4 t0 = acc[0], t1 = acc[1], t2 = acc[2];
#pragma unroll
            for (int kk = 0; kk < NB / 4; ++kk) {
                const double b0 = B1[(16 * rb0 + l15) * CT_LDT + 4 * kk + l4];
                t0 = __builtin_amdgcn_mfma_f64_16x16x4f64(-B1[(16 * cbA + l15) * CT_LDT + 4 * kk + l4], b0, t0, 0, 0, 0);
                const double b1 = w == 3 ? B1[(16 * rowB + l15) * CT_LDT + 4 * kk + l4] : b0;
                t1 = __builtin_amdgcn_mfma_f64_16x16x4f64(-B1[(16 * colB + l15) * CT_LDT + 4 * kk + l4], b1, t1, 0, 0, 0);
                if (w < 2) t2 = __builtin_amdgcn_mfma_f64_16x16x4f64(-B1[(16 * cb1 + l15) * CT_LDT + 4 * kk + l4], B1[(16 * rb1 + l15) * CT_LDT + 4 * kk + l4], t2, 0, 0, 0);
            }
            // B0 (M_k) is free since the barrier above
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                B0[(16 * rb0 + l15) * CT_LDT + 16 * cbA + l4 + 4 * v] = t0[v];
                B0[(16 * rowB + l15) * CT_LDT + 16 * colB + l4 + 4 * v] = t1[v];
                if (w < 2) B0[(16 * rb1 + l15) * CT_LDT + 16 * cb1 + l4 + 4 * v] = t2[v];
            }
        }
    } else {
        for (int idx = tid; idx < NB * NB; idx += 256) { const int r = idx % NB, c = idx / NB; B0[r * CT_LDT + c] = r >= c ? AT(ib + r, ib + c) : 0.0; }
    }
    __syncthreads();
    const int bad = chol_tile_factor(B0, X, ED, V, ib, d);
    if (bad != 0 && lane == 0 && w == 0) atomicCAS(info, 0, bad);
    double* Mn = minv + (size_t)(k + 1) * NB * NB;
    {
        const int r = tid & (NB - 1), q = r >> 4;
#pragma unroll
        for (int m = 0; m < NB / 4; ++m) {
            const int c = (tid >> 6) + 4 * m, cb = c >> 4;
            const double sc = X[c];             // 1 / sqrt(pivot): the tile is factored with unscaled columns
            const double u = sc * B0[r * CT_LDT + c];
            const double e = sc * ED[(q * CT_PB + (r & 15)) * CT_LDP + (c & 15)];
            if (r >= c) AT(ib + r, ib + c) = u;
            Mn[r + c * NB] = q < cb ? u : q == cb ? e : 0.0;
        }
    }
}

// rhs -> augmented row d (row-major column d); padded diagonal is already 1
// "not computed yet" marker of the one-launch back substitution (k_chol_backsolve): a signalling-NaN bit pattern no computation produces
constexpr unsigned long long CHOL_X_PENDING = 0x7FF4C0DEC0DE0001ull;
// rhs becomes row d of the matrix (the forward substitution rides along with the factorisation); its first `pending` entries are then
// marked "not computed yet" for the back substitution, which overwrites rhs with the solution
// ... and the padding rows d + 1 .. ld - 1 are written afresh (unit diagonal, zeros): the factorisation works in place and stores what it computes
// for them -- zeros and ones as long as everything is finite, but a factorisation that ran into an indefinite or near-singular matrix (fp32 Jacobians on
// a barely determined problem: tests/fuzz_parity.py found it) leaves 0 * inf = NaN there, nothing else ever rewrites those rows, and the last diagonal
// tile of EVERY later factorisation contains them: one invalid step used to make every following step of the handle invalid.
__global__ void k_augment(double* __restrict__ A, int ld, int d, double* __restrict__ rhs, int pending) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < d) AT(d, c) = rhs[c];
    if (c < pending) reinterpret_cast<unsigned long long*>(rhs)[c] = CHOL_X_PENDING;
    if (c < ld) for (int i = max(c, d + 1); i < ld; ++i) AT(i, c) = i == c ? 1.0 : 0.0;
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

// Back substitution in ONE launch (fused path, <= CHOL_FUSED_MAX_BLOCKS block columns).  The fused factorisation leaves the
// off-diagonal tiles of block column j as the RAW (fully updated) A_ij with L_ij = A_ij M_j implied, and the right-hand side as row d
// of the last block row, so   y_j = M_j^T a_j (a_j = A(d, column block j)),   sum_{i>j} L_ij^T x_i = M_j^T sum_{i>j} A_ij^T x_i   and
//   x_j = M_j M_j^T (a_j - sum_{i>j} A_ij^T x_i)                  (last block column: x = M y with y = row d of the stored factor).
// Workgroup j prefetches its tiles A_ij and M_j, then consumes the x_i in the order they appear (i = nblk-1 ... j+1).  The data
// announce themselves: k_augment marked every entry of x "not computed yet" (CHOL_X_PENDING), the producer writes its 64 values with
// agent-scope atomic stores and the consumer's 64 lanes poll them with agent-scope atomic loads -- one round trip per step, no flag,
// no release / acquire fence (an L2 write-back and an invalidate on the chain of every step with the flag version: 51 -> 26 us).
// Workgroup j waits only for workgroups dispatched BEFORE it (block index nblk-1-j), so the chain cannot deadlock however few of
// them are resident.  Per block column the chain is  x_{j+1} arrives -> 64x64 product -> two M_j products -> x_j leaves  instead of a
// kernel boundary plus the same products.
__global__ __launch_bounds__(256) void k_chol_backsolve(const double* __restrict__ A, int ld, int d, const double* __restrict__ minv,
                                                        double* __restrict__ x, int nblk) {
    __shared__ double xi[NB];
    __shared__ double part[4][NB];
    __shared__ double v[NB];
    const int j = nblk - 1 - (int)blockIdx.x;
    const int tid = threadIdx.x, c = tid & (NB - 1), seg = tid >> 6;
    const int jb = j * NB;
    const bool last = j == nblk - 1;
    const double* M = minv + (size_t)j * NB * NB;
    double mrow[16], mcol[16], t[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        mrow[m] = M[c + (size_t)(seg * 16 + m) * NB];          // row c of M_j, this thread's 16 columns
        mcol[m] = M[(seg * 16 + m) + (size_t)c * NB];          // column c of M_j, this thread's 16 rows
    }
    if (!last) {
#pragma unroll
        for (int m = 0; m < 16; ++m) t[m] = AT((nblk - 1) * NB + seg * 16 + m, jb + c);  // column c of A_ij, this thread's 16 rows
    }
    const double aj = (tid < NB && jb + tid < d) ? AT(d, jb + tid) : 0.0;      // (before the chain starts, not on it)
    double acc = 0.0;
    for (int i = nblk - 1; i > j; --i) {
        double tn[16];
        if (i - 1 > j) {
#pragma unroll
            for (int m = 0; m < 16; ++m) tn[m] = AT((i - 1) * NB + seg * 16 + m, jb + c);
        }
        // x_i: every entry carries its own "ready" (k_augment marked the vector; the producer's stores and these loads are agent-scope
        // atomics): one round trip per step instead of flag, then data, and no release / acquire (L2 write-back / invalidate) pair
        if (tid < NB) {
            const unsigned long long* xp = reinterpret_cast<const unsigned long long*>(x) + i * NB + tid;
            unsigned long long b = __hip_atomic_load(xp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (b == CHOL_X_PENDING) { __builtin_amdgcn_s_sleep(1); b = __hip_atomic_load(xp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            xi[tid] = __longlong_as_double((long long)b);
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < 16; ++m) acc = fma(t[m], xi[seg * 16 + m], acc);
        if (i - 1 > j) {
#pragma unroll
            for (int m = 0; m < 16; ++m) t[m] = tn[m];
        }
        __syncthreads();          // xi is rewritten by the next step
    }
    part[seg][c] = acc;
    __syncthreads();
    // w = a_j - sum (raw row d of the panel tile; in the last block column row d of the stored factor, which is y itself)
    if (tid < NB) v[tid] = aj - (part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid]);
    __syncthreads();
    if (!last) {
        // y-part: v <- M_j^T w
        double sy = 0.0;
#pragma unroll
        for (int m = 0; m < 16; ++m) sy = fma(mcol[m], v[seg * 16 + m], sy);
        __syncthreads();
        part[seg][c] = sy;
        __syncthreads();
        if (tid < NB) v[tid] = part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid];
        __syncthreads();
    }
    double sx = 0.0;
#pragma unroll
    for (int m = 0; m < 16; ++m) sx = fma(mrow[m], v[seg * 16 + m], sx);
    __syncthreads();
    part[seg][c] = sx;
    __syncthreads();
    if (tid < NB) {
        const double xr = part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid];
        __hip_atomic_store(&x[jb + tid], jb + tid < d ? xr : 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// The fused factorisation followed by the step-by-step back substitution (SFMBA_CHOL_BACKSOLVE=0, or more than 64 block columns):
// that one expects L_ik in place, so the panel tiles are multiplied by M_k once, after the last step (one workgroup per tile).
__global__ __launch_bounds__(256) void k_chol_apply_minv(double* __restrict__ A, int ld, int nblk, const double* __restrict__ minv) {
    __shared__ double Ti[NB][NB + 1];
    __shared__ double Mk[NB][NB + 1];
    // tile (i, k), i > k, from the linear index
    const int t = blockIdx.x;
    int i = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((i + 1) * (i + 2) / 2 <= t) ++i;
    while (i * (i + 1) / 2 > t) --i;
    const int k = t - i * (i + 1) / 2;
    ++i;                                            // (i - 1, k) enumerates the lower triangle incl. its diagonal: shift to i > k
    if (i >= nblk) return;
    const double* M = minv + (size_t)k * NB * NB;
    for (int idx = threadIdx.x; idx < NB * NB; idx += 256) { const int r = idx % NB, c = idx / NB; Ti[r][c] = AT(i * NB + r, k * NB + c); Mk[r][c] = M[r + c * NB]; }
    __syncthreads();
    for (int idx = threadIdx.x; idx < NB * NB; idx += 256) {
        const int r = idx % NB, c = idx / NB;
        double sum = 0.0;
        for (int q = 0; q <= c; ++q) sum = fma(Ti[r][q], Mk[q][c], sum);       // M_k is upper triangular
        AT(i * NB + r, k * NB + c) = sum;
    }
}

// d < 64 (up to ten cameras: the reference's own data sets start there): the whole solve in ONE launch of one workgroup -- the tile
// is read together with the right-hand side as its row d, factored with the inverse riding along, and x = L^-T y follows from the
// rows already in LDS.  Replaces k_augment + k_chol_step + k_chol_backsolve (three launches of a launch-bound LM iteration).
__global__ __launch_bounds__(256, 1) void k_chol_small(double* __restrict__ A, int ld, int d, double* __restrict__ rhs, double* __restrict__ minv,
                                                      int* __restrict__ info) {
    extern __shared__ __align__(16) double cs_lds[];
    double* U = cs_lds;
    double* X = U + CS_TILE;
    double* ED = X + CT_NB * CT_LDP;
    double* V = ED + (CT_NB / CT_PB + 1) * CT_PB * CT_LDP;
    __shared__ double yv[NB];
    const int tid = threadIdx.x;
    for (int idx = tid; idx < NB * NB; idx += 256) {
        const int r = idx % NB, c = idx / NB;
        double v = r >= c ? AT(r, c) : 0.0;
        if (r == d && c < d) v = rhs[c];             // the augmented row (the padded diagonal is already 1)
        if (r > d) v = r == c ? 1.0 : 0.0;           // padding rows: never what an earlier (failed) factorisation stored there (see k_augment)
        U[r * CT_LDT + c] = v;
    }
    __syncthreads();
    const int bad = chol_tile_factor(U, X, ED, V, 0, d);
    if (bad != 0 && tid == 0) atomicCAS(info, 0, bad);
    // y = scaled row d of the factor; x = L^-T y with L^-T = (s(col) E)(row, col), E from U (block row < block column) and ED (diagonal blocks)
    if (tid < NB) yv[tid] = tid < d ? X[tid] * U[d * CT_LDT + tid] : 0.0;
    for (int idx = tid; idx < NB * NB; idx += 256) {          // the factor and its inverse go where the step-by-step path keeps them
        const int r = idx % NB, c = idx / NB, q = r >> 4, cb = c >> 4;
        const double sc = X[c];
        const double u = sc * U[r * CT_LDT + c];
        if (r >= c) AT(r, c) = u;
        minv[r + c * NB] = q < cb ? u : q == cb ? sc * ED[(q * CT_PB + (r & 15)) * CT_LDP + (c & 15)] : 0.0;
    }
    __syncthreads();
    if (tid < NB) {
        const int r = tid, q = r >> 4;
        double sum = 0.0;
        for (int c = r; c < d; ++c) {
            const int cb = c >> 4;
            const double e = q < cb ? U[r * CT_LDT + c] : ED[(q * CT_PB + (r & 15)) * CT_LDP + (c & 15)];
            sum = fma(X[c] * e, yv[c], sum);
        }
        if (r < ld) rhs[r] = r < d ? sum : 0.0;
    }
}

void dense_cholesky_solve(hipStream_t s, DenseSolver* ws, double* S, double* rhs, int* info_dev, Profiler* prof) {
    const int ld = ws->ld, d = ws->d, nblk = ld / NB;
    if (nblk == 1) {
        static bool small_attr_set = false;
        if (!small_attr_set) { (void)hipFuncSetAttribute((const void*)k_chol_small, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * CS_LDS_DOUBLES)); small_attr_set = true; }
        ProfScope ps(prof, KID_CHOL_PANEL, s);
        hipLaunchKernelGGL(k_chol_small, dim3(1), dim3(256), sizeof(double) * CS_LDS_DOUBLES, s, S, ld, d, rhs, ws->minv, info_dev);
        return;
    }
    // (the two-kernel factorisation and the step-by-step back substitution take over beyond CHOL_FUSED_MAX_BLOCKS block columns: chosen by size only)
    const bool one_launch_back = nblk <= CHOL_FUSED_MAX_BLOCKS && nblk <= 64;
    { ProfScope ps(prof, KID_CHOL_AUGMENT, s);
      const int pending = one_launch_back ? nblk * NB : 0;
      hipLaunchKernelGGL(k_augment, dim3((ld + 255) / 256), dim3(256), 0, s, S, ld, d, rhs, pending); }
    if (nblk <= CHOL_FUSED_MAX_BLOCKS) {
        // one launch per block column (k_chol_step); beyond ~2500 unknowns the redundant panel GEMMs of the fused step cost more than
        // the launch they save and the two-kernel form below takes over
        static bool attr_set = false;
        if (!attr_set) { (void)hipFuncSetAttribute((const void*)k_chol_step, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * CS_LDS_DOUBLES)); attr_set = true; }
        for (int k = -1; k < nblk - 1; ++k) {
            ProfScope ps(prof, KID_CHOL_PANEL, s);
            const int m = nblk - k - 1;          // block rows below block column k
            hipLaunchKernelGGL(k_chol_step, dim3(k < 0 ? 1 : m * (m + 1) / 2), dim3(256), sizeof(double) * CS_LDS_DOUBLES, s, S, ld, k, d, ws->minv, info_dev);
        }
    } else
    for (int k = 0; k < nblk; ++k) {
        { ProfScope ps(prof, KID_CHOL_PANEL, s);
          const int ntask = nblk - k;
          hipLaunchKernelGGL(k_chol_panel, dim3((ntask + 2) / 3), dim3(256), 0, s, S, ld, k, d, ntask, ws->minv, info_dev); }
        const int m = nblk - k - 1;
        if (m > 0) { ProfScope ps(prof, KID_CHOL_UPDATE, s);
          hipLaunchKernelGGL(k_chol_update, dim3(m * (m + 1) / 2), dim3(256), 0, s, S, ld, k); }
    }
    if (one_launch_back) {
        ProfScope ps(prof, KID_CHOL_BACKSTEP, s);
        hipLaunchKernelGGL(k_chol_backsolve, dim3(nblk), dim3(256), 0, s, S, ld, d, ws->minv, rhs, nblk);
        return;
    }
    if (nblk <= CHOL_FUSED_MAX_BLOCKS && nblk > 1)
        hipLaunchKernelGGL(k_chol_apply_minv, dim3(nblk * (nblk - 1) / 2), dim3(256), 0, s, S, ld, nblk, ws->minv);
    { ProfScope ps(prof, KID_CHOL_EXTRACT, s);
      hipLaunchKernelGGL(k_extract_y, dim3((ld + 255) / 256), dim3(256), 0, s, S, ld, d, ws->y); }
    for (int k = nblk - 1; k >= 0; --k) {
        ProfScope ps(prof, KID_CHOL_BACKSTEP, s);
        hipLaunchKernelGGL(k_chol_backstep, dim3(k + 1), dim3(256), 0, s, S, ld, k, ws->minv, ws->y, rhs, d);
    }
}

}  // namespace sfmba
