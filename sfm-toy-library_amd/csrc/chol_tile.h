// chol_tile.h -- one workgroup factors one 64x64 SPD tile held in LDS and forms the inverse of the factor on the way
// (device functions shared by dense_solver.hip and tools/micro/chol_tile_bench.hip).
//
// The diagonal tile of a blocked Cholesky is the serial part of the factorisation: 64 pivots, each one a dependent chain
//   pivot -> 1/sqrt -> scaled column -> update of the next pivot
// (a dependent fp64 operation is 32 cycles on gfx950 and v_rsq_f64 48, tools/micro/lat_bench.hip).  What is not on that chain is
// kept out of its way:
//   * block columns of 16, swept by wave 0 with the 16 columns in registers.  The uniform values of a step (the pivot and the
//     entries L(c, j) of the 16 x 16 diagonal block D) never leave the vector registers: EVERY row of 16 lanes carries its own
//     copy of D (lane 16 q + r: row r of D), so they are row-local DPP broadcasts (v_mov_b32_dpp row_newbcast: ~10 cycles;
//     v_readlane into an SGPR was measured at ~18 cycles each, an LDS round trip is ~150 on the chain);
//   * the identity rides along ([T; I] -> [L; L^-T]): the same steps with the same broadcasts, so the inverse costs no sweep
//     and no chain of its own.  Of the 128 rows only 64 matter in a given block column b -- row 16 q + r of T for q > b
//     (below the diagonal block), of E = L^-T for q <= b (E is upper triangular) -- so lane 16 q + r carries exactly that row:
//     two FMAs per broadcast (copy of D, own row);
//   * the rank-16 updates of everything right of the panel (T rows below, E rows above: again one row per lane) are done by
//     all four waves from LDS.
// The b loop is NOT unrolled: one pass is ~2k instructions that stay in the instruction cache for the other three.
#pragma once
#include <hip/hip_runtime.h>
#ifdef CT_DEBUG_CLK
__device__ __forceinline__ long long ct_clk() { __builtin_amdgcn_sched_barrier(0); const long long t = clock64(); __builtin_amdgcn_sched_barrier(0); return t; }
#endif
#include <utility>

namespace sfmba {

constexpr int CT_NB = 64;          // tile
constexpr int CT_PB = 16;          // panel (block column) width = DPP row
constexpr int CT_LDT = CT_NB + 1;  // LDS row pitch of the tile (doubles)
constexpr int CT_LDP = CT_PB + 1;  // LDS row pitch of the panel
#ifndef SFMBA_CT_NEWTON
#define SFMBA_CT_NEWTON 2          // Newton steps on v_rcp_f64 on the pivot chain (estimate ~5e-8 relative; one step ~4e-15, two ~2e-16)
#endif
// LDS of chol_tile_factor (doubles): U tile, X panel, ED diagonal blocks of E, V published columns of one panel
constexpr int CT_LDS_DOUBLES = CT_NB * CT_LDT + CT_NB * CT_LDP + (CT_NB / CT_PB + 1) * CT_PB * CT_LDP + (CT_NB / CT_PB + 1) * CT_PB * CT_PB;

// value of lane L of the caller's row of 16 lanes
template <int L> __device__ __forceinline__ double ct_bcast16(double v) {
    // old = the source itself: every lane of a row has a valid source lane, and a constant would cost a v_mov per half
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), 0x150 + L, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), 0x150 + L, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// One pivot of the panel sweep, square-root free on the critical path: with v = the unscaled column I of the diagonal block and
// p = v(I) its pivot, the rank-1 update is  a(r, c) -= v(r) v(c) / p.  The next pivot is  a(I+1, I+1) - (v(I+1) v(I+1)) / p  with
// the product formed while the reciprocal is still in flight, so the chain per pivot is  broadcast -> rcp (+ Newton) -> one FMA
// instead of  broadcast -> rsq -> scaled column -> broadcast -> FMA.  The columns further right take v(c) from LDS: every row of
// lanes publishes v (identical copies, no branch), every lane reads copy 0 back as broadcasts -- the wave's LDS operations
// execute in order -- and their updates are deferred by one step: they fill the issue slots under the NEXT pivot's reciprocal
// chain.  The wave issues in order, so the interleaving is spelled out (scheduling barriers between the groups): left to itself
// the compiler emits the 4-deep reciprocal chain and, worse, the 7-deep 1/sqrt chain of the column scaling back to back
// (measured 407 cycles per pivot instead of ~250).  Columns stay UNSCALED throughout the tile (L(:, c) = v_c / sqrt(p_c) is applied
// by whoever stores the result): nothing inside the tile needs the square roots, only v v^T / p.
// The reciprocal chain as volatile asm: plain builtins are sunk to their first use by the compiler's own passes (the scheduling
// barriers only bind the machine scheduler), which puts the whole chain back to back at the end of the step.
__device__ __forceinline__ double ct_chain_rcp(double p) {
    double y;
    asm volatile("v_rcp_f64 %0, %1\n\ts_nop 1" : "=v"(y) : "v"(p));     // trans-op result read by a VALU op: wait states by hand
    return y;
}
__device__ __forceinline__ double ct_chain_err(double p, double y) {      // 1 - p y
    double e;
    asm volatile("v_fma_f64 %0, -%1, %2, 1.0" : "=v"(e) : "v"(p), "v"(y));
    return e;
}
__device__ __forceinline__ double ct_chain_fix(double y, double e) {      // y + y e
    double z;
    asm volatile("v_fma_f64 %0, %1, %2, %1" : "=v"(z) : "v"(y), "v"(e));
    return z;
}
// pivot as the sweep used it (a non-positive or non-finite one is replaced by 1), its reciprocal and 1 / sqrt
__device__ __forceinline__ double ct_pivot_fixed(double p) { return __builtin_amdgcn_class(p, 0x100 | 0x080) ? p : 1.0; }
__device__ __forceinline__ double ct_rcp(double p) {
    double y = __builtin_amdgcn_rcp(p);
    y = fma(y, fma(-p, y, 1.0), y);
    return fma(y, fma(-p, y, 1.0), y);
}
__device__ __forceinline__ double ct_rsqrt(double p) {
    double y = __builtin_amdgcn_rsq(p);
    y = fma(0.5 * y, fma(-p * y, y, 1.0), y);
    return fma(0.5 * y, fma(-p * y, y, 1.0), y);
}
struct CtCarry { double u, ux; };       // (own entry of the previous pivot column) / (previous pivot), both register sets

template <int I>
__device__ __forceinline__ void ct_pivot_step(double (&dc)[CT_PB], double (&x)[CT_PB], CtCarry& prev,
                                              double* __restrict__ V, int q, int r, unsigned& badbits) {
    // columns I + 1 .. 15 of the previous pivot column (published one step ago)
    double vf[CT_PB];
#pragma unroll
    for (int c = I + 1; c < CT_PB; ++c) vf[c] = I >= 1 ? V[(I >= 1 ? I - 1 : 0) * CT_PB + c] : 0.0;
    // rows above the diagonal (r < I) are not masked: whatever they hold only ever reaches entries above the diagonal of D
    const double v = dc[I];
    const double xi = x[I];
    const double p = ct_bcast16<I>(v);
    double y = ct_chain_rcp(p);
    // positive and finite?  (padding / augmented columns and failed pivots: keep going with 1); off the chain until the select
    const bool ok = __builtin_amdgcn_class(p, 0x100 | 0x080);      // +normal | +denormal
    badbits |= ok ? 0u : (1u << I);
    double w = 0.0, wx = 0.0;
    if constexpr (I + 1 < CT_PB) {
        const double vb = ct_bcast16<(I + 1 < CT_PB ? I + 1 : I)>(v);
        w = v * vb; wx = xi * vb;
        asm volatile("" : "+v"(w), "+v"(wx));      // formed here, under the reciprocal, not after it
    }
    if constexpr (I + 2 < CT_PB) V[(q * CT_PB + I) * CT_PB + r] = v;
    __builtin_amdgcn_sched_barrier(0);
    double e = ct_chain_err(p, y);
#pragma unroll
    for (int c = I + 1; c < CT_PB; c += 4) if (I >= 1) { dc[c] = fma(-prev.u, vf[c], dc[c]); x[c] = fma(-prev.ux, vf[c], x[c]); }
    __builtin_amdgcn_sched_barrier(0);
    y = ct_chain_fix(y, e);
#pragma unroll
    for (int c = I + 2; c < CT_PB; c += 4) if (I >= 1) { dc[c] = fma(-prev.u, vf[c], dc[c]); x[c] = fma(-prev.ux, vf[c], x[c]); }
    __builtin_amdgcn_sched_barrier(0);
#if SFMBA_CT_NEWTON >= 2
    e = ct_chain_err(p, y);
#endif
#pragma unroll
    for (int c = I + 3; c < CT_PB; c += 4) if (I >= 1) { dc[c] = fma(-prev.u, vf[c], dc[c]); x[c] = fma(-prev.ux, vf[c], x[c]); }
    __builtin_amdgcn_sched_barrier(0);
#if SFMBA_CT_NEWTON >= 2
    y = ct_chain_fix(y, e);
#endif
#pragma unroll
    for (int c = I + 4; c < CT_PB; c += 4) if (I >= 1) { dc[c] = fma(-prev.u, vf[c], dc[c]); x[c] = fma(-prev.ux, vf[c], x[c]); }
    __builtin_amdgcn_sched_barrier(0);
    y = ok ? y : 1.0;
    if constexpr (I + 1 < CT_PB) {
        dc[I + 1] = fma(-w, y, dc[I + 1]);
        x[I + 1] = fma(-wx, y, x[I + 1]);
    }
    prev.u = v * y; prev.ux = xi * y;
    V[((CT_NB / CT_PB) * CT_PB + I) * CT_PB + r] = y;      // 1 / pivot for the trailing update (16 copies: no branch, no conflict)
}

template <int... Is>
__device__ __forceinline__ unsigned ct_sweep(double (&dc)[CT_PB], double (&x)[CT_PB], double* __restrict__ V, int q, int r,
                                             std::integer_sequence<int, Is...>) {
    CtCarry prev = { 0.0, 0.0 };
    unsigned badbits = 0;      // bit I: pivot I of this panel was not positive and finite
    (ct_pivot_step<Is>(dc, x, prev, V, q, r, badbits), ...);
    return badbits;
}

// Trailing update of panel b: at most NJ blocks per wave, entries w, w + 4, w + 8 of the packed list (4 bits per block: q | cb << 2).
// No branch between the loads and the last matrix instruction: a wave with fewer than NJ blocks repeats its first one and skips
// the store (a branch per block makes the compiler shuttle every accumulator through the AGPRs around every v_mfma).
template <int NJ>
__device__ __forceinline__ void ct_trailing(double* __restrict__ U, const double* __restrict__ X, const double* __restrict__ V,
                                            unsigned long long list, int nblk, int b, int w, int lane) {
    typedef double ct_d4 __attribute__((ext_vector_type(4)));
    const int l15 = lane & 15, l4 = lane >> 4;
    const int n = (nblk - w + 3) >> 2;
    int bq[NJ], bc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const unsigned e = (unsigned)(list >> (4 * (w + (j < n ? 4 * j : 0)))) & 15u;
        bq[j] = e & 3; bc[j] = e >> 2;
    }
    double yv[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) yv[kk] = -V[((CT_NB / CT_PB) * CT_PB + 4 * kk + l4) * CT_PB];
    ct_d4 acc[NJ];
    double av[NJ][4], bv[NJ][4];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const double* Y = U + (bq[j] * CT_PB + l4) * CT_LDT + bc[j] * CT_PB + l15;
        const double* Xa = X + (bq[j] * CT_PB + l15) * CT_LDP + l4;
        const double* Xb = X + (bc[j] * CT_PB + l15) * CT_LDP + l4;
#pragma unroll
        for (int v = 0; v < 4; ++v) { const double y0 = Y[4 * v * CT_LDT]; acc[j][v] = bq[j] == b ? 0.0 : y0; }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) { av[j][kk] = Xa[4 * kk]; bv[j][kk] = Xb[4 * kk]; }
    }
    // k outermost: the accumulators are independent, so the matrix instructions issue back to back
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[j][kk] * yv[kk], bv[j][kk], acc[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        if (j < n) {
            double* Y = U + (bq[j] * CT_PB + l4) * CT_LDT + bc[j] * CT_PB + l15;
#pragma unroll
            for (int v = 0; v < 4; ++v) Y[4 * v * CT_LDT] = acc[j][v];
        }
    }
}

// U: LDS tile [CT_NB][CT_LDT] holding the SPD tile (lower triangle significant), X: panel [CT_NB][CT_LDP], ED: [4 + 1][CT_PB][CT_LDP] (the last
// block is a dump), V: [4 + 1][CT_PB][CT_PB] scratch
// (four copies of the published pivot columns, then the reciprocal pivots of the panel).
// On return, with q = row / 16, cb = column / 16 and the column scale s(col) = X[col] = 1 / sqrt(pivot of that column):
//   L(row, col)      = s(col) U[row][col]                      for q >= cb  (the strict upper part of the diagonal blocks: garbage)
//   L^-T(row, col)   = s(col) U[row][col]                      for q <  cb
//                    = s(col) ED[q][row % 16][col % 16]        for q == cb  (upper triangular: zeros below the diagonal)
//                    = 0                                       for q >  cb
// col0: global index of the tile's first column, d: number of true columns (a non-positive pivot in a column >= d -- padding,
// the augmented right-hand side -- is replaced by 1 without a report).  Returns, in every thread of wave 0 (0 elsewhere), the
// 1-based global column of the first non-positive pivot, or 0.  Must be called by all 256 threads; U must be complete
// (__syncthreads() before the call is the caller's business); ends with a __syncthreads().
__device__ __forceinline__ int chol_tile_factor(double* __restrict__ U, double* __restrict__ X, double* __restrict__ ED, double* __restrict__ V, int col0, int d) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave index as a scalar: branches on it are uniform, and the compiler must know
    const int q = lane >> 4, r = lane & 15;
    int badcol = 0;
#ifdef CT_DEBUG_CLK
    long long clk_sweep = 0, clk_load = 0, clk_trail = 0, clk_store = 0;
#endif
#pragma unroll 1
    for (int b = 0; b < CT_NB / CT_PB; ++b) {
        const int c0 = b * CT_PB;
#ifdef CT_DEBUG_CLK
        const long long k0 = ct_clk();
        long long k1 = k0, k2 = k0;
#endif
        if (w == 0) {
            double dc[CT_PB], x[CT_PB];
#pragma unroll
            for (int c = 0; c < CT_PB; ++c) {
                dc[c] = U[(c0 + r) * CT_LDT + c0 + c];
                // own row: T below the diagonal block, E above it (zero until an earlier panel touched it), identity in it
                const double own = U[lane * CT_LDT + c0 + c];
                x[c] = q == b ? (r == c ? 1.0 : 0.0) : own;
            }
#ifdef CT_DEBUG_CLK
            k1 = ct_clk();
#endif
            unsigned bad = ct_sweep(dc, x, V, q, r, std::make_integer_sequence<int, CT_PB>());
            // a non-positive pivot in a column >= d (padding, the augmented right-hand side) is not reported
            const int nreal = d - (col0 + c0);
            bad &= nreal >= CT_PB ? 0xffffu : nreal <= 0 ? 0u : ((1u << nreal) - 1u);
            if (bad != 0 && badcol == 0) badcol = col0 + c0 + __builtin_ctz(bad) + 1;
#ifdef CT_DEBUG_CLK
            k2 = ct_clk();
#endif
            // branch-free stores (a branch here would split the sweep's basic block and strand the final column scalings behind it)
            const int edrow = ((q == b ? b : CT_NB / CT_PB) * CT_PB + r) * CT_LDP;
#pragma unroll
            for (int c = 0; c < CT_PB; ++c) {
                X[lane * CT_LDP + c] = x[c];
                U[lane * CT_LDT + c0 + c] = q == b ? dc[c] : x[c];
                ED[edrow + c] = x[c];
            }
        }
        __syncthreads();
#ifdef CT_DEBUG_CLK
        const long long k3 = ct_clk();
#endif
        // trailing update on the fp64 matrix cores: Y[row][cc] -= sum_i X[row][i] X[cc][i] / p_i for cc >= c0 + 16, in 16 x 16
        // blocks (q, cb): Y = T for the blocks on and below the diagonal (q >= cb), Y = E = L^-T for the block rows q <= b (a block
        // row of E enters as zero in its own panel).  X[cc] is a row of the T panel; 1 / p_i was left in V by the sweep.  One block =
        // 4 x v_mfma_f64_16x16x4_f64 (k = 16); the blocks are dealt round-robin to the four waves (at most three each) and a wave
        // issues the operand loads of all its blocks before the first matrix instruction.  Operand layout
        // (cdna_hip_programming.md): A[m = lane & 15][k = lane >> 4], B[k = lane >> 4][n = lane & 15], D[m = (lane >> 4) + 4 v][n = lane & 15].
        if (b == 0) ct_trailing<3>(U, X, V, 0xFCBA87654ull, 9, 0, w, lane);
        else if (b == 1) ct_trailing<2>(U, X, V, 0xFDCBA98ull, 7, 1, w, lane);
        else if (b == 2) ct_trailing<1>(U, X, V, 0xFEDCull, 4, 2, w, lane);
        __syncthreads();
#ifdef CT_DEBUG_CLK
        const long long k4 = ct_clk(); clk_load += k1 - k0; clk_sweep += k2 - k1; clk_store += k3 - k2; clk_trail += k4 - k3;
#endif
    }
    // column scales 1 / sqrt(p_c) for whoever stores the result (the panel buffer is free now)
    if (tid < CT_NB) X[tid] = ct_rsqrt(ct_pivot_fixed(U[tid * (CT_LDT + 1)]));
    __syncthreads();
#ifdef CT_DEBUG_CLK
    if (tid == 0 && blockIdx.x == 0) { CT_DEBUG_CLK[0] = clk_load; CT_DEBUG_CLK[1] = clk_sweep; CT_DEBUG_CLK[2] = clk_trail; CT_DEBUG_CLK[3] = clk_store; }
#endif
    return badcol;
}

}  // namespace sfmba
