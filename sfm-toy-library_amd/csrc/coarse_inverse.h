// coarse_inverse.h -- the 8 x 8 inverse of the coarse matrix E = W~^T S~ W~ of the two-level CG preconditioner, shared by the one-GPU CG
// kernels (dense_solver.hip) and the distributed CG of the sharded solve (dist_cg.hip).
#pragma once
#include "sfmba_device.h"

namespace sfmba {

// LDS traffic of one wave is processed in order: this stops the compiler from moving the accesses and waits for the wave's outstanding LDS writes
__device__ __forceinline__ void chol_wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

// Einv by a Jacobi-scaled Gauss-Jordan elimination spread over the 64 lanes of ONE wave (lane = one entry; eight dependent
// steps instead of a ~600-deep serial chain: 28 -> ~3 us).  Returns this lane's entry (row lane / 8, column lane % 8).
// A pivot below 1e-10 of the unit diagonal means the vector depends on the earlier ones (degenerate configuration, fewer
// cameras than gauge freedoms): the vector is dropped (its row and column of Einv are zero).  sa, sb: 64 doubles of LDS each.
__device__ __forceinline__ double coarse_invert_wave(const double* tot, double* sa, double* sb) {
    constexpr int N = 8;
    const int lane = threadIdx.x & 63;
    const int i = lane >> 3, j = lane & 7;
    const double dii = tot[i * N + i], djj = tot[j * N + j];
    const bool ki = dii > 0.0 && dii <= 1.7e308, kj = djj > 0.0 && djj <= 1.7e308;
    const double si = ki ? fast_rsq(dii) : 0.0, sj = kj ? fast_rsq(djj) : 0.0;
    double a = (i == j) ? 1.0 : 0.5 * (tot[i * N + j] + tot[j * N + i]) * si * sj;
    double b = (i == j) ? 1.0 : 0.0;
    unsigned dropped = 0;                                 // bit k: vector k dropped (identical in every lane)
#pragma unroll
    for (int k = 0; k < N; ++k) if (!(tot[k * N + k] > 0.0 && tot[k * N + k] <= 1.7e308)) dropped |= 1u << k;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        sa[i * N + j] = a; sb[i * N + j] = b;
        chol_wave_fence();
        const double piv = sa[k * N + k];
        const bool ok = !((dropped >> k) & 1u) && piv > 1e-10;
        if (!ok) dropped |= 1u << k;
        const double aik = ok ? sa[i * N + k] : 0.0;
        const double akj = ok ? sa[k * N + j] : (k == j ? 1.0 : 0.0), bkj = ok ? sb[k * N + j] : 0.0;
        const double ip = ok ? fast_rcp(piv) : 1.0;
        if (i == k) { a = akj * ip; b = bkj * ip; }
        else { a = fma(-aik * ip, akj, a); b = fma(-aik * ip, bkj, b); }
        if (!ok && (i == k || j == k)) { a = (i == j) ? 1.0 : 0.0; b = 0.0; }
        chol_wave_fence();
    }
    const bool gone = ((dropped >> i) & 1u) || ((dropped >> j) & 1u);
    return gone ? 0.0 : b * si * sj;
}


}  // namespace sfmba
