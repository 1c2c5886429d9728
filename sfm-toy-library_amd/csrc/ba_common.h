// ba_common.h -- device helpers shared by the kernel files of the bundle-adjustment path (ba_kernels.hip, implicit_schur.hip,
// shard_exchange.hip): block / slot reductions, launch geometry of the point and camera passes, the stores into the CG's matrix, the
// post-linearisation bookkeeping, and the evaluation of one observation in the factored form of sfmba_device.h.
#pragma once
#include "ba_kernels.h"
#include "sfmba_device.h"
#include "../../include/sfmba.h"

namespace sfmba {

#define BLK 256

__device__ __forceinline__ double wave_max(double v) {
    v = fmax(v, xlane_get<32>(v)); v = fmax(v, xlane_get<16>(v)); v = fmax(v, xlane_get<8>(v));
    v = fmax(v, xlane_get<4>(v)); v = fmax(v, xlane_get<2>(v)); v = fmax(v, xlane_get<1>(v));
    return v;
}

__device__ __forceinline__ bool finite_d(double v) { return fabs(v) <= DBL_MAX; }

// N block-wide sums at once: the wave reductions advance in lock step (a shuffle is ~50 cycles of latency: N dependent chains
// of six in sequence, each with its own pair of barriers, were ~1 us at the tail of every wave of the point passes), one
// barrier; thread t < N returns the sum of value t (other threads: 0).  scratch: >= (blockDim.x / 64) * N doubles.
template <int N>
__device__ __forceinline__ double block_sums(double (&v)[N], double* scratch) {
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = wave_allsum(v[k]);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < N; ++k) scratch[w * N + k] = v[k];
    }
    __syncthreads();
    double s = 0.0;
    if ((int)threadIdx.x < N) {
        const int nw = (blockDim.x + 63) >> 6;
        for (int i = 0; i < nw; ++i) s += scratch[i * N + threadIdx.x];
    }
    return s;
}

// slotted accumulators: one atomic per value per workgroup, spread over NSLOT addresses
__device__ __forceinline__ double* slot_ptr(const DeviceBuffers& db, int which) {
    return db.slots + (size_t)(blockIdx.x % (unsigned)db.nslot) * SLOT_W + which;
}
// sum (or max for ACC_GMAX) of one accumulator over the NSLOT (= 64) slots, then clear it.
// Must be called by all 64 lanes of one wave; every lane returns the result.
__device__ inline double slots_take(const DeviceBuffers& db, int which) {
    const int lane = threadIdx.x & 63;
    if (which == ACC_GMAX) {
        double v = 0.0;
        for (int i = lane; i < db.nslot; i += 64) {
            double* p = db.slots + (size_t)i * SLOT_W + which;
            const double o = *p;
            *p = 0.0;
            v = (o > v || o != o) ? o : v;
        }
#define SFMBA_MAXSTEP(OFF) { const double o = xlane_get<OFF>(v); v = (o > v || o != o) ? o : v; }
        SFMBA_MAXSTEP(32) SFMBA_MAXSTEP(16) SFMBA_MAXSTEP(8) SFMBA_MAXSTEP(4) SFMBA_MAXSTEP(2) SFMBA_MAXSTEP(1)
#undef SFMBA_MAXSTEP
        return v;
    }
    double v = 0.0;
    for (int i0 = lane; i0 < db.nslot; i0 += 512) {    // fixed order: lane-strided (eight loads in flight), then the butterfly
        double t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int i = i0 + 64 * u; t[u] = db.slots[(size_t)(i < db.nslot ? i : lane) * SLOT_W + which]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int i = i0 + 64 * u; if (i < db.nslot) { v += t[u]; db.slots[(size_t)i * SLOT_W + which] = 0.0; } }
    }
    return wave_sum(v);
}

// N accumulators at once: all slot loads in flight together, then the clears, then the N butterflies in lock step.  Back-to-back
// slots_take() calls cannot overlap (the clearing stores of one fence off the loads of the next): four of them were most of
// k_finalize's 14 us (the focal wave), five of them of k_lm_control's 7.  Same summation order as slots_take (bitwise identical).
template <int N>
__device__ __forceinline__ void slots_take_n(const DeviceBuffers& db, const int (&which)[N], double (&out)[N]) {
    const int lane = threadIdx.x & 63;
    double v[N];
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = 0.0;
    for (int i0 = lane; i0 < db.nslot; i0 += 512) {
        double t[N][8];
#pragma unroll
        for (int k = 0; k < N; ++k)
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int i = i0 + 64 * u; t[k][u] = (u == 0 || db.nslot > 64) ? db.slots[(size_t)(i < db.nslot ? i : lane) * SLOT_W + which[k]] : 0.0; }
#pragma unroll
        for (int k = 0; k < N; ++k)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + 64 * u;
                if (i < db.nslot) {
                    if (which[k] == ACC_GMAX) v[k] = (t[k][u] > v[k] || t[k][u] != t[k][u]) ? t[k][u] : v[k]; else v[k] += t[k][u];
                    db.slots[(size_t)i * SLOT_W + which[k]] = 0.0;
                }
            }
    }
#define SFMBA_TAKE_STEP(OFF) \
    _Pragma("unroll") for (int k = 0; k < N; ++k) { \
        const double o = xlane_get<OFF>(v[k]); \
        if (which[k] == ACC_GMAX) v[k] = (o > v[k] || o != o) ? o : v[k]; else v[k] += o; }
    SFMBA_TAKE_STEP(32) SFMBA_TAKE_STEP(16) SFMBA_TAKE_STEP(8) SFMBA_TAKE_STEP(4) SFMBA_TAKE_STEP(2) SFMBA_TAKE_STEP(1)
#undef SFMBA_TAKE_STEP
#pragma unroll
    for (int k = 0; k < N; ++k) out[k] = v[k];
}

template <typename T>
__device__ __forceinline__ void load_obs(const void* base, int q, double& ox, double& oy) {
    const typename ObsXY<T>::type v = reinterpret_cast<const typename ObsXY<T>::type*>(base)[q];
    ox = (double)v.x; oy = (double)v.y;
}


// workgroup of the two point passes: PBK / 64 waves that share nothing but the final block reduction
#ifndef PBK
#define PBK 128
#endif
#define WPB (PBK / 64)
// minimum waves per SIMD the point passes are compiled for (register budget 512 / this); 0 = let the compiler decide
#ifndef SFMBA_PB_WAVES
#define SFMBA_PB_WAVES 0
#endif
#if SFMBA_PB_WAVES > 0
#define PB_BOUNDS __launch_bounds__(PBK, SFMBA_PB_WAVES)
#else
#define PB_BOUNDS __launch_bounds__(PBK)
#endif

__device__ __forceinline__ void wave_lds_fence() {
    // LDS traffic of one wave is processed in order; this only stops the compiler from moving the
    // accesses and waits for outstanding LDS writes before other lanes of the wave read them
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

#define PB_LPP 4    // lanes per point of the point passes (k_point_build, k_point_update, k_imp_points)

#define CD_N 48      // Sjj(21) udiag(6) Sjf(6) bc(6) rhs(6) uff bf + pad
#define CD_BLK 256   // threads of a workgroup; a lane takes SFMBA_CAM_CHUNK / CD_BLK observations of the chunk, one after the other
#define CD_OBS (SFMBA_CAM_CHUNK / CD_BLK)
static_assert(SFMBA_CAM_CHUNK % CD_BLK == 0, "camera chunk length");



// one entry of the preconditioned reduced matrix (fp64, or fp32 when the streaming CG path asked for it)
__device__ __forceinline__ void store_F(const DeviceBuffers& db, size_t idx, double v) {
    if (db.pcg_F32) db.pcg_F32[idx] = (float)v; else db.pcg_F[idx] = v;
}

// one entry of an off-diagonal block of the preconditioned matrix: both triangles of the CG's matrix, or -- sharded CG path -- the
// block's slot in the all-reduce buffer (k_shard_offdiag's layout; the matrix is written after the sum over the ranks)
__device__ __forceinline__ void store_block_entry(const DeviceStructure& ds, const DeviceBuffers& db, int b, int2 cj, int r, int c, double v) {
    if (db.shard_blocks32 || db.shard_blocks) {
        const size_t o = (size_t)((long long)(b - cj.x - 1) + (db.shard_row_shift ? db.shard_row_shift[cj.x] : 0)) * 36 + 6 * r + c;
        if (db.shard_blocks32) db.shard_blocks32[o] = (float)v; else db.shard_blocks[o] = v;
        return;
    }
    store_F(db, (size_t)(6 * cj.x + r) * ds.ld + 6 * cj.y + c, v);
#ifndef SFMBA_WHATIF_UPPER_ONLY       // (timing what-if: the lower triangle never written -- wrong results where a kernel reads it)
    if (!db.pcg_upper_only) store_F(db, (size_t)(6 * cj.y + c) * ds.ld + 6 * cj.x + r, v);
#endif
}

// after a linearisation: initial cost (iteration 0), gradient tolerance, evaluation failure.  One wave.
__device__ inline void post_linearisation(const DeviceStructure& ds, const DeviceBuffers& db) {
    LMState* st = db.st;
    const int lin_acc[3] = { ACC_GMAX, ACC_BAD_LIN, ACC_LIN_COST };
    double lin[3];
    slots_take_n<3>(db, lin_acc, lin);
    const double gmax = lin[0], bad_lin = lin[1], lin_cost = lin[2];
    if ((threadIdx.x & 63) != 0) return;
    if (st->termination != -1) return;
    if (bad_lin != 0.0) {
        st->termination = SFMBA_FAILURE;
        st->message = st->iter == 0 ? MSG_INITIAL_EVAL_FAILED : MSG_EVAL_FAILED;
    }
    if (st->x_is_new) {
        st->jacobian_evals++;
        st->gmax = gmax;
        if (st->iter == 0) {
            st->cost = 0.5 * lin_cost;
            if (db.trace_cap > 0) {
                TraceRow row = {};
                row.iteration = 0; row.cost = st->cost; row.gradient_max_norm = gmax; row.trust_region_radius = st->radius;
                db.trace[0] = row;
            }
        } else if (st->iter < db.trace_cap) {
            db.trace[st->iter].gradient_max_norm = gmax;
        }
        if (st->termination == -1 && gmax <= st->gradient_tolerance) {
            st->termination = SFMBA_CONVERGENCE;
            st->message = MSG_GRADIENT_TOL;
        }
        st->x_is_new = 0;
    }
}

// one observation in the factored form of sfmba_device.h, as the back-substitution and the implicit Schur product need it: projection,
// X_g, u = P (Q dw x X_g + dt) for the direction in `dr` (step-table layout), C = (P R) L~
struct ImpObs { double xg[3], u[2], fz, xp, yp; };
template <typename T, typename CamPtr, typename DirPtr>
__device__ __forceinline__ void imp_eval(const CamPtr& ct, const DirPtr& dr, double focal, const PtRecA<T>& pa, ImpObs& o, T (&C)[6]) {
    const double rx = ct[CT_R + 0] * pa.X[0] + ct[CT_R + 1] * pa.X[1] + ct[CT_R + 2] * pa.X[2];
    const double ry = ct[CT_R + 3] * pa.X[0] + ct[CT_R + 4] * pa.X[1] + ct[CT_R + 5] * pa.X[2];
    const double rz = ct[CT_R + 6] * pa.X[0] + ct[CT_R + 7] * pa.X[1] + ct[CT_R + 8] * pa.X[2];
    Proj pr;
    pr.iz = fast_rcp(rz + ct[CT_T + 2]);
    pr.xp = (rx + ct[CT_T + 0]) * pr.iz;
    pr.yp = (ry + ct[CT_T + 1]) * pr.iz;
    const bool first_order = dr[ST_SMALL] != 0.0;
    o.xg[0] = first_order ? pa.X[0] : rx; o.xg[1] = first_order ? pa.X[1] : ry; o.xg[2] = first_order ? pa.X[2] : rz;
    const double dq0 = dr[ST_DQ], dq1 = dr[ST_DQ + 1], dq2 = dr[ST_DQ + 2];
    const double v0 = dq1 * o.xg[2] - dq2 * o.xg[1] + dr[ST_DT], v1 = dq2 * o.xg[0] - dq0 * o.xg[2] + dr[ST_DT + 1], v2 = dq0 * o.xg[1] - dq1 * o.xg[0] + dr[ST_DT + 2];
    o.fz = focal * pr.iz; o.xp = pr.xp; o.yp = pr.yp;
    o.u[0] = o.fz * (v0 - pr.xp * v2);
    o.u[1] = o.fz * (v1 - pr.yp * v2);
    T B[6];
    point_block<T>(ct, pr, focal, B);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const T b0 = B[3 * r], b1 = B[3 * r + 1], b2 = B[3 * r + 2];
        C[3 * r + 0] = b0 * pa.L[0];
        C[3 * r + 1] = b0 * pa.L[1] + b1 * pa.L[2];
        C[3 * r + 2] = b0 * pa.L[3] + b1 * pa.L[4] + b2 * pa.L[5];
    }
}


}  // namespace sfmba
