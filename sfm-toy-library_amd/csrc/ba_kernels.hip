// ba_kernels.hip -- hand-written gfx950 kernels of the bundle-adjustment hot path.
//
// Reference path replaced: everything ceres::Solve does per LM iteration for the problem that
// SfMToyLib/SfMBundleAdjustmentUtils.cpp:142-179 builds (residual+Jacobian evaluation, Schur
// elimination of the points, back-substitution, trial-point cost) -- see DESIGN.md for the
// kernel-by-kernel map and SURVEY.md Appendix A for the math.
//
//   k_point_build     (point-major)  per point: V, b, E_f, Cholesky inverse of V + D^2, the per-point table entry; NOTHING per observation
//   k_cam_diag_f      (camera-major) U_jj minus the self terms, S_jf, b_c, rhs, undamped diagonal of one camera (re-evaluating)
//   k_schur_pairs     one wave (k_schur_pairs_sub_f: one 16-lane group) per off-diagonal 6x6 block over a precomputed list of pair points
//   k_finalize        damping of the reduced diagonal, gradient max-norm, block-Jacobi factors (PCG)
//   k_cam_update / k_point_update  back-substitution, trial point, trial cost, model cost change
//   k_lm_control   accept/reject + trust-region update on the device (no host round trip needed)
#include "ba_kernels.h"
#include "sfmba_device.h"
#include "ba_common.h"
#include "../../include/sfmba.h"
#include <cstdlib>
#include <algorithm>

namespace sfmba {

// ------------------------------------------------------------------------------------------
// camera tables
// ------------------------------------------------------------------------------------------
__device__ void make_cam_table(const double cam[6], const double* scale6, double* ct) {
    const double w0 = cam[0], w1 = cam[1], w2 = cam[2];
    const double theta2 = w0 * w0 + w1 * w1 + w2 * w2;
    double R[9], K[9];
    double small = 0.0;
    double cq = 1.0 / 12.0;
    if (theta2 > DBL_EPSILON) {
        const double theta = sqrt(theta2);
        double s, c;
        sincos(theta, &s, &c);
        if (theta2 > 1e-8) { const double den = 2.0 * theta * s; cq = fabs(den) > 1e-12 ? 1.0 / theta2 - (1.0 + c) / den : 0.0; }
        const double ti = 1.0 / theta;
        const double k0 = w0 * ti, k1 = w1 * ti, k2 = w2 * ti;
        const double oc = 1.0 - c;
        R[0] = c + k0 * k0 * oc;       R[1] = k0 * k1 * oc - k2 * s;  R[2] = k0 * k2 * oc + k1 * s;
        R[3] = k0 * k1 * oc + k2 * s;  R[4] = c + k1 * k1 * oc;       R[5] = k1 * k2 * oc - k0 * s;
        R[6] = k0 * k2 * oc - k1 * s;  R[7] = k1 * k2 * oc + k0 * s;  R[8] = c + k2 * k2 * oc;
        // K' = (w w^T + (R^T - I) [w]x) / theta^2,  [w]x = [[0,-w2,w1],[w2,0,-w0],[-w1,w0,0]]
        const double Wx[9] = { 0.0, -w2, w1, w2, 0.0, -w0, -w1, w0, 0.0 };
        const double w[3] = { w0, w1, w2 };
        const double it2 = 1.0 / theta2;
        for (int r = 0; r < 3; ++r)
            for (int cc = 0; cc < 3; ++cc) {
                double acc = w[r] * w[cc];
                for (int m = 0; m < 3; ++m) {
                    const double rtmi = R[m * 3 + r] - (m == r ? 1.0 : 0.0);   // (R^T - I)[r][m]
                    acc += rtmi * Wx[m * 3 + cc];
                }
                K[r * 3 + cc] = acc * it2;
            }
    } else {
        R[0] = 1.0;  R[1] = -w2;  R[2] = w1;
        R[3] = w2;   R[4] = 1.0;  R[5] = -w0;
        R[6] = -w1;  R[7] = w0;   R[8] = 1.0;
        for (int e = 0; e < 9; ++e) K[e] = (e % 4 == 0) ? 1.0 : 0.0;
        small = 1.0;
    }
    for (int e = 0; e < 9; ++e) { ct[CT_R + e] = R[e]; ct[CT_K + e] = K[e]; }
    ct[CT_T + 0] = cam[3]; ct[CT_T + 1] = cam[4]; ct[CT_T + 2] = cam[5];
    ct[CT_SMALL] = small;
    for (int e = 0; e < 6; ++e) ct[CT_SCALE + e] = scale6 ? scale6[e] : 1.0;
    for (int c = 0; c < 3; ++c)
        for (int a = 0; a < 3; ++a) {
            double q = (a == c) ? 1.0 : 0.0;
            if (small == 0.0) q = R[3 * c + 0] * K[0 + a] + R[3 * c + 1] * K[3 + a] + R[3 * c + 2] * K[6 + a];
            ct[CT_QD + 3 * c + a] = q;
        }
    for (int e = CT_QD + 9; e < CT_STRIDE; ++e) ct[e] = 0.0;
    // coefficient of [w]x^2 in Jr(w)^-1 = I + [w]x / 2 + cq [w]x^2 (gauge vectors, k_finalize): 1 / theta^2 - (1 + cos) / (2 theta sin), 1 / 12
    // in the limit; formed here, where sin and cos of theta are at hand anyway (theta near pi: the term is dropped, any vector will do)
    ct[CT_CQ] = cq;
}

// [R | t] of a camera from its six parameters: the rotation part of make_cam_table (same expressions, same branches), for a pass that would
// rather rebuild R per observation than gather nine more doubles of it (k_point_update's trial sweep: the pass is bound by its gather instructions)
__device__ __forceinline__ void pose_from_params(const double (&cam)[6], double (&RT)[12]) {
    const double w0 = cam[0], w1 = cam[1], w2 = cam[2];
    const double theta2 = w0 * w0 + w1 * w1 + w2 * w2;
    if (theta2 > DBL_EPSILON) {
        const double theta = sqrt(theta2);
        double s, c;
        sincos(theta, &s, &c);
        const double ti = 1.0 / theta;
        const double k0 = w0 * ti, k1 = w1 * ti, k2 = w2 * ti;
        const double oc = 1.0 - c;
        RT[0] = c + k0 * k0 * oc;       RT[1] = k0 * k1 * oc - k2 * s;  RT[2] = k0 * k2 * oc + k1 * s;
        RT[3] = k0 * k1 * oc + k2 * s;  RT[4] = c + k1 * k1 * oc;       RT[5] = k1 * k2 * oc - k0 * s;
        RT[6] = k0 * k2 * oc - k1 * s;  RT[7] = k1 * k2 * oc + k0 * s;  RT[8] = c + k2 * k2 * oc;
    } else {
        RT[0] = 1.0;  RT[1] = -w2;  RT[2] = w1;
        RT[3] = w2;   RT[4] = 1.0;  RT[5] = -w0;
        RT[6] = -w1;  RT[7] = w0;   RT[8] = 1.0;
    }
    RT[9] = cam[3]; RT[10] = cam[4]; RT[11] = cam[5];
}

__global__ void k_cam_setup(int ncam, const double* __restrict__ cam, const double* __restrict__ cscale, double* __restrict__ camtab) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ncam) return;
    double c6[6], ct[CT_STRIDE];
    for (int e = 0; e < 6; ++e) c6[e] = cam[6 * j + e];
    make_cam_table(c6, cscale + 6 * j, ct);
    for (int e = 0; e < CT_STRIDE; ++e) camtab[cam_tab_index(e, j, ncam)] = ct[e];
}

template <typename T>
void launch_cam_setup(hipStream_t s, const DeviceStructure& ds, const DeviceBuffers& db, int which) {
    hipLaunchKernelGGL(k_cam_setup, dim3((ds.ncam + 63) / 64), dim3(64), 0, s, ds.ncam, db.cam[which], db.cscale, db.camtab[which]);
}
template void launch_cam_setup<float>(hipStream_t, const DeviceStructure&, const DeviceBuffers&, int);
template void launch_cam_setup<double>(hipStream_t, const DeviceStructure&, const DeviceBuffers&, int);

// First launch of a solve: the LM state arrives as a kernel argument (no H2D copy); the same launch clears the linear-solver
// status word, the Jacobi scales, the column-norm accumulators and the slotted accumulators and builds the camera tables
// (instead of a copy, two memsets, a fill kernel, k_cam_setup and k_iter0).
// cam_src / pts_src non-null: a pending sfmba_problem_reset() -- the initial parameters are copied into the current buffers by this
// launch (instead of two device-to-device copies and a state upload enqueued ahead of it: three more host calls in front of a solve
// whose first kernels are all a few microseconds long)
__global__ void k_begin(LMState st, DeviceStructure ds, DeviceBuffers db, const double* __restrict__ cam_src, const double* __restrict__ pts_src) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e == 0) { *db.st = st; *db.lin_info = 0; *db.fin_counter = 0; }
    if (e < 6 * ds.ncam) db.cscale[e] = 1.0;
    if (e < ds.ld) db.udiag[e] = 0.0;
    for (int k = e; k < db.nslot * SLOT_W; k += gridDim.x * blockDim.x) db.slots[k] = 0.0;
    if (pts_src) { for (size_t k = e; k < (size_t)3 * ds.npt; k += (size_t)gridDim.x * blockDim.x) db.pts[st.cur][k] = pts_src[k]; }
    if (e < ds.ncam) {
        double c6[6], ct[CT_STRIDE];
        for (int k = 0; k < 6; ++k) c6[k] = cam_src ? cam_src[6 * e + k] : db.cam[st.cur][6 * e + k];
        if (cam_src) { for (int k = 0; k < 6; ++k) db.cam[st.cur][6 * e + k] = c6[k]; }
        make_cam_table(c6, nullptr, ct);
        for (int k = 0; k < CT_STRIDE; ++k) db.camtab[st.cur][cam_tab_index(k, e, ds.ncam)] = ct[k];
    }
}
void launch_begin(hipStream_t s, const DeviceStructure& ds, const DeviceBuffers& db, const LMState& st, const double* cam_src, const double* pts_src) {
    int nb = std::max(std::max(6 * ds.ncam, ds.ld), NSLOT * SLOT_W);      // (a larger slot array is cleared by a strided loop)
    if (pts_src) nb = std::max(nb, std::min(3 * ds.npt, 1 << 20));         // one point coordinate per thread up to 4096 workgroups
    hipLaunchKernelGGL(k_begin, dim3((nb + 255) / 256), dim3(256), 0, s, st, ds, db, cam_src, pts_src);
}

// ||x||^2 of the current parameters -> acc[ACC_XNEW2]
__global__ void k_xnorm(DeviceStructure ds, DeviceBuffers db) {
    __shared__ double scratch[BLK / 64];
    const int cur = db.st->cur;
    const double* cam = db.cam[cur];
    const double* pts = db.pts[cur];
    const int nc = 6 * ds.ncam, np = 3 * ds.npt;
    double s = 0.0;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < nc + np; e += gridDim.x * blockDim.x) {
        const double v = e < nc ? cam[e] : pts[(size_t)3 * ds.pt_base + (e - nc)];
        s += (e < nc ? db.shared_weight : 1.0) * v * v;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { const double f = db.st->focal[cur]; s += db.shared_weight * f * f; }
    s = block_sum(s, scratch);
    if (threadIdx.x == 0) atomicAdd(slot_ptr(db, ACC_XNEW2), s);
}

void launch_xnorm(hipStream_t s, const DeviceStructure& ds, const DeviceBuffers& db) {
    int blocks = (6 * ds.ncam + 3 * ds.npt + BLK - 1) / BLK;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_xnorm, dim3(blocks), dim3(BLK), 0, s, ds, db);
}

// ------------------------------------------------------------------------------------------
// Jacobi column scaling (iteration 0): s = 1 / (1 + ||J_col||)   [Ceres-upstream EstimateScale]
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(BLK) void k_colnorm_points(DeviceStructure ds, DeviceBuffers db, int jacobi) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ds.npt) return;
    const int cur = db.st->cur;
    const double* tab = db.camtab[cur];
    const double focal = db.st->focal[cur];
    const double X[3] = { db.pts[cur][3 * i], db.pts[cur][3 * i + 1], db.pts[cur][3 * i + 2] };
    double n0 = 0, n1 = 0, n2 = 0;
    if (jacobi) {
        for (int q = ds.pt_ptr[i]; q < ds.pt_ptr[i + 1]; ++q) {
            const CamRow ct = { tab + 4 * (size_t)(ds.obs_cam[q]), ds.ncam };
            const Proj pr = project_point(ct, CT_R, CT_T, X);
            T B[6];
            point_block<T>(ct, pr, focal, B);
            n0 += (double)B[0] * (double)B[0] + (double)B[3] * (double)B[3];
            n1 += (double)B[1] * (double)B[1] + (double)B[4] * (double)B[4];
            n2 += (double)B[2] * (double)B[2] + (double)B[5] * (double)B[5];
        }
        db.pscale[3 * i] = 1.0 / (1.0 + sqrt(n0));
        db.pscale[3 * i + 1] = 1.0 / (1.0 + sqrt(n1));
        db.pscale[3 * i + 2] = 1.0 / (1.0 + sqrt(n2));
    } else {
        db.pscale[3 * i] = db.pscale[3 * i + 1] = db.pscale[3 * i + 2] = 1.0;
    }
}

// squared column norms of the camera / focal columns -> udiag (atomics), one block per chunk
// with_xnorm: the launch also sums ||x||^2 of the current parameters (k_xnorm's job: a launch of its own, 4.6 us per solve, for one strided pass
// over 2.4 MB) -- every workgroup takes a stride of the parameter arrays beside its chunk
template <typename T>
__global__ __launch_bounds__(BLK) void k_colnorm_cams(DeviceStructure ds, DeviceBuffers db, int with_xnorm) {
    __shared__ double scratch[(BLK / 64) * 8];
    const int4 ch = ds.chunks_coarse[ds.coarse_order[blockIdx.x]];
    const int j = ch.x;
    const int cur = db.st->cur;
    const CamRow ct = { db.camtab[cur] + 4 * (size_t)(j), ds.ncam };
    const double focal = db.st->focal[cur];
    const typename ObsXY<T>::type* oxy = reinterpret_cast<const typename ObsXY<T>::type*>(ds.obs_xy);
    (void)oxy;
    double n[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    if (with_xnorm) {
        const double* cam = db.cam[cur];
        const double* pts = db.pts[cur];
        const int nc = 6 * ds.ncam, np = 3 * ds.npt;
        double s = 0.0;
        for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < nc + np; e += gridDim.x * blockDim.x) {
            const double v = e < nc ? cam[e] : pts[(size_t)3 * ds.pt_base + (e - nc)];
            s += (e < nc ? db.shared_weight : 1.0) * v * v;
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) s += db.shared_weight * focal * focal;
        n[7] = s;
    }
    for (int e = ch.y + threadIdx.x; e < ch.z; e += blockDim.x) {
        const int i = ds.cam_obs_pt[e];
        const double X[3] = { db.pts[cur][3 * i], db.pts[cur][3 * i + 1], db.pts[cur][3 * i + 2] };
        const Proj pr = project_point(ct, CT_R, CT_T, X);
        T B[6], A[12];
        point_block<T>(ct, pr, focal, B);
        camera_block<T>(ct, pr, focal, X, B, A);
#pragma unroll
        for (int c = 0; c < 6; ++c) n[c] += (double)A[c] * (double)A[c] + (double)A[6 + c] * (double)A[6 + c];
        n[6] += pr.xp * pr.xp + pr.yp * pr.yp;
    }
    const double tot = block_sums<8>(n, scratch);            // (one barrier instead of eight pairs of them)
    if (threadIdx.x < 7) atomicAdd(threadIdx.x < 6 ? &db.udiag[6 * j + threadIdx.x] : slot_ptr(db, ACC_UDF), tot);
    if (with_xnorm && threadIdx.x == 7) atomicAdd(slot_ptr(db, ACC_XNEW2), tot);
}

__global__ void k_colnorm_finish(DeviceStructure ds, DeviceBuffers db, int jacobi, int finish_xnorm) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.x == 0 && threadIdx.x < 64) {      // focal column: summed over the slots by one wave
        const double n2 = slots_take(db, ACC_UDF);
        if (threadIdx.x == 0) db.st->fscale = jacobi ? 1.0 / (1.0 + sqrt(n2)) : 1.0;
        if (finish_xnorm) {                          // ||x|| of the starting point (k_iter0's job outside a solve)
            const double x2 = slots_take(db, ACC_XNEW2);
            if (threadIdx.x == 0) db.st->x_norm = sqrt(x2);
        }
    }
    if (e >= ds.d - 1) return;
    db.cscale[e] = jacobi ? 1.0 / (1.0 + sqrt(db.udiag[e])) : 1.0;
}

void launch_colnorm_points_only(hipStream_t s, const DeviceStructure& ds, const DeviceBuffers& db, int jacobi, int f32) {
    if (f32) hipLaunchKernelGGL(k_colnorm_points<float>, dim3((ds.npt + BLK - 1) / BLK), dim3(BLK), 0, s, ds, db, jacobi);
    else hipLaunchKernelGGL(k_colnorm_points<double>, dim3((ds.npt + BLK - 1) / BLK), dim3(BLK), 0, s, ds, db, jacobi);
}
void launch_colnorm_cams_only(hipStream_t s, const DeviceStructure& ds, const DeviceBuffers& db, int jacobi, int f32, bool clear_udiag) {
    if (clear_udiag) (void)hipMemsetAsync(db.udiag, 0, sizeof(double) * ds.ld, s);
    if (!jacobi || ds.nchunk_coarse <= 0) return;
    if (f32) hipLaunchKernelGGL(k_colnorm_cams<float>, dim3(ds.nchunk_coarse), dim3(BLK), 0, s, ds, db, 0);
    else hipLaunchKernelGGL(k_colnorm_cams<double>, dim3(ds.nchunk_coarse), dim3(BLK), 0, s, ds, db, 0);
}
void launch_colnorm_finish(hipStream_t s, const DeviceStructure& ds, const DeviceBuffers& db, int jacobi) {
    hipLaunchKernelGGL(k_colnorm_finish, dim3((ds.d + 255) / 256), dim3(256), 0, s, ds, db, jacobi, 0);
}

// with_xnorm: ||x||^2 is summed by the camera pass itself (launch_colnorm_sums_xnorm says whether that pass runs: else launch_xnorm first)
template <typename T>
void launch_colnorm(hipStream_t s, const DeviceStructure& ds, const DeviceBuffers& db, int jacobi, bool clear_udiag, bool points, bool finish_xnorm, bool with_xnorm) {
    if (clear_udiag) (void)hipMemsetAsync(db.udiag, 0, sizeof(double) * ds.ld, s);
    if (points) hipLaunchKernelGGL(k_colnorm_points<T>, dim3((ds.npt + BLK - 1) / BLK), dim3(BLK), 0, s, ds, db, jacobi);
    if (jacobi && ds.nchunk_coarse > 0) hipLaunchKernelGGL(k_colnorm_cams<T>, dim3(ds.nchunk_coarse), dim3(BLK), 0, s, ds, db, with_xnorm ? 1 : 0);
    hipLaunchKernelGGL(k_colnorm_finish, dim3((ds.d + 255) / 256), dim3(256), 0, s, ds, db, jacobi, finish_xnorm ? 1 : 0);
}
template void launch_colnorm<float>(hipStream_t, const DeviceStructure&, const DeviceBuffers&, int, bool, bool, bool, bool);
template void launch_colnorm<double>(hipStream_t, const DeviceStructure&, const DeviceBuffers&, int, bool, bool, bool, bool);

// ------------------------------------------------------------------------------------------
// 3x3 SPD: L^-1 (lower, 6 values l00 l10 l11 l20 l21 l22 of the INVERSE factor). Returns false if not PD.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ bool chol3_inverse(const double V[6] /* v00 v10 v11 v20 v21 v22 */, double Li[6]) {
    // L^-1 directly from reciprocal square roots of the pivots (no sqrt / divide chains)
    const double i00 = fast_rsq(V[0]);
    const double l10 = V[1] * i00, l20 = V[3] * i00;
    const double d1 = V[2] - l10 * l10;
    const double i11 = fast_rsq(d1);
    const double l21 = (V[4] - l20 * l10) * i11;
    const double d2 = V[5] - l20 * l20 - l21 * l21;
    const double i22 = fast_rsq(d2);
    Li[0] = i00;
    Li[1] = -l10 * i00 * i11;
    Li[2] = i11;
    Li[4] = -l21 * i11 * i22;
    Li[3] = -(l20 * i00 + l21 * Li[1]) * i22;
    Li[5] = i22;
    return (V[0] > 0.0) && (d1 > 0.0) && (d2 > 0.0);
}

// ------------------------------------------------------------------------------------------
// The blocks of one observation in registers (obs_record, sfmba_device.h): 16 values
//                 [0..5] A_w = Aproj G (2x3, unscaled)  [6] fz = f/pz  [7] xp  [8] yp
//                 [9..14] C = B~ L^-T (2x3)             [15] unused
// The whitened Schur block of two observations a, b of one point is
//   Y_a Y_b^T = S_a A_a^T (C_a C_b^T) A_b S_b,   A = [A_w | Aproj],  S = Jacobi scale of the camera.
// ------------------------------------------------------------------------------------------
// unscaled camera block A (2x6, row-major) from a record
template <typename T>
__device__ __forceinline__ void rec_camera_block(const T rec[YREC], T A[12]) {
    A[0] = rec[0]; A[1] = rec[1]; A[2] = rec[2];
    A[3] = rec[6]; A[4] = (T)0; A[5] = -rec[6] * rec[7];
    A[6] = rec[3]; A[7] = rec[4]; A[8] = rec[5];
    A[9] = (T)0; A[10] = rec[6]; A[11] = -rec[6] * rec[8];
}

// Per point, once its sums over the observations are known: Jacobi scales, LM damping, the inverse Cholesky factor of V + D^2, t, y_f, M and
// the table entry every other pass re-evaluates from.  `store`: this lane writes the point's results (lanes that share a point compute the
// same values; one of them stores and contributes to the block sums).
template <typename T>
__device__ __forceinline__ void point_finish(const DeviceBuffers& db, const LMState* st, size_t i, int ps_mode, const double (&Xl)[3], double (&V)[6], double (&bp)[3],
                                             double (&Ef)[3], double (&sp)[3], bool store, double& gmax, double& sff, double& rhsf, double& bad) {
    const double radius = st->radius;
    // Jacobi scales of the point's three columns: loaded, or -- first linearisation of a solve -- formed here from the
    // column norms this lane has just summed (s = 1 / (1 + ||J_col||), [Ceres-upstream] EstimateScale)
    if (ps_mode != 0) {
        sp[0] = ps_mode == 1 ? 1.0 / (1.0 + sqrt(V[0])) : 1.0;
        sp[1] = ps_mode == 1 ? 1.0 / (1.0 + sqrt(V[2])) : 1.0;
        sp[2] = ps_mode == 1 ? 1.0 / (1.0 + sqrt(V[5])) : 1.0;
        if (store) { db.pscale[3 * i] = sp[0]; db.pscale[3 * i + 1] = sp[1]; db.pscale[3 * i + 2] = sp[2]; }
    }
    if (store) {
#pragma unroll
        for (int c = 0; c < 3; ++c) gmax = fmax(gmax, fabs(bp[c]));           // gradient of the unscaled problem
    }
    V[0] *= sp[0] * sp[0]; V[1] *= sp[1] * sp[0]; V[2] *= sp[1] * sp[1];
    V[3] *= sp[2] * sp[0]; V[4] *= sp[2] * sp[1]; V[5] *= sp[2] * sp[2];
#pragma unroll
    for (int c = 0; c < 3; ++c) { bp[c] *= sp[c]; Ef[c] *= sp[c]; }
    // LM damping D^2 = clamp(diag(J~^T J~)) / radius   [LevenbergMarquardtStrategy::ComputeStep]
    V[0] += fmin(fmax(V[0], st->min_diag), st->max_diag) / radius;
    V[2] += fmin(fmax(V[2], st->min_diag), st->max_diag) / radius;
    V[5] += fmin(fmax(V[5], st->min_diag), st->max_diag) / radius;
    double Li[6];
    const bool pd = chol3_inverse(V, Li);
    const double t0 = Li[0] * bp[0];
    const double t1 = Li[1] * bp[0] + Li[2] * bp[1];
    const double t2 = Li[3] * bp[0] + Li[4] * bp[1] + Li[5] * bp[2];
    const double y0 = Li[0] * Ef[0];
    const double y1 = Li[1] * Ef[0] + Li[2] * Ef[1];
    const double y2 = Li[3] * Ef[0] + Li[4] * Ef[1] + Li[5] * Ef[2];
    if (!store) return;
    db.pt_t[3 * i] = t0; db.pt_t[3 * i + 1] = t1; db.pt_t[3 * i + 2] = t2;
    // M = diag(s_p) L^-T for the back-substitution (k_point_update): dX = M (t - sum C^T u)
    db.pt_M[6 * i] = sp[0] * Li[0]; db.pt_M[6 * i + 1] = sp[0] * Li[1]; db.pt_M[6 * i + 2] = sp[0] * Li[3];
    db.pt_M[6 * i + 3] = sp[1] * Li[2]; db.pt_M[6 * i + 4] = sp[1] * Li[4]; db.pt_M[6 * i + 5] = sp[2] * Li[5];
    sff -= y0 * y0 + y1 * y1 + y2 * y2;
    rhsf -= y0 * t0 + y1 * t1 + y2 * t2;
    if (!pd || !finite_d(t0 + t1 + t2 + y0 + y1 + y2)) bad = 1.0;
    // the per-point table (sfmba_device.h): the point itself; L^-1 with the point scales folded in, so that C = B~ L^-T =
    // B diag(s) L^-T comes from the UNSCALED point block of an observation; t and y_f in the precision of the Jacobian blocks
    PtRecA<T> ra;
    ra.X[0] = Xl[0]; ra.X[1] = Xl[1]; ra.X[2] = Xl[2];
    ra.L[0] = (T)(Li[0] * sp[0]); ra.L[1] = (T)(Li[1] * sp[0]); ra.L[2] = (T)(Li[2] * sp[1]);
    ra.L[3] = (T)(Li[3] * sp[0]); ra.L[4] = (T)(Li[4] * sp[1]); ra.L[5] = (T)(Li[5] * sp[2]);
    if (sizeof(T) == 8) reinterpret_cast<double*>(&ra)[9] = 0.0;
    reinterpret_cast<PtRecA<T>*>(db.PA)[i] = ra;
    PtRecB<T> rb;
    rb.t[0] = (T)t0; rb.t[1] = (T)t1; rb.t[2] = (T)t2;
    rb.yf[0] = (T)y0; rb.yf[1] = (T)y1; rb.yf[2] = (T)y2;
    reinterpret_cast<PtRecB<T>*>(db.PB)[i] = rb;
}

// K1: point pass.  The point pass leaves NOTHING per observation behind (rounds 1 - 3 wrote a 64-byte record per observation for the
// back-substitution: 64 MB written and read per LM iteration at BASELINE config 3): per point the table entry the reduced-system passes
// and the back-substitution re-evaluate from (PtRecA / PtRecB), t, M.  LPP = 4 LANES PER POINT: a wave owns 16 points, the four lanes of
// a point take its observations in turn (4 at a time) and keep the point's sums in registers; one quad reduction (DPP) and the
// per-point arithmetic on every lane of the quad.  (Rounds 1 - 3 and the first half of round 4 gave every observation a lane and every
// wave 64 consecutive observations: of its ~750 wave instructions ~360 were a serial row loop through LDS and ~130 the per-point phase,
// both at a tenth of the lanes -- 34.5 against 22.7 us at BASELINE config 3, 150 against 98 at config 5.)  Waves take points in the order
// of ds.pt_order (sorted by number of rounds of four observations: the quads of a wave then loop alike whatever the track lengths).  A
// lane needs the camera's R and t only (three component quads of the table: K' is the reduced-system passes' business).
template <typename T>
__global__ PB_BOUNDS void k_point_build(DeviceStructure ds, DeviceBuffers db, int ps_mode_flags) {
    if ((ps_mode_flags & 4) && (db.st->termination != -1 || db.st->retry != 0)) return;
    const int ps_mode = ps_mode_flags & 3;
    __shared__ double scratch[WPB * 4];
    const LMState* st = db.st;
    const int cur = st->cur;
    const double* tab = db.camtab[cur];
    const double focal = st->focal[cur];
    const T fscale = (T)st->fscale;
    const double* pts = db.pts[cur];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int gw = blockIdx.x * WPB + w;
    double lin_cost = 0.0, sff = 0.0, rhsf = 0.0, gmax = 0.0, bad = 0.0;
    // clear what k_cam_diag_f (and the duplicate-pair pass) accumulate with atomics: per camera the 6x6 diagonal block,
    // its focal column, the undamped diagonal, the scaled gradient and the reduced right-hand side
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < ds.ncam * 60; e += gridDim.x * blockDim.x) {
        const int j = e / 60, k = e - 60 * j, row0 = 6 * j;
        if (k < 36) db.S[(size_t)(row0 + k / 6) * ds.ld + row0 + k % 6] = 0.0;
        else if (k < 42) db.udiag[row0 + k - 36] = 0.0;
        else if (k < 48) db.bc[row0 + k - 42] = 0.0;
        else if (k < 54) db.rhs[row0 + k - 48] = 0.0;
        else db.S[(size_t)(row0 + k - 54) * ds.ld + ds.d - 1] = 0.0;
    }
    const int sub = lane & (PB_LPP - 1);
    const int slot = gw * (64 / PB_LPP) + (lane / PB_LPP);
    const bool have = slot < ds.npt;
    const int ip = have ? (ds.pt_order ? ds.pt_order[slot] : slot) : 0;
    const size_t i = (size_t)ip;
    const int q0 = have ? ds.pt_ptr[ip] : 0, q1 = have ? ds.pt_ptr[ip + 1] : 0;
    double Xl[3] = { pts[3 * i], pts[3 * i + 1], pts[3 * i + 2] };
    double sp[3] = { 1.0, 1.0, 1.0 };
    if (ps_mode == 0) { sp[0] = db.pscale[3 * i]; sp[1] = db.pscale[3 * i + 1]; sp[2] = db.pscale[3 * i + 2]; }
    T Va[6] = { (T)0, (T)0, (T)0, (T)0, (T)0, (T)0 }, Ea[3] = { (T)0, (T)0, (T)0 };
    double bp[3] = { 0, 0, 0 };
    // the camera and the coordinates of the NEXT round's observation are fetched one round ahead: a round costs one dependent memory
    // level (the camera's table row)
    int q = q0 + sub;
    int j_next = q < q1 ? ds.obs_cam[q] : 0;
    double ox_next = 0.0, oy_next = 0.0;
    if (q < q1) load_obs<T>(ds.obs_xy, q, ox_next, oy_next);
    while (__any(q < q1)) {
        const bool act = q < q1;
        const int j = j_next;
        const double ox = ox_next, oy = oy_next;
        const CamRow ct = { tab + 4 * (size_t)(j), ds.ncam };
        double Rt[12];
#pragma unroll
        for (int e = 0; e < 12; ++e) Rt[e] = ct[CT_R + e];
        q += PB_LPP;
        if (q < q1) { j_next = ds.obs_cam[q]; load_obs<T>(ds.obs_xy, q, ox_next, oy_next); }
        if (act) {
            const Proj pr = project_point(Rt, 0, 9, Xl);
            const double r0 = focal * pr.xp - ox, r1 = focal * pr.yp - oy;
            lin_cost += r0 * r0 + r1 * r1;
            T B[6];
            point_block<T>(Rt, pr, focal, B);
            const T g0 = (T)pr.xp * fscale, g1 = (T)pr.yp * fscale;
            Va[0] += B[0] * B[0] + B[3] * B[3];
            Va[1] += B[1] * B[0] + B[4] * B[3];
            Va[2] += B[1] * B[1] + B[4] * B[4];
            Va[3] += B[2] * B[0] + B[5] * B[3];
            Va[4] += B[2] * B[1] + B[5] * B[4];
            Va[5] += B[2] * B[2] + B[5] * B[5];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                Ea[c] += B[c] * g0 + B[3 + c] * g1;
                bp[c] += (double)B[c] * r0 + (double)B[3 + c] * r1;
            }
        }
    }
    // the point's sums over its quad (every lane of the quad ends up with them)
    double V[6], Ef[3];
#pragma unroll
    for (int c = 0; c < 6; ++c) { double v = (double)Va[c]; v = xlane_add<1>(v); v = xlane_add<2>(v); V[c] = v; }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double v = (double)Ea[c]; v = xlane_add<1>(v); v = xlane_add<2>(v); Ef[c] = v;
        double u = bp[c]; u = xlane_add<1>(u); u = xlane_add<2>(u); bp[c] = u;
    }
    if (have) point_finish<T>(db, st, i, ps_mode, Xl, V, bp, Ef, sp, sub == 0, gmax, sff, rhsf, bad);
    if (!finite_d(lin_cost)) bad = 1.0;
    const double gm = wave_max(gmax);
    if ((threadIdx.x & 63) == 0 && gm > 0.0) atomic_max_nonneg(slot_ptr(db, ACC_GMAX), gm);
    double sums[4] = { lin_cost, sff, rhsf, bad };
    const double tot = block_sums<4>(sums, scratch);
    if (threadIdx.x < 4) {
        const int which = threadIdx.x == 0 ? ACC_LIN_COST : threadIdx.x == 1 ? ACC_SFF : threadIdx.x == 2 ? ACC_RHSF : ACC_BAD_LIN;
        if (threadIdx.x < 3 || tot != 0.0) atomicAdd(slot_ptr(db, which), tot);
    }
}

// ------------------------------------------------------------------------------------------
// K2a: reduced-system pass over camera pairs.  One wave per 6x6 block (ja < jb): the pairs of observations (one of camera ja,
// one of camera jb, same point) were listed once at build time (structure_build.hip: per pair its POINT, grouped by block),
// so the block is the plain sum
//   -S_a [ sum_pairs A_a^T (C_a C_b^T) A_b ] S_b
// -- no atomics, every block written exactly once per iteration (empty blocks are written as zero:
// the in-place Cholesky destroyed the previous contents).  Workgroups are grouped so that all
// blocks of one block-row run on one XCD (blockIdx % 8).  Nothing is stored per observation (rounds 1 / 2 gathered two 64-byte
// records per pair: 19x the algorithmic traffic): both observations of a pair are RE-EVALUATED from the two camera rows (scalar
// registers) and one 64-byte point-table entry, in the FACTORED form of sfmba_device.h (obs_factored).
// ------------------------------------------------------------------------------------------
// Sum of N per-lane values over the lanes that differ in the bits OFF, OFF/2, ..., 1: at every level a lane keeps one half of
// the values and sends the other half to its partner, so the whole reduction moves N/2 + N/4 + ... values instead of N per level.
// On return v[0 .. len) of this lane are the sums of values base .. base + len - 1 (len may be <= 0).
// VALU64: fp64 values exchanged in the VALU as well (two 32-bit halves per value: DPP / permlane swaps) -- the pair pass's block sums
template <typename V, int N, int OFF, bool VALU64 = false>
struct HalvingReduceT {
    static __device__ __forceinline__ void run(V* v, int lane, int& base, int& len) {
        constexpr int H = (N + 1) / 2;
        const bool up = (lane & OFF) != 0;
#pragma unroll
        for (int k = 0; k < H; ++k) {
            const V lo = v[k];
            const V hi = (H + k < N) ? v[H + k] : (V)0;
            if constexpr (sizeof(V) == 8 && !VALU64) {
                // fp64 (camera pass: 24 doubles): through the LDS pipe as before.  The permlane / DPP forms cost that pass its occupancy
                // (62 -> 144 registers: two registers in and two out per swapped word), and it is bound by loads in flight, not by issue.
                const V send = up ? lo : hi;
                const V keep = up ? hi : lo;
                v[k] = keep + __shfl_xor(send, OFF, 64);
            } else if constexpr (OFF >= 16) {
                v[k] = xlane_pairsum<OFF>(lo, hi);                           // the swap does the selects (sfmba_device.h)
            } else {
                const V send = up ? lo : hi;
                const V keep = up ? hi : lo;
                v[k] = keep + xlane_get<OFF>(send);                          // partner: lane ^ OFF (lane ^ 7 at OFF = 4)
            }
        }
        base += up ? H : 0;
        len = up ? len - H : (len < H ? len : H);
        HalvingReduceT<V, H, OFF / 2, VALU64>::run(v, lane, base, len);
    }
};
template <typename V, int N, bool VALU64>
struct HalvingReduceT<V, N, 0, VALU64> {
    static __device__ __forceinline__ void run(V*, int, int&, int&) {}
};
// The block sums of the pair passes: the lane's own sums (at most eight pair products each in the wave-per-chunk pass) are WIDENED to fp64
// before the first cross-lane step, so that everything summed across lanes, chunks and ranks is summed in fp64 ("fp32 Jacobian blocks,
// fp64 accumulation", BASELINE config 3).  The first halving level converts on the fly: only N / 2 doubles are ever live.
//   SFMBA_PAIR_ACC = 0: the round-4 form (lane sums AND butterfly in T; fp64 from the chunk boundary on)
//                    1: lane sums in T, butterfly in fp64 (default)
//                    2: fp64 lane accumulators as well (every pair product widened before it is added)
#ifndef SFMBA_PAIR_ACC
#define SFMBA_PAIR_ACC 1
#endif
template <typename T> struct PairAccSel { typedef T type; };
#if SFMBA_PAIR_ACC == 2
template <> struct PairAccSel<float> { typedef double type; };
#endif
template <typename T> using PairAcc = typename PairAccSel<T>::type;
template <typename T, int N, int OFF>
__device__ __forceinline__ double pair_block_reduce(T (&acc)[N], int lane, int& base, int& len) {
    if constexpr (sizeof(T) == 8 || SFMBA_PAIR_ACC == 0) {
        HalvingReduceT<T, N, OFF, SFMBA_PAIR_ACC == 2>::run(acc, lane, base, len);
        return len >= 1 ? (double)acc[0] : 0.0;
    } else {
        constexpr int H = (N + 1) / 2;
        const bool up = (lane & OFF) != 0;
        double w[H];
#pragma unroll
        for (int k = 0; k < H; ++k) {
            const double lo = (double)acc[k];
            const double hi = (H + k < N) ? (double)acc[H + k] : 0.0;
            if constexpr (OFF >= 16) w[k] = xlane_pairsum<OFF>(lo, hi);
            else { const double send = up ? lo : hi, keep = up ? hi : lo; w[k] = keep + xlane_get<OFF>(send); }
        }
        base += up ? H : 0;
        len = up ? len - H : (len < H ? len : H);
        HalvingReduceT<double, H, OFF / 2, true>::run(w, lane, base, len);
        return len >= 1 ? w[0] : 0.0;
    }
}
template <int N, int OFF> using HalvingReduce = HalvingReduceT<double, N, OFF>;

// the same for a lane GROUP of LPB lanes (sixteen-lane pair pass): the group's lanes end up with up to three sums each (own[0 .. len))
template <typename T, int LPB>
__device__ __forceinline__ void pair_group_reduce(T (&acc)[36], int lane, int& base, int& len, double (&own)[3]) {
    if constexpr (sizeof(T) == 8 || SFMBA_PAIR_ACC == 0) {
        HalvingReduceT<T, 36, LPB / 2, SFMBA_PAIR_ACC == 2>::run(acc, lane, base, len);
#pragma unroll
        for (int k = 0; k < 3; ++k) own[k] = (double)acc[k];
    } else {
        constexpr int OFF = LPB / 2, H = 18;
        const bool up = (lane & OFF) != 0;
        double w[H];
#pragma unroll
        for (int k = 0; k < H; ++k) {
            const double lo = (double)acc[k], hi = (double)acc[H + k];
            const double send = up ? lo : hi, keep = up ? hi : lo;
            w[k] = keep + xlane_get<OFF>(send);
        }
        base += up ? H : 0;
        len = up ? len - H : (len < H ? len : H);
        HalvingReduceT<double, H, OFF / 2, true>::run(w, lane, base, len);
#pragma unroll
        for (int k = 0; k < 3; ++k) own[k] = w[k];
    }
}

// one pair of observations: acc += A_a^T (C_a C_b^T) A_b   (unscaled; the camera scales are applied once at the end)
template <typename T>
__device__ __forceinline__ void pair_product(const T ra[YREC], const T rb[YREC], bool diag, T acc[36]) {
    const T m00 = ra[9] * rb[9] + ra[10] * rb[10] + ra[11] * rb[11];
    const T m01 = ra[9] * rb[12] + ra[10] * rb[13] + ra[11] * rb[14];
    const T m10 = ra[12] * rb[9] + ra[13] * rb[10] + ra[14] * rb[11];
    const T m11 = ra[12] * rb[12] + ra[13] * rb[13] + ra[14] * rb[14];
    T Aa[12], Ab[12], Tm[12];
    rec_camera_block<T>(ra, Aa);
    rec_camera_block<T>(rb, Ab);
#pragma unroll
    for (int c = 0; c < 6; ++c) { Tm[c] = m00 * Ab[c] + m01 * Ab[6 + c]; Tm[6 + c] = m10 * Ab[c] + m11 * Ab[6 + c]; }
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const T v = Aa[r] * Tm[c] + Aa[6 + r] * Tm[6 + c];
            acc[6 * r + c] += v;
            if (diag) acc[6 * c + r] += v;   // same camera twice: Y_a Y_b^T + Y_b Y_a^T
        }
}

// the two-sided transform of one block's sums (factored coordinates, sfmba_device.h): S_IJ = G_I [sum] G_J^T with the per-camera
// G = Lw D E^T that k_finalize (PCG: Lw = Linv, so this IS the preconditioned block) or k_pair_factors (exact solver: Lw = I) left in
// pair_G.  tile: the 36 sums (negated) in LDS; lanes 0..35 write one entry each.
template <int MODE>
__device__ __forceinline__ void pair_epilogue(const DeviceStructure& ds, const DeviceBuffers& db, int b, int2 cj, const double* tile, int lane) {
    if (lane < 36) {
        const int r = lane / 6, c = lane - 6 * r;
        double Gi[6], Gj[6];
#pragma unroll
        for (int a = 0; a < 6; ++a) { Gi[a] = db.pair_G[(size_t)cj.x * 36 + 6 * r + a]; Gj[a] = db.pair_G[(size_t)cj.y * 36 + 6 * c + a]; }
        double v = 0.0;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            double u = 0.0;
#pragma unroll
            for (int bb = 0; bb < 6; ++bb) u += tile[6 * a + bb] * Gj[bb];
            v += Gi[a] * u;
        }
        if (MODE == 0) db.S[(size_t)(6 * cj.x + r) * ds.ld + 6 * cj.y + c] = v;
        else store_block_entry(ds, db, b, cj, r, c, v);
    }
}

// One wave per CHUNK of a block: at most SFMBA_PAIR_CHUNK pairs (eight rounds of 64), so lane-local sums in T never pile up more than eight
// terms and no wave runs longer than eight rounds whatever the co-visibility (structure_build.hip, build_pair_chunks).  The per-camera
// factor diag(R K', I) of the camera blocks is applied once per block in the epilogue (pair_G), the pair loop works on [ -[R X]x | I ], the
// projection Jacobian and C: 295 wave instructions per 64 pairs, 128 registers (four waves per SIMD) in fp32 mode.  A block of several
// chunks leaves its partial sums in pair_partial; k_schur_combine (the next launch) adds them and runs the epilogue.
template <typename T, int MODE>
__global__ __launch_bounds__(64, (sizeof(T) == 4 ? 4 : 2)) void k_schur_pairs(DeviceStructure ds, DeviceBuffers db) {
    __shared__ double tile[36];
    const int lane = threadIdx.x & 63;
    const int4 dsc = ds.pwg_desc[blockIdx.x];       // one load: block, row camera, pair range of the chunk
    const int2 chunk = ds.pwg_chunk[blockIdx.x];    // {row of pair_partial (blocks of several chunks), chunks of the block}
    const int b = dsc.x, pbeg = dsc.z, p1 = dsc.w;
    int2 cj;
    cj.x = dsc.y;
    cj.y = dsc.y + (b - (int)((long long)dsc.y * ds.ncam - (long long)dsc.y * (dsc.y - 1) / 2));
    const int fo = ds.d - 1;
    if (cj.x == cj.y) {
        if (MODE == 1) {
            // k_finalize(pcg = 1) left the post-linearisation bookkeeping (gradient tolerance, cost of iteration 0, failed
            // evaluation) to this launch, which starts after it in stream order: no arrival counter, no fences there
            if (cj.x == 0) post_linearisation(ds, db);
            // glue of the block-Jacobi transform for camera j: S~_jj = I, S~_jf = Linv_j S_jf / sqrt(S_ff), b~_j = Linv_j rhs_j
            const int j = cj.x, row0 = 6 * j;
            const double* Li = db.pcg_binv + (size_t)j * 36;
            const double linv_f = 1.0 / sqrt(db.S[(size_t)fo * ds.ld + fo]);
            if (lane < 36) {
                const int r = lane / 6, c = lane - 6 * r;
                store_F(db, (size_t)(row0 + r) * ds.ld + row0 + c, (r == c) ? 1.0 : 0.0);
            }
            if (lane < 6) {
                double vf = 0.0, vb = 0.0;
                for (int a = 0; a <= lane; ++a) { vf += Li[lane * 6 + a] * db.S[(size_t)(row0 + a) * ds.ld + fo]; vb += Li[lane * 6 + a] * db.rhs[row0 + a]; }
                vf *= linv_f;
                store_F(db, (size_t)(row0 + lane) * ds.ld + fo, vf);
                store_F(db, (size_t)fo * ds.ld + row0 + lane, vf);
                db.pcg_bt[row0 + lane] = vb;
            }
            if (j == 0 && lane == 63) {
                store_F(db, (size_t)fo * ds.ld + fo, 1.0);
                db.pcg_bt[fo] = db.rhs[fo] * linv_f;
                db.pcg_binv[(size_t)ds.ncam * 36] = linv_f;
            }
        }
        return;     // pairs inside diagonal blocks (duplicates) were added by k_schur_dups before k_finalize
    }
    const int s = lane & 3, g = lane >> 2;
    PairAcc<T> acc[36];
#pragma unroll
    for (int e = 0; e < 36; ++e) acc[e] = (PairAcc<T>)0;
    {
        const LMState* st = db.st;
        const int cur = st->cur;
        const double focal = st->focal[cur];
        CamG<T> ca, cb;
        load_cam_g<T>(db.camtab[cur], __builtin_amdgcn_readfirstlane(cj.x), ds.ncam, ca);
        load_cam_g<T>(db.camtab[cur], __builtin_amdgcn_readfirstlane(cj.y), ds.ncam, cb);
        const PtRecA<T>* PA = reinterpret_cast<const PtRecA<T>*>(db.PA);
        // lane (s, g) owns pair p0 + 16 s + g of a round; the point slot of the NEXT round's pair is fetched one round ahead (one
        // dependent memory level per round: the point-table entry)
        const int mine = 16 * s + g;
        int pt_next = ds.pair_pt[pbeg < p1 ? (pbeg + mine < p1 ? pbeg + mine : p1 - 1) : 0];
        for (int p0 = pbeg; p0 < p1; p0 += 64) {
            const PtRecA<T> pa = load_ptrec(PA + pt_next);
            { const int p = p0 + 64 + mine; pt_next = ds.pair_pt[p < p1 ? p : p1 - 1]; }
            T ga[GREC], gb[GREC];
            obs_factored<T>(ca, focal, pa.X, pa.L, ga);
            obs_factored<T>(cb, focal, pa.X, pa.L, gb);
            if (p0 + mine >= p1) ga[3] = (T)0;      // this lane's pair lies beyond the chunk: contribute nothing (N carries f_a / p_z)
            pair_product_factored<T, PairAcc<T>>(ga, gb, acc);
        }
    }
    // Sum of the 36 entries over the 64 lanes by a halving butterfly: afterwards lane `base` -- 36 of the 64 lanes -- owns ONE
    // entry of the 6x6 block.
    int base = 0, len = 36;
    const double total = pair_block_reduce<PairAcc<T>, 36, 32>(acc, lane, base, len);
    if (chunk.y > 1) {                              // one of several chunks: the partial sums of this one
        if (len >= 1) db.pair_partial[(size_t)chunk.x * 36 + base] = -total;
        return;
    }
    if (len >= 1) tile[base] = -total;
    wave_lds_fence();
    pair_epilogue<MODE>(ds, db, b, cj, tile, lane);
}

// the blocks of several chunks: their partial sums added in chunk order, then the epilogue of k_schur_pairs
template <int MODE>
__global__ __launch_bounds__(64) void k_schur_combine(DeviceStructure ds, DeviceBuffers db) {
    __shared__ double tile[36];
    const int lane = threadIdx.x & 63;
    const int slot0 = ds.multi_slots[blockIdx.x];
    const int4 dsc = ds.pwg_desc[slot0];
    const int2 rows = ds.pwg_chunk[slot0];          // {first row of pair_partial, chunks}
    const int nch = rows.y;
    const int b = dsc.x;
    int2 cj;
    cj.x = dsc.y;
    cj.y = dsc.y + (b - (int)((long long)dsc.y * ds.ncam - (long long)dsc.y * (dsc.y - 1) / 2));
    if (lane < 36) {
        double v = 0.0;
        for (int c = 0; c < nch; ++c) v += db.pair_partial[(size_t)(rows.x + c) * 36 + lane];
        tile[lane] = v;
    }
    wave_lds_fence();
    pair_epilogue<MODE>(ds, db, b, cj, tile, lane);
}

// Pairs INSIDE a diagonal block: one camera observing a point twice (two features matched to the same 3D point).  Both observations
// of such a pair have the same camera and the same point, hence the same Jacobian blocks (only their coordinates differ):
// Y_a Y_b^T + Y_b Y_a^T = 2 Y Y^T, one evaluation per pair.  One wave per diagonal block that has pairs (usually none); added to the
// upper part of the block with atomics BEFORE k_finalize damps and factors it.
template <typename T>
__global__ __launch_bounds__(64) void k_schur_dups(DeviceStructure ds, DeviceBuffers db) {
    const int lane = threadIdx.x & 63;
    const int b = ds.dup_blocks[blockIdx.x].x;
    const int j = ds.blk_cams[b].x;
    const int pbeg = ds.blk_ptr[b], p1 = ds.blk_ptr[b + 1];
    const LMState* st = db.st;
    const int cur = st->cur;
    const double focal = st->focal[cur];
    CamRegs ct;
    load_cam_regs(db.camtab[cur], __builtin_amdgcn_readfirstlane(j), ds.ncam, ct);
    const PtRecA<T>* PA = reinterpret_cast<const PtRecA<T>*>(db.PA);
    double total[36];
#pragma unroll
    for (int e = 0; e < 36; ++e) total[e] = 0.0;
    for (int p0 = pbeg; p0 < p1; p0 += 64) {
        const int p = p0 + lane;
        const PtRecA<T> pa = load_ptrec(PA + ds.pair_pt[p < p1 ? p : p1 - 1]);
        T rec[YREC], acc[36];
        obs_record<T>(ct, focal, pa.X, pa.L, rec);
        if (p >= p1) {
#pragma unroll
            for (int e = 9; e < 15; ++e) rec[e] = (T)0;
        }
#pragma unroll
        for (int e = 0; e < 36; ++e) acc[e] = (T)0;
        pair_product<T>(rec, rec, true, acc);
#pragma unroll
        for (int e = 0; e < 36; ++e) total[e] += (double)acc[e];
    }
#pragma unroll
    for (int e = 0; e < 36; ++e) {
        const double v = wave_allsum(total[e]);
        const int r = e / 6, c = e - 6 * r;
        if (lane == e && c >= r) atomicAdd(db.S + (size_t)(6 * j + r) * ds.ld + 6 * j + c, -v * db.cscale[6 * j + r] * db.cscale[6 * j + c]);
    }
}

// The pair pass for SMALL blocks, LPB = 16 lanes per 6x6 block and 64 / LPB blocks per wave -- BASELINE config 5's 500k blocks of ~45
// pairs, every problem with more than ~210 cameras, and every rank of a sharded solve (which owns every block with 1/N of its pairs).
// Nothing is gathered per observation and there is no (qa, qb) pair list: per pair a lane reads the pair's point slot (pair_pt, coalesced
// inside its lane group), ONE 64-byte point-table entry, and re-evaluates both observations (obs_factored).  The blocks of a wave are
// consecutive blocks of ONE block row (build_structure), so the row camera sits in scalar registers for the whole wave; the column
// camera differs per lane group and is held per lane.  The 36 sums are reduced over the LPB lanes of a group by the VALU-only halving
// butterfly (DPP row operations never leave a row of 16 lanes), and the per-camera factors G = Lw D E^T (pair_G) are applied from both
// sides in the epilogue of all NG blocks side by side.
#ifndef SFMBA_SUBF_WPS
#define SFMBA_SUBF_WPS 3
#endif
template <typename T, int MODE, int LPB>
__global__ __launch_bounds__(64, (sizeof(T) == 4 ? SFMBA_SUBF_WPS : 2)) void k_schur_pairs_sub_f(DeviceStructure ds, DeviceBuffers db) {
    static_assert(LPB == 16, "one DPP row per block");
    constexpr int NG = 64 / LPB;
    __shared__ double tile[NG][36];
    const int lane = threadIdx.x & 63;
    if (MODE == 1 && blockIdx.x == 0) post_linearisation(ds, db);        // block (0,0) is in the first workgroup; all 64 lanes here
    const int sub = lane / LPB, li = lane % LPB;
    const int4 dsc = ds.pwg_desc[(size_t)blockIdx.x * NG + sub];         // one load: block, row camera, pair range
    const bool have = dsc.x >= 0;
    const int b = have ? dsc.x : 0;
    int2 cj;
    cj.x = dsc.y;
    cj.y = dsc.y + (b - (int)((long long)dsc.y * ds.ncam - (long long)dsc.y * (dsc.y - 1) / 2));
    const bool diag = cj.x == cj.y;
    const int fo = ds.d - 1;
    if (MODE == 1 && have && diag) {
        // glue of the block-Jacobi transform for camera j (see k_schur_pairs)
        const int j = cj.x, row0 = 6 * j;
        const double* Li = db.pcg_binv + (size_t)j * 36;
        const double linv_f = 1.0 / sqrt(db.S[(size_t)fo * ds.ld + fo]);
        for (int e = li; e < 36; e += LPB) {
            const int r = e / 6, c = e - 6 * r;
            store_F(db, (size_t)(row0 + r) * ds.ld + row0 + c, (r == c) ? 1.0 : 0.0);
        }
        for (int l = li; l < 6; l += LPB) {
            double vf = 0.0, vb = 0.0;
            for (int a = 0; a <= l; ++a) { vf += Li[l * 6 + a] * db.S[(size_t)(row0 + a) * ds.ld + fo]; vb += Li[l * 6 + a] * db.rhs[row0 + a]; }
            vf *= linv_f;
            store_F(db, (size_t)(row0 + l) * ds.ld + fo, vf);
            store_F(db, (size_t)fo * ds.ld + row0 + l, vf);
            db.pcg_bt[row0 + l] = vb;
        }
        if (j == 0 && li == LPB - 1) {
            store_F(db, (size_t)fo * ds.ld + fo, 1.0);
            db.pcg_bt[fo] = db.rhs[fo] * linv_f;
            db.pcg_binv[(size_t)ds.ncam * 36] = linv_f;
        }
    }
    const bool work = have && !diag;
    if (!__any(work)) return;
    const LMState* st = db.st;
    const int cur = st->cur;
    const double focal = st->focal[cur];
    const double* tab = db.camtab[cur];
    // lane group 0 always holds a block of the workgroup's row (the descriptors of a workgroup are filled from the front).  Both rows in
    // the precision of the Jacobian blocks (obs_factored_t): the row camera's wave-uniform, the column camera's per lane.
    T Ra[12], Rb[12];
    {
        const int ja = __builtin_amdgcn_readfirstlane(dsc.y), jb = work ? cj.y : 0;
#pragma unroll
        for (int k = 0; k < 12; ++k) { Ra[k] = to_uniform((T)tab[cam_tab_index(k, ja, ds.ncam)]); Rb[k] = (T)tab[cam_tab_index(k, jb, ds.ncam)]; }
    }
    const bool fo_a = tab[cam_tab_index(CT_SMALL, __builtin_amdgcn_readfirstlane(dsc.y), ds.ncam)] != 0.0;
    const bool fo_b = tab[cam_tab_index(CT_SMALL, work ? cj.y : 0, ds.ncam)] != 0.0;
    const T focal_t = (T)focal;
    const PtRecA<T>* PA = reinterpret_cast<const PtRecA<T>*>(db.PA);
    PairAcc<T> acc[36];
#pragma unroll
    for (int e = 0; e < 36; ++e) acc[e] = (PairAcc<T>)0;
    int p0 = work ? dsc.z : 0;
    const int p1 = work ? dsc.w : 0;
    // the point slot of a round's pair is fetched one round ahead (a round then costs one dependent memory level: the point-table entry);
    // lanes beyond the block's last pair re-read it and contribute nothing
    const bool nonempty = p0 < p1;
    int pt_next = nonempty ? ds.pair_pt[p0 + li < p1 ? p0 + li : p1 - 1] : 0;
    // No branch around the body: a lane group whose block is finished (or empty) keeps evaluating its last pair (point 0 if it never had
    // one) and adds zeros -- a divergent `if` here costs a register copy of all 36 sums per round.
    while (__any(p0 < p1)) {
        const PtRecA<T> pa = load_ptrec(PA + pt_next);
        const bool mine = p0 + li < p1;
        if (nonempty) { const int p = p0 + LPB + li; pt_next = ds.pair_pt[p < p1 ? p : p1 - 1]; }
        T ga[GREC], gb[GREC];
        const T X0 = (T)pa.X[0], X1 = (T)pa.X[1], X2 = (T)pa.X[2];
        obs_factored_t<T>(Ra, fo_a, focal_t, X0, X1, X2, pa.L, ga);
        obs_factored_t<T>(Rb, fo_b, focal_t, X0, X1, X2, pa.L, gb);
        if (!mine) ga[3] = (T)0;                       // (N carries f_a / p_z)
        pair_product_factored<T, PairAcc<T>>(ga, gb, acc);
        p0 += LPB;
    }
    if (!nonempty) {                                   // never had a pair: whatever point 0 gave under these two cameras (0 x inf) is not a sum
#pragma unroll
        for (int e = 0; e < 36; ++e) acc[e] = (PairAcc<T>)0;
    }
    int base = 0, len = 36;
    double own[3];                                     // afterwards lane li owns entries base .. base + len - 1 (len <= 3)
    pair_group_reduce<PairAcc<T>, LPB>(acc, lane, base, len, own);
    if (work) {
#pragma unroll
        for (int k = 0; k < 3; ++k) if (k < len) tile[sub][base + k] = -own[k];
    }
    wave_lds_fence();
    if (work) {
        // S_IJ = G_I [sum] G_J^T (see k_schur_pairs)
        for (int e = li; e < 36; e += LPB) {
            const int r = e / 6, c = e - 6 * r;
            double Gi[6], Gj[6];
#pragma unroll
            for (int a = 0; a < 6; ++a) { Gi[a] = db.pair_G[(size_t)cj.x * 36 + 6 * r + a]; Gj[a] = db.pair_G[(size_t)cj.y * 36 + 6 * c + a]; }
            double v = 0.0;
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                double u = 0.0;
#pragma unroll
                for (int bb = 0; bb < 6; ++bb) u += tile[sub][6 * a + bb] * Gj[bb];
                v += Gi[a] * u;
            }
            if (MODE == 0) db.S[(size_t)(6 * cj.x + r) * ds.ld + 6 * cj.y + c] = v;
            else store_block_entry(ds, db, b, cj, r, c, v);
        }
    }
}

// ------------------------------------------------------------------------------------------
// K2b: camera-diagonal pass.  One lane per observation of the camera (no loop, two dependent memory
// levels), SFMBA_CAM_CHUNK (256) lanes per workgroup = one chunk of one camera.  Each lane forms its 47 terms in T;
// the sums over lanes are carried in fp64: halving butterfly inside the wave, LDS across the waves,
// one atomic per value per workgroup.
//   S_jj += A~^T (I - C C^T) A~   (U_jj minus the self term Y_a Y_a^T), undamped diagonal, S_jf, b_c, rhs
// ------------------------------------------------------------------------------------------
// the 47 terms (ACCUM: added to v, else stored) one observation contributes to its camera's diagonal block, focal column, gradient and right-hand side, from its packed
// record and side values z = {C t (2), C y_f (2), residual (2)} -- shared by the record-gathering and the re-evaluating camera pass
template <typename T, bool ACCUM>
__device__ __forceinline__ void cam_diag_terms(const T (&rec)[YREC], const T (&z)[8], const double* __restrict__ cscale6, T fscale, T (&v)[CD_N]) {
        T A[12];
        rec_camera_block<T>(rec, A);
#pragma unroll
        for (int c = 0; c < 6; ++c) { const T s = (T)cscale6[c]; A[c] *= s; A[6 + c] *= s; }
        const T r0 = z[4], r1 = z[5];
        const T g0 = rec[7] * fscale, g1 = rec[8] * fscale;
        // N = I - C C^T
        const T n00 = (T)1 - (rec[9] * rec[9] + rec[10] * rec[10] + rec[11] * rec[11]);
        const T n01 = -(rec[9] * rec[12] + rec[10] * rec[13] + rec[11] * rec[14]);
        const T n11 = (T)1 - (rec[12] * rec[12] + rec[13] * rec[13] + rec[14] * rec[14]);
        // Y v = A~^T (C v): C t and C y_f come from the side record
        const T ct0 = z[0], ct1 = z[1], cy0 = z[2], cy1 = z[3];
        int u = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            const T p0 = n00 * A[a] + n01 * A[6 + a], p1 = n01 * A[a] + n11 * A[6 + a];
#pragma unroll
            for (int b = a; b < 6; ++b) { const T t_ = p0 * A[b] + p1 * A[6 + b]; if (ACCUM) v[u] += t_; else v[u] = t_; ++u; }
        }
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            const T ar = A[a] * r0 + A[6 + a] * r1;
            { const T t_ = A[a] * A[a] + A[6 + a] * A[6 + a]; if (ACCUM) v[21 + a] += t_; else v[21 + a] = t_; }                               // undamped diagonal
            { const T t_ = (A[a] * g0 + A[6 + a] * g1) - (A[a] * cy0 + A[6 + a] * cy1); if (ACCUM) v[27 + a] += t_; else v[27 + a] = t_; }     // S[j,f]
            { const T t_ = ar; if (ACCUM) v[33 + a] += t_; else v[33 + a] = t_; }                                                              // b_c (scaled gradient)
            { const T t_ = ar - (A[a] * ct0 + A[6 + a] * ct1); if (ACCUM) v[39 + a] += t_; else v[39 + a] = t_; }                              // reduced rhs
        }
        { const T t_ = g0 * g0 + g1 * g1; if (ACCUM) v[45] += t_; else v[45] = t_; }
        { const T t_ = g0 * r0 + g1 * r1; if (ACCUM) v[46] += t_; else v[46] = t_; }
}

// sums of the 47 terms over the workgroup (fp64: halving butterfly inside the wave, LDS across the waves) and one atomic per value
// per workgroup -- or, deterministic mode, the per-chunk slot that k_finalize / k_cd_fold add in chunk order
template <typename T>
__device__ __forceinline__ void cam_diag_finish(const DeviceStructure& ds, const DeviceBuffers& db, int j, int chunk_id, T (&v)[CD_N], double (*red)[CD_N]) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // first halving step on the T values (one 32-bit shuffle each), the rest in fp64
    double acc[CD_N / 2];
    {
        const bool up = (lane & 32) != 0;
#pragma unroll
        for (int k = 0; k < CD_N / 2; ++k) {
            const T lo = v[k], hi = v[CD_N / 2 + k];
            const T send = up ? lo : hi, keep = up ? hi : lo;
            acc[k] = (double)keep + (double)__shfl_xor(send, 32, 64);        // (LDS pipe: see HalvingReduceT, fp64)
        }
    }
    int base = (lane & 32) ? CD_N / 2 : 0, len = CD_N / 2;
    HalvingReduce<CD_N / 2, 16>::run(acc, lane, base, len);
    if (len >= 1) red[w][base] = acc[0];
    __syncthreads();
    if (threadIdx.x < 47) {
        const int k = threadIdx.x;
        double s = 0.0;
#pragma unroll
        for (int ww = 0; ww < CD_BLK / 64; ++ww) s += red[ww][k];
        const int row0 = 6 * j, fo = ds.d - 1;
        if (db.cd_part && k < 45) { db.cd_part[(size_t)chunk_id * 48 + k] = s; return; }      // deterministic mode: k_finalize adds the chunks in order
        if (k < 21) {
            int a = 0, rem = k;
            while (rem >= 6 - a) { rem -= 6 - a; ++a; }
            atomicAdd(&db.S[(size_t)(row0 + a) * ds.ld + row0 + a + rem], s);
        } else if (k < 27) {
            atomicAdd(&db.udiag[row0 + k - 21], s);
        } else if (k < 33) {
            atomicAdd(&db.S[(size_t)(row0 + k - 27) * ds.ld + fo], s);
        } else if (k < 39) {
            atomicAdd(&db.bc[row0 + k - 33], s);
        } else if (k < 45) {
            atomicAdd(&db.rhs[row0 + k - 39], s);
        } else if (k == 45) {
            atomicAdd(slot_ptr(db, ACC_SFF), s);
            atomicAdd(slot_ptr(db, ACC_UDF), s);
        } else {
            atomicAdd(slot_ptr(db, ACC_RHSF), s);
            atomicAdd(slot_ptr(db, ACC_BCF), s);
        }
    }
}

// Nothing is stored per observation: the camera's table row sits in scalar registers (one camera per workgroup), a lane gathers its
// observation's point-table entries (64 + 24 bytes from a table that stays in L2), reads the observation's coordinates from the
// camera-major copy (coalesced), and re-evaluates blocks and residual with the expressions of the point pass (obs_record).
template <typename T>
__global__ __launch_bounds__(CD_BLK) void k_cam_diag_f(DeviceStructure ds, DeviceBuffers db) {
    __shared__ double red[CD_BLK / 64][CD_N];
    const int chunk_id = ds.chunk_order[blockIdx.x];
    const int4 ch = ds.chunks[chunk_id];
    const int j = ch.x;
    const LMState* st = db.st;
    const int cur = st->cur;
    const double focal = st->focal[cur];
    const T fscale = (T)st->fscale;
    CamRegs ct;
    load_cam_regs(db.camtab[cur], j, ds.ncam, ct);
    T v[CD_N];
#pragma unroll
    for (int k = 0; k < CD_N; ++k) v[k] = (T)0;
    const typename ObsXY<T>::type* xy = reinterpret_cast<const typename ObsXY<T>::type*>(ds.cam_obs_xy);
    const PtRecA<T>* PA = reinterpret_cast<const PtRecA<T>*>(db.PA);
    const PtRecB<T>* PB = reinterpret_cast<const PtRecB<T>*>(db.PB);
    // SFMBA_CAM_CHUNK / CD_BLK observations per lane (1 by default).  With several, the loop is software-pipelined: the point slot is fetched
    // two rounds ahead and the point-table entries one round ahead, so that a round's arithmetic runs under the next round's gathers.
    int e = ch.y + threadIdx.x;
    int i_nn = 0;
    PtRecA<T> pa_n = PA[0]; PtRecB<T> pb_n = PB[0];
    typename ObsXY<T>::type oxy_n = xy[e < ch.z ? e : ch.y];
    if (e < ch.z) { const int i0 = ds.cam_obs_pt[e]; pa_n = PA[i0]; pb_n = PB[i0]; }
    if (CD_OBS > 1 && e + CD_BLK < ch.z) i_nn = ds.cam_obs_pt[e + CD_BLK];
#pragma unroll 1
    for (; e < ch.z; e += CD_BLK) {
        const PtRecA<T> pa = pa_n;
        const PtRecB<T> pb = pb_n;
        const typename ObsXY<T>::type oxy = oxy_n;
        if (CD_OBS > 1) {
            if (e + CD_BLK < ch.z) { pa_n = PA[i_nn]; pb_n = PB[i_nn]; oxy_n = xy[e + CD_BLK]; }
            if (e + 2 * CD_BLK < ch.z) i_nn = ds.cam_obs_pt[e + 2 * CD_BLK];
        }
        T rec[YREC], z[8];
        obs_record<T>(ct, focal, pa.X, pa.L, rec);
        // the residual as the point pass forms it: fp64 projection, fp64 subtraction, then rounded to T
        typename ObsXY<T>::type rr;
        { const Proj pr = project_point(ct, CT_R, CT_T, pa.X); rr.x = (T)(focal * pr.xp - (double)oxy.x); rr.y = (T)(focal * pr.yp - (double)oxy.y); }
        // side values: C t, C y_f, residual
        z[0] = rec[9] * pb.t[0] + rec[10] * pb.t[1] + rec[11] * pb.t[2];
        z[1] = rec[12] * pb.t[0] + rec[13] * pb.t[1] + rec[14] * pb.t[2];
        z[2] = rec[9] * pb.yf[0] + rec[10] * pb.yf[1] + rec[11] * pb.yf[2];
        z[3] = rec[12] * pb.yf[0] + rec[13] * pb.yf[1] + rec[14] * pb.yf[2];
        z[4] = rr.x; z[5] = rr.y; z[6] = (T)0; z[7] = (T)0;
        cam_diag_terms<T, (CD_OBS > 1)>(rec, z, db.cscale + 6 * j, fscale, v);
    }
    cam_diag_finish<T>(ds, db, j, chunk_id, v, red);
}



template <typename T>
void launch_point_build(hipStream_t s, const DeviceStructure& ds, const DeviceBuffers& db, int ps_mode) {
    const int per_wg = WPB * (64 / PB_LPP);
    // (at least one workgroup: the launch also clears what the camera pass accumulates -- a row-sharded rank may own no point)
    hipLaunchKernelGGL(k_point_build<T>, dim3(std::max(1, (ds.npt + per_wg - 1) / per_wg)), dim3(PBK), 0, s, ds, db, ps_mode);
}
template void launch_point_build<float>(hipStream_t, const DeviceStructure&, const DeviceBuffers&, int);
template void launch_point_build<double>(hipStream_t, const DeviceStructure&, const DeviceBuffers&, int);

// Per-camera factor of the factored pair pass (sfmba_device.h, obs_factored): G = Lw D E^T, E = diag(Q, I), Q = R K' (I for a camera on
// the first-order branch), D the Jacobi scales, Lw = Linv (PCG: the block-Jacobi transform) or I.  The rotation part of D E^T is in the
// camera table (CT_QD, make_cam_table).  Row-major 6 x 6 per camera: pair_G[36 j + 6 r + c] (entry-major would let k_finalize's lanes
// store side by side, -1 us there, but costs the pair pass's epilogue +5 us: measured).
template <bool HAVE_L>
__device__ __forceinline__ void pair_factor(const DeviceBuffers& db, int j, const double (&Lw)[6][6], const double (&Q9)[9], const double (&cs)[6]) {
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            double v;
            if (c < 3) {            // sum_{a < 3} Lw[r][a] (D E^T)[a][c],  (D E^T)[a][c] = cs[a] Q[c][a]
                v = 0.0;
#pragma unroll
                for (int a = 0; a < 3; ++a) v += (HAVE_L ? Lw[r][a] : (r == a ? 1.0 : 0.0)) * (cs[a] * Q9[3 * c + a]);
            } else {
                v = (HAVE_L ? Lw[r][c] : (r == c ? 1.0 : 0.0)) * cs[c];
            }
            db.pair_G[(size_t)j * 36 + 6 * r + c] = v;
        }
}
__global__ void k_pair_factors(DeviceStructure ds, DeviceBuffers db) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ds.ncam) return;
    const double none[6][6] = {};
    double Q9[9], cs[6];
#pragma unroll
    for (int e = 0; e < 9; ++e) Q9[e] = db.camtab[db.st->cur][cam_tab_index(CT_QD + e, j, ds.ncam)];
#pragma unroll
    for (int e = 0; e < 6; ++e) cs[e] = db.cscale[6 * j + e];       // (not the table's copy: the first linearisation's table predates the scales)
    pair_factor<false>(db, j, none, Q9, cs);
}

// mode 0: off-diagonal blocks of S (upper triangle; exact solver);  mode 1: the same blocks written straight into S~ = Lb^-1 S Lb^-T
// (both triangles, or the exchange buffer of a sharded solve) + the per-camera glue of the block-Jacobi transform;  mode 2: the
// duplicate pairs inside diagonal blocks.  One wave per block, or -- small blocks (ds.pair_lpb == 16) -- 16 lanes per block.
template <typename T>
void launch_schur_pairs(hipStream_t s, const DeviceStructure& ds, const DeviceBuffers& db, int mode) {
    const dim3 grid(ds.npairwg);
    if (mode != 2 && ds.npairwg <= 0) return;      // (a row-sharded rank without a block row)
    if (mode == 2) {
        if (ds.ndupwg > 0) hipLaunchKernelGGL(k_schur_dups<T>, dim3(ds.ndupwg), dim3(64), 0, s, ds, db);
        return;
    }
    if (mode == 0) hipLaunchKernelGGL(k_pair_factors, dim3((ds.ncam + 63) / 64), dim3(64), 0, s, ds, db);      // D E^T per camera (PCG: k_finalize wrote Linv D E^T)
    if (ds.pair_lpb == 16) {
        if (mode == 1) hipLaunchKernelGGL((k_schur_pairs_sub_f<T, 1, 16>), grid, dim3(64), 0, s, ds, db);
        else hipLaunchKernelGGL((k_schur_pairs_sub_f<T, 0, 16>), grid, dim3(64), 0, s, ds, db);
    } else {
        if (mode == 1) hipLaunchKernelGGL((k_schur_pairs<T, 1>), grid, dim3(64), 0, s, ds, db);
        else hipLaunchKernelGGL((k_schur_pairs<T, 0>), grid, dim3(64), 0, s, ds, db);
        if (ds.nmulti > 0) {
            if (mode == 1) hipLaunchKernelGGL(k_schur_combine<1>, dim3(ds.nmulti), dim3(64), 0, s, ds, db);
            else hipLaunchKernelGGL(k_schur_combine<0>, dim3(ds.nmulti), dim3(64), 0, s, ds, db);
        }
    }
}
template void launch_schur_pairs<float>(hipStream_t, const DeviceStructure&, const DeviceBuffers&, int);
template void launch_schur_pairs<double>(hipStream_t, const DeviceStructure&, const DeviceBuffers&, int);

template <typename T>
void launch_cam_diag(hipStream_t s, const DeviceStructure& ds, const DeviceBuffers& db) {
    if (ds.nchunk > 0) hipLaunchKernelGGL(k_cam_diag_f<T>, dim3(ds.nchunk), dim3(CD_BLK), 0, s, ds, db);
}
template void launch_cam_diag<float>(hipStream_t, const DeviceStructure&, const DeviceBuffers&);
template void launch_cam_diag<double>(hipStream_t, const DeviceStructure&, const DeviceBuffers&);

// ------------------------------------------------------------------------------------------
// finalize: damping of the reduced diagonal, camera/focal part of the gradient max-norm, padding
// ------------------------------------------------------------------------------------------
// Rows of camera j in the 8 gauge vectors of the problem, in the unknowns of the block-Jacobi transformed reduced system
// (coarse space of the two-level CG preconditioner, dense_solver.hip).  adjustBundle() holds no block constant
// (BA.cpp:160-164), so the undamped problem does not change under a similarity transform of the scene; in camera
// parameters (p = R X + t), to first order:
//   world translation a   (X -> X + a):        dt = -R a,  dw = 0
//   world rotation phi    (X -> Exp(phi) X):   R -> R Exp(-phi)  =>  dw = -Jr(w)^-1 phi,  dt = 0
//                         Jr^-1 = I + [w]x / 2 + (1/theta^2 - (1 + cos theta) / (2 theta sin theta)) [w]x^2
//   scale s               (X -> s X):          dt = t
// plus the weakly determined focal / depth direction (df = f, dt_z = t_z: a longer lens further away).  Unknowns are
// Jacobi-scaled (x = s x_s) and transformed by the block factor (x~ = Lb^T x_s, Lb^-1 = Li): w~ solves Li^T w~ = w / s.
// Values are rounded to fp32 so that every consumer (LDS copies included) sees the same numbers; any vectors are a valid
// coarse space, they only have to be close to the slow directions.
// (the camera's parameters, the rows of R and the Jacobi scales arrive preloaded: k_finalize issues every global load of a camera
// before its first dependent instruction -- read where they are used they were one more L2 round trip each on a kernel of 200 lanes)
__device__ __forceinline__ void gauge_vectors_pre(const DeviceStructure& ds, const DeviceBuffers& db, int j, const double (&Li)[6][6],
                                                  const double (&cam)[6], const double (&Rm)[9], const double (&cs6)[6]) {
    const double w0 = cam[0], w1 = cam[1], w2 = cam[2];
    const double th2 = w0 * w0 + w1 * w1 + w2 * w2;
    double cq = 1.0 / 12.0;                              // limit of the [w]x^2 coefficient for theta -> 0
    if (th2 > 1e-8) {
        const double th = sqrt(th2);
        double sn, cs;
        sincos(th, &sn, &cs);
        const double den = 2.0 * th * sn;
        cq = fabs(den) > 1e-12 ? 1.0 / th2 - (1.0 + cs) / den : 0.0;     // theta near pi: drop the term, any vector will do
    }
    const double K[3][3] = { { 0.0, -w2, w1 }, { w2, 0.0, -w0 }, { -w1, w0, 0.0 } };
    double Ji[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double k2 = 0.0;
#pragma unroll
            for (int m = 0; m < 3; ++m) k2 += K[r][m] * K[m][c];
            Ji[r][c] = (r == c ? 1.0 : 0.0) + 0.5 * K[r][c] + cq * k2;
        }
    double s6[6], ild[6];       // 1 / Jacobi scale; 1 / Li[r][r]: reciprocals once, not a division per back-substitution step
#pragma unroll
    for (int e = 0; e < 6; ++e) { s6[e] = fast_rcp(cs6[e]); ild[e] = fast_rcp(Li[e][e]); }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        double wv[6] = { 0, 0, 0, 0, 0, 0 };
        if (k < 3) { for (int r = 0; r < 3; ++r) wv[3 + r] = -Rm[3 * r + k]; }
        else if (k < 6) { for (int r = 0; r < 3; ++r) wv[r] = -Ji[r][k - 3]; }
        else if (k == 6) { for (int r = 0; r < 3; ++r) wv[3 + r] = cam[3 + r]; }
        else wv[5] = cam[5];
#pragma unroll
        for (int e = 0; e < 6; ++e) wv[e] *= s6[e];
        // back substitution with the upper triangular Li^T
        double wt[6];
#pragma unroll
        for (int r = 5; r >= 0; --r) {
            double v = wv[r];
#pragma unroll
            for (int t = 5; t > r; --t) v -= Li[t][r] * wt[t];
            wt[r] = v * ild[r];
        }
#pragma unroll
        for (int r = 0; r < 6; ++r) db.pcg_W[(size_t)k * ds.ld + 6 * j + r] = (double)(float)wt[r];
    }
}
__device__ void gauge_vectors(const DeviceStructure& ds, const DeviceBuffers& db, int j, const double (&Li)[6][6]) {
    const int cur = db.st->cur;
    double cam[6], Rm[9], cs6[6];
#pragma unroll
    for (int e = 0; e < 6; ++e) { cam[e] = db.cam[cur][6 * (size_t)j + e]; cs6[e] = db.cscale[6 * j + e]; }
#pragma unroll
    for (int e = 0; e < 9; ++e) Rm[e] = db.camtab[cur][cam_tab_index(CT_R + e, j, ds.ncam)];
    gauge_vectors_pre(ds, db, j, Li, cam, Rm, cs6);
}

// The gauge vectors for a reduced system whose block factors were formed outside k_finalize (sharded solve: the factors come
// from the all-reduced system, dense_pcg_transform): Linv of every camera block from db.pcg_binv, focal row from its last entry.
__global__ __launch_bounds__(64) void k_gauge(DeviceStructure ds, DeviceBuffers db) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < ds.ncam) {
        double Li[6][6];
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = 0; c < 6; ++c) Li[r][c] = db.pcg_binv[(size_t)j * 36 + r * 6 + c];
        gauge_vectors(ds, db, j, Li);
    }
    if (j == 0) {
        const LMState* st = db.st;
        const int fo = ds.d - 1;
        const double lf = 1.0 / db.pcg_binv[(size_t)ds.ncam * 36];        // sqrt(S_ff)
        for (int k = 0; k < 8; ++k) {
            db.pcg_W[(size_t)k * ds.ld + fo] = k == 7 ? (double)(float)(lf * st->focal[st->cur] / st->fscale) : 0.0;
            for (int e = ds.d; e < ds.ld; ++e) db.pcg_W[(size_t)k * ds.ld + e] = 0.0;
        }
    }
}
void launch_gauge(hipStream_t s, const DeviceStructure& ds, const DeviceBuffers& db) {
    if (db.pcg_W) hipLaunchKernelGGL(k_gauge, dim3((ds.ncam + 63) / 64), dim3(64), 0, s, ds, db);
}

// ONE gauge vector (k = 0 .. 7, a per-lane value) of camera j: the body of gauge_vectors_pre for a runtime k -- the vector's entries are
// picked by selects (no dynamically indexed registers), the back substitution with Li^T is the same for every k.
__device__ __forceinline__ void gauge_vector_k(const DeviceStructure& ds, const DeviceBuffers& db, int j, int k, const double (&Li)[6][6], const double (&cam)[6],
                                               const double (&Rm)[9], const double (&cs6)[6], double cq) {
    const double w0 = cam[0], w1 = cam[1], w2 = cam[2];
    const double K[3][3] = { { 0.0, -w2, w1 }, { w2, 0.0, -w0 }, { -w1, w0, 0.0 } };
    double wv[6];
    const int c3 = k - 3;                                     // world rotation: column k - 3 of -Jr^-1
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        double k2c[3], ji[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            k2c[c] = 0.0;
#pragma unroll
            for (int m = 0; m < 3; ++m) k2c[c] += K[r][m] * K[m][c];
            ji[c] = (r == c ? 1.0 : 0.0) + 0.5 * K[r][c] + cq * k2c[c];
        }
        // (0 / 1 weights instead of select chains: the compiler turns a chain of selects over array elements into a dynamically indexed
        // array -- in scratch memory)
        const double jsel = (c3 == 0 ? 1.0 : 0.0) * ji[0] + (c3 == 1 ? 1.0 : 0.0) * ji[1] + (c3 == 2 ? 1.0 : 0.0) * ji[2];
        wv[r] = -jsel;
        const double rsel = (k == 0 ? 1.0 : 0.0) * Rm[3 * r] + (k == 1 ? 1.0 : 0.0) * Rm[3 * r + 1] + (k == 2 ? 1.0 : 0.0) * Rm[3 * r + 2];
        wv[3 + r] = -rsel + (k == 6 ? 1.0 : 0.0) * cam[3 + r] + ((k == 7 && r == 2) ? 1.0 : 0.0) * cam[5];
    }
    double wt[6];
#pragma unroll
    for (int e = 0; e < 6; ++e) wv[e] *= fast_rcp(cs6[e]);
#pragma unroll
    for (int r = 5; r >= 0; --r) {                            // back substitution with the upper triangular Li^T
        double v = wv[r];
#pragma unroll
        for (int t = 5; t > r; --t) v -= Li[t][r] * wt[t];
        wt[r] = v * fast_rcp(Li[r][r]);
    }
#pragma unroll
    for (int r = 0; r < 6; ++r) db.pcg_W[(size_t)k * ds.ld + 6 * j + r] = (double)(float)wt[r];
}

// damping of the reduced diagonal, camera/focal part of the gradient max-norm, padding; in PCG mode also Linv of every damped 6x6
// diagonal block (the block-Jacobi preconditioner), the pair pass's per-camera factor and the gauge vectors.  EIGHT LANES PER CAMERA
// (round 4; one lane per camera made this kernel a 1000-deep dependent fp64 chain on 200 lanes: 14 us): every lane of a camera's group
// loads the block and factors it (the same instructions, no divergence), then lane t writes row t of the pair factor (rows picked by
// selects), gauge vector t and its share of Linv.  The focal entries are owned by the last wave of the last workgroup.  pcg = 0: the last
// workgroup to arrive (agent-scope release/acquire around an arrival counter) runs post_linearisation.
constexpr int FIN_LANES = 8;
__global__ __launch_bounds__(256) void k_finalize(DeviceStructure ds, DeviceBuffers db, int pcg) {
    __shared__ int is_last;
    const int gt = blockIdx.x * blockDim.x + threadIdx.x;
    const LMState* st = db.st;
    if (pcg && db.pcg_zero) { for (int i = gt; i < db.pcg_zero_n; i += gridDim.x * blockDim.x) db.pcg_zero[i] = 0.0; }      // (the symmetric CG's S~ W~ is accumulated with atomics)
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x >= 192) {
        // focal-focal entries were accumulated in the slotted buffer: the last wave of the last block owns them
        const int foc_acc[4] = { ACC_SFF, ACC_RHSF, ACC_UDF, ACC_BCF };
        double foc[4];
        slots_take_n<4>(db, foc_acc, foc);
        const double sff = foc[0], rhsf = foc[1], udf = foc[2], bcf = foc[3];
        if (threadIdx.x == 192) {
            const int fo = ds.d - 1;
            const double dd = fmin(fmax(udf, st->min_diag), st->max_diag) / st->radius;
            db.S[(size_t)fo * ds.ld + fo] = sff + dd;
            db.rhs[fo] = rhsf; db.udiag[fo] = udf; db.bc[fo] = bcf;
            const double gg = fabs(bcf / st->fscale);
            if (gg > 0.0) atomic_max_nonneg(slot_ptr(db, ACC_GMAX), gg);
            if (!finite_d(sff + dd) || !finite_d(rhsf)) atomicAdd(slot_ptr(db, ACC_BAD_LIN), 1.0);
            if (pcg && !(sff + dd > 0.0)) atomicCAS(db.lin_info, 0, ds.d);
            if (pcg && db.pcg_W) {
                // focal row of the gauge vectors: only the focal/depth vector (7) touches the focal; x~_f = sqrt(S_ff) x_f
                const double lf = sqrt(sff + dd > 0.0 ? sff + dd : 1.0);
                for (int k = 0; k < 8; ++k) {
                    db.pcg_W[(size_t)k * ds.ld + fo] = k == 7 ? (double)(float)(lf * st->focal[st->cur] / st->fscale) : 0.0;
                    for (int e = ds.d; e < ds.ld; ++e) db.pcg_W[(size_t)k * ds.ld + e] = 0.0;
                }
            }
        }
    }
    double gm = 0.0;
    const int g = gt / FIN_LANES, t = gt % FIN_LANES;         // camera, lane of its group
    if (g < ds.ncam) {
        const int row0 = 6 * g, fo = ds.d - 1;
        const bool writer = t == 0;
        // every global load of this camera first (one L2 round trip): diagonal block, focal column, undamped diagonal, gradient, scales,
        // and what the gauge vectors and the pair pass's camera factor need
        double Sb[6][6], ud[6], bcv[6], csv[6], rhv[6], sjf[6], camv[6], Rm[9], Q9[9], cq = 0.0;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            ud[a] = db.udiag[row0 + a]; bcv[a] = db.bc[row0 + a]; csv[a] = db.cscale[row0 + a]; rhv[a] = db.rhs[row0 + a];
            sjf[a] = db.cd_part ? db.S[(size_t)(row0 + a) * ds.ld + fo] : 0.0;
#pragma unroll
            for (int b = 0; b < 6; ++b) Sb[a][b] = (b >= a) ? db.S[(size_t)(row0 + a) * ds.ld + row0 + b] : 0.0;
        }
        if (pcg) {
            const double* tab = db.camtab[st->cur];
#pragma unroll
            for (int e = 0; e < 6; ++e) camv[e] = db.pcg_W ? db.cam[st->cur][6 * (size_t)g + e] : 0.0;
#pragma unroll
            for (int e = 0; e < 9; ++e) { Rm[e] = db.pcg_W ? tab[cam_tab_index(CT_R + e, g, ds.ncam)] : 0.0; Q9[e] = db.pair_G ? tab[cam_tab_index(CT_QD + e, g, ds.ncam)] : 0.0; }
            cq = db.pcg_W ? tab[cam_tab_index(CT_CQ, g, ds.ncam)] : 0.0;
        }
        if (db.cd_part) {
            // deterministic mode: this camera's k_cam_diag_f chunks, in chunk order, on top of what the (single-writer) passes left -- added in
            // registers by every lane of the group, written back by one (the exact solver and the AUTO fallback read the block from memory)
            double acc[45];
#pragma unroll
            for (int k = 0; k < 45; ++k) acc[k] = 0.0;
            for (int c = ds.cam_chunk_ptr[g]; c < ds.cam_chunk_ptr[g + 1]; ++c) {
#pragma unroll
                for (int k = 0; k < 45; ++k) acc[k] += db.cd_part[(size_t)c * 48 + k];
            }
            int u = 0;
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int b = a; b < 6; ++b) Sb[a][b] += acc[u++];
#pragma unroll
            for (int a = 0; a < 6; ++a) { ud[a] += acc[21 + a]; sjf[a] += acc[27 + a]; bcv[a] += acc[33 + a]; rhv[a] += acc[39 + a]; }
            if (writer) {
#pragma unroll
                for (int a = 0; a < 6; ++a) {
                    db.udiag[row0 + a] = ud[a]; db.S[(size_t)(row0 + a) * ds.ld + fo] = sjf[a]; db.bc[row0 + a] = bcv[a]; db.rhs[row0 + a] = rhv[a];
#pragma unroll
                    for (int b = a + 1; b < 6; ++b) db.S[(size_t)(row0 + a) * ds.ld + row0 + b] = Sb[a][b];
                }
            }
        }
        bool bad = false;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            const int e = row0 + a;
            const double dd = fmin(fmax(ud[a], st->min_diag), st->max_diag) / st->radius;
            const double v = Sb[a][a] + dd;
            Sb[a][a] = v;
            if (writer) db.S[(size_t)e * ds.ld + e] = v;
            gm = fmax(gm, fabs(bcv[a] / csv[a]));
            bad = bad || !finite_d(v) || !finite_d(rhv[a]);
        }
        if (bad && writer) atomicAdd(slot_ptr(db, ACC_BAD_LIN), 1.0);
        if (pcg) {
            // Linv of the damped block (row-major lower, zeros above), see dense_solver.hip
            double L[6][6], Li[6][6];
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int c = 0; c < 6; ++c) L[r][c] = (c <= r) ? Sb[c][r] : 0.0;
            bool ok = true;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                double dj = L[j][j];
#pragma unroll
                for (int tt = 0; tt < 6; ++tt) if (tt < j) dj -= L[j][tt] * L[j][tt];
                ok = ok && (dj > 0.0);
                const double lji = fast_rsq(dj > 0.0 ? dj : 1.0);
                L[j][j] = lji;              // the RECIPROCAL of the pivot is what the inverse needs
#pragma unroll
                for (int i = 0; i < 6; ++i) if (i > j) {
                    double v = L[i][j];
#pragma unroll
                    for (int tt = 0; tt < 6; ++tt) if (tt < j) v -= L[i][tt] * L[j][tt];
                    L[i][j] = v * lji;
                }
            }
            if (!ok && writer) atomicCAS(db.lin_info, 0, row0 + 1);
#pragma unroll
            for (int c = 0; c < 6; ++c)
#pragma unroll
                for (int r = 0; r < 6; ++r) {
                    double v = (r == c) ? 1.0 : 0.0;
#pragma unroll
                    for (int tt = 0; tt < 6; ++tt) if (tt >= c && tt < r) v -= L[r][tt] * Li[tt][c];
                    Li[r][c] = (r < c) ? 0.0 : v * L[r][r];
                }
            // lane t < 6 owns row t of Linv (store) and of the pair factor G = Linv D E^T (pair_factor<true>): the row by selects
            double lrow[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                double v = 0.0;
#pragma unroll
                for (int r = 0; r < 6; ++r) v += (t == r ? 1.0 : 0.0) * Li[r][c];      // (weights, not selects: see gauge_vector_k)
                lrow[c] = v;
            }
            if (t < 6) {
#pragma unroll
                for (int c = 0; c < 6; ++c) db.pcg_binv[(size_t)g * 36 + t * 6 + c] = lrow[c];
                if (db.pair_G) {
#pragma unroll
                    for (int c = 0; c < 6; ++c) {
                        double v;
                        if (c < 3) {            // sum_{a < 3} Lw[r][a] (D E^T)[a][c],  (D E^T)[a][c] = cs[a] Q[c][a]
                            v = 0.0;
#pragma unroll
                            for (int a = 0; a < 3; ++a) v += lrow[a] * (csv[a] * Q9[3 * c + a]);
                        } else {
                            v = lrow[c] * csv[c];
                        }
                        db.pair_G[(size_t)g * 36 + 6 * t + c] = v;
                    }
                }
            }
            if (db.pcg_W) gauge_vector_k(ds, db, g, t, Li, camv, Rm, csv, cq);
        }
        if (!writer) gm = 0.0;
    } else if (gt - ds.ncam * FIN_LANES < ds.ld - ds.d) {
        const int e = ds.d + (gt - ds.ncam * FIN_LANES);
        db.S[(size_t)e * ds.ld + e] = 1.0;
        db.rhs[e] = 0.0;
    }
    gm = wave_max(gm);
    if ((threadIdx.x & 63) == 0 && gm > 0.0) atomic_max_nonneg(slot_ptr(db, ACC_GMAX), gm);
    if (pcg == 1) return;        // post_linearisation runs in the pair pass that follows (see k_schur_pairs, MODE 1)
    // ---- arrival: every wave drains its stores, one lane releases and takes a ticket ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int ticket = __hip_atomic_fetch_add(db.fin_counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        is_last = (ticket == (int)gridDim.x - 1);
        if (is_last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_store(db.fin_counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    if (is_last && threadIdx.x < 64) post_linearisation(ds, db);
}

// Deterministic mode, sharded solve: the per-chunk sums of k_cam_diag are folded into the reduced system (in chunk order, one thread
// per camera) BEFORE the partial system is packed for the exchange; k_finalize, which does this on one GPU, runs behind the all-reduce
// there and is then called with cd_part = null.
__global__ __launch_bounds__(64) void k_cd_fold(DeviceStructure ds, DeviceBuffers db) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ds.ncam || !db.cd_part) return;
    const int row0 = 6 * g, fo = ds.d - 1;
    double acc[45];
#pragma unroll
    for (int k = 0; k < 45; ++k) acc[k] = 0.0;
    for (int c = ds.cam_chunk_ptr[g]; c < ds.cam_chunk_ptr[g + 1]; ++c) {
#pragma unroll
        for (int k = 0; k < 45; ++k) acc[k] += db.cd_part[(size_t)c * 48 + k];
    }
    int u = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = a; b < 6; ++b) db.S[(size_t)(row0 + a) * ds.ld + row0 + b] += acc[u++];
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        db.udiag[row0 + a] += acc[21 + a];
        db.S[(size_t)(row0 + a) * ds.ld + fo] += acc[27 + a];
        db.bc[row0 + a] += acc[33 + a];
        db.rhs[row0 + a] += acc[39 + a];
    }
}
void launch_cd_fold(hipStream_t s, const DeviceStructure& ds, const DeviceBuffers& db) {
    if (db.cd_part) hipLaunchKernelGGL(k_cd_fold, dim3((ds.ncam + 63) / 64), dim3(64), 0, s, ds, db);
}

void launch_finalize(hipStream_t s, const DeviceStructure& ds, const DeviceBuffers& db, int pcg) {
    const int work = ds.ncam * FIN_LANES + (ds.ld - ds.d) + 64;     // eight lanes per camera, padding rows, room for the focal wave
    hipLaunchKernelGGL(k_finalize, dim3((work + 255) / 256), dim3(256), 0, s, ds, db, pcg);
}

// iteration 0 bookkeeping: x_norm from the accumulated ||x||^2
__global__ void k_iter0(DeviceBuffers db) {
    LMState* st = db.st;
    const double x2 = slots_take(db, ACC_XNEW2);
    for (int e = 0; e < SLOT_W; ++e) if (e != ACC_XNEW2) (void)slots_take(db, e);
    if (threadIdx.x == 0) { st->x_norm = sqrt(x2); *db.fin_counter = 0; }
}
void launch_iter0(hipStream_t s, const DeviceStructure& ds, const DeviceBuffers& db) {
    (void)ds;
    hipLaunchKernelGGL(k_iter0, dim3(1), dim3(64), 0, s, db);
}

// ------------------------------------------------------------------------------------------
// back-substitution + trial point
// ------------------------------------------------------------------------------------------
// One thread per camera: delta = scale * y ; trial camera = camera - delta ; step table; table of the trial camera
__global__ __launch_bounds__(BLK) void k_cam_update(DeviceStructure ds, DeviceBuffers db) {
    __shared__ double scratch[BLK / 64];
    if (db.cg_gate && !db.cg_force && db.cg_gate[0] == 0) return;      // the CG batch in front of this launch was too short
    LMState* st = db.st;
    const int cur = st->cur, nxt = cur ^ 1;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    double step2 = 0.0, xn2 = 0.0, gdot = 0.0;
    if (j < ds.ncam) {
        double dlt[6], cn[6], z[6];
        // every load of this camera before the first store (the stores below may alias as far as the compiler can tell: a load behind
        // one of them waits for its own round trip -- six in a row in the loop this replaces)
        double c0[6], cs[6], xin[6], Lt[6][6], Q9[9];
#pragma unroll
        for (int e = 0; e < 6; ++e) { c0[e] = db.cam[cur][6 * j + e]; cs[e] = db.cscale[6 * j + e]; }
        // Q = R K' and the first-order flag of the camera AT THE LINEARISATION POINT: the back-substitution re-evaluates the camera
        // block of an observation in the factored form A = P [ -[X_g]x | I ] diag(Q, I) (sfmba_device.h), so the step arrives as Q dw
#pragma unroll
        for (int e = 0; e < 9; ++e) Q9[e] = db.camtab[cur][cam_tab_index(CT_QD + e, j, ds.ncam)];
        const double small_cur = db.camtab[cur][cam_tab_index(CT_SMALL, j, ds.ncam)];
        double Rt_cur[12], bcj[6];
        if (db.pu32) {
#pragma unroll
            for (int e = 0; e < 12; ++e) Rt_cur[e] = db.camtab[cur][cam_tab_index(CT_R + e, j, ds.ncam)];
#pragma unroll
            for (int e = 0; e < 6; ++e) bcj[e] = db.bc[6 * j + e];
        }
        if (db.pcg_vec) {          // z_j = Linv_j^T x~_j  (block-Jacobi transformed unknowns)
            const double* x = db.pcg_vec + (size_t)db.pcg_flags[2] * ds.ld + 6 * j;
            const double* Li = db.pcg_linv + (size_t)j * 36;
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                xin[t] = x[t];
#pragma unroll
                for (int c = 0; c < 6; ++c) Lt[t][c] = (c <= t) ? Li[t * 6 + c] : 0.0;
            }
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                double v = 0.0;
#pragma unroll
                for (int t = 0; t < 6; ++t) if (t >= c) v += Lt[t][c] * xin[t];
                z[c] = v;
            }
        } else {
#pragma unroll
            for (int c = 0; c < 6; ++c) z[c] = db.rhs[6 * j + c];
        }
#pragma unroll
        for (int e = 0; e < 6; ++e) {
            dlt[e] = cs[e] * z[e];
            cn[e] = c0[e] - dlt[e];
            const double df = c0[e] - cn[e];
            step2 += df * df;
            xn2 += cn[e] * cn[e];
        }
#pragma unroll
        for (int e = 0; e < 6; ++e) db.cam[nxt][6 * j + e] = cn[e];
        double ctn[CT_STRIDE];
        make_cam_table(cn, cs, ctn);
        for (int e = 0; e < CT_STRIDE; ++e) db.camtab[nxt][cam_tab_index(e, j, ds.ncam)] = ctn[e];
        double stb[ST_STRIDE] = {};
        for (int e = 0; e < 9; ++e) stb[ST_RN + e] = ctn[CT_R + e];
        for (int e = 0; e < 3; ++e) {
            stb[ST_DQ + e] = Q9[3 * e] * dlt[0] + Q9[3 * e + 1] * dlt[1] + Q9[3 * e + 2] * dlt[2];
            stb[ST_DT + e] = dlt[3 + e]; stb[ST_TN + e] = ctn[CT_T + e];
        }
        stb[ST_SMALL] = small_cur;
        for (int e = 0; e < ST_STRIDE; ++e) db.steptab[cam_tab_index(e, j, ds.ncam)] = stb[e];
        if (db.pu32) {
            // the first sweep of k_point_update in F32J mode: what it needs of this camera as ONE 80-byte fp32 record; and the part of the model
            // cost change that sweep no longer forms per observation: sum_obs u . r = (step) . (gradient) = sum z_i bc_i in the scaled unknowns
            float* rec = db.pu32 + 20 * (size_t)j;
#pragma unroll
            for (int e = 0; e < 12; ++e) rec[e] = (float)Rt_cur[e];
#pragma unroll
            for (int e = 0; e < 8; ++e) rec[12 + e] = (float)stb[e];
#pragma unroll
            for (int e = 0; e < 6; ++e) gdot += z[e] * bcj[e];
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const double f0 = st->focal[cur];
        double zf = db.rhs[ds.d - 1];
        if (db.pcg_vec) zf = db.pcg_linv[(size_t)ds.ncam * 36] * db.pcg_vec[(size_t)db.pcg_flags[2] * ds.ld + ds.d - 1];
        const double fn = f0 - st->fscale * zf;
        st->focal[nxt] = fn;
        const double df = f0 - fn;
        step2 += df * df;
        xn2 += fn * fn;
        if (db.pu32) gdot += zf * db.bc[ds.d - 1];
    }
    const double s2 = block_sum(step2, scratch);
    const double x2 = block_sum(xn2, scratch);
    if (threadIdx.x == 0) { atomicAdd(slot_ptr(db, ACC_STEP2), db.shared_weight * s2); atomicAdd(slot_ptr(db, ACC_XNEW2), db.shared_weight * x2); }
    if (db.pu32) {
        const double gd = block_sum(gdot, scratch);
        if (threadIdx.x == 0) atomicAdd(slot_ptr(db, ACC_MODEL), db.shared_weight * gd);      // (sharded: the cameras are replicated, rank 0 counts them)
    }
}

// ------------------------------------------------------------------------------------------
// LM control: the accept/reject logic of ceres::internal::TrustRegionMinimizer::Minimize()
// [Ceres-upstream], one thread.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void lm_post(int* mb, int seq, int termination, int message, int iter, int cg_iters = 0) {
    if (!mb) return;
    __hip_atomic_store(mb + 4, cg_iters, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(mb + 1, termination, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(mb + 2, message, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(mb + 3, iter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(mb, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// (all 64 lanes of ONE wave; every accumulator it reads must be complete and visible: a kernel of its own behind the passes)
__device__ __forceinline__ void lm_control_body(const DeviceBuffers& db) {
    LMState* st = db.st;
    if (db.cg_gate && !db.cg_force && db.cg_gate[0] == 0) {
        // the linear solve has not converged within the launches enqueued so far: tell the host (termination code -2),
        // touch nothing -- it will enqueue more CG iterations followed by the same three kernels
        if ((threadIdx.x & 63) == 0) { st->retry = 1; const int seq = ++st->mail_seq; lm_post(db.lm_mailbox, seq, -2, 0, st->iter); }
        return;
    }
    double trial2, model, step2, xnew2, bad_trial;
    if (db.shard_scal) {
        // sharded solve: the sums over the ranks sit in the all-reduced scalar block (k_shard_pack emptied the slots)
        trial2 = db.shard_scal[0]; model = db.shard_scal[1]; step2 = db.shard_scal[2]; xnew2 = db.shard_scal[3]; bad_trial = db.shard_scal[4];
    } else {
        const int ctl_acc[5] = { ACC_TRIAL_COST, ACC_MODEL, ACC_STEP2, ACC_XNEW2, ACC_BAD_TRIAL };
        double ctl[5];
        slots_take_n<5>(db, ctl_acc, ctl);
        trial2 = ctl[0]; model = ctl[1]; step2 = ctl[2]; xnew2 = ctl[3]; bad_trial = ctl[4];
    }
    if ((threadIdx.x & 63) != 0) return;
    st->retry = 0;
    st->lin_info = *db.lin_info;
    *db.lin_info = 0;
    const int seq = ++st->mail_seq;
    const int cg_iters = db.cg_gate ? db.cg_gate[1] : 0;          // CG iterations of this LM iteration (device-side count)
    if (st->termination != -1) { if (db.st_mirror) *db.st_mirror = *st; lm_post(db.lm_mailbox, seq, st->termination, st->message, st->iter, cg_iters); return; }
    const int it = ++st->iter;
    TraceRow row = {};
    row.iteration = it;
    const bool lin_fail = st->lin_info != 0 || !finite_d(step2) || !finite_d(model);
    const bool step_valid = !lin_fail && model > 0.0;
    row.step_is_valid = step_valid;
    row.gradient_max_norm = st->gmax;
    double report_cost = st->cost;
    st->last_step_successful = 0;
    if (!step_valid) {
        if (++st->consecutive_invalid >= st->max_consecutive_invalid) {
            st->termination = SFMBA_FAILURE;
            st->message = MSG_INVALID_STEPS;
        } else {
            st->radius *= st->invalid_shrink;
            st->unsuccessful++;
        }
    } else {
        st->consecutive_invalid = 0;
        double cand = 0.5 * trial2;
        if (bad_trial != 0.0 || !finite_d(cand)) cand = DBL_MAX;
        st->residual_evals++;
        row.step_norm = sqrt(step2);
        const double step_tol = st->parameter_tolerance * (st->x_norm + st->parameter_tolerance);
        if (row.step_norm <= step_tol) {
            st->termination = SFMBA_CONVERGENCE;
            st->message = MSG_PARAMETER_TOL;
        } else {
            row.cost_change = st->cost - cand;
            if (fabs(row.cost_change) <= st->function_tolerance * st->cost) {
                st->termination = SFMBA_CONVERGENCE;
                st->message = MSG_FUNCTION_TOL;
            } else {
                row.relative_decrease = row.cost_change / model;
                if (row.relative_decrease > st->min_relative_decrease) {
                    row.step_is_successful = 1;
                    st->last_step_successful = 1;
                    st->cur ^= 1;
                    st->cost = cand;
                    st->x_norm = sqrt(xnew2);
                    const double t = 2.0 * row.relative_decrease - 1.0;
                    st->radius = st->radius / fmax(1.0 / 3.0, 1.0 - t * t * t);
                    st->radius = fmin(st->max_radius, st->radius);
                    st->decrease_factor = 2.0;
                    st->successful++;
                    st->x_is_new = 1;
                    report_cost = cand;
                } else {
                    st->radius = st->radius / st->decrease_factor;
                    st->decrease_factor *= 2.0;
                    st->unsuccessful++;
                    report_cost = cand;
                }
            }
        }
    }
    if (st->termination == -1 && st->radius <= st->min_radius) {
        st->termination = SFMBA_CONVERGENCE;
        st->message = MSG_MIN_RADIUS;
    }
    row.cost = report_cost;
    row.trust_region_radius = st->radius;
    if (it < db.trace_cap) db.trace[it] = row;
    st->lin_info = 0;
    if (db.st_mirror) *db.st_mirror = *st;      // plain stores; the release store of the sequence number in lm_post orders them
    lm_post(db.lm_mailbox, seq, st->termination, st->message, st->iter, cg_iters);
}

__global__ void k_lm_control(DeviceBuffers db) { lm_control_body(db); }

// Back-substitution + trial point, four lanes per point like k_point_build:
//   y_p = (V + D^2)^-1 (b_p - W^T y_c), trial point, model cost change, trial cost.
// With V + D^2 = L L^T, t = L^-1 b_p and C = B~ L^-T (left behind per POINT by k_point_build: pt_t, M = diag(s_p) L^-T, the table entry):
//   u   = A (camera step) + g (focal step)     per observation
//   z   = t - sum_obs C^T u                      per point
//   dX  = M z ;   J step = -(u + C z)            (model cost change; B~ y_p = C L^T y_p = C z)
// Nothing per observation is read but its camera and coordinates (rounds 2 / 3 streamed a 64-byte record per observation here): the
// projection at the linearisation point is re-evaluated in fp64 from the camera's R, t and the point-table entry, the camera block acts
// on the step in the factored form of sfmba_device.h,  A [dw; dt] = P (Q dw x X_g + dt),  P = (f / p_z) [[1, 0, -x_p], [0, 1, -y_p]],
// with Q dw formed once per camera by k_cam_update, and C = (P R) L~ in the precision of the Jacobian blocks, exactly as the
// reduced-system passes form it.  The quad's lanes take the point's observations in turn and keep
//   sum C^T u (3),  sum u.r,  sum |u|^2,  sum C^T C (6)
// in registers; after ONE quad reduction every lane of the quad has z = t - sum C^T u, the trial point, and the point's share of the model
// cost change in closed form --
//   sum_obs [ (u + C z).r - |u + C z|^2 / 2 ] = sum u.r + z.t - sum |u|^2 / 2 - z.(sum C^T u) - z^T (sum C^T C) z / 2      (sum C^T r = L^-1 b_p = t)
// -- so the second sweep over the observations only evaluates the TRIAL residual (projection with the trial pose at the trial point): nothing
// per observation has to survive the first sweep, no LDS, and the per-point arithmetic runs on all lanes (the lane-per-observation form
// of the first half of round 4: 33.6 against 29.9 us at BASELINE config 3, 242 against 189 at config 5).
// (Measured in round 5 and not kept: the LM control logic run by the LAST WORKGROUP TO ARRIVE of this launch instead of a launch of its own.
// With a release fence per workgroup the launch went from 27 to 83 us at BASELINE config 3 (747 from 190 at config 5: buffer_wbl2 3 125 times);
// with the slots read as agent atomics and no release, one ticket counter serialised the 3 125 arrivals (57 us); with two-level tickets on
// separate cache lines 36.5 us against 29.9 + 5.6 for the two launches: 4 260 against 4 270 LM iterations/s -- a wash, so the simpler form stays.)
#ifndef SFMBA_TRIAL_POSE_FROM_PARAMS
#define SFMBA_TRIAL_POSE_FROM_PARAMS 1
#endif
#ifndef SFMBA_TRIAL_POSE_MIN_CAMS
#define SFMBA_TRIAL_POSE_MIN_CAMS 400     // 96 bytes of [R | t] per camera against a 32 KB L1: between the two measured sizes (200: table, 1 000: parameters)
#endif
template <typename T>
__global__ __launch_bounds__(PBK, 4) void k_point_update(DeviceStructure ds, DeviceBuffers db) {
    __shared__ double scratch[WPB * 5];
    if (db.cg_gate && !db.cg_force && db.cg_gate[0] == 0) return;      // see k_cam_update
    const LMState* st = db.st;
    const int cur = st->cur, nxt = cur ^ 1;
    const double* tab = db.camtab[cur];
    const double* stab = db.steptab;
    const double focal = st->focal[cur], focal_n = st->focal[nxt];
    const double dfoc = focal - focal_n;           // unscaled focal step to SUBTRACT (= fscale * y_f)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int gw = blockIdx.x * WPB + w;
    double trial = 0.0, model = 0.0, step2 = 0.0, xn2 = 0.0, bad = 0.0;
    const int sub = lane & (PB_LPP - 1);
    const int slot = gw * (64 / PB_LPP) + (lane / PB_LPP);
    const bool have = slot < ds.npt;
    const int ip = have ? (ds.pt_order ? ds.pt_order[slot] : slot) : 0;
    const size_t i = (size_t)ip;
    const int q0 = have ? ds.pt_ptr[ip] : 0, q1 = have ? ds.pt_ptr[ip + 1] : 0;
    const PtRecA<T> pa = load_ptrec(reinterpret_cast<const PtRecA<T>*>(db.PA) + i);
    double zacc[3] = { 0, 0, 0 }, ur = 0.0, uu = 0.0;
    T G[6] = { (T)0, (T)0, (T)0, (T)0, (T)0, (T)0 };       // (sum C^T C: a second-order term of the model cost change; summed in the precision of C)
    if (sizeof(T) == 4 && db.pu32) {
        // F32J, unsharded: the camera's R, t and step as ONE fp32 record (k_cam_update), five 16-byte gathers instead of ten -- the pass is bound by the
        // number of gather instructions (every lane another camera: 64 lines each; TA_TA_BUSY 92 % at BASELINE config 5).  sum u . r is not formed here.
        const float4* recs = reinterpret_cast<const float4*>(db.pu32);
        int q = q0 + sub;
        int j_next = q < q1 ? ds.obs_cam[q] : 0;
        while (__any(q < q1)) {
            const bool act = q < q1;
            const float4* rec = recs + 5 * (size_t)j_next;
            const float4 a0 = rec[0], a1 = rec[1], a2 = rec[2], a3 = rec[3], a4 = rec[4];
            const double Rt[12] = { a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z, a2.w };
            const double dr[8] = { a3.x, a3.y, a3.z, a3.w, a4.x, a4.y, a4.z, a4.w };
            q += PB_LPP;
            if (q < q1) j_next = ds.obs_cam[q];
            if (act) {
                ImpObs o; T C[6];
                imp_eval<T>(Rt, dr, focal, pa, o, C);
                const double u0 = o.u[0] + o.xp * dfoc, u1 = o.u[1] + o.yp * dfoc;
                uu += u0 * u0 + u1 * u1;
                zacc[0] += (double)C[0] * u0 + (double)C[3] * u1; zacc[1] += (double)C[1] * u0 + (double)C[4] * u1; zacc[2] += (double)C[2] * u0 + (double)C[5] * u1;
                G[0] += C[0] * C[0] + C[3] * C[3]; G[1] += C[0] * C[1] + C[3] * C[4]; G[2] += C[0] * C[2] + C[3] * C[5];
                G[3] += C[1] * C[1] + C[4] * C[4]; G[4] += C[1] * C[2] + C[4] * C[5]; G[5] += C[2] * C[2] + C[5] * C[5];
            }
        }
    } else {
        int q = q0 + sub;
        int j_next = q < q1 ? ds.obs_cam[q] : 0;
        double ox_next = 0.0, oy_next = 0.0;
        if (q < q1) load_obs<T>(ds.obs_xy, q, ox_next, oy_next);
        while (__any(q < q1)) {
            const bool act = q < q1;
            const int j = j_next;
            const double ox = ox_next, oy = oy_next;
            const CamRow ct = { tab + 4 * (size_t)(j), ds.ncam };
            const CamRow stb = { stab + 4 * (size_t)(j), ds.ncam };
            double Rt[12], dr[8];
#pragma unroll
            for (int e = 0; e < 12; ++e) Rt[e] = ct[CT_R + e];
#pragma unroll
            for (int e = 0; e < 8; ++e) dr[e] = stb[e];
            q += PB_LPP;
            if (q < q1) { j_next = ds.obs_cam[q]; load_obs<T>(ds.obs_xy, q, ox_next, oy_next); }
            if (act) {
                ImpObs o; T C[6];
                imp_eval<T>(Rt, dr, focal, pa, o, C);                    // projection, X_g, u = P (Q dw x X_g + dt), C = (P R) L~
                const double u0 = o.u[0] + o.xp * dfoc, u1 = o.u[1] + o.yp * dfoc;      // + the focal step
                const double r0 = focal * o.xp - ox, r1 = focal * o.yp - oy;
                ur += u0 * r0 + u1 * r1;
                uu += u0 * u0 + u1 * u1;
                zacc[0] += (double)C[0] * u0 + (double)C[3] * u1; zacc[1] += (double)C[1] * u0 + (double)C[4] * u1; zacc[2] += (double)C[2] * u0 + (double)C[5] * u1;
                G[0] += C[0] * C[0] + C[3] * C[3]; G[1] += C[0] * C[1] + C[3] * C[4]; G[2] += C[0] * C[2] + C[3] * C[5];
                G[3] += C[1] * C[1] + C[4] * C[4]; G[4] += C[1] * C[2] + C[4] * C[5]; G[5] += C[2] * C[2] + C[5] * C[5];
            }
        }
    }
#define SFMBA_QUADSUM(x) { x = xlane_add<1>(x); x = xlane_add<2>(x); }
#pragma unroll
    for (int c = 0; c < 3; ++c) SFMBA_QUADSUM(zacc[c])
    SFMBA_QUADSUM(ur) SFMBA_QUADSUM(uu)
    double Gd[6];            // (the lane's own sum in the precision of C, across the quad in fp64 like every cross-lane sum)
#pragma unroll
    for (int c = 0; c < 6; ++c) { Gd[c] = (double)G[c]; SFMBA_QUADSUM(Gd[c]) }
#undef SFMBA_QUADSUM
    // (the point's t and M: loaded here, behind the sweep -- in front of it they cost the sweep its fourth wave per SIMD)
    double tp[3], Mp[6];
#pragma unroll
    for (int c = 0; c < 3; ++c) tp[c] = db.pt_t[3 * i + c];
#pragma unroll
    for (int c = 0; c < 6; ++c) Mp[c] = db.pt_M[6 * i + c];
    const double z0 = tp[0] - zacc[0], z1 = tp[1] - zacc[1], z2 = tp[2] - zacc[2];
    const double dX[3] = { Mp[0] * z0 + Mp[1] * z1 + Mp[2] * z2, Mp[3] * z1 + Mp[4] * z2, Mp[5] * z2 };
    double Xn[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) Xn[c] = pa.X[c] - dX[c];
    if (have && sub == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double df = pa.X[c] - Xn[c];
            step2 += df * df;
            xn2 += Xn[c] * Xn[c];
            db.pts[nxt][3 * i + c] = Xn[c];
        }
        const double zGz = Gd[0] * z0 * z0 + Gd[3] * z1 * z1 + Gd[5] * z2 * z2 + 2.0 * (Gd[1] * z0 * z1 + Gd[2] * z0 * z2 + Gd[4] * z1 * z2);
        model += ur + (z0 * tp[0] + z1 * tp[1] + z2 * tp[2]) - 0.5 * uu - (z0 * zacc[0] + z1 * zacc[1] + z2 * zacc[2]) - 0.5 * zGz;
    }
    // Trial sweep.  F32J with more cameras than an L1 holds rows of (SFMBA_TRIAL_POSE_MIN_CAMS): three gathers of the trial camera's PARAMETERS and R rebuilt per
    // observation (fp64 sincos on a VALU that is a third busy) instead of six gathers of the stored [R | t] -- measured 142 -> 116 us at BASELINE config 5
    // (1 000 cameras: the pass is bound by its gather instructions), but 23.4 -> 25.4 us at config 3 (200 cameras: it is not), hence the threshold.
    const bool pose_from_cam = SFMBA_TRIAL_POSE_FROM_PARAMS && sizeof(T) == 4 && ds.ncam >= SFMBA_TRIAL_POSE_MIN_CAMS;
    {
        int q = q0 + sub;
        int j_next = q < q1 ? ds.obs_cam[q] : 0;
        double ox_next = 0.0, oy_next = 0.0;
        if (q < q1) load_obs<T>(ds.obs_xy, q, ox_next, oy_next);
        while (__any(q < q1)) {
            const bool act = q < q1;
            const int j = j_next;
            const double ox = ox_next, oy = oy_next;
            double RTn[12];
            if (pose_from_cam) {
                // the trial camera's six parameters (three 16-byte gathers) and R rebuilt here, instead of six gathers of the stored [R | t]
                const double2* cp = reinterpret_cast<const double2*>(db.cam[nxt] + 6 * (size_t)j);
                const double2 c0 = cp[0], c1 = cp[1], c2 = cp[2];
                const double cn6[6] = { c0.x, c0.y, c1.x, c1.y, c2.x, c2.y };
                pose_from_params(cn6, RTn);
            } else {
                const CamRow stb = { stab + 4 * (size_t)(j), ds.ncam };
#pragma unroll
                for (int e = 0; e < 12; ++e) RTn[e] = stb[ST_RN + e];
            }
            q += PB_LPP;
            if (q < q1) { j_next = ds.obs_cam[q]; load_obs<T>(ds.obs_xy, q, ox_next, oy_next); }
            if (act) {
                const Proj pn = project_point(RTn, 0, 9, Xn);
                const double n0 = focal_n * pn.xp - ox, n1 = focal_n * pn.yp - oy;
                if (!finite_d(n0) || !finite_d(n1)) bad = 1.0;
                trial += n0 * n0 + n1 * n1;
            }
        }
    }
    double sums[5] = { trial, model, step2, xn2, bad };
    const double tot = block_sums<5>(sums, scratch);
    if (threadIdx.x < 5) {
        const int which = threadIdx.x == 0 ? ACC_TRIAL_COST : threadIdx.x == 1 ? ACC_MODEL : threadIdx.x == 2 ? ACC_STEP2 : threadIdx.x == 3 ? ACC_XNEW2 : ACC_BAD_TRIAL;
        if (threadIdx.x < 4 || tot != 0.0) atomicAdd(slot_ptr(db, which), tot);
    }
}

void launch_cam_update(hipStream_t s, const DeviceStructure& ds, const DeviceBuffers& db) {
    hipLaunchKernelGGL(k_cam_update, dim3((ds.ncam + BLK - 1) / BLK), dim3(BLK), 0, s, ds, db);
}

template <typename T>
void launch_point_update(hipStream_t s, const DeviceStructure& ds, const DeviceBuffers& db) {
    const int per_wg = WPB * (64 / PB_LPP);
    hipLaunchKernelGGL(k_point_update<T>, dim3(std::max(1, (ds.npt + per_wg - 1) / per_wg)), dim3(PBK), 0, s, ds, db);
}
template void launch_point_update<float>(hipStream_t, const DeviceStructure&, const DeviceBuffers&);
template void launch_point_update<double>(hipStream_t, const DeviceStructure&, const DeviceBuffers&);

void launch_control(hipStream_t s, const DeviceStructure& ds, const DeviceBuffers& db) {
    (void)ds;
    hipLaunchKernelGGL(k_lm_control, dim3(1), dim3(64), 0, s, db);
}

// ------------------------------------------------------------------------------------------
// kernel-level entry points used by the parity tests (C ABI: sfmba_problem_eval_*)
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(BLK) void k_eval_residuals(DeviceStructure ds, DeviceBuffers db, const int* __restrict__ obs_pt,
                                                        const int* __restrict__ perm, double* __restrict__ res_out, double* cost_out) {
    __shared__ double scratch[BLK / 64];
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    const int cur = db.st->cur;
    double c = 0.0;
    if (q < ds.nobs) {
        const int i = obs_pt[q];
        const CamRow ct = { db.camtab[cur] + 4 * (size_t)(ds.obs_cam[q]), ds.ncam };
        const double X[3] = { db.pts[cur][3 * i], db.pts[cur][3 * i + 1], db.pts[cur][3 * i + 2] };
        double ox, oy;
        load_obs<T>(ds.obs_xy, q, ox, oy);
        const Proj pr = project_point(ct, CT_R, CT_T, X);
        const double f = db.st->focal[cur];
        const double r0 = f * pr.xp - ox, r1 = f * pr.yp - oy;
        if (res_out) { res_out[2 * (size_t)perm[q]] = r0; res_out[2 * (size_t)perm[q] + 1] = r1; }
        c = r0 * r0 + r1 * r1;
    }
    c = block_sum(c, scratch);
    if (threadIdx.x == 0 && cost_out) atomicAdd(cost_out, 0.5 * c);
}

template <typename T>
__global__ __launch_bounds__(BLK) void k_eval_jacobian(DeviceStructure ds, DeviceBuffers db, const int* __restrict__ obs_pt,
                                                       const int* __restrict__ perm, double* jc, double* jp, double* jf) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= ds.nobs) return;
    const int cur = db.st->cur;
    const int i = obs_pt[q];
    const CamRow ct = { db.camtab[cur] + 4 * (size_t)(ds.obs_cam[q]), ds.ncam };
    const double X[3] = { db.pts[cur][3 * i], db.pts[cur][3 * i + 1], db.pts[cur][3 * i + 2] };
    const Proj pr = project_point(ct, CT_R, CT_T, X);
    const double f = db.st->focal[cur];
    T B[6], A[12];
    point_block<T>(ct, pr, f, B);
    camera_block<T>(ct, pr, f, X, B, A);
    const size_t k = (size_t)perm[q];
    if (jc) for (int e = 0; e < 12; ++e) jc[12 * k + e] = (double)A[e];
    if (jp) for (int e = 0; e < 6; ++e) jp[6 * k + e] = (double)B[e];
    if (jf) { jf[2 * k] = pr.xp; jf[2 * k + 1] = pr.yp; }
}

template <typename T>
void launch_eval_residuals(hipStream_t s, const DeviceStructure& ds, const DeviceBuffers& db, const int* obs_pt_and_perm,
                           double* res_out, double* cost_out) {
    // obs_pt_and_perm: [2*nobs] = point slot per point-major obs, then perm (point-major position -> caller index)
    hipLaunchKernelGGL(k_eval_residuals<T>, dim3((ds.nobs + BLK - 1) / BLK), dim3(BLK), 0, s, ds, db,
                       obs_pt_and_perm, obs_pt_and_perm + ds.nobs, res_out, cost_out);
}
template void launch_eval_residuals<float>(hipStream_t, const DeviceStructure&, const DeviceBuffers&, const int*, double*, double*);
template void launch_eval_residuals<double>(hipStream_t, const DeviceStructure&, const DeviceBuffers&, const int*, double*, double*);

template <typename T>
void launch_eval_jacobian(hipStream_t s, const DeviceStructure& ds, const DeviceBuffers& db, const int* obs_pt, const int* perm,
                          double* jc, double* jp, double* jf) {
    hipLaunchKernelGGL(k_eval_jacobian<T>, dim3((ds.nobs + BLK - 1) / BLK), dim3(BLK), 0, s, ds, db, obs_pt, perm, jc, jp, jf);
}
template void launch_eval_jacobian<float>(hipStream_t, const DeviceStructure&, const DeviceBuffers&, const int*, const int*, double*, double*, double*);
template void launch_eval_jacobian<double>(hipStream_t, const DeviceStructure&, const DeviceBuffers&, const int*, const int*, double*, double*, double*);

// full symmetric unpadded copy of the reduced system + the scale vector (sfmba_problem_build_reduced)
__global__ void k_mirror_scale(DeviceStructure ds, DeviceBuffers db, double* S_full, double* scale_out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = blockIdx.y;
    if (c >= ds.d || r >= ds.d) return;
    const double v = c >= r ? db.S[(size_t)r * ds.ld + c] : db.S[(size_t)c * ds.ld + r];
    S_full[(size_t)r * ds.d + c] = v;
    if (r == 0 && scale_out) scale_out[c] = c < ds.d - 1 ? db.cscale[c] : db.st->fscale;
}

void launch_mirror_scale(hipStream_t s, const DeviceStructure& ds, const DeviceBuffers& db, double* S_full, double* scale_out) {
    hipLaunchKernelGGL(k_mirror_scale, dim3((ds.d + 255) / 256, ds.d), dim3(256), 0, s, ds, db, S_full, scale_out);
}

}  // namespace sfmba
