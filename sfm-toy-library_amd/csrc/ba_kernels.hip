// ba_kernels.hip -- hand-written gfx950 kernels of the bundle-adjustment hot path.
//
// Reference path replaced: everything ceres::Solve does per LM iteration for the problem that
// SfMToyLib/SfMBundleAdjustmentUtils.cpp:142-179 builds (residual+Jacobian evaluation, Schur
// elimination of the points, back-substitution, trial-point cost) -- see DESIGN.md for the
// kernel-by-kernel map and SURVEY.md Appendix A for the math.
//
//   k_point_build  (point-major)  per point: V, b, E_f, LDL of V+D^2, whitened blocks Y = A~^T B~ L^-T
//   k_cam_schur    (camera-major) per camera row block of S: U_jj, -sum Y_a Y_b^T accumulated in LDS
//   k_finalize     damping of the reduced diagonal, gradient max-norm
//   k_cam_update / k_point_update  back-substitution, trial point, trial cost, model cost change
//   k_lm_control   accept/reject + trust-region update on the device (no host round trip needed)
#include "ba_kernels.h"
#include "sfmba_device.h"
#include "../../include/sfmba.h"

namespace sfmba {

#define BLK 256

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    return v;
}

__device__ __forceinline__ bool finite_d(double v) { return fabs(v) <= DBL_MAX; }

// ------------------------------------------------------------------------------------------
// camera tables
// ------------------------------------------------------------------------------------------
__device__ void make_cam_table(const double cam[6], const double* scale6, double* ct) {
    const double w0 = cam[0], w1 = cam[1], w2 = cam[2];
    const double theta2 = w0 * w0 + w1 * w1 + w2 * w2;
    double R[9], K[9];
    double small = 0.0;
    if (theta2 > DBL_EPSILON) {
        const double theta = sqrt(theta2);
        double s, c;
        sincos(theta, &s, &c);
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
}

__global__ void k_cam_setup(int ncam, const double* __restrict__ cam, const double* __restrict__ cscale, double* __restrict__ camtab) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ncam) return;
    double c6[6], ct[CT_STRIDE];
    for (int e = 0; e < 6; ++e) c6[e] = cam[6 * j + e];
    make_cam_table(c6, cscale + 6 * j, ct);
    for (int e = 0; e < CT_STRIDE; ++e) camtab[(size_t)j * CT_STRIDE + e] = ct[e];
}

template <typename T>
void launch_cam_setup(hipStream_t s, const DeviceStructure& ds, const DeviceBuffers& db, int which) {
    hipLaunchKernelGGL(k_cam_setup, dim3((ds.ncam + 63) / 64), dim3(64), 0, s, ds.ncam, db.cam[which], db.cscale, db.camtab[which]);
}
template void launch_cam_setup<float>(hipStream_t, const DeviceStructure&, const DeviceBuffers&, int);
template void launch_cam_setup<double>(hipStream_t, const DeviceStructure&, const DeviceBuffers&, int);

// ||x||^2 of the current parameters -> acc[ACC_XNEW2]
__global__ void k_xnorm(DeviceStructure ds, DeviceBuffers db) {
    __shared__ double scratch[BLK / 64];
    const int cur = db.st->cur;
    const double* cam = db.cam[cur];
    const double* pts = db.pts[cur];
    const int nc = 6 * ds.ncam, np = 3 * ds.npt;
    double s = 0.0;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < nc + np; e += gridDim.x * blockDim.x) {
        const double v = e < nc ? cam[e] : pts[e - nc];
        s += v * v;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { const double f = db.st->focal[cur]; s += f * f; }
    s = block_sum(s, scratch);
    if (threadIdx.x == 0) atomicAdd(&db.st->acc[ACC_XNEW2], s);
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
            const double* ct = tab + (size_t)ds.obs_cam[q] * CT_STRIDE;
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
template <typename T>
__global__ __launch_bounds__(BLK) void k_colnorm_cams(DeviceStructure ds, DeviceBuffers db) {
    __shared__ double scratch[BLK / 64];
    const int4 ch = ds.chunks[blockIdx.x];
    const int j = ch.x;
    const int wlo = ch.w * ds.win_cams;
    if (!(wlo <= j && j < wlo + ds.win_cams)) return;   // only the primary window chunk of each range
    const int cur = db.st->cur;
    const double* ct = db.camtab[cur] + (size_t)j * CT_STRIDE;
    const double focal = db.st->focal[cur];
    const typename ObsXY<T>::type* oxy = reinterpret_cast<const typename ObsXY<T>::type*>(ds.obs_xy);
    (void)oxy;
    double n[7] = { 0, 0, 0, 0, 0, 0, 0 };
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
    for (int c = 0; c < 7; ++c) {
        const double s = block_sum(n[c], scratch);
        if (threadIdx.x == 0) atomicAdd(c < 6 ? &db.udiag[6 * j + c] : &db.udiag[ds.d - 1], s);
    }
}

__global__ void k_colnorm_finish(DeviceStructure ds, DeviceBuffers db, int jacobi) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= ds.d) return;
    const double s = jacobi ? 1.0 / (1.0 + sqrt(db.udiag[e])) : 1.0;
    if (e < ds.d - 1) db.cscale[e] = s; else db.st->fscale = s;
}

template <typename T>
void launch_colnorm(hipStream_t s, const DeviceStructure& ds, const DeviceBuffers& db, int jacobi) {
    (void)hipMemsetAsync(db.udiag, 0, sizeof(double) * ds.ld, s);
    hipLaunchKernelGGL(k_colnorm_points<T>, dim3((ds.npt + BLK - 1) / BLK), dim3(BLK), 0, s, ds, db, jacobi);
    if (jacobi) hipLaunchKernelGGL(k_colnorm_cams<T>, dim3(ds.nchunk), dim3(BLK), 0, s, ds, db);
    hipLaunchKernelGGL(k_colnorm_finish, dim3((ds.d + 255) / 256), dim3(256), 0, s, ds, db, jacobi);
}
template void launch_colnorm<float>(hipStream_t, const DeviceStructure&, const DeviceBuffers&, int);
template void launch_colnorm<double>(hipStream_t, const DeviceStructure&, const DeviceBuffers&, int);

// ------------------------------------------------------------------------------------------
// 3x3 SPD: L^-1 (lower, 6 values l00 l10 l11 l20 l21 l22 of the INVERSE factor). Returns false if not PD.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ bool chol3_inverse(const double V[6] /* v00 v10 v11 v20 v21 v22 */, double Li[6]) {
    const double l00 = sqrt(V[0]);
    const double l10 = V[1] / l00, l20 = V[3] / l00;
    const double d1 = V[2] - l10 * l10;
    const double l11 = sqrt(d1);
    const double l21 = (V[4] - l20 * l10) / l11;
    const double d2 = V[5] - l20 * l20 - l21 * l21;
    const double l22 = sqrt(d2);
    const double i00 = 1.0 / l00, i11 = 1.0 / l11, i22 = 1.0 / l22;
    Li[0] = i00;
    Li[1] = -l10 * i00 * i11;
    Li[2] = i11;
    Li[4] = -l21 * i11 * i22;
    Li[3] = -(l20 * i00 + l21 * Li[1]) * i22;
    Li[5] = i22;
    return (V[0] > 0.0) && (d1 > 0.0) && (d2 > 0.0);
}

template <typename T>
__device__ __forceinline__ void load_obs(const void* base, int q, double& ox, double& oy) {
    const typename ObsXY<T>::type v = reinterpret_cast<const typename ObsXY<T>::type*>(base)[q];
    ox = (double)v.x; oy = (double)v.y;
}

// ------------------------------------------------------------------------------------------
// K1: point pass.  One thread per point.
// ------------------------------------------------------------------------------------------
template <typename T, bool LDS_TAB>
__global__ __launch_bounds__(BLK) void k_point_build(DeviceStructure ds, DeviceBuffers db) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    __shared__ double scratch[BLK / 64];
    const LMState* st = db.st;
    const int cur = st->cur;
    const double* gtab = db.camtab[cur];
    const double* tab = gtab;
    if (LDS_TAB) {
        double* ltab = reinterpret_cast<double*>(smem_raw);
        for (int e = threadIdx.x; e < ds.ncam * CT_STRIDE; e += blockDim.x) ltab[e] = gtab[e];
        __syncthreads();
        tab = ltab;
    }
    const double focal = st->focal[cur];
    const T fscale = (T)st->fscale;
    const double radius = st->radius;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = i < ds.npt;

    double lin_cost = 0.0, sff = 0.0, rhsf = 0.0, gmax = 0.0, bad = 0.0;
    if (active) {
        const double* P = db.pts[cur] + 3 * (size_t)i;
        const double X[3] = { P[0], P[1], P[2] };
        const T sp[3] = { (T)db.pscale[3 * i], (T)db.pscale[3 * i + 1], (T)db.pscale[3 * i + 2] };
        const int q0 = ds.pt_ptr[i], q1 = ds.pt_ptr[i + 1];
        double V[6] = { 0, 0, 0, 0, 0, 0 }, bp[3] = { 0, 0, 0 }, Ef[3] = { 0, 0, 0 };
        for (int q = q0; q < q1; ++q) {
            const double* ct = tab + (size_t)ds.obs_cam[q] * CT_STRIDE;
            double ox, oy;
            load_obs<T>(ds.obs_xy, q, ox, oy);
            const Proj pr = project_point(ct, CT_R, CT_T, X);
            const double r0 = focal * pr.xp - ox, r1 = focal * pr.yp - oy;
            lin_cost += r0 * r0 + r1 * r1;
            T B[6];
            point_block<T>(ct, pr, focal, B);
#pragma unroll
            for (int c = 0; c < 3; ++c) { B[c] *= sp[c]; B[3 + c] *= sp[c]; }
            const T g0 = (T)pr.xp * fscale, g1 = (T)pr.yp * fscale;
            V[0] += (double)(B[0] * B[0] + B[3] * B[3]);
            V[1] += (double)(B[1] * B[0] + B[4] * B[3]);
            V[2] += (double)(B[1] * B[1] + B[4] * B[4]);
            V[3] += (double)(B[2] * B[0] + B[5] * B[3]);
            V[4] += (double)(B[2] * B[1] + B[5] * B[4]);
            V[5] += (double)(B[2] * B[2] + B[5] * B[5]);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                bp[c] += (double)B[c] * r0 + (double)B[3 + c] * r1;
                Ef[c] += (double)(B[c] * g0 + B[3 + c] * g1);
            }
        }
        // unscaled gradient of the point block (for the gradient-tolerance test)
#pragma unroll
        for (int c = 0; c < 3; ++c) gmax = fmax(gmax, fabs(bp[c] / (double)sp[c]));
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
        db.pt_t[3 * i] = t0; db.pt_t[3 * i + 1] = t1; db.pt_t[3 * i + 2] = t2;
        db.pt_yf[3 * i] = y0; db.pt_yf[3 * i + 1] = y1; db.pt_yf[3 * i + 2] = y2;
        sff = -(y0 * y0 + y1 * y1 + y2 * y2);
        rhsf = -(y0 * t0 + y1 * t1 + y2 * t2);
        if (!pd || !finite_d(t0 + t1 + t2 + y0 + y1 + y2) || !finite_d(lin_cost)) bad = 1.0;

        // second sweep: Y_a = A~_a^T (B~_a L^-T), 6x3 per observation
        const T l00 = (T)Li[0], l10 = (T)Li[1], l11 = (T)Li[2], l20 = (T)Li[3], l21 = (T)Li[4], l22 = (T)Li[5];
        T* Yout = reinterpret_cast<T*>(db.Y);
        for (int q = q0; q < q1; ++q) {
            const int j = ds.obs_cam[q];
            const double* ct = tab + (size_t)j * CT_STRIDE;
            const Proj pr = project_point(ct, CT_R, CT_T, X);
            T B[6], A[12];
            point_block<T>(ct, pr, focal, B);
            camera_block<T>(ct, pr, focal, X, B, A);
            T C[6];   // C = B~ L^-T : C[r][c] = sum_m B~[r][m] Linv[c][m]
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const T b0 = B[3 * r] * sp[0], b1 = B[3 * r + 1] * sp[1], b2 = B[3 * r + 2] * sp[2];
                C[3 * r + 0] = b0 * l00;
                C[3 * r + 1] = b0 * l10 + b1 * l11;
                C[3 * r + 2] = b0 * l20 + b1 * l21 + b2 * l22;
            }
            T rec[YREC];
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                const T s = (T)ct[CT_SCALE + a];
                const T a0 = A[a] * s, a1 = A[6 + a] * s;
#pragma unroll
                for (int c = 0; c < 3; ++c) rec[3 * a + c] = a0 * C[c] + a1 * C[3 + c];
            }
            if (sizeof(T) == 4) { rec[18] = (T)__int_as_float(j); } else { rec[18] = (T)__longlong_as_double((long long)j); }
            rec[19] = (T)0;
            T* dst = Yout + (size_t)q * YREC;
            if (sizeof(T) == 4) {
                float4* d4 = reinterpret_cast<float4*>(dst);
                const float* rf = reinterpret_cast<const float*>(rec);
#pragma unroll
                for (int v = 0; v < 5; ++v) d4[v] = make_float4(rf[4 * v], rf[4 * v + 1], rf[4 * v + 2], rf[4 * v + 3]);
            } else {
                double2* d2 = reinterpret_cast<double2*>(dst);
                const double* rd = reinterpret_cast<const double*>(rec);
#pragma unroll
                for (int v = 0; v < 10; ++v) d2[v] = make_double2(rd[2 * v], rd[2 * v + 1]);
            }
        }
    }
    // block reductions -> global accumulators
    const double c_sum = block_sum(lin_cost, scratch);
    const double f_sum = block_sum(sff, scratch);
    const double r_sum = block_sum(rhsf, scratch);
    const double b_sum = block_sum(bad, scratch);
    const double gm = wave_max(gmax);
    if ((threadIdx.x & 63) == 0) atomic_max_nonneg(&db.st->acc[ACC_GMAX], gm);
    if (threadIdx.x == 0) {
        atomicAdd(&db.st->acc[ACC_LIN_COST], c_sum);
        atomicAdd(&db.S[(size_t)(ds.d - 1) * ds.ld + (ds.d - 1)], f_sum);
        atomicAdd(&db.rhs[ds.d - 1], r_sum);
        if (b_sum != 0.0) atomicAdd(&db.st->acc[ACC_BAD_LIN], b_sum);
    }
}

// ------------------------------------------------------------------------------------------
// K2: camera pass.  One block per (camera, range of its observations, column window).
// The 6 x (6*win_cams) row block of S is accumulated in LDS with ds_add_f64, then flushed.
// ------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void load_yrec(const T* Y, int q, T rec[18], int& cam) {
    const T* src = Y + (size_t)q * YREC;
    if (sizeof(T) == 4) {
        const float4* s4 = reinterpret_cast<const float4*>(src);
        float tmp[20];
#pragma unroll
        for (int v = 0; v < 5; ++v) { const float4 x = s4[v]; tmp[4 * v] = x.x; tmp[4 * v + 1] = x.y; tmp[4 * v + 2] = x.z; tmp[4 * v + 3] = x.w; }
#pragma unroll
        for (int e = 0; e < 18; ++e) rec[e] = (T)tmp[e];
        cam = __float_as_int(tmp[18]);
    } else {
        const double2* s2 = reinterpret_cast<const double2*>(src);
        double tmp[20];
#pragma unroll
        for (int v = 0; v < 10; ++v) { const double2 x = s2[v]; tmp[2 * v] = x.x; tmp[2 * v + 1] = x.y; }
#pragma unroll
        for (int e = 0; e < 18; ++e) rec[e] = (T)tmp[e];
        cam = (int)__double_as_longlong(tmp[18]);
    }
}

#define NRED 41   // Ujj(21) Sjf(6) bc(6) rhs(6) uff bf

template <typename T>
__global__ __launch_bounds__(BLK) void k_cam_schur(DeviceStructure ds, DeviceBuffers db) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    double* acc = reinterpret_cast<double*>(smem_raw);
    const int4 ch = ds.chunks[blockIdx.x];
    const int j = ch.x;
    const int wlo = ch.w * ds.win_cams;
    const int whi = min(ds.ncam, wlo + ds.win_cams);
    const int wcols = 6 * (whi - wlo);
    double* red = acc + 6 * 6 * ds.win_cams;   // [BLK/64][NRED]
    for (int e = threadIdx.x; e < 6 * wcols; e += blockDim.x) acc[e] = 0.0;
    __syncthreads();

    const bool primary = (wlo <= j && j < whi);
    const LMState* st = db.st;
    const int cur = st->cur;
    const double* ct = db.camtab[cur] + (size_t)j * CT_STRIDE;
    const double focal = st->focal[cur];
    const double fscale = st->fscale;
    const T* Y = reinterpret_cast<const T*>(db.Y);

    double loc[NRED];
#pragma unroll
    for (int e = 0; e < NRED; ++e) loc[e] = 0.0;

    for (int e = ch.y + threadIdx.x; e < ch.z; e += blockDim.x) {
        const int q = ds.cam_obs[e];
        const int i = ds.cam_obs_pt[e];
        T Ya[18];
        int ca;
        load_yrec<T>(Y, q, Ya, ca);
        if (primary) {
            const double X[3] = { db.pts[cur][3 * i], db.pts[cur][3 * i + 1], db.pts[cur][3 * i + 2] };
            double ox, oy;
            load_obs<T>(ds.obs_xy, q, ox, oy);
            const Proj pr = project_point(ct, CT_R, CT_T, X);
            const double r0 = focal * pr.xp - ox, r1 = focal * pr.yp - oy;
            T B[6], A[12];
            point_block<T>(ct, pr, focal, B);
            camera_block<T>(ct, pr, focal, X, B, A);
#pragma unroll
            for (int a = 0; a < 6; ++a) { const T s = (T)ct[CT_SCALE + a]; A[a] *= s; A[6 + a] *= s; }
            const T g0 = (T)(pr.xp * fscale), g1 = (T)(pr.yp * fscale);
            int u = 0;
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int b = a; b < 6; ++b) loc[u++] += (double)(A[a] * A[b] + A[6 + a] * A[6 + b]);
            const double t0 = db.pt_t[3 * i], t1 = db.pt_t[3 * i + 1], t2 = db.pt_t[3 * i + 2];
            const double y0 = db.pt_yf[3 * i], y1 = db.pt_yf[3 * i + 1], y2 = db.pt_yf[3 * i + 2];
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                const double ar = (double)A[a] * r0 + (double)A[6 + a] * r1;
                const double yt = (double)Ya[3 * a] * t0 + (double)Ya[3 * a + 1] * t1 + (double)Ya[3 * a + 2] * t2;
                const double yf = (double)Ya[3 * a] * y0 + (double)Ya[3 * a + 1] * y1 + (double)Ya[3 * a + 2] * y2;
                loc[21 + a] += (double)(A[a] * g0 + A[6 + a] * g1) - yf;   // S[j,f]
                loc[27 + a] += ar;                                          // b_c (scaled gradient)
                loc[33 + a] += ar - yt;                                     // reduced rhs
            }
            loc[39] += (double)(g0 * g0 + g1 * g1);
            loc[40] += (double)g0 * r0 + (double)g1 * r1;
        }
        // partners: observations of the same point with camera slot >= j (sorted ascending inside a point)
        const int qend = ds.pt_ptr[i + 1];
        for (int qb = q; qb < qend; ++qb) {
            T Yb[18];
            int cb;
            if (qb == q) {
                cb = j;
#pragma unroll
                for (int m = 0; m < 18; ++m) Yb[m] = Ya[m];
            } else {
                load_yrec<T>(Y, qb, Yb, cb);
            }
            if (cb < wlo) continue;
            if (cb >= whi) break;
            double* dst = acc + 6 * (cb - wlo);
            const bool sym = (cb == j) && (qb != q);   // same camera observing the point twice
#pragma unroll
            for (int r = 0; r < 6; ++r) {
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    T v = Ya[3 * r] * Yb[3 * c] + Ya[3 * r + 1] * Yb[3 * c + 1] + Ya[3 * r + 2] * Yb[3 * c + 2];
                    if (sym) v += Ya[3 * c] * Yb[3 * r] + Ya[3 * c + 1] * Yb[3 * r + 1] + Ya[3 * c + 2] * Yb[3 * r + 2];
                    atomicAdd(&dst[r * wcols + c], -(double)v);
                }
            }
        }
    }

    if (primary) {
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
        for (int e = 0; e < NRED; ++e) {
            const double s = wave_sum(loc[e]);
            if (lane == 0) red[w * NRED + e] = s;
        }
    }
    __syncthreads();
    const int row0 = 6 * j;
    if (primary && threadIdx.x < NRED) {
        double s = 0.0;
        for (int w = 0; w < BLK / 64; ++w) s += red[w * NRED + threadIdx.x];
        const int e = threadIdx.x;
        const int fo = ds.d - 1;
        if (e < 21) {
            int a = 0, rem = e;
            while (rem >= 6 - a) { rem -= 6 - a; ++a; }
            const int b = a + rem;
            atomicAdd(&db.S[(size_t)(row0 + a) * ds.ld + row0 + b], s);
            if (a == b) atomicAdd(&db.udiag[row0 + a], s);
        } else if (e < 27) {
            atomicAdd(&db.S[(size_t)(row0 + e - 21) * ds.ld + fo], s);
        } else if (e < 33) {
            atomicAdd(&db.bc[row0 + e - 27], s);
        } else if (e < 39) {
            atomicAdd(&db.rhs[row0 + e - 33], s);
        } else if (e == 39) {
            atomicAdd(&db.S[(size_t)fo * ds.ld + fo], s);
            atomicAdd(&db.udiag[fo], s);
        } else {
            atomicAdd(&db.bc[fo], s);
            atomicAdd(&db.rhs[fo], s);
        }
    }
    // flush the LDS row block (upper triangle only: column >= row)
    for (int e = threadIdx.x; e < 6 * wcols; e += blockDim.x) {
        const int r = e / wcols, c = e - r * wcols;
        const int gc = 6 * wlo + c;
        const double v = acc[e];
        if (gc >= row0 + r && v != 0.0) atomicAdd(&db.S[(size_t)(row0 + r) * ds.ld + gc], v);
    }
}

static bool use_lds_table(const DeviceStructure& ds, int stride) { return (size_t)ds.ncam * stride * sizeof(double) <= 56 * 1024; }

size_t build_lds_bytes(const DeviceStructure& ds) {
    return sizeof(double) * (size_t)(36 * ds.win_cams + (BLK / 64) * NRED);
}

void launch_zero_system(hipStream_t s, const DeviceStructure& ds, const DeviceBuffers& db) {
    // S | rhs | udiag | bc are contiguous; only the upper triangle of S is ever read, but a flat memset is cheapest
    (void)hipMemsetAsync(db.S, 0, sizeof(double) * ((size_t)ds.ld * ds.ld + 3 * (size_t)ds.ld), s);
}

template <typename T>
void launch_point_build(hipStream_t s, const DeviceStructure& ds, const DeviceBuffers& db) {
    const dim3 grid((ds.npt + BLK - 1) / BLK);
    if (use_lds_table(ds, CT_STRIDE))
        hipLaunchKernelGGL((k_point_build<T, true>), grid, dim3(BLK), sizeof(double) * ds.ncam * CT_STRIDE, s, ds, db);
    else
        hipLaunchKernelGGL((k_point_build<T, false>), grid, dim3(BLK), 0, s, ds, db);
}
template void launch_point_build<float>(hipStream_t, const DeviceStructure&, const DeviceBuffers&);
template void launch_point_build<double>(hipStream_t, const DeviceStructure&, const DeviceBuffers&);

template <typename T>
void launch_cam_schur(hipStream_t s, const DeviceStructure& ds, const DeviceBuffers& db) {
    hipLaunchKernelGGL(k_cam_schur<T>, dim3(ds.nchunk), dim3(BLK), build_lds_bytes(ds), s, ds, db);
}
template void launch_cam_schur<float>(hipStream_t, const DeviceStructure&, const DeviceBuffers&);
template void launch_cam_schur<double>(hipStream_t, const DeviceStructure&, const DeviceBuffers&);

// ------------------------------------------------------------------------------------------
// finalize: damping of the reduced diagonal, camera/focal part of the gradient max-norm, padding
// ------------------------------------------------------------------------------------------
__global__ void k_finalize(DeviceStructure ds, DeviceBuffers db) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const LMState* st = db.st;
    double g = 0.0;
    if (e < ds.d) {
        const double dd = fmin(fmax(db.udiag[e], st->min_diag), st->max_diag) / st->radius;
        db.S[(size_t)e * ds.ld + e] += dd;
        const double sc = e < ds.d - 1 ? db.cscale[e] : st->fscale;
        g = fabs(db.bc[e] / sc);
        if (!finite_d(db.S[(size_t)e * ds.ld + e]) || !finite_d(db.rhs[e])) atomicAdd(&db.st->acc[ACC_BAD_LIN], 1.0);
    } else if (e < ds.ld) {
        db.S[(size_t)e * ds.ld + e] = 1.0;
        db.rhs[e] = 0.0;
    }
    g = wave_max(g);
    if ((threadIdx.x & 63) == 0 && g > 0.0) atomic_max_nonneg(&db.st->acc[ACC_GMAX], g);
}

// after a linearisation: initial cost (iteration 0), gradient tolerance, evaluation failure
__global__ void k_post_lin(DeviceStructure ds, DeviceBuffers db) {
    LMState* st = db.st;
    if (st->termination != -1) return;
    const double gmax = __longlong_as_double((long long)reinterpret_cast<unsigned long long*>(st->acc)[ACC_GMAX]);
    if (st->acc[ACC_BAD_LIN] != 0.0) {
        st->termination = SFMBA_FAILURE;
        st->message = st->iter == 0 ? MSG_INITIAL_EVAL_FAILED : MSG_EVAL_FAILED;
    }
    if (st->x_is_new) {
        st->jacobian_evals++;
        st->gmax = gmax;
        if (st->iter == 0) {
            st->cost = 0.5 * st->acc[ACC_LIN_COST];
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
    st->acc[ACC_GMAX] = 0.0;
    st->acc[ACC_BAD_LIN] = 0.0;
    st->acc[ACC_LIN_COST] = 0.0;
}

void launch_finalize(hipStream_t s, const DeviceStructure& ds, const DeviceBuffers& db) {
    hipLaunchKernelGGL(k_finalize, dim3((ds.ld + 255) / 256), dim3(256), 0, s, ds, db);
}
void launch_post_lin(hipStream_t s, const DeviceStructure& ds, const DeviceBuffers& db) {
    hipLaunchKernelGGL(k_post_lin, dim3(1), dim3(1), 0, s, ds, db);
}

// iteration 0 bookkeeping: x_norm from the accumulated ||x||^2
__global__ void k_iter0(DeviceBuffers db) {
    LMState* st = db.st;
    st->x_norm = sqrt(st->acc[ACC_XNEW2]);
    for (int e = 0; e < ACC_COUNT; ++e) st->acc[e] = 0.0;
}
void launch_iter0(hipStream_t s, const DeviceStructure& ds, const DeviceBuffers& db) {
    (void)ds;
    hipLaunchKernelGGL(k_iter0, dim3(1), dim3(1), 0, s, db);
}

// ------------------------------------------------------------------------------------------
// back-substitution + trial point
// ------------------------------------------------------------------------------------------
// One thread per camera: delta = scale * y ; trial camera = camera - delta ; step table; table of the trial camera
__global__ void k_cam_update(DeviceStructure ds, DeviceBuffers db) {
    __shared__ double scratch[BLK / 64];
    LMState* st = db.st;
    const int cur = st->cur, nxt = cur ^ 1;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    double step2 = 0.0, xn2 = 0.0;
    if (j < ds.ncam) {
        const double* ct = db.camtab[cur] + (size_t)j * CT_STRIDE;
        double dlt[6], cn[6];
        for (int e = 0; e < 6; ++e) {
            const double c0 = db.cam[cur][6 * j + e];
            dlt[e] = db.cscale[6 * j + e] * db.rhs[6 * j + e];
            cn[e] = c0 - dlt[e];
            const double df = c0 - cn[e];
            step2 += df * df;
            xn2 += cn[e] * cn[e];
            db.cam[nxt][6 * j + e] = cn[e];
        }
        double ctn[CT_STRIDE];
        make_cam_table(cn, db.cscale + 6 * j, ctn);
        double* outc = db.camtab[nxt] + (size_t)j * CT_STRIDE;
        for (int e = 0; e < CT_STRIDE; ++e) outc[e] = ctn[e];
        double* stb = db.steptab + (size_t)j * ST_STRIDE;
        for (int e = 0; e < 9; ++e) { stb[ST_R + e] = ct[CT_R + e]; stb[ST_RN + e] = ctn[CT_R + e]; }
        for (int e = 0; e < 3; ++e) { stb[ST_T + e] = ct[CT_T + e]; stb[ST_TN + e] = ctn[CT_T + e]; stb[ST_DT + e] = dlt[3 + e]; }
        if (ct[CT_SMALL] != 0.0) {
            for (int e = 0; e < 3; ++e) stb[ST_KV + e] = dlt[e];
        } else {
            for (int r = 0; r < 3; ++r)
                stb[ST_KV + r] = ct[CT_K + 3 * r] * dlt[0] + ct[CT_K + 3 * r + 1] * dlt[1] + ct[CT_K + 3 * r + 2] * dlt[2];
        }
        stb[ST_SMALL] = ct[CT_SMALL];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const double f0 = st->focal[cur];
        const double fn = f0 - st->fscale * db.rhs[ds.d - 1];
        st->focal[nxt] = fn;
        const double df = f0 - fn;
        step2 += df * df;
        xn2 += fn * fn;
    }
    const double s2 = block_sum(step2, scratch);
    const double x2 = block_sum(xn2, scratch);
    if (threadIdx.x == 0) { atomicAdd(&st->acc[ACC_STEP2], s2); atomicAdd(&st->acc[ACC_XNEW2], x2); }
}

// One thread per point: y_p = (V + D^2)^-1 (b_p - W^T y_c), trial point, model cost change, trial cost
template <typename T, bool LDS_TAB>
__global__ __launch_bounds__(BLK) void k_point_update(DeviceStructure ds, DeviceBuffers db) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    __shared__ double scratch[BLK / 64];
    const LMState* st = db.st;
    const int cur = st->cur, nxt = cur ^ 1;
    const double* gtab = db.steptab;
    const double* tab = gtab;
    if (LDS_TAB) {
        double* ltab = reinterpret_cast<double*>(smem_raw);
        for (int e = threadIdx.x; e < ds.ncam * ST_STRIDE; e += blockDim.x) ltab[e] = gtab[e];
        __syncthreads();
        tab = ltab;
    }
    const double focal = st->focal[cur], focal_n = st->focal[nxt];
    const double dfoc = focal - focal_n;           // unscaled focal step to SUBTRACT (= fscale * y_f)
    const double radius = st->radius;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double trial = 0.0, model = 0.0, step2 = 0.0, xn2 = 0.0, bad = 0.0;
    if (i < ds.npt) {
        const double* P = db.pts[cur] + 3 * (size_t)i;
        const double X[3] = { P[0], P[1], P[2] };
        const T sp[3] = { (T)db.pscale[3 * i], (T)db.pscale[3 * i + 1], (T)db.pscale[3 * i + 2] };
        const int q0 = ds.pt_ptr[i], q1 = ds.pt_ptr[i + 1];
        double V[6] = { 0, 0, 0, 0, 0, 0 }, bp[3] = { 0, 0, 0 };
        for (int q = q0; q < q1; ++q) {
            const double* stb = tab + (size_t)ds.obs_cam[q] * ST_STRIDE;
            double ox, oy;
            load_obs<T>(ds.obs_xy, q, ox, oy);
            const Proj pr = project_point(stb, ST_R, ST_T, X);
            const double r0 = focal * pr.xp - ox, r1 = focal * pr.yp - oy;
            T B[6];
            point_block<T>(stb, pr, focal, B);     // ST_R == CT_R == 0
            // u = A (scale*y_c) + g (fscale*y_f):  A dc = Aproj (G dw + dt), G dw = R_J (kv x X)
            const T kx = (T)stb[ST_KV], ky = (T)stb[ST_KV + 1], kz = (T)stb[ST_KV + 2];
            const T c0 = ky * (T)X[2] - kz * (T)X[1], c1 = kz * (T)X[0] - kx * (T)X[2], c2 = kx * (T)X[1] - ky * (T)X[0];
            T dp0, dp1, dp2;
            if (stb[ST_SMALL] != 0.0) { dp0 = c0; dp1 = c1; dp2 = c2; }
            else {
                dp0 = (T)stb[ST_R + 0] * c0 + (T)stb[ST_R + 1] * c1 + (T)stb[ST_R + 2] * c2;
                dp1 = (T)stb[ST_R + 3] * c0 + (T)stb[ST_R + 4] * c1 + (T)stb[ST_R + 5] * c2;
                dp2 = (T)stb[ST_R + 6] * c0 + (T)stb[ST_R + 7] * c1 + (T)stb[ST_R + 8] * c2;
            }
            dp0 += (T)stb[ST_DT]; dp1 += (T)stb[ST_DT + 1]; dp2 += (T)stb[ST_DT + 2];
            const T fz = (T)(focal * pr.iz), xp = (T)pr.xp, yp = (T)pr.yp;
            const T u0 = fz * (dp0 - xp * dp2) + xp * (T)dfoc;
            const T u1 = fz * (dp1 - yp * dp2) + yp * (T)dfoc;
#pragma unroll
            for (int c = 0; c < 3; ++c) { B[c] *= sp[c]; B[3 + c] *= sp[c]; }
            V[0] += (double)(B[0] * B[0] + B[3] * B[3]);
            V[1] += (double)(B[1] * B[0] + B[4] * B[3]);
            V[2] += (double)(B[1] * B[1] + B[4] * B[4]);
            V[3] += (double)(B[2] * B[0] + B[5] * B[3]);
            V[4] += (double)(B[2] * B[1] + B[5] * B[4]);
            V[5] += (double)(B[2] * B[2] + B[5] * B[5]);
#pragma unroll
            for (int c = 0; c < 3; ++c) bp[c] += (double)B[c] * (r0 - (double)u0) + (double)B[3 + c] * (r1 - (double)u1);
        }
        V[0] += fmin(fmax(V[0], st->min_diag), st->max_diag) / radius;
        V[2] += fmin(fmax(V[2], st->min_diag), st->max_diag) / radius;
        V[5] += fmin(fmax(V[5], st->min_diag), st->max_diag) / radius;
        double Li[6];
        chol3_inverse(V, Li);
        // y_p = L^-T L^-1 (b_p - c)
        const double t0 = Li[0] * bp[0];
        const double t1 = Li[1] * bp[0] + Li[2] * bp[1];
        const double t2 = Li[3] * bp[0] + Li[4] * bp[1] + Li[5] * bp[2];
        const double y2 = Li[5] * t2;
        const double y1 = Li[2] * t1 + Li[4] * t2;
        const double y0 = Li[0] * t0 + Li[1] * t1 + Li[3] * t2;
        const double dX[3] = { (double)sp[0] * y0, (double)sp[1] * y1, (double)sp[2] * y2 };
        double Xn[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            Xn[c] = X[c] - dX[c];
            const double df = X[c] - Xn[c];
            step2 += df * df;
            xn2 += Xn[c] * Xn[c];
            db.pts[nxt][3 * (size_t)i + c] = Xn[c];
        }
        for (int q = q0; q < q1; ++q) {
            const double* stb = tab + (size_t)ds.obs_cam[q] * ST_STRIDE;
            double ox, oy;
            load_obs<T>(ds.obs_xy, q, ox, oy);
            const Proj pr = project_point(stb, ST_R, ST_T, X);
            const double r0 = focal * pr.xp - ox, r1 = focal * pr.yp - oy;
            // model residual m = J step = -(u + B dX), all unscaled quantities; dpt = total unscaled change of p
            const double kx = stb[ST_KV], ky = stb[ST_KV + 1], kz = stb[ST_KV + 2];
            const double c0 = ky * X[2] - kz * X[1], c1 = kz * X[0] - kx * X[2], c2 = kx * X[1] - ky * X[0];
            double dp0, dp1, dp2;
            if (stb[ST_SMALL] != 0.0) { dp0 = c0; dp1 = c1; dp2 = c2; }
            else {
                dp0 = stb[ST_R + 0] * c0 + stb[ST_R + 1] * c1 + stb[ST_R + 2] * c2;
                dp1 = stb[ST_R + 3] * c0 + stb[ST_R + 4] * c1 + stb[ST_R + 5] * c2;
                dp2 = stb[ST_R + 6] * c0 + stb[ST_R + 7] * c1 + stb[ST_R + 8] * c2;
            }
            dp0 += stb[ST_DT] + stb[ST_R + 0] * dX[0] + stb[ST_R + 1] * dX[1] + stb[ST_R + 2] * dX[2];
            dp1 += stb[ST_DT + 1] + stb[ST_R + 3] * dX[0] + stb[ST_R + 4] * dX[1] + stb[ST_R + 5] * dX[2];
            dp2 += stb[ST_DT + 2] + stb[ST_R + 6] * dX[0] + stb[ST_R + 7] * dX[1] + stb[ST_R + 8] * dX[2];
            const double fz = focal * pr.iz;
            const double m0 = -(fz * (dp0 - pr.xp * dp2) + pr.xp * dfoc);
            const double m1 = -(fz * (dp1 - pr.yp * dp2) + pr.yp * dfoc);
            model -= m0 * (r0 + 0.5 * m0) + m1 * (r1 + 0.5 * m1);
            // trial residual
            const Proj pn = project_point(stb, ST_RN, ST_TN, Xn);
            const double n0 = focal_n * pn.xp - ox, n1 = focal_n * pn.yp - oy;
            if (!finite_d(n0) || !finite_d(n1)) bad = 1.0;
            trial += n0 * n0 + n1 * n1;
        }
    }
    const double a = block_sum(trial, scratch);
    const double b = block_sum(model, scratch);
    const double c = block_sum(step2, scratch);
    const double d = block_sum(xn2, scratch);
    const double e = block_sum(bad, scratch);
    if (threadIdx.x == 0) {
        atomicAdd(&db.st->acc[ACC_TRIAL_COST], a);
        atomicAdd(&db.st->acc[ACC_MODEL], b);
        atomicAdd(&db.st->acc[ACC_STEP2], c);
        atomicAdd(&db.st->acc[ACC_XNEW2], d);
        if (e != 0.0) atomicAdd(&db.st->acc[ACC_BAD_TRIAL], e);
    }
}

void launch_cam_update(hipStream_t s, const DeviceStructure& ds, const DeviceBuffers& db) {
    hipLaunchKernelGGL(k_cam_update, dim3((ds.ncam + BLK - 1) / BLK), dim3(BLK), 0, s, ds, db);
}

template <typename T>
void launch_point_update(hipStream_t s, const DeviceStructure& ds, const DeviceBuffers& db) {
    const dim3 grid((ds.npt + BLK - 1) / BLK);
    if (use_lds_table(ds, ST_STRIDE))
        hipLaunchKernelGGL((k_point_update<T, true>), grid, dim3(BLK), sizeof(double) * ds.ncam * ST_STRIDE, s, ds, db);
    else
        hipLaunchKernelGGL((k_point_update<T, false>), grid, dim3(BLK), 0, s, ds, db);
}
template void launch_point_update<float>(hipStream_t, const DeviceStructure&, const DeviceBuffers&);
template void launch_point_update<double>(hipStream_t, const DeviceStructure&, const DeviceBuffers&);

// ------------------------------------------------------------------------------------------
// LM control: the accept/reject logic of ceres::internal::TrustRegionMinimizer::Minimize()
// [Ceres-upstream], one thread.
// ------------------------------------------------------------------------------------------
__global__ void k_lm_control(DeviceBuffers db) {
    LMState* st = db.st;
    if (st->termination != -1) return;
    const int it = ++st->iter;
    TraceRow row = {};
    row.iteration = it;
    const double model = st->acc[ACC_MODEL];
    const double step2 = st->acc[ACC_STEP2];
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
            st->radius *= 0.5;
            st->unsuccessful++;
        }
    } else {
        st->consecutive_invalid = 0;
        double cand = 0.5 * st->acc[ACC_TRIAL_COST];
        if (st->acc[ACC_BAD_TRIAL] != 0.0 || !finite_d(cand)) cand = DBL_MAX;
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
                    st->x_norm = sqrt(st->acc[ACC_XNEW2]);
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
    st->acc[ACC_TRIAL_COST] = 0.0;
    st->acc[ACC_MODEL] = 0.0;
    st->acc[ACC_STEP2] = 0.0;
    st->acc[ACC_XNEW2] = 0.0;
    st->acc[ACC_BAD_TRIAL] = 0.0;
    st->lin_info = 0;
}

void launch_control(hipStream_t s, const DeviceStructure& ds, const DeviceBuffers& db) {
    (void)ds;
    hipLaunchKernelGGL(k_lm_control, dim3(1), dim3(1), 0, s, db);
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
        const double* ct = db.camtab[cur] + (size_t)ds.obs_cam[q] * CT_STRIDE;
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
    const double* ct = db.camtab[cur] + (size_t)ds.obs_cam[q] * CT_STRIDE;
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
